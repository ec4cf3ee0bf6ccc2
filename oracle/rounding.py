"""ctypes wrapper of oracle/rounding.c (TEST INFRASTRUCTURE; built by __graft_entry__.build())."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "librounding.so")


def build():
    src = os.path.join(HERE, "rounding.c")
    if os.path.exists(SO) and os.path.getmtime(SO) >= os.path.getmtime(src):
        return SO
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", SO, src, "-lm"])
    return SO


def rounding_ref(x: np.ndarray, W: np.ndarray, tgt: np.ndarray | None = None, want_logits=False):
    build()
    L = C.CDLL(SO)
    x = np.ascontiguousarray(x, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    M, K = x.shape
    V = W.shape[0]
    am = np.zeros(M, np.int64)
    mx = np.zeros(M, np.float32)
    lse = np.zeros(M, np.float64)
    tl = np.zeros(M, np.float32)
    logits = np.zeros((M, V), np.float32) if want_logits else None
    t = np.ascontiguousarray(tgt, np.int64) if tgt is not None else None
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    L.rounding_ref(p(x), p(W), M, V, K, p(t), p(am), p(mx), p(lse), p(tl), p(logits))
    return dict(argmax=am, maxlogit=mx, lse=lse, tgt_logit=tl, logits=logits)
