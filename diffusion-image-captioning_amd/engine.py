"""`DistilBertModel` drop-in (ref CLIP-DDPM.py:227-323) driven entirely by the HIP library.

There is no autograd here: `encode()` keeps the activations the backward kernels need in a pre-allocated
workspace and `backward()` walks the layers in reverse, launching the hand-written HIP kernels through the C-ABI
(`include/dic_hip.h`).  torch is used for device memory, the current stream and (in parallel.py) RCCL only.

Batching trick: the reference runs up to three encoder passes per training step (x_t rows, the classifier-free-
guidance subset again with the text key unmasked, and the x_1 rows; ref :312, :316, :426).  The encoder is
row-independent, so this engine stacks them into ONE batch with a per-sequence key mask; every weight then sees
one forward GEMM, one dX GEMM and one dW GEMM per step, which is what fills 256 CUs at small B.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import (DIC_BF16, DIC_F32, EPI_AFFINE, EPI_BIAS_GELU, EPI_BIAS_GELU_D, EPI_CE_DLOGITS, EPI_CE_EXP, EPI_CE_PARTIAL, EPI_GELU_BWD, EPI_MUL_AUX,
                   GemmParams)
from .config import LOSS_KINDS, cfg
from .params import ParamStore

from .options import OPT

LN_EPS = 1e-12
NPART = OPT.ln_npart          # persistent blocks (= partial rows) of the LayerNorm backward kernels (2 per CU)


def _p(t):
    return 0 if t is None else t.data_ptr()


class Ops:
    """Thin callers of the C-ABI on the current torch stream."""

    def __init__(self, dtype_flag: int):
        self.L = _lib.lib()
        self.dt = dtype_flag
        self._gp = GemmParams()
        self.stream = None
        self.default_cu_cap = 0          # >0: bf16 GEMMs issued without an explicit cu_cap keep to this many CUs (set by parallel.GradReducer
                                         # while a gradient slice is on the wire, DIC_DP_CU_CAP)

    def begin(self):
        self.stream = torch.cuda.current_stream().cuda_stream

    def gemm(self, A, B, Cc, M, N, K, lda, ldb, ldc, a_km=0, b_km=0, epi=EPI_AFFINE, bias=0, R=0, ldr=0, aux=0,
             ldaux=0, p_drop=0.0, seed=0, out_f32=0, accumulate=0, tgt=0, lse=0, partial=0, tgt_logit=0,
             ce_rows_a=0, ce_scale_a=0.0, ce_scale_b=0.0, dtype=None, split_k=1, split_ws=0, colsum_out=0, tile=None, cu_cap=0, B2=0, b2_col0=0, bias2=0):
        g = self._gp
        g.A, g.B, g.C = A, B, Cc
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, K, lda, ldb, ldc
        g.bias, g.R, g.ldr, g.aux, g.ldaux = bias, R, ldr, aux, ldaux
        g.p_drop, g.seed, g.out_f32, g.accumulate = p_drop, seed, out_f32, accumulate
        g.tgt, g.lse, g.partial, g.tgt_logit = tgt, lse, partial, tgt_logit
        g.ce_rows_a, g.ce_scale_a, g.ce_scale_b = ce_rows_a, ce_scale_a, ce_scale_b
        g.split_k, g.split_ws, g.colsum_out, g.B2, g.b2_col0, g.bias2 = split_k, split_ws, colsum_out, B2, b2_col0, bias2
        dt = self.dt if dtype is None else dtype
        if tile is None:
            tile = choose_tile(M, N, split_k, epi) if (dt == DIC_BF16 and epi != EPI_CE_PARTIAL and (not a_km or M % 256 == 0)) else 128
        g.tile = tile
        g.cu_cap = cu_cap if cu_cap else self.default_cu_cap
        _lib.check(self.L.dic_gemm(self.dt if dtype is None else dtype, a_km, b_km, epi, C.byref(g), self.stream), "gemm")


import math

# Every switch lives in options.py (one record, shipped values = its defaults); what remains here are names for the hot ones.
DIC_U_F32, DIC_RES_F32, OUT_F32_RES_F32 = 0x100, 0x200, 3
N_CU = 256


def choose_tile(M, N, split_k=1, epi=EPI_AFFINE):
    """256x256 workgroup tiles (8 waves, one workgroup per CU) pull half the bytes per flop through L2 of the 128x128 ones
    (measured: 1.16-1.26 vs 0.83-0.95 PFLOP/s on large squares), but there are 4x fewer of them and their epilogue is not hidden
    behind a second resident workgroup: use them when they still give every CU work.  The GELU' epilogue of the fp32 engine (one extra
    113 MB input) measured faster on 128-tiles (145 vs 154 us); bias+GELU (two 113 MB outputs) and the multiply epilogue on 256-tiles."""
    if OPT.gemm_tile in ("128", "256"):
        return int(OPT.gemm_tile) if (M >= 256 and N >= 256) else 128
    if M < 256 or N < 256 or epi == EPI_GELU_BWD:
        return 128
    units = ((M + 255) // 256) * ((N + 255) // 256) * max(split_k, 1)
    return 256 if units >= int(0.75 * N_CU) else 128


def pick_split_k(M, N, K, bk=64, max_split=32):
    """dW GEMMs have few output tiles (768x768 -> 36) but a long contraction (all tokens): cut K so that one round of resident
    workgroups covers the launch (2 per CU for 128-tiles, 1 per CU for 256-tiles), each slice keeping >= 8 K-steps.
    Measured at K = 18432: 768x768 -> 128-tiles x14 (468 TF) beats 256-tiles x28 (404); 3072x768 -> 256-tiles x7 (836 TF)
    beats 128-tiles x3 (781); 2304x768 -> 256-tiles x9 (736) ~ 128-tiles x4 (731).  Returns (split_k, tile)."""
    nk = (K + bk - 1) // bk
    big = OPT.gemm_tile != "128" and M % 256 == 0 and N >= 256 and M * N >= 2304 * 768
    if OPT.gemm_tile == "256" and M % 256 == 0 and N >= 256:
        big = True
    tile, resident = (256, N_CU) if big else (128, 2 * N_CU)
    tiles = ((M + tile - 1) // tile) * ((N + tile - 1) // tile)
    return max(1, min(max_split, resident // max(tiles, 1), nk // 8)), tile


class Denoiser:
    """Same constructor shape as the reference: `DistilBertModel(embedding, projection, config=...)`.

    embedding / projection: objects with `.weight` ([V,768]; nn.Embedding / nn.Linear work) or raw arrays/tensors.
    The projection bias is zeroed as ref :247 does.  `config` may be a HF DistilBertConfig-like object or a dict with
    `n_layers`, `dropout`, `attention_dropout`.
    dtype (keyword-only; the reference is fp32):
      "fp32"  exact-fp32 MFMA GEMMs: token ids identical to the CPU reference's, losses to 1e-6;
      "bf16"  (default; torch.bfloat16 and the older name "bf16m" mean the same) bf16 MFMA operands and activations, fp32 master weights --
              evaluated so that no bf16 rounding is COMMON to all token rows: every forward Linear adds back the row-common part of the
              weights' rounding (mean input row x lo weight half, dic_lin_prep), the residual stream is stored centred on predicted mean rows
              (dic_ln_fwd_cen), the MLM-head pre-activation stays fp32, the rounding head sees mean-centred rows.  Every loss term within
              1e-4 of fp32 at the initial weights, at trained weights and along a run, ~4 % slower than "bf16r";
      "bf16r" the raw bf16 engine without those corrections: the fastest, 1-3e-4 from fp32 (the A/B partner; what rounds 1-4 benchmarked);
      "bf16w" the exact form of "bf16": the lo weight halves as a second pass of every forward GEMM's K loop + an fp32 residual stream (the
              reference the mean-row form is checked against; ~25 % slower).
    An unknown value raises ValueError.
    """

    def __init__(self, embedding=None, projection=None, config=None, *, dtype="bf16", device="cuda:0", seed=0, split_weights=None):
        if dtype not in ("fp32", "bf16", "bf16r", "bf16w", "bf16m", torch.float32, torch.bfloat16):
            raise ValueError(f"dtype must be 'fp32', 'bf16' (= 'bf16m'), 'bf16r' or 'bf16w', not {dtype!r}")
        _lib.require_gpu()
        get = (lambda k, d: config.get(k, d)) if isinstance(config, dict) else (lambda k, d: getattr(config, k, d))
        self.n_layers = int(get("n_layers", 6)) if config is not None else 6
        self.p_hidden = float(get("dropout", 0.1)) if config is not None else 0.1
        self.p_attn = float(get("attention_dropout", 0.1)) if config is not None else 0.1
        self.n_heads, self.dim, self.hidden = 12, 768, 3072
        self.device = torch.device(device)
        self.bf16 = dtype in ("bf16", "bf16r", "bf16w", "bf16m", torch.bfloat16)
        # SPLIT WEIGHTS (dtype="bf16w" / split_weights=True / DIC_SPLIT_W=1; bf16 engine only): the forward Linears multiply by hi + lo bf16 halves
        # of the fp32 master weights (two passes of the GEMM's K loop, include/dic_hip.h DicGemmParams.B2) instead of by their bf16 rounding.
        # This is the fast mode that meets north_star's 1e-4 loss tolerance: the weights' rounding error is the same for every sample and does
        # not average out of a batch-mean loss, the activations' does (profiles/r04_weight_rounding_probe.txt).  The backward is unchanged.
        if split_weights is None:
            split_weights = dtype in ("bf16", "bf16m", "bf16w", torch.bfloat16)
        self.split_w = bool(split_weights) and self.bf16
        # HOW the lo halves enter (DIC_LO_MODE): "pass2" = a second pass of the K loop (DicGemmParams.B2, dtype="bf16w"); "mean" = only their
        # row-common part, mean row of the Linear's input times the lo half, added to the bias (dic_lo_mean_bias, dtype="bf16m"): two small
        # launches per Linear instead of doubling its flops
        self.lo_mode = ("pass2" if dtype == "bf16w" else "mean") if self.split_w else None
        self.lo_row_stride = OPT.lo_row_stride      # rows sampled for the mean row: every 16th
        # WHICH forward Linears take the lo half (DIC_SPLIT_SET; profiles/r04_split_alloc_trajectory_dense.txt: 19 states along a training run):
        # "all" (default): every Linear; "vo2t" = only the value third of q|k|v, the attention output projection, FFN lin2 and the MLM-head
        # transform -- at B = 512 FFN lin1 and the query / key projections make no measurable difference to any loss term at any state (54 % of the
        # second-pass flops), at small batches they do, hence not the default.
        # split_slots (slot -> bool) overrides the set, split_qk overrides the q / k choice (the probes use both).
        # "auto" (round 5): the exact form bf16w corrects every Linear; the mean-row form bf16m leaves out FFN lin1 -- its correction never moved a
        # loss term at any state (round 4, B = 512; 7e-5 at B = 16 with q / k left out as well), and it is one launch per layer on the critical path
        self.split_set = OPT.split_set if OPT.split_set != "auto" else ("vo2t" if self.lo_mode == "mean" else "all")
        self.split_slots = None
        self.split_qk = None
        self.dt = DIC_BF16 if self.bf16 else DIC_F32
        # bf16 engines keep the MLM-head pre-activation (vocab_transform's output) in fp32: include/dic_hip.h, DIC_U_F32 (DIC_UVT32=0: A/B)
        self.uvt32 = self.bf16 and OPT.uvt32
        self.dt_u = (self.dt | DIC_U_F32) if self.uvt32 else self.dt
        # fp32 residual stream: the pre-LayerNorm sums and the residual operands of the two residual GEMMs of a block in fp32, bf16 MFMA operands
        # CENTRED residual stream (round 5, the parity mode dtype="bf16m"): the same roundings removed at bf16 bytes -- the pre-LayerNorm sums and the
        # residual operands are stored as bf16(value - reference row), one fp32 reference row per tensor predicted by dic_lin_prep
        # (include/dic_hip.h; DIC_CEN=0: the round-4 form of bf16m, fp32 copies)
        self.cen = self.split_w and self.lo_mode == "mean" and OPT.cen
        self.res32 = self.bf16 and not self.cen and (OPT.res32 == "1" or (OPT.res32 == "auto" and self.split_w))
        self.dt_ln = (self.dt | DIC_RES_F32) if self.res32 else self.dt
        self.tdtype = torch.bfloat16 if self.bf16 else torch.float32
        self.es = 2 if self.bf16 else 4
        self.concat = cfg.CLIP_ADDING_METHOD == "concat"
        if cfg.CLIP_ADDING_METHOD not in ("concat", "add"):
            raise NotImplementedError(cfg.CLIP_ADDING_METHOD)
        self.training = True
        self.ops = Ops(self.dt)
        # TRAIN_EMBEDDING ablation (ref :98-102, 238-243): learned 16-d embedding / head + projections, see train_embedding.py
        self.te = bool(cfg.TRAIN_EMBEDDING)
        te_kw = dict(train_embedding_vocab=int(cfg.VOCAB_SIZE), in_channel=int(cfg.IN_CHANNEL)) if self.te else {}
        self.temb = bool(cfg.TIMESTEP_EMBEDDING)
        if self.temb and self.te:
            raise NotImplementedError("cfg.TIMESTEP_EMBEDDING is not wired into the TRAIN_EMBEDDING ablation (its loss / sampling paths pass no "
                                      "timestep): the table would silently stay untrained")
        if self.temb:
            te_kw["timestep_embedding"] = int(cfg.STEP_TOT)
        self.params = ParamStore(self.n_layers, self.device, concat=self.concat, bf16_shadow=self.bf16, split_shadow=self.split_w, **te_kw)
        self.params.init_like_reference(seed)
        if self.te:
            assert cfg.IN_CHANNEL % 4 == 0 and cfg.IN_CHANNEL <= 32, "TRAIN_EMBEDDING: IN_CHANNEL must be a multiple of 4, at most 32"
            self.vocab = int(cfg.VOCAB_SIZE)
            self.vpad = (self.vocab + 127) // 128 * 128
            self.E = self.W_lm = self.W_lm_c = None
            self._te_pending = None
        else:
            self._set_embedding(embedding, projection)
        self.refresh_shadows()
        self._ws = {}            # encoder workspaces, keyed by capacity (a few; least recently created is dropped first)
        self._ce_ws = {}         # rounding-head workspaces, keyed by row count (never dropped with the encoder's)
        self._te_ws = {}         # TRAIN_EMBEDDING scratch (train_embedding._buffers)
        self.dropout_seed_base = 0x5EED0000 + seed
        self._seed = self.dropout_seed_base
        self._saved = None
        self._pending = False
        self._side = None
        self._id_err = None
        self.wgrad_stream_enabled = True
        self.rank_rows_forced = True     # CFG forces rows 0/1 to unguided/guided (ref :408-409); DP: rank 0 only

    # ------------------------------------------------------------------ frozen embedding / rounding head
    def _set_embedding(self, embedding, projection):
        def weight_of(x, default_seed):
            if x is None:
                from . import synth
                return torch.from_numpy(synth.vocab_embedding(cfg.VOCAB_SIZE, 768, default_seed))
            w = getattr(x, "weight", x)
            w = torch.from_numpy(np.ascontiguousarray(w)) if isinstance(w, np.ndarray) else w
            return w.detach()
        E = weight_of(embedding, 0).to(self.device, torch.float32).contiguous()
        W = E if projection is None else weight_of(projection, 0).to(self.device, torch.float32).contiguous()
        self.E = E
        self.vocab = E.shape[0]
        self.vpad = (self.vocab + 127) // 128 * 128
        # rounding-head operand, rows zero-padded to a tile multiple so it can be read k-major in the dX GEMM
        self.W_lm = torch.zeros(self.vpad, 768, dtype=torch.float32, device=self.device)
        self.W_lm[:self.vocab].copy_(W)
        self.W_lm_c = self.W_lm.to(torch.bfloat16).contiguous() if self.bf16 else self.W_lm

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def set_dropout_seed(self, seed: int):
        """(Re)start the dropout-mask stream.  Data parallel, every rank gets its own (parallel.configure_model_for_rank): the masks are
        keyed by (seed, element index), so ranks sharing a seed would apply identical masks to their shards."""
        self._seed = int(seed) & 0x7FFFFFFFFFFFFFFF

    def refresh_shadows(self):
        """bf16 copies of the parameters for the MFMA operands (dic_adamw keeps them fresh itself)."""
        if self.bf16:
            self.ops.begin()
            _lib.check(self.ops.L.dic_cast_bf16_hl(_p(self.params.P), _p(self.params.Pb), _p(self.params.Pl), self.params.numel, self.ops.stream), "cast")

    # ------------------------------------------------------------------ reference-shaped API
    def parameters(self):
        return self.params.parameters()

    def named_parameters(self):
        return self.params.named_parameters()

    def load_state(self, state):
        self.params.load_state(state)
        self.refresh_shadows()

    def state_dict(self):
        """Trainable tensors under the reference's parameter names (`nn.Module.state_dict` for the parts the step trains)."""
        return self.params.state_dict()

    def load_state_dict(self, state, strict=True):
        missing = [n for n in self.params.names if n not in state]
        if missing and strict:
            raise KeyError(f"load_state_dict: missing {missing[:3]}{'...' if len(missing) > 3 else ''}")
        self.load_state(state)

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    def check_ids(self, sync=False):
        """Raise IndexError if an embedding() call met a token id outside [0, vocab) (nn.Embedding's behaviour; the kernel zero-fills the
        row and reports through a device-visible pinned word).  Without `sync` this looks at what has reached the host so far --
        embedding() calls it before every lookup, so a bad batch surfaces at the next step at the latest; sync=True waits for the stream."""
        if self._id_err is None:
            return
        if sync:
            torch.cuda.current_stream().synchronize()
        n = int(self._id_err[0])
        if n:
            pos = int(self._id_err[1]) - 1
            self._id_err.zero_()
            raise IndexError(f"embedding: {n} token id(s) outside [0, {self.vocab}) (last at flat position {pos})")

    def embedding(self, ids):
        """ref :459 -- nn.Embedding lookup -> fp32 [..., IN_CHANNEL] (frozen 768-d table, or the learned 16-d one)."""
        if self._id_err is None:
            self._id_err = torch.zeros(2, dtype=torch.int32).pin_memory()
        self.check_ids()
        ids = ids.to(self.device, torch.int64).contiguous()
        d = self.params.in_channel if self.te else 768
        table = self.params.slot_view(self.params.P, "E16") if self.te else self.E
        out = torch.empty(*ids.shape, d, dtype=torch.float32, device=self.device)
        self.ops.begin()
        _lib.check(self.ops.L.dic_embed_gather(_p(ids), _p(table), _p(out), ids.numel(), d, self.vocab, self._id_err.data_ptr(), self.ops.stream), "embed")
        return out

    def lm_head(self, h):
        """ref :323 -- logits = h @ W^T (bias is zero); materialises [.., V] fp32 (API use; training never does)."""
        if self.te:
            from . import train_embedding
            return train_embedding.lm_head(self, h)
        shp = h.shape[:-1]
        x = h.reshape(-1, 768).to(self.device, self.tdtype).contiguous()
        M = x.shape[0]
        out = torch.empty(M, self.vocab + (-self.vocab) % 4, dtype=torch.float32, device=self.device)
        self.ops.begin()
        self.ops.gemm(_p(x), _p(self.W_lm_c), _p(out), M, out.shape[1], 768, 768, 768, out.shape[1], out_f32=1)
        return out[:, :self.vocab].reshape(*shp, self.vocab)

    # ------------------------------------------------------------------ workspace
    @staticmethod
    def _evict(cache, keep):
        """Drop the oldest workspaces beyond `keep` -- never one a captured hipGraph points into (graph.GraphedTrainStep pins them)."""
        for k in [k for k, w in cache.items() if not w.get("pinned")]:
            if len(cache) <= keep:
                break
            cache.pop(k)

    def _workspace(self, N, L, drop_txt=False, cap=None):
        """Activations + backward scratch for a stacked batch of N sequences.  Buffers are sized for `cap` >= N sequences and cached by
        capacity: with classifier-free guidance N = S*B + (number of guided rows) + B changes every step, and a cache keyed by N
        would allocate a fresh multi-GB workspace per step -- the caller passes the worst case instead and the kernels get the live N."""
        Tk = (L + 1 if drop_txt else L + 2) if self.concat else L
        cap = N if cap is None else max(int(cap), N)
        key = (cap, L, Tk)
        ws = self._ws.get(key)
        if ws is not None:
            ws["N"], ws["T"] = N, N * Tk
            return ws
        self._evict(self._ws, 2)
        live_N, N = N, cap
        T, D, Hd, dev, td = N * Tk, self.dim, self.hidden, self.device, self.tdtype
        e = lambda *s, dtype=td: torch.empty(*s, dtype=dtype, device=dev)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        ws = dict(N=N, L=L, Tk=Tk, T=T, mode=(2 if drop_txt else 0) if self.concat else 1)
        ws["img_in"], ws["txt_in"] = f(N, 512), f(N, 512)
        ws["img_p"], ws["txt_p"] = f(N, D), f(N, D)
        ws["xin"] = f(N, L, D)
        ws["kmask"] = torch.empty(N, Tk, dtype=torch.uint8, device=dev)
        ws["addtxt"] = torch.zeros(N, dtype=torch.uint8, device=dev)
        ws["tidx"] = torch.full((N,), -1, dtype=torch.int32, device=dev) if self.temb else None
        ws["h"] = [e(T, D) for _ in range(self.n_layers + 1)]
        ws["mean0"], ws["rstd0"] = f(T), f(T)
        ey = f if self.res32 else e                      # fp32 residual stream: pre-LayerNorm sums in fp32 + fp32 copies of the LayerNorm outputs
        ws["layers"] = [dict(qkv=e(T, 3 * D), ctx=e(T, D), y1=ey(T, D), m1=f(T), r1=f(T), sa=e(T, D), u=e(T, Hd), g=e(T, Hd),
                             y2=ey(T, D), m2=f(T), r2=f(T)) for _ in range(self.n_layers)]
        if self.lo_mode == "mean":
            ws["beff"] = f(self.n_layers * (10 * D + 2 * Hd) + 4 * D)       # effective biases of the forward Linears (bias + mean-row lo correction, LayerNorm tails)
            ws["lomean_ws"] = f(64 * Hd)
        if self.cen:
            # centred residual stream: reference rows [layer][y1 | sa | y2 | h_next][768] (+ a zero row: the embedding LayerNorm's output is stored
            # as it is), the bias rows that go behind FFN-2's dropout, the centred residual copies of sa / h
            ws["refs"] = torch.zeros(self.n_layers + 1, 4, D, dtype=torch.float32, device=dev)
            ws["bpost"] = f(self.n_layers, 2, D)
            if not OPT.cen_operand:                   # A/B form: an uncentred operand copy next to the centred residual copy (one more 27 MB write per LayerNorm)
                ws["hc"] = [None] + [e(T, D) for _ in range(self.n_layers - 1)]
                for Lw in ws["layers"]:
                    Lw["sac"] = e(T, D)
        if self.res32:
            ws["h32"] = [f(T, D) for _ in range(self.n_layers)]          # residual operand of layer i's out-proj (the last LayerNorm's output has no reader)
            for Lw in ws["layers"]:
                Lw["sa32"] = f(T, D)
        ws["uvt"], ws["mv"], ws["rv"] = (f(T, D) if self.uvt32 else e(T, D)), f(T), f(T)
        ws["x_out"] = f(N, Tk, D)
        # backward scratch (shared by all layers)
        ws["dx_out"] = f(N, Tk, D)
        ws["dHa"], ws["dHb"] = e(T, D), e(T, D)
        # gradients that a weight-gradient GEMM consumes exist twice (layer parity): those GEMMs run on a second stream and may
        # still be reading layer i+1's copy while the main stream produces layer i's
        npar = OPT.n_bwd_sets           # sets of the gradient buffers the weight-gradient stream reads: layer i reuses the set of layer i + npar
        ws["dy"], ws["dyd"], ws["dy1"] = [e(T, D) for _ in range(npar)], [e(T, D) for _ in range(npar)], [e(T, D) for _ in range(npar)]
        ws["du"], ws["dqkv"] = [e(T, Hd) for _ in range(npar)], [e(T, 3 * D) for _ in range(npar)]
        ws["dsa"], ws["dctx"] = e(T, D), e(T, D)
        ws["dy0"] = f(N, Tk, D)
        ws["partial"] = [f(NPART, 3 * D) for _ in range(2 * npar + 1)]       # [layer parity][which LayerNorm] (folded on the side stream) + embeddings LN
        ws["cs_ws"] = f(64 * max(Tk * D, Hd))
        ws["dimg"], ws["dtxt"] = f(N, D), f(N, D)
        ws["splitk_tail"] = f(8 * (768 * 512 + 768))   # fp32 CLIP-projection weight gradients (main stream, while the side stream owns "splitk")
        ws["splitk"] = f(64 * 1024 * 1024)          # 256 MB: split_k * M * N fp32 partial tiles of one dW GEMM / the rounding dX GEMM
        ws["cap"], ws["N"], ws["T"] = cap, live_N, live_N * Tk
        self._ws[key] = ws
        return ws

    def _ce_workspace(self, M):
        ws = self._ce_ws.get(M)
        if ws is None:
            self._evict(self._ce_ws, 3)
            dev = self.device
            np_ = max(self.ops.L.dic_ce_n_partials(self.vocab, 128), self.ops.L.dic_ce_n_partials(self.vocab, 256))   # either tile size
            ws = dict(M=M, np=np_, xr=torch.empty(M, 768, dtype=self.tdtype, device=dev),
                      partial=torch.empty(M, np_, 4, dtype=torch.float32, device=dev),
                      tgt_logit=torch.zeros(M, dtype=torch.float32, device=dev), lse=torch.empty(M, dtype=torch.float32, device=dev),
                      argmax=torch.empty(M, dtype=torch.int64, device=dev), nll=torch.empty(M, dtype=torch.float32, device=dev),
                      tgt=torch.empty(M, dtype=torch.int64, device=dev), dxr=torch.empty(M, 768, dtype=torch.float32, device=dev),
                      dlogits=None, cref=torch.empty(M, dtype=torch.float32, device=dev), inv_z=torch.empty(M, dtype=torch.float32, device=dev),
                      fused=False, centered=False,
                      # mean-centred head input (center_head_input): mean row, its logits (the head GEMM's bias), partial column sums
                      xbar=torch.zeros(768, dtype=torch.float32, device=dev), cvec=torch.zeros(self.vpad + 256, dtype=torch.float32, device=dev),
                      hc_ws=torch.empty(self.ops.L.dic_head_center_ws_bytes(768) // 4, dtype=torch.float32, device=dev))
            self._ce_ws[M] = ws
        return ws

    # ------------------------------------------------------------------ encoder forward (hf:92-118, 150-259, 501-513)
    def encode(self, x, image_clip, text_clip, key_mask, add_txt=None, drop_txt=False, cap=None, tidx=None, x_view=None, inputs_ready=False, raw=False):
        """x [N,L,768] fp32; image_clip/text_clip [N,512]; key_mask [N,Tk] uint8 -> x_out [N,Tk,768] fp32.
        Saves what backward() needs.  Dropout (hidden p, attention p) is active iff self.training.
        drop_txt (concat fusion, no guided row in the batch): run with Tk = L+1, leaving the never-read text row out.
        x_view = (data_ptr, elements between sequences, N, L): read the L input rows of every sequence in place from a larger fp32 tensor
        (the sampling loop's feedback of x_out[:, :L], ref :613-620) instead of from `x`; inputs_ready: the workspace's CLIP rows, key mask
        and CLIP projections are those of the previous call (constant over a sampling loop) -- neither copied nor recomputed.
        raw: run this pass as the "bf16r" engine would (no mean-row corrections, plain bf16 residual stream) whatever the engine's dtype --
        forward-only use (sample(): per-row argmax, no batch mean whose common-mode rounding would need protecting)."""
        if x_view is not None:
            x_ptr, x_stride, N, L = x_view
        else:
            N, L, _ = x.shape
            x_ptr, x_stride = None, L * self.dim
        ws = self._workspace(N, L, drop_txt, cap)
        Tk, T, D, Hd = ws["Tk"], ws["T"], self.dim, self.hidden
        o, P, lib = self.ops, self.params, self.ops.L
        o.begin()
        st = o.stream
        wsrc = "Pb" if self.bf16 else "P"
        ph = self.p_hidden if self.training else 0.0
        pa = self.p_attn if self.training else 0.0
        self._seed += 64
        seed = self._seed
        ws["seed"], ws["ph"], ws["pa"] = seed, ph, pa
        sel = self.split_slots
        if sel is None and self.split_set != "all":
            sel = lambda slot: not slot.endswith("W1")
        lo = (lambda slot: P.ptr(slot, "Pl") if (sel is None or sel(slot)) else 0) if (self.split_w and self.lo_mode == "pass2" and not raw) else (lambda slot: 0)   # low-order weight halves
        lo_mean = self.split_w and self.lo_mode == "mean" and not raw
        beff_off = [0]
        # every lo_row_stride-th row feeds the mean row -- but never fewer than ~1000 rows (a 16-caption batch has 544 token rows: all of them)
        lo_stride = max(1, min(self.lo_row_stride, T // 1024))          # rows are [sequence][token]: a stride that shares a factor with Tk would visit only some token positions
        while math.gcd(lo_stride, Tk) != 1:
            lo_stride += 1

        cen = self.cen and not raw
        tails = {}                                     # Linear -> bias row (b + W h_ref) that the LayerNorm launch writing its CENTRED input has left for it
        ws["raw_fwd"] = bool(raw) and (self.cen or self.res32 or self.split_w)      # backward() refuses a pass that skipped the parity corrections it differentiates

        def slot(n):
            out = _p(ws["beff"]) + beff_off[0] * 4
            beff_off[0] += n
            return out

        def bias_of(wslot, bslot, a_ptr, K, Nn, resid=None, bias_ptr=None):
            """bias pointer of a forward Linear; in the mean-row mode: bias + lo . mean row of the input (sampled rows).
            resid = (r_ref, y_ref, bias_post, fold): the centred residual stream's reference rows for a residual Linear (dic_lin_prep).
            bias_ptr: the bias row to start from when it is not the parameter itself (a LayerNorm tail has already added W h_ref to it)."""
            b0 = P.ptr(bslot) if bias_ptr is None else bias_ptr
            if not lo_mean or (resid is None and sel is not None and not sel(wslot)):
                return b0
            out = slot(Nn)
            if cen:
                r_ref, y_ref, b_post, fold = resid if resid is not None else (0, 0, 0, 0)
                _lib.check(lib.dic_lin_prep(a_ptr, T, K, lo_stride, K, P.ptr(wslot, "Pb") if resid is not None else 0, P.ptr(wslot, "Pl"), K, Nn, b0,
                                            r_ref, fold, out, b_post, y_ref, _p(ws["lomean_ws"]), st), "lin_prep")
                return out
            _lib.check(lib.dic_lo_mean_bias(a_ptr, T, K, lo_stride, K, P.ptr(wslot, "Pl"), K, Nn, P.ptr(bslot), out, _p(ws["lomean_ws"]), st),
                       "lo_mean_bias")
            return out
        qk_lo = self.split_qk if self.split_qk is not None else (self.split_set == "all" or self.split_slots is not None)
        v_col0 = 0 if qk_lo else 2 * D                     # q|k|v GEMM: first output column whose weight rows take the second pass
        keep_u = torch.is_grad_enabled()            # the FFN pre-activation is only read by the backward: forward-only calls (no_grad) skip its store
        ws["has_u"] = keep_u
        gelu_d = ws["gelu_d"] = self.bf16 and OPT.gelu_d and not OPT.gemm_v1          # Lw["u"] then holds gelu'(u), not u
        # forward-only calls (sampling, validation): nothing reads gelu'(u), so the epilogue that also evaluates it is not used (same g, bit for bit)
        gelu_epi = EPI_BIAS_GELU_D if (gelu_d and keep_u) else EPI_BIAS_GELU
        if x_ptr is None:
            if x.data_ptr() != ws["xin"].data_ptr():
                ws["xin"][:N].copy_(x)
            x_ptr = _p(ws["xin"])
        if image_clip is not None and not inputs_ready:                     # None: the caller filled the workspace's input buffers itself
            ws["img_in"][:N].copy_(image_clip.reshape(N, 512))
            ws["txt_in"][:N].copy_(text_clip.reshape(N, 512))
            ws["kmask"][:N].copy_(key_mask)
            if add_txt is not None:
                ws["addtxt"][:N].copy_(add_txt)
            else:
                ws["addtxt"][:N].zero_()               # a reused workspace must not keep an earlier batch's guided-row flags
        if self.temb:                                  # per-sequence timestep for the optional timestep embedding (None: no injection)
            if tidx is not None:
                ws["tidx"][:N].copy_(tidx.reshape(N))
            else:
                ws["tidx"][:N].fill_(-1)
        temb_p, tidx_p = (P.ptr("temb"), _p(ws["tidx"])) if self.temb else (0, 0)
        mode = ws["mode"]
        # K3: CLIP projections, exact fp32 MFMA (tiny)
        if not inputs_ready:
            o.gemm(_p(ws["img_in"]), P.ptr("Wimg"), _p(ws["img_p"]), N, D, 512, 512, 512, D, bias=P.ptr("bimg"), out_f32=1, dtype=DIC_F32)
            if mode != 2:
                o.gemm(_p(ws["txt_in"]), P.ptr("Wtxt"), _p(ws["txt_p"]), N, D, 512, 512, 512, D, bias=P.ptr("btxt"), out_f32=1, dtype=DIC_F32)
        # K4: concat/add fusion + segment + position + LayerNorm (+ dropout)
        _lib.check(lib.dic_fuse_ln_fwd_x(self.dt, mode, x_ptr, x_stride, _p(ws["img_p"]), _p(ws["txt_p"]), _p(ws["addtxt"]),
                                         P.ptr("seg") if self.concat else 0, P.ptr("pos"), temb_p, tidx_p, P.ptr("eln_g"), P.ptr("eln_b"),
                                         _p(ws["h"][0]), _p(ws["mean0"]), _p(ws["rstd0"]), N, L, D, LN_EPS, ph, seed, st), "fuse_ln_fwd")
        r32 = self.res32 and not raw
        ws["cen_fwd"] = cen
        if r32:          # the embedding LayerNorm once more in fp32 (same dropout mask: it is a function of seed and position): layer 0's residual
            _lib.check(lib.dic_fuse_ln_fwd_x(DIC_F32, mode, x_ptr, x_stride, _p(ws["img_p"]), _p(ws["txt_p"]), _p(ws["addtxt"]),
                                             P.ptr("seg") if self.concat else 0, P.ptr("pos"), temb_p, tidx_p, P.ptr("eln_g"), P.ptr("eln_b"),
                                             _p(ws["h32"][0]), _p(ws["mean0"]), _p(ws["rstd0"]), N, L, D, LN_EPS, ph, seed, st), "fuse_ln_fwd")
        of = OUT_F32_RES_F32 if r32 else 0
        if cen:
            refs = ws["refs"]
            ref = lambda i, j: refs.data_ptr() + ((i * 4 + j) * D) * 4          # j: 0 y1_ref, 1 sa_ref, 2 y2_ref, 3 h_ref of layer i + 1
            zero_ref = ref(self.n_layers, 0)                                      # (the last block of `refs` is never written: zeros)
            bpost = lambda i, j: ws["bpost"].data_ptr() + ((i * 2 + j) * D) * 4
        for i in range(self.n_layers):
            Lw, h = ws["layers"][i], ws["h"][i]
            pre = f"L{i}."
            if cen:
                # The CENTRED stream: h (layers >= 1) and sa hold bf16(value - reference row); that ONE tensor is the MFMA operand of the next Linear
                # (whose bias carries W h_ref, written by the LayerNorm launch that produced the tensor: `tails`) and the residual operand of the next
                # residual GEMM; y1, y2 hold bf16(sum - predicted mean row).  Layer 0's h is the embedding LayerNorm's output as it is (reference 0).
                h_ref = zero_ref if i == 0 else ref(i - 1, 3)
                if not OPT.cen_operand:
                    # A/B form (options.cen_operand = False): LayerNorm writes the uncentred bf16 operand copy AND the centred residual copy
                    tb = tails.pop(pre + "Wqkv", None)
                    o.gemm(_p(h), P.ptr(pre + "Wqkv", wsrc), _p(Lw["qkv"]), T, 3 * D, D, D, D, 3 * D,
                           bias=tb if (tb is not None and OPT.qkv_pred) else bias_of(pre + "Wqkv", pre + "bqkv", _p(h), D, 3 * D))
                    _lib.check(lib.dic_attn_fwd(self.dt, _p(Lw["qkv"]), _p(ws["kmask"]), _p(Lw["ctx"]), N, Tk, self.n_heads, 64, pa, seed + 4 * i + 1, st), "attn_fwd")
                    o.gemm(_p(Lw["ctx"]), P.ptr(pre + "Wo", wsrc), _p(Lw["y1"]), T, D, D, D, D, D,
                           bias=bias_of(pre + "Wo", pre + "bo", _p(Lw["ctx"]), D, D, resid=(h_ref, ref(i, 0), 0, 1)), R=_p(h) if i == 0 else _p(ws["hc"][i]), ldr=D)
                    _lib.check(lib.dic_ln_fwd_cen(_p(Lw["y1"]), ref(i, 0), P.ptr(pre + "ln1g"), P.ptr(pre + "ln1b"), _p(Lw["sa"]), _p(Lw["sac"]), ref(i, 1),
                                                  _p(Lw["m1"]), _p(Lw["r1"]), T, D, LN_EPS, 0, 0, 0, 0, 0, 0, st), "ln_fwd_cen")
                    o.gemm(_p(Lw["sa"]), P.ptr(pre + "W1", wsrc), _p(Lw["g"]), T, Hd, D, D, D, Hd, epi=gelu_epi,
                           bias=bias_of(pre + "W1", pre + "b1", _p(Lw["sa"]), D, Hd), aux=_p(Lw["u"]) if keep_u else 0, ldaux=Hd)
                    drop2 = ph > 0.0
                    o.gemm(_p(Lw["g"]), P.ptr(pre + "W2", wsrc), _p(Lw["y2"]), T, D, Hd, Hd, Hd, D,
                           bias=bias_of(pre + "W2", pre + "b2", _p(Lw["g"]), Hd, D, resid=(ref(i, 1), ref(i, 2), bpost(i, 1) if drop2 else 0, 0 if drop2 else 1)),
                           bias2=bpost(i, 1) if drop2 else 0, R=_p(Lw["sac"]), ldr=D, p_drop=ph, seed=seed + 4 * i + 2)
                    last = i + 1 == self.n_layers
                    nx = (0, 0, 0, 0, 0, 0)
                    if not last and OPT.qkv_pred:
                        t2 = slot(3 * D)
                        tails[f"L{i + 1}.Wqkv"] = t2
                        nx = (0, P.ptr(f"L{i + 1}.Wqkv", "Pl"), D, 3 * D, P.ptr(f"L{i + 1}.bqkv"), t2)
                    _lib.check(lib.dic_ln_fwd_cen(_p(Lw["y2"]), ref(i, 2), P.ptr(pre + "ln2g"), P.ptr(pre + "ln2b"), _p(ws["h"][i + 1]),
                                                  0 if last else _p(ws["hc"][i + 1]), ref(i, 3), _p(Lw["m2"]), _p(Lw["r2"]), T, D, LN_EPS, *nx, st), "ln_fwd_cen")
                    continue
                tb = tails.pop(pre + "Wqkv", None)
                if tb is None:
                    qkv_bias = bias_of(pre + "Wqkv", pre + "bqkv", _p(h), D, 3 * D)
                else:              # (options.qkv_pred: the lo half's share of the mean-row correction stays at the tail's prediction W_lo h_ref)
                    qkv_bias = tb if OPT.qkv_pred else bias_of(pre + "Wqkv", pre + "bqkv", _p(h), D, 3 * D, bias_ptr=tb)
                o.gemm(_p(h), P.ptr(pre + "Wqkv", wsrc), _p(Lw["qkv"]), T, 3 * D, D, D, D, 3 * D, bias=qkv_bias)
                _lib.check(lib.dic_attn_fwd(self.dt, _p(Lw["qkv"]), _p(ws["kmask"]), _p(Lw["ctx"]), N, Tk, self.n_heads, 64, pa, seed + 4 * i + 1, st), "attn_fwd")
                o.gemm(_p(Lw["ctx"]), P.ptr(pre + "Wo", wsrc), _p(Lw["y1"]), T, D, D, D, D, D,
                       bias=bias_of(pre + "Wo", pre + "bo", _p(Lw["ctx"]), D, D, resid=(h_ref, ref(i, 0), 0, 1)), R=_p(h), ldr=D)
                t1 = slot(Hd)              # FFN lin1: b1 + W_hi sa_ref (it takes no lo correction, measured or predicted: options.split_set)
                lo1 = P.ptr(pre + "W1", "Pl") if (sel is None or sel(pre + "W1")) else 0
                _lib.check(lib.dic_ln_fwd_cen(_p(Lw["y1"]), ref(i, 0), P.ptr(pre + "ln1g"), P.ptr(pre + "ln1b"), 0, _p(Lw["sa"]), ref(i, 1),
                                              _p(Lw["m1"]), _p(Lw["r1"]), T, D, LN_EPS, P.ptr(pre + "W1", "Pb"), lo1, D, Hd, P.ptr(pre + "b1"), t1, st), "ln_fwd_cen")
                o.gemm(_p(Lw["sa"]), P.ptr(pre + "W1", wsrc), _p(Lw["g"]), T, Hd, D, D, D, Hd, epi=gelu_epi,
                       bias=bias_of(pre + "W1", pre + "b1", _p(Lw["sa"]), D, Hd, bias_ptr=t1) if lo1 else t1, aux=_p(Lw["u"]) if keep_u else 0, ldaux=Hd)
                drop2 = ph > 0.0
                o.gemm(_p(Lw["g"]), P.ptr(pre + "W2", wsrc), _p(Lw["y2"]), T, D, Hd, Hd, Hd, D,
                       bias=bias_of(pre + "W2", pre + "b2", _p(Lw["g"]), Hd, D, resid=(ref(i, 1), ref(i, 2), bpost(i, 1) if drop2 else 0, 0 if drop2 else 1)),
                       bias2=bpost(i, 1) if drop2 else 0, R=_p(Lw["sa"]), ldr=D, p_drop=ph, seed=seed + 4 * i + 2)
                nxt = ("Wvt", "bvt", D) if i + 1 == self.n_layers else (f"L{i + 1}.Wqkv", f"L{i + 1}.bqkv", 3 * D)
                t2 = slot(nxt[2])
                tails[nxt[0]] = t2
                lo2 = P.ptr(nxt[0], "Pl") if (sel is None or sel(nxt[0])) else 0
                _lib.check(lib.dic_ln_fwd_cen(_p(Lw["y2"]), ref(i, 2), P.ptr(pre + "ln2g"), P.ptr(pre + "ln2b"), 0, _p(ws["h"][i + 1]), ref(i, 3),
                                              _p(Lw["m2"]), _p(Lw["r2"]), T, D, LN_EPS, P.ptr(nxt[0], "Pb"), lo2, D, nxt[2], P.ptr(nxt[1]), t2, st), "ln_fwd_cen")
                continue
            # K5: q|k|v projections as one GEMM
            o.gemm(_p(h), P.ptr(pre + "Wqkv", wsrc), _p(Lw["qkv"]), T, 3 * D, D, D, D, 3 * D, bias=bias_of(pre + "Wqkv", pre + "bqkv", _p(h), D, 3 * D),
                   B2=lo(pre + "Wqkv"), b2_col0=v_col0)
            # K6: attention
            _lib.check(lib.dic_attn_fwd(self.dt, _p(Lw["qkv"]), _p(ws["kmask"]), _p(Lw["ctx"]), N, Tk, self.n_heads, 64, pa, seed + 4 * i + 1, st), "attn_fwd")
            # K7: out-proj + bias + residual, then LayerNorm
            o.gemm(_p(Lw["ctx"]), P.ptr(pre + "Wo", wsrc), _p(Lw["y1"]), T, D, D, D, D, D, bias=bias_of(pre + "Wo", pre + "bo", _p(Lw["ctx"]), D, D), R=_p(ws["h32"][i]) if r32 else _p(h), ldr=D,
                   B2=lo(pre + "Wo"), out_f32=of)
            if r32:
                _lib.check(lib.dic_ln_fwd_r32(_p(Lw["y1"]), P.ptr(pre + "ln1g"), P.ptr(pre + "ln1b"), _p(Lw["sa"]), _p(Lw["sa32"]), _p(Lw["m1"]), _p(Lw["r1"]), T, D, LN_EPS, st), "ln_fwd")
            else:
                _lib.check(lib.dic_ln_fwd(self.dt, _p(Lw["y1"]), P.ptr(pre + "ln1g"), P.ptr(pre + "ln1b"), _p(Lw["sa"]), _p(Lw["m1"]), _p(Lw["r1"]), T, D, LN_EPS, st), "ln_fwd")
            # K8: FFN
            o.gemm(_p(Lw["sa"]), P.ptr(pre + "W1", wsrc), _p(Lw["g"]), T, Hd, D, D, D, Hd, epi=gelu_epi, bias=bias_of(pre + "W1", pre + "b1", _p(Lw["sa"]), D, Hd),
                   aux=_p(Lw["u"]) if keep_u else 0, ldaux=Hd, B2=lo(pre + "W1"))
            o.gemm(_p(Lw["g"]), P.ptr(pre + "W2", wsrc), _p(Lw["y2"]), T, D, Hd, Hd, Hd, D, bias=bias_of(pre + "W2", pre + "b2", _p(Lw["g"]), Hd, D), R=_p(Lw["sa32"]) if r32 else _p(Lw["sa"]), ldr=D,
                   p_drop=ph, seed=seed + 4 * i + 2, B2=lo(pre + "W2"), out_f32=of)
            if r32:
                _lib.check(lib.dic_ln_fwd_r32(_p(Lw["y2"]), P.ptr(pre + "ln2g"), P.ptr(pre + "ln2b"), _p(ws["h"][i + 1]),
                                              _p(ws["h32"][i + 1]) if i + 1 < self.n_layers else 0, _p(Lw["m2"]), _p(Lw["r2"]), T, D, LN_EPS, st), "ln_fwd")
            else:
                _lib.check(lib.dic_ln_fwd(self.dt, _p(Lw["y2"]), P.ptr(pre + "ln2g"), P.ptr(pre + "ln2b"), _p(ws["h"][i + 1]), _p(Lw["m2"]), _p(Lw["r2"]), T, D, LN_EPS, st), "ln_fwd")
        # K9: MLM-head transform: Linear -> GELU -> LayerNorm
        o.gemm(_p(ws["h"][-1]), P.ptr("Wvt", wsrc), _p(ws["uvt"]), T, D, D, D, D, D, bias=bias_of("Wvt", "bvt", _p(ws["h"][-1]), D, D, bias_ptr=tails.pop("Wvt", None)),
               B2=lo("Wvt"), out_f32=int(self.uvt32))
        _lib.check(lib.dic_gelu_ln_fwd(self.dt_u, _p(ws["uvt"]), P.ptr("vln_g"), P.ptr("vln_b"), _p(ws["x_out"]), _p(ws["mv"]), _p(ws["rv"]), T, D, LN_EPS, st), "gelu_ln_fwd")
        self._saved = ws
        return ws["x_out"][:N]

    # ------------------------------------------------------------------ encoder backward
    def backward(self, dx_out=None, layer_done=None):
        """dx_out [N,Tk,768] fp32 (defaults to the workspace buffer the loss kernels filled).  Accumulates nothing:
        every parameter gradient in `params.G` is overwritten (pos rows >= Tk stay zero from zero_grad).
        layer_done(i): optional callback fired once layer i's gradients are complete (data-parallel overlap)."""
        ws = self._saved
        assert ws is not None, "backward() without a saved forward"
        assert ws.get("has_u", True), "backward() after a forward run under torch.no_grad() (the FFN pre-activations were not kept)"
        if ws.get("raw_fwd", False):
            raise RuntimeError("backward() after encode(raw=True): the raw pass is forward-only (it skipped the reference rows / fp32 residual copies / "
                               "lo-weight corrections this engine's backward reads)")
        N, L, Tk, T, D, Hd = ws["N"], ws["L"], ws["Tk"], ws["T"], self.dim, self.hidden
        o, P, lib = self.ops, self.params, self.ops.L
        o.begin()
        st = o.stream
        wsrc = "Pb" if self.bf16 else "P"
        seed, ph, pa = ws["seed"], ws["ph"], ws["pa"]
        dx = ws["dx_out"] if dx_out is None else dx_out
        csw = _p(ws["cs_ws"])
        parts = [_p(t_) for t_ in ws["partial"]]

        skw = _p(ws["splitk"])
        skcap = ws["splitk"].numel()

        # ---- weight gradients on a second stream.  dW = dY^T X depends only on dY and on activations saved by the forward, never
        # feeds the dX chain, and is MFMA-bound, while the chain it leaves behind alternates MFMA-bound GEMMs with HBM-bound
        # LayerNorm / attention kernels and store-heavy epilogues: letting the two streams share the CUs overlaps those phases.
        if P.zero_pending:          # zero_grad(): clear what this backward will not overwrite
            P.zero_pending = False
            pos_rows = P._slots["pos"][1][0]
            _lib.check(lib.dic_zero(P.ptr("pos", "G") + Tk * D * 4, (pos_rows - Tk) * D * 4, st), "zero")
            if ws["mode"] == 2:
                _lib.check(lib.dic_zero(P.ptr("Wtxt", "G"), D * 512 * 4, st), "zero")
                _lib.check(lib.dic_zero(P.ptr("btxt", "G"), D * 4, st), "zero")
        main = torch.cuda.current_stream()
        use_side = self.bf16 and OPT.wgrad_stream and self.wgrad_stream_enabled
        side = self._side_stream() if use_side else None
        done = {}                                     # layer -> event on the side stream after that layer's dW launches

        # Each side launch is handed over as soon as its inputs exist (one event on the main stream per hand-over; batching the hand-overs
        # to two per layer saved events but started the side work later: 1.5 % slower, round 2).
        pending = []
        # (the hand-over events stay alive until the backward returns: under hipGraph capture a destroyed event's handle is reused by the next one,
        # and a step with ~40 more hand-overs than before replayed with the embedding gradients computed too early -- round 5)
        evs = []

        def flush_side():
            if not pending:
                return
            if use_side:
                ev = torch.cuda.Event()
                evs.append(ev)
                ev.record(main)
                side.wait_event(ev)
                o.stream = side.cuda_stream
            try:
                for fn in pending:
                    fn()
            finally:
                o.stream = st
                pending.clear()

        def on_side(fn):
            pending.append(fn)
            flush_side()

        # Grouped weight gradients (dic_wgrad_group: one K-slice count for all the tiles of several Linears, one launch + one fold).
        # DIC_WGRAD_GROUP=1 groups all four Linears of a layer: 2.4 % faster with everything on one stream (15.79 vs 16.16 ms) but 1-2 %
        # SLOWER with the weight gradients on their second stream -- a 256-workgroup persistent kernel holding 128 KB of LDS per CU for
        # ~400 us keeps the main stream's GEMMs off the CUs for that long, ~100 us launches interleave with them.  The default, 2, groups only
        # out-proj + qkv, the two with too few tiles to split well on their own (9 + 27 tiles x 7 slices = one round): 0.5-0.8 % on the step.
        # Round 4 re-measured with the halves (DIC_WGRAD_GROUP_HALVES, below): "1" = [lin2, lin1] as one launch and [out-proj, qkv] as another
        # ties "2" on the two-stream step (13.93 vs 13.93 ms, profiles/r04_wgrad_group_ab.txt) with 24 launches fewer per step -> default "1".
        # Round 5, "pair": the split-K slabs (2.4 GB written + re-read per step) and the 25 fold launches go away when ONE launch carries two layers:
        # 216 tiles of 272 K-steps = one round of the 256 CUs at 84 %, each tile written once, in place (dic_wgrad_group picks one slice per tile).
        gmode = OPT.wgrad_group
        pair = gmode == "pair" and self.bf16 and len(ws["dy"]) >= 4
        if gmode == "pair" and not pair:
            gmode = "1"
        group = self.bf16 and gmode in ("1", "2", "pair")
        items = []
        open_layers = []                              # pair mode: layers whose weight-gradient items wait for the launch

        def flush_group():
            if not items:
                return
            cap_ = OPT.wgrad_cu_cap if use_side else 0
            batches = [list(items)]
            items.clear()
            arr0 = (_lib.WgradItem * len(batches[0]))(*batches[0])
            if len(batches[0]) > 4 and lib.dic_wgrad_group_ws_bytes(arr0, len(batches[0]), T, cap_) > skcap * 4:
                batches = [batches[0][:4], batches[0][4:]]        # (a two-layer group whose K cut needs more slab space than the workspace has: one launch per layer)
            arrs = [(_lib.WgradItem * len(b_))(*b_) for b_ in batches]
            todo = list(rank1)
            rank1.clear()

            def launch():          # ONE hand-over to the weight-gradient stream: the group launch(es), then the rank-one completions they feed
                for arr in arrs:
                    _lib.check(lib.dic_wgrad_group(arr, len(arr), T, skw, skcap * 4, cap_, o.stream), "wgrad_group")
                for dW_, db_, xr_, M_, N_ in todo:
                    _lib.check(lib.dic_rank1_add(dW_, db_, xr_, M_, N_, o.stream), "rank1_add")
            on_side(launch)

        rank1 = []                                    # (dW, db, x_ref, M, N): weight gradients whose X was stored centred

        def flush_rank1(todo):
            """dW += db x_ref^T for the Linears that read a CENTRED tensor (X = X_c + 1 x_ref^T; the GEMM contracted dY with X_c): behind
            the launch that wrote dW and db, on the same stream.  `todo` = the entries of THAT launch only -- entries of grouped items still
            waiting in `items` stay in `rank1` until flush_group launches them (dic_wgrad_group overwrites dW in place)."""
            if todo:
                def launch():
                    for dW_, db_, xr_, M_, N_ in todo:
                        _lib.check(lib.dic_rank1_add(dW_, db_, xr_, M_, N_, o.stream), "rank1_add")
                on_side(launch)

        def wgrad(dY, X, slot, M, N, lda, ldb, bias_slot=None, x_ref=0, db_slot=None):
            """dW[M][N] = dY^T X over all T tokens: (k-major, k-major) GEMM, split along K to fill the chip; in bf16 mode the
            bias gradient colsum(dY) comes out of the same launch (fp32 mode: separate dic_colsum).
            x_ref: X holds bf16(input - x_ref) (centred stream): dW is completed by db x_ref^T (db: the gradient of bias_slot / db_slot)."""
            r1 = [(P.ptr(slot, "G"), P.ptr(bias_slot if bias_slot is not None else db_slot, "G"), x_ref, M, N)] if x_ref else []
            if group and M % 256 == 0 and N % 8 == 0 and (gmode in ("1", "pair") or slot.endswith(("Wo", "Wqkv"))):
                items.append(_lib.WgradItem(dY=dY, ldy=lda, X=X, ldx=ldb, dW=P.ptr(slot, "G"), db=P.ptr(bias_slot, "G") if bias_slot is not None else 0, M=M, N=N))
                rank1.extend(r1)                      # completed behind the group launch that writes this dW (flush_group)
                return
            sk, tile = pick_split_k(M, N, T, 64 if self.bf16 else 32)
            if not self.bf16:
                tile = 128
            while sk > 1 and sk * (M * N + M) > skcap:
                sk -= 1
            cs = P.ptr(bias_slot, "G") if (bias_slot is not None and self.bf16) else 0

            def launch():
                o.gemm(dY, X, P.ptr(slot, "G"), M, N, T, lda, ldb, N, a_km=1, b_km=1, out_f32=1, split_k=sk, split_ws=skw if sk > 1 else 0,
                       colsum_out=cs, tile=tile, cu_cap=OPT.wgrad_cu_cap if use_side else 0)
            on_side(launch)
            flush_rank1(r1)
            if bias_slot is not None and not self.bf16:
                colsum(self.dt, dY, T, M, lda, P.ptr(bias_slot, "G"))

        def colsum(in_dtype, src, rows, cols, ld, dst, acc=0):
            _lib.check(lib.dic_colsum(in_dtype, src, rows, cols, ld, dst, acc, csw, st), "colsum")

        def fold(pbuf, cols, dst):
            """Column sums of a LayerNorm-backward partial buffer -> gamma/beta/bias gradients.  Nothing on the dX chain reads them,
            so the fold runs on the weight-gradient stream (one launch + one dependent-launch gap less on the main stream per LayerNorm)."""
            on_side(lambda: _lib.check(lib.dic_colsum(DIC_F32, pbuf, NPART, cols, cols, dst, 0, csw, o.stream), "colsum"))

        def fold2(pbuf0, dst0, pbuf1, dst1, cols):
            """The two LayerNorm folds of one encoder layer as ONE launch (round 4: 12 launches fewer per step)."""
            on_side(lambda: _lib.check(lib.dic_colsum_pair(pbuf0, dst0, pbuf1, dst1, NPART, cols, cols, o.stream), "colsum_pair"))

        def finish_layer(j):
            """dW launches of layer j are queued: mark it, and hand the layer's gradient slice to the data-parallel reducer.
            Pair mode: the launch goes out after every second encoder layer (and after the last one); both layers are marked then."""
            open_layers.append(j)
            if pair and j < self.n_layers and j > 0 and len(open_layers) < 2:
                flush_side()                          # (this layer's LayerNorm folds go out now; its weight gradients wait for the next layer's)
                return
            flush_group()
            flush_side()
            ev_side = None
            if use_side:
                ev_side = torch.cuda.Event()
                ev_side.record(side)
            for jj in open_layers:
                if use_side:
                    done[jj] = ev_side
                if layer_done is not None and jj < self.n_layers:
                    if use_side:
                        ev = torch.cuda.Event()
                        evs.append(ev)
                        ev.record(main)                   # the LayerNorm / bias gradients of this layer come from the main stream
                        side.wait_event(ev)
                        with torch.cuda.stream(side):
                            layer_done(jj)
                    else:
                        layer_done(jj)
            open_layers.clear()

        # head: GELU+LN backward, vocab_transform
        nl = self.n_layers
        npar = len(ws["dy"])
        sp = nl % npar
        dyb = ws["dy"][sp]
        _lib.check(lib.dic_gelu_ln_bwd(self.dt_u, _p(dx), _p(ws["uvt"]), P.ptr("vln_g"), _p(ws["mv"]), _p(ws["rv"]), _p(dyb), parts[2 * sp], NPART, T, D, st), "gelu_ln_bwd")
        fold(parts[2 * sp], 3 * D, P.ptr("vln_g", "G"))                                      # [vln_g | vln_b | bvt]
        cenb = bool(ws.get("cen_fwd")) and OPT.cen_operand
        xref = (lambda i_, j_: ws["refs"].data_ptr() + ((i_ * 4 + j_) * D) * 4) if cenb else (lambda i_, j_: 0)
        wgrad(_p(dyb), _p(ws["h"][-1]), "Wvt", D, D, D, D, x_ref=xref(nl - 1, 3), db_slot="bvt")
        finish_layer(nl)
        dH, dHn = ws["dHa"], ws["dHb"]
        o.gemm(_p(dyb), P.ptr("Wvt", wsrc), _p(dH), T, D, D, D, D, D, b_km=1)
        for i in reversed(range(nl)):
            Lw, h = ws["layers"][i], ws["h"][i]
            pre = f"L{i}."
            use_drop = ph > 0.0
            sp = i % npar
            if (i + npar) in done:
                main.wait_event(done[i + npar])       # the dW GEMMs of layer i + npar have finished with this set's buffers
            dy_, dyd_, dy1_, du_, dqkv_ = ws["dy"][sp], ws["dyd"][sp], ws["dy1"][sp], ws["du"][sp], ws["dqkv"][sp]
            # output_layer_norm backward; bias grad of lin2 folded in
            if ws["cen_fwd"]:
                yref = lambda j: ws["refs"].data_ptr() + ((i * 4 + j) * D) * 4
                _lib.check(lib.dic_ln_bwd_cen(_p(dH), _p(Lw["y2"]), yref(2), P.ptr(pre + "ln2g"), _p(Lw["m2"]), _p(Lw["r2"]), _p(dy_),
                                              _p(dyd_) if use_drop else 0, ph, seed + 4 * i + 2, parts[2 * sp], NPART, T, D, st), "ln_bwd")
            else:
                _lib.check(lib.dic_ln_bwd(self.dt_ln, _p(dH), _p(Lw["y2"]), P.ptr(pre + "ln2g"), _p(Lw["m2"]), _p(Lw["r2"]), _p(dy_),
                                          _p(dyd_) if use_drop else 0, ph, seed + 4 * i + 2, parts[2 * sp], NPART, T, D, st), "ln_bwd")
            dyd = dyd_ if use_drop else dy_
            wgrad(_p(dyd), _p(Lw["g"]), pre + "W2", D, Hd, D, Hd)
            o.gemm(_p(dyd), P.ptr(pre + "W2", wsrc), _p(du_), T, Hd, D, D, Hd, Hd, b_km=1, epi=EPI_MUL_AUX if ws["gelu_d"] else EPI_GELU_BWD,
                   aux=_p(Lw["u"]), ldaux=Hd)
            wgrad(_p(du_), _p(Lw["sa"]), pre + "W1", Hd, D, Hd, D, bias_slot=pre + "b1", x_ref=xref(i, 1))             # dW1 (+ db1)
            if not pair:
                flush_group()                         # the two FFN gradients go out now (72 tiles), out-proj + qkv at the end of the layer (36):
            flush_side()                              # one launch per layer starts the side stream too late to hide behind this layer's chain
            o.gemm(_p(du_), P.ptr(pre + "W1", wsrc), _p(ws["dsa"]), T, D, Hd, Hd, D, D, b_km=1, R=_p(dy_), ldr=D)       # + residual
            # sa_layer_norm backward; bias grad of out_lin folded in
            if ws["cen_fwd"]:
                _lib.check(lib.dic_ln_bwd_cen(_p(ws["dsa"]), _p(Lw["y1"]), yref(0), P.ptr(pre + "ln1g"), _p(Lw["m1"]), _p(Lw["r1"]), _p(dy1_),
                                              0, 0.0, 0, parts[2 * sp + 1], NPART, T, D, st), "ln_bwd")
            else:
                _lib.check(lib.dic_ln_bwd(self.dt_ln, _p(ws["dsa"]), _p(Lw["y1"]), P.ptr(pre + "ln1g"), _p(Lw["m1"]), _p(Lw["r1"]), _p(dy1_),
                                          0, 0.0, 0, parts[2 * sp + 1], NPART, T, D, st), "ln_bwd")
            # [ln2g | ln2b | b2] and [ln1g | ln1b | bo] in one launch
            fold2(parts[2 * sp], P.ptr(pre + "ln2g", "G"), parts[2 * sp + 1], P.ptr(pre + "ln1g", "G"), 3 * D)
            wgrad(_p(dy1_), _p(Lw["ctx"]), pre + "Wo", D, D, D, D)
            o.gemm(_p(dy1_), P.ptr(pre + "Wo", wsrc), _p(ws["dctx"]), T, D, D, D, D, D, b_km=1)
            _lib.check(lib.dic_attn_bwd(self.dt, _p(Lw["qkv"]), _p(ws["kmask"]), _p(ws["dctx"]), _p(dqkv_), N, Tk, self.n_heads, 64, pa,
                                        seed + 4 * i + 1, st), "attn_bwd")
            wgrad(_p(dqkv_), _p(h), pre + "Wqkv", 3 * D, D, 3 * D, D, bias_slot=pre + "bqkv", x_ref=xref(i - 1, 3) if i > 0 else 0)   # dWqkv (+ dbqkv)
            o.gemm(_p(dqkv_), P.ptr(pre + "Wqkv", wsrc), _p(dHn), T, D, 3 * D, 3 * D, D, D, b_km=1, R=_p(dy1_), ldr=D)
            dH, dHn = dHn, dH
            finish_layer(i)
        # embeddings LayerNorm + fusion backward (touches none of the side stream's buffers: runs under layer 0's weight gradients)
        mode = ws["mode"]
        _lib.check(lib.dic_fuse_ln_bwd(self.dt, mode, _p(ws["xin"]), _p(ws["img_p"]), _p(ws["txt_p"]), _p(ws["addtxt"]),
                                       P.ptr("seg") if self.concat else 0, P.ptr("pos"), P.ptr("temb") if self.temb else 0,
                                       _p(ws["tidx"]) if self.temb else 0, P.ptr("eln_g"), _p(dH), _p(ws["mean0"]), _p(ws["rstd0"]),
                                       _p(ws["dy0"]), parts[2 * npar], NPART, N, L, D, ph, seed, st), "fuse_ln_bwd")
        dy0 = _p(ws["dy0"])
        if self.temb:
            _lib.check(lib.dic_temb_grad(dy0, _p(ws["tidx"]), N, Tk, D, P.temb_steps, P.ptr("temb", "G"), st), "temb_grad")
        small_rows = N <= 1024                   # dic_colsum's single-launch path needs no workspace (the two-stage path shares `csw`)

        def embedding_grads():                   # feed only G: under the CLIP-projection GEMMs below, on the side stream
            s_ = o.stream
            _lib.check(lib.dic_colsum(DIC_F32, parts[2 * npar], NPART, 2 * D, 2 * D, P.ptr("eln_g", "G"), 0, csw, s_), "colsum")   # [eln_g | eln_b]
            _lib.check(lib.dic_colsum(DIC_F32, dy0, N, Tk * D, Tk * D, P.ptr("pos", "G"), 0, csw, s_), "colsum")             # dpos[0:Tk]
            if self.concat:
                gpos = P.ptr("pos", "G")
                _lib.check(lib.dic_colsum(DIC_F32, gpos, L, D, D, P.ptr("seg", "G"), 0, csw, s_), "colsum")                  # dseg[0] = sum_{t<L}
                _lib.check(lib.dic_colsum(DIC_F32, gpos + L * D * 4, Tk - L, D, D, P.ptr("seg", "G") + D * 4, 0, csw, s_), "colsum")   # dseg[1]
        if small_rows:
            on_side(embedding_grads)
        else:
            embedding_grads()
        if self.concat:
            dimg, dtxt, ldd = dy0 + L * D * 4, dy0 + (L + 1) * D * 4, Tk * D
        else:
            # "add" fusion: the projected CLIP rows were broadcast over the sequence -> sum the row gradients
            _lib.check(lib.dic_seq_sum(dy0, _p(ws["addtxt"]), _p(ws["dimg"]), _p(ws["dtxt"]), N, L, D, st), "seq_sum")
            dimg, dtxt, ldd = _p(ws["dimg"]), _p(ws["dtxt"]), D
        # 768 x 512 outputs = 24 tiles of the fp32 kernel: cut the contraction over the N sequences so the launch is not 24 workgroups
        skt = max(1, min(8, N // 128))
        skt_ws = _p(ws["splitk_tail"]) if skt > 1 else 0
        o.gemm(dimg, _p(ws["img_in"]), P.ptr("Wimg", "G"), D, 512, N, ldd, 512, 512, a_km=1, b_km=1, out_f32=1, dtype=DIC_F32, split_k=skt, split_ws=skt_ws)
        colsum(DIC_F32, dimg, N, D, ldd, P.ptr("bimg", "G"))
        if mode != 2:      # text row dropped: text_linear's gradient is exactly zero (G was zeroed by zero_grad)
            o.gemm(dtxt, _p(ws["txt_in"]), P.ptr("Wtxt", "G"), D, 512, N, ldd, 512, 512, a_km=1, b_km=1, out_f32=1, dtype=DIC_F32, split_k=skt, split_ws=skt_ws)
            colsum(DIC_F32, dtxt, N, D, ldd, P.ptr("btxt", "G"))
        if use_side:
            main.wait_stream(side)                    # every weight gradient is in G before anything downstream (AdamW, all-reduce tail)

    # ------------------------------------------------------------------ rounding head: streaming CE / argmax (ref :323, 436-437, 620)
    @property
    def head_centered(self):
        return self.bf16 and not self.te and (OPT.head_center == "1" or (OPT.head_center in ("w", "auto") and self.split_w))

    def center_head_input(self, cw, x_a, n_a, x_b, n_b, L, Tk):
        """bf16 engines: rewrite cw["xr"] as bf16(x - xbar) over the head rows (rows t < L of the n_a sequences at x_a and the n_b at x_b, fp32
        [n][Tk][768]) and leave xbar W^T (fp32) in cw["cvec"]: `rounding` / `rounding_train` then hand it to the head GEMM as its bias
        (include/dic_hip.h, dic_head_center -- why: DESIGN.md section 4).  DIC_HEAD_CENTER=0 switches it off (A/B)."""
        if not self.head_centered:
            cw["centered"] = False
            return
        o = self.ops
        o.begin()
        _lib.check(o.L.dic_head_center(x_a, n_a, x_b, n_b, L, Tk, 768, _p(self.W_lm), self.vpad, _p(cw["hc_ws"]), _p(cw["xbar"]), _p(cw["cvec"]),
                                       _p(cw["xr"]), o.stream), "head_center")
        cw["centered"] = True

    def rounding(self, xr, M, tgt=None, ce_ws=None, dtype=None):
        """xr [M,768] (compute dtype) -> (lse[M], argmax[M], nll[M] or None) without materialising the logits."""
        cw = ce_ws or self._ce_workspace(M)
        cw["fused"] = False
        o = self.ops
        o.begin()
        cbias = _p(cw["cvec"]) if (cw.get("centered") and ce_ws is not None and dtype is None) else 0
        W = self.W_lm_c if dtype is None else (self.W_lm if dtype == DIC_F32 else self.W_lm_c)
        f32 = dtype == DIC_F32 or (dtype is None and not self.bf16)
        tile = 128 if (f32 or OPT.gemm_v1) else choose_tile(M, self.vocab, 1, EPI_CE_PARTIAL)
        np_ = o.L.dic_ce_n_partials(self.vocab, tile)
        o.gemm(_p(xr), _p(W), 0, M, self.vocab, 768, 768, 768, 0, epi=EPI_CE_PARTIAL, tgt=_p(tgt) if tgt is not None else 0,
               partial=_p(cw["partial"]), tgt_logit=_p(cw["tgt_logit"]), dtype=dtype, tile=tile, bias=cbias)
        _lib.check(o.L.dic_ce_combine(_p(cw["partial"]), _p(cw["tgt_logit"]), M, np_, _p(cw["lse"]), _p(cw["argmax"]),
                                      _p(cw["nll"]) if tgt is not None else 0, o.stream), "ce_combine")
        return cw["lse"], cw["argmax"], (cw["nll"] if tgt is not None else None)

    # shift of the per-row reference point of rounding_train's exponentials above the target logit (include/dic_hip.h, dic_ce_target_logit)
    CE_REF_SHIFT = 40.0

    @property
    def ce_fused(self):
        """bf16 engine: the training forward of the rounding loss leaves exp(logit - c) behind, so the backward needs no second logits GEMM
        (DIC_CE_FUSED=0: the recompute path, kept as the A/B partner and for the fp32 engine)."""
        return self.bf16 and not OPT.gemm_v1 and OPT.ce_fused

    def rounding_train(self, xr, M, tgt, ce_ws=None):
        """Training form of `rounding` (ref :323, 436-437 with their backward in mind): one GEMM over the vocabulary writes
        E = exp(logit - c) in bf16 (1 GB at M = 16 384, the buffer the recompute path uses for dlogits) + per-slab sums; lse / nll come from the
        sums, and E with its target entries patched is (softmax - onehot) up to the per-row factor 1/Z that `rounding_backward`'s caller
        applies.  Saves the backward's recompute of the logits (0.7 ms of a 14.7 ms step)."""
        cw = ce_ws or self._ce_workspace(M)
        o = self.ops
        o.begin()
        if cw["dlogits"] is None:
            cw["dlogits"] = torch.empty(M, self.vpad, dtype=self.tdtype, device=self.device)
        L = o.L
        tile = choose_tile(M, self.vocab, 1, EPI_CE_EXP)
        np_ = L.dic_ce_n_partials(self.vocab, tile)
        cbias = _p(cw["cvec"]) if (cw.get("centered") and ce_ws is not None) else 0
        _lib.check(L.dic_ce_target_logit(_p(xr), _p(self.W_lm_c), _p(tgt), M, self.vocab, 768, self.CE_REF_SHIFT, _p(cw["tgt_logit"]), _p(cw["cref"]),
                                         cbias, o.stream), "ce_target_logit")
        o.gemm(_p(xr), _p(self.W_lm_c), _p(cw["dlogits"]), M, self.vocab, 768, 768, 768, self.vpad, epi=EPI_CE_EXP, tgt=_p(tgt), lse=_p(cw["cref"]),
               partial=_p(cw["partial"]), tgt_logit=_p(cw["tgt_logit"]), tile=tile, bias=cbias)
        _lib.check(L.dic_ce_exp_combine(_p(cw["partial"]), np_, _p(cw["cref"]), _p(cw["tgt_logit"]), _p(tgt), M, self.vocab, _p(cw["dlogits"]), self.vpad,
                                        _p(cw["lse"]), _p(cw["nll"]), _p(cw["inv_z"]), o.stream), "ce_exp_combine")
        cw["fused"] = True
        return cw["lse"], None, cw["nll"]

    def rounding_backward(self, cw, M, rows_a, scale_a, scale_b):
        """dxr = ((softmax - onehot) * row_scale) @ W  via a recompute GEMM with the dlogits epilogue + one (KC,KM) GEMM.  After
        `rounding_train` (cw["fused"]) the first GEMM is skipped and dxr comes back WITHOUT the per-row factor row_scale / Z: the caller
        folds it into the add of dxr to the encoder-output gradient (dic_add_rows_scaled with cw["inv_z"])."""
        o = self.ops
        o.begin()
        if cw["dlogits"] is None:
            cw["dlogits"] = torch.empty(M, self.vpad, dtype=self.tdtype, device=self.device)
        ws_split = self._saved["splitk"]
        if not cw.get("fused"):          # (a mean-centred head input: the recomputed logits need the same column bias as the forward's)
            o.gemm(_p(cw["xr"]), _p(self.W_lm_c), _p(cw["dlogits"]), M, self.vocab, 768, 768, 768, self.vpad, epi=EPI_CE_DLOGITS,
                   tgt=_p(cw["tgt"]), lse=_p(cw["lse"]), ce_rows_a=rows_a, ce_scale_a=scale_a, ce_scale_b=scale_b,
                   bias=_p(cw["cvec"]) if cw.get("centered") else 0)
        # 64 x 3 = 192 256-tiles are 0.75 of a round and x2 slices 1.5 rounds: four slices of the 30592-deep contraction fill
        # three rounds exactly (measured at M = 16384: 862 us vs 1064 us for two slices, 1032 us unsplit)
        sk = next((k for k in (4, 2) if M * 768 * k <= ws_split.numel() and M >= 2048), 1)
        o.gemm(_p(cw["dlogits"]), _p(self.W_lm_c), _p(cw["dxr"]), M, 768, self.vpad, self.vpad, 768, 768, b_km=1, out_f32=1,
               split_k=sk, split_ws=_p(ws_split) if sk > 1 else 0)
        return cw["dxr"]

    # ------------------------------------------------------------------ reference forward (ref :271-323)
    def _key_masks(self, mask, concat_mask, n):
        guided = (concat_mask[:, 1] == 1)
        m = (mask != 0).to(torch.uint8)
        if self.concat:
            one = torch.ones(n, 1, dtype=torch.uint8, device=self.device)
            plain = torch.cat([m, one, torch.zeros_like(one)], 1)
            gmask = torch.cat([m, one, one], 1)
        else:
            plain = gmask = m
        return guided, plain, gmask

    @torch.no_grad()
    def forward(self, x, image_clip, text_clip, mask, concat_mask, with_logits=True, *, t=None):
        """Inference-shaped call with the reference's signature: returns (vocab_logits [N,L,V], x_out [N,Tk,768]).
        Training goes through `diffusion.loss`, which shares `encode`/`rounding` but never builds the logits.
        t ([N] ints, keyword-only): the timestep of every row, REQUIRED when cfg.TIMESTEP_EMBEDDING is on (the reference's forward has no
        such input, ref :271; a model trained with the table must not silently run without it)."""
        if self.temb and t is None:
            raise ValueError("this model was built with cfg.TIMESTEP_EMBEDDING: forward() needs t= (the timestep of every row)")
        n = x.shape[0]
        L = cfg.MAX_LENGTH
        assert x.shape == (n, L, cfg.IN_CHANNEL)
        assert image_clip.shape == text_clip.shape == (n, 1, 512)
        assert mask.shape == (n, L)
        assert concat_mask.shape == (n, 2)
        if self.te:
            from . import train_embedding
            return train_embedding.forward(self, x, image_clip, text_clip, mask, concat_mask, with_logits)
        dev = self.device
        x, mask, concat_mask = x.to(dev, torch.float32), mask.to(dev), concat_mask.to(dev)
        image_clip, text_clip = image_clip.to(dev, torch.float32), text_clip.to(dev, torch.float32)
        guided, plain, gmask = self._key_masks(mask, concat_mask, n)
        w = cfg.CLASSIFIER_FREE_WEIGHT
        gi = guided.nonzero().squeeze(1) if (w > 0 and bool(guided.any())) else None
        if gi is not None:
            xs = torch.cat([x, x[gi]]); ic = torch.cat([image_clip, image_clip[gi]]); tc = torch.cat([text_clip, text_clip[gi]])
            km = torch.cat([plain, gmask[gi]])
            add_txt = torch.cat([torch.zeros(n, dtype=torch.uint8, device=dev), torch.ones(len(gi), dtype=torch.uint8, device=dev)])
        else:
            xs, ic, tc, km, add_txt = x, image_clip, text_clip, plain, torch.zeros(n, dtype=torch.uint8, device=dev)
        tidx = None
        if self.temb:
            tt = torch.as_tensor(t).to(dev, torch.int32).reshape(n)
            tidx = torch.cat([tt, tt[gi]]) if gi is not None else tt
        x_all = self.encode(xs.contiguous(), ic, tc, km, add_txt, tidx=tidx)
        if gi is not None:
            Tk = x_all.shape[1]
            _lib.check(self.ops.L.dic_cfg_mix_fwd(_p(x_all), _p(x_all[n:]), _p(gi), len(gi), Tk * 768, float(w), self.ops.stream), "cfg_mix")
        x_out = x_all[:n].clone()
        logits = self.lm_head(x_out[:, :L, :]) if with_logits else None
        return logits, x_out

    __call__ = forward
