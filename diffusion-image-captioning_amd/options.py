"""ONE record of every switch of the hot path.

Rounds 1-4 grew ~40 `DIC_*` environment variables read at import time in three modules and in the C library; a benchmark and a test could
end up on different code without either saying so.  Now: `Options` below is the whole list, with the SHIPPED value as each field's default
(`tests/test_host_cpu.py` pins them); nothing else in the package or in the library reads a switch from the environment.  A measurement that
wants another value sets

    DIC_OPTIONS="wgrad_group=1,cen=0"          (comma-separated name=value; unknown names raise)

or -- so that the A/B scripts of earlier rounds keep working -- the legacy variable of that option (`LEGACY_ENV`).  `bench.py` prints
`non_default()` in its JSON line and the GPU tests assert it is empty unless a test sets an option itself, so every number and every
assertion names the configuration it ran on.  Switches whose A/B is settled were deleted with their losing branch (see DESIGN.md section 7:
DIC_SIDE_BATCH, DIC_SIDE_PRIO, DIC_WGRAD_GROUP_HALVES, DIC_PAIR_FOLDS, DIC_MUL_AUX_TILE, DIC_GELU_FWD_TILE, DIC_GELU_BWD_TILE,
DIC_WGRAD_MAX_SPLIT, DIC_WGRAD_TILE, DIC_LO_MODE, DIC_SPLIT_W, DIC_SAMPLE_GRAPH_OFF, DIC_GEMM_PERSIST, DIC_GEMM_ROWS).

Not options: what `torchrun` provides (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), `DIC_DIST_BACKEND` / `DIC_DIST_SHARE_GPU` (how the test
harness places ranks, parallel.init_distributed), `DIC_HIP_LIB` / `HIPCC` (which library file is loaded / how it is built, _lib.py).
"""
from __future__ import annotations

import dataclasses
import os
from dataclasses import dataclass


@dataclass
class Options:
    # ---- engine (engine.py)
    wgrad_stream: bool = True        # weight gradients (and the LayerNorm folds, AdamW slices) on a second stream
    wgrad_group: str = "pair"        # "pair": the Linears of TWO layers per launch, one K-slice per tile | "1": two launches + folds per layer | "2": out-proj + qkv only | "0"
    bwd_sets: int = 0                # gradient-buffer sets shared with the weight-gradient stream; 0 = what wgrad_group needs (4 for "pair", else 2)
    wgrad_cu_cap: int = 0            # > 0: weight-gradient GEMMs keep to this many CUs
    ln_npart: int = 512              # persistent blocks (= partial rows) of the LayerNorm backward kernels
    gemm_tile: str = "auto"          # "128" | "256" | "auto"
    gemm_v1: bool = False            # bf16 on the register-staged v1 kernel (128-tiles only)
    gelu_d: bool = True              # FFN-1 leaves gelu'(u) behind (MUL_AUX backward epilogue) instead of u
    ce_fused: bool = True            # rounding loss: training forward keeps exp(logit - c), no logits recompute
    head_center: str = "1"           # mean-centred rounding-head input: "1" every bf16 engine | "w" split-weight modes only | "0"
    uvt32: bool = True               # bf16 engines keep the MLM-head pre-activation in fp32
    split_set: str = "auto"          # which forward Linears take the lo-weight correction: "auto" (bf16: all but FFN lin1; bf16w: all) | "all" | "vo2t"
    lo_row_stride: int = 16          # rows sampled for the mean row of a Linear's input: every 16th
    qkv_pred: bool = True            # q|k|v of layers >= 1: the mean row comes PREDICTED out of the LayerNorm launch that writes their input (dic_ln_fwd_cen tail)
    cen: bool = True                 # parity mode: centred bf16 residual stream + dic_lin_prep (False: round 4's fp32 residual stream + dic_lo_mean_bias)
    cen_operand: bool = True         # the centred tensor is ALSO the next Linear's MFMA operand (its bias carries W h_ref); False: a separate uncentred operand copy
    res32: str = "auto"              # fp32 residual stream: "auto" = the exact form bf16w (and bf16 with cen=False) | "1" | "0"
    sample_raw: bool = True          # sample(): encoder passes without the parity mode's corrections (per-row argmax: no batch mean to protect)
    # ---- step / sampling loop (diffusion.py)
    streamed_adamw: bool = True      # AdamW per finished gradient slice on the weight-gradient stream
    sample_graph: bool = True        # passes 3..K of a sampling loop replayed as one hipGraph
    sample_w4a: bool = True          # the four-wave asm GEMM inside sample()
    sample_two_heights: bool = True  # two tile heights per launch inside sample()
    # ---- C library (pushed through dic_set_option when the library is loaded)
    gemm_w4a: bool = True            # the four-wave asm GEMM for every eligible launch, training included (round 5: 14.76 -> 14.35 ms per step once the
                                     # weight gradients run as one launch per two layers; 18.5 -> 17.1 J per step -- profiles/r05_power_ab.txt)
    gemm_w4a_mask: int = 0x73        # which (layout, epilogue) forms may take it: bit 4 * b_km + {0 plain, 1 + residual, 2 x aux, 3 dropout + residual}, bit 8 GELU
                                     # (forward-only FFN lin1), bit 9 GELU + GELU' (training FFN lin1).  0x73 = the forms the last FULL GPU suite and the round-5
                                     # collection ran.  0x173 adds the forward-only GELU body (measured: 7.24 -> 7.21 ms per sampling pass, profiles/r05_w4a_mask_ab.txt);
                                     # the dropout + residual and the two-output GELU forms tie with the 8-wave kernel in the step
    gemm_w4a_rows: int = 256         # tile height of the asm GEMM: 256 | 224 forced, 0 = per launch (256 or 224 rows, by rounds x height; measured 14.41 -> 14.27 ms per
                                     # step, 7.73 -> 7.52 ms per pass, profiles/r05_tile_rows_probe.txt).  Round 6 ships 256 / 0x73 -- what was last verified END TO
                                     # END on hardware: the 224-row and GELU bodies passed their own GPU tests (104 cases) and the benches, but the pool closed before
                                     # the full suite and the collection could be repeated on them, and stayed closed for the whole of round 6.
                                     # DIC_OPTIONS="gemm_w4a_rows=0,gemm_w4a_mask=0x173" is the faster setting once `scripts/gpu_round.sh` has been run on it.
    gemm_w4n: bool = False           # the NARROW-tile asm GEMM (256 x 128 tiles, epilogue under the next tile's K loop: csrc/gemm_w4n.h) where eligible -- off until measured on hardware
    gemm_w4n_mask: int = 0x740       # which (layout, epilogue) forms may take it: the bits of gemm_w4a_mask + bit 10: the rounding-head forward (CE_EXP).  0x740 = the forms the issue
                                     # model favours (profiles/r06_w4n_issue_model.txt): GELU, GELU + GELU', CE_EXP and the k-major x aux form (FFN lin2's input gradient); the light forms and
                                     # everything that runs at N = 768 (one wide tile per CU already) stay on the wide bodies.  To be replaced by what scripts/experiments/w4n_ab.sh measures
    gemm_w4n_kmax: int = 1024        # launches with K above this keep the 256 x 256 bodies
    gemm_w4n_flat: bool = True       # K = 768 launches of the narrow GEMM take its loop-free bodies (False: the loop form everywhere -- A/B)
    gemm_two_heights: bool = False   # two tile heights per launch everywhere
    # ---- data parallel (parallel.py)
    dp_group: int = 3                # encoder layers per gradient slice / collective
    dp_single: bool = False          # exactly one all-reduce of the whole flat buffer after the backward
    dp_cu_cap: int = 0               # > 0: the backward's persistent GEMMs keep to this many CUs while a slice is on the wire
    dp_timing: bool = False          # bracket every collective with events
    force_reducer: bool = False      # run the exchange path at world size 1 (single-GPU test of the data-parallel code path)
    dp_timeout_s: int = 600          # rendezvous / collective timeout of the process group: a missing or dead peer ends the job with a message

    def non_default(self) -> dict:
        ref = Options()
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self) if getattr(self, f.name) != getattr(ref, f.name)}

    @property
    def n_bwd_sets(self) -> int:
        return max(2, self.bwd_sets) if self.bwd_sets else (4 if self.wgrad_group == "pair" else 2)


# legacy environment variable -> option (the measurement scripts of rounds 1-4 set these; DIC_OPTIONS wins over them)
LEGACY_ENV = {
    "DIC_WGRAD_STREAM": "wgrad_stream", "DIC_WGRAD_GROUP": "wgrad_group", "DIC_BWD_PARITY": "bwd_sets", "DIC_WGRAD_CU_CAP": "wgrad_cu_cap",
    "DIC_LN_NPART": "ln_npart", "DIC_GEMM_TILE": "gemm_tile", "DIC_GEMM": "gemm_v1", "DIC_GELU_D": "gelu_d", "DIC_CE_FUSED": "ce_fused",
    "DIC_HEAD_CENTER": "head_center", "DIC_UVT32": "uvt32", "DIC_SPLIT_SET": "split_set", "DIC_LO_ROW_STRIDE": "lo_row_stride", "DIC_CEN": "cen",
    "DIC_RES32": "res32", "DIC_STREAMED_ADAMW": "streamed_adamw", "DIC_SAMPLE_GRAPH": "sample_graph", "DIC_GEMM_W4A": "gemm_w4a",
    "DIC_GEMM_TWO_HEIGHTS": "gemm_two_heights", "DIC_DP_GROUP": "dp_group", "DIC_DP_SINGLE": "dp_single",
    "DIC_DP_CU_CAP": "dp_cu_cap", "DIC_DP_TIMING": "dp_timing", "DIC_FORCE_REDUCER": "force_reducer", "DIC_SAMPLE_RAW": "sample_raw",
}
# switches deleted with their losing branch: setting one means the caller expects a code path that no longer exists -- refuse instead of ignoring
DELETED_ENV = ("DIC_SIDE_BATCH", "DIC_SIDE_PRIO", "DIC_WGRAD_GROUP_HALVES", "DIC_PAIR_FOLDS", "DIC_MUL_AUX_TILE", "DIC_GELU_FWD_TILE", "DIC_GELU_BWD_TILE",
               "DIC_WGRAD_MAX_SPLIT", "DIC_WGRAD_TILE", "DIC_LO_MODE", "DIC_SPLIT_W", "DIC_SAMPLE_GRAPH_OFF", "DIC_GEMM_PERSIST", "DIC_GEMM_ROWS", "DIC_GEMM_PP")
_LIB_OPTIONS = ("gemm_v1", "gemm_w4a", "gemm_w4a_mask", "gemm_w4a_rows", "gemm_w4n", "gemm_w4n_mask", "gemm_w4n_kmax", "gemm_w4n_flat", "gemm_two_heights")


def _coerce(name: str, text: str):
    kind = {f.name: f.type for f in dataclasses.fields(Options)}.get(name)
    if kind is None:
        raise ValueError(f"unknown option {name!r} (diffusion-image-captioning_amd/options.py lists them)")
    kind = {"bool": bool, "int": int, "str": str}.get(kind, kind)
    if kind is bool:
        if text.strip().lower() in ("1", "true", "on", "yes"):
            return True
        if text.strip().lower() in ("0", "false", "off", "no", ""):
            return False
        raise ValueError(f"option {name}: {text!r} is not a boolean")
    return int(text.strip(), 0) if kind is int else kind(text.strip())


def from_env(env=None) -> Options:
    env = os.environ if env is None else env
    o = Options()
    gone = [v for v in DELETED_ENV if v in env]
    if gone:
        raise ValueError(f"{', '.join(gone)}: this switch was deleted together with the code path it selected (options.py lists the live ones)")
    for var, name in LEGACY_ENV.items():
        if var in env:
            if var == "DIC_GEMM_W4A" and env[var] == "0":       # (legacy meaning of "0": off inside sample() too)
                o.sample_w4a = False
            if var == "DIC_GEMM_TWO_HEIGHTS" and env[var] == "0":
                o.sample_two_heights = False
            setattr(o, name, _coerce(name, env[var]))
    for item in filter(None, (s.strip() for s in env.get("DIC_OPTIONS", "").split(","))):
        if "=" not in item:
            raise ValueError(f"DIC_OPTIONS: {item!r} is not name=value")
        k, v = item.split("=", 1)
        setattr(o, k.strip(), _coerce(k.strip(), v))
    return o


OPT = from_env()


def push_to_library(L) -> None:
    """The C library keeps four process-global switches of its own (include/dic_hip.h, dic_set_option): set them from the record."""
    for name in _LIB_OPTIONS:
        v = getattr(OPT, name)
        rc = L.dic_set_option(name.encode(), int(v))
        if rc:
            raise RuntimeError(f"dic_set_option({name}, {int(v)}) failed: {L.dic_last_error().decode(errors='replace')}")


def set_option(name: str, value) -> None:
    """Change one option at run time.  The C library's own switches (`_LIB_OPTIONS`) are process-global state of the LOADED library: assigning
    `OPT.gemm_w4a = ...` would change what `non_default()` reports but not what runs, so they go through here (tests, A/B scripts)."""
    if name not in {f.name for f in dataclasses.fields(Options)}:
        raise ValueError(f"unknown option {name!r} (diffusion-image-captioning_amd/options.py lists them)")
    if isinstance(value, str):
        value = _coerce(name, value)
    if name in _LIB_OPTIONS:
        from . import _lib
        if _lib.loaded():
            L = _lib.lib()
            rc = L.dic_set_option(name.encode(), int(value))
            if rc:
                raise RuntimeError(f"dic_set_option({name}, {int(value)}) failed: {L.dic_last_error().decode(errors='replace')}")
    setattr(OPT, name, value)
