#!/usr/bin/env python3
"""Generator of the NARROW-tile four-wave GEMM bodies (csrc/gemm_w4n_asm.inc), round 6 -- DESIGN.md section 7.0001.  Proven on the CPU (scripts/w4n_emulate.py,
scripts/w4n_hazard_check.py) and assembled by hipcc; NOT yet run on hardware when this was written (GPU use was closed): the library keeps it behind options.gemm_w4n.

Why a second geometry.  The 256 x 256 bodies of scripts/gen_w4a.py spend, on a K = 768 problem, 12 K-steps at the L2 -> LDS latency floor of a
two-stage pipeline and then an epilogue that NOTHING overlaps (one wave per SIMD, all 256 accumulation registers in use): 9 % of a tile for a plain
epilogue, 39-42 % for the GELU forms (profiles/r05_w4a_instruction_mix.txt).  Here:

  * tile 256 x 128 x 64, four waves as 2 x 2, wave tile 128 x 64 = 8 x 4 fragments = 128 accumulation registers: a SECOND set fits.  The K loop always
    accumulates in a[0:127]; during the first K-step of the next tile the finished tile moves to a[128:255] (v_accvgpr_mov, 16 per row block, in the
    MFMA slots in front of the row block's first C = 0 MFMA) and its epilogue -- bias / residual / dropout / GELU arithmetic, DPP chunk exchange, full-line
    stores -- is a QUEUE of instructions drained a few at a time from the MFMA slots of the next tile's first and last three K-steps;
  * a K-step is 48 KB (A 32 + B 16), so THREE LDS stages fit (144 KB + 16 KB tile table = the CU's 160 KB): K-step k reads stage k mod 3, the DMA of
    K-step k + 3 goes into the stage K-step k has just finished with -- two K-steps in flight instead of 1.2, and ONE barrier per K-step instead of two:
    "all my reads of stage s have returned (lgkmcnt 0) and my pieces of K-step k + 1 have landed (vmcnt N)" -> s_barrier -> K-step k + 1 is visible to
    everybody and stage s is free for everybody's DMA;
  * two forms of the K loop.  LOOP form (any K = 192 n >= 576): unrolled by three -- first / second / middle ... / last triple.  The middle triple is the only loop and
    carries no epilogue work, and the second triple is its twin, so the loop's counted waits see the same VMEM history on every entry (asserted); the queue drains in the
    first and the last triple only.  FLAT form (`flat=12`: K = 768, the model's only short K; what csrc/gemm_w4n.h takes for it): twelve K-steps of straight-line text, the
    queue paced evenly over all of them but the first (`Pacer`) -- scripts/w4n_issue_model.py shows why: with a GELU queue the loop form's five draining K-steps are
    issue-bound at ~3 000 cycles while its clean ones idle at the DMA bound (0.89 of the wide bodies' cost), the flat form lands every K-step near the DMA bound (0.69;
    0.85 in that script's stricter model).
    Every vmcnt is DERIVED from the order in which the generator has issued loads, LDS-DMA pieces and stores (gfx950: one in-order counter);
  * the last three K-steps' DMA slots carry the NEXT tile's first three K-steps (descriptor swap), as in the wide bodies; the kernel prologue replays the last
    triple's VMEM sequence (real pieces of tile 0, null stores for everything else; two-pass generation) so that the first tile's waits see the history every later tile sees;
  * epilogue forms: the wide bodies' six + `ceexp`, the rounding-head forward (DIC_EPI_CE_EXP: bf16(exp(acc + bias - c_row)), zeros beyond a ragged N, the unrounded sums per
    64-column slab, the target's logit; per-row loads, v_permlane swaps and out-of-range lane offsets instead of EXEC masks -- see epilogue_queue).

LDS (bytes): A stages [0, 32K), [32K, 64K), [64K, 96K); B stages [96K, 112K), [112K, 128K), [128K, 144K); tile table [144K, +16K).
LDS images, swizzle keys, B-row permutation, MFMA operand order (D = Bfrag x Afrag), DPP exchange and full-line stores are those of gen_w4a.py / the 8-wave
kernel restricted to one 64-column slab per wave, so the three kernels produce the same values.

Per K-step and wave: 64 MFMAs (1 024 cycles at the matrix peak), 24 fragment reads, 12 LDS-DMA pieces.  Budget: DESIGN.md section 7.0001.

    python scripts/gen_w4n.py diffusion-image-captioning_amd/csrc/gemm_w4n_asm.inc
"""
import struct
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_w4a as W  # noqa: E402
from gen_w4a import (Asm, OPS, OP, GELU_CONSTS, GC_K1, GC_NC, GC_A5, GC_A4, GC_A3, GC_A2, GC_A1, GC_PHI,  # noqa: E402,F401
                     S_ARG, S_A, S_Bp, S_C, S_BIAS, S_R, S_M, S_N, S_Kd, S_LDA, S_LDB, S_LDC, S_LDR, S_ABYTES, S_BBYTES, S_CBYTES, S_RBYTES, S_KSTEPB,
                     S_NPAIRS, S_LDA64, S_LDBP, S_LDC2, S_LDR2, S_BIASBYTES, S_RSA, S_RSB, S_NXA, S_NXB, S_RSC, S_RSBIAS, S_RSR, S_NULL, S_NXC_OFF,
                     S_NXN0, S_CUR_C_OFF, S_CUR_N0, S_KA, S_KB, S_PAIRS, S_TILE, S_TIDX, S_M0A, S_M0B, S_LDC16, S_LDC8, S_SOFF, S_T, S_LDR16, S_LDR8,
                     S_NXPAIR, S_NXR_OFF, S_CUR_R_OFF, S_N8, S_HC2, S_HC1, S_CURPAIR, S_EXTRA, S0, S_LAST)

NW = 4
NI, NJ = 8, 4
A_ST = [0, 32768, 65536]
B_BASE = 98304
B_ST = [98304, 114688, 131072]
TABLE_OFF = 147456
LDS_BYTES = TABLE_OFF + 16384
ACC1 = 128               # the finished tile's accumulators

# ---- vector registers ----------------------------------------------------------------------------------------------
V_B = [0, 16]            # B fragment buffers (kk parity), 4 fragments x 4
V_A = 32                 # A fragments, 8 x 4
V_BIAS = 64              # 16: bias quads of the wave's 4 B fragments (loaded by the epilogue queue, from the FINISHED tile's descriptor)
V_T = 80                 # 24 temporaries of the epilogue
V_VOA = 104              # 8 DMA source offsets (A pieces)
V_VOB = 112              # 4 (B pieces)
V_AA = [116, 117]        # A fragment address kk0 / kk1, stages 0 and 1 (ds_read offsets have 16 bits)
V_AA2 = [118, 119]       # ... stage 2 (+ 64K)
V_AB = [120, 121]        # B fragment addresses: KC kk0 / kk1; KM column pair 0 / 1 (B region base folded in)
V_CST, V_BOFF, V_TBL, V_RST, V_PAIRB = 122, 123, 124, 125, 126
R_BASE = 128             # side-input tile of the wave: 8 row blocks x 8 registers (rows 0-7 / 8-15 of the block, 8 columns per lane each)
V_GC = 192               # gelu / gelud: 8 constant pairs
V_GX = 208               # gelu / gelud: second fragment + scratch pairs (16)
V_T2 = 224               # gelud: second output, 8 packed registers + 4 exchange temporaries
V_LAST = W.V_LAST
S_EPAIR = S_CURPAIR      # the FINISHED tile's first pair index (dropout mask of its epilogue)
S_RUNPAIR = S_LDC2       # the running tile's (gen_w4a.py computes 2 ldc there and never uses it)
# ---- the CE_EXP form (rounding-head forward: C = bf16(exp(acc + bias - c_row)), slab sums, target logit; k-contiguous B only).  It has no side input and no dropout:
V_CROW = 128             # 8: the row reference points c[m] of the lane's row in each row block
V_TG = 136               # 16: the rows' target ids (int64: lo, hi)
V_MASK = 152             # 16: 1.0 / 0.0 per (fragment, element): column < N
V_L2E, V_C2, V_SUM, V_TV, V_D, V_OWN, V_POFF, V_NL, V_ADR, V_S2 = 168, 170, 172, 176, 177, 178, 179, 180, 182, 184
S_DLSE, S_DTL, S_DPART = 16, 36, 76      # buffer resources over the WHOLE lse / tgt_logit / partial arrays (constant for the kernel): s16..19 (dropout scalars), s36..39 (K, lda, ldb,
                                          # ldc: prologue only), s76..79 (side-input descriptor)
S_TGTP = S_R             # s32..33: the tgt pointer (csrc/gemm_w4n.h passes it in the side-input slot; read with global loads: no fourth resource is free)
S_EPOFF, S_EROW, S_EN0, S_NP64, S_M1 = S_LDR8, S_NXPAIR, S_RBYTES, S_LDR2, S_N8     # finished tile: partial byte offset, first row, first column; np * 64; M - 1  (all dropout / side-input scalars)
S_CNT = S_PAIRS          # middle-triple counter
S_TRIP = S_NPAIRS        # K-steps / 3


def acc(i, j):
    return 4 * (NJ * i + j)


class Queue:
    """The epilogue as data: instructions in program order, replayed into the K loop's MFMA slots (Gen.drain).  VMEM operations keep their tag so that the
    main Asm tracks the TRUE issue order; waits for the queue's own loads are markers resolved at replay time against that order."""

    def __init__(self):
        self.items = []
        self.n_head = 0

    def __call__(self, text):
        self.items.append(("ins", text))

    def vmem(self, tag, text):
        self.items.append(("vmem", tag, text))

    def wait_vm(self, tag):
        self.items.append(("waitvm", tag))

    def __len__(self):
        return len(self.items)


def parse_opts(opts):
    """measurement options (NOT in the shipped .inc; `python scripts/gen_w4n.py OUT bar=2 quota=4` writes a variant file for scripts/build_variant.sh):
         bar=g     the K-step's barrier stands in front of phase-1 group g (default 3; 2 ... 4)
         quota=n   at least n queued epilogue instructions per MFMA slot (default 2: light epilogues spread into the last triple; 4 keeps a plain /
                   residual epilogue -- and all of its stores -- inside the tile's first triple)
       and the FORM of the K loop (both forms are shipped: csrc/gemm_w4n.h takes the flat one where K allows):
         flat=n    no loop: the tile's n K-steps (n = 12: K = 768, the model's only short K) are straight-line text and ALL of them but the first drain the epilogue queue
                   -- with the loop form the clean middle K-steps idle at the DMA bound while the few that drain a GELU queue are issue-bound at twice that
                   (scripts/w4n_issue_model.py)
         pk=1      keep the epilogue's packed fp32 arithmetic (v_pk_fma / v_pk_mul / v_pk_add_f32) as the wide bodies have it.  Default 0: every packed operation of the queue is
                   emitted as its two scalar halves (same IEEE operation per element: same bits) -- /opt/skills/guides/MI355X_MICROARCH.md measures packed fp32 VALU operations
                   BESIDE MFMAs at +22-26 cycles beyond their issue slot ("an anti-lever beside MFMAs"; a transcendental costs ~2 there).  In the wide bodies the epilogue runs
                   alone and packing won 2 %; here the queue runs between MFMAs."""
    d = dict(bar=3, quota=2, flat=0, pk=0)
    for o in opts:
        k, _, v = o.partition("=")
        assert k in d, o
        d[k] = int(v)
    assert d["pk"] in (0, 1) and 2 <= d["bar"] <= 4 and 1 <= d["quota"] <= 16 and (d["flat"] == 0 or (d["flat"] % 3 == 0 and d["flat"] >= 9))
    return d


def unpack_pk(text):
    """One packed fp32 instruction of the queue -> its two scalar halves (low, high), or None if `text` is not one."""
    import re
    op, _, rest = text.partition(" ")
    if op not in ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32"):
        return None
    n_src = 3 if op == "v_pk_fma_f32" else 2
    toks = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", rest)]
    mods = " ".join(" ".join(t.split()[1:]) for t in toks)
    toks = [t.split()[0] for t in toks]
    dst, srcs = toks[0], toks[1:1 + n_src]

    def flags(name):
        m = re.search(name + r":\[([\d,]+)\]", mods)
        return [int(x) for x in m.group(1).split(",")] if m else None
    sel_hi, neg_lo, neg_hi = flags("op_sel_hi") or [1] * n_src, flags("neg_lo") or [0] * n_src, flags("neg_hi") or [0] * n_src
    d0 = int(re.fullmatch(r"v\[(\d+):(\d+)\]", dst).group(1))
    out = []
    for half in (0, 1):
        ops = []
        for k, s_ in enumerate(srcs):
            m = re.fullmatch(r"v\[(\d+):(\d+)\]", s_)
            if m:
                assert sel_hi[k] == 1, text           # (a register pair whose high half is taken from its low dword would make the two halves order-dependent)
                tok = f"v{int(m.group(1)) + half}"
            else:
                tok = s_                              # an inline constant: the same value for both halves
            ops.append(("-" if (neg_hi if half else neg_lo)[k] else "") + tok)
        name = {"v_pk_add_f32": "v_add_f32", "v_pk_mul_f32": "v_mul_f32", "v_pk_fma_f32": "v_fma_f32"}[op]
        if name != "v_fma_f32" and not ops[1].lstrip("-").startswith("v"):
            ops = ops[::-1]                           # (VOP2: a constant may only be the first source)
        out.append(f"{name} v{d0 + half}, " + ", ".join(ops))
    return out


class Pacer:
    """How many queued instructions the next MFMA slot drains: the queue's `total` instructions spread evenly over `slots` slots (the loop-free form: the K-steps
    then all cost the same instead of the first ones carrying a whole number per slot and the last ones nothing)."""

    def __init__(self, total, slots):
        self.total, self.slots, self.k, self.done = total, slots, 0, 0

    def __call__(self):
        self.k += 1
        want = min(self.total, -(-self.total * self.k // self.slots))
        n, self.done = want - self.done, want
        return n


class Gen:
    def __init__(self, bkm, epi, opts=()):
        self.bkm, self.epi = bkm, epi
        self.o = parse_opts(opts)
        self.BAR = self.o["bar"]
        self.a = Asm()
        self.gen = 0
        self.side = epi in ("resid", "mulaux", "dropres")
        self.drop = epi == "dropres"
        self.gelu = epi in ("gelu", "gelud")
        self.two_out = epi == "gelud"
        self.ce = epi == "ceexp"
        assert not (self.ce and bkm)
        self.loads_closed = set()
        self.region = 0
        self.vm_region = []           # region of every VMEM operation (parallel to a.vm)

    # ---------------------------------------------------------------- bookkeeping on top of Asm
    def vmem(self, tag, text):
        self.a.vmem(tag, text)
        self.vm_region.append(self.region)

    def younger(self, tag, same_region=True):
        a = self.a
        idx = max(i for i, t in enumerate(a.vm) if t == tag)
        assert not same_region or self.vm_region[idx] == self.region, ("a counted wait across a loop boundary", tag)
        return len(a.vm) - 1 - idx

    # ---------------------------------------------------------------- fragment reads / DMA
    def read_b(self, j, kk, stage, gen):
        a = self.a
        r = V_B[kk] + 4 * j
        st = B_ST[stage] - B_BASE
        if not self.bkm:
            a.read(("B", gen, kk, j), f"ds_read_b128 v[{r}:{r + 3}], v{V_AB[kk]} offset:{st + (32 * (j >> 1) + 4 * (j & 1)) * 128}")
        else:   # k-major tile [64 k][256 B]: two transpose reads (k rows 8g+{0..3} and +4), column pair j>>1 has its own address, parity = +8 B
            off = st + kk * 32 * 256 + (j & 1) * 8
            a.read(("B", gen, kk, j), f"ds_read_b64_tr_b16 v[{r}:{r + 1}], v{V_AB[j >> 1]} offset:{off}")
            a.read(("B", gen, kk, j), f"ds_read_b64_tr_b16 v[{r + 2}:{r + 3}], v{V_AB[j >> 1]} offset:{off + 4 * 256}")

    def read_a(self, i, kk, stage, gen):
        r = V_A + 4 * i
        if stage < 2:
            self.a.read(("A", gen, kk, i), f"ds_read_b128 v[{r}:{r + 3}], v{V_AA[kk]} offset:{A_ST[stage] + i * 2048}")
        else:
            self.a.read(("A", gen, kk, i), f"ds_read_b128 v[{r}:{r + 3}], v{V_AA2[kk]} offset:{i * 2048}")

    def dma_piece(self, p, stage, gen):
        a = self.a
        if p < NI:                                            # A: 8 pieces of 32 rows (4 waves x 8 rows x 128 B)
            a(f"s_add_u32 m0, s{S_M0A}, {A_ST[stage] + p * NW * 1024}")
            a("s_nop 0")
            self.vmem(("dma", gen), f"buffer_load_dwordx4 v{V_VOA + p}, s[{S_RSA}:{S_RSA + 3}], s{S_KA} offen lds")
        else:                                                 # B: 4 pieces (KC: 32 rows = 4 waves x 8 rows x 128 B; KM: 16 k-rows = 4 waves x 4 k-rows x 256 B)
            q = p - NI
            a(f"s_add_u32 m0, s{S_M0B}, {(B_ST[stage] - B_BASE) + q * 1024}")
            a("s_nop 0")
            self.vmem(("dma", gen), f"buffer_load_dwordx4 v{V_VOB + q}, s[{S_RSB}:{S_RSB + 3}], s{S_KB} offen lds")

    def mfma(self, i, j, kk, first):
        d = acc(i, j)
        b = V_B[kk] + 4 * j
        av = V_A + 4 * i
        c = "0" if first else f"a[{d}:{d + 3}]"
        self.a(f"v_mfma_f32_16x16x32_bf16 a[{d}:{d + 3}], v[{b}:{b + 3}], v[{av}:{av + 3}], {c}")

    # ---------------------------------------------------------------- the epilogue of a finished tile, as a queue
    def gelu_pairs(self, q, X, D, G, P):
        """gen_w4a.Gen.gelu_pairs (common.h gelu_fast_parts2 / gelu_fast_with_grad4 as the compiler emits them), into the queue"""
        pr = lambda r: f"v[{r}:{r + 1}]"
        gc = lambda n: pr(V_GC + 2 * n)
        R2 = range(len(X))
        if self.o["pk"]:
            for k in R2:
                q(f"v_and_b32 v{D[k]}, 0x7fffffff, v{X[k]}")
                q(f"v_and_b32 v{D[k] + 1}, 0x7fffffff, v{X[k] + 1}")
            for k in R2:
                q(f"v_pk_fma_f32 {pr(D[k])}, {pr(D[k])}, {gc(GC_K1)}, 1.0 op_sel_hi:[1,1,0]")
        else:                                                 # scalar queue: |u| is a source modifier of the fma (one VALU instruction less per element; same value)
            for k in R2:
                for h in range(2):
                    q(f"v_fma_f32 v{D[k] + h}, |v{X[k] + h}|, v{V_GC + 2 * GC_K1 + h}, 1.0")
        for k in R2:
            q(f"v_pk_mul_f32 {pr(G[k])}, {pr(X[k])}, {pr(X[k])}")
        for k in R2:
            q(f"v_rcp_f32 v{D[k]}, v{D[k]}")
            q(f"v_rcp_f32 v{D[k] + 1}, v{D[k] + 1}")
        for k in R2:
            q(f"v_pk_mul_f32 {pr(G[k])}, {pr(G[k])}, {gc(GC_NC)}")
        for k in R2:
            q(f"v_pk_fma_f32 {pr(P[k])}, {pr(D[k])}, {gc(GC_A5)}, {gc(GC_A4)}")
        for k in R2:
            q(f"v_exp_f32 v{G[k]}, v{G[k]}")
            q(f"v_exp_f32 v{G[k] + 1}, v{G[k] + 1}")
        for c in (GC_A3, GC_A2, GC_A1):
            for k in R2:
                q(f"v_pk_fma_f32 {pr(P[k])}, {pr(P[k])}, {pr(D[k])}, {gc(c)}")
        for k in R2:
            q(f"v_pk_mul_f32 {pr(P[k])}, {pr(P[k])}, {pr(D[k])} neg_lo:[0,1] neg_hi:[0,1]")
        for k in R2:
            q(f"v_pk_fma_f32 {pr(P[k])}, {pr(P[k])}, {pr(G[k])}, 1.0 op_sel_hi:[1,1,0]")
        for k in R2:
            q(f"v_bfi_b32 v{P[k]}, s{S_HC1}, v{P[k]}, v{X[k]}")
            q(f"v_bfi_b32 v{P[k] + 1}, s{S_HC1}, v{P[k] + 1}, v{X[k] + 1}")
        for k in R2:
            q(f"v_pk_fma_f32 {pr(P[k])}, {pr(P[k])}, 0.5, 0.5 op_sel_hi:[1,0,0]")
        if self.two_out:
            for k in R2:
                q(f"v_pk_mul_f32 {pr(D[k])}, {pr(X[k])}, {gc(GC_PHI)}")
            for k in R2:
                q(f"v_pk_fma_f32 {pr(D[k])}, {pr(D[k])}, {pr(G[k])}, {pr(P[k])}")
        for k in R2:
            q(f"v_pk_mul_f32 {pr(X[k])}, {pr(X[k])}, {pr(P[k])}")

    def epilogue_queue(self):
        """Everything the finished tile still needs, reading a[128:255] through the descriptors S_RSC / S_RSBIAS / S_RSR that were set when its K loop ended."""
        q = Queue()
        T = V_T
        if self.epi != "mulaux":
            for j in range(NJ):
                off = (32 * (j >> 1) + 4 * (j & 1)) * 4
                q.vmem("bias", f"buffer_load_dwordx4 v[{V_BIAS + 4 * j}:{V_BIAS + 4 * j + 3}], v{V_BOFF}, s[{S_RSBIAS}:{S_RSBIAS + 3}], 0 offen offset:{off}")
        if self.side:
            for blk in range(NI):
                for n in range(2):                            # rows 0-7 / 8-15 of the block
                    r = R_BASE + 8 * blk + 4 * n
                    q(f"s_mul_i32 s{S_T}, s{S_LDR16}, {blk}")
                    if n:
                        q(f"s_add_u32 s{S_T}, s{S_T}, s{S_LDR8}")
                    q.vmem(("side", blk), f"buffer_load_dwordx4 v[{r}:{r + 3}], v{V_RST}, s[{S_RSR}:{S_RSR + 3}], s{S_T} offen")
        if self.ce:
            # per row block: the rows' reference points (through the lse resource) and target ids (global loads: 64-bit address = tgt + 8 min(row, M - 1))
            for i in range(NI):
                q(f"s_lshl_b32 s{S_T}, s{S_EROW}, 2")
                q(f"s_add_u32 s{S_T}, s{S_T}, {64 * i}")
                q.vmem(("row", i), f"buffer_load_dword v{V_CROW + i}, v{V_RST}, s[{S_DLSE}:{S_DLSE + 3}], s{S_T} offen")
                q(f"s_add_u32 s{S_T}, s{S_EROW}, {16 * i}")
                q(f"v_lshrrev_b32 v{V_ADR}, 2, v{V_RST}")
                q(f"v_add_u32 v{V_ADR}, s{S_T}, v{V_ADR}")
                q(f"v_min_u32 v{V_ADR}, s{S_M1}, v{V_ADR}")
                q(f"v_lshlrev_b32 v{V_ADR}, 3, v{V_ADR}")
                q(f"v_mov_b32 v{V_ADR + 1}, s{S_TGTP + 1}")
                q(f"v_add_co_u32 v{V_ADR}, vcc, s{S_TGTP}, v{V_ADR}")
                q(f"v_addc_co_u32 v{V_ADR + 1}, vcc, 0, v{V_ADR + 1}, vcc")
                q.vmem(("row", i), f"global_load_dwordx2 v[{V_TG + 2 * i}:{V_TG + 2 * i + 1}], v[{V_ADR}:{V_ADR + 1}], off")
        q.n_head = len(q)                                     # the load section: issued during the first K-step, consumed from the second on (a wait for
        q(f"s_mov_b32 s{S_SOFF}, 0")                          # these loads also waits for every OLDER operation -- the DMA pieces of the K-steps in flight)
        if self.epi != "mulaux":
            q.wait_vm("bias")
        if self.ce:
            # column masks of the finished tile: 1.0 where the column exists (n < N), else 0.0 -- E is written as zeros there and the slab sums skip it
            q(f"v_add_u32 v{V_NL}, s{S_EN0}, v{V_PAIRB}")
            for j in range(NJ):
                for r in range(4):
                    q(f"v_add_u32 v{V_D}, {32 * (j >> 1) + 4 * (j & 1) + r}, v{V_NL}")
                    q(f"v_cmp_gt_u32 vcc, s{S_N}, v{V_D}")
                    q(f"v_cndmask_b32 v{V_MASK + 4 * j + r}, 0, 1.0, vcc")
        for i in range(NI):
            if self.ce:
                q.wait_vm(("row", i))
                q(f"v_mul_f32 v{V_C2}, 0x3fb8aa3b, v{V_CROW + i}")                     # c log2(e)
                q(f"v_mov_b32 v{V_C2 + 1}, v{V_C2}")
                q(f"v_sub_u32 v{V_D}, v{V_TG + 2 * i}, v{V_NL}")                       # target column relative to the lane's first column
                q(f"v_and_b32 v{V_OWN}, 0xffffffd8, v{V_D}")                           # the lane owns relative columns 0..7 and 32..39 (and only for 0 <= tgt < 2^32)
                q(f"v_or_b32 v{V_OWN}, v{V_OWN}, v{V_TG + 2 * i + 1}")
                q(f"v_cmp_le_u32 vcc, s{S_N}, v{V_TG + 2 * i}")                        # ... and tgt < N (a padded column of the last tile is nobody's target)
                q(f"v_cndmask_b32 v{V_S2}, 0, 1, vcc")
                q(f"v_or_b32 v{V_OWN}, v{V_OWN}, v{V_S2}")
                q(f"v_cmp_eq_u32 vcc, 0, v{V_OWN}")
                q(f"v_mov_b32 v{V_OWN}, 0x80000000")
                q(f"v_cndmask_b32 v{V_OWN}, v{V_OWN}, v{V_RST}, vcc")                  # store offset of the target logit: the row's, or out of every range
                q(f"v_mov_b32 v{V_TV}, 0")
                for r in range(4):
                    q(f"v_mov_b32 v{V_SUM + r}, 0")
            if self.side:
                q.wait_vm(("side", i))
                # the lane's side chunks of this block: rows 0-7 / 8-15 in line order -> q' = 0 / 1 chunks of row t (swap with lane t ^ 8)
                L0, L1 = R_BASE + 8 * i, R_BASE + 8 * i + 4
                for r in range(4):
                    q(f"v_mov_b32 v{T + 16 + r}, v{L0 + r}")
                q("s_nop 1")
                for r in range(4):
                    q(f"v_mov_b32_dpp v{L0 + r}, v{L1 + r} row_ror:8 row_mask:0xf bank_mask:0xc")
                for r in range(4):
                    q(f"v_mov_b32_dpp v{L1 + r}, v{T + 16 + r} row_ror:8 row_mask:0xf bank_mask:0x3")
            # P0 / P1: the lane's 8 consecutive columns of column groups q' = 0 / 1 (fragments 2 q' + e, e = 0, 1)
            for qp in range(2):
                if self.gelu:
                    XB = V_GX
                    for e in range(2):
                        j = 2 * qp + e
                        d = ACC1 + acc(i, j)
                        x0 = T + 8 if e == 0 else XB
                        for r in range(4):
                            q(f"v_accvgpr_read_b32 v{x0 + r}, a{d + r}")
                        q(f"v_pk_add_f32 v[{x0}:{x0 + 1}], v[{x0}:{x0 + 1}], v[{V_BIAS + 4 * j}:{V_BIAS + 4 * j + 1}]")
                        q(f"v_pk_add_f32 v[{x0 + 2}:{x0 + 3}], v[{x0 + 2}:{x0 + 3}], v[{V_BIAS + 4 * j + 2}:{V_BIAS + 4 * j + 3}]")
                    X = [T + 8, T + 10, XB, XB + 2]
                    D = [T + 16, T + 20, XB + 4, XB + 6]
                    G = [T + 18, T + 22, XB + 8, XB + 10]
                    P = [T + 12, T + 14, XB + 12, XB + 14]
                    self.gelu_pairs(q, X, D, G, P)
                    for e in range(2):
                        q(f"v_cvt_pk_bf16_f32 v{T + 4 * qp + 2 * e}, v{X[2 * e]}, v{X[2 * e] + 1}")
                        q(f"v_cvt_pk_bf16_f32 v{T + 4 * qp + 2 * e + 1}, v{X[2 * e + 1]}, v{X[2 * e + 1] + 1}")
                        if self.two_out:
                            q(f"v_cvt_pk_bf16_f32 v{V_T2 + 4 * qp + 2 * e}, v{D[2 * e]}, v{D[2 * e] + 1}")
                            q(f"v_cvt_pk_bf16_f32 v{V_T2 + 4 * qp + 2 * e + 1}, v{D[2 * e + 1]}, v{D[2 * e + 1] + 1}")
                    continue
                for e in range(2):
                    j = 2 * qp + e
                    d = ACC1 + acc(i, j)
                    for r in range(4):
                        q(f"v_accvgpr_read_b32 v{T + 8 + r}, a{d + r}")
                    if self.epi != "mulaux":
                        q(f"v_pk_add_f32 v[{T + 8}:{T + 9}], v[{T + 8}:{T + 9}], v[{V_BIAS + 4 * j}:{V_BIAS + 4 * j + 1}]")
                        q(f"v_pk_add_f32 v[{T + 10}:{T + 11}], v[{T + 10}:{T + 11}], v[{V_BIAS + 4 * j + 2}:{V_BIAS + 4 * j + 3}]")
                    if self.ce:
                        cj = 32 * qp + 4 * e
                        for r in range(4):                    # the target's logit, if this is its column
                            q(f"v_cmp_eq_u32 vcc, {cj + r}, v{V_D}")
                            q(f"v_cndmask_b32 v{V_TV}, v{V_TV}, v{T + 8 + r}, vcc")
                        # exp2(min(x log2 e - c log2 e, 100)) (gemm.hip CE_EXP: the cap keeps a row's fp32 sums finite), zero beyond column N, summed unrounded
                        q(f"v_pk_fma_f32 v[{T + 8}:{T + 9}], v[{T + 8}:{T + 9}], v[{V_L2E}:{V_L2E + 1}], v[{V_C2}:{V_C2 + 1}] neg_lo:[0,0,1] neg_hi:[0,0,1]")
                        q(f"v_pk_fma_f32 v[{T + 10}:{T + 11}], v[{T + 10}:{T + 11}], v[{V_L2E}:{V_L2E + 1}], v[{V_C2}:{V_C2 + 1}] neg_lo:[0,0,1] neg_hi:[0,0,1]")
                        for r in range(4):
                            q(f"v_min_f32 v{T + 8 + r}, 0x42c80000, v{T + 8 + r}")
                        for r in range(4):
                            q(f"v_exp_f32 v{T + 8 + r}, v{T + 8 + r}")
                        q(f"v_pk_mul_f32 v[{T + 8}:{T + 9}], v[{T + 8}:{T + 9}], v[{V_MASK + 4 * j}:{V_MASK + 4 * j + 1}]")
                        q(f"v_pk_mul_f32 v[{T + 10}:{T + 11}], v[{T + 10}:{T + 11}], v[{V_MASK + 4 * j + 2}:{V_MASK + 4 * j + 3}]")
                        q(f"v_pk_add_f32 v[{V_SUM}:{V_SUM + 1}], v[{V_SUM}:{V_SUM + 1}], v[{T + 8}:{T + 9}]")
                        q(f"v_pk_add_f32 v[{V_SUM + 2}:{V_SUM + 3}], v[{V_SUM + 2}:{V_SUM + 3}], v[{T + 10}:{T + 11}]")
                    if self.drop:
                        # common.h dropout4 / pair_hash, bit for bit (gen_w4a.py): pair = (m N + n) / 2; one 32-bit hash decides two elements
                        P, H0, H1, TT = T + 16, T + 17, T + 18, T + 19
                        q(f"s_mul_i32 s{S_T}, s{S_N8}, {i}")
                        q(f"s_add_u32 s{S_T}, s{S_T}, {16 * qp + 2 * e}")
                        q(f"s_add_u32 s{S_T}, s{S_T}, s{S_EPAIR}")
                        q(f"v_add_u32 v{P}, s{S_T}, v{V_PAIRB}")
                        q(f"v_xor_b32 v{H0}, {OP['dkey']}, v{P}")
                        q(f"v_add_u32 v{H1}, 1, v{P}")
                        q(f"v_xor_b32 v{H1}, {OP['dkey']}, v{H1}")
                        for sh, mul in ((16, S_HC1), (15, S_HC2), (16, None)):
                            for H in (H0, H1):
                                q(f"v_lshrrev_b32 v{TT}, {sh}, v{H}")
                                q(f"v_xor_b32 v{H}, v{H}, v{TT}")
                                if mul is not None:
                                    q(f"v_mul_lo_u32 v{H}, v{H}, s{mul}")
                        for r in range(4):
                            q(f"v_mul_f32 v{T + 8 + r}, {OP['dinv']}, v{T + 8 + r}")
                        for r in range(4):
                            H = H0 if r < 2 else H1
                            if r & 1:
                                q(f"v_lshrrev_b32 v{TT}, 16, v{H}")
                            else:
                                q(f"v_and_b32 v{TT}, 0xffff, v{H}")
                            q(f"v_cmp_le_u32 vcc, {OP['dthr']}, v{TT}")
                            q(f"v_cndmask_b32 v{T + 8 + r}, 0, v{T + 8 + r}, vcc")
                    if self.side:
                        src = (R_BASE + 8 * i + (4 if qp else 0)) + 2 * e            # two dwords: columns 4e..4e+3 of the chunk
                        q(f"v_lshlrev_b32 v{T + 20}, 16, v{src}")
                        q(f"v_and_b32 v{T + 21}, 0xffff0000, v{src}")
                        q(f"v_lshlrev_b32 v{T + 22}, 16, v{src + 1}")
                        q(f"v_and_b32 v{T + 23}, 0xffff0000, v{src + 1}")
                        op = "v_pk_mul_f32" if self.epi == "mulaux" else "v_pk_add_f32"
                        q(f"{op} v[{T + 8}:{T + 9}], v[{T + 8}:{T + 9}], v[{T + 20}:{T + 21}]")
                        q(f"{op} v[{T + 10}:{T + 11}], v[{T + 10}:{T + 11}], v[{T + 22}:{T + 23}]")
                    q(f"v_cvt_pk_bf16_f32 v{T + 4 * qp + 2 * e}, v{T + 8}, v{T + 9}")
                    q(f"v_cvt_pk_bf16_f32 v{T + 4 * qp + 2 * e + 1}, v{T + 10}, v{T + 11}")
            # D0 = P0 with lanes t >= 8 taking P1 of lane t - 8; D1 = P1 with lanes t < 8 taking P0 of lane t + 8
            for r in range(4):
                q(f"v_mov_b32 v{T + 12 + r}, v{T + r}")
            q("s_nop 1")
            for r in range(4):
                q(f"v_mov_b32_dpp v{T + r}, v{T + 4 + r} row_ror:8 row_mask:0xf bank_mask:0xc")
            for r in range(4):
                q(f"v_mov_b32_dpp v{T + 4 + r}, v{T + 12 + r} row_ror:8 row_mask:0xf bank_mask:0x3")
            nt = " nt" if (self.gelu or self.ce) else ""
            q.vmem("store", f"buffer_store_dwordx4 v[{T}:{T + 3}], v{V_CST}, s[{S_RSC}:{S_RSC + 3}], s{S_SOFF} offen{nt}")
            q(f"s_add_u32 s{S_T}, s{S_SOFF}, s{S_LDC8}")
            q.vmem("store", f"buffer_store_dwordx4 v[{T + 4}:{T + 7}], v{V_CST}, s[{S_RSC}:{S_RSC + 3}], s{S_T} offen{nt}")
            if self.two_out:                                  # the same exchange and two full-line stores for gelu'(u) -> aux (descriptor / strides of the side input)
                U = V_T2
                for r in range(4):
                    q(f"v_mov_b32 v{U + 8 + r}, v{U + r}")
                q("s_nop 1")
                for r in range(4):
                    q(f"v_mov_b32_dpp v{U + r}, v{U + 4 + r} row_ror:8 row_mask:0xf bank_mask:0xc")
                for r in range(4):
                    q(f"v_mov_b32_dpp v{U + 4 + r}, v{U + 8 + r} row_ror:8 row_mask:0xf bank_mask:0x3")
                q(f"s_mul_i32 s{S_T}, s{S_LDR16}, {i}")
                q.vmem("store", f"buffer_store_dwordx4 v[{U}:{U + 3}], v{V_RST}, s[{S_RSR}:{S_RSR + 3}], s{S_T} offen nt")
                q(f"s_add_u32 s{S_T}, s{S_T}, s{S_LDR8}")
                q.vmem("store", f"buffer_store_dwordx4 v[{U + 4}:{U + 7}], v{V_RST}, s[{S_RSR}:{S_RSR + 3}], s{S_T} offen nt")
            if self.ce:
                # the row's sum over this wave's 64 columns: the lane's 16 values, then the four lanes that share the row (16 and 32 lanes apart:
                # v_permlane32_swap / v_permlane16_swap, as the 8-wave kernel does); lanes g = 0 store it into partial[m][slab]
                q(f"v_add_f32 v{V_SUM}, v{V_SUM}, v{V_SUM + 1}")
                q(f"v_add_f32 v{V_SUM + 2}, v{V_SUM + 2}, v{V_SUM + 3}")
                q(f"v_add_f32 v{V_SUM}, v{V_SUM}, v{V_SUM + 2}")
                q(f"v_mov_b32 v{V_S2}, v{V_SUM}")
                q("s_nop 1")
                q(f"v_permlane32_swap_b32 v{V_SUM}, v{V_S2}")
                q("s_nop 1")
                q(f"v_add_f32 v{V_SUM}, v{V_SUM}, v{V_S2}")
                q(f"v_mov_b32 v{V_S2}, v{V_SUM}")
                q("s_nop 1")
                q(f"v_permlane16_swap_b32 v{V_SUM}, v{V_S2}")
                q("s_nop 1")
                q(f"v_add_f32 v{V_SUM}, v{V_SUM}, v{V_S2}")
                q(f"s_mul_i32 s{S_T}, s{S_NP64}, {i}")
                q(f"s_add_u32 s{S_T}, s{S_T}, s{S_EPOFF}")
                q.vmem("store", f"buffer_store_dword v{V_SUM}, v{V_POFF}, s[{S_DPART}:{S_DPART + 3}], s{S_T} offen")
                q(f"s_lshl_b32 s{S_T}, s{S_EROW}, 2")
                q(f"s_add_u32 s{S_T}, s{S_T}, {64 * i}")
                q.vmem("store", f"buffer_store_dword v{V_TV}, v{V_OWN}, s[{S_DTL}:{S_DTL + 3}], s{S_T} offen")
            q(f"s_add_u32 s{S_SOFF}, s{S_SOFF}, s{S_LDC16}")
        if not self.o["pk"]:                                  # the queue runs BETWEEN MFMAs: no packed fp32 arithmetic there (parse_opts)
            items = []
            for it in q.items:
                halves = unpack_pk(it[1]) if it[0] == "ins" else None
                items += [("ins", h) for h in halves] if halves else [it]
            q.items = items
        return q

    def drain(self, q, n):
        """replay up to n queued instructions here"""
        a = self.a
        while n > 0 and q.items:
            it = q.items.pop(0)
            n -= 1
            if it[0] == "ins":
                a(it[1])
            elif it[0] == "vmem":
                self.vmem(it[1], it[2])
            else:
                tag = it[1]
                if tag in self.loads_closed:
                    continue                                  # (covered by the wait that closed the first triple)
                a(f"s_waitcnt vmcnt({min(63, self.younger(tag))})")

    def close_loads(self, q_tags=("bias",) + tuple(("side", i) for i in range(NI)) + tuple(("row", i) for i in range(NI))):
        """end of a straight-line region: everything the queue has loaded so far is waited for here, so that no later region needs a count across the loop"""
        a = self.a
        live = [t for t in q_tags if t in a.vm and t not in self.loads_closed]
        if live:
            n = min(self.younger(t) for t in live)
            a(f"s_waitcnt vmcnt({min(63, n)})")
        self.loads_closed.update(live)

    # ---------------------------------------------------------------- one K-step
    def step(self, stage, first=False, q=None, quota=0, q_from_phase=0):
        """first: accumulators start from 0 and the finished tile moves to a[128:255] in front of them.  q / quota: epilogue queue drained `quota`
        instructions per MFMA slot (from phase q_from_phase on)."""
        a = self.a
        g0 = self.gen
        self.gen += 1
        nxt = (stage + 1) % 3
        fill = {}

        def put(key, f):
            fill.setdefault(key, []).append(f)
        # ---- phase 0 fillers
        put((0, 0, 1), lambda: self.read_a(NI - 1, 0, stage, g0))                       # the last A(kk0) fragment (its registers were busy until now)
        for j, key in enumerate(((0, 0, 2), (0, 0, 3), (0, 1, 2), (0, 1, 3))):
            put(key, lambda j=j: self.read_b(j, 1, stage, g0))                          # (not the very first slots: the previous phase's last MFMAs read this buffer)
        for i in range(NI - 1):
            put((0, i + 1, 1), lambda i=i: self.read_a(i, 1, stage, g0))                # A(kk1)[i] over A(kk0)[i], after its MFMAs
        if first:
            def moves(i, part):
                for r in range(4 * part, 4 * part + 4):
                    a(f"v_accvgpr_mov_b32 a{ACC1 + 16 * i + r}, a{16 * i + r}")
            for i in range(1, NI):
                for part in range(4):
                    put((0, i - 1, part), lambda i=i, part=part: moves(i, part))
        # ---- phase 1 fillers
        put((1, 0, 1), lambda: self.read_a(NI - 1, 1, stage, g0))
        for key, kind, n in self.next_frag_schedule():
            if kind == "B":
                put(key, lambda n=n: self.read_b(n, 0, nxt, g0 + 1))
            else:
                put(key, lambda n=n: self.read_a(n, 0, nxt, g0 + 1))
        assert self.BAR + 1 <= 5                              # (A(kk0)[0..3] of the next step go into registers phase-1 groups 0-3 have finished with)
        dma_slots = [(1, g_, sl) for g_ in range(self.BAR, NI) for sl in ((1, 3) if (g_ <= self.BAR + 1 and self.BAR < 4) else (1, 2, 3))][:12]
        assert len(dma_slots) == 12
        for p, key in enumerate(dma_slots):
            put(key, lambda p=p: self.dma_piece(p, stage, g0 + 3))                      # K-step k + 3 into the stage this step has finished with
        # ---- the step
        if first:
            for r in range(16):
                a(f"v_accvgpr_mov_b32 a{ACC1 + r}, a{r}")
        for ph in range(2):
            for i in range(NI):
                if i == 0:
                    a.wait_lds(("B", g0, ph, NJ - 1), ("A", g0, ph, 0))
                else:
                    a.wait_lds(("A", g0, ph, i))
                if ph == 1 and i == self.BAR:
                    # every read of this stage is back (lgkmcnt 0) and this wave's pieces of K-step k + 1 have landed -> barrier: K-step k + 1 is visible to
                    # everybody, and this stage is free for everybody's DMA of K-step k + 3
                    a(f"s_waitcnt vmcnt({min(63, self.younger(('dma', g0 + 1), same_region=False))}) lgkmcnt(0)")
                    a("s_barrier")
                for j in range(NJ):
                    self.mfma(i, j, ph, first and ph == 0)
                    for f in fill.get((ph, i, j), []):
                        f()
                    if q is not None and ph >= q_from_phase:
                        self.drain(q, quota() if callable(quota) else quota)
        a(f"s_add_u32 s{S_KA}, s{S_KA}, 128")
        a(f"s_add_u32 s{S_KB}, s{S_KB}, s{S_KSTEPB}")

    def next_frag_schedule(self):
        """(MFMA slot, operand, fragment) of the NEXT K-step's first fragments, in issue order: behind the barrier B(kk0)[0..3], then A(kk0)[i] once phase-1
        group i has been issued ([7]: first group of the next step)"""
        sch = [((1, self.BAR, j), "B", j) for j in range(NJ)] + [((1, self.BAR + 1, i), "A", i) for i in range(4)] + \
              [((1, 5, 0), "A", 4), ((1, 6, 0), "A", 5), ((1, 7, 0), "A", 6)]
        return sorted(sch, key=lambda e: e[0])                # (stable: fragments sharing a slot keep this order)

    def first_frags(self, stage, gen):
        """the same reads in the same order without a K-step around them (kernel prologue)"""
        for _, kind, n in self.next_frag_schedule():
            if kind == "B":
                self.read_b(n, 0, stage, gen)
            else:
                self.read_a(n, 0, stage, gen)

    # ---------------------------------------------------------------- the whole body
    def body(self, mimic=None):
        """mimic: VMEM kinds ('dma' / other) of the last triple from its first DMA piece on -- the kernel prologue replays them (two-pass generation: main())"""
        a = self.a
        bkm = self.bkm
        a("s_nop 4")
        a(f"s_load_dwordx16 s[{S_ARG}:{S_ARG + 15}], {OP['karg']}, 0")
        a(f"s_load_dword s{S_LDR}, {OP['karg']}, 64")
        a(f"v_mov_b32 v{V_TBL}, {OP['tbl']}")
        a(f"v_mov_b32 v{V_CST}, {OP['cst']}")
        a(f"v_mov_b32 v{V_BOFF}, {OP['boff']}")
        a(f"v_mov_b32 v{V_RST}, {OP['rst']}")
        a(f"v_mov_b32 v{V_PAIRB}, {OP['pairb']}")
        if self.gelu:
            a(f"s_mov_b32 s{S_HC1}, 0x7fffffff")
            for n, c in enumerate(GELU_CONSTS):
                bits = struct.unpack("<I", struct.pack("<f", c))[0]
                a(f"v_mov_b32 v{V_GC + 2 * n}, 0x{bits:08x}")
                a(f"v_mov_b32 v{V_GC + 2 * n + 1}, 0x{bits:08x}")
        else:
            a(f"s_mov_b32 s{S_HC1}, 0x7feb352d")
            a(f"s_mov_b32 s{S_HC2}, 0x846ca68b")
        a(f"v_mov_b32 v{V_AA[0]}, {OP['aA0']}")
        a(f"v_xor_b32 v{V_AA[1]}, 64, {OP['aA0']}")
        a(f"v_add_u32 v{V_AA2[0]}, {A_ST[2]}, v{V_AA[0]}")
        a(f"v_add_u32 v{V_AA2[1]}, {A_ST[2]}, v{V_AA[1]}")
        a(f"v_mov_b32 v{V_AB[0]}, {OP['aB0']}")
        a(f"v_xor_b32 v{V_AB[1]}, 64, {OP['aB0']}")                      # KC: kk1 = chunk index + 4; KM: column pair 1 = chunk index + 4
        a(f"s_mov_b32 s{S_M0A}, {OP['m0A']}")
        a(f"s_mov_b32 s{S_M0B}, {OP['m0B']}")
        a(f"s_mov_b32 s{S_TILE}, {OP['ntiles']}")
        a("s_waitcnt lgkmcnt(0)")
        for ptr in (S_A, S_Bp, S_C, S_BIAS, S_R):
            a(f"s_and_b32 s{ptr + 1}, s{ptr + 1}, 0xffff")
        # derived scalars
        a(f"s_lshl_b32 s{S_LDA64}, s{S_LDA}, 6")
        a(f"s_lshl_b32 s{S_LDBP}, s{S_LDB}, {3 if bkm else 4}")          # bytes between a wave's consecutive B pieces (KM: 4 k-rows, KC: 8 rows)
        a(f"s_lshl_b32 s{S_LDC8}, s{S_LDC}, 4")
        a(f"s_lshl_b32 s{S_LDC16}, s{S_LDC}, 5")
        a(f"s_lshl_b32 s{S_LDR8}, s{S_LDR}, 4")
        a(f"s_lshl_b32 s{S_LDR16}, s{S_LDR}, 5")
        a(f"s_lshr_b32 s{S_T}, s{S_Kd}, 6")                               # K-steps / 3 (K-steps < 2^15: x * 0xAAAB >> 17)
        a(f"s_mul_i32 s{S_T}, s{S_T}, 0xaaab")
        a(f"s_lshr_b32 s{S_TRIP}, s{S_T}, 17")
        if bkm:
            a(f"s_lshl_b32 s{S_KSTEPB}, s{S_LDB}, 7")                     # 64 k-rows further
        else:
            a(f"s_mov_b32 s{S_KSTEPB}, 128")
        a(f"s_lshl_b32 s{S_BIASBYTES}, s{S_N}, 2")
        a(f"s_lshl_b32 s{S_N8}, s{S_N}, 3")

        def bytes_of(dst, rows, ld, cols):                    # ((rows - 1) * ld + cols) * 2
            a(f"s_sub_u32 s{S_T}, s{rows}, 1")
            a(f"s_mul_i32 s{S_T}, s{S_T}, s{ld}")
            a(f"s_add_u32 s{S_T}, s{S_T}, s{cols}")
            a(f"s_lshl_b32 s{dst}, s{S_T}, 1")
        bytes_of(S_ABYTES, S_M, S_LDA, S_Kd)
        if bkm:
            bytes_of(S_BBYTES, S_Kd, S_LDB, S_N)
        else:
            bytes_of(S_BBYTES, S_N, S_LDB, S_Kd)
        bytes_of(S_CBYTES, S_M, S_LDC, S_LDC if self.ce else S_N)              # (CE_EXP writes zeros into columns [N, ldc) of every row)
        bytes_of(S_RBYTES, S_M, S_LDR, S_N)
        a(f"s_or_b32 s{S_T}, s{S_BIAS}, s{S_BIAS + 1}")
        a(f"s_cmp_eq_u32 s{S_T}, 0")
        a(f"s_cselect_b32 s{S_BIASBYTES}, 0, s{S_BIASBYTES}")
        a(f"s_or_b32 s{S_T}, s{S_R}, s{S_R + 1}")
        a(f"s_cmp_eq_u32 s{S_T}, 0")
        a(f"s_cselect_b32 s{S_RBYTES}, 0, s{S_RBYTES}")
        # DMA source offsets
        a(f"v_mov_b32 v{V_VOA}, {OP['voA0']}")
        for p in range(1, NI):                                # A pieces are 32 rows apart (piece row = 32 p + 8 wave + r8), same swizzle key
            a(f"v_add_u32 v{V_VOA + p}, v{V_VOA + p - 1}, s{S_LDA64}")
        for j in range(4):
            # KC: piece j covers rows 8 j further, key bits 1-2 = j & 3;  KM: k-rows 4 j further, key = 8 ((j >> 1) & 1) (+ 2 r in chunkx)
            kx = (8 * ((j >> 1) & 1)) if bkm else 2 * (j & 3)
            a(f"v_xor_b32 v{V_T}, {kx}, {OP['chunkx']}")
            a(f"v_lshlrev_b32 v{V_T}, 4, v{V_T}")
            a(f"s_mul_i32 s{S_T}, s{S_LDBP}, {j}")
            a(f"v_add3_u32 v{V_VOB + j}, {OP['voBbase']}, v{V_T}, s{S_T}")
        a(f"s_mov_b32 s{S_NULL}, 0")
        a(f"s_mov_b32 s{S_NULL + 1}, 0")
        a(f"s_mov_b32 s{S_NULL + 2}, 0")
        a(f"s_mov_b32 s{S_NULL + 3}, 0x00020000")
        for k in range(4):                                    # nothing is finished in front of the first tile: its "epilogue" loads zeros and stores nowhere
            a(f"s_mov_b32 s{S_RSC + k}, s{S_NULL + k}")
            a(f"s_mov_b32 s{S_RSBIAS + k}, s{S_NULL + k}")
            if not self.ce:
                a(f"s_mov_b32 s{S_RSR + k}, s{S_NULL + k}")
        if not self.ce:
            a(f"s_mov_b32 s{S_EPAIR}, 0")
        else:
            # ---- CE_EXP: three more pointers behind the common arguments (lse, partial, tgt_logit; tgt came in the side-input slot, np in ldr), constant resources
            # over the whole arrays, lane constants.  The registers they take were the prologue's (K, lda, ldb, ldc) or belong to forms this one is not.
            a(f"s_load_dwordx2 s[{S_RSA}:{S_RSA + 1}], {OP['karg']}, 104")           # (three naturally aligned 8-byte loads: no assumption about the segment's alignment)
            a(f"s_load_dwordx2 s[{S_RSA + 2}:{S_RSA + 3}], {OP['karg']}, 112")
            a(f"s_load_dwordx2 s[{S_RSB}:{S_RSB + 1}], {OP['karg']}, 120")
            a("s_waitcnt lgkmcnt(0)")
            a(f"s_sub_u32 s{S_M1}, s{S_M}, 1")
            a(f"s_lshl_b32 s{S_NP64}, s{S_LDR}, 6")
            a(f"s_lshl_b32 s{S_T}, s{S_M}, 2")                                # M * 4
            a(f"s_mov_b32 s{S_DLSE}, s{S_RSA}")
            a(f"s_and_b32 s{S_DLSE + 1}, s{S_RSA + 1}, 0xffff")
            a(f"s_mov_b32 s{S_DLSE + 2}, s{S_T}")
            a(f"s_mov_b32 s{S_DLSE + 3}, 0x00020000")
            a(f"s_mov_b32 s{S_DPART}, s{S_RSA + 2}")
            a(f"s_and_b32 s{S_DPART + 1}, s{S_RSA + 3}, 0xffff")
            a(f"s_mul_i32 s{S_DPART + 2}, s{S_T}, s{S_LDR}")                  # M * np * 4
            a(f"s_mov_b32 s{S_DPART + 3}, 0x00020000")
            a(f"s_mov_b32 s{S_DTL}, s{S_RSB}")
            a(f"s_and_b32 s{S_DTL + 1}, s{S_RSB + 1}, 0xffff")
            a(f"s_mov_b32 s{S_DTL + 2}, s{S_T}")
            a(f"s_mov_b32 s{S_DTL + 3}, 0x00020000")
            a(f"s_or_b32 s{S_T}, s{S_TGTP}, s{S_TGTP + 1}")                  # no targets: nothing is picked, nothing is stored
            a(f"s_cmp_eq_u32 s{S_T}, 0")
            a(f"s_cselect_b32 s{S_DTL + 2}, 0, s{S_DTL + 2}")
            a(f"s_or_b32 s{S_T}, s{S_RSB}, s{S_RSB + 1}")
            a(f"s_cmp_eq_u32 s{S_T}, 0")
            a(f"s_cselect_b32 s{S_DTL + 2}, 0, s{S_DTL + 2}")
            a(f"v_mov_b32 v{V_L2E}, 0x3fb8aa3b")
            a(f"v_mov_b32 v{V_L2E + 1}, 0x3fb8aa3b")
            # partial[m][slab]: lane offset (row np + wn) * 4 for the lanes g = 0 (V_RST = 4 row, V_PAIRB = 64 wn + 8 g), out of every range for the others
            a(f"v_mul_lo_u32 v{V_POFF}, v{V_RST}, s{S_LDR}")
            a(f"v_lshrrev_b32 v{V_T}, 4, v{V_PAIRB}")
            a(f"v_and_b32 v{V_T}, 4, v{V_T}")
            a(f"v_add_u32 v{V_POFF}, v{V_POFF}, v{V_T}")
            a(f"v_and_b32 v{V_T}, 0x38, v{V_PAIRB}")
            a(f"v_cmp_eq_u32 vcc, 0, v{V_T}")
            a(f"v_mov_b32 v{V_T}, 0x80000000")
            a(f"v_cndmask_b32 v{V_POFF}, v{V_T}, v{V_POFF}, vcc")
            # the phantom tile in front of the first one: offsets of 2^31 -- its loads read zeros, its sums and target logits are stored nowhere
            a(f"s_mov_b32 s{S_EPOFF}, 0x80000000")
            a(f"s_mov_b32 s{S_EROW}, 0x20000000")
            a(f"s_mov_b32 s{S_EN0}, 0")
        a(f"s_mov_b32 s{S_TIDX}, 0")

        def load_next():
            """table entry S_TIDX -> next-tile descriptors (null past the end); entry = {a_off, b_off, c_off, n0}; second half {r_off, first pair}"""
            l_no, l_done = a.label("nonext"), a.label("nextdone")
            a(f"s_cmp_lt_u32 s{S_TIDX}, {OP['ntiles']}")
            a(f"s_cbranch_scc0 {l_no}")
            a(f"s_lshl_b32 s{S_T}, s{S_TIDX}, 4")
            a(f"v_add_u32 v{V_T}, s{S_T}, v{V_TBL}")
            a(f"ds_read_b128 v[{V_T + 4}:{V_T + 7}], v{V_T}")
            a(f"ds_read_b64 v[{V_T + 8}:{V_T + 9}], v{V_T} offset:8192")
            a("s_waitcnt lgkmcnt(0)")
            a(f"v_readfirstlane_b32 s{S_T}, v{V_T + 4}")
            a(f"v_readfirstlane_b32 s{S_T + 1}, v{V_T + 5}")
            a(f"v_readfirstlane_b32 s{S_NXC_OFF}, v{V_T + 6}")
            a(f"v_readfirstlane_b32 s{S_NXN0}, v{V_T + 7}")
            a(f"v_readfirstlane_b32 s{S_NXR_OFF}, v{V_T + 8}")
            if not self.ce:
                a(f"v_readfirstlane_b32 s{S_NXPAIR}, v{V_T + 9}")
            a("s_nop 3")
            a(f"s_add_u32 s{S_NXA}, s{S_A}, s{S_T}")
            a(f"s_addc_u32 s{S_NXA + 1}, s{S_A + 1}, 0")
            a(f"s_sub_u32 s{S_NXA + 2}, s{S_ABYTES}, s{S_T}")
            a(f"s_mov_b32 s{S_NXA + 3}, 0x00020000")
            a(f"s_add_u32 s{S_NXB}, s{S_Bp}, s{S_T + 1}")
            a(f"s_addc_u32 s{S_NXB + 1}, s{S_Bp + 1}, 0")
            a(f"s_sub_u32 s{S_NXB + 2}, s{S_BBYTES}, s{S_T + 1}")
            a(f"s_mov_b32 s{S_NXB + 3}, 0x00020000")
            a(f"s_branch {l_done}")
            a(f"{l_no}:")
            for k in range(4):
                a(f"s_mov_b32 s{S_NXA + k}, s{S_NULL + k}")
                a(f"s_mov_b32 s{S_NXB + k}, s{S_NULL + k}")
            a(f"{l_done}:")
            a(f"s_add_u32 s{S_TIDX}, s{S_TIDX}, 1")

        def next_to_cur():
            for k in range(4):
                a(f"s_mov_b32 s{S_RSA + k}, s{S_NXA + k}")
                a(f"s_mov_b32 s{S_RSB + k}, s{S_NXB + k}")

        def desc(dst, base, total, off):
            a(f"s_add_u32 s{dst}, s{base}, s{off}")
            a(f"s_addc_u32 s{dst + 1}, s{base + 1}, 0")
            a(f"s_sub_u32 s{dst + 2}, s{total}, s{off}")
            a(f"s_max_i32 s{dst + 2}, s{dst + 2}, 0")
            a(f"s_mov_b32 s{dst + 3}, 0x00020000")

        def take_next_offsets():
            a(f"s_mov_b32 s{S_CUR_C_OFF}, s{S_NXC_OFF}")
            a(f"s_mov_b32 s{S_CUR_N0}, s{S_NXN0}")
            a(f"s_mov_b32 s{S_CUR_R_OFF}, s{S_NXR_OFF}")
            a(f"s_mov_b32 s{S_RUNPAIR}, s{S_NXPAIR}")

        # ---- kernel prologue: tile 0's descriptors; the last triple's VMEM history replayed (its DMA pieces = tile 0's first three K-steps)
        load_next()
        next_to_cur()
        take_next_offsets()
        load_next()
        a(f"s_mov_b32 s{S_KA}, 0")
        a(f"s_mov_b32 s{S_KB}, 0")
        a("s_nop 4")
        pgen = self.gen                                       # generations pgen .. pgen + 2: tile 0's K-steps 0, 1, 2
        seq = mimic if mimic is not None else ["dma"] * 36
        assert seq.count("dma") == 36
        n_p = 0
        for kind in seq:
            if kind == "dma":
                self.dma_piece(n_p % 12, n_p // 12, pgen + n_p // 12)
                n_p += 1
                if n_p % 12 == 0:
                    a(f"s_add_u32 s{S_KA}, s{S_KA}, 128")
                    a(f"s_add_u32 s{S_KB}, s{S_KB}, s{S_KSTEPB}")
            else:
                self.vmem("null", f"buffer_store_dword v{V_T}, v{V_CST}, s[{S_NULL}:{S_NULL + 3}], 0 offen")
        a(f"s_waitcnt vmcnt({min(63, self.younger(('dma', pgen)))})")
        a("s_barrier")
        self.first_frags(0, pgen)
        n_tail = (2 * NJ if bkm else NJ) + 7                  # the next K-step's first fragments: what every step's text ends with
        lds_tail = [(t[0],) + t[2:] for t in a.lds[-n_tail:]]
        self.gen = pgen

        # ---- tile loop
        l_tile, l_mid, l_last, l_done = a.label("tile"), a.label("mid"), a.label("last"), a.label("done")
        q = self.epilogue_queue()
        total = len(q)
        # MFMA slots that drain the queue: the first K-step's second phase issues its loads (the first phase carries the accumulator moves), the
        # arithmetic and the stores go into the other two K-steps of the first triple and the three of the last
        flat = self.o["flat"]
        n_slots = (flat - 1) * 64 if flat else 5 * 64
        quota = max(self.o["quota"] if not flat else 1, -(-(total - q.n_head) // (n_slots - 24)))            # (a margin of 24 slots: wait markers that turn into nothing still use their turn)
        self.quota = quota if not flat else round((total - q.n_head) / (n_slots - 16), 2)
        a(f"{l_tile}:")
        if not flat:
            a(f"s_sub_u32 s{S_CNT}, s{S_TRIP}, 3")
        self.region += 1
        head = Queue()
        head.items, q.items = q.items[:q.n_head], q.items[q.n_head:]
        self.step(0, first=True, q=head, quota=-(-q.n_head // 32), q_from_phase=1)
        assert not head.items
        if flat:
            # ---- loop-free form: K-steps 1 .. flat - 1 all drain the queue; the last three carry the next tile's first three K-steps in their DMA slots
            quota = Pacer(len(q), n_slots - 16)               # (the last 16 slots stay free: a margin, and the tile's last stores are not its very last instructions)
            for k in range(1, flat - 3):
                self.step(k % 3, q=q, quota=quota)
            next_to_cur()
            a(f"s_mov_b32 s{S_KA}, 0")
            a(f"s_mov_b32 s{S_KB}, 0")
            vm0 = len(a.vm)
            for k in range(flat - 3, flat):
                self.step(k % 3, q=q, quota=quota)
        else:
            self.step(1, q=q, quota=quota)
            self.step(2, q=q, quota=quota)
            self.close_loads()
            self.region += 1
            vm0 = len(a.vm)
            self.step(0)
            self.step(1)
            self.step(2)
            second_vm = [("dma" if isinstance(t, tuple) and t[0] == "dma" else t) for t in a.vm[vm0:]]
            assert lds_tail == [(t[0],) + t[2:] for t in a.lds[-n_tail:]]          # (the prologue's order of the first fragments is every step's)
            a(f"s_cmp_eq_u32 s{S_CNT}, 0")
            a(f"s_cbranch_scc1 {l_last}")
            a(f"{l_mid}:")
            vm0 = len(a.vm)
            self.step(0)
            self.step(1)
            self.step(2)
            assert second_vm == [("dma" if isinstance(t, tuple) and t[0] == "dma" else t) for t in a.vm[vm0:]]          # the loop sees the history its first entry sees
            assert lds_tail == [(t[0],) + t[2:] for t in a.lds[-n_tail:]]
            a(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
            a(f"s_cmp_eq_u32 s{S_CNT}, 0")
            a(f"s_cbranch_scc0 {l_mid}")
            a(f"{l_last}:")
            next_to_cur()                                         # last triple: its DMA slots carry the next tile's first three K-steps
            a(f"s_mov_b32 s{S_KA}, 0")
            a(f"s_mov_b32 s{S_KB}, 0")
            vm0 = len(a.vm)
            self.step(0, q=q, quota=quota)
            self.step(1, q=q, quota=quota)
            self.step(2, q=q, quota=quota)
        assert not q.items, (len(q.items), total, quota)
        assert lds_tail == [(t[0],) + t[2:] for t in a.lds[-n_tail:]]
        last_vm = a.vm[vm0:]
        k0 = next(i for i, t in enumerate(last_vm) if isinstance(t, tuple) and t[0] == "dma")
        self.last_triple_vm = ["dma" if (isinstance(t, tuple) and t[0] == "dma") else "x" for t in last_vm[k0:]]
        if mimic is not None:
            assert self.last_triple_vm == list(mimic), "the prologue does not replay the last triple's VMEM history"
        # ---- tile switch: the finished tile's output descriptors (its epilogue runs under the next K loop), the next tile's offsets
        a("s_nop 2")
        desc(S_RSC, S_C, S_CBYTES, S_CUR_C_OFF)
        a(f"s_lshl_b32 s{S_T + 2}, s{S_CUR_N0}, 2")
        desc(S_RSBIAS, S_BIAS, S_BIASBYTES, S_T + 2)
        if not self.ce:
            desc(S_RSR, S_R, S_RBYTES, S_CUR_R_OFF)
            a(f"s_mov_b32 s{S_EPAIR}, s{S_RUNPAIR}")
        else:                                                 # first row (the table's side-input slot carries it), first column, partial[m0][n0 / 64] of the finished tile
            a(f"s_mov_b32 s{S_EROW}, s{S_CUR_R_OFF}")
            a(f"s_mov_b32 s{S_EN0}, s{S_CUR_N0}")
            a(f"s_mul_i32 s{S_T}, s{S_EROW}, s{S_LDR}")
            a(f"s_lshr_b32 s{S_T + 1}, s{S_EN0}, 6")
            a(f"s_add_u32 s{S_T}, s{S_T}, s{S_T + 1}")
            a(f"s_lshl_b32 s{S_EPOFF}, s{S_T}, 2")
        a(f"s_sub_u32 s{S_TILE}, s{S_TILE}, 1")
        a(f"s_cmp_eq_u32 s{S_TILE}, 0")
        a(f"s_cbranch_scc1 {l_done}")
        take_next_offsets()
        load_next()
        a(f"s_branch {l_tile}")
        a(f"{l_done}:")
        # ---- the last tile's epilogue has no K loop to hide under
        self.region += 1
        self.loads_closed = set()
        a("s_nop 7")
        for r in range(128):
            a(f"v_accvgpr_mov_b32 a{ACC1 + r}, a{r}")
        a("s_nop 1")
        qf = self.epilogue_queue()
        # (fresh tags for the final queue's loads: younger() looks for the LAST operation carrying the tag, which is this region's)
        self.drain(qf, len(qf) + 1)
        a("s_waitcnt vmcnt(0) lgkmcnt(0)")
        return a.l


def generate(bkm, epi, opts=()):
    """two passes: the first learns the last triple's VMEM sequence, the second replays it in the prologue"""
    g1 = Gen(bkm, epi, opts)
    g1.body()
    g2 = Gen(bkm, epi, opts)
    lines = g2.body(mimic=g1.last_triple_vm)
    return lines, g2


def lint(lines, name):
    W.lint(lines, name)


BODIES = [(False, e) for e in ("plain", "resid", "mulaux", "dropres", "gelu", "gelud", "ceexp")] + [(True, e) for e in ("plain", "resid", "mulaux")]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
    opts = tuple(sys.argv[2:])
    with open(out, "w") as f:
        f.write("// GENERATED by scripts/gen_w4n.py -- do not edit; the schedule is described there\n")
        f.write(f"#define W4N_N_OPERANDS {len(OPS)}\n")
        f.write("// operand order: " + " ".join(OPS) + "\n")
        f.write("#define W4N_CLOBBERS " + ", ".join([f'"v{i}"' for i in range(V_LAST + 1)] + [f'"a{i}"' for i in range(256)] +
                                                    [f'"s{i}"' for i in S_EXTRA + list(range(S0, S_LAST + 1))] + ['"vcc"', '"scc"', '"m0"', '"memory"']) + "\n")
        for form, fopts in (("", ()), ("12", ("flat=12",))):          # the loop form (any K = 192 n >= 576) and the loop-free form for K = 768
            for bkm, epi in BODIES:
                lines, g = generate(bkm, epi, tuple(opts) + fopts)
                lint(lines, (bkm, epi))
                name = f"W4N_BODY{form}_{'KM' if bkm else 'KC'}_{epi.upper()}"
                f.write(f"#define {name} \\\n")
                f.write(" \\\n".join('    "%s\\n\\t"' % x for x in lines))
                f.write("\n")
                print(f"{name}: {len(lines)} asm lines, epilogue queue drained {g.quota} per MFMA slot", file=sys.stderr)


if __name__ == "__main__":
    main()
