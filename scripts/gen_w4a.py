#!/usr/bin/env python3
"""Generator of the hand-scheduled four-wave GEMM main loops (csrc/gemm_w4a_asm.inc), round 4.

Why a generator: each loop is ONE inline-asm statement per kernel -- persistent tile loop, software-pipelined K loop, epilogue -- with
explicit register numbers (accumulators in a[0:255], operand fragments in v[0:95]; nothing is left to the compiler's allocator or
scheduler: round-3 review item 2).  A few hundred MFMAs with their fillers and counted waits are not something to type by hand; this script
places them and COUNTS the waits (every `s_waitcnt lgkmcnt(n)` is derived from the issue order it has just generated).

Geometry (fixed): workgroup 256 threads = 4 waves as 2 x 2, tile 256 x 256 x 64, wave tile 128 x 128 = 8 x 8 fragments of 16 x 16,
`v_mfma_f32_16x16x32_bf16` with the operands swapped (D = Bfrag x Afrag: a lane holds 4 consecutive output columns), LDS image and
output mapping those of the 8-wave kernel in gemm.hip (same swizzle keys, same B-row permutation, same full-line stores and side-input
loads), so the two kernels produce the same results (the bias is added after the K loop here: last-bit differences).

Variants (one asm body each): B operand k-contiguous (KC: nn.Linear forward) or k-major (KM: input gradients, read with
ds_read_b64_tr_b16) x epilogue `plain` (+ bias), `resid` (+ bias + residual R), `mulaux` (x aux, the GELU' factor left by the forward), `dropres` (k-contiguous B
only: dropout(acc + bias) + R, the mask regenerated bit for bit from common.h's pair hash), `gelu` (k-contiguous B only: C = GELU(acc + bias), the FFN lin1
forward of a forward-only call) and `gelud` (C = GELU(u), second output aux = GELU'(u), u = acc + bias: the training forward; the instruction sequence is the one
the compiler emits for common.h's gelu_fast_with_grad4 -- Abramowitz-Stegun 7.1.26 on packed fp32 -- so both kernels produce the same values).

LDS (bytes): A stage 0 [0, 32K), A stage 1 [32K, 64K), B stage 0 [64K, 96K), B stage 1 [96K, 128K), tile table [128K, +16K: 512 tiles x {16 B of A / B / C offsets + first column, 16 B holding the side-input offset}).

Pipeline per K-step k (stage s = k & 1), two phases of 64 MFMAs (kk = 0, 1), group i = the 8 MFMAs of A fragment i:
  phase 0: MFMAs on B(kk0) x A(kk0)[i]; the B(kk1) set is read into the second B buffer during groups 0-3; A(kk1)[i] is read INTO A(kk0)[i]'s
           registers half a group after group i (A fragments need one buffer, B fragments two: 96 fragment registers).
  phase 1: MFMAs on B(kk1) x A(kk1)[i].  At group 2 every read of stage s has returned (lgkmcnt 0) -> barrier M -> stage s is free: the
           LDS-DMA of K-step k+2 goes into it, 8 pieces before and 8 after barrier E.  At group 5: `s_waitcnt vmcnt(N)`, N = the VMEM
           operations younger than K-step k+1's last DMA piece (the counter is in order) -> barrier E -> stage s^1 (K-step k+1) is visible: its
           B(kk0) set and A(kk0)[0..6] are read during groups 5-7 into registers phase 1 has finished with ([7]: first group of the next step).
  A K-step's DMA has 1.1-1.3 K-steps to land with 128 KB of LDS (the 8-wave kernel: 1.0, and its K-step then waits for the slowest piece).
Tiles follow each other without a prologue: during a tile's last two K-steps the DMA slots carry the NEXT tile's first two K-steps
(descriptor swap), and the epilogue's stores are YOUNGER than every load the next K loop waits for (gfx950 has one in-order vmcnt for
loads and stores: a wait never has to sit out a store's ~2 us acknowledge unless it asks for something issued after it).
Side inputs (`resid`, `mulaux`): the wave's 128 x 128 tile of R / aux (128 registers) is requested during the tile's LAST K-step, into
registers that step has finished with (the first B buffer, the A fragments as their groups retire) and 64 high registers -- before any store of
the epilogue, so its wait does not sit behind one; those variants re-read the next tile's first fragments after the epilogue instead.
"""
import sys

NW = 4
A_ST = [0, 32768]
B_ST = [65536, 98304]

# ---- fixed registers -----------------------------------------------------------------------------------------------
V_B = [0, 32]            # B fragment buffers (kk parity), 8 fragments x 4
V_A = 64                 # A fragments, 8 x 4
V_BIAS = 96              # 32: bias quads of the wave's 8 B fragments (added in the epilogue)
V_T = 128                # temporaries (24)
V_VOA = 152              # 8 DMA source offsets (A pieces)
V_VOB = 160              # 8 (B pieces)
V_AA = [168, 169]        # A fragment address kk0 / kk1
V_AB = [170, 171, 172, 173]   # B fragment addresses: KC kk0 / kk1; KM column pair 0..3 (B region base folded in)
V_CST = 174              # store lane offset (bytes)
V_BOFF = 175             # bias lane offset (bytes)
V_TBL = 176              # LDS address of the tile table (same in every lane)
V_RST = 177              # side-input lane offset (bytes)
V_PAIRB = 242            # dropout: the lane's element-pair index inside the tile
V_HOLD = 178             # defer_stores: 8 units x 8 packed registers (v178..v241)
S_PREVC = 76             # defer_stores: the previous tile's C descriptor (s76..s79 = S_RSR, the side-input descriptor: plain bodies have no side input; a buffer resource must be 4-aligned --
                         # round 5 had it in s15..s18, which the emulator and the race checker accept and the ASSEMBLER refuses: lint() now checks the alignment)
V_GC = 178               # gelu / gelud: 8 constant pairs (both halves the same value) in v178..v193
V_GX = 210               # gelu / gelud: the column group's second fragment (4 values) and the scratch pairs of its two pairs (v210..v225)
V_T2 = 198               # gelud: the second output's 8 packed registers (v198..v205) and its 4 exchange temporaries (v206..v209)
GELU_CONSTS = [0.3275911 * 0.70710678118654752, -0.72134752044448170, 1.061405429, -1.453152027, 1.421413741, -0.284496736, 0.254829592, 0.3989422804014327]
GC_K1, GC_NC, GC_A5, GC_A4, GC_A3, GC_A2, GC_A1, GC_PHI = range(8)
V_LAST = 242             # (v243..v255 stay with the compiler: the statement's ten vector operands live there)
R_BLOCK = [0, 16, 178, 194, 210, 226, 64, 80]     # first register of side-input block i (16 registers: slab 0 {rows 0-7, rows 8-15}, slab 1 {..})

S0 = 24
S_ARG = 24               # 24..39: A, B, C, bias, R (lo, hi each), M, N, K, lda, ldb, ldc
S_A, S_Bp, S_C, S_BIAS, S_R, S_M, S_N, S_Kd, S_LDA, S_LDB, S_LDC = 24, 26, 28, 30, 32, 34, 35, 36, 37, 38, 39
S_LDR, S_ABYTES, S_BBYTES, S_CBYTES, S_RBYTES, S_KSTEPB, S_NPAIRS, S_LDA64, S_LDBP, S_LDC2, S_LDR2, S_BIASBYTES = 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51
S_RSA, S_RSB, S_NXA, S_NXB, S_RSC, S_RSBIAS, S_RSR, S_NULL = 52, 56, 60, 64, 68, 72, 76, 80
S_NXC_OFF, S_NXN0, S_CUR_C_OFF, S_CUR_N0, S_KA, S_KB, S_PAIRS, S_TILE, S_TIDX, S_M0A, S_M0B, S_LDC16, S_LDC8, S_SOFF = range(84, 98)
S_T = 98                 # 98..101 scratch
S_LDR16, S_LDR8, S_NXPAIR, S_NXR_OFF, S_CUR_R_OFF = 19, 20, 21, 22, 23     # (below S0: listed separately in the clobbers)
S_N8, S_HC2, S_HC1, S_CURPAIR = 15, 16, 17, 18           # dropout: 8 N (pairs per 16 rows), the hash multipliers, the tile's first pair index
S_EXTRA = [15, 16, 17, 18, 19, 20, 21, 22, 23]
S_LAST = 101

OPS = ["tbl", "voA0", "voBbase", "chunkx", "aA0", "aB0", "cst", "boff", "rst", "pairb",  # "v"
       "karg", "ntiles", "m0A", "m0B", "dkey", "dthr", "dinv"]                         # "s" (karg: 64-bit)
OP = {n: f"%{i}" for i, n in enumerate(OPS)}


class Asm:
    def __init__(self):
        self.l = []
        self.lds = []          # tags of the LDS reads issued so far, in order
        self.vm = []           # tags of the VMEM operations (loads, LDS-DMA pieces, stores) issued so far, in order: one in-order vmcnt
        self.uid = 0

    def __call__(self, s):
        self.l.append(s)

    def read(self, tag, text, n=1):
        for _ in range(n):
            self.lds.append(tag)
        self(text)

    def wait_lds(self, *tags):
        """wait until the reads tagged `tags` have returned: LDS reads return in order, so everything issued after the youngest of them may
        stay outstanding.  The counter has 4 bits: with 15 or more younger reads the hardware has already stalled until the target returned."""
        assert all(t in self.lds for t in tags), tags
        idx = max(i for i, t in enumerate(self.lds) if t in tags)
        n = len(self.lds) - 1 - idx
        if n < 15:
            self(f"s_waitcnt lgkmcnt({n})")

    def vmem(self, tag, text):
        self.vm.append(tag)
        self(text)

    def younger_vm(self, tag):
        """VMEM operations issued after the last one tagged `tag` (straight-line code only: the caller knows nothing older is in a loop body)"""
        idx = max(i for i, t in enumerate(self.vm) if t == tag)
        return len(self.vm) - 1 - idx

    def label(self, stem):
        self.uid += 1
        return f"L{stem}{self.uid}_%="


def acc(i, j):
    return 4 * (8 * i + j)


def mfma(a, i, j, kk, first):
    d = acc(i, j)
    b = V_B[kk] + 4 * j
    av = V_A + 4 * i
    c = "0" if first else f"a[{d}:{d + 3}]"
    a(f"v_mfma_f32_16x16x32_bf16 a[{d}:{d + 3}], v[{b}:{b + 3}], v[{av}:{av + 3}], {c}")


class Gen:
    def __init__(self, bkm, epi, ni=8, opts=()):
        """ni: A fragments per wave = tile height / 32 (8: 256-row tiles; 7: 224-row tiles -- 17 408 tokens are 78 x 224: 234 / 702 / 936 tiles fill
        1 / 3 / 4 rounds of 256 CUs where 204 / 612 / 816 tiles of 256 rows leave the last round 20-80 % empty)"""
        assert ni in (7, 8)
        self.bkm, self.epi, self.ni = bkm, epi, ni
        self.a = Asm()
        self.gen = 0
        self.side = epi in ("resid", "mulaux", "dropres")
        self.drop = epi == "dropres"
        self.gelu = epi in ("gelu", "gelud")
        self.two_out = epi == "gelud"
        # measurement options (NOT in the shipped .inc; `python scripts/gen_w4a.py OUT block_waits early_side` writes a variant file for scripts/build_variant.sh):
        #   block_waits: the epilogue waits for side-input block i in front of row block i instead of for blocks 0-6 in front of row block 0
        #   early_side:  the side-input blocks that live in registers the K loop never uses (2-5: v178..v241) are requested one K-step earlier
        #   defer_stores (plain bodies): the packed output of the tile's last four row blocks waits in v178..v241 and is stored from the MFMA slots of the NEXT
        #                tile's first K-step (the epilogue's store burst halves: 64 KB per CU instead of 128 KB in front of the next K loop's loads)
        self.opts = frozenset(opts)
        assert self.opts <= {"block_waits", "early_side", "defer_stores"}
        self.defer = "defer_stores" in self.opts and epi == "plain"

    # ---------------------------------------------------------------- fragment reads / DMA
    def read_b(self, j, kk, stage, gen):
        a = self.a
        r = V_B[kk] + 4 * j
        st = B_ST[stage] - B_ST[0]
        if not self.bkm:
            a.read(("B", gen, kk, j), f"ds_read_b128 v[{r}:{r + 3}], v{V_AB[kk]} offset:{st + (32 * (j >> 1) + 4 * (j & 1)) * 128}")
        else:   # k-major tile [64 k][512 B]: two transpose reads (k rows 8g+{0..3} and +4), column pair j>>1 has its own address, parity = +8 B
            off = st + kk * 32 * 512 + (j & 1) * 8
            a.read(("B", gen, kk, j), f"ds_read_b64_tr_b16 v[{r}:{r + 1}], v{V_AB[j >> 1]} offset:{off}")
            a.read(("B", gen, kk, j), f"ds_read_b64_tr_b16 v[{r + 2}:{r + 3}], v{V_AB[j >> 1]} offset:{off + 4 * 512}")

    def read_a(self, i, kk, stage, gen):
        r = V_A + 4 * i
        self.a.read(("A", gen, kk, i), f"ds_read_b128 v[{r}:{r + 3}], v{V_AA[kk]} offset:{A_ST[stage] + i * 2048}")

    def dma_piece(self, p, stage):
        a = self.a
        if p < self.ni:                                       # A: ni pieces of 32 rows (4 waves x 8 rows x 128 B)
            a(f"s_add_u32 m0, s{S_M0A}, {A_ST[stage] + p * NW * 1024}")
            a("s_nop 0")
            a.vmem("dma", f"buffer_load_dwordx4 v{V_VOA + p}, s[{S_RSA}:{S_RSA + 3}], s{S_KA} offen lds")
        else:
            q = p - self.ni
            a(f"s_add_u32 m0, s{S_M0B}, {(B_ST[stage] - B_ST[0]) + q * 1024}")
            a("s_nop 0")
            a.vmem("dma", f"buffer_load_dwordx4 v{V_VOB + q}, s[{S_RSB}:{S_RSB + 3}], s{S_KB} offen lds")

    def side_load(self, blk, n):
        """load n (0..3) of side-input block blk: slab n >> 1, row half n & 1"""
        r = R_BLOCK[blk] + 4 * n
        a = self.a
        a(f"s_mul_i32 s{S_T}, s{S_LDR16}, {blk}")
        a(f"s_add_u32 s{S_T}, s{S_T}, {128 * (n >> 1)}")
        if n & 1:
            a(f"s_add_u32 s{S_T}, s{S_T}, s{S_LDR8}")
        a.vmem(("side", blk), f"buffer_load_dwordx4 v[{r}:{r + 3}], v{V_RST}, s[{S_RSR}:{S_RSR + 3}], s{S_T} offen")

    def gelu_pairs(self, X, D, G, P):
        """register pairs X[k] (u = acc + bias) -> GELU(u) in place; gelud: GELU'(u) left in D[k].  The arithmetic of common.h gelu_fast_parts2 /
        gelu_fast_with_grad4, instruction for instruction as the compiler emits it for the 8-wave kernel; the len(X) pairs are interleaved
        (one wave per SIMD: nothing else hides a dependent instruction's latency, and a transcendental's result may not be read by the very
        next VALU instruction on gfx940+).  D, G, P: scratch pairs."""
        a = self.a
        pr = lambda r: f"v[{r}:{r + 1}]"
        gc = lambda n: pr(V_GC + 2 * n)
        R2 = range(len(X))
        for k in R2:
            a(f"v_and_b32 v{D[k]}, 0x7fffffff, v{X[k]}")
            a(f"v_and_b32 v{D[k] + 1}, 0x7fffffff, v{X[k] + 1}")
        for k in R2:
            a(f"v_pk_fma_f32 {pr(D[k])}, {pr(D[k])}, {gc(GC_K1)}, 1.0 op_sel_hi:[1,1,0]")       # 1 + p |u| / sqrt 2
        for k in R2:
            a(f"v_pk_mul_f32 {pr(G[k])}, {pr(X[k])}, {pr(X[k])}")
        for k in R2:
            a(f"v_rcp_f32 v{D[k]}, v{D[k]}")                                                    # t
            a(f"v_rcp_f32 v{D[k] + 1}, v{D[k] + 1}")
        for k in R2:
            a(f"v_pk_mul_f32 {pr(G[k])}, {pr(G[k])}, {gc(GC_NC)}")
        for k in R2:
            a(f"v_pk_fma_f32 {pr(P[k])}, {pr(D[k])}, {gc(GC_A5)}, {gc(GC_A4)}")
        for k in R2:
            a(f"v_exp_f32 v{G[k]}, v{G[k]}")                                                    # e^{-u^2 / 2}
            a(f"v_exp_f32 v{G[k] + 1}, v{G[k] + 1}")
        for c in (GC_A3, GC_A2, GC_A1):
            for k in R2:
                a(f"v_pk_fma_f32 {pr(P[k])}, {pr(P[k])}, {pr(D[k])}, {gc(c)}")
        for k in R2:
            a(f"v_pk_mul_f32 {pr(P[k])}, {pr(P[k])}, {pr(D[k])} neg_lo:[0,1] neg_hi:[0,1]")     # -(poly t)
        for k in R2:
            a(f"v_pk_fma_f32 {pr(P[k])}, {pr(P[k])}, {pr(G[k])}, 1.0 op_sel_hi:[1,1,0]")        # erf(|u| / sqrt 2)
        for k in R2:
            a(f"v_bfi_b32 v{P[k]}, s{S_HC1}, v{P[k]}, v{X[k]}")                                 # copysign(., u)
            a(f"v_bfi_b32 v{P[k] + 1}, s{S_HC1}, v{P[k] + 1}, v{X[k] + 1}")
        for k in R2:
            a(f"v_pk_fma_f32 {pr(P[k])}, {pr(P[k])}, 0.5, 0.5 op_sel_hi:[1,0,0]")               # Phi(u)
        if self.two_out:
            for k in R2:
                a(f"v_pk_mul_f32 {pr(D[k])}, {pr(X[k])}, {gc(GC_PHI)}")
            for k in R2:
                a(f"v_pk_fma_f32 {pr(D[k])}, {pr(D[k])}, {pr(G[k])}, {pr(P[k])}")               # Phi(u) + u phi(u)
        for k in R2:
            a(f"v_pk_mul_f32 {pr(X[k])}, {pr(X[k])}, {pr(P[k])}")

    def flush_unit(self, u):
        """deferred stores of unit u = 2 (row block - (ni - 4)) + slab: two full-line stores through the PREVIOUS tile's C descriptor (s15..s18)"""
        a = self.a
        i, slab = self.ni - 4 + (u >> 1), u & 1
        H = V_HOLD + 8 * u
        a(f"s_mul_i32 s{S_T}, s{S_LDC16}, {i}")
        a(f"s_add_u32 s{S_T}, s{S_T}, {128 * slab}")
        a.vmem("store", f"buffer_store_dwordx4 v[{H}:{H + 3}], v{V_CST}, s[{S_PREVC}:{S_PREVC + 3}], s{S_T} offen")
        a(f"s_add_u32 s{S_T}, s{S_T}, s{S_LDC8}")
        a.vmem("store", f"buffer_store_dwordx4 v[{H + 4}:{H + 7}], v{V_CST}, s[{S_PREVC}:{S_PREVC + 3}], s{S_T} offen")

    # ---------------------------------------------------------------- one K-step
    def step(self, stage, first=False, n_e=8, bias_loads=False, last_of_tile=False, early_side=False):
        """first: accumulators start from 0.  n_e: vmcnt count of barrier E (None: no wait, only the barrier).  bias_loads: the tile's 8 bias
        quads ride in phase 0.  last_of_tile (side-input variants): no prefetch of the next tile's fragments; the side tile is requested."""
        a = self.a
        ni = self.ni
        g0 = self.gen
        self.gen += 1
        fill = {}

        def put(key, f):
            fill.setdefault(key, []).append(f)
        for j in range(8):
            put((j // 2, 2 + j % 2), lambda j=j: self.read_b(j, 1, stage, g0))      # (not the first slots: the previous phase's last MFMAs read this buffer)
        put((0, 5), lambda: self.read_a(ni - 1, 0, stage, g0))                        # the last A(kk0) fragment (its registers were busy until now)
        for i in range(ni - 1):
            put((i + 1, 5), lambda i=i: self.read_a(i, 1, stage, g0))                 # A(kk1)[i] over A(kk0)[i], half a group after its MFMAs
        if first and self.defer:
            free = [(g, sl) for g in range(ni) for sl in (4, 6, 7)][:8]
            for u, key in enumerate(free):
                put(key, lambda u=u: self.flush_unit(u))
        if bias_loads:
            for j in range(8):
                off = (32 * (j >> 1) + 4 * (j & 1)) * 4
                put((ni - 4 + j // 2, j % 2), lambda j=j, off=off: a.vmem("bias",
                    f"buffer_load_dwordx4 v[{V_BIAS + 4 * j}:{V_BIAS + 4 * j + 3}], v{V_BOFF}, s[{S_RSBIAS}:{S_RSBIAS + 3}], 0 offen offset:{off}"))
        for i in range(ni):
            if i == 0:
                a.wait_lds(("B", g0, 0, 7), ("A", g0, 0, 0))
            else:
                a.wait_lds(("A", g0, 0, i))
            for j in range(8):
                mfma(a, i, j, 0, first)
                for f in fill.get((i, j), []):
                    f()
        # ---------------- phase 1
        side_q = []
        if last_of_tile and self.side:
            # side-input blocks 0-5 (24 loads) into the first B buffer and the high registers: before barrier E; block 6 (A[0..3]) after group 4
            side_q = [(blk, n) for blk in ((0, 1) if "early_side" in self.opts else range(6)) for n in range(4)]
        if early_side and self.side:
            side_q = [(blk, n) for blk in (2, 3, 4, 5) for n in range(4)]
        n_side_before_e = len(side_q)
        # DMA pieces of K-step k+2 behind barrier E (8 went out in front of it): (group, slot)
        after_e = ([(6, 5), (6, 6), (6, 7), (7, 1), (7, 2), (7, 3), (7, 4), (7, 5)] if ni == 8 else
                   [(5, 1), (5, 3), (5, 5), (5, 7), (6, 5), (6, 6), (6, 7)])
        assert len(after_e) == ni + 8 - 8
        for i in range(ni):
            slots = {}

            def sput(j, f):
                slots.setdefault(j, []).append(f)
            if i == 0:
                a.wait_lds(("B", g0, 1, 7), ("A", g0, 1, 0))
                sput(5, lambda: self.read_a(ni - 1, 1, stage, g0))
            elif i == 1:
                a.wait_lds(("A", g0, 1, 1))
            elif i == 2:
                a("s_waitcnt lgkmcnt(0)")                     # every read of this stage is back ...
                a("s_barrier")                                # ... barrier M: in every wave -> the stage is free
            if i <= 4 and side_q:
                for s_ in (0, 2, 4, 6, 7)[: (5 if i < 4 else 4)]:
                    if side_q:
                        blk, n = side_q.pop(0)
                        sput(s_, lambda blk=blk, n=n: self.side_load(blk, n))
            if 2 <= i <= 4:                                   # DMA of K-step k+2, pieces 0-7 before barrier E
                lo, hi = {2: (0, 3), 3: (3, 6), 4: (6, 8)}[i]
                for n, p in enumerate(range(lo, hi)):
                    sput(1 + 2 * n, lambda p=p: self.dma_piece(p, stage))
            if i == 5:
                assert not side_q
                if n_e is not None:
                    # K-step k+1 has landed (this wave's pieces) ...  (the counter has 6 bits: behind a two-output epilogue more than 63 operations are
                    # younger, and the strongest expressible wait also asks for that epilogue's first stores -- issued microseconds earlier)
                    a(f"s_waitcnt vmcnt({min(63, n_e + n_side_before_e)})")
                a("s_barrier")                                # ... barrier E: everybody's
                if not (last_of_tile and self.side):
                    for j in range(8):
                        sput(j, lambda j=j: self.read_b(j, 0, stage ^ 1, g0 + 1))
                else:
                    for n in range(4):
                        sput(2 * n, lambda n=n: self.side_load(6, n))
            if i == 6:
                if not (last_of_tile and self.side):
                    for q in range(5):
                        sput(q, lambda q=q: self.read_a(q, 0, stage ^ 1, g0 + 1))
            if i == 7:
                if not (last_of_tile and self.side):
                    sput(0, lambda: self.read_a(5, 0, stage ^ 1, g0 + 1))
            for n, (gi, sl) in enumerate(after_e):
                if gi == i:
                    sput(sl, lambda p=8 + n: self.dma_piece(p, stage))
            for j in range(8):
                mfma(a, i, j, 1, False)
                for f in slots.get(j, []):
                    f()
        if not (last_of_tile and self.side):
            self.read_a(ni - 2, 0, stage ^ 1, g0 + 1)
        elif ni == 8:
            for n in range(4):
                self.side_load(7, n)
        a(f"s_add_u32 s{S_KA}, s{S_KA}, 128")
        a(f"s_add_u32 s{S_KB}, s{S_KB}, s{S_KSTEPB}")

    def first_frags(self, stage):
        """B(kk0) set and A(kk0)[0..6] of a tile's first K-step (kernel prologue; side-input variants: after every epilogue)"""
        g0 = self.gen
        for j in range(8):
            self.read_b(j, 0, stage, g0)
        for i in range(self.ni - 1):
            self.read_a(i, 0, stage, g0)

    # ---------------------------------------------------------------- the whole body
    def body(self):
        a = self.a
        bkm = self.bkm
        ni = self.ni
        a("s_nop 4")
        a(f"s_load_dwordx16 s[{S_ARG}:{S_ARG + 15}], {OP['karg']}, 0")
        a(f"s_load_dword s{S_LDR}, {OP['karg']}, 64")
        a(f"v_mov_b32 v{V_TBL}, {OP['tbl']}")
        a(f"v_mov_b32 v{V_CST}, {OP['cst']}")
        a(f"v_mov_b32 v{V_BOFF}, {OP['boff']}")
        a(f"v_mov_b32 v{V_RST}, {OP['rst']}")
        a(f"v_mov_b32 v{V_PAIRB}, {OP['pairb']}")
        if self.gelu:
            import struct
            a(f"s_mov_b32 s{S_HC1}, 0x7fffffff")              # (the dropout multipliers' registers: no dropout in these variants) copysign mask
            for n, c in enumerate(GELU_CONSTS):
                bits = struct.unpack("<I", struct.pack("<f", c))[0]
                a(f"v_mov_b32 v{V_GC + 2 * n}, 0x{bits:08x}")
                a(f"v_mov_b32 v{V_GC + 2 * n + 1}, 0x{bits:08x}")
        else:
            a(f"s_mov_b32 s{S_HC1}, 0x7feb352d")
            a(f"s_mov_b32 s{S_HC2}, 0x846ca68b")
        a(f"v_mov_b32 v{V_AA[0]}, {OP['aA0']}")
        a(f"v_xor_b32 v{V_AA[1]}, 64, {OP['aA0']}")
        a(f"v_mov_b32 v{V_AB[0]}, {OP['aB0']}")
        if not bkm:
            a(f"v_xor_b32 v{V_AB[1]}, 64, {OP['aB0']}")
        else:
            for pp in range(1, 4):                            # column pair pp: chunk index + 4 pp (bits the swizzle key does not carry into) -> address ^ (pp << 6)
                a(f"v_xor_b32 v{V_AB[pp]}, {pp << 6}, {OP['aB0']}")
        a(f"s_mov_b32 s{S_M0A}, {OP['m0A']}")
        a(f"s_mov_b32 s{S_M0B}, {OP['m0B']}")
        a(f"s_mov_b32 s{S_TILE}, {OP['ntiles']}")
        a("s_waitcnt lgkmcnt(0)")
        for ptr in (S_A, S_Bp, S_C, S_BIAS, S_R):
            a(f"s_and_b32 s{ptr + 1}, s{ptr + 1}, 0xffff")
        # derived scalars
        a(f"s_lshl_b32 s{S_LDA64}, s{S_LDA}, 6")
        a(f"s_lshl_b32 s{S_LDBP}, s{S_LDB}, {2 if bkm else 4}")          # bytes between a wave's consecutive B pieces (KM: 2 k-rows, KC: 8 rows)
        a(f"s_lshl_b32 s{S_LDC2}, s{S_LDC}, 1")
        a(f"s_lshl_b32 s{S_LDR2}, s{S_LDR}, 1")
        a(f"s_lshl_b32 s{S_LDC8}, s{S_LDC}, 4")
        a(f"s_lshl_b32 s{S_LDC16}, s{S_LDC}, 5")
        a(f"s_lshl_b32 s{S_LDR8}, s{S_LDR}, 4")
        a(f"s_lshl_b32 s{S_LDR16}, s{S_LDR}, 5")
        a(f"s_lshr_b32 s{S_NPAIRS}, s{S_Kd}, 7")
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
        bytes_of(S_CBYTES, S_M, S_LDC, S_N)
        bytes_of(S_RBYTES, S_M, S_LDR, S_N)
        a(f"s_or_b32 s{S_T}, s{S_BIAS}, s{S_BIAS + 1}")
        a(f"s_cmp_eq_u32 s{S_T}, 0")
        a(f"s_cselect_b32 s{S_BIASBYTES}, 0, s{S_BIASBYTES}")             # no bias: zero records -> every bias load reads 0
        a(f"s_or_b32 s{S_T}, s{S_R}, s{S_R + 1}")
        a(f"s_cmp_eq_u32 s{S_T}, 0")
        a(f"s_cselect_b32 s{S_RBYTES}, 0, s{S_RBYTES}")
        # DMA source offsets
        a(f"v_mov_b32 v{V_VOA}, {OP['voA0']}")
        for p in range(1, 8):                                  # A pieces are 32 rows apart (piece q = 4 p + wave), same swizzle key
            a(f"v_add_u32 v{V_VOA + p}, v{V_VOA + p - 1}, s{S_LDA64}")
        for j in range(8):
            # KC: piece j covers rows 8 j further, key bits 1-2 = j & 3;  KM: k-rows 2 j further, key = 4 (j & 1) + 8 ((j >> 2) & 1) (+ 2 r1 in chunkx)
            kx = (4 * (j & 1) + 8 * ((j >> 2) & 1)) if bkm else 2 * (j & 3)
            a(f"v_xor_b32 v{V_T}, {kx}, {OP['chunkx']}")
            a(f"v_lshlrev_b32 v{V_T}, 4, v{V_T}")
            a(f"s_mul_i32 s{S_T}, s{S_LDBP}, {j}")
            a(f"v_add3_u32 v{V_VOB + j}, {OP['voBbase']}, v{V_T}, s{S_T}")
        a(f"s_mov_b32 s{S_NULL}, 0")
        a(f"s_mov_b32 s{S_NULL + 1}, 0")
        a(f"s_mov_b32 s{S_NULL + 2}, 0")
        a(f"s_mov_b32 s{S_NULL + 3}, 0x00020000")
        if self.defer:                                        # nothing is held in front of the first tile: its flush goes through a null descriptor
            for k in range(4):
                a(f"s_mov_b32 s{S_PREVC + k}, s{S_NULL + k}")
        a(f"s_mov_b32 s{S_TIDX}, 0")

        def load_next():
            """table entry S_TIDX -> next-tile descriptors (null past the end); entry = {a_off, b_off, c_off, n0}; side-input offset = c_off scaled"""
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

        def cur_output_descriptors():
            """C / bias / side-input descriptors of the tile whose K loop starts now"""
            desc(S_RSC, S_C, S_CBYTES, S_CUR_C_OFF)
            a(f"s_lshl_b32 s{S_T + 2}, s{S_CUR_N0}, 2")
            desc(S_RSBIAS, S_BIAS, S_BIASBYTES, S_T + 2)
            if not self.defer:                                # (defer_stores keeps the held tile's C descriptor in these registers)
                desc(S_RSR, S_R, S_RBYTES, S_CUR_R_OFF)

        def take_next_offsets():
            a(f"s_mov_b32 s{S_CUR_C_OFF}, s{S_NXC_OFF}")
            a(f"s_mov_b32 s{S_CUR_N0}, s{S_NXN0}")
            a(f"s_mov_b32 s{S_CUR_R_OFF}, s{S_NXR_OFF}")
            if not self.defer:                                # (s18 is part of the held tile's descriptor there)
                a(f"s_mov_b32 s{S_CURPAIR}, s{S_NXPAIR}")

        # ---- kernel prologue: tile 0's descriptors, its first two K-steps, 32 null stores (the first K-step's vmcnt count assumes an epilogue
        # before it; side-input variants: + the 4 loads of side block 7), its first fragments
        load_next()
        next_to_cur()
        take_next_offsets()
        load_next()
        cur_output_descriptors()
        a(f"s_mov_b32 s{S_KA}, 0")
        a(f"s_mov_b32 s{S_KB}, 0")
        a("s_nop 4")
        for p in range(ni + 8):
            self.dma_piece(p, 0)
        a(f"s_add_u32 s{S_KA}, s{S_KA}, 128")
        a(f"s_add_u32 s{S_KB}, s{S_KB}, s{S_KSTEPB}")
        for p in range(ni + 8):
            self.dma_piece(p, 1)
        a(f"s_add_u32 s{S_KA}, s{S_KA}, 128")
        a(f"s_add_u32 s{S_KB}, s{S_KB}, s{S_KSTEPB}")
        a(f"s_waitcnt vmcnt({ni + 8})")
        a("s_barrier")
        # VMEM operations between a tile's last DMA piece and the next tile's first K-step: the epilogue's stores (4 per A fragment row) and, with
        # 8 side blocks, the four loads of the block requested behind the last piece (checked against the generated order below)
        n_epi_vm = 4 * ni * (2 if self.two_out else 1) + (4 if (self.side and ni == 8) else 0) - (16 if self.defer else 0)
        for _ in range(n_epi_vm):
            a(f"buffer_store_dword v{V_T}, v{V_CST}, s[{S_NULL}:{S_NULL + 3}], 0 offen")
        a("s_nop 1")
        if not self.side:
            self.first_frags(0)

        # ---- tile loop
        l_tile, l_mid, l_last, l_done = a.label("tile"), a.label("mid"), a.label("last"), a.label("done")
        a(f"{l_tile}:")
        if self.side:
            self.first_frags(0)                               # (the last K-step of the previous tile used these registers for its side tile)
        a(f"s_mov_b32 s{S_PAIRS}, s{S_NPAIRS}")
        # first pair.  vmcnt of its first barrier E: younger than this tile's second K-step (issued during the previous tile's last step) are
        # the previous epilogue's VMEM operations, the 8 bias loads of phase 0 and the 8 DMA pieces issued before the barrier
        self.step(0, first=True, n_e=n_epi_vm + (16 if self.defer else 0) + 8 + 8, bias_loads=True)
        self.step(1)
        tail_state = [(t[0],) + t[2:] for t in a.lds[-24:]]
        a(f"s_sub_u32 s{S_PAIRS}, s{S_PAIRS}, 1")
        a(f"s_cmp_eq_u32 s{S_PAIRS}, 1")
        a(f"s_cbranch_scc1 {l_last}")
        a(f"{l_mid}:")
        self.step(0)
        self.step(1)
        assert tail_state == [(t[0],) + t[2:] for t in a.lds[-24:]]
        a(f"s_sub_u32 s{S_PAIRS}, s{S_PAIRS}, 1")
        a(f"s_cmp_eq_u32 s{S_PAIRS}, 1")
        a(f"s_cbranch_scc0 {l_mid}")
        a(f"{l_last}:")
        next_to_cur()                                         # last pair: its DMA slots carry the next tile's first two K-steps
        a(f"s_mov_b32 s{S_KA}, 0")
        a(f"s_mov_b32 s{S_KB}, 0")
        self.step(0, early_side="early_side" in self.opts)
        self.step(1, last_of_tile=True)
        if not self.side:
            assert tail_state == [(t[0],) + t[2:] for t in a.lds[-24:]]

        # ---- epilogue of the finished tile (descriptor S_RSC still its own)
        a("s_nop 15")
        a(f"s_mov_b32 s{S_SOFF}, 0")
        T = V_T
        n_store = 0
        for i in range(ni):
            if self.side and "block_waits" in self.opts:
                a(f"s_waitcnt vmcnt({min(63, a.younger_vm(('side', i)))})")       # side block i is back
            elif self.side:
                if i == 0:
                    n_y = a.younger_vm(("side", 6))
                    assert n_y == (12 if ni == 8 else 4), n_y
                    a(f"s_waitcnt vmcnt({n_y})")              # side blocks 0-6 are back (younger: the DMA pieces issued behind block 6, block 7)
                if i == 7:
                    assert a.younger_vm(("side", 7)) == n_store
                    a(f"s_waitcnt vmcnt({n_store})")          # block 7 (everything older than the stores issued so far)
            for slab in range(2):
                if self.side:
                    # the lane's side chunks of this (block, slab): rows 0-7 / 8-15 in line order -> q' = 0 / 1 chunks of row t (swap with lane t ^ 8)
                    L0, L1 = R_BLOCK[i] + 8 * slab, R_BLOCK[i] + 8 * slab + 4
                    for r in range(4):
                        a(f"v_mov_b32 v{T + 16 + r}, v{L0 + r}")
                    a("s_nop 1")
                    for r in range(4):
                        a(f"v_mov_b32_dpp v{L0 + r}, v{L1 + r} row_ror:8 row_mask:0xf bank_mask:0xc")
                    for r in range(4):
                        a(f"v_mov_b32_dpp v{L1 + r}, v{T + 16 + r} row_ror:8 row_mask:0xf bank_mask:0x3")
                # P0 / P1: the lane's 8 consecutive columns of column groups q' = 0 / 1 of the slab (fragments 4 slab + 2 q' + e, e = 0, 1)
                for qp in range(2):
                    if self.gelu:
                        # both fragments (e = 0, 1) of the column group at once: four independent pairs through the GELU arithmetic
                        XB = V_GX                                  # second fragment's values; scratch pairs of pairs 2, 3 behind it
                        for e in range(2):
                            j = 4 * slab + 2 * qp + e
                            d = acc(i, j)
                            x0 = T + 8 if e == 0 else XB
                            for r in range(4):
                                a(f"v_accvgpr_read_b32 v{x0 + r}, a{d + r}")
                            a(f"v_pk_add_f32 v[{x0}:{x0 + 1}], v[{x0}:{x0 + 1}], v[{V_BIAS + 4 * j}:{V_BIAS + 4 * j + 1}]")
                            a(f"v_pk_add_f32 v[{x0 + 2}:{x0 + 3}], v[{x0 + 2}:{x0 + 3}], v[{V_BIAS + 4 * j + 2}:{V_BIAS + 4 * j + 3}]")
                        X = [T + 8, T + 10, XB, XB + 2]
                        D = [T + 16, T + 20, XB + 4, XB + 6]
                        G = [T + 18, T + 22, XB + 8, XB + 10]
                        P = [T + 12, T + 14, XB + 12, XB + 14]
                        self.gelu_pairs(X, D, G, P)
                        for e in range(2):
                            a(f"v_cvt_pk_bf16_f32 v{T + 4 * qp + 2 * e}, v{X[2 * e]}, v{X[2 * e] + 1}")
                            a(f"v_cvt_pk_bf16_f32 v{T + 4 * qp + 2 * e + 1}, v{X[2 * e + 1]}, v{X[2 * e + 1] + 1}")
                            if self.two_out:
                                a(f"v_cvt_pk_bf16_f32 v{V_T2 + 4 * qp + 2 * e}, v{D[2 * e]}, v{D[2 * e] + 1}")
                                a(f"v_cvt_pk_bf16_f32 v{V_T2 + 4 * qp + 2 * e + 1}, v{D[2 * e + 1]}, v{D[2 * e + 1] + 1}")
                        continue
                    for e in range(2):
                        j = 4 * slab + 2 * qp + e
                        d = acc(i, j)
                        for r in range(4):
                            a(f"v_accvgpr_read_b32 v{T + 8 + r}, a{d + r}")
                        if self.epi != "mulaux":
                            a(f"v_pk_add_f32 v[{T + 8}:{T + 9}], v[{T + 8}:{T + 9}], v[{V_BIAS + 4 * j}:{V_BIAS + 4 * j + 1}]")
                            a(f"v_pk_add_f32 v[{T + 10}:{T + 11}], v[{T + 10}:{T + 11}], v[{V_BIAS + 4 * j + 2}:{V_BIAS + 4 * j + 3}]")
                        if self.drop:
                            # dropout on (acc + bias), mask from the 2-multiply avalanche hash of (seed, element pair) -- common.h dropout4 / pair_hash,
                            # bit for bit (ln_bwd regenerates the same mask): pair = (m N + n) / 2; one 32-bit hash decides two elements (16-bit thresholds)
                            P, H0, H1, TT = T + 16, T + 17, T + 18, T + 19
                            a(f"s_mul_i32 s{S_T}, s{S_N8}, {i}")
                            a(f"s_add_u32 s{S_T}, s{S_T}, {32 * slab + 16 * qp + 2 * e}")
                            a(f"s_add_u32 s{S_T}, s{S_T}, s{S_CURPAIR}")
                            a(f"v_add_u32 v{P}, s{S_T}, v{V_PAIRB}")
                            a(f"v_xor_b32 v{H0}, {OP['dkey']}, v{P}")
                            a(f"v_add_u32 v{H1}, 1, v{P}")
                            a(f"v_xor_b32 v{H1}, {OP['dkey']}, v{H1}")
                            for sh, mul in ((16, S_HC1), (15, S_HC2), (16, None)):
                                for H in (H0, H1):
                                    a(f"v_lshrrev_b32 v{TT}, {sh}, v{H}")
                                    a(f"v_xor_b32 v{H}, v{H}, v{TT}")
                                    if mul is not None:
                                        a(f"v_mul_lo_u32 v{H}, v{H}, s{mul}")
                            for r in range(4):
                                a(f"v_mul_f32 v{T + 8 + r}, {OP['dinv']}, v{T + 8 + r}")
                            for r in range(4):
                                H = H0 if r < 2 else H1
                                if r & 1:
                                    a(f"v_lshrrev_b32 v{TT}, 16, v{H}")
                                else:
                                    a(f"v_and_b32 v{TT}, 0xffff, v{H}")
                                a(f"v_cmp_le_u32 vcc, {OP['dthr']}, v{TT}")
                                a(f"v_cndmask_b32 v{T + 8 + r}, 0, v{T + 8 + r}, vcc")
                        if self.side:
                            src = (R_BLOCK[i] + 8 * slab + (4 if qp else 0)) + 2 * e          # two dwords: columns 4e..4e+3 of the chunk
                            a(f"v_lshlrev_b32 v{T + 20}, 16, v{src}")
                            a(f"v_and_b32 v{T + 21}, 0xffff0000, v{src}")
                            a(f"v_lshlrev_b32 v{T + 22}, 16, v{src + 1}")
                            a(f"v_and_b32 v{T + 23}, 0xffff0000, v{src + 1}")
                            op = "v_pk_mul_f32" if self.epi == "mulaux" else "v_pk_add_f32"
                            a(f"{op} v[{T + 8}:{T + 9}], v[{T + 8}:{T + 9}], v[{T + 20}:{T + 21}]")
                            a(f"{op} v[{T + 10}:{T + 11}], v[{T + 10}:{T + 11}], v[{T + 22}:{T + 23}]")
                        a(f"v_cvt_pk_bf16_f32 v{T + 4 * qp + 2 * e}, v{T + 8}, v{T + 9}")
                        a(f"v_cvt_pk_bf16_f32 v{T + 4 * qp + 2 * e + 1}, v{T + 10}, v{T + 11}")
                # D0 = P0 with lanes t >= 8 taking P1 of lane t - 8; D1 = P1 with lanes t < 8 taking P0 of lane t + 8
                for r in range(4):
                    a(f"v_mov_b32 v{T + 12 + r}, v{T + r}")
                a("s_nop 1")
                for r in range(4):
                    a(f"v_mov_b32_dpp v{T + r}, v{T + 4 + r} row_ror:8 row_mask:0xf bank_mask:0xc")
                for r in range(4):
                    a(f"v_mov_b32_dpp v{T + 4 + r}, v{T + 12 + r} row_ror:8 row_mask:0xf bank_mask:0x3")
                if self.defer and i >= ni - 4:
                    H = V_HOLD + 8 * (2 * (i - (ni - 4)) + slab)
                    for r in range(8):
                        a(f"v_mov_b32 v{H + r}, v{T + r}")
                    continue
                a(f"s_add_u32 s{S_T}, s{S_SOFF}, {128 * slab}")
                nt = " nt" if self.gelu else ""            # (the FFN activations: 107 MB per output that nothing reads before the next GEMM -- streaming stores, as in the 8-wave kernel)
                a.vmem("store", f"buffer_store_dwordx4 v[{T}:{T + 3}], v{V_CST}, s[{S_RSC}:{S_RSC + 3}], s{S_T} offen{nt}")
                a(f"s_add_u32 s{S_T}, s{S_T}, s{S_LDC8}")
                a.vmem("store", f"buffer_store_dwordx4 v[{T + 4}:{T + 7}], v{V_CST}, s[{S_RSC}:{S_RSC + 3}], s{S_T} offen{nt}")
                a("s_nop 1")
                n_store += 2
                if self.two_out:                              # the same exchange and two full-line stores for gelu'(u) -> aux (descriptor / strides of the side input)
                    U = V_T2
                    for r in range(4):
                        a(f"v_mov_b32 v{U + 8 + r}, v{U + r}")
                    a("s_nop 1")
                    for r in range(4):
                        a(f"v_mov_b32_dpp v{U + r}, v{U + 4 + r} row_ror:8 row_mask:0xf bank_mask:0xc")
                    for r in range(4):
                        a(f"v_mov_b32_dpp v{U + 4 + r}, v{U + 8 + r} row_ror:8 row_mask:0xf bank_mask:0x3")
                    a(f"s_mul_i32 s{S_T}, s{S_LDR16}, {i}")
                    a(f"s_add_u32 s{S_T}, s{S_T}, {128 * slab}")
                    a.vmem("store", f"buffer_store_dwordx4 v[{U}:{U + 3}], v{V_RST}, s[{S_RSR}:{S_RSR + 3}], s{S_T} offen nt")
                    a(f"s_add_u32 s{S_T}, s{S_T}, s{S_LDR8}")
                    a.vmem("store", f"buffer_store_dwordx4 v[{U + 4}:{U + 7}], v{V_RST}, s[{S_RSR}:{S_RSR + 3}], s{S_T} offen nt")
                    a("s_nop 1")
                    n_store += 2
            a(f"s_add_u32 s{S_SOFF}, s{S_SOFF}, s{S_LDC16}")
        assert a.younger_vm("dma") == n_epi_vm, (a.younger_vm("dma"), n_epi_vm)
        if self.defer:
            for k in range(4):
                a(f"s_mov_b32 s{S_PREVC + k}, s{S_RSC + k}")
        # ---- next tile
        a(f"s_sub_u32 s{S_TILE}, s{S_TILE}, 1")
        a(f"s_cmp_eq_u32 s{S_TILE}, 0")
        a(f"s_cbranch_scc1 {l_done}")
        take_next_offsets()
        cur_output_descriptors()
        load_next()
        a(f"s_branch {l_tile}")
        a(f"{l_done}:")
        if self.defer:
            for u in range(8):
                self.flush_unit(u)
        a("s_waitcnt vmcnt(0) lgkmcnt(0)")
        return a.l


def lint(lines, name):
    """Static checks on a generated body: every register inside the ranges the clobber list declares (v243..v255 and s0..s14 belong to the compiler:
    the statement's operands live there), 64-bit VGPR operands of packed instructions even-aligned, counted waits inside their counters' widths."""
    import re
    for ln in lines:
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", ln):
            lo = int(m.group(1) if m.group(1) is not None else m.group(3))
            hi = int(m.group(2) if m.group(2) is not None else m.group(3))
            assert 0 <= lo <= hi <= V_LAST, (name, ln)
            if ln.startswith("v_pk_") and m.group(1) is not None:
                assert lo % 2 == 0 and hi == lo + 1, (name, ln)
        for m in re.finditer(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b", ln):
            hi = int(m.group(2) if m.group(2) is not None else m.group(3))
            assert hi <= 255, (name, ln)
        for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", ln):
            lo = int(m.group(1) if m.group(1) is not None else m.group(3))
            hi = int(m.group(2) if m.group(2) is not None else m.group(3))
            assert all((r in S_EXTRA) or (S0 <= r <= S_LAST) for r in range(lo, hi + 1)), (name, ln)
        if ln.startswith("buffer_"):                             # a buffer resource is four SGPRs aligned to four (the assembler refuses anything else)
            quads = [(int(a_), int(b_)) for a_, b_ in re.findall(r"\bs\[(\d+):(\d+)\]", ln)]
            assert quads and all(b_ - a_ == 3 and a_ % 4 == 0 for a_, b_ in quads), (name, ln)
        m = re.search(r"vmcnt\((\d+)\)", ln)
        assert m is None or int(m.group(1)) <= 63, (name, ln)
        m = re.search(r"lgkmcnt\((\d+)\)", ln)
        assert m is None or int(m.group(1)) <= 15, (name, ln)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
    opts = tuple(sys.argv[2:])
    with open(out, "w") as f:
        f.write("// GENERATED by scripts/gen_w4a.py -- do not edit; the schedule is described there\n")
        f.write(f"#define W4A_N_OPERANDS {len(OPS)}\n")
        f.write("// operand order: " + " ".join(OPS) + "\n")
        f.write("#define W4A_CLOBBERS " + ", ".join([f'"v{i}"' for i in range(V_LAST + 1)] + [f'"a{i}"' for i in range(256)] +
                                                    [f'"s{i}"' for i in S_EXTRA + list(range(S0, S_LAST + 1))] + ['"vcc"', '"scc"', '"m0"', '"memory"']) + "\n")
        for ni, bkm in ((8, False), (8, True), (7, False), (7, True)):
            for epi in ("plain", "resid", "mulaux") + (() if bkm else ("dropres", "gelu", "gelud")):
                lines = Gen(bkm, epi, ni, opts).body()
                lint(lines, (ni, bkm, epi))
                name = f"W4A_BODY{'' if ni == 8 else ni}_{'KM' if bkm else 'KC'}_{epi.upper()}"
                f.write(f"#define {name} \\\n")
                f.write(" \\\n".join('    "%s\\n\\t"' % x for x in lines))
                f.write("\n")
                print(f"{name}: {len(lines)} asm lines", file=sys.stderr)


if __name__ == "__main__":
    main()
