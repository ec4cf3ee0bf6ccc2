#!/usr/bin/env python3
"""Generator of the hand-scheduled four-wave GEMM main loop (csrc/gemm_w4a_asm.inc), round 4.

Why a generator: the loop is ONE inline-asm statement per kernel -- persistent tile loop, software-pipelined K loop, epilogue -- with
explicit register numbers (accumulators in a[0:255], operand fragments in v[0:95]; nothing is left to the compiler's allocator or
scheduler: round-3 review item 2).  A few hundred MFMAs with their fillers and counted waits are not something to type by hand; this script
places them and COUNTS the waits (every `s_waitcnt lgkmcnt(n)` is derived from the issue order it has just generated).

Geometry (fixed): workgroup 256 threads = 4 waves as 2 x 2, tile 256 x 256 x 64, wave tile 128 x 128 = 8 x 8 fragments of 16 x 16,
`v_mfma_f32_16x16x32_bf16` with the operands swapped (D = Bfrag x Afrag: a lane holds 4 consecutive output columns), LDS image and
output mapping those of the 8-wave kernel in gemm.hip (same swizzle keys, same B-row permutation, same full-line stores), so the two
kernels produce identical results.

LDS (bytes): A stage 0 [0, 32K), A stage 1 [32K, 64K), B stage 0 [64K, 96K), B stage 1 [96K, 128K), tile table [128K, +16K).

Pipeline per K-step k (stage s = k & 1), two phases of 64 MFMAs (kk = 0, 1), group i = the 8 MFMAs of A fragment i:
  phase 0: MFMAs on B(kk0) x A(kk0)[i]; the B(kk1) set is read into the second B buffer during groups 0-3; A(kk1)[i] is read INTO A(kk0)[i]'s
           registers half a group after group i (A fragments need one buffer, B fragments two: 96 fragment registers).
  phase 1: MFMAs on B(kk1) x A(kk1)[i].  At group 2 every read of stage s has returned (lgkmcnt 0) -> barrier M -> stage s is free: the
           LDS-DMA of K-step k+2 goes into it, 8 pieces before and 8 after barrier E.  At group 5: `s_waitcnt vmcnt(N)`, N = the VMEM
           operations younger than K-step k+1's last DMA piece (the counter is in order) -> barrier E -> stage s^1 (K-step k+1) is visible: its
           B(kk0) set and A(kk0)[0..6] are read during groups 5-7 into registers phase 1 has finished with ([7]: first group of the next step).
  A K-step's DMA is issued 0.6-1.0 K-steps before the step before it ends and is needed 0.7 K-steps into that step: 1.1-1.3 K-steps of
  latency budget with 128 KB of LDS (the 8-wave kernel: 1.0, and its K-step then waits for the slowest piece).
Tiles follow each other without a prologue: during a tile's last two K-steps the DMA slots carry the NEXT tile's first two K-steps
(descriptor swap), so the next tile's fragments are in registers when the epilogue starts, and the epilogue's stores are YOUNGER than
every load the next K loop waits for (gfx950 has one in-order vmcnt for loads and stores: a wait never has to sit out a store's ~2 us
acknowledge unless it asks for something issued after it).
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
V_AB = [170, 171]        # B fragment address kk0 / kk1 (B region base folded in)
V_CST = 172              # store lane offset (bytes)
V_BOFF = 173             # bias lane offset (bytes)
V_TBL = 174              # LDS address of the tile table (same in every lane)
V_LAST = 179
S0 = 36
S_RSA, S_RSB = 36, 40            # DMA descriptors in use (current tile, or the next tile's during the last pair)
S_NXA, S_NXB = 44, 48            # next tile's
S_RSC, S_RSBIAS = 52, 56         # output / bias descriptors of the tile being finished
S_NXC_OFF, S_NXN0 = 60, 61       # next tile's output offset / first column
S_K = 62                         # k byte offset of the next DMA (soffset)
S_PAIRS = 63                     # remaining K-step pairs of the tile
S_TILE = 64                      # tiles left
S_TMP = 65                       # 65..71 scratch
S_CUR_C_OFF, S_CUR_N0 = 72, 73
S_M0A, S_M0B = 74, 75            # LDS destinations of this wave's first A / B piece in stage 0
S_LDC16, S_LDC8 = 76, 77         # 16 / 8 output rows in bytes
S_SOFF = 78                      # store soffset cursor
S_TIDX = 79                      # index of the next table entry to read
S_NULL = 80                      # 80..83: null descriptor (num_records 0)
S_BIASB = 84                     # 84,85: bias base or 0; 86: bias bytes
S_LAST = 91

# operands of the asm statement (gemm_w4a.h must pass them in this order)
OPS = ["tbl", "voA0", "voBbase", "chunkx", "aA0", "aB0", "cst", "boff",                      # "v"
       "A_lo", "A_hi", "B_lo", "B_hi", "C_lo", "C_hi", "bias_lo", "bias_hi",                 # "s"
       "a_bytes", "b_bytes", "c_bytes", "lda64", "ldb16", "ldc2", "pairs", "ntiles", "m0A", "m0B", "bias_bytes"]
OP = {n: f"%{i}" for i, n in enumerate(OPS)}


class Asm:
    def __init__(self):
        self.l = []
        self.lds = []          # tags of the LDS reads issued so far, in order
        self.uid = 0

    def __call__(self, s):
        self.l.append(s)

    def read(self, tag, text):
        self.lds.append(tag)
        self(text)

    def wait_lds(self, *tags):
        """wait until the reads tagged `tags` have returned: LDS reads return in order, so everything issued after the youngest of them
        may stay outstanding"""
        idx = max(i for i, t in enumerate(self.lds) if t in tags)
        assert all(t in self.lds for t in tags), tags
        n = len(self.lds) - 1 - idx
        assert n <= 15, (tags, n)
        self(f"s_waitcnt lgkmcnt({n})")

    def wait_lds_all(self):
        self("s_waitcnt lgkmcnt(0)")

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


def b_frag_off(j):
    return (32 * (j >> 1) + 4 * (j & 1)) * 128


def read_b(a, j, kk, stage, gen):
    r = V_B[kk] + 4 * j
    a.read(("B", gen, kk, j), f"ds_read_b128 v[{r}:{r + 3}], v{V_AB[kk]} offset:{(B_ST[stage] - B_ST[0]) + b_frag_off(j)}")


def read_a(a, i, kk, stage, gen):
    r = V_A + 4 * i
    a.read(("A", gen, kk, i), f"ds_read_b128 v[{r}:{r + 3}], v{V_AA[kk]} offset:{A_ST[stage] + i * 2048}")


def dma_piece(a, p, stage):
    """piece p of the wave's 16 per K-step (0-7: A rows, 8-15: B rows) into `stage`; source = descriptors in use, soffset S_K"""
    if p < 8:
        a(f"s_add_u32 m0, s{S_M0A}, {A_ST[stage] + p * NW * 1024}")
        a("s_nop 0")
        a(f"buffer_load_dwordx4 v{V_VOA + p}, s[{S_RSA}:{S_RSA + 3}], s{S_K} offen lds")
    else:
        q = p - 8
        a(f"s_add_u32 m0, s{S_M0B}, {(B_ST[stage] - B_ST[0]) + q * 1024}")
        a("s_nop 0")
        a(f"buffer_load_dwordx4 v{V_VOB + q}, s[{S_RSB}:{S_RSB + 3}], s{S_K} offen lds")


GEN = [0]


def gen_step(a, stage, first, n_e, bias_loads=False):
    """One K-step on `stage`.  first: the accumulators start from 0 (phase 0 takes the inline constant as C).  n_e: vmcnt count of barrier E.
    bias_loads: the 8 bias-quad loads of this tile ride in phase 0 (older than everything the tile's later waits ask for)."""
    g0 = GEN[0]            # generation number of this step's fragments (tags)
    GEN[0] += 1
    # ---------------- phase 0: fillers by (group, slot)
    fill = {}
    for j in range(8):
        fill[(j // 2, 2 + j % 2)] = lambda j=j: read_b(a, j, 1, stage, g0)   # (not in the first slots: the previous phase's last MFMAs read this buffer)
    fill[(0, 5)] = lambda: read_a(a, 7, 0, stage, g0)                      # the last A(kk0) fragment (its registers were busy until now)
    for i in range(7):
        fill[(i + 1, 5)] = lambda i=i: read_a(a, i, 1, stage, g0)          # A(kk1)[i] over A(kk0)[i], half a group after its MFMAs
    if bias_loads:
        for j in range(8):
            off = (32 * (j >> 1) + 4 * (j & 1)) * 4
            fill[(4 + j // 2, 1 + j % 2)] = lambda j=j, off=off: a(
                f"buffer_load_dwordx4 v[{V_BIAS + 4 * j}:{V_BIAS + 4 * j + 3}], v{V_BOFF}, s[{S_RSBIAS}:{S_RSBIAS + 3}], 0 offen offset:{off}")
    for i in range(8):
        if i == 0:
            a.wait_lds(("B", g0, 0, 7), ("A", g0, 0, 0))
        else:
            a.wait_lds(("A", g0, 0, i))
        for j in range(8):
            mfma(a, i, j, 0, first)
            if (i, j) in fill:
                fill[(i, j)]()
    # ---------------- phase 1
    for i in range(8):
        slots = {}
        if i == 0:
            a.wait_lds(("B", g0, 1, 7), ("A", g0, 1, 0))
            slots[5] = [lambda: read_a(a, 7, 1, stage, g0)]
        elif i == 1:
            a.wait_lds(("A", g0, 1, 1))
        elif i == 2:
            a.wait_lds_all()                              # every read of this stage is back ...
            a("s_barrier")                                # ... barrier M: in every wave -> the stage is free
        if 2 <= i <= 4:                                   # DMA of K-step k+2, pieces 0-7 before barrier E
            lo, hi = {2: (0, 3), 3: (3, 6), 4: (6, 8)}[i]
            for n, p in enumerate(range(lo, hi)):
                slots[1 + 2 * n] = [lambda p=p: dma_piece(a, p, stage)]
        if i == 5:
            a(f"s_waitcnt vmcnt({n_e})")                  # K-step k+1 has landed (this wave's pieces) ...
            a("s_barrier")                                # ... barrier E: everybody's
            for j in range(8):
                slots[j] = [lambda j=j: read_b(a, j, 0, stage ^ 1, g0 + 1)]
        if i == 6:
            for q in range(5):
                slots[q] = [lambda q=q: read_a(a, q, 0, stage ^ 1, g0 + 1)]
            for n, p in enumerate(range(8, 11)):
                slots[5 + n] = [lambda p=p: dma_piece(a, p, stage)]
        if i == 7:
            slots[0] = [lambda: read_a(a, 5, 0, stage ^ 1, g0 + 1)]
            for n, p in enumerate(range(11, 16)):
                slots[1 + n] = [lambda p=p: dma_piece(a, p, stage)]
        for j in range(8):
            mfma(a, i, j, 1, False)
            for f in slots.get(j, []):
                f()
    read_a(a, 6, 0, stage ^ 1, g0 + 1)
    a(f"s_add_u32 s{S_K}, s{S_K}, 128")


def main():
    a = Asm()
    a("s_nop 4")
    a(f"v_mov_b32 v{V_TBL}, {OP['tbl']}")
    a(f"v_mov_b32 v{V_VOA}, {OP['voA0']}")
    for p in range(1, 8):                                  # A pieces are 32 rows apart (piece q = 4 p + wave), same swizzle key
        a(f"v_add_u32 v{V_VOA + p}, v{V_VOA + p - 1}, {OP['lda64']}")
    for j in range(8):                                     # B piece j: rows 8 j further, key bits 1-2 = j & 3
        a(f"v_xor_b32 v{V_T}, {2 * (j & 3)}, {OP['chunkx']}")
        a(f"v_lshlrev_b32 v{V_T}, 4, v{V_T}")
        a(f"s_mul_i32 s{S_TMP}, {OP['ldb16']}, {j}")
        a(f"v_add3_u32 v{V_VOB + j}, {OP['voBbase']}, v{V_T}, s{S_TMP}")
    a(f"v_mov_b32 v{V_AA[0]}, {OP['aA0']}")
    a(f"v_xor_b32 v{V_AA[1]}, 64, {OP['aA0']}")
    a(f"v_mov_b32 v{V_AB[0]}, {OP['aB0']}")
    a(f"v_xor_b32 v{V_AB[1]}, 64, {OP['aB0']}")
    a(f"v_mov_b32 v{V_CST}, {OP['cst']}")
    a(f"v_mov_b32 v{V_BOFF}, {OP['boff']}")
    a(f"s_mov_b32 s{S_M0A}, {OP['m0A']}")
    a(f"s_mov_b32 s{S_M0B}, {OP['m0B']}")
    a(f"s_lshl_b32 s{S_LDC8}, {OP['ldc2']}, 3")
    a(f"s_lshl_b32 s{S_LDC16}, {OP['ldc2']}, 4")
    a(f"s_mov_b32 s{S_TILE}, {OP['ntiles']}")
    a(f"s_mov_b32 s{S_NULL}, 0")
    a(f"s_mov_b32 s{S_NULL + 1}, 0")
    a(f"s_mov_b32 s{S_NULL + 2}, 0")
    a(f"s_mov_b32 s{S_NULL + 3}, 0x00020000")
    a(f"s_mov_b32 s{S_TIDX}, 0")
    a(f"s_mov_b32 s{S_BIASB}, {OP['bias_lo']}")
    a(f"s_mov_b32 s{S_BIASB + 1}, {OP['bias_hi']}")
    a(f"s_or_b32 s{S_TMP}, {OP['bias_lo']}, {OP['bias_hi']}")
    a(f"s_cmp_eq_u32 s{S_TMP}, 0")
    a(f"s_cselect_b32 s{S_BIASB + 2}, 0, {OP['bias_bytes']}")           # no bias: zero records -> every bias load reads 0

    def load_next():
        """table entry S_TIDX -> next-tile descriptors (null descriptors past the end); entry = {a_off, b_off, c_off, n0}"""
        l_no, l_done = a.label("nonext"), a.label("nextdone")
        a(f"s_cmp_lt_u32 s{S_TIDX}, {OP['ntiles']}")
        a(f"s_cbranch_scc0 {l_no}")
        a(f"s_lshl_b32 s{S_TMP}, s{S_TIDX}, 4")
        a(f"v_add_u32 v{V_T}, s{S_TMP}, v{V_TBL}")
        a(f"ds_read_b128 v[{V_T + 4}:{V_T + 7}], v{V_T}")
        a("s_waitcnt lgkmcnt(0)")
        for k in range(4):
            a(f"v_readfirstlane_b32 s{S_TMP + 1 + k}, v{V_T + 4 + k}")     # a_off, b_off, c_off, n0
        a("s_nop 3")
        a(f"s_add_u32 s{S_NXA}, {OP['A_lo']}, s{S_TMP + 1}")
        a(f"s_addc_u32 s{S_NXA + 1}, {OP['A_hi']}, 0")
        a(f"s_sub_u32 s{S_NXA + 2}, {OP['a_bytes']}, s{S_TMP + 1}")
        a(f"s_mov_b32 s{S_NXA + 3}, 0x00020000")
        a(f"s_add_u32 s{S_NXB}, {OP['B_lo']}, s{S_TMP + 2}")
        a(f"s_addc_u32 s{S_NXB + 1}, {OP['B_hi']}, 0")
        a(f"s_sub_u32 s{S_NXB + 2}, {OP['b_bytes']}, s{S_TMP + 2}")
        a(f"s_mov_b32 s{S_NXB + 3}, 0x00020000")
        a(f"s_mov_b32 s{S_NXC_OFF}, s{S_TMP + 3}")
        a(f"s_mov_b32 s{S_NXN0}, s{S_TMP + 4}")
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

    def cur_output_descriptors():
        """C / bias descriptors of the tile whose K loop starts now"""
        a(f"s_add_u32 s{S_RSC}, {OP['C_lo']}, s{S_CUR_C_OFF}")
        a(f"s_addc_u32 s{S_RSC + 1}, {OP['C_hi']}, 0")
        a(f"s_sub_u32 s{S_RSC + 2}, {OP['c_bytes']}, s{S_CUR_C_OFF}")
        a(f"s_mov_b32 s{S_RSC + 3}, 0x00020000")
        a(f"s_lshl_b32 s{S_TMP}, s{S_CUR_N0}, 2")
        a(f"s_add_u32 s{S_RSBIAS}, s{S_BIASB}, s{S_TMP}")
        a(f"s_addc_u32 s{S_RSBIAS + 1}, s{S_BIASB + 1}, 0")
        a(f"s_sub_u32 s{S_RSBIAS + 2}, s{S_BIASB + 2}, s{S_TMP}")
        a(f"s_max_i32 s{S_RSBIAS + 2}, s{S_RSBIAS + 2}, 0")
        a(f"s_mov_b32 s{S_RSBIAS + 3}, 0x00020000")

    # ---- kernel prologue: tile 0's descriptors, its first two K-steps, 32 null stores (the first K-step's vmcnt count assumes an epilogue
    # before it), its first fragments
    load_next()
    next_to_cur()
    a(f"s_mov_b32 s{S_CUR_C_OFF}, s{S_NXC_OFF}")
    a(f"s_mov_b32 s{S_CUR_N0}, s{S_NXN0}")
    load_next()
    cur_output_descriptors()
    a(f"s_mov_b32 s{S_K}, 0")
    a("s_nop 4")
    for p in range(16):
        dma_piece(a, p, 0)
    a(f"s_add_u32 s{S_K}, s{S_K}, 128")
    for p in range(16):
        dma_piece(a, p, 1)
    a(f"s_add_u32 s{S_K}, s{S_K}, 128")
    a("s_waitcnt vmcnt(16)")
    a("s_barrier")
    GEN[0] = 0
    for j in range(8):
        read_b(a, j, 0, 0, 0)
    for i in range(7):
        read_a(a, i, 0, 0, 0)
    for _ in range(32):
        a(f"buffer_store_dword v{V_T}, v{V_CST}, s[{S_NULL}:{S_NULL + 3}], 0 offen")
    a("s_nop 1")

    # ---- tile loop
    l_tile, l_pair, l_noswap, l_done = a.label("tile"), a.label("pair"), a.label("noswap"), a.label("done")
    a(f"{l_tile}:")
    a(f"s_mov_b32 s{S_PAIRS}, {OP['pairs']}")
    # first pair.  vmcnt of the first barrier E: younger than this tile's second K-step (issued during the previous tile's last step) are
    # the previous epilogue's 32 stores, the 8 bias loads of phase 0 and the 8 DMA pieces issued before the barrier
    gen_step(a, 0, True, 32 + 8 + 8, bias_loads=True)
    gen_step(a, 1, False, 8)
    a(f"s_sub_u32 s{S_PAIRS}, s{S_PAIRS}, 1")
    lds_state = list(a.lds)
    a(f"{l_pair}:")
    a(f"s_cmp_eq_u32 s{S_PAIRS}, 1")
    a(f"s_cbranch_scc0 {l_noswap}")
    next_to_cur()                                           # last pair: its DMA slots carry the next tile's first two K-steps
    a(f"s_mov_b32 s{S_K}, 0")
    a(f"{l_noswap}:")
    gen_step(a, 0, False, 8)
    gen_step(a, 1, False, 8)
    a(f"s_sub_u32 s{S_PAIRS}, s{S_PAIRS}, 1")
    a(f"s_cmp_eq_u32 s{S_PAIRS}, 0")
    a(f"s_cbranch_scc0 {l_pair}")
    # (the read-tracking state at the loop's back edge and at its entry must describe the same tail: both are "end of a step on stage 1")
    assert [t[0] + str(t[2:]) for t in lds_state[-24:]] == [t[0] + str(t[2:]) for t in a.lds[-24:]]

    # ---- epilogue of the finished tile (descriptor S_RSC still its own); the next tile's first fragments are already in v[0:95]
    a("s_nop 15")
    a(f"s_mov_b32 s{S_SOFF}, 0")
    T = V_T
    for i in range(8):
        for slab in range(2):
            # P0 / P1: the lane's 8 consecutive columns of column groups q' = 0 / 1 of the slab (fragments 4 slab + 2 q' + e, e = 0, 1)
            for qp in range(2):
                for e in range(2):
                    j = 4 * slab + 2 * qp + e
                    d = acc(i, j)
                    for r in range(4):
                        a(f"v_accvgpr_read_b32 v{T + 8 + r}, a{d + r}")
                    a(f"v_pk_add_f32 v[{T + 8}:{T + 9}], v[{T + 8}:{T + 9}], v[{V_BIAS + 4 * j}:{V_BIAS + 4 * j + 1}]")
                    a(f"v_pk_add_f32 v[{T + 10}:{T + 11}], v[{T + 10}:{T + 11}], v[{V_BIAS + 4 * j + 2}:{V_BIAS + 4 * j + 3}]")
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
            a(f"s_add_u32 s{S_TMP}, s{S_SOFF}, {128 * slab}")
            a(f"buffer_store_dwordx4 v[{T}:{T + 3}], v{V_CST}, s[{S_RSC}:{S_RSC + 3}], s{S_TMP} offen")
            a(f"s_add_u32 s{S_TMP}, s{S_TMP}, s{S_LDC8}")
            a(f"buffer_store_dwordx4 v[{T + 4}:{T + 7}], v{V_CST}, s[{S_RSC}:{S_RSC + 3}], s{S_TMP} offen")
            a("s_nop 1")
        a(f"s_add_u32 s{S_SOFF}, s{S_SOFF}, s{S_LDC16}")
    # ---- next tile
    a(f"s_sub_u32 s{S_TILE}, s{S_TILE}, 1")
    a(f"s_cmp_eq_u32 s{S_TILE}, 0")
    a(f"s_cbranch_scc1 {l_done}")
    a(f"s_mov_b32 s{S_CUR_C_OFF}, s{S_NXC_OFF}")
    a(f"s_mov_b32 s{S_CUR_N0}, s{S_NXN0}")
    cur_output_descriptors()
    load_next()
    a(f"s_branch {l_tile}")
    a(f"{l_done}:")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    out = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
    with open(out, "w") as f:
        f.write("// GENERATED by scripts/gen_w4a.py -- do not edit; the schedule is described there\n")
        f.write(f"#define W4A_N_OPERANDS {len(OPS)}\n")
        f.write("// operand order: " + " ".join(OPS) + "\n")
        f.write("#define W4A_CLOBBERS " + ", ".join([f'"v{i}"' for i in range(V_LAST + 1)] + [f'"a{i}"' for i in range(256)] +
                                                    [f'"s{i}"' for i in range(S0, S_LAST + 1)] + ['"scc"', '"m0"', '"memory"']) + "\n")
        f.write("#define W4A_ASM_BODY \\\n")
        f.write(" \\\n".join('    "%s\\n\\t"' % x for x in a.l))
        f.write("\n")
    print(f"{len(a.l)} asm lines", file=sys.stderr)


if __name__ == "__main__":
    main()
