// gemm_pp.h -- the "ping-pong" K loop of the bf16 256-column kernel.  Included by gemm.hip inside its anonymous namespace (it uses
// that file's tile bookkeeping, LDS swizzle keys and epilogues).
//
// Why a second K loop.  gemm_bf16_body keeps its 8 waves in lock step: at the head of every 64-deep K-step each wave issues the ~8 LDS-DMA
// pieces of the next stage back to back, the CU's vector-memory pipe (36 B/clk at best = 28 cycles per 1 KiB piece) backs up, and both
// waves of a SIMD sit in front of their first MFMA for ~1 000 cycles (s_memtime trace, profiles/r02_gemm_phase_trace.txt: 2 884 cycles per
// K-step against 1 792 of MFMA).  Here the K-step is cut into four phases and the two wave rows (waves 0-3 and 4-7 = one wave per SIMD
// each) run half a phase apart, every segment fenced by s_barrier:
//
//      waves 0-3:   L0 | M0 | L1 | M1 | L2 | M2 | L3 | M3 |
//      waves 4-7:        L0 | M0 | L1 | M1 | L2 | M2 | L3 | M3
//
// M = 8-16 MFMAs with every operand already in registers (the SIMD's matrix pipe runs them back to back), L = the fragment reads of the
// next M segment + two LDS-DMA pieces.  While one wave of a SIMD computes, its partner loads, and the DMA issue is spread evenly over the
// K-step (2 pieces per segment) instead of arriving as one burst.
//
// The DMA stream runs 1-2 K-steps ahead of the MFMAs at sub-stage granularity.  An operand stage has three regions: B (read completely
// into registers in L0, both 32-deep halves), A-front (the fragments of phases 0/1) and A-back (phases 2/3).  A region is refilled as soon
// as every wave has read it, not when the whole stage is done:
//      L0(k): A-back (k+1)          -- the back region of the other stage died with L3(k-1)
//      L1(k): B      (k+2), half    -- this stage's B died with L0(k)
//      L2(k): A-front(k+2)          -- this stage's front died with L1(k)
//      L3(k): B      (k+2), rest
// so every piece has about one full K-step between its issue and the (counted) s_waitcnt vmcnt(N) in front of its first reader, and
// the stream does not stop at a tile boundary: the cursor that generates the source addresses simply walks on into the workgroup's next
// (tile, K-slice) unit, so the next tile's first two K-steps are in LDS (or in flight) before the current tile's epilogue starts.  The
// epilogue waits for them once, BEFORE its first store (gfx950 counts loads and stores in one vmcnt; a wait behind the stores would also
// wait for their write acknowledgement), and the first K-step of the next tile runs without any vmcnt wait while the stores drain.
//
// Hazards (intervals between barriers numbered globally, waves 4-7 one behind): reads of a segment are retired (lgkmcnt(0)) before the
// barrier that ends the segment, so a region's last reader is done one barrier before the earliest refill above; a refill is waited for
// (vmcnt) in the L segment one phase before its first reader, by every wave that issued a piece of it, and at least one barrier lies
// between that wait and the read.  The two wave rows are re-aligned around every epilogue (the first row idles one segment at the end of
// a tile, the second one at the start of the next) so that both write their output at the same time.

template <bool AKM, bool BKM, int EPI, int CNT, bool GROUP>
__device__ __forceinline__ void gemm_pp_body(DicGemmParams& p, const WgradGroupDev* grp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = T256;
    using G = Geo<C>;
    using T = bf16_t;
    constexpr int S = 2, BK = 64;
    static_assert(CNT >= 4 && CNT <= G::FM && (!AKM || CNT == G::FM), "fragments per wave");
    static_assert(G::FN == 4 && G::NW == 8, "two wave rows of four waves, four B fragments per wave");
    static_assert(!GROUP || (AKM && BKM && EPI == DIC_EPI_AFFINE), "grouped launches are weight gradients");
    constexpr int FH = CNT >= 5 ? 4 : CNT / 2, BH = CNT - FH;                 // A fragments per wave in the front / back region
    constexpr int NF = 4 * FH / 8;                                            // front pieces (1 KiB) per wave and K-step
    constexpr int NBS = (4 * BH + 7) / 8;                                     // back piece slots per wave
    constexpr bool BACK_EVEN = (4 * BH) % 8 == 0;                             // else waves 4..7 (the second wave row) own one piece fewer
    constexpr int NBK0 = NBS, NBK1 = BACK_EVEN ? NBS : NBS - 1;
    static_assert((4 * FH) % 8 == 0, "front pieces divide evenly over the waves");
    constexpr int ROWB_A = AKM ? 256 : 128, ROWB_B = BKM ? G::BN * 2 : 128;   // LDS row pitch (k-major A: two half tiles of 128 rows)
    constexpr int tile_rows = 32 * CNT;
    constexpr bool COLSUM = AKM && BKM && EPI == DIC_EPI_AFFINE;              // weight gradients: the bias gradient rides along (see gemm_bf16_body)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int g = lane >> 4, t = lane & 15;
    const int row0_w = wm * CNT * 16;

    // ---- per-lane fragment addresses inside a stage (see gemm_bf16_body; k-major A is stored as two [64 k][128 rows] half tiles: the front
    // half holds rows 0-63 and 128-191 of the tile = fragments 0-3 of both wave rows, the back half the rest)
    int ofA[AKM ? CNT : 1], ofB[BKM ? G::FN : 1];
    if constexpr (!AKM) { const int row = row0_w + t; ofA[0] = row * 128 + ((g ^ key_a(row)) << 4); }
    else {
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const int half = i >> 2, mp = wm * 64 + 16 * (i & 3), rho = 8 * g + (t >> 2), c = (mp >> 3) + ((t & 3) >> 1);
            ofA[i] = half * 16384 + rho * ROWB_A + ((c ^ km_key(rho)) << 4) + (t & 1) * 8;
        }
    }
    if constexpr (!BKM) { const int row = wn * G::WCOLS + 8 * (t >> 2) + (t & 3); ofB[0] = row * 128 + ((g ^ key_b(row)) << 4); }
    else {
#pragma unroll
        for (int j = 0; j < G::FN; ++j) {
            const int base = wn * G::WCOLS + 32 * (j >> 1), rho = 8 * g + (t >> 2), c = (base >> 3) + (t & 3);
            ofB[j] = rho * ROWB_B + ((c ^ km_key(rho)) << 4) + (j & 1) * 8;
        }
    }

    // ---- DMA cursor: the (unit, K-step) whose operand regions are issued next; runs ahead of the MFMAs across unit boundaries ------------
    const int total = GROUP ? grp->split * grp->tiles : total_units(p, tile_rows, G::BN);
    i32x4 rsA, rsB;
    unsigned voF[NF], voK[NBS], voB[4];
    unsigned stepA = AKM ? (unsigned)BK * (unsigned)p.lda * 2u : BK * 2u;
    unsigned stepB = BKM ? (unsigned)BK * (unsigned)p.ldb * 2u : BK * 2u;
    int c_unit = blockIdx.x, c_k = 0, c_k1 = 0, c_par = 0;
    bool c_valid = true;
    auto make_rsrc = [](const void* base, long long bytes) {
        const unsigned long long b = (unsigned long long)base;
        i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        r[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
        r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
        r[3] = 0x00020000;
        return r;
    };
    // 64 lanes x 16 B -> LDS [lds_addr, lds_addr + 1 KiB).  (s_nop 3: m0 write -> LDS-DMA needs one wait state, and a descriptor SGPR fresh
    // from v_readfirstlane five before a VMEM instruction reads it -- nothing pads inside an asm statement.)
    auto dma16 = [](unsigned voff, const i32x4& rsrc, unsigned lds_addr) {
#ifdef DIC_PP_NODMA             // timing ablations (scripts/experiments/pp_ablate.sh): results are garbage
        return;
#endif
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
    };
    // tile-local piece numbers of this wave (front slot j / back slot j) and their LDS offsets inside a stage.  k-contiguous A: a piece is
    // 8 rows; the front region is rows [0, 16 FH) of each wave row.  k-major A: a piece is 4 k-rows of a half tile.
    unsigned dF[NF], dK[NBS];
    int qF[NF], qK[NBS];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int f = j * 8 + wave;
        qF[j] = AKM ? f : (f < 2 * FH ? f : 2 * CNT + (f - 2 * FH));
        dF[j] = __builtin_amdgcn_readfirstlane((unsigned)qF[j] * 1024u);
    }
#pragma unroll
    for (int j = 0; j < NBS; ++j) {
        const int b = j * 8 + wave;
        qK[j] = AKM ? b : (b < 2 * BH ? 2 * FH + b : 2 * CNT + 2 * FH + (b - 2 * BH));
        dK[j] = __builtin_amdgcn_readfirstlane(AKM ? 16384u + (unsigned)b * 1024u : (unsigned)qK[j] * 1024u);
    }
    const unsigned dB = __builtin_amdgcn_readfirstlane((unsigned)(G::A_BYTES + wave * 4096));
    const void* cA = p.A;
    const void* cB = p.B;
    int c_lda = p.lda, c_ldb = p.ldb, c_M = p.M, c_N = p.N;
    auto cursor_setup = [&](const TileId& tl) {          // descriptors anchored at the tile origin: out-of-range rows / k read as zero
        const int m0 = tl.bm * tile_rows, n0 = tl.bn * G::BN;
        const T* Ab = (const T*)cA + (AKM ? (size_t)m0 : (size_t)m0 * c_lda);
        const T* Bb = (const T*)cB + (BKM ? (size_t)n0 : (size_t)n0 * c_ldb);
        long long a_bytes = AKM ? ((long long)(p.K - 1) * c_lda + (c_M - m0)) * S : ((long long)(c_M - m0 - 1) * c_lda + p.K) * S;
        long long b_bytes = BKM ? ((long long)(p.K - 1) * c_ldb + (c_N - n0)) * S : ((long long)(c_N - n0 - 1) * c_ldb + p.K) * S;
        if (a_bytes > 0xFFFFFFF0ll) a_bytes = 0xFFFFFFF0ll;
        if (b_bytes > 0xFFFFFFF0ll) b_bytes = 0xFFFFFFF0ll;
        rsA = make_rsrc(Ab, a_bytes);
        rsB = make_rsrc(Bb, b_bytes);
        const unsigned ka = (unsigned)tl.kt0 * stepA, kb = (unsigned)tl.kt0 * stepB;
        auto a_off = [&](int q, int half_base) -> unsigned {
            if constexpr (!AKM) {
                const int row = 8 * q + (lane >> 3);
                return ka + (unsigned)row * (unsigned)c_lda * 2u + (((lane & 7) ^ key_a(row)) << 4);
            } else {          // LDS chunk position (lane & 15) of k-row krow holds the source chunk cs = position ^ key
                const int krow = 4 * q + (lane >> 4), cs = (lane & 15) ^ km_key(krow), m = half_base + (cs < 8 ? 8 * cs : 128 + 8 * (cs - 8));
                return ka + (unsigned)krow * (unsigned)c_lda * 2u + (unsigned)m * 2u;
            }
        };
#pragma unroll
        for (int j = 0; j < NF; ++j) voF[j] = a_off(qF[j], 0);
#pragma unroll
        for (int j = 0; j < NBS; ++j) voK[j] = a_off(qK[j], 64);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = wave * 4 + j;
            if constexpr (!BKM) { const int row = 8 * q + (lane >> 3); voB[j] = kb + (unsigned)row * (unsigned)c_ldb * 2u + (((lane & 7) ^ key_b(row)) << 4); }
            else { constexpr int CPRW = G::BN / 8; const int row = q * (64 / CPRW) + lane / CPRW; voB[j] = kb + (unsigned)row * (unsigned)c_ldb * 2u + (((lane % CPRW) ^ km_key(row)) << 4); }
        }
    };
    int slab = 0;
    auto group_tile = [&](int u, int& slab_out, int& pi_out) {          // grouped launch: unit u -> (problem, tile, K range); see gemm_bf16_body
        const int nwg = grp->split * grp->tiles;
        const int q_ = nwg >> 3, r_ = nwg & 7, xcd = u & 7, slot = u >> 3;
        const int lu = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + slot;
        const int kz = lu / grp->tiles, tg = lu - kz * grp->tiles;
        int pi = 0;
        while (pi + 1 < grp->n && tg >= grp->tile0[pi + 1]) ++pi;
        const int lt = tg - grp->tile0[pi], nbn_ = grp->nbn[pi];
        TileId t_;
        t_.bm = lt / nbn_; t_.bn = lt - t_.bm * nbn_; t_.nbn = nbn_; t_.kz = kz;
        t_.kt0 = kz * grp->per < grp->nk ? kz * grp->per : grp->nk;
        t_.kt1 = t_.kt0 + grp->per < grp->nk ? t_.kt0 + grp->per : grp->nk;
        slab_out = lu; pi_out = pi;
        return t_;
    };
    auto cursor_tile = [&](int u) {
        if constexpr (GROUP) {
            int sl, pi;
            const TileId t_ = group_tile(u, sl, pi);
            cA = grp->A[pi]; cB = grp->B[pi]; c_lda = grp->lda[pi]; c_ldb = grp->ldb[pi]; c_M = grp->M[pi]; c_N = grp->N[pi];
            stepA = (unsigned)BK * (unsigned)c_lda * 2u; stepB = (unsigned)BK * (unsigned)c_ldb * 2u;
            return t_;
        } else {
            return tile_of_unit(p, BK, u, tile_rows, G::BN);
        }
    };
    auto compute_tile = [&](int u) {
        if constexpr (GROUP) {
            int pi;
            const TileId t_ = group_tile(u, slab, pi);
            p.M = grp->M[pi]; p.N = grp->N[pi]; p.colsum_out = grp->cs[pi];
            return t_;
        } else {
            return tile_of_unit(p, BK, u, tile_rows, G::BN);
        }
    };
    auto cursor_open = [&]() {                          // position the cursor on the first K-step of c_unit (skipping units without K-steps)
        for (;;) {
            if (c_unit >= total) { c_valid = false; return; }
            const TileId t_ = cursor_tile(c_unit);
            if (t_.kt0 < t_.kt1) { c_k = t_.kt0; c_k1 = t_.kt1; cursor_setup(t_); return; }
            c_unit += (int)gridDim.x;
        }
    };
    auto advance = [&]() {
        if (!c_valid) return;
        ++c_par;
        if (++c_k >= c_k1) { c_unit += (int)gridDim.x; cursor_open(); }
    };
    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_PTR(char))smem;
    auto issue_front = [&]() {
        if (!c_valid) return;
        const unsigned sb = lds_base + (unsigned)(c_par & 1) * G::STAGE;
#pragma unroll
        for (int j = 0; j < NF; ++j) { dma16(voF[j], rsA, sb + dF[j]); voF[j] += stepA; }
    };
    auto issue_back = [&]() {
        if (!c_valid) return;
        const unsigned sb = lds_base + (unsigned)(c_par & 1) * G::STAGE;
#pragma unroll
        for (int j = 0; j < NBS; ++j) {
            if (BACK_EVEN || j < NBS - 1 || wm == 0) dma16(voK[j], rsA, sb + dK[j]);
            voK[j] += stepA;
        }
    };
    auto issue_b = [&](auto j0_c) {
        constexpr int j0 = decltype(j0_c)::value;
        if (!c_valid) return;
        const unsigned sb = lds_base + (unsigned)(c_par & 1) * G::STAGE + dB;
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j) { dma16(voB[j], rsB, sb + j * 1024); voB[j] += stepB; }
    };
    // counted wait: everything this wave issued except the youngest `n0` (first wave row) / `n1` (second) pieces has landed; once the cursor
    // has run off the end of the work the assumed younger pieces no longer exist, so everything is waited for.  (Loads return in order; a
    // store in flight -- the previous tile's output -- only makes the wait stricter, never weaker.)
    auto wait_vm = [&](auto n0_c, auto n1_c) {
        constexpr int n0 = decltype(n0_c)::value, n1 = decltype(n1_c)::value;
#ifdef DIC_PP_NOVMWAIT
        return;
#endif
        if (!c_valid) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n0 == n1 || wm == 0) asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(n0) : "memory");
        else asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(n1) : "memory");
    };
    auto bar = []() {
        __builtin_amdgcn_sched_barrier(0);
#ifndef DIC_PP_NOBAR
        asm volatile("s_barrier" ::: "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- fragment reads (inline asm: neither order nor waits are the compiler's to change) -----------------------------------------------
    auto read_kc = [](bf16x8& dst, unsigned addr, auto imm_c) {
        constexpr int imm = decltype(imm_c)::value;
        i32x4 v;
#ifdef DIC_PP_NOREAD
        asm volatile("; no read %0 %1" : "=v"(v) : "v"(addr));
#else
        asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(v) : "v"(addr), "i"(imm));
#endif
        dst = __builtin_bit_cast(bf16x8, v);
    };
    auto read_km = [](bf16x8& dst, unsigned addr, auto imm_c, auto hi_c) {
        constexpr int imm = decltype(imm_c)::value, hi = decltype(hi_c)::value;
        s16x4 lo, hi4;
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c3\n\tds_read_b64_tr_b16 %1, %2 offset:%c4" : "=&v"(lo), "=v"(hi4) : "v"(addr), "i"(imm), "i"(imm + hi));
        dst = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    f32x4 acc[CNT][G::FN];
    bf16x8 fb[2][G::FN], fa[2][4];          // fb[32-deep half][column fragment]; fa[phase parity][fragment of the phase]
    constexpr int RA = AKM ? 2 : 1, RB = BKM ? 2 : 1;                  // LDS instructions per fragment read
    constexpr int cap15 = 15;
    constexpr int N_FB = (FH * RA + G::FN * RB) < cap15 ? (FH * RA + G::FN * RB) : cap15;      // reads of a "front + B" batch (lgkmcnt counts to 15)
    constexpr int N_BK = BH * RA;
    auto rd_a = [&](bf16x8& dst, unsigned sA, auto i_c, auto kk_c) {     // fragment i (0..CNT-1), 32-deep half kk, of the A tile at LDS address sA
        constexpr int i = decltype(i_c)::value, kk = decltype(kk_c)::value;
        if constexpr (!AKM) read_kc(dst, sA + ((unsigned)ofA[0] ^ (unsigned)(kk * 64)), std::integral_constant<int, i * 2048>{});
        else read_km(dst, sA + (unsigned)ofA[i], std::integral_constant<int, kk * 32 * ROWB_A>{}, std::integral_constant<int, 4 * ROWB_A>{});
    };
    auto rd_b = [&](bf16x8& dst, unsigned sB, auto j_c, auto kk_c) {
        constexpr int j = decltype(j_c)::value, kk = decltype(kk_c)::value;
        if constexpr (!BKM) read_kc(dst, sB + ((unsigned)ofB[0] ^ (unsigned)(kk * 64)), std::integral_constant<int, (32 * (j >> 1) + 4 * (j & 1)) * 128>{});
        else read_km(dst, sB + (unsigned)ofB[j], std::integral_constant<int, kk * 32 * ROWB_B>{}, std::integral_constant<int, 4 * ROWB_B>{});
    };
    // batch reads: `first` A fragment, `n` fragments, half kk -> fa[set][0..n-1]; all four B fragments of half kk -> fb[kk]
    auto read_a_batch = [&](unsigned sA, auto first_c, auto n_c, auto kk_c, auto set_c) {
        constexpr int first = decltype(first_c)::value, n = decltype(n_c)::value, set = decltype(set_c)::value;
        if constexpr (n >= 1) rd_a(fa[set][0], sA, std::integral_constant<int, first>{}, kk_c);
        if constexpr (n >= 2) rd_a(fa[set][1], sA, std::integral_constant<int, first + 1>{}, kk_c);
        if constexpr (n >= 3) rd_a(fa[set][2], sA, std::integral_constant<int, first + 2>{}, kk_c);
        if constexpr (n >= 4) rd_a(fa[set][3], sA, std::integral_constant<int, first + 3>{}, kk_c);
    };
    auto read_b_batch = [&](unsigned sB, auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
        rd_b(fb[kk][0], sB, std::integral_constant<int, 0>{}, kk_c); rd_b(fb[kk][1], sB, std::integral_constant<int, 1>{}, kk_c);
        rd_b(fb[kk][2], sB, std::integral_constant<int, 2>{}, kk_c); rd_b(fb[kk][3], sB, std::integral_constant<int, 3>{}, kk_c);
    };
    // fused bias gradient (weight gradients): column sums of the A tile that is in LDS anyway.  Thread -> source chunk cs = tid & 15 (8 tile rows)
    // of k-rows (tid >> 4) and (tid >> 4) + 32 of each half tile; the chunk sits at position cs ^ key(k).
    bool do_cs = false;
    f32x4 csF0{0.f, 0.f, 0.f, 0.f}, csF1 = csF0, csK0 = csF0, csK1 = csF0;
    i32x4 csr0, csr1;
    auto cs_read = [&csr0, &csr1, tid](unsigned sA, auto half_c) {      // (explicit captures: an asm operand alone does not make a generic lambda capture)
        constexpr int half = decltype(half_c)::value;
        const int k0 = tid >> 4, cs = tid & 15;
        const unsigned a0 = sA + half * 16384 + k0 * 256 + ((cs ^ km_key(k0)) << 4);      // km_key(k + 32) == km_key(k)
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192" : "=&v"(csr0), "=v"(csr1) : "v"(a0));
    };
    auto cs_add = [&csr0, &csr1](f32x4& s0, f32x4& s1) {
        f32x4 a, b;
        unpack8(csr0, a, b); s0 += a; s1 += b;
        unpack8(csr1, a, b); s0 += a; s1 += b;
    };
    // Counted wait at the end of a load segment: the operands of the NEXT compute segment (read one segment earlier) have returned; the n
    // LDS instructions issued in this segment may still be in flight (LDS returns in order).  The statement names what it validates, so no
    // consumer can be scheduled above it.
    auto wait_ops = [&fa, &fb, &csr0, &csr1](auto set_c, auto n_c, auto nfrag_c, auto with_b_c) {
        constexpr int set = decltype(set_c)::value, n = decltype(n_c)::value < 15 ? decltype(n_c)::value : 15, nf = decltype(nfrag_c)::value;
        constexpr bool wb = decltype(with_b_c)::value;
        // (only the registers the coming compute segment reads are named: naming a dead register would keep it allocated)
        if constexpr (wb) asm volatile("; operands B" : "+v"(fb[set][0]), "+v"(fb[set][1]), "+v"(fb[set][2]), "+v"(fb[set][3]));
        if constexpr (COLSUM) asm volatile("; operands cs" : "+v"(csr0), "+v"(csr1));
        if constexpr (nf == 4) asm volatile("s_waitcnt lgkmcnt(%c4)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fa[set][2]), "+v"(fa[set][3]) : "i"(n));
        else if constexpr (nf == 3) asm volatile("s_waitcnt lgkmcnt(%c3)" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fa[set][2]) : "i"(n));
        else if constexpr (nf == 2) asm volatile("s_waitcnt lgkmcnt(%c2)" : "+v"(fa[set][0]), "+v"(fa[set][1]) : "i"(n));
        else if constexpr (nf == 1) asm volatile("s_waitcnt lgkmcnt(%c1)" : "+v"(fa[set][0]) : "i"(n));
        else asm volatile("s_waitcnt lgkmcnt(%c0)" ::"i"(n));
        if constexpr (wb) asm volatile("; operands B" : "+v"(fb[set][0]), "+v"(fb[set][1]), "+v"(fb[set][2]), "+v"(fb[set][3]));
        if constexpr (COLSUM) asm volatile("; operands cs" : "+v"(csr0), "+v"(csr1));
    };
    // compute segment: n fragments starting at `first`, operands fa[set] / fb[set]
    auto phase_mfma = [&](auto first_c, auto n_c, auto set_c) {
        constexpr int first = decltype(first_c)::value, n = decltype(n_c)::value, set = decltype(set_c)::value;
#ifndef DIC_PP_NOPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int s = 0; s < n; ++s)
#pragma unroll
            for (int j = 0; j < G::FN; ++j) {
#ifdef DIC_PP_NOMFMA
                asm volatile("; no mfma %0 %1 %2" : "+v"(acc[first + s][j]) : "v"(fb[set][j]), "v"(fa[set][s]));
#else
                acc[first + s][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[set][j], fa[set][s], acc[first + s][j], 0, 0, 0);
#endif
            }
#ifndef DIC_PP_NOPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
#ifdef DIC_PP_TRACE      // measurement build (scripts/experiments/pp_trace.py): s_memtime of waves 0 and 4 of workgroup 0 around every barrier of K-steps 4..11
    unsigned long long* trace = (unsigned long long*)p.tgt_logit;
    int trace_n = -1;        // >= 0: recording
#define PP_STAMP() do { if (trace_n >= 0 && trace_n < 256) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if (lane == 0) trace[(wave >> 2) * 256 + trace_n] = t__; ++trace_n; } } while (0)
#else
#define PP_STAMP() do { } while (0)
#endif
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using IFH = std::integral_constant<int, FH>;
    using IBH = std::integral_constant<int, BH>;
    using Tt = std::true_type;
    using Ff = std::false_type;
    // The operands of a compute segment are read one load segment EARLIER (into the other fa set / the other half of fb), so a load segment
    // never waits for the latency of its own reads -- it only issues -- and the compute order is: front fragments x first half, front x second
    // half, back x first half, back x second half (both halves of an accumulator are two segments apart).
    //   L0: reads [front, B] second half        | waits back(k) landed        | M0: front x half 0
    //   L1: reads back first half               | DMA back(k+1), cursor on    | M1: front x half 1
    //   L2: reads back second half              | waits front/B(k+1) landed, DMA front(k+2) + half of B(k+2) | M2: back x half 0
    //   L3: reads [front, B] first half of k+1  | DMA rest of B(k+2)          | M3: back x half 1
    auto kstep = [&](int stage, bool fresh, bool last) {
        const unsigned sbA = lds_base + (unsigned)stage * G::STAGE, sbB = sbA + G::A_BYTES;
        const unsigned nbA = lds_base + (unsigned)(stage ^ 1) * G::STAGE, nbB = nbA + G::A_BYTES;
        // ---- L0 | M0
        read_a_batch(sbA, I0{}, IFH{}, I1{}, I1{});
        read_b_batch(sbB, I1{});
        if constexpr (COLSUM) { if (do_cs) cs_read(sbA, I0{}); }
        if (!fresh) wait_vm(std::integral_constant<int, NF + 4>{}, std::integral_constant<int, NF + 4>{});
        if constexpr (COLSUM) { if (do_cs) wait_ops(I0{}, std::integral_constant<int, N_FB + 2>{}, IFH{}, Tt{}); else wait_ops(I0{}, std::integral_constant<int, N_FB>{}, IFH{}, Tt{}); }
        else wait_ops(I0{}, std::integral_constant<int, N_FB>{}, IFH{}, Tt{});
        PP_STAMP(); bar(); PP_STAMP();
        phase_mfma(I0{}, IFH{}, I0{});
        PP_STAMP(); bar(); PP_STAMP();
        // ---- L1 | M1
        read_a_batch(sbA, IFH{}, IBH{}, I0{}, I0{});
        issue_back();
        advance();
        wait_ops(I1{}, std::integral_constant<int, N_BK>{}, IFH{}, Tt{});
        if constexpr (COLSUM) { if (do_cs) cs_add(csF0, csF1); }
        PP_STAMP(); bar(); PP_STAMP();
        phase_mfma(I0{}, IFH{}, I1{});
        PP_STAMP(); bar(); PP_STAMP();
        // ---- L2 | M2
        read_a_batch(sbA, IFH{}, IBH{}, I1{}, I1{});
        if constexpr (COLSUM) { if (do_cs) cs_read(sbA, I1{}); }
        if (!fresh) wait_vm(std::integral_constant<int, NBK0>{}, std::integral_constant<int, NBK1>{});
        issue_front();
        issue_b(I0{});
        if constexpr (COLSUM) { if (do_cs) wait_ops(I0{}, std::integral_constant<int, N_BK + 2>{}, IBH{}, Ff{}); else wait_ops(I0{}, std::integral_constant<int, N_BK>{}, IBH{}, Ff{}); }
        else wait_ops(I0{}, std::integral_constant<int, N_BK>{}, IBH{}, Ff{});
        PP_STAMP(); bar(); PP_STAMP();
        phase_mfma(IFH{}, IBH{}, I0{});
        PP_STAMP(); bar(); PP_STAMP();
        // ---- L3 | M3
        if (!last) {
            read_a_batch(nbA, I0{}, IFH{}, I0{}, I0{});
            read_b_batch(nbB, I0{});
        }
        issue_b(I2{});
        if (!last) wait_ops(I1{}, std::integral_constant<int, N_FB>{}, IBH{}, Ff{}); else wait_ops(I1{}, I0{}, IBH{}, Ff{});
        if constexpr (COLSUM) { if (do_cs) cs_add(csK0, csK1); }
        PP_STAMP(); bar(); PP_STAMP();
        phase_mfma(IFH{}, IBH{}, I1{});
        PP_STAMP(); bar(); PP_STAMP();
    };

    // ---- prologue: the stream's first two K-steps (all of the first; B and front of the second) -------------------------------------------
    cursor_open();
    issue_b(I0{}); issue_b(I2{}); issue_front(); issue_back();
    advance();
    issue_b(I0{}); issue_b(I2{}); issue_front();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();

    int unit = blockIdx.x, par = 0;
    bool fresh = true;
    for (;;) {
        const TileId tl = compute_tile(unit);
#pragma unroll
        for (int i = 0; i < CNT; ++i)
#pragma unroll
            for (int j = 0; j < G::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (COLSUM) {
            do_cs = p.colsum_out != nullptr && tl.bn == 0;
            csF0 = f32x4{0.f, 0.f, 0.f, 0.f}; csF1 = csF0; csK0 = csF0; csK1 = csF0;
        }
        if (tl.kt0 < tl.kt1) {                  // the first compute segment's operands (later ones are read one segment ahead inside the loop)
            const unsigned sbA0 = lds_base + (unsigned)(par & 1) * G::STAGE;
            read_a_batch(sbA0, I0{}, IFH{}, I0{}, I0{});
            read_b_batch(sbA0 + G::A_BYTES, I0{});
        }
        if (wm == 1) bar();                     // second wave row: half a phase behind
        for (int kt = tl.kt0; kt < tl.kt1; ++kt) {
#ifdef DIC_PP_TRACE
            if (trace && blockIdx.x == 0 && (wave & 3) == 0 && unit == (int)blockIdx.x) { if (kt - tl.kt0 == 4) trace_n = 0; if (kt - tl.kt0 == 12) trace_n = -1; }
#endif
            kstep(par & 1, fresh, kt + 1 == tl.kt1);
            fresh = false;
            ++par;
        }
        if (wm == 0) bar();                     // re-aligned: both rows write their output together
        DicGemmParams pe = p;
        if constexpr (GROUP) {
            pe.C = grp->ws + (size_t)slab * (G::BM * G::BN + G::BM);
            pe.ldc = G::BN; pe.M = G::BM; pe.N = G::BN; pe.out_f32 = 1; pe.accumulate = 0; pe.bias = nullptr; pe.R = nullptr; pe.p_drop = 0.f;
        } else {
            if (p.split_k > 1) redirect_to_slab(pe, tl.kz);
        }
        if constexpr (COLSUM) {
            if (do_cs) {
                // lanes l, l+16, l+32, l+48 hold the same rows for different k: fold them, then the 8 waves through LDS in fixed order.  The
                // scratch is the back region of the stage the next K-step does NOT use: its refill is issued in that K-step's L0.
                auto fold4 = [](f32x4& v) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] += __shfl_xor(v[r], 16, 64); v[r] += __shfl_xor(v[r], 32, 64); }
                };
                fold4(csF0); fold4(csF1); fold4(csK0); fold4(csK1);
                float* red = (float*)(smem + ((par & 1) ^ 1) * G::STAGE + 16384);
                if (lane < 16) {
                    const int m = lane < 8 ? 8 * lane : 128 + 8 * (lane - 8);
                    float* r_ = red + wave * G::BM;
                    *(f32x4*)(r_ + m) = csF0; *(f32x4*)(r_ + m + 4) = csF1;
                    *(f32x4*)(r_ + 64 + m) = csK0; *(f32x4*)(r_ + 64 + m + 4) = csK1;
                }
                barrier_lds_only();
                if (tid < G::BM) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < G::NW; ++w) v += red[w * G::BM + tid];
                    const int m = tl.bm * G::BM + tid;
                    if constexpr (GROUP) ((float*)pe.C)[G::BM * G::BN + tid] = v;
                    else if (m < p.M) {
                        if (p.split_k > 1) ((float*)pe.C)[(size_t)p.M * p.ldc + m] = v;
                        else p.colsum_out[m] = p.accumulate ? p.colsum_out[m] + v : v;
                    }
                }
                barrier_lds_only();
            }
        }
        // everything the stream has issued (the next tile's first K-steps) has landed BEFORE the first store of this tile goes out
        auto pre_store = []() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        const int m_first = (GROUP ? 0 : tl.bm * tile_rows) + row0_w, n_first = (GROUP ? 0 : tl.bn * G::BN) + wn * G::WCOLS;
        if constexpr (EPI == DIC_EPI_CE_PARTIAL) {
            epilogue_ce_partial<C, CNT>(acc, pe, m_first, n_first, wn, lane, tl.bn, tl.nbn, pre_store);
        } else {
            epilogue_direct<C, EPI, !AKM, CNT, false>(acc, pe, m_first, n_first, lane, pre_store, []() {});
        }
        fresh = true;
        unit += (int)gridDim.x;
        if (unit >= total) break;
    }
}

template <bool AKM, bool BKM, int EPI, int CNT>
__global__ __launch_bounds__(Geo<T256>::NTH, 2) void gemm_pp_kernel(DicGemmParams p) {
    if (p.step_ctr) p.seed += (uint64_t)(p.step_ctr[0] - p.step_ctr0) * DIC_STRIDE_DROP;
    gemm_pp_body<AKM, BKM, EPI, CNT, false>(p, nullptr);
}
__global__ __launch_bounds__(Geo<T256>::NTH, 2) void wgrad_group_pp_kernel(DicGemmParams p, WgradGroupDev grp) {
    gemm_pp_body<true, true, DIC_EPI_AFFINE, Geo<T256>::FM, true>(p, &grp);
}
