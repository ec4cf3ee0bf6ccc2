// gemm_w4.h -- 256 x 256 x 64 tiles on FOUR waves (one per SIMD, up to 512 registers each): the configuration the vendor library's own
// assembly kernel uses for these shapes on gfx950 (rocprofv3 of torch.matmul: "Custom_Cijk_..._MT256x256x64_MI16x16x1", workgroup 256, 130 KB LDS;
// profiles/r03_vendor_gemm_reference.txt).  Measured experiment of round 3 for the k-contiguous (forward) layout; included by gemm.hip inside its
// anonymous namespace.  Selected with dic_gemm_set_variant(2) / DIC_GEMM_PP=2.
//
// Against the 8-wave geometry: a wave owns 128 (CNT x 16) rows x 128 columns = CNT x 8 accumulator fragments (256 registers at CNT = 8, which is
// why it needs the whole register file), reads 25 % less LDS per flop, and has no partner wave on its SIMD -- every ds_read, every LDS-DMA issue
// and every wait has to sit between its own MFMAs.  The K-step is one asm-ordered stream: B fragments of a 32-deep half (8 reads), A fragments
// three ahead of their use, eight MFMAs per A fragment, and ONE LDS-DMA piece of the next stage behind every MFMA group (16 pieces per K-step
// and wave: a burst at the head of the K-step, as the 8-wave kernel issues it, would stop this wave's matrix pipe for ~1 500 cycles).

template <int EPI, int CNT>
__device__ __forceinline__ void gemm_w4_body(DicGemmParams& p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Geo<T256>;                                  // LDS image and the per-64-column output mapping are the 8-wave geometry's
    using T = bf16_t;
    constexpr int S = 2, BK = 64, NW = 4, FNW = 8;        // waves, B fragments per wave (128 columns)
    static_assert(CNT >= 4 && CNT <= 8, "A fragments per wave");
    constexpr int tile_rows = 32 * CNT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, t = lane & 15;
    const int row0_w = wm * CNT * 16;
    int ofA, ofB;
    { const int row = row0_w + t; ofA = row * 128 + ((g ^ key_a(row)) << 4); }
    { const int row = wn * 128 + 8 * (t >> 2) + (t & 3); ofB = row * 128 + ((g ^ key_b(row)) << 4); }

    i32x4 rsA, rsB;
    auto make_rsrc = [](const void* base, long long bytes) {
        const unsigned long long b = (unsigned long long)base;
        i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        r[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
        r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
        r[3] = 0x00020000;
        return r;
    };
    auto dma16 = [](unsigned voff, const i32x4& rsrc, unsigned lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
    };
    auto dma_barrier = []() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    constexpr int PA = CNT, PB = 8, NPC = PA + PB;        // 1 KiB pieces per wave and K-step: A rows (8 per piece, dealt round-robin), B rows
    unsigned voA[PA], voB[PB];
    auto setup = [&](const TileId& tl) {
        const int m0 = tl.bm * tile_rows, n0 = tl.bn * G::BN;
        const T* Ab = (const T*)p.A + (size_t)m0 * p.lda;
        const T* Bb = (const T*)p.B + (size_t)n0 * p.ldb;
        long long a_bytes = ((long long)(p.M - m0 - 1) * p.lda + p.K) * S, b_bytes = ((long long)(p.N - n0 - 1) * p.ldb + p.K) * S;
        if (a_bytes > 0xFFFFFFF0ll) a_bytes = 0xFFFFFFF0ll;
        if (b_bytes > 0xFFFFFFF0ll) b_bytes = 0xFFFFFFF0ll;
        rsA = make_rsrc(Ab, a_bytes);
        rsB = make_rsrc(Bb, b_bytes);
        const unsigned k0 = (unsigned)tl.kt0 * BK * 2u;
#pragma unroll
        for (int j = 0; j < PA; ++j) { const int q = j * NW + wave, row = 8 * q + (lane >> 3); voA[j] = k0 + (unsigned)row * (unsigned)p.lda * 2u + (((lane & 7) ^ key_a(row)) << 4); }
#pragma unroll
        for (int j = 0; j < PB; ++j) { const int q = wave * PB + j, row = 8 * q + (lane >> 3); voB[j] = k0 + (unsigned)row * (unsigned)p.ldb * 2u + (((lane & 7) ^ key_b(row)) << 4); }
    };
    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_PTR(char))smem;
    auto piece = [&](auto k_c, int stage) {               // piece k of the wave's NPC pieces into `stage`; advances its source by one K-step
        constexpr int k = decltype(k_c)::value;
        const unsigned sb = lds_base + (unsigned)stage * G::STAGE;
        if constexpr (k < PA) { dma16(voA[k], rsA, sb + (unsigned)(k * NW + wave) * 1024u); voA[k] += BK * 2u; }
        else { dma16(voB[k - PA], rsB, sb + G::A_BYTES + (unsigned)(wave * PB + (k - PA)) * 1024u); voB[k - PA] += BK * 2u; }
    };
    auto issue_all = [&](int stage) {
        [&]<int... Ks>(std::integer_sequence<int, Ks...>) { (piece(std::integral_constant<int, Ks>{}, stage), ...); }(std::make_integer_sequence<int, NPC>{});
    };

    f32x4 acc[CNT][FNW];
    auto compute = [&](int stage, bool more_k) {
        const unsigned sbA = lds_base + (unsigned)stage * G::STAGE, sbB = sbA + G::A_BYTES;
        const unsigned aA[2] = {sbA + (unsigned)ofA, sbA + ((unsigned)ofA ^ 64u)}, aB[2] = {sbB + (unsigned)ofB, sbB + ((unsigned)ofB ^ 64u)};
        constexpr int NS = 2 * CNT, PFD = 3, RING = PFD + 1;
        bf16x8 fb[2][FNW], fa[RING];
        auto read_frag = [&](bf16x8& dst, unsigned addr, auto imm_c) {
            constexpr int imm = decltype(imm_c)::value;
            i32x4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(v) : "v"(addr), "i"(imm));
            dst = __builtin_bit_cast(bf16x8, v);
        };
        auto issue_b = [&](auto kk_c) {
            constexpr int kk = decltype(kk_c)::value;
            [&]<int... Js>(std::integer_sequence<int, Js...>) {
                (read_frag(fb[kk][Js], aB[kk], std::integral_constant<int, (32 * (Js >> 1) + 4 * (Js & 1)) * 128>{}), ...);
            }(std::make_integer_sequence<int, FNW>{});
        };
        auto issue_a = [&](auto s_c) {
            constexpr int sidx = decltype(s_c)::value, kk = sidx / CNT, i = sidx % CNT;
            if constexpr (sidx == CNT) issue_b(std::integral_constant<int, 1>{});
            read_frag(fa[sidx % RING], aA[kk], std::integral_constant<int, i * 2048>{});
        };
        auto step = [&](auto s_c) {
            constexpr int sidx = decltype(s_c)::value, kk = sidx / CNT, i = sidx % CNT;
            constexpr int last = (sidx + PFD - 1 < NS - 1) ? sidx + PFD - 1 : NS - 1;          // reads issued after A_sidx when this step starts
            constexpr int after = (last - sidx) + ((sidx < CNT && last >= CNT) ? FNW : 0);
            static_assert(after <= 15, "lgkmcnt is a 4-bit counter");
            if constexpr (i == 0) {
                asm volatile("; B operands" : "+v"(fb[kk][4]), "+v"(fb[kk][5]), "+v"(fb[kk][6]), "+v"(fb[kk][7]));
                asm volatile("s_waitcnt lgkmcnt(%c5)" : "+v"(fa[sidx % RING]), "+v"(fb[kk][0]), "+v"(fb[kk][1]), "+v"(fb[kk][2]), "+v"(fb[kk][3]) : "i"(after));
                asm volatile("; B operands" : "+v"(fb[kk][4]), "+v"(fb[kk][5]), "+v"(fb[kk][6]), "+v"(fb[kk][7]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(%c1)" : "+v"(fa[sidx % RING]) : "i"(after));
            }
            // (a finer interleave -- one read / one DMA piece pinned between single MFMAs with sched_barrier -- made the register allocator
            // park accumulators in scratch inside the loop: 70 TFLOP/s; profiles/r03_gemm_w4_experiment.txt.  The order below is left to the scheduler.)
#pragma unroll
            for (int j = 0; j < FNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[sidx % RING], acc[i][j], 0, 0, 0);
            using std::integral_constant;
            if constexpr (sidx + PFD < NS) issue_a(integral_constant<int, sidx + PFD>{});      // fragment three steps ahead (ring slot of the step before this one)
            constexpr int k0 = sidx * NPC / NS, k1 = (sidx + 1) * NPC / NS;                     // the next stage's pieces, spread over the K-step
            if (more_k) {
                if constexpr (k1 > k0) piece(integral_constant<int, k0>{}, stage ^ 1);
                if constexpr (k1 > k0 + 1) piece(integral_constant<int, k0 + 1>{}, stage ^ 1);
            }
        };
        issue_b(std::integral_constant<int, 0>{});
        [&]<int... Is>(std::integer_sequence<int, Is...>) { (issue_a(std::integral_constant<int, Is>{}), ...); }(std::make_integer_sequence<int, PFD>{});
        [&]<int... Is>(std::integer_sequence<int, Is...>) { (step(std::integral_constant<int, Is>{}), ...); }(std::make_integer_sequence<int, NS>{});
    };

    const int total = total_units(p, tile_rows, G::BN);
    int unit = blockIdx.x;
    TileId tl = tile_of_unit(p, BK, unit, tile_rows, G::BN);
    setup(tl);
    if (tl.kt0 < tl.kt1) issue_all(0);
    for (;;) {
        const bool more = unit + (int)gridDim.x < total;
        TileId tl_next = tl;
        if (more) tl_next = tile_of_unit(p, BK, unit + (int)gridDim.x, tile_rows, G::BN);
        dma_barrier();
#pragma unroll
        for (int i = 0; i < CNT; ++i)
#pragma unroll
            for (int j = 0; j < FNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int cur = 0;
        for (int kt = tl.kt0; kt < tl.kt1; ++kt) {
            compute(cur, kt + 1 < tl.kt1);
            dma_barrier();
            cur ^= 1;
        }
        DicGemmParams pe = p;
        if (p.split_k > 1) redirect_to_slab(pe, tl.kz);
        bool issued = false;
        auto issue_next = [&]() {
            if (more && !issued) { issued = true; setup(tl_next); if (tl_next.kt0 < tl_next.kt1) issue_all(0); }
        };
        const int m_first = tl.bm * tile_rows + row0_w;
#pragma unroll
        for (int slab = 0; slab < 2; ++slab) {            // the wave's two 64-column slabs go through the 8-wave geometry's epilogues unchanged
            f32x4 a4[CNT][4];
#pragma unroll
            for (int i = 0; i < CNT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) a4[i][j] = acc[i][4 * slab + j];
            const int n_first = tl.bn * G::BN + wn * 128 + 64 * slab;
            if constexpr (EPI == DIC_EPI_CE_PARTIAL) epilogue_ce_partial<T256, CNT>(a4, pe, m_first, n_first, 2 * wn + slab, lane, tl.bn, tl.nbn, issue_next);
            else epilogue_direct<T256, EPI, true, CNT, false>(a4, pe, m_first, n_first, lane, issue_next, []() {});
        }
        unit += (int)gridDim.x;
        tl = tl_next;
        if (!more) break;
    }
}

template <int EPI, int CNT>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(DicGemmParams p) {
    if (p.step_ctr) p.seed += (uint64_t)(p.step_ctr[0] - p.step_ctr0) * DIC_STRIDE_DROP;
    gemm_w4_body<EPI, CNT>(p);
}
