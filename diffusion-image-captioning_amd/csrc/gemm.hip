// MFMA GEMM for gfx950 with fused epilogues -- the FLOP carrier of the denoiser (K3,K5,K7,K8,K9,K11,K14).
//
//   C[m][n] = sum_k A(m,k) * B(n,k)           (then an epilogue)
//
// Operand storage is chosen per operand:
//   k-contiguous ("KC"):  X[row][k]   -- activations as A in the forward, Linear weights [out][in] as B
//   k-major      ("KM"):  X[k][row]   -- weights [out][in] as B of dX = dY.W, and both operands of dW = dY^T.X
// so forward (KC,KC), input-gradient (KC,KM) and weight-gradient (KM,KM) all run on this one kernel without
// ever materialising a transposed tensor in HBM.
//
// Tile: 128x128 per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 4x4 MFMA fragments),
// K-step 128 bytes (64 bf16 / 32 f32).  Global->register->LDS staging with `buffer_load_dwordx4`
// (the buffer descriptor's bounds check zero-fills rows beyond M/N/K, so ragged shapes need no branches),
// double-buffered LDS, one barrier per K-step.
//   bf16: v_mfma_f32_16x16x32_bf16.  KC fragments are two ds_read_b64 (k = 4g..4g+3 and 16+4g..), KM fragments
//         are two ds_read_b64_tr_b16 hardware-transpose reads giving the same k-slot mapping, so any mix of
//         KC/KM operands contracts consistently.  LDS row strides (144 B KC, 288 B KM) make both reads
//         bank-conflict-free.
//   f32:  v_mfma_f32_16x16x4_f32 -- exact fp32, accumulating k in ascending order (bit-identical to an
//         fmaf chain), which is what lets the rounding head reproduce the oracle's token ids bit-for-bit.
// The MFMA is issued with operands swapped (D = Bfrag x Afrag) so that each lane ends up holding 4
// CONSECUTIVE n for one m: epilogue loads/stores are 8-byte (bf16) or 16-byte (f32) vectors.
#include <cstring>
#include "common.h"
#include "../../include/dic_hip.h"
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include <hip/hip_ext.h>

#ifndef DIC_GEMM_PF
#define DIC_GEMM_PF 3            // A fragments read ahead of their MFMAs in the bf16 kernel
#endif
#ifndef DIC_GEMM_ISSUE_AT
#define DIC_GEMM_ISSUE_AT -1     // fragment step after whose MFMAs the next K-step's LDS-DMA is issued; < 0: before the fragment reads
#endif

namespace {

constexpr int BM = 128, BN = 128, NT = 256;
constexpr int TILE_BYTES = 18432;      // one operand tile in LDS (either layout, either dtype)
constexpr int KC_STRIDE = 144;         // bytes: 128 data + 16 pad
template <typename T> struct KCfg;
template <> struct KCfg<bf16_t> { static constexpr int BK = 64, KM_STRIDE = 288, KM_CHUNKS_LOG2 = 4; };
template <> struct KCfg<float>  { static constexpr int BK = 32, KM_STRIDE = 576, KM_CHUNKS_LOG2 = 5; };

__device__ __forceinline__ i32x4 buf_load16(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
}

// ---- staging: global -> 4 x 16B registers per operand per thread ---------------------------------
template <typename T, bool KM>
__device__ __forceinline__ void stage_load(i32x4 (&r)[4], __amdgpu_buffer_rsrc_t rs, int ld, int k0, int tid) {
    constexpr int S = sizeof(T);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int c = tid + NT * j;
        unsigned off;
        if (!KM) {
            int row = c >> 3, kc = c & 7;
            off = ((unsigned)row * (unsigned)ld + (unsigned)k0) * S + kc * 16;
        } else {
            constexpr int L2 = KCfg<T>::KM_CHUNKS_LOG2;
            int row = c >> L2, cc = c & ((1 << L2) - 1);
            off = ((unsigned)(k0 + row) * (unsigned)ld) * S + cc * 16;
        }
        r[j] = buf_load16(rs, off);
    }
}
template <typename T, bool KM>
__device__ __forceinline__ void stage_store(const i32x4 (&r)[4], char* lds, int tid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int c = tid + NT * j;
        int off;
        if (!KM) {
            off = (c >> 3) * KC_STRIDE + (c & 7) * 16;
        } else {
            constexpr int L2 = KCfg<T>::KM_CHUNKS_LOG2;
            off = (c >> L2) * KCfg<T>::KM_STRIDE + (c & ((1 << L2) - 1)) * 16;
        }
        *(i32x4*)(lds + off) = r[j];
    }
}

// ---- fragment reads ------------------------------------------------------------------------------
// bf16: returns the 8-element operand for k sub-step kk (32 k's) of 16 rows/cols starting at `base`.
template <bool KM>
__device__ __forceinline__ bf16x8 frag_bf16(const char* lds, int base, int kk, int lane) {
    int g = lane >> 4, t = lane & 15;
    s16x4 lo, hi;
    if (!KM) {
        const char* p = lds + (base + t) * KC_STRIDE + (kk * 32 + 4 * g) * 2;
        lo = *(const s16x4*)p;
        hi = *(const s16x4*)(p + 32);
    } else {
        const char* p = lds + (kk * 32 + 4 * g + (t >> 2)) * 288 + (base + 4 * (t & 3)) * 2;
        lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(p));
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(p + 16 * 288));
    }
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// f32: one value per lane for k = k4*4 + (lane>>4)
template <bool KM>
__device__ __forceinline__ float frag_f32(const char* lds, int base, int k4, int lane) {
    int g = lane >> 4, t = lane & 15;
    if (!KM) return *(const float*)(lds + (base + t) * KC_STRIDE + (k4 * 4 + g) * 4);
    return *(const float*)(lds + (k4 * 4 + g) * 576 + (base + t) * 4);
}

__device__ __forceinline__ float row_scale(const DicGemmParams& p, int m) { return m < p.ce_rows_a ? p.ce_scale_a : p.ce_scale_b; }

// ---- block -> (tile, K-slice) ----------------------------------------------------------------------
// XCD-aware order: consecutive logical ids (same A row-panel, then the K-slices of one tile) share one XCD's L2.
struct TileId { int bm, bn, nbn, kz, kt0, kt1; };
// (mrows: the number of rows the units cover when that is a row range of the problem instead of all p.M rows -- the two-height launch)
__device__ __forceinline__ int total_units(const DicGemmParams& p, int bm_ = BM, int bn_ = BN, int mrows = -1) {
    const int M_ = mrows >= 0 ? mrows : p.M;
    return ((p.N + bn_ - 1) / bn_) * ((M_ + bm_ - 1) / bm_) * (p.split_k > 1 ? p.split_k : 1);
}
__device__ __forceinline__ TileId tile_of_unit(const DicGemmParams& p, int BK, int pid, int bm_ = BM, int bn_ = BN, int mrows = -1) {
    const int M_ = mrows >= 0 ? mrows : p.M;
    const int nbn = (p.N + bn_ - 1) / bn_, nbm = (M_ + bm_ - 1) / bm_;
    const int split = p.split_k > 1 ? p.split_k : 1;
    const int nwg = nbm * nbn * split;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = pid & 7, slot = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    // Logical order (what one XCD's L2 sees over time), chosen from the rocprof FETCH_SIZE of the first version (the rounding
    // GEMM pulled 6 GB through the fabric for 72 MB of operands: with n fastest, every workgroup of a row-panel streamed a
    // different 196 KB slice of the 47 MB vocabulary matrix):
    //   * K-slice slowest: the workgroups resident together share one K range, hence the same A and B slices;
    //   * inside a K-slice, column groups of 8 tiles, m fastest across the group: the ~64 workgroups an XCD runs at once form
    //     an 8x8 patch (8 A panels + 8 B panels ~ 3 MB, inside the 4 MB L2) instead of a 1x64 strip (65 panels).
    TileId t;
    const int ntiles = nbm * nbn;
    t.kz = pid / ntiles;
    const int lin = pid - t.kz * ntiles;
#ifndef DIC_GEMM_GW
#define DIC_GEMM_GW 8
#endif
    constexpr int GW = DIC_GEMM_GW;
#ifndef DIC_GEMM_WIDE_ORDER
#define DIC_GEMM_WIDE_ORDER 1
#endif
    if (DIC_GEMM_WIDE_ORDER && nbn >= 4 * nbm) {
        // WIDE problems (the rounding head: 64 row tiles x 120 column tiles): the patch an XCD works on moves along the LONG dimension, so the
        // GW row panels of A it holds (8 x 393 KB at K = 768) stay in its L2 while the column panels of the vocabulary matrix stream through --
        // (nbm / GW) x |B| + |A| of L2 misses instead of (nbn / GW) x |A| + |B| (round 4; PMC: profiles/r04_pmc_hbm_traffic.txt)
        const int grp = lin / (GW * nbn), rem = lin - grp * (GW * nbn);
        const int h = min(GW, nbm - grp * GW);
        t.bn = rem / h;
        t.bm = grp * GW + (rem - t.bn * h);
    } else {
        const int grp = lin / (GW * nbm), rem = lin - grp * (GW * nbm);
        const int w = min(GW, nbn - grp * GW);
        t.bm = rem / w;
        t.bn = grp * GW + (rem - t.bm * w);
    }
    t.nbn = nbn;
    const int nk = (p.K + BK - 1) / BK, per = (nk + split - 1) / split;
    t.kt0 = t.kz * per;
    t.kt1 = min(nk, t.kt0 + per);
    return t;
}
__device__ __forceinline__ TileId tile_of_block(const DicGemmParams& p, int BK) { return tile_of_unit(p, BK, blockIdx.x); }
// split-K: slice kz writes its plain fp32 partial tile into slab kz of the workspace; dic_gemm folds the slabs after.
__device__ __forceinline__ size_t slab_stride(const DicGemmParams& p) { return (size_t)p.M * p.ldc + (p.colsum_out ? p.M : 0); }
__device__ __forceinline__ void redirect_to_slab(DicGemmParams& p, int kz) {
    p.C = (float*)p.split_ws + (size_t)kz * slab_stride(p);
    p.bias = nullptr; p.R = nullptr; p.p_drop = 0.f; p.out_f32 = 1; p.accumulate = 0;
}

// ---- epilogues, shared by both kernels -----------------------------------------------------------
// acc[i][j][r]  <->  m = m0 + wm*64 + i*16 + (lane&15),  n = n0 + wn*64 + j*16 + (lane>>4)*4 + r
template <typename T, int EPI>
__device__ __forceinline__ void epilogue(f32x4 (&acc)[4][4], const DicGemmParams& p, int m0, int n0, int wm, int wn, int lane, int bn, int nbn) {
    const int g = lane >> 4, t = lane & 15;
    if constexpr (EPI == DIC_EPI_AFFINE || EPI == DIC_EPI_BIAS_GELU || EPI == DIC_EPI_GELU_BWD) {
        const float inv_keep = drop_inv_keep(p.p_drop);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + t;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + g * 4;
                if (n >= p.N) continue;
                f32x4 v = acc[i][j];
                if constexpr (EPI == DIC_EPI_AFFINE) {
                    if (p.bias) v += *(const f32x4*)(p.bias + n);
                    if (p.p_drop > 0.f) v = dropout4(v, p.seed, (unsigned long long)m * p.N + n, p.p_drop, inv_keep);
                    if (p.R) v += Elem<T>::ld4((const T*)p.R + (size_t)m * p.ldr + n);
                    if (p.out_f32) {
                        float* c = (float*)p.C + (size_t)m * p.ldc + n;
                        if (p.accumulate) v += *(const f32x4*)c;
                        *(f32x4*)c = v;
                    } else {
                        Elem<T>::st4((T*)p.C + (size_t)m * p.ldc + n, v);
                    }
                } else if constexpr (EPI == DIC_EPI_BIAS_GELU) {
                    v += *(const f32x4*)(p.bias + n);
                    if (p.aux) Elem<T>::st4((T*)p.aux + (size_t)m * p.ldaux + n, v);   // pre-activation u (for GELU'); NULL: forward-only call
                    f32x4 gl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) gl[r] = sizeof(T) == 2 ? gelu_fast(v[r]) : gelu_f(v[r]);
                    Elem<T>::st4((T*)p.C + (size_t)m * p.ldc + n, gl);
                } else {   // GELU_BWD: dU = acc * gelu'(U)
                    f32x4 u = Elem<T>::ld4((const T*)p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= sizeof(T) == 2 ? gelu_grad_fast(u[r]) : gelu_grad_f(u[r]);
                    Elem<T>::st4((T*)p.C + (size_t)m * p.ldc + n, v);
                }
            }
        }
    } else if constexpr (EPI == DIC_EPI_CE_PARTIAL) {
        // per (row, 64-column half-tile): running max / first argmax / sum exp(x - max); target logit scattered
        const int np = 2 * nbn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + t;
            const long long tg = (m < p.M && p.tgt) ? p.tgt[m] : -1;
            float mx = -INFINITY;
            int ix = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * 64 + j * 16 + g * 4 + r;
                    const float x = acc[i][j][r];
                    if (n < p.N) {
                        if (x > mx) { mx = x; ix = n; }
                        if ((long long)n == tg) p.tgt_logit[m] = x;
                    }
                }
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * 64 + j * 16 + g * 4 + r;
                    if (n < p.N) sm += __expf(acc[i][j][r] - mx);
                }
            // merge the 4 lanes (g = 0..3) that share row m
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                float mx2 = __shfl_xor(mx, o, 64), sm2 = __shfl_xor(sm, o, 64);
                int ix2 = __shfl_xor(ix, o, 64);
                float M = fmaxf(mx, mx2);
                float s1 = (mx == -INFINITY) ? 0.f : sm * __expf(mx - M);
                float s2 = (mx2 == -INFINITY) ? 0.f : sm2 * __expf(mx2 - M);
                ix = (mx > mx2) ? ix : (mx2 > mx) ? ix2 : min(ix, ix2);
                mx = M; sm = s1 + s2;
            }
            if (g == 0 && m < p.M) {
                float4 o4 = make_float4(mx, sm, __int_as_float(ix), 0.f);
                *(float4*)(p.partial + ((size_t)m * np + bn * 2 + wn) * 4) = o4;
            }
        }
    } else if constexpr (EPI == DIC_EPI_CE_DLOGITS) {
        // dlogits = (softmax - onehot) * row_scale ; columns >= V (padding up to ldc) are written as zeros
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + t;
            if (m >= p.M) continue;
            const float lse = p.lse[m], sc = row_scale(p, m);
            const long long tg = p.tgt[m];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + g * 4;
                if (n >= p.ldc) continue;
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = (n + r < p.N) ? __expf(acc[i][j][r] - lse) : 0.f;
                    if ((long long)(n + r) == tg) pr -= 1.0f;
                    v[r] = pr * sc;
                }
                Elem<T>::st4((T*)p.C + (size_t)m * p.ldc + n, v);
            }
        }
    }
}

template <typename T, bool AKM, bool BKM, int EPI>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(DicGemmParams p) {   // v1: register-staged; fp32 parity path (and bf16 A/B reference)
    if (p.step_ctr) p.seed += (uint64_t)(p.step_ctr[0] - p.step_ctr0) * DIC_STRIDE_DROP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int S = sizeof(T);
    constexpr int BK = KCfg<T>::BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    TileId tl = tile_of_block(p, BK);
    const int bm = tl.bm, bn = tl.bn, nbn = tl.nbn;
    const int m0 = bm * BM, n0 = bn * BN;
    if (p.split_k > 1) redirect_to_slab(p, tl.kz);

    // buffer descriptors anchored at the tile origin; num_records ends at the matrix end => OOB rows read 0
    const T* Ab = (const T*)p.A + (AKM ? (size_t)m0 : (size_t)m0 * p.lda);
    const T* Bb = (const T*)p.B + (BKM ? (size_t)n0 : (size_t)n0 * p.ldb);
    long long a_bytes = AKM ? ((long long)(p.K - 1) * p.lda + (p.M - m0)) * S : ((long long)(p.M - m0 - 1) * p.lda + p.K) * S;
    long long b_bytes = BKM ? ((long long)(p.K - 1) * p.ldb + (p.N - n0)) * S : ((long long)(p.N - n0 - 1) * p.ldb + p.K) * S;
    if (a_bytes > 0xFFFFFFF0ll) a_bytes = 0xFFFFFFF0ll;
    if (b_bytes > 0xFFFFFFF0ll) b_bytes = 0xFFFFFFF0ll;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)b_bytes, 0x00020000);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int kt0 = tl.kt0, nk = tl.kt1;
    i32x4 ra[4], rb[4];
    stage_load<T, AKM>(ra, rsA, p.lda, kt0 * BK, tid);
    stage_load<T, BKM>(rb, rsB, p.ldb, kt0 * BK, tid);
    stage_store<T, AKM>(ra, smem + (kt0 & 1) * (2 * TILE_BYTES), tid);
    stage_store<T, BKM>(rb, smem + (kt0 & 1) * (2 * TILE_BYTES) + TILE_BYTES, tid);
    __syncthreads();

    for (int kt = kt0; kt < nk; ++kt) {
        const char* la = smem + (kt & 1) * (2 * TILE_BYTES);
        const char* lb = la + TILE_BYTES;
        if (kt + 1 < nk) {
            stage_load<T, AKM>(ra, rsA, p.lda, (kt + 1) * BK, tid);
            stage_load<T, BKM>(rb, rsB, p.ldb, (kt + 1) * BK, tid);
        }
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = frag_bf16<AKM>(la, wm * 64 + i * 16, kk, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = frag_bf16<BKM>(lb, wn * 64 + j * 16, kk, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                float fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = frag_f32<AKM>(la, wm * 64 + i * 16, k4, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = frag_f32<BKM>(lb, wn * 64 + j * 16, k4, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) {
            char* na = smem + ((kt + 1) & 1) * (2 * TILE_BYTES);
            stage_store<T, AKM>(ra, na, tid);
            stage_store<T, BKM>(rb, na + TILE_BYTES, tid);
        }
        __syncthreads();
    }
    epilogue<T, EPI>(acc, p, m0, n0, wm, wn, lane, bn, nbn);
}

// ---- bf16 kernel: output mapping and epilogue -------------------------------------------------------------------------------------
// The MFMA is issued with the operands swapped (D = Bfrag x Afrag), so a lane holds 4 values of ONE output row for 4 of the B
// fragment's rows (= output columns).  Which LDS rows feed a B fragment is free, so fragment pair (2p, 2p+1) of a wave's 64 columns is
// loaded with the rows   n = 32 p + 8 (idx>>2) + 4 q + (idx&3)   (q = fragment parity, idx = 0..15 the fragment row): lane group
// g = lane>>4 then owns the 8 CONSECUTIVE columns 32 p + 8 g .. + 7 of its row across the pair, i.e. 16 bytes of bf16 output.  The
// accumulators go to HBM straight from registers as 16-byte stores (a wave-instruction covers 16 rows x 64 contiguous bytes) and the
// side inputs (residual, GELU' pre-activation) come in with the same shape.  Round 1 parked the fp32 tile in LDS to get row-contiguous
// stores (32 ds_write_b128 + 2 barriers + 16 ds_read_b128 per wave on the critical path of every tile, and the LDS was not free for
// the next tile's operands meanwhile); with 16 bytes per lane straight from the accumulators that detour is gone and the next tile's
// first K-step is already in flight while this one is written out.
__device__ __forceinline__ void unpack8(i32x4 r, f32x4& a, f32x4& b) {
    a[0] = __uint_as_float((unsigned)r[0] << 16); a[1] = __uint_as_float((unsigned)r[0] & 0xffff0000u);
    a[2] = __uint_as_float((unsigned)r[1] << 16); a[3] = __uint_as_float((unsigned)r[1] & 0xffff0000u);
    b[0] = __uint_as_float((unsigned)r[2] << 16); b[1] = __uint_as_float((unsigned)r[2] & 0xffff0000u);
    b[2] = __uint_as_float((unsigned)r[3] << 16); b[3] = __uint_as_float((unsigned)r[3] & 0xffff0000u);
}
__device__ __forceinline__ i32x4 pack8f(const f32x4& a, const f32x4& b) {
    i32x4 r;
    r[0] = (int)pack2bf(a[0], a[1]); r[1] = (int)pack2bf(a[2], a[3]);
    r[2] = (int)pack2bf(b[0], b[1]); r[3] = (int)pack2bf(b[2], b[3]);
    return r;
}

// Tile geometries of the bf16 kernel.  T128: 128x128, 4 waves (2x2, 64x64 each), 64 KB LDS, two workgroups per CU.
// T256: 256x256, 8 waves (2x4, 128x64 each), 128 KB LDS, one workgroup per CU -- twice the flop per byte pulled from L2
// (128 vs 64 flop/B: the 128x128 kernel saturates near 0.9 PFLOP/s on L2->LDS bandwidth) and 25 % fewer LDS reads per MFMA.
// BM is the MAXIMUM tile height: the launch picks the height actually used (a multiple of 16 rows) so that the tiles fill whole
// rounds of resident workgroups (pick_tile_rows below).
struct T128 { static constexpr int BM = 128, BN = 128, WM = 2, WN = 2; };
struct T256 { static constexpr int BM = 256, BN = 256, WM = 2, WN = 4; };
template <class C> struct Geo {
    static constexpr int BM = C::BM, BN = C::BN, WM = C::WM, WN = C::WN;
    static_assert(WM == 2, "tile rows are split over two wave rows");
    static constexpr int NW = WM * WN, NTH = 64 * NW;
    static constexpr int FM = BM / WM / 16, FN = BN / WN / 16;           // MFMA fragments per wave (FM: at most)
    static constexpr int NP = FN / 2;                                    // fragment pairs = 8-column groups per lane
    static constexpr int WCOLS = BN / WN;                                // columns per wave
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES, LDS = 2 * STAGE;
    static constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;             // 1 KiB DMA pieces per wave per operand
};
// output column (relative to the wave's first) of accumulator element r of fragment j held by lane group g
__device__ __forceinline__ int frag_col(int j, int g, int r) { return 32 * (j >> 1) + 8 * g + 4 * (j & 1) + r; }

// workgroup barrier that waits for this wave's LDS traffic only: outstanding global STORES (epilogue output) keep draining across
// it.  (__syncthreads() carries vmcnt(0), i.e. a full round trip of every store issued so far.)
__device__ __forceinline__ void barrier_lds_only() {
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0), vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
}

// Lanes t and t^8 of a 16-lane row trade one 16-byte chunk (DPP row_ror:8 with a bank mask: only half of the lanes take the rotated
// value, the others keep `keep` -- no select instructions).  HI: lanes 8..15 of each row receive, else lanes 0..7.
template <bool HI>
__device__ __forceinline__ i32x4 take8(const i32x4& keep, const i32x4& from_partner) {
    i32x4 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = __builtin_amdgcn_update_dpp(keep[r], from_partner[r], 0x128, 0xF, HI ? 0xC : 0x3, false);
    return y;
}

// One wave's accumulators -> HBM.  m_first: global row of the wave's fragment 0 (CNT fragments of 16 rows); n_first: global column of
// the wave's first column.
// gfx950 has ONE counter (vmcnt) for loads and stores, and they complete out of order with respect to each other: any wait for a
// load drains every store issued before it (measured in round 1: 12 us to write a 28 MB output that a fill kernel writes in 4 when a
// side input was loaded per row).  So all side inputs of the wave's tile are loaded first, one vmcnt(0) covers them, and the rest is
// math + stores with nothing to wait on.  (The bias is added to the accumulators before the side tile is requested: its registers
// are free again by then.)
// `issue_next` starts the next tile (address set-up + the first K-step's LDS-DMA).  It is called once, AFTER the last load this epilogue
// waits for and BEFORE its first store: a wait for the bias / residual / pre-activation loads is a `vmcnt(0)` and would otherwise also
// wait ~2.4 k cycles for that DMA (the s_memtime trace showed it in front of every epilogue), while behind the wait the DMA's latency
// hides under the stores.
template <class C, int EPI, bool PF, int CNT, bool BIAS_IN_ACC, class IssueNext, class Stamp>
__device__ __forceinline__ void epilogue_direct(f32x4 (&acc)[CNT][Geo<C>::FN], const DicGemmParams& p, int m_first, int n_first, int lane, IssueNext&& issue_next_,
                                                Stamp&& stamp) {
    auto issue_next = [&]() { stamp(); issue_next_(); stamp(); };          // (stamp: s_memtime in the trace build, nothing otherwise)
    using G = Geo<C>;
    using T = bf16_t;
    const int g = lane >> 4, t = lane & 15;
    // FULL-LINE STORES.  In the accumulator layout lane (g,t) owns, per 16-row fragment i, row t and the two 16-byte chunks g and 4+g of the
    // wave's 128-byte output row, so a store instruction per (i, chunk set) writes 16 rows x 64 B: half cache lines.  Measured on an
    // otherwise idle CU (scripts/experiments/stprobe.hip): 32 B/clk with that map, 82-86 B/clk when one instruction writes 8 rows x 128 B.
    // So lanes t and t^8 swap one chunk (4 DPP moves): instruction 0 then writes rows 0-7 of the fragment, instruction 1 rows 8-15,
    // lane (g,t) at chunk g + 4*(t>>3) of row (t&7).  Same trick for fp32 outputs (a lane's 8 columns are 32 B: chunks 2g, 2g+1 of the
    // 128-byte line of a 32-column group) and, mirrored, for the side-input loads (residual, GELU' pre-activation).
    // Streaming (nt) stores for the FFN pre-activation / activation pair only: 214 MB written per launch pushed the weight panels out of the
    // XCD's L2 between rounds of the persistent grid (PMC: 149 MB fetched for 31 MB of operands); with nt the launch is 14 % faster
    // (130 -> 112 us, cold operands).  NOT for the other outputs: split-K slabs are re-read by the fold right away (nt: +20-50 %), the
    // GELU' and residual epilogues measured 0-6 % slower with nt.
#ifndef DIC_CE_EXP_NT
#define DIC_CE_EXP_NT 1       // streaming stores for the rounding head's 1 GB of exp(logit - c): nothing reads it before the backward, and it must not push the operand panels out of L2
#endif
    constexpr bool NT = EPI == DIC_EPI_BIAS_GELU || EPI == DIC_EPI_BIAS_GELU_D || (DIC_CE_EXP_NT && EPI == DIC_EPI_CE_EXP);
    static_assert(G::NP == 2, "line stores pair the two 8-column groups of a 64-column wave slab");
    const int lrow = t & 7, lcol = 8 * (g + 4 * (t >> 3));          // row within an 8-row half fragment; column of this lane's chunk in line order
    // Buffer addressing (32-bit lane offsets against a descriptor anchored at the wave's first row): no 64-bit address arithmetic and no
    // exec-mask branches per store -- rows >= M fall outside the descriptor's range and are dropped (loads: read as zero) by the hardware,
    // a lane whose 16-byte chunk lies beyond the last column gets an offset that is out of range for every row.  (The pointer form
    // cost ~12 of the ~25 instructions per store, and the epilogue is issue-bound: s_memtime puts 4-13 k cycles of VALU per wave behind
    // every tile, two waves per SIMD.)
    struct LineBuf { __amdgpu_buffer_rsrc_t rs; unsigned off, row8; };
    auto line_buf = [&](const void* base, int ld, int es, int col, int n_lim) {
        long long bytes = (long long)(p.M - m_first) * ld * es;
        bytes = bytes < 0 ? 0 : (bytes > 0x7FFFFFFFll ? 0x7FFFFFFFll : bytes);          // 16 rows x ld never get near 2 GB; 0x80000000 + that never wraps
        LineBuf Lb;
        Lb.rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + (size_t)m_first * ld * es), 0, (int)bytes, 0x00020000);
        Lb.row8 = 8u * (unsigned)ld * (unsigned)es;
        Lb.off = col < n_lim ? ((unsigned)lrow * (unsigned)ld + (unsigned)col) * (unsigned)es : 0x80000000u;
        return Lb;
    };
    constexpr int AUX = NT ? 2 : 0;                                   // cache-policy bits of the buffer instruction: 2 = nt
    // bf16 row pair of fragment i: P0 / P1 = this lane's chunks for column groups q = 0 / 1
    auto put_lines = [&](int i, const LineBuf& Lb, const i32x4& P0, const i32x4& P1) {
        const i32x4 D0 = take8<true>(P0, P1), D1 = take8<false>(P1, P0);      // rows 0-7: lanes t >= 8 carry lane t-8's second chunk; rows 8-15: lanes t < 8 carry lane t+8's first
        const unsigned o = Lb.off + (unsigned)(2 * i) * Lb.row8;
        __builtin_amdgcn_raw_buffer_store_b128(D0, Lb.rs, (int)o, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b128(D1, Lb.rs, (int)(o + Lb.row8), 0, AUX);
    };
    // mirrored load: full-line loads now, this lane's chunks (q = 0, q = 1) of row t of fragment i after get_lines_finish
    auto get_lines_issue = [&](int i, const LineBuf& Lb, i32x4& L0, i32x4& L1) {
        const unsigned o = Lb.off + (unsigned)(2 * i) * Lb.row8;
        L0 = __builtin_amdgcn_raw_buffer_load_b128(Lb.rs, (int)o, 0, 0);
        L1 = __builtin_amdgcn_raw_buffer_load_b128(Lb.rs, (int)(o + Lb.row8), 0, 0);
    };
    auto get_lines_finish = [&](i32x4& L0, i32x4& L1) {          // in place: L0 -> chunk of q = 0, L1 -> chunk of q = 1
        const i32x4 Q0 = take8<true>(L0, L1), Q1 = take8<false>(L1, L0);
        L0 = Q0; L1 = Q1;
    };
    // fp32 pair of (fragment i, column group q): x0 / x1 = this lane's columns 8g..8g+3 / 8g+4..8g+7; Lb is built for column group q
    auto put_lines_f32 = [&](int i, const LineBuf& Lb, const f32x4& x0, const f32x4& x1) {
        const i32x4 X0 = __builtin_bit_cast(i32x4, x0), X1 = __builtin_bit_cast(i32x4, x1);
        const i32x4 D0 = take8<true>(X0, X1), D1 = take8<false>(X1, X0);
        const unsigned o = Lb.off + (unsigned)(2 * i) * Lb.row8;
        __builtin_amdgcn_raw_buffer_store_b128(D0, Lb.rs, (int)o, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(D1, Lb.rs, (int)(o + Lb.row8), 0, 0);
    };
    const int fcol = 4 * (2 * g + (t >> 3));                          // this lane's 4-column chunk inside a 32-column group (fp32 line order)
    int nc[G::NP];
    bool v0[G::NP], v1[G::NP];
#pragma unroll
    for (int q = 0; q < G::NP; ++q) { nc[q] = n_first + 32 * q + 8 * g; v0[q] = nc[q] < p.N; v1[q] = nc[q] + 4 < p.N; }
    if constexpr ((EPI == DIC_EPI_AFFINE || EPI == DIC_EPI_BIAS_GELU || EPI == DIC_EPI_BIAS_GELU_D) && !BIAS_IN_ACC) {       // (forward Linears: the accumulators started from the bias)
        if (p.bias) {
#pragma unroll
            for (int q = 0; q < G::NP; ++q) {
                f32x4 b0{0.f, 0.f, 0.f, 0.f}, b1{0.f, 0.f, 0.f, 0.f};
                if (v0[q]) b0 = *(const f32x4*)(p.bias + nc[q]);
                if (v1[q]) b1 = *(const f32x4*)(p.bias + nc[q] + 4);
#pragma unroll
                for (int i = 0; i < CNT; ++i) { acc[i][2 * q] += b0; acc[i][2 * q + 1] += b1; }
            }
        }
    }
    if constexpr (EPI == DIC_EPI_AFFINE) {
        const float inv_keep = drop_inv_keep(p.p_drop);
        // fp32 output + fp32 residual (+ dropout): the fp32 residual stream of the bf16 engines (include/dic_hip.h, DIC_RES_F32)
        const bool r32 = PF && p.R != nullptr && (p.out_f32 & DIC_RES_IS_F32) != 0 && (p.N & 7) == 0 && !p.accumulate;
        const bool general = p.accumulate || (p.R != nullptr && !PF) || ((p.N & 7) != 0 && (p.R != nullptr || !p.out_f32)) ||
                             (p.out_f32 && (p.R != nullptr || p.p_drop > 0.f) && !r32);   // other fp32 outputs with a residual / dropout: nothing on the path asks for them
        if (r32) {
            // The residual tile in fp32 is as large as the accumulators (CNT x 16 registers): it is fetched in two halves, each with its own
            // vmcnt(0) in front of its stores.  The second wait also drains the first half's stores (one counter for loads and stores, see the
            // top of this function) -- once per tile; the next tile's first DMA goes out behind that second wait, so the wait does not cover it.
            const LineBuf bF0 = line_buf(p.C, p.ldc, 4, n_first + fcol, p.N), bF1 = line_buf(p.C, p.ldc, 4, n_first + 32 + fcol, p.N);
            const LineBuf bR0 = line_buf(p.R, p.ldr, 4, n_first + fcol, p.N), bR1 = line_buf(p.R, p.ldr, 4, n_first + 32 + fcol, p.N);
            constexpr int H = (CNT + 1) / 2;
            i32x4 pre[H][G::NP][2];
            auto phase = [&](auto i0_c, auto i1_c, auto drop_c) {
                constexpr int I0 = decltype(i0_c)::value, I1 = decltype(i1_c)::value;
                constexpr bool DROP = decltype(drop_c)::value;
#pragma unroll
                for (int i = I0; i < I1; ++i) {
                    get_lines_issue(i, bR0, pre[i - I0][0][0], pre[i - I0][0][1]);
                    get_lines_issue(i, bR1, pre[i - I0][1][0], pre[i - I0][1][1]);
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);
                if constexpr (I1 == CNT) issue_next();
#pragma unroll
                for (int i = I0; i < I1; ++i) {
                    const int m = m_first + 16 * i + t;
#pragma unroll
                    for (int q = 0; q < G::NP; ++q) {
                        get_lines_finish(pre[i - I0][q][0], pre[i - I0][q][1]);          // -> this lane's columns 8g..8g+3 / 8g+4..8g+7 of row t
                        f32x4 x0 = acc[i][2 * q], x1 = acc[i][2 * q + 1];
                        if constexpr (DROP) {
                            x0 = dropout4(x0, p.seed, (unsigned long long)m * p.N + nc[q], p.p_drop, inv_keep);
                            x1 = dropout4(x1, p.seed, (unsigned long long)m * p.N + nc[q] + 4, p.p_drop, inv_keep);
                        }
                        x0 += __builtin_bit_cast(f32x4, pre[i - I0][q][0]);
                        x1 += __builtin_bit_cast(f32x4, pre[i - I0][q][1]);
                        put_lines_f32(i, q == 0 ? bF0 : bF1, x0, x1);
                    }
                }
            };
            using I0c = std::integral_constant<int, 0>; using IHc = std::integral_constant<int, H>; using INc = std::integral_constant<int, CNT>;
            if (p.p_drop > 0.f) { phase(I0c{}, IHc{}, std::true_type{}); if constexpr (H < CNT) phase(IHc{}, INc{}, std::true_type{}); }
            else { phase(I0c{}, IHc{}, std::false_type{}); if constexpr (H < CNT) phase(IHc{}, INc{}, std::false_type{}); }
            if constexpr (H == CNT) { /* one phase: issue_next() ran inside it */ }
        } else if (!general) {
            // The launch-uniform switches (dropout, residual, fp32 output) select one of eight straight-line bodies: left as run-time tests
            // inside the unrolled (fragment, column group) loops they cost ~10 scalar branches per store instruction, with the dropout hash
            // code to jump over each time.
            const T* R = (const T*)p.R;
            auto body = [&](auto drop_c, auto resid_c, auto f32_c, auto post_c) {
                constexpr bool DROP = decltype(drop_c)::value, RESID = decltype(resid_c)::value, F32 = decltype(f32_c)::value;
                constexpr bool POST = decltype(post_c)::value;            // DicGemmParams.bias2: a second bias row BEHIND the dropout (centred residual stream)
                i32x4 pre[RESID ? CNT : 1][RESID ? G::NP : 1];
                f32x4 post[POST ? G::NP : 1][2];
                if constexpr (POST) {
#pragma unroll
                    for (int q = 0; q < G::NP; ++q) {
                        post[q][0] = v0[q] ? *(const f32x4*)(p.bias2 + nc[q]) : f32x4{0.f, 0.f, 0.f, 0.f};
                        post[q][1] = v1[q] ? *(const f32x4*)(p.bias2 + nc[q] + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                const LineBuf bC = F32 ? LineBuf{} : line_buf(p.C, p.ldc, 2, n_first + lcol, p.N);
                const LineBuf bF0 = F32 ? line_buf(p.C, p.ldc, 4, n_first + fcol, p.N) : LineBuf{};
                const LineBuf bF1 = F32 ? line_buf(p.C, p.ldc, 4, n_first + 32 + fcol, p.N) : LineBuf{};
                // (full-height tiles with a residual: the set-up's registers do not fit next to 128 accumulators + 64 residual registers,
                // so there the DMA goes out first and the wait below covers it as well)
                constexpr bool EARLY = RESID && CNT * G::FN * 4 + CNT * G::NP * 4 >= 192;
                if constexpr (EARLY) issue_next();
                if constexpr (RESID) {
                    const LineBuf bR = line_buf(R, p.ldr, 2, n_first + lcol, p.N);
#pragma unroll
                    for (int i = 0; i < CNT; ++i) get_lines_issue(i, bR, pre[i][0], pre[i][1]);
                    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the residual tile is in registers
                }
                if constexpr (!EARLY) issue_next();
#pragma unroll
                for (int i = 0; i < CNT; ++i) {
                    const int m = m_first + 16 * i + t;
                    i32x4 P[G::NP];
                    if constexpr (RESID) get_lines_finish(pre[i][0], pre[i][1]);
#pragma unroll
                    for (int q = 0; q < G::NP; ++q) {
                        f32x4 x0 = acc[i][2 * q], x1 = acc[i][2 * q + 1];
                        if constexpr (DROP) {
                            x0 = dropout4(x0, p.seed, (unsigned long long)m * p.N + nc[q], p.p_drop, inv_keep);
                            x1 = dropout4(x1, p.seed, (unsigned long long)m * p.N + nc[q] + 4, p.p_drop, inv_keep);
                        }
                        if constexpr (POST) { x0 += post[q][0]; x1 += post[q][1]; }
                        if constexpr (RESID) { f32x4 r0, r1; unpack8(pre[i][q], r0, r1); x0 += r0; x1 += r1; }
                        if constexpr (F32) put_lines_f32(i, q == 0 ? bF0 : bF1, x0, x1);
                        else P[q] = pack8f(x0, x1);
                    }
                    if constexpr (!F32) put_lines(i, bC, P[0], P[1]);
                }
            };
            const bool drop = p.p_drop > 0.f, resid = PF && R != nullptr, f32o = p.out_f32 != 0;
            using Tt = std::true_type; using Ff = std::false_type;
            if (f32o) body(Ff{}, Ff{}, Tt{}, Ff{});
            else if constexpr (PF) {
                if (resid) { if (drop) { if (p.bias2) body(Tt{}, Tt{}, Ff{}, Tt{}); else body(Tt{}, Tt{}, Ff{}, Ff{}); } else body(Ff{}, Tt{}, Ff{}, Ff{}); }
                else { if (drop) body(Tt{}, Ff{}, Ff{}, Ff{}); else body(Ff{}, Ff{}, Ff{}, Ff{}); }
            } else {                                            // weight gradients: never a residual in this path
                if (drop) body(Tt{}, Ff{}, Ff{}, Ff{}); else body(Ff{}, Ff{}, Ff{}, Ff{});
            }
        } else {
            // rare combinations (accumulating into an fp32 C; a residual with N % 8 != 0): loads inside the loop.  Still fully unrolled:
            // a run-time index into the accumulator array would move it to scratch memory.
            issue_next();
#pragma unroll
            for (int i = 0; i < CNT; ++i) {
                const int m = m_first + 16 * i + t;
#pragma unroll
                for (int q = 0; q < G::NP; ++q) {
                    if (m >= p.M || !v0[q]) continue;
                    f32x4 x0 = acc[i][2 * q], x1 = acc[i][2 * q + 1];
                    if (p.p_drop > 0.f) {
                        x0 = dropout4(x0, p.seed, (unsigned long long)m * p.N + nc[q], p.p_drop, inv_keep);
                        x1 = dropout4(x1, p.seed, (unsigned long long)m * p.N + nc[q] + 4, p.p_drop, inv_keep);
                    }
                    if (p.R) {
                        const T* rp = (const T*)p.R + (size_t)m * p.ldr + nc[q];
                        x0 += Elem<T>::ld4(rp);
                        if (v1[q]) x1 += Elem<T>::ld4(rp + 4);
                    }
                    if (p.out_f32) {
                        float* c = (float*)p.C + (size_t)m * p.ldc + nc[q];
                        if (p.accumulate) { x0 += *(const f32x4*)c; if (v1[q]) x1 += *(const f32x4*)(c + 4); }
                        *(f32x4*)c = x0;
                        if (v1[q]) *(f32x4*)(c + 4) = x1;
                    } else {
                        T* c = (T*)p.C + (size_t)m * p.ldc + nc[q];
                        Elem<T>::st4(c, x0);
                        if (v1[q]) Elem<T>::st4(c + 4, x1);
                    }
                }
            }
        }
    } else if constexpr (EPI == DIC_EPI_BIAS_GELU_D) {               // g = gelu(u) -> C, gelu'(u) -> aux (what the backward multiplies by)
        issue_next();
        const bool keep_d = p.aux != nullptr;
        const LineBuf bD = line_buf(keep_d ? p.aux : p.C, keep_d ? p.ldaux : p.ldc, 2, n_first + lcol, p.N), bC = line_buf(p.C, p.ldc, 2, n_first + lcol, p.N);
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            i32x4 Dq[G::NP], A[G::NP];
#pragma unroll
            for (int q = 0; q < G::NP; ++q) {
                f32x4 x0 = acc[i][2 * q], x1 = acc[i][2 * q + 1], d0, d1;
                gelu_fast_with_grad4(x0, d0); gelu_fast_with_grad4(x1, d1);
                Dq[q] = pack8f(d0, d1);
                A[q] = pack8f(x0, x1);
            }
            if (keep_d) put_lines(i, bD, Dq[0], Dq[1]);
            put_lines(i, bC, A[0], A[1]);
        }
    } else if constexpr (EPI == DIC_EPI_MUL_AUX) {                   // dU = acc * aux  (aux = gelu'(u) left by BIAS_GELU_D)
        i32x4 pre[CNT][G::NP];
        const LineBuf bU = line_buf(p.aux, p.ldaux, 2, n_first + lcol, p.N), bC = line_buf(p.C, p.ldc, 2, n_first + lcol, p.N);
        constexpr bool EARLY = CNT * G::FN * 4 + CNT * G::NP * 4 >= 192;          // (as for the residual epilogue: registers of the set-up vs the side tile)
        if constexpr (EARLY) issue_next();
#pragma unroll
        for (int i = 0; i < CNT; ++i) get_lines_issue(i, bU, pre[i][0], pre[i][1]);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if constexpr (!EARLY) issue_next();
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            get_lines_finish(pre[i][0], pre[i][1]);
            i32x4 P[G::NP];
#pragma unroll
            for (int q = 0; q < G::NP; ++q) {
                f32x4 x0 = acc[i][2 * q], x1 = acc[i][2 * q + 1], u0, u1;
                unpack8(pre[i][q], u0, u1);
                x0 *= u0; x1 *= u1;
                P[q] = pack8f(x0, x1);
            }
            put_lines(i, bC, P[0], P[1]);
        }
    } else if constexpr (EPI == DIC_EPI_BIAS_GELU) {                 // N % 8 == 0 is required for this epilogue
        issue_next();
        // aux == NULL: a forward-only call (sampling / validation under no_grad) -- the pre-activation is not kept: half the epilogue's stores
        const bool keep_u = p.aux != nullptr;
        const LineBuf bU = line_buf(keep_u ? p.aux : p.C, keep_u ? p.ldaux : p.ldc, 2, n_first + lcol, p.N), bC = line_buf(p.C, p.ldc, 2, n_first + lcol, p.N);
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            i32x4 U[G::NP], A[G::NP];
#pragma unroll
            for (int q = 0; q < G::NP; ++q) {
                f32x4 x0 = acc[i][2 * q], x1 = acc[i][2 * q + 1];
                U[q] = pack8f(x0, x1);                                // pre-activation u (for GELU')
                gelu_fast4(x0); gelu_fast4(x1);
                A[q] = pack8f(x0, x1);
            }
            if (keep_u) put_lines(i, bU, U[0], U[1]);
            put_lines(i, bC, A[0], A[1]);
        }
    } else if constexpr (EPI == DIC_EPI_GELU_BWD) {                  // dU = acc * gelu'(U)
        i32x4 pre[CNT][G::NP];
        const LineBuf bU = line_buf(p.aux, p.ldaux, 2, n_first + lcol, p.N), bC = line_buf(p.C, p.ldc, 2, n_first + lcol, p.N);
#pragma unroll
        for (int i = 0; i < CNT; ++i) get_lines_issue(i, bU, pre[i][0], pre[i][1]);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        issue_next();
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            get_lines_finish(pre[i][0], pre[i][1]);
            i32x4 P[G::NP];
#pragma unroll
            for (int q = 0; q < G::NP; ++q) {
                f32x4 x0 = acc[i][2 * q], x1 = acc[i][2 * q + 1], u0, u1;
                unpack8(pre[i][q], u0, u1);
                gelu_grad_mul4(x0, u0); gelu_grad_mul4(x1, u1);
                P[q] = pack8f(x0, x1);
            }
            put_lines(i, bC, P[0], P[1]);
        }
    } else if constexpr (EPI == DIC_EPI_CE_DLOGITS) {                // (softmax - onehot) * row_scale; columns in [N, ldc) are written as zeros
        float r_lse[CNT], r_sc[CNT];
        long long r_tg[CNT];
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const int m = m_first + 16 * i + t;
            const bool ok = m < p.M;
            r_lse[i] = ok ? p.lse[m] : 0.f;
            r_tg[i] = ok ? p.tgt[m] : -1;
            r_sc[i] = ok ? row_scale(p, m) : 0.f;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        issue_next();
        const LineBuf bC = line_buf(p.C, p.ldc, 2, n_first + lcol, p.ldc);
        // (a leaner per-element sequence -- fma + v_exp, 32-bit target compare, no bounds tests off the last tile column -- changed nothing:
        // this epilogue is bound by its 1 GB of stores, not by VALU)
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const int m = m_first + 16 * i + t;
            const float lse = r_lse[i], sc = r_sc[i];
            const long long tg = r_tg[i];
            i32x4 P[G::NP];
#pragma unroll
            for (int q = 0; q < G::NP; ++q) {
                f32x4 x0 = acc[i][2 * q], x1 = acc[i][2 * q + 1];
                const int n = nc[q];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float q0 = (n + r < p.N) ? __expf(x0[r] - lse) : 0.f, q1 = (n + 4 + r < p.N) ? __expf(x1[r] - lse) : 0.f;
                    if ((long long)(n + r) == tg) q0 -= 1.0f;
                    if ((long long)(n + 4 + r) == tg) q1 -= 1.0f;
                    x0[r] = q0 * sc; x1[r] = q1 * sc;
                }
                P[q] = pack8f(x0, x1);
            }
            put_lines(i, bC, P[0], P[1]);          // only rows < M are written
        }
    } else if constexpr (EPI == DIC_EPI_CE_EXP) {
        // Training forward of the rounding head: E = exp(logit - c_row) as bf16 (c_row = p.lse[m], the caller's per-row reference point; columns
        // in [N, ldc) are written as zeros), the sum of the UNROUNDED values of this wave's 64 columns per row into partial[m][n_first / 64],
        // and the fp32 logit of the target column into tgt_logit[m].  dic_ce_exp_combine turns E into the (unnormalised) gradient operand,
        // so the backward needs no second pass over the vocabulary.
        // The store-bound part (exp + 1 GB of E) is the loop; everything else is kept out of it: the row sums are reduced over the four lanes
        // of a row with v_permlane16/32_swap (no LDS round trip, no lgkmcnt wait) and stored through a buffer descriptor (no exec-mask branches),
        // and the target logit is picked in a branch that a wave takes for 3 % of its fragments (a first version compared every element's
        // column with the target: SGPR-mask bookkeeping spilled, 950 us against 780 us for the CE_DLOGITS epilogue that writes the same bytes).
        static_assert(G::WCOLS == 64, "one partial sum per 64-column wave slab");
        float r_c[CNT];
        int r_tg[CNT];
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            // (unconditional loads from a clamped row: a load inside an `m < M` branch gets its own vmcnt(0) when the compiler sinks the
            //  first use into the branch -- eight serialised round trips per tile, measured +120 us on this launch)
            const int m = m_first + 16 * i + t, mc = m < p.M ? m : p.M - 1;
            r_c[i] = p.lse[mc];
            const long long tg = p.tgt ? p.tgt[mc] : -1;
            r_tg[i] = (m < p.M && tg >= 0 && tg < (long long)p.N) ? (int)tg : -0x40000000;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        issue_next();
        const LineBuf bC = line_buf(p.C, p.ldc, 2, n_first + lcol, p.ldc);
        const int np = ((p.N + G::BN - 1) / G::BN) * G::WN, slot = n_first / G::WCOLS;
        constexpr float L2E = 1.4426950408889634f;
        float sums[CNT];
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const float c2 = r_c[i] * L2E;
            float sm = 0.f;
            i32x4 P[G::NP];
#pragma unroll
            for (int q = 0; q < G::NP; ++q) {
                f32x4 x0 = acc[i][2 * q], x1 = acc[i][2 * q + 1];
                const int n = nc[q];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (exponent capped at 2^100: a logit more than 69 + shift nats above its row's reference point -- a per-token loss beyond ~109 with
                    //  the shift of 40 the engine uses -- saturates instead of overflowing the fp32 sums of this row and of E @ W)
                    const float e0 = (n + r < p.N) ? __builtin_amdgcn_exp2f(__builtin_fminf(__builtin_fmaf(x0[r], L2E, -c2), 100.f)) : 0.f;
                    const float e1 = (n + 4 + r < p.N) ? __builtin_amdgcn_exp2f(__builtin_fminf(__builtin_fmaf(x1[r], L2E, -c2), 100.f)) : 0.f;
                    sm += e0 + e1;
                    x0[r] = e0; x1[r] = e1;
                }
                P[q] = pack8f(x0, x1);
            }
            put_lines(i, bC, P[0], P[1]);          // only rows < M are written
            sums[i] = sm;
        }
        // row sums over the four lanes (g = 0..3) that share a row: lanes 16 and 32 apart
        const long long pbytes = (long long)(p.M - m_first) * np * 4;
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.partial + (size_t)m_first * np), 0,
                                                                             (int)(pbytes < 0 ? 0 : (pbytes > 0x7FFFFFFFll ? 0x7FFFFFFFll : pbytes)), 0x00020000);
        const unsigned poff = g == 0 ? ((unsigned)t * (unsigned)np + (unsigned)slot) * 4u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            // (inline asm: with the same value passed for both operands of __builtin_amdgcn_permlane32_swap this compiler adds result 0 to
            //  itself -- seen in the ISA, and in the sums.  The s_nop covers the VALU-write -> permlane-swap wait states the asm hides.)
            float a0 = sums[i], a1 = sums[i];
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));      // a0 = [lo, lo], a1 = [hi, hi] (lane halves)
            float b0 = a0 + a1, b1 = b0;
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(b0), "+v"(b1));      // rows 0<->1, 2<->3
            const float s4 = b0 + b1;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, s4), rsP, (int)(poff + (unsigned)(16 * i) * (unsigned)np * 4u), 0, 0);
        }
        // the target's logit: a lane owns columns nc[0]..+7 and nc[1]..+7 of row t
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const unsigned d0 = (unsigned)(r_tg[i] - nc[0]), d1 = (unsigned)(r_tg[i] - nc[1]);
            if (d0 < 8u || d1 < 8u) {
                const unsigned d = d0 < 8u ? d0 : d1;
                const int q2 = d0 < 8u ? 0 : 2;
                float tv = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a0 = acc[i][j][r], a1 = acc[i][2 + j][r];
                        tv = (d == (unsigned)(4 * j + r)) ? (q2 == 0 ? a0 : a1) : tv;
                    }
                p.tgt_logit[m_first + 16 * i + t] = tv;
            }
        }
    }
}

// CE_PARTIAL for the bf16 geometries: every wave owns its rows x 64 columns of the tile and emits one (max, sum exp, first argmax)
// record per row; record slot = bn * WN + wn, so a row has nbn * WN records for ce_combine.
template <class C, int CNT, class IssueNext>
__device__ __forceinline__ void epilogue_ce_partial(f32x4 (&acc)[CNT][Geo<C>::FN], const DicGemmParams& p, int m_first, int n_first,
                                                    int wn, int lane, int bn, int nbn, IssueNext&& issue_next) {
    using G = Geo<C>;
    static_assert(G::WCOLS == 64 && G::FN == 4, "one 64-column record per wave");
    const int g = lane >> 4, t = lane & 15;
    const int np = G::WN * nbn;
    // all target ids first: a load inside the row loop would wait on vmcnt, which also counts the stores of the rows before it
    long long tgs[CNT];
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int m = m_first + i * 16 + t;
        tgs[i] = (m < p.M && p.tgt) ? p.tgt[m] : -1;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // the target ids are in; only now the next tile's DMA goes out (see epilogue_direct)
    issue_next();
    constexpr float L2E = 1.4426950408889634f;
    const bool full = n_first + G::WCOLS <= p.N;            // wave-uniform: only the vocabulary's last tile column is ragged
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int m = m_first + i * 16 + t;
        const long long tg = tgs[i];
        float mx = -INFINITY, sm = 0.f;
        int ix = 0x7fffffff;
        if (full) {
            // short path (no bounds tests): max by a max3 chain, first argmax by a descending equality scan over compile-time column
            // offsets, target logit by a 32-bit compare against the lane-relative target column, exp as fma + v_exp
            const long long tgl = tg - (long long)(n_first + 8 * g);
            const int tl = (tgl >= 0 && tgl < G::WCOLS) ? (int)tgl : -1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[i][j][r]);
            int il = 0;
            float tv = 0.f;
#pragma unroll
            for (int j = 3; j >= 0; --j)
#pragma unroll
                for (int r = 3; r >= 0; --r) {
                    const int c = 32 * (j >> 1) + 4 * (j & 1) + r;                    // column relative to n_first + 8g (= frag_col(j, g, r) - 8g)
                    il = (acc[i][j][r] == mx) ? c : il;
                    tv = (tl == c) ? acc[i][j][r] : tv;
                }
            ix = n_first + 8 * g + il;
            if (((tl >= 0 && tl < 8) || (tl >= 32 && tl < 40)) && m < p.M) p.tgt_logit[m] = tv;
            const float mx2 = mx * L2E;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) sm += __builtin_amdgcn_exp2f(__builtin_fmaf(acc[i][j][r], L2E, -mx2));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)            // (j, r) ascending = column ascending within a lane: the first maximum keeps the lowest index
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n_first + frag_col(j, g, r);
                    const float x = acc[i][j][r];
                    if (n < p.N) {
                        if (x > mx) { mx = x; ix = n; }
                        if ((long long)n == tg) p.tgt_logit[m] = x;
                    }
                }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n_first + frag_col(j, g, r);
                    if (n < p.N) sm += __expf(acc[i][j][r] - mx);
                }
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {       // merge the 4 lanes (g = 0..3) that share row m
            float mx2 = __shfl_xor(mx, o, 64), sm2 = __shfl_xor(sm, o, 64);
            int ix2 = __shfl_xor(ix, o, 64);
            float M = fmaxf(mx, mx2);
            float s1 = (mx == -INFINITY) ? 0.f : sm * __expf(mx - M);
            float s2 = (mx2 == -INFINITY) ? 0.f : sm2 * __expf(mx2 - M);
            ix = (mx > mx2) ? ix : (mx2 > mx) ? ix2 : min(ix, ix2);
            mx = M; sm = s1 + s2;
        }
        if (g == 0 && m < p.M) *(float4*)(p.partial + ((size_t)m * np + bn * G::WN + wn) * 4) = make_float4(mx, sm, __int_as_float(ix), 0.f);
    }
}

// =====================================================================================================
// bf16 kernel: operand tiles go HBM/L2 -> LDS directly with `buffer_load_dwordx4 ... lds` (LDS-DMA): no staging VGPRs and
// no ds_write pass (the register-staged v1 spent ~830 of every ~1340 LDS cycles per K-step on ds_write_b128); the next
// tile's DMA is in flight while the MFMAs of the current tile run.  The DMA destination is lane-linear (wave base +
// lane*16), so the bank-conflict-free LDS image is produced by permuting the per-lane SOURCE address and applying the
// same XOR on the fragment reads (scripts/experiments/lds_bank_check.py replays every fragment read against the bank model of
// MI355X_MICROARCH.md):
//   KC tile [rows][128 B]:       16-byte chunk c of row r lives at chunk c ^ key(r); fragment = ONE ds_read_b128 (8 consecutive k),
//                                conflict-free.  A fragments read 16 consecutive rows (key_a); B fragments read the rows of the
//                                output mapping above, 8 (t>>2) + (t&3) (+ 4q + 32p), which needs a key of its own (key_b).
//   KM tile [64 k][rows x 2 B]:  chunk c of row k lives at c ^ km_key(k); fragment = two ds_read_b64_tr_b16 on rows
//                                8g+{0..3} and 8g+4+{0..3}  => lane group g = lane>>4 holds k = 8g..8g+7 in BOTH layouts.  An A
//                                fragment reads 16 consecutive columns (conflict-free); a B fragment reads the 4-column groups
//                                8c + 4q of the output mapping: one half of four 16-byte chunks, 2-way (the minimum for that shape).
// Everything lane-dependent (DMA source offsets, fragment addresses) is computed once per workgroup; the K loop is DMA
// issues, LDS reads, MFMAs, a few integer adds and one barrier per 64-deep step.
__device__ __forceinline__ int key_a(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int key_b(int row) { return ((((row >> 3) & 3) << 1) | ((row >> 1) & 1)) & 7; }
__device__ __forceinline__ int km_key(int k) { return 2 * ((k & 3) | (((k >> 3) & 1) << 2)); }

// ---- grouped weight gradients (dic_wgrad_group) ------------------------------------------------------------------------------------------
// The weight gradients of one encoder layer -- dW = dY^T X for the qkv, out-proj and the two FFN Linears: 27 + 9 + 36 + 36 tiles of 256 x 256,
// each with the whole token dimension (272 K-steps at 17 408 tokens) as its contraction -- are ONE launch: every tile of every problem is cut
// into the same number S of K-slices, and the S x 108 (slice, tile) units are walked by a persistent grid slice-slowest, so the workgroups
// resident together read the same token range of dY / X (L2 reuse across the tiles that share a panel) and S is chosen for the WHOLE group
// (S = 7: 756 units = 2.95 rounds of 256) instead of per problem (9 / 28 / 7 / 7, four launches each rounded up to whole rounds, four folds).
// (A stream-K cut -- equal consecutive K-step ranges of the tile-major sequence, ~1/3 of the partial tiles -- was tried first: 2.5 % slower on
// the step, because neighbouring workgroups then sit at unrelated token positions and every operand panel is streamed from HBM once per tile.)
constexpr int WG_MAX_PROBLEMS = 8;
struct WgradGroupDev {
    int n, nk, split, per;                     // problems, K-steps per tile, K-slices per tile, K-steps per slice
    int tiles, pad_;
    int tile0[WG_MAX_PROBLEMS + 1];            // first global tile of each problem
    int nbn[WG_MAX_PROBLEMS];
    const void* A[WG_MAX_PROBLEMS];            // dY [T][lda] (k-major)
    const void* B[WG_MAX_PROBLEMS];            // X  [T][ldb] (k-major)
    float* Cout[WG_MAX_PROBLEMS];
    float* cs[WG_MAX_PROBLEMS];                // bias gradient (column sums of dY) or NULL
    int M[WG_MAX_PROBLEMS], N[WG_MAX_PROBLEMS], lda[WG_MAX_PROBLEMS], ldb[WG_MAX_PROBLEMS];
    float* ws;                                 // [split * tiles] slabs of BM*BN + BM floats, slab = slice * tiles + tile
};

template <class C, bool AKM, bool BKM, int EPI, int CNT, bool GROUP>
__device__ __forceinline__ void gemm_bf16_body(DicGemmParams& p, const WgradGroupDev* grp, int row_base = 0, int row_cnt = -1) {
    // row_base / row_cnt: this call covers rows [row_base, row_base + row_cnt) of the problem (row_cnt < 0: all of it) -- gemm_bf16_kernel2 runs
    // the body twice with two tile heights
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Geo<C>;
    using T = bf16_t;
    constexpr int S = 2, BK = 64;
    constexpr int ROWB_A = AKM ? G::BM * 2 : 128, ROWB_B = BKM ? G::BN * 2 : 128;     // LDS row pitch of each operand tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / G::WN, wn = wave % G::WN;
    const int g = lane >> 4, t = lane & 15;
    // tile height: CNT 16-row fragments for each of the two wave rows (a per-launch choice, see pick_tile_rows; k-major A always takes BM).
    // CNT is a template parameter: with a run-time count the K loop needs either a branch between fragment reads and MFMAs or a switch
    // over unrolled bodies, and the latter made the register allocator spill hundreds of VGPRs.
    static_assert(CNT >= 1 && CNT <= G::FM && (!AKM || CNT == G::FM), "fragments per wave");
    constexpr int tile_rows = 32 * CNT, cnt = CNT;
    const int row0_w = wm * CNT * 16;

    // ---- per-lane fragment addresses inside an operand tile for the first 32-deep half-step (same for every tile this
    // workgroup processes); the second half-step is `^ 64` (KC: chunk index bit 2) or `+ 32 rows` (KM: same swizzle key)
    // k-contiguous tiles: the swizzle key is the same for every fragment of an operand (A fragments are 16 rows apart, B fragments
    // 4 / 32 rows), so fragment i is a compile-time byte offset from fragment 0 -- one address register per operand, the rest are
    // instruction immediates.  k-major tiles XOR the key into the column chunk, which differs per fragment: one register each.
    int ofA[AKM ? CNT : 1], ofB[BKM ? G::FN : 1];
    if constexpr (!AKM) { const int row = row0_w + t; ofA[0] = row * 128 + ((g ^ key_a(row)) << 4); }
    else {
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            const int base = row0_w + i * 16, rho = 8 * g + (t >> 2), c = (base >> 3) + ((t & 3) >> 1);
            ofA[i] = rho * ROWB_A + ((c ^ km_key(rho)) << 4) + (t & 1) * 8;
        }
    }
    if constexpr (!BKM) { const int row = wn * G::WCOLS + 8 * (t >> 2) + (t & 3); ofB[0] = row * 128 + ((g ^ key_b(row)) << 4); }
    else {
#pragma unroll
        for (int j = 0; j < G::FN; ++j) {           // fragment pair's first column; fragment parity picks the 4-column half
            const int base = wn * G::WCOLS + 32 * (j >> 1), rho = 8 * g + (t >> 2), c = (base >> 3) + (t & 3);
            ofB[j] = rho * ROWB_B + ((c ^ km_key(rho)) << 4) + (j & 1) * 8;
        }
    }
    auto offA = [&](int i, int kk) { return AKM ? ofA[AKM ? i : 0] + kk * 32 * ROWB_A : (ofA[0] ^ (kk * 64)) + i * (16 * 128); };
    auto offB = [&](int j, int kk) { return BKM ? ofB[BKM ? j : 0] + kk * 32 * ROWB_B : (ofB[0] ^ (kk * 64)) + (32 * (j >> 1) + 4 * (j & 1)) * 128; };
    // ---- per-lane DMA source offsets of the 1 KiB pieces this wave stages per operand advance by a uniform step.  (Kept in
    // VGPRs rather than the scalar offset operand: the descriptor's bounds check covers only the vector offset, and it is
    // that check which zero-fills ragged M/N/K.)
    unsigned stepA = AKM ? (unsigned)BK * (unsigned)p.lda * 2u : BK * 2u;          // (re-derived per problem by the grouped launch)
    unsigned stepB = BKM ? (unsigned)BK * (unsigned)p.ldb * 2u : BK * 2u;

    // The LDS-DMA is issued through inline asm on purpose.  With the builtin, the compiler's waitcnt pass cannot tell which LDS bytes a
    // pending `buffer_load ... lds` will write, so it puts `s_waitcnt vmcnt(0)` in front of EVERY later ds_read: the K loop then
    // waits for the DMA of the next stage before it reads the current one, and the two stages run back to back instead of
    // overlapping (measured: 1.58 us per K-step = 1.21 us of LDS reads + MFMA plus most of the 0.90 us DMA).  Hidden from the
    // pass, the DMA is ordered by the explicit `s_waitcnt vmcnt(0)` + `s_barrier` that publishes a stage (dma_barrier below).
    i32x4 rsA, rsB;                                   // buffer resource words: base, base_hi (stride 0), num_records, flags
    // SPLIT WEIGHTS (p.B2, forward Linears only): C = A (B + B2)^T with B2 = bf16(W - bf16(W)), the low-order half of an fp32 weight whose
    // bf16 rounding is B.  The K loop simply runs twice over the tile's K range -- pass 1 against B, pass 2 against B2, same A -- into the
    // same accumulators: at the pass boundary the DMA cursors rewind and B's descriptor is swapped, nothing else changes.  Why it exists:
    // the bf16 rounding of the WEIGHTS is one fixed perturbation shared by every sample, so its first-order effect on a batch-mean loss does
    // not average out (2.6e-4 relative at the bench shape), while activation roundings are independent per element and do (< 4e-5;
    // profiles/r04_weight_rounding_probe.txt).  With hi + lo the weights carry 16 mantissa bits and the bf16 engine's losses sit within
    // north_star's 1e-4 of the fp32 engine's at twice the forward-GEMM K loop.
    constexpr bool TWO_PASS = !GROUP && !AKM && !BKM && (EPI == DIC_EPI_AFFINE || EPI == DIC_EPI_BIAS_GELU || EPI == DIC_EPI_BIAS_GELU_D);
    i32x4 rsB2 = {0, 0, 0, 0};
    auto make_rsrc = [](const void* base, long long bytes) {
        const unsigned long long b = (unsigned long long)base;
        i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        r[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
        r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
        r[3] = 0x00020000;
        return r;
    };
    auto dma16 = [](unsigned voff, const i32x4& rsrc, unsigned lds_addr) {   // 64 lanes x 16 B -> LDS [lds_addr, lds_addr + 1 KiB)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
    };
    // every wave's DMA pieces of the stage being published have landed, and every wave is done reading the stage being recycled
    auto dma_barrier = []() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // A pieces (8 rows of a k-contiguous tile) are dealt round-robin over the waves so that a short tile still spreads its DMA issue;
    // pieces beyond the tile height are not issued at all
    constexpr int n_pieces_a = AKM ? G::BM / 8 : tile_rows >> 3;
    // per-lane source offsets of the wave's pieces, swizzle term included, advanced by one K-step per issue.  (Recomputing them from one
    // base per operand saved six registers and cost ~40 VALU instructions at the head of every K-step, in front of the first MFMA of
    // both waves of a SIMD: 25 % slower on the K = 768 GEMMs.)
    unsigned voA[G::PA], voB[G::PB];
    auto setup = [&](const TileId& tl) {          // descriptors anchored at the tile origin: OOB rows/k read as zero
        const int m0 = row_base + tl.bm * tile_rows, n0 = tl.bn * G::BN;
        const T* Ab = (const T*)p.A + (AKM ? (size_t)m0 : (size_t)m0 * p.lda);
        const T* Bb = (const T*)p.B + (BKM ? (size_t)n0 : (size_t)n0 * p.ldb);
        long long a_bytes = AKM ? ((long long)(p.K - 1) * p.lda + (p.M - m0)) * S : ((long long)(p.M - m0 - 1) * p.lda + p.K) * S;
        long long b_bytes = BKM ? ((long long)(p.K - 1) * p.ldb + (p.N - n0)) * S : ((long long)(p.N - n0 - 1) * p.ldb + p.K) * S;
        if (a_bytes > 0xFFFFFFF0ll) a_bytes = 0xFFFFFFF0ll;
        if (b_bytes > 0xFFFFFFF0ll) b_bytes = 0xFFFFFFF0ll;
        rsA = make_rsrc(Ab, a_bytes);
        rsB = make_rsrc(Bb, b_bytes);
        if constexpr (TWO_PASS) { if (p.B2) rsB2 = make_rsrc((const T*)p.B2 + (size_t)n0 * p.ldb, b_bytes); }
        const unsigned ka = (unsigned)tl.kt0 * stepA, kb = (unsigned)tl.kt0 * stepB;
#pragma unroll
        for (int j = 0; j < G::PA; ++j) {
            if (!AKM) { const int q = j * G::NW + wave, row = 8 * q + (lane >> 3); voA[j] = ka + (unsigned)row * (unsigned)p.lda * 2u + (((lane & 7) ^ key_a(row)) << 4); }
            else { constexpr int CPRW = G::BM / 8; const int q = wave * G::PA + j, row = q * (64 / CPRW) + lane / CPRW; voA[j] = ka + (unsigned)row * (unsigned)p.lda * 2u + (((lane % CPRW) ^ km_key(row)) << 4); }
        }
#pragma unroll
        for (int j = 0; j < G::PB; ++j) {
            const int q = wave * G::PB + j;
            if (!BKM) { const int row = 8 * q + (lane >> 3); voB[j] = kb + (unsigned)row * (unsigned)p.ldb * 2u + (((lane & 7) ^ key_b(row)) << 4); }
            else { constexpr int CPRW = G::BN / 8; const int row = q * (64 / CPRW) + lane / CPRW; voB[j] = kb + (unsigned)row * (unsigned)p.ldb * 2u + (((lane % CPRW) ^ km_key(row)) << 4); }
        }
    };
    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_PTR(char))smem;
    auto issue = [&](int stage) {
        const unsigned dstA = __builtin_amdgcn_readfirstlane(lds_base + stage * G::STAGE + (AKM ? wave * (G::PA * 1024) : wave * 1024));
        const unsigned dstB = __builtin_amdgcn_readfirstlane(lds_base + stage * G::STAGE + G::A_BYTES + wave * (G::PB * 1024));
#pragma unroll
        for (int j = 0; j < G::PA; ++j) {
            if (AKM || j * G::NW + wave < n_pieces_a) dma16(voA[j], rsA, dstA + j * (AKM ? 1024 : G::NW * 1024));
            voA[j] += stepA;
        }
#pragma unroll
        for (int j = 0; j < G::PB; ++j) {
            dma16(voB[j], rsB, dstB + j * 1024);
            voB[j] += stepB;
        }
    };
    auto frag = [&](const char* tile, int off, bool km, int rowb) -> bf16x8 {
        if (!km) { i32x4 v = *(const i32x4*)(tile + off); return __builtin_bit_cast(bf16x8, v); }
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + off));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + off + 4 * rowb));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    f32x4 acc[CNT][G::FN];
    // fused bias gradient (weight-gradient GEMMs only): db[m] = sum_k A[k][m] is a column sum of the A tile that is already
    // in LDS; the workgroups of the first tile column (bn == 0) add it up on the side (4 x ds_read_b128 + 32 adds per thread
    // per K-step) instead of a separate pass that re-reads dY from HBM.
    bool do_cs = false;
    f32x4 cs0{0.f, 0.f, 0.f, 0.f}, cs1{0.f, 0.f, 0.f, 0.f};
    constexpr int CS_CPR = G::BM / 8, CS_GROUPS = G::NTH / CS_CPR;        // 16-byte chunks per A row; thread groups over k
    auto compute = [&](int stage, bool more_k) {
        const int sb = stage * G::STAGE;
        const char* la = smem + sb;
        const char* lb = la + G::A_BYTES;
        if constexpr (AKM && BKM && EPI == DIC_EPI_AFFINE) {
            if (do_cs) {
#pragma unroll
                for (int r = 0; r < 64 / CS_GROUPS; ++r) {
                    const int rho = tid / CS_CPR + CS_GROUPS * r;
                    f32x4 a, b;
                    unpack8(*(const i32x4*)(la + rho * ROWB_A + (((tid % CS_CPR) ^ km_key(rho)) << 4)), a, b);
                    cs0 += a; cs1 += b;
                }
            }
        }
        // Fragment reads run DIC_GEMM_PF A fragments ahead of the MFMAs that consume them, with counted waits.  Left to the compiler every A
        // fragment is `ds_read -> s_waitcnt lgkmcnt(0) -> 4 MFMAs`: a wave alone keeps the matrix pipe below 50 % busy and the K-step ends
        // with the younger wave of each SIMD finishing on its own (s_memtime stamps: the older wave waits ~25 % of every K-step at the barrier).
        // Issue order: B(kk=0) x FN, A0, A1, | A2 ... A(CNT-1), B(kk=1) x FN, A(CNT) ... ; LDS returns in order, so before the MFMAs of
        // A_s everything but the reads issued after A_s may be outstanding.  The reads are inline asm so that neither the order nor the
        // counts are the compiler's to change (it tracks none of them); every wait names the registers it validates.
        constexpr int RA = AKM ? 2 : 1, RB = BKM ? 2 : 1, NS = 2 * CNT;
        constexpr int PFD_MAX = (15 - G::FN * RB) / RA, PFD = DIC_GEMM_PF < PFD_MAX ? DIC_GEMM_PF : PFD_MAX, RING = PFD + 1;      // lgkmcnt counts to 15
        const unsigned sbA = lds_base + sb, sbB = sbA + G::A_BYTES;
        unsigned aA[2] = {0, 0}, aB[2] = {0, 0};
        if constexpr (!AKM) { aA[0] = sbA + (unsigned)ofA[0]; aA[1] = sbA + ((unsigned)ofA[0] ^ 64u); }
        if constexpr (!BKM) { aB[0] = sbB + (unsigned)ofB[0]; aB[1] = sbB + ((unsigned)ofB[0] ^ 64u); }
        auto read_frag = [&](bf16x8& dst, bool km, unsigned addr, auto imm_c, auto rowb4_c) {
            constexpr int imm = decltype(imm_c)::value, hi = decltype(rowb4_c)::value;
            if (!km) {
                i32x4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(v) : "v"(addr), "i"(imm));
                dst = __builtin_bit_cast(bf16x8, v);
            } else {
                s16x4 lo, hi4;
                asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c3\n\tds_read_b64_tr_b16 %1, %2 offset:%c4" : "=&v"(lo), "=v"(hi4) : "v"(addr), "i"(imm), "i"(imm + hi));
                dst = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        };
        bf16x8 fb[2][G::FN], fa[RING];
        auto issue_b = [&](auto kk_c) {
            constexpr int kk = decltype(kk_c)::value;
#pragma unroll
            for (int j = 0; j < G::FN; ++j) {
                if constexpr (!BKM) {
                    auto rd = [&](auto j_c) { constexpr int jj = decltype(j_c)::value;
                        read_frag(fb[kk][jj], false, aB[kk], std::integral_constant<int, (32 * (jj >> 1) + 4 * (jj & 1)) * 128>{}, std::integral_constant<int, 0>{}); };
                    if (j == 0) rd(std::integral_constant<int, 0>{});
                    if (j == 1) rd(std::integral_constant<int, 1>{});
                    if (j == 2) rd(std::integral_constant<int, 2>{});
                    if (j == 3) rd(std::integral_constant<int, 3>{});
                } else {
                    read_frag(fb[kk][j], true, sbB + (unsigned)ofB[j], std::integral_constant<int, kk * 32 * ROWB_B>{}, std::integral_constant<int, 4 * ROWB_B>{});
                }
            }
        };
        auto issue_a = [&](auto s_c) {          // A fragment number sidx of the K-step; the B set of the second half-step goes out right before its first A fragment
            constexpr int sidx = decltype(s_c)::value, kk = sidx / CNT, i = sidx % CNT;
            if constexpr (sidx == CNT) issue_b(std::integral_constant<int, 1>{});
            if constexpr (!AKM) read_frag(fa[sidx % RING], false, aA[kk], std::integral_constant<int, i * 2048>{}, std::integral_constant<int, 0>{});
            else read_frag(fa[sidx % RING], true, sbA + (unsigned)ofA[i], std::integral_constant<int, kk * 32 * ROWB_A>{}, std::integral_constant<int, 4 * ROWB_A>{});
        };
        auto step = [&](auto s_c) {
            constexpr int sidx = decltype(s_c)::value, kk = sidx / CNT, i = sidx % CNT;
            if constexpr (sidx + PFD < NS) issue_a(std::integral_constant<int, sidx + PFD>{});
            // reads issued after A_s: A_{s+1} .. A_{s+PFD} and, when it lies among them, the B set of the second half-step
            constexpr int last = (sidx + PFD < NS - 1) ? sidx + PFD : NS - 1;
            constexpr int after = (last - sidx) * RA + ((sidx < CNT && last >= CNT) ? G::FN * RB : 0);
            static_assert(after <= 15, "lgkmcnt is a 4-bit counter");
            if constexpr (i == 0) {
                asm volatile("s_waitcnt lgkmcnt(%c5)" : "+v"(fa[sidx % RING]), "+v"(fb[kk][0]), "+v"(fb[kk][1]), "+v"(fb[kk][2]), "+v"(fb[kk][3]) : "i"(after));
            } else {
                asm volatile("s_waitcnt lgkmcnt(%c1)" : "+v"(fa[sidx % RING]) : "i"(after));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < G::FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[sidx % RING], acc[i][j], 0, 0, 0);
            // Where in the K-step the next stage's DMA is issued (DIC_GEMM_ISSUE_AT).  With operands resident in L2 / the infinity cache, issuing
            // behind the first few MFMA groups is 2-5 % faster (at the head of the K-step the ~8 x 60-180 issue cycles per wave sit in front of
            // the first MFMA of both waves of a SIMD); with operands coming from HBM -- the training step: every operand was just written by
            // another kernel or is streamed once -- the K-step is bound by the DMA's arrival and issuing FIRST wins by 5-10 % (weight
            // gradients: +10 % slower when issued after fragment 3).  Default: first.  Also measured: staggering the two waves of a SIMD (one
            // half issuing in the first half-step, the other in the second) 5-8 % slower; static s_setprio for the younger half: noise.
            if constexpr (DIC_GEMM_ISSUE_AT >= 0 && sidx == (DIC_GEMM_ISSUE_AT < NS ? DIC_GEMM_ISSUE_AT : NS - 1)) { if (more_k) issue(stage ^ 1); }
        };
        static_assert(G::FN == 4, "issue_b / the first wait of a half-step name four B fragments");
        if constexpr (DIC_GEMM_ISSUE_AT < 0) { if (more_k) issue(stage ^ 1); }
        issue_b(std::integral_constant<int, 0>{});
        [&]<int... Is>(std::integer_sequence<int, Is...>) { (issue_a(std::integral_constant<int, Is>{}), ...); }(std::make_integer_sequence<int, (PFD < NS ? PFD : NS)>{});
        [&]<int... Is>(std::integer_sequence<int, Is...>) { (step(std::integral_constant<int, Is>{}), ...); }(std::make_integer_sequence<int, NS>{});
    };

    // ---- persistent loop over (tile, K-slice) units: the grid is capped at the number of co-resident workgroups, so
    // addressing set-up is paid once per workgroup and the tail of the launch is balanced by unit order, not dispatch order.
    const int total = GROUP ? 0 : total_units(p, tile_rows, G::BN, row_cnt);
    if (!GROUP && (int)blockIdx.x >= total) return;          // (only the second call of a two-height launch has fewer units than workgroups)
    int unit = blockIdx.x;
    // grouped launch: unit u of split * tiles, slice-slowest after the XCD-aware remap (consecutive logical units share an XCD's L2)
    int slab = 0;
    auto group_unit = [&](int u) {             // sets p (operands, shapes) and returns the unit's tile / K range
        const int nwg = grp->split * grp->tiles;
        const int q_ = nwg >> 3, r_ = nwg & 7, xcd = u & 7, slot = u >> 3;
        const int lu = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + slot;
        const int kz = lu / grp->tiles, tg = lu - kz * grp->tiles;
        int pi = 0;
        while (pi + 1 < grp->n && tg >= grp->tile0[pi + 1]) ++pi;
        p.A = grp->A[pi]; p.B = grp->B[pi]; p.M = grp->M[pi]; p.N = grp->N[pi]; p.lda = grp->lda[pi]; p.ldb = grp->ldb[pi];
        p.colsum_out = grp->cs[pi]; p.C = grp->Cout[pi];
        stepA = (unsigned)BK * (unsigned)p.lda * 2u; stepB = (unsigned)BK * (unsigned)p.ldb * 2u;
        const int lt = tg - grp->tile0[pi], nbn_ = grp->nbn[pi];
        TileId t_;
        t_.bm = lt / nbn_; t_.bn = lt - t_.bm * nbn_; t_.nbn = nbn_; t_.kz = kz;
        t_.kt0 = kz * grp->per < grp->nk ? kz * grp->per : grp->nk;
        t_.kt1 = t_.kt0 + grp->per < grp->nk ? t_.kt0 + grp->per : grp->nk;
        slab = lu;
        return t_;
    };
#ifdef DIC_GEMM_TRACE      // measurement build only (scripts/experiments/gemm_trace.py): s_memtime stamps of wave 0 at the phase boundaries of every tile
    unsigned long long* trace = (unsigned long long*)p.tgt_logit + (size_t)blockIdx.x * 64;
    int trace_n = 0;
#define DIC_STAMP() do { if (trace && tid == 0 && trace_n < 61) trace[trace_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DIC_STAMP() do { } while (0)
#endif
    DIC_STAMP();
#ifdef DIC_GEMM_TRACE
    if (trace && tid == 0) trace[62] = __builtin_amdgcn_s_memrealtime();     // 100 MHz wall clock next to the shader clock: their ratio is the GPU clock under this kernel's load
#endif
    // Forward Linears (k-contiguous A and B): the accumulators START from the bias instead of zero.  The bias row of the tile's 64 columns
    // per wave is loaded where nothing waits for it -- behind the previous tile's stores (before the first tile: behind the first DMA) --
    // and the epilogue has no load and no wait left in front of its first store (s_memtime: ~2 k cycles per tile for that wait).
    // (Only without the cross-tile prefetch below: with it nothing waits behind the stores any more, so the bias is added in the epilogue.)
#ifndef DIC_GEMM_XT
#define DIC_GEMM_XT 0          // measured in round 3 (profiles/r03_gemm_xt_ab.txt): 3-7 % SLOWER on every shape, K loop included -- off
#endif
    constexpr bool XT = DIC_GEMM_XT != 0;
    // (the rounding-head epilogues take a per-column bias too: the row-common part of the logits of a mean-centred head input, dic_head_center)
    constexpr bool BIAS_INIT = !XT && !GROUP && !AKM && !BKM && (EPI == DIC_EPI_AFFINE || EPI == DIC_EPI_BIAS_GELU || EPI == DIC_EPI_BIAS_GELU_D ||
                                                                 EPI == DIC_EPI_CE_PARTIAL || EPI == DIC_EPI_CE_EXP || EPI == DIC_EPI_CE_DLOGITS);
    f32x4 binit[BIAS_INIT ? G::FN : 1];
    auto load_bias = [&](const TileId& t_) {
        if constexpr (BIAS_INIT) {
#pragma unroll
            for (int j = 0; j < G::FN; ++j) binit[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                const int nf = t_.bn * G::BN + wn * G::WCOLS + 8 * g;
#pragma unroll
                for (int q = 0; q < G::NP; ++q) {
                    if (nf + 32 * q < p.N) binit[2 * q] = *(const f32x4*)(p.bias + nf + 32 * q);
                    if (nf + 32 * q + 4 < p.N) binit[2 * q + 1] = *(const f32x4*)(p.bias + nf + 32 * q + 4);
                }
            }
        }
    };
    TileId tl;
    if constexpr (GROUP) {
        tl = group_unit(unit);
    } else {
        tl = tile_of_unit(p, BK, unit, tile_rows, G::BN, row_cnt);
    }
    setup(tl);
    if (tl.kt0 < tl.kt1) issue(0);
    load_bias(tl);
    // CROSS-TILE PREFETCH (XT).  The tile's last K-step has no successor inside the tile, so its DMA slot carries the NEXT unit's first K-step
    // (into the stage that step is not reading), and the end-of-step barrier publishes it BEFORE this tile's output is stored.  The epilogue
    // then issues the next unit's second K-step into the stage the last step has just released, still in front of its first store.  The next
    // tile therefore starts without any wait: its first vmcnt(0) -- which on gfx950 also waits for every store issued before it -- comes at
    // the end of its first K-step, a full K-step (~2.9 k cycles) after the last store went out, instead of at the loop top (s_memtime trace
    // of round 2: 5-8 k cycles per tile between the epilogue and the first MFMA).  The stage parity runs on across tiles.
    int cur = 0;
    bool k0_ready = false;         // this tile's first K-step was published by the previous tile's last barrier
    bool k1_issued = false;        // this tile's second K-step was issued by the previous tile's epilogue
    for (;;) {
        // the NEXT unit's tile coordinates (a handful of integer divisions) are worked out here, where the wave waits for the first DMA anyway
        TileId tl_next = tl;
        bool more_next = false;
        if constexpr (!GROUP) {
            more_next = unit + (int)gridDim.x < total;
            if (more_next) tl_next = tile_of_unit(p, BK, unit + (int)gridDim.x, tile_rows, G::BN, row_cnt);
        } else {
            more_next = unit + (int)gridDim.x < grp->split * grp->tiles;
        }
#ifdef DIC_GEMM_TRACE
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        DIC_STAMP();
        asm volatile("s_barrier" ::: "memory");
#else
        if (!k0_ready) dma_barrier();        // the tile's first K-step has landed (without XT: and the previous tile's output stores have drained)
#endif
        DIC_STAMP();
#pragma unroll
        for (int i = 0; i < CNT; ++i)
#pragma unroll
            for (int j = 0; j < G::FN; ++j) acc[i][j] = BIAS_INIT ? binit[BIAS_INIT ? j : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (AKM && BKM && EPI == DIC_EPI_AFFINE) {
            do_cs = p.colsum_out != nullptr && tl.bn == 0;
            cs0 = f32x4{0.f, 0.f, 0.f, 0.f}; cs1 = cs0;
        }
        const int nk = tl.kt1;
        // (split weights: a second pass over the same K range; K-step kt >= nk reads A's step kt - (nk - kt0) again, against B2)
        // (b2_col0: only the tiles at output columns >= b2_col0 run the second pass -- the value projection's third of a fused q|k|v weight)
        const int kend = (TWO_PASS && p.B2 && tl.bn * G::BN >= p.b2_col0) ? nk + (nk - tl.kt0) : nk;
        const TileId done = tl;
        const int slab_done = slab;
        DicGemmParams pe = p;                        // (grouped launches: group_unit() below re-points p at the next problem)
        bool next_set = false, next_k0 = false;      // next unit's addressing is set up / its first K-step is in flight or landed
        // (A software-pipelined variant -- barrier before the last MFMA group, first fragments of the next stage prefetched across it,
        // DMA re-armed 1.75 K-steps ahead -- was measured in round 1: +2-5 % on isolated GEMMs, -2 % on the training step; not kept.)
        for (int kt = tl.kt0; kt < kend; ++kt) {     // ONE loop body (a hand-unrolled pair with an early exit made the
            bool do_issue = kt + 1 < kend;             //  register allocator keep two copies of the accumulator tile); issues the next K-step's DMA
            if constexpr (TWO_PASS) {
                if (kt + 1 == nk && kend > nk) {       // the next DMA is pass 2's first K-step: rewind both cursors, B -> B2
                    const unsigned backA = (unsigned)(nk - tl.kt0) * stepA, backB = (unsigned)(nk - tl.kt0) * stepB;
#pragma unroll
                    for (int j = 0; j < G::PA; ++j) voA[j] -= backA;
#pragma unroll
                    for (int j = 0; j < G::PB; ++j) voB[j] -= backB;
                    rsB = rsB2;
                }
            }
            if (do_issue) { if (k1_issued && kt == tl.kt0) do_issue = false; }
            else if (XT && more_next) {
                if constexpr (GROUP) tl_next = group_unit(unit + (int)gridDim.x);
                setup(tl_next);
                next_set = true;
                do_issue = next_k0 = tl_next.kt0 < tl_next.kt1;
            }
            compute(cur, do_issue);
#ifdef DIC_GEMM_TRACE
            if (p.partial && (kt - tl.kt0) < 16 && unit == (int)blockIdx.x && (wave == 0 || wave == G::NW - 1)) {
                unsigned long long* kst = (unsigned long long*)p.partial + (((size_t)blockIdx.x * 2 + (wave != 0)) * 16 + (kt - tl.kt0)) * 4;
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                const unsigned long long t1 = __builtin_amdgcn_s_memtime();
                asm volatile("s_barrier" ::: "memory");
                const unsigned long long t2 = __builtin_amdgcn_s_memtime();
                if (lane == 0) { kst[0] = t0; kst[1] = t1; kst[2] = t2; }
            } else
#endif
            dma_barrier();
            cur ^= 1;
        }
        DIC_STAMP();
        // `cur` is now the stage that holds (or will hold) the next unit's first K-step; the other stage is free.
        bool more = more_next;
        // grouped launch with ONE K-slice per tile (round 5: the weight gradients of two layers, 216 tiles = one round of 256 workgroups): the
        // tile is complete, so it goes straight to its dW (and its column sums to db) -- no slab, no fold launch
        const bool grp_direct = GROUP && grp->split == 1;
        if constexpr (GROUP) {
            pe.out_f32 = 1; pe.accumulate = 0; pe.bias = nullptr; pe.R = nullptr; pe.p_drop = 0.f;
            if (grp_direct) {
                pe.ldc = pe.N;
            } else {
                // the unit's partial tile goes to its own slab, in tile-local coordinates (the fold kernel adds a tile's slices up)
                pe.C = grp->ws + (size_t)slab_done * (G::BM * G::BN + G::BM);
                pe.ldc = G::BN; pe.M = G::BM; pe.N = G::BN;
            }
        } else {
            if (p.split_k > 1) redirect_to_slab(pe, done.kz);
        }
        unit += gridDim.x;
        // The next tile's remaining start-up goes out from inside the epilogue: behind its last load wait, in front of its first store.
        auto issue_next = [&]() {
            if (!more) return;
            if (!next_set) {                         // no K-step of this tile could carry the prefetch (a unit without K-steps), or XT is off
                if constexpr (GROUP) tl_next = group_unit(unit);
                setup(tl_next);
                if (tl_next.kt0 < tl_next.kt1) issue(cur);
            } else if (next_k0 && tl_next.kt0 + 1 < tl_next.kt1) {
                issue(cur ^ 1);
            }
        };
        if constexpr (AKM && BKM && EPI == DIC_EPI_AFFINE) {
            if (do_cs) {       // fold the thread groups through LDS (the free stage), fixed order
                float* red = (float*)(smem + (cur ^ 1) * G::STAGE);
                *(f32x4*)(red + (tid / CS_CPR) * G::BM + (tid % CS_CPR) * 8) = cs0;
                *(f32x4*)(red + (tid / CS_CPR) * G::BM + (tid % CS_CPR) * 8 + 4) = cs1;
                barrier_lds_only();
                if (tid < G::BM) {
                    float v = 0.f;
#pragma unroll
                    for (int gq = 0; gq < CS_GROUPS; ++gq) v += red[gq * G::BM + tid];
                    const int m = done.bm * G::BM + tid;
                    if (GROUP && !grp_direct) ((float*)pe.C)[G::BM * G::BN + tid] = v;
                    else if (GROUP) { if (m < pe.M) pe.colsum_out[m] = v; }
                    else if (m < pe.M) {
                        if (pe.split_k > 1) ((float*)pe.C)[(size_t)pe.M * pe.ldc + m] = v;
                        else pe.colsum_out[m] = pe.accumulate ? pe.colsum_out[m] + v : v;
                    }
                }
                barrier_lds_only();
            }
        }
        const int m_first = ((GROUP && !grp_direct) ? 0 : row_base + done.bm * tile_rows) + row0_w, n_first = ((GROUP && !grp_direct) ? 0 : done.bn * G::BN) + wn * G::WCOLS;
        if constexpr (EPI == DIC_EPI_CE_PARTIAL) {
            epilogue_ce_partial<C, CNT>(acc, pe, m_first, n_first, wn, lane, done.bn, done.nbn, issue_next);
        } else {
            epilogue_direct<C, EPI, !AKM, CNT, BIAS_INIT>(acc, pe, m_first, n_first, lane, issue_next, [&]() { DIC_STAMP(); });
        }
        k0_ready = more && next_set && next_k0;
        k1_issued = k0_ready && tl_next.kt0 + 1 < tl_next.kt1;
        if (more && next_set && !next_k0) k0_ready = true;          // a unit without K-steps: nothing to wait for
        tl = tl_next;
        if constexpr (BIAS_INIT) { if (more) load_bias(tl); }       // behind this tile's stores; the loop-top wait covers it
        DIC_STAMP();
        if (!more) break;
    }
#ifdef DIC_GEMM_TRACE
    if (trace && tid == 0) { trace[63] = __builtin_amdgcn_s_memrealtime(); trace[61] = __builtin_amdgcn_s_memtime(); }
#endif
}

// CLOCK PROBE (measurement build only, -DDIC_CLOCK_PROBE; scripts/power_ab.py): workgroup 0 of every GEMM launch stamps the shader-clock counter
// (s_memtime) and the 100 MHz wall clock (s_memrealtime) at its start and at its end into the next record of a caller-provided buffer
// (dic_clock_probe_set) -- their ratio is the shader clock this launch actually ran at, measured inside the launch and per launch, which
// rocm-smi's 1 Hz samples cannot give.  Record: {memtime0, realtime0, memtime1, realtime1, M, N, K, tag} (tag: 1000 a_km + 100 b_km + epi; 5000+ = asm).
#ifdef DIC_CLOCK_PROBE
__device__ unsigned long long* g_clk_buf = nullptr;
__device__ unsigned g_clk_cap = 0, g_clk_n = 0;
struct ClkProbe {
    unsigned long long* rec = nullptr;
    __device__ __forceinline__ void begin(int M, int N, int K, int tag) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && g_clk_buf) {
            const unsigned slot = atomicAdd(&g_clk_n, 1u);
            if (slot < g_clk_cap) {
                rec = g_clk_buf + (size_t)slot * 8;
                rec[4] = (unsigned long long)M; rec[5] = (unsigned long long)N; rec[6] = (unsigned long long)K; rec[7] = (unsigned long long)tag;
                rec[0] = __builtin_amdgcn_s_memtime(); rec[1] = __builtin_amdgcn_s_memrealtime();
            }
        }
    }
    __device__ __forceinline__ void end() {
        if (rec) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); rec[2] = __builtin_amdgcn_s_memtime(); rec[3] = __builtin_amdgcn_s_memrealtime(); }
    }
};
extern "C" int dic_clock_probe_set(void* buf, int capacity) {
    unsigned long long* b = (unsigned long long*)buf;
    unsigned cap = (unsigned)capacity, zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_clk_buf), &b, sizeof(b)) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(g_clk_cap), &cap, sizeof(cap)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(g_clk_n), &zero, sizeof(zero)) != hipSuccess) { dic_set_error("dic_clock_probe_set: hipMemcpyToSymbol failed"); return 1008; }
    return 0;
}
extern "C" int dic_clock_probe_count(void) {
    unsigned n = 0;
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_clk_n), sizeof(n));
    return (int)n;
}
#else
struct ClkProbe { __device__ __forceinline__ void begin(int, int, int, int) {} __device__ __forceinline__ void end() {} };
#endif

template <class C, bool AKM, bool BKM, int EPI, int CNT>
__global__ __launch_bounds__(Geo<C>::NTH, 2) void gemm_bf16_kernel(DicGemmParams p) {
    if (p.step_ctr) p.seed += (uint64_t)(p.step_ctr[0] - p.step_ctr0) * DIC_STRIDE_DROP;
    ClkProbe cp; cp.begin(p.M, p.N, p.K, 1000 * AKM + 100 * BKM + EPI);
    gemm_bf16_body<C, AKM, BKM, EPI, CNT, false>(p, nullptr);
    cp.end();
}
// TWO TILE HEIGHTS IN ONE LAUNCH.  A persistent grid runs its units in rounds of `slots` workgroups; with one tile height the last round is
// usually partial (17 408 x 2304: 702 units of 224 rows = 2.74 rounds; 34 816 x 768: 1.83).  Here the first `row_split` rows -- whole rounds of
// CA-fragment tiles -- are followed by ONE round of shorter CB-fragment tiles over the remaining rows, so that round costs (CB + fixed) / (CA +
// fixed) of a full one instead of leaving CUs idle.  Each workgroup simply runs the body twice; tiles are independent, so nothing but the
// reuse of its own LDS orders the two calls.
template <class C, bool BKM, int EPI, int CA, int CB>
__global__ __launch_bounds__(Geo<C>::NTH, 2) void gemm_bf16_kernel2(DicGemmParams p, int row_split) {
    if (p.step_ctr) p.seed += (uint64_t)(p.step_ctr[0] - p.step_ctr0) * DIC_STRIDE_DROP;
    ClkProbe cp; cp.begin(p.M, p.N, p.K, 100 * BKM + EPI);
    gemm_bf16_body<C, false, BKM, EPI, CA, false>(p, nullptr, 0, row_split);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    gemm_bf16_body<C, false, BKM, EPI, CB, false>(p, nullptr, row_split, p.M - row_split);
    cp.end();
}
__global__ __launch_bounds__(Geo<T256>::NTH, 2) void wgrad_group_kernel(DicGemmParams p, WgradGroupDev grp) {
    ClkProbe cp; cp.begin(grp.tiles, grp.split, p.K, 9000);
    gemm_bf16_body<T256, true, true, DIC_EPI_AFFINE, Geo<T256>::FM, true>(p, &grp);
    cp.end();
}

// (The two K-loop alternatives measured in round 3 -- ping-pong loop, C++ four-wave 256 x 256 kernel: equal or slower, DESIGN.md section 7.0 -- are no
// longer part of the library: scripts/experiments/attic/ keeps their text.  The shipped library carries ONE 8-wave K loop + the generated asm kernel.)
// Fold of a grouped launch: tile t = sum of its K-slices' slabs in slice order (deterministic).  Block = (tile, 16-row chunk); 256 threads x
// (4 rows x 4 columns).  The bias gradient rides in each slab's tail.
__global__ __launch_bounds__(256) void wgrad_group_fold_kernel(WgradGroupDev grp) {
    using G = Geo<T256>;
    const int tg = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    int pi = 0;
    while (pi + 1 < grp.n && tg >= grp.tile0[pi + 1]) ++pi;
    const int lt = tg - grp.tile0[pi], nbn_ = grp.nbn[pi], bm = lt / nbn_, bn = lt - bm * nbn_;
    const size_t slab_f = (size_t)G::BM * G::BN + G::BM;
    const int r0 = chunk * 16 + (tid >> 6) * 4, col = (tid & 63) * 4;
    f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 csum{0.f, 0.f, 0.f, 0.f};
    const bool do_cs = grp.cs[pi] != nullptr && bn == 0 && chunk == 0 && tid < 64;
    for (int kz = 0; kz < grp.split; ++kz) {
        const float* sl = grp.ws + ((size_t)kz * grp.tiles + tg) * slab_f;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += *(const f32x4*)(sl + (size_t)(r0 + r) * G::BN + col);
        if (do_cs) csum += *(const f32x4*)(sl + (size_t)G::BM * G::BN + tid * 4);
    }
    float* Cp = grp.Cout[pi];
    const int M = grp.M[pi], N = grp.N[pi];
    const int n = bn * G::BN + col;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = bm * G::BM + r0 + r;
        if (m < M && n < N) *(f32x4*)(Cp + (size_t)m * N + n) = acc[r];
    }
    if (do_cs) { const int m = bm * G::BM + tid * 4; if (m < M) *(f32x4*)(grp.cs[pi] + m) = csum; }
}

// fold split-K slabs: out[i] (+)= sum_s ws[s][i]   (fixed order => deterministic); the optional tail of each slab holds the
// fused bias-gradient partial and goes to cs_out
__global__ void reduce_slabs_kernel(const float* ws, int nslab, long long n4, long long stride4, float* out, int accumulate,
                                    float* cs_out, long long n4_cs) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4 + n4_cs; i += (long long)gridDim.x * blockDim.x) {
        // slabs are read eight at a time (independent loads in flight: one memory round trip per eight slabs, not one each) and
        // added in slab order, so the result does not depend on the batching
        f32x4 a = ((const f32x4*)ws)[i];
        int s = 1;
        for (; s + 8 <= nslab; s += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ((const f32x4*)ws)[(size_t)(s + u) * stride4 + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; s < nslab; ++s) a += ((const f32x4*)ws)[(size_t)s * stride4 + i];
        f32x4* dst = i < n4 ? (f32x4*)out + i : (f32x4*)cs_out + (i - n4);
        if (accumulate) a += *dst;
        *dst = a;
    }
}

// ---- per-launch timing for bench.py's roofline leg (dic_prof_begin / dic_prof_end): while a record is armed for the calling thread, the
// (at most two) kernels of a dic_gemm / dic_wgrad_group call are launched through hipExtLaunchKernelGGL with a start and a stop event each,
// which stamp the kernel's own begin and end -- the same interval rocprofv3 reports.  (Events recorded around the launch added ~4 us of
// dispatch latency per call: 0.56 ms per step over 140 launches.)
struct ProfRec { hipEvent_t a, b, a2, b2; double flops, bytes; int used; };
thread_local ProfRec* tl_prof = nullptr;
template <typename... Args, typename F = void (*)(Args...)>
void launch_timed(F kernel, dim3 grid, dim3 block, unsigned lds, hipStream_t st, Args... args) {
    ProfRec* r = tl_prof;
    if (r && r->used < 2) {
        hipExtLaunchKernelGGL(kernel, grid, block, lds, st, r->used == 0 ? r->a : r->a2, r->used == 0 ? r->b : r->b2, 0u, args...);
        ++r->used;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
    }
}

// Process-global measurement switches (dic_set_option; the library reads NO environment variable -- the one record of every switch is
// diffusion-image-captioning_amd/options.py, which pushes its values through dic_set_option when the library is loaded).
// gemm_v1 = 1 runs bf16 on the register-staged v1 kernel (kept for within-run A/B measurements); default = LDS-DMA kernel.
int g_gemm_v1 = 0, g_persist = 1, g_rows = 1;
bool bf16_on_v1() { return g_gemm_v1 == 1; }

int device_cus() {
    static int cus = -1;
    if (cus < 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return cus;
}
bool persist_enabled() { return g_persist == 1; }
bool rows_enabled() { return g_rows == 1; }          // 0: always full-height tiles (A/B measurements)

#include "gemm_w4a.h"         // gemm_w4a_kernel: the hand-scheduled four-wave kernel (round 4; DIC_GEMM_W4A)
#include "gemm_w4n.h"         // gemm_w4n_kernel: its narrow-tile form with the epilogue under the next tile's K loop (round 6; option gemm_w4n)

// Tile height for a k-contiguous A (the token dimension of the forward / input-gradient GEMMs).  With full-height tiles the tile count
// rarely fills whole rounds of resident workgroups (17 408 tokens x 768 columns = 204 tiles of 256x256 on 256 CUs: one round at 80 %;
// x 2304: 612 = 2.39 rounds run as 3), and a persistent grid cannot rebalance inside a round.  Tiles may be any multiple of 32 rows
// (one 16-row fragment for each of the two wave rows) from BM/2 to BM, so the launch takes the height that minimises
// rounds x (K-loop time of one tile + per-tile fixed cost): 224 rows at 17 408 tokens (234 / 702 / 936 tiles = 1 / 3 / 4 rounds at 91 %).
int pick_tile_rows(int M, int nbn_split, int slots, int BM, int K) {
    const int CMAX = BM / 32, CMIN = CMAX / 2;          // fragments per wave
    if (!rows_enabled()) return BM;
    if (M <= BM) { int c = (M + 31) / 32; return 32 * (c < CMIN ? CMIN : c); }
    double fixed = 2.0 * 768.0 / (K > 0 ? K : 768);      // prologue + epilogue + one B tile, in units of one fragment pair's K loop (K = 768: ~1/4 tile)
    fixed = fixed < 0.25 ? 0.25 : (fixed > 3.0 ? 3.0 : fixed);
    double best = 1e30;
    int best_c = CMAX;
    for (int R = 1; R <= 4096; ++R) {
        const long long nbm = (long long)R * slots / nbn_split;
        if (nbm < 1) continue;
        int c = (int)((((long long)M + nbm - 1) / nbm + 31) / 32);
        if (c > CMAX) continue;
        const bool last = c <= CMIN;
        if (last) c = CMIN;
        const long long tiles = ((long long)M + 32 * c - 1) / (32 * c) * nbn_split;
        const double cost = (double)((tiles + slots - 1) / slots) * (c + fixed);
        if (cost < best - 1e-9) { best = cost; best_c = c; }
        if (last) break;
    }
    return 32 * best_c;
}

template <class C, bool AKM, bool BKM, int E, int CNT>
void launch_bf16_cnt(const DicGemmParams& q, hipStream_t st, int grid) {
    using G = Geo<C>;
    static bool attr_set[2][64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[0][dev]) {           // per device: one process may drive several GPUs
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<C, AKM, BKM, E, CNT>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        attr_set[0][dev] = true;
    }
    launch_timed(gemm_bf16_kernel<C, AKM, BKM, E, CNT>, dim3(grid), dim3(G::NTH), (unsigned)G::LDS, st, q);
}

// Two tile heights (gemm_bf16_kernel2): R whole rounds of cA-fragment tiles + one round of cB-fragment tiles over the rest, if that beats the best
// single height by 3 % in the cost model of pick_tile_rows.  OFF by default, ON inside diffusion.sample() (dic_gemm_set_two_heights; env
// DIC_GEMM_TWO_HEIGHTS=1 turns it on everywhere).  Measured (profiles/r03_gemm_two_heights_ab.txt): FFN-1 at 17 408 rows 108.0 -> 105.2 us, QKV
// unchanged, a sampling pass 7.56 -> 7.47 ms -- but the TRAINING step gets 0.8 % slower (13.78 -> 13.91 ms): there the CUs a partial round leaves
// idle are not idle, they run the weight-gradient stream's kernels.
struct TwoHeights { int cA = 0, cB = 0, row_split = 0; };
int g_two_heights = 0;
bool two_heights_enabled() { return g_two_heights == 1; }
TwoHeights plan_two_heights(int M, int nbn, int slots, int K, int rows_single, bool ignore_switch = false) {
    TwoHeights best;
    if ((!ignore_switch && !two_heights_enabled()) || !rows_enabled() || nbn > slots) return best;
    double fixed = 2.0 * 768.0 / (K > 0 ? K : 768);
    fixed = fixed < 0.25 ? 0.25 : (fixed > 3.0 ? 3.0 : fixed);
    const int c1 = rows_single / 32;
    const long long units1 = ((long long)M + rows_single - 1) / rows_single * nbn;
    if (units1 <= slots) return best;                                   // a single round: nothing to balance
    double best_cost = 0.97 * (double)((units1 + slots - 1) / slots) * (c1 + fixed);
    const int per_round = slots / nbn;                                  // row tiles one round can hold
    for (int cA = 7; cA <= 8; ++cA)
        for (int R = 1; R <= 64; ++R) {
            const long long nA = (long long)R * slots / nbn;
            const long long rowsA = nA * 32 * cA;
            if (nA < 1) continue;
            if (rowsA >= M) break;
            const long long Mr = M - rowsA;
            int cB = (int)((Mr + 32ll * per_round - 1) / (32ll * per_round));
            if (cB > 7) continue;                                       // the rest does not fit one round of shorter tiles
            if (cB < 4) cB = 4;
            const double cost = R * (cA + fixed) + (cB + fixed);
            if (cost < best_cost - 1e-9) { best_cost = cost; best.cA = cA; best.cB = cB; best.row_split = (int)rowsA; }
        }
    return best;
}
template <class C, bool BKM, int E, int CA, int CB>
void launch_bf16_two(const DicGemmParams& q, hipStream_t st, int grid, int row_split) {
    using G = Geo<C>;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel2<C, BKM, E, CA, CB>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        attr_set[dev] = true;
    }
    launch_timed(gemm_bf16_kernel2<C, BKM, E, CA, CB>, dim3(grid), dim3(G::NTH), (unsigned)G::LDS, st, q, row_split);
}

template <class C, bool AKM, bool BKM, int E>
void launch_bf16(const DicGemmParams& q, hipStream_t st) {
    using G = Geo<C>;
    int cus = device_cus();
    if (q.cu_cap > 0 && q.cu_cap < cus) cus = q.cu_cap;             // spatial share of the chip (two streams interleaving two half-batches)
    const int resident = cus * (160 * 1024 / G::LDS);                // co-resident workgroups (LDS-limited): 2 per CU at 64 KB, 1 at 128 KB
    const int nbn_split = ((q.N + G::BN - 1) / G::BN) * (q.split_k > 1 ? q.split_k : 1);
    const int rows = AKM ? G::BM : pick_tile_rows(q.M, nbn_split, resident, G::BM, q.K);
    const int units = nbn_split * ((q.M + rows - 1) / rows);
    const int grid = ((persist_enabled() || q.cu_cap > 0) && units > resident) ? resident : units;
    if constexpr (std::is_same_v<C, T256> && !AKM && !BKM && (E == DIC_EPI_AFFINE || E == DIC_EPI_BIAS_GELU || E == DIC_EPI_BIAS_GELU_D)) {
        if (q.split_k <= 1 && persist_enabled()) {
            const TwoHeights th = plan_two_heights(q.M, nbn_split, resident, q.K, rows);
            if (th.cA) {
#define DIC_TWO(A_, B_) if (th.cA == A_ && th.cB == B_) { launch_bf16_two<C, BKM, E, A_, B_>(q, st, resident, th.row_split); return; }
                DIC_TWO(7, 4) DIC_TWO(7, 5) DIC_TWO(7, 6) DIC_TWO(7, 7) DIC_TWO(8, 4) DIC_TWO(8, 5) DIC_TWO(8, 6) DIC_TWO(8, 7)
#undef DIC_TWO
            }
        }
    }
    if constexpr (AKM) {
        launch_bf16_cnt<C, AKM, BKM, E, G::FM>(q, st, grid);
    } else if constexpr (G::FM == 8) {
        switch (rows / 32) {
            case 8: launch_bf16_cnt<C, AKM, BKM, E, 8>(q, st, grid); break;
            case 7: launch_bf16_cnt<C, AKM, BKM, E, 7>(q, st, grid); break;
            case 6: launch_bf16_cnt<C, AKM, BKM, E, 6>(q, st, grid); break;
            case 5: launch_bf16_cnt<C, AKM, BKM, E, 5>(q, st, grid); break;
            default: launch_bf16_cnt<C, AKM, BKM, E, 4>(q, st, grid); break;
        }
    } else {
        switch (rows / 32) {
            case 4: launch_bf16_cnt<C, AKM, BKM, E, 4>(q, st, grid); break;
            case 3: launch_bf16_cnt<C, AKM, BKM, E, 3>(q, st, grid); break;
            default: launch_bf16_cnt<C, AKM, BKM, E, 2>(q, st, grid); break;
        }
    }
}

template <typename T, bool AKM, bool BKM, int E>
void launch_one(hipStream_t st, const DicGemmParams& q) {
    if constexpr (sizeof(T) == 2) {
        if (!bf16_on_v1()) {
            if (q.tile == 256) { launch_bf16<T256, AKM, BKM, E>(q, st); return; }
            launch_bf16<T128, AKM, BKM, E>(q, st);
            return;
        }
    }
    constexpr size_t lds = 4 * TILE_BYTES;   // 72 KB: two stages x (A tile + B tile) -> 2 workgroups per CU
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<T, AKM, BKM, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev] = true;
    }
    const int nbn = (q.N + BN - 1) / BN, nbm = (q.M + BM - 1) / BM;
    launch_timed(gemm_kernel<T, AKM, BKM, E>, dim3(nbm * nbn * (q.split_k > 1 ? q.split_k : 1)), dim3(NT), (unsigned)lds, st, q);
}

template <typename T, bool AKM, bool BKM>
int launch_epi(const DicGemmParams& p, int epi, hipStream_t st) {
    const int split = p.split_k > 1 ? p.split_k : 1;
    // built combinations: every epilogue for (k-contiguous, k-contiguous) = nn.Linear forward and the rounding head; the affine and GELU
    // epilogues for a k-major B (input gradients); the affine one for (k-major, k-major) = weight gradients.  Nothing on the path
    // asks for the others.
    if (epi < DIC_EPI_AFFINE || epi > DIC_EPI_MUL_AUX) { dic_set_error("dic_gemm: unknown epilogue"); return 1002; }
    if ((AKM && epi != DIC_EPI_AFFINE) || (BKM && (epi == DIC_EPI_CE_PARTIAL || epi == DIC_EPI_CE_DLOGITS || epi == DIC_EPI_CE_EXP || epi == DIC_EPI_BIAS_GELU_D))) {
        dic_set_error("dic_gemm: this epilogue is not built for this operand layout (weight gradients: AFFINE only; k-major B: no rounding-head epilogues)");
        return 1005;
    }
    switch (epi) {
        case DIC_EPI_AFFINE: launch_one<T, AKM, BKM, DIC_EPI_AFFINE>(st, p); break;
        case DIC_EPI_BIAS_GELU: if constexpr (!AKM) launch_one<T, AKM, BKM, DIC_EPI_BIAS_GELU>(st, p); break;
        case DIC_EPI_GELU_BWD: if constexpr (!AKM) launch_one<T, AKM, BKM, DIC_EPI_GELU_BWD>(st, p); break;
        case DIC_EPI_CE_PARTIAL: if constexpr (!AKM && !BKM) launch_one<T, AKM, BKM, DIC_EPI_CE_PARTIAL>(st, p); break;
        case DIC_EPI_CE_DLOGITS: if constexpr (!AKM && !BKM) launch_one<T, AKM, BKM, DIC_EPI_CE_DLOGITS>(st, p); break;
        case DIC_EPI_CE_EXP:                       // bf16 LDS-DMA kernels only (dic_gemm_impl checks)
            if constexpr (!AKM && !BKM && sizeof(T) == 2) {
                if (p.tile == 256) launch_bf16<T256, false, false, DIC_EPI_CE_EXP>(p, st); else launch_bf16<T128, false, false, DIC_EPI_CE_EXP>(p, st);
            }
            break;
        case DIC_EPI_BIAS_GELU_D:                  // bf16 LDS-DMA kernels only
            if constexpr (!AKM && !BKM && sizeof(T) == 2) {
                if (p.tile == 256) launch_bf16<T256, false, false, DIC_EPI_BIAS_GELU_D>(p, st); else launch_bf16<T128, false, false, DIC_EPI_BIAS_GELU_D>(p, st);
            }
            break;
        case DIC_EPI_MUL_AUX:                      // bf16 LDS-DMA kernels only
            if constexpr (!AKM && sizeof(T) == 2) {
                if (p.tile == 256) launch_bf16<T256, false, BKM, DIC_EPI_MUL_AUX>(p, st); else launch_bf16<T128, false, BKM, DIC_EPI_MUL_AUX>(p, st);
            }
            break;
    }
    if (split > 1) {
        const long long n4 = (long long)p.M * p.ldc / 4, n4cs = p.colsum_out ? p.M / 4 : 0;
        int g = (int)((n4 + n4cs + 255) / 256);
        if (g > 2048) g = 2048;
        launch_timed(reduce_slabs_kernel, dim3(g), dim3(256), 0u, st, (const float*)p.split_ws, split, n4, n4 + n4cs, (float*)p.C, p.accumulate,
                     p.colsum_out, n4cs);
    }
    DIC_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int launch_layout(const DicGemmParams& p, int a_km, int b_km, int epi, hipStream_t st) {
    if (!a_km && !b_km) return launch_epi<T, false, false>(p, epi, st);
    if (!a_km && b_km) return launch_epi<T, false, true>(p, epi, st);
    if (a_km && b_km) return launch_epi<T, true, true>(p, epi, st);
    dic_set_error("dic_gemm: (A k-major, B k-contiguous) is not used by the path and not built");
    return 1003;
}

}  // namespace

// claims the next timing record of the measurement hooks below (false when profiling is off); thread-safe
static ProfRec* prof_slot(double flops, double bytes = 0.0);

// ---- grouped weight gradients: host side ---------------------------------------------------------------------------------------------------
namespace {
struct WgradPlan { WgradGroupDev dev; int grid; size_t ws_bytes; };
// Collects the problems' tiles into one sequence and picks the number of K-slices for the whole group.
int plan_wgrad_group(const DicWgradItem* items, int n, int T, int cu_cap, WgradPlan& pl) {
    using G = Geo<T256>;
    DIC_REQUIRE(n >= 1 && n <= WG_MAX_PROBLEMS && T > 0, "dic_wgrad_group: 1..8 problems");
    WgradGroupDev& d = pl.dev;
    d.n = n; d.nk = (T + 63) / 64; d.pad_ = 0;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        const DicWgradItem& it = items[i];
        DIC_REQUIRE(it.M > 0 && it.N > 0 && it.M % 256 == 0 && it.N % 8 == 0 && it.ldy % 8 == 0 && it.ldx % 8 == 0, "dic_wgrad_group: M must be a multiple of 256, N / ldy / ldx of 8");
        DIC_REQUIRE(((uintptr_t)it.dY % 16) == 0 && ((uintptr_t)it.X % 16) == 0 && ((uintptr_t)it.dW % 16) == 0, "dic_wgrad_group: operands must be 16-byte aligned");
        DIC_REQUIRE((long long)T * it.ldy * 2 < 0x7FFFFFFFll && (long long)T * it.ldx * 2 < 0x7FFFFFFFll, "dic_wgrad_group: operand too large for 32-bit buffer offsets");
        d.tile0[i] = tiles;
        d.nbn[i] = (it.N + G::BN - 1) / G::BN;
        tiles += (it.M / G::BM) * d.nbn[i];
        d.A[i] = it.dY; d.B[i] = it.X; d.Cout[i] = it.dW; d.cs[i] = it.db; d.M[i] = it.M; d.N[i] = it.N; d.lda[i] = it.ldy; d.ldb[i] = it.ldx;
    }
    for (int i = n; i <= WG_MAX_PROBLEMS; ++i) d.tile0[i] = tiles;
    for (int i = n; i < WG_MAX_PROBLEMS; ++i) { d.nbn[i] = 1; d.A[i] = d.B[i] = nullptr; d.Cout[i] = d.cs[i] = nullptr; d.M[i] = d.N[i] = d.lda[i] = d.ldb[i] = 0; }
    d.tiles = tiles;
    int cus = device_cus();
    if (cu_cap > 0 && cu_cap < cus) cus = cu_cap;
    // K-slices per tile: minimise rounds x (slice length + per-unit fixed cost ~ 8 K-steps: first DMA + a 256 KB partial tile written at ~16 B/clk)
    double best = 1e30, cost1 = 1e30;
    int best_s = 1;
    for (int S = 1; S <= 64 && (S == 1 || d.nk / S >= 8); ++S) {
        const long long units = (long long)tiles * S;
        // (S = 1 writes no slab and needs no fold: its fixed cost is the epilogue alone, and it wins ties)
        const double cost = (double)((units + cus - 1) / cus) * ((d.nk + S - 1) / S + (S == 1 ? 2.0 : 8.0));
        if (S == 1) cost1 = cost;
        if (cost < best - 1e-9) { best = cost; best_s = S; }
    }
    // what the model does not see: S > 1 writes S x tiles partial tiles and reads them back in a second launch (0.8 GB for 216 tiles x 7 slices
    // at 48 756 tokens) -- one slice per tile is taken whenever the model puts it within 12 % of the best cut
    if (best_s > 1 && cost1 <= 1.12 * best) best_s = 1;
    d.split = best_s;
    d.per = (d.nk + best_s - 1) / best_s;
    const long long units = (long long)tiles * best_s;
    pl.grid = (int)(units < cus ? units : cus);
    pl.ws_bytes = (size_t)units * ((size_t)G::BM * G::BN + G::BM) * sizeof(float);
    return 0;
}
}  // namespace

extern "C" size_t dic_wgrad_group_ws_bytes(const DicWgradItem* items, int n, int T, int cu_cap) {
    WgradPlan pl;
    if (plan_wgrad_group(items, n, T, cu_cap, pl) != 0) return 0;
    return pl.ws_bytes;
}
extern "C" int dic_wgrad_group(const DicWgradItem* items, int n, int T, void* ws, size_t ws_bytes, int cu_cap, void* stream) {
    using G = Geo<T256>;
    WgradPlan pl;
    int rc = plan_wgrad_group(items, n, T, cu_cap, pl);
    if (rc) return rc;
    DIC_REQUIRE(ws != nullptr && ws_bytes >= pl.ws_bytes && ((uintptr_t)ws % 16) == 0, "dic_wgrad_group: workspace too small (dic_wgrad_group_ws_bytes)");
    pl.dev.ws = (float*)ws;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)wgrad_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        attr_set[dev] = true;
    }
    DicGemmParams q{};
    q.K = T; q.out_f32 = 1; q.split_k = 1; q.tile = 256;
    hipStream_t st = (hipStream_t)stream;
    tl_prof = prof_slot([&] { double f = 0; for (int i = 0; i < n; ++i) f += 2.0 * items[i].M * items[i].N * T; return f; }(),
                        [&] { double b = 0; for (int i = 0; i < n; ++i) b += ((double)items[i].M + items[i].N) * T * 2.0 + (double)items[i].M * items[i].N * 4.0; return b; }());
    launch_timed(wgrad_group_kernel, dim3(pl.grid), dim3(G::NTH), (unsigned)G::LDS, st, q, pl.dev);
    const int tiles = pl.dev.tiles;
    if (pl.dev.split > 1) launch_timed(wgrad_group_fold_kernel, dim3(tiles, G::BM / 16), dim3(256), 0u, st, pl.dev);      // (one slice per tile: written in place)
    tl_prof = nullptr;
    DIC_CHECK_LAUNCH();
    return 0;
}

// kept for ABI stability: the alternative K loops (round 3) left the library; 0 is the only variant
extern "C" int dic_gemm_set_variant(int v) {
    if (v == 0) return 0;
    dic_set_error("dic_gemm_set_variant: the library carries one K loop (the round-3 alternatives are in scripts/experiments/attic/)");
    return 1006;
}
// One setter for every process-global switch of the library (include/dic_hip.h lists the names); returns 0, or 1007 for an unknown name.
extern "C" int dic_set_option(const char* name, int value) {
    if (!name) { dic_set_error("dic_set_option: name is NULL"); return 1007; }
    if (!strcmp(name, "gemm_v1")) g_gemm_v1 = value ? 1 : 0;
    else if (!strcmp(name, "gemm_persist")) g_persist = value ? 1 : 0;
    else if (!strcmp(name, "gemm_rows")) g_rows = value ? 1 : 0;
    else if (!strcmp(name, "gemm_two_heights")) g_two_heights = value ? 1 : 0;
    else if (!strcmp(name, "gemm_w4a")) g_w4a = value ? 1 : 0;
    else if (!strcmp(name, "gemm_w4a_mask")) g_w4a_mask = value & 0x3FF;
    else if (!strcmp(name, "gemm_w4n")) g_w4n = value ? 1 : 0;
    else if (!strcmp(name, "gemm_w4n_mask")) g_w4n_mask = value & 0x7FF;
    else if (!strcmp(name, "gemm_w4n_flat")) g_w4n_flat = value ? 1 : 0;
    else if (!strcmp(name, "gemm_w4n_kmax")) { if (value < 576) { dic_set_error("dic_set_option: gemm_w4n_kmax is at least 576 (the narrow bodies need nine K-steps)"); return 1007; } g_w4n_kmax = value; }
    else if (!strcmp(name, "gemm_w4a_rows")) { if (value != 0 && value != 224 && value != 256) { dic_set_error("dic_set_option: gemm_w4a_rows is 0 (per launch), 224 or 256"); return 1007; } g_w4a_rows = value; }
    else { dic_set_error("dic_set_option: unknown option"); return 1007; }
    return 0;
}
// process-global measurement / test switch: 1 = eligible forward GEMMs run on the hand-scheduled four-wave kernel (gemm_w4a.h).  Returns the previous value.
extern "C" int dic_gemm_set_w4a(int on) {
    const int prev = w4a_mode();
    g_w4a = on ? 1 : 0;
    return prev;
}
// process-global measurement / test switch: 1 = launches may use two tile heights (default), 0 = one height per launch.  Returns the previous value.
extern "C" int dic_gemm_set_two_heights(int on) {
    const int prev = two_heights_enabled() ? 1 : 0;
    g_two_heights = on ? 1 : 0;
    return prev;
}

// host-only query: what dic_gemm would do with a forward (k-contiguous) bf16 problem of the 256-column geometry on this device --
// out[0] = fragments per wave of the tall tiles (0: one height), out[1] = of the last round's tiles, out[2] = rows covered by the tall tiles
extern "C" int dic_gemm_two_heights_plan(int M, int N, int K, int cu_cap, int* out) {
    int cus = device_cus();
    if (cu_cap > 0 && cu_cap < cus) cus = cu_cap;
    const int nbn = (N + Geo<T256>::BN - 1) / Geo<T256>::BN;
    const TwoHeights th = plan_two_heights(M, nbn, cus, K, pick_tile_rows(M, nbn, cus, Geo<T256>::BM, K), true);     // (whatever the switch says)
    out[0] = th.cA; out[1] = th.cB; out[2] = th.row_split;
    return 0;
}

// host-only query (a pure function of its arguments: no device needed): the tile height -- 256 or 224 rows -- the four-wave asm kernel takes for an
// M x N x K problem on `cus` compute units under the current "gemm_w4a_rows" option (gemm_w4a.h, w4a_pick_ni)
extern "C" int dic_gemm_w4a_rows_plan(int M, int N, int K, int cus) {
    if (M <= 0 || N <= 0 || K <= 0 || cus <= 0) { dic_set_error("dic_gemm_w4a_rows_plan: M, N, K, cus must be positive"); return -1; }
    return 32 * w4a_pick_ni(M, N, K, cus);
}

// ---- optional per-launch timing (bench.py roofline leg), see launch_timed above
namespace {
ProfRec* g_prof = nullptr;
int g_prof_cap = 0, g_prof_n = 0;
}  // namespace
extern "C" int dic_prof_begin(int max_launches) {
    if (g_prof) return 0;
    g_prof = new ProfRec[max_launches];
    for (int i = 0; i < max_launches; ++i) {
        (void)hipEventCreate(&g_prof[i].a); (void)hipEventCreate(&g_prof[i].b); (void)hipEventCreate(&g_prof[i].a2); (void)hipEventCreate(&g_prof[i].b2);
        g_prof[i].used = 0; g_prof[i].flops = 0;
    }
    g_prof_cap = max_launches; g_prof_n = 0;
    return 0;
}
// Sums the recorded launches (caller has synchronised the stream), frees the events.
extern "C" int dic_prof_end(double* total_ms, double* total_flops, int* n_launches) {
    double ms = 0, fl = 0;
    for (int i = 0; i < g_prof_n; ++i) {
        float t = 0;
        if (g_prof[i].used >= 1) { (void)hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b); ms += t; }
        if (g_prof[i].used >= 2) { (void)hipEventElapsedTime(&t, g_prof[i].a2, g_prof[i].b2); ms += t; }
        fl += g_prof[i].flops;
    }
    *total_ms = ms; *total_flops = fl; *n_launches = g_prof_n;
    for (int i = 0; i < g_prof_cap; ++i) {
        (void)hipEventDestroy(g_prof[i].a); (void)hipEventDestroy(g_prof[i].b); (void)hipEventDestroy(g_prof[i].a2); (void)hipEventDestroy(g_prof[i].b2);
    }
    delete[] g_prof; g_prof = nullptr; g_prof_cap = g_prof_n = 0;
    return 0;
}

static std::mutex g_prof_mu;
static ProfRec* prof_slot(double flops, double bytes) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof || g_prof_n >= g_prof_cap) return nullptr;
    ProfRec& r = g_prof[g_prof_n++];
    r.flops = flops; r.bytes = bytes; r.used = 0;
    return &r;
}
// One recorded launch (caller has synchronised; call before dic_prof_end): kernel time incl. its slab fold, flops, algorithmic bytes.  Returns 0 past the end.
extern "C" int dic_prof_get(int i, double* ms, double* flops, double* bytes) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof || i < 0 || i >= g_prof_n) return 0;
    float t = 0, t2 = 0;
    if (g_prof[i].used >= 1) (void)hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b);
    if (g_prof[i].used >= 2) (void)hipEventElapsedTime(&t2, g_prof[i].a2, g_prof[i].b2);
    *ms = (double)t + t2; *flops = g_prof[i].flops; *bytes = g_prof[i].bytes;
    return 1;
}
// ALGORITHMIC bytes of the launches recorded so far (call before dic_prof_end): every operand, side input and output of a GEMM once --
// what a launch would move if nothing were re-fetched and split-K partial sums never left the chip.  bench.py puts it next to the PMC traffic.
extern "C" double dic_prof_algorithmic_bytes(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double b = 0;
    for (int i = 0; i < g_prof_n; ++i) b += g_prof[i].bytes;
    return b;
}
static int dic_gemm_impl(int dtype, int a_km, int b_km, int epi, const DicGemmParams* pp, void* stream);
extern "C" int dic_gemm(int dtype, int a_km, int b_km, int epi, const DicGemmParams* pp, void* stream) {
    {
        const double es = dtype == DIC_BF16 ? 2.0 : 4.0, M = pp->M, N = pp->N, K = pp->K;
        double by = (M * K + N * K * (pp->B2 ? 2.0 : 1.0)) * es;
        if (pp->C) by += M * (epi == DIC_EPI_CE_EXP || epi == DIC_EPI_CE_DLOGITS ? (double)pp->ldc : N) * (pp->out_f32 ? 4.0 : es);
        if (pp->R) by += M * N * es;
        if (pp->aux) by += M * N * es;
        if (pp->accumulate) by += M * N * 4.0;
        tl_prof = prof_slot(2.0 * M * N * K * (pp->B2 ? 2.0 : 1.0), by);          // (executed flops: a split-weight launch runs its K loop twice)
    }
    const int rc = dic_gemm_impl(dtype, a_km, b_km, epi, pp, stream);
    tl_prof = nullptr;
    return rc;
}

static int dic_gemm_impl(int dtype, int a_km, int b_km, int epi, const DicGemmParams* pp, void* stream) {
    DicGemmParams p_ = *pp;
    { const DicStepCtx c = dic_step_ctx(); p_.step_ctr = (const int64_t*)c.ctr; p_.step_ctr0 = c.ctr0; }     // (see common.h: seeds under hipGraph replay)
    const DicGemmParams& p = p_;
    DIC_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "dic_gemm: empty problem");
    const int es = dtype == DIC_BF16 ? 2 : 4;
    const int bk = dtype == DIC_BF16 ? 64 : 32;
    if (!a_km || !b_km) DIC_REQUIRE(p.K % bk == 0, "dic_gemm: K must be a multiple of the K-step for k-contiguous operands (pad with zeros)");
    DIC_REQUIRE(((size_t)p.lda * es) % 16 == 0 && ((size_t)p.ldb * es) % 16 == 0, "dic_gemm: leading dimensions must be 16-byte multiples");
    DIC_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0, "dic_gemm: operands must be 16-byte aligned");
    if (a_km) DIC_REQUIRE((long long)p.K * p.lda * es < 0x7FFFFFFFll && p.M % 8 == 0, "dic_gemm: k-major A too large for 32-bit buffer offsets");
    if (b_km) DIC_REQUIRE((long long)p.K * p.ldb * es < 0x7FFFFFFFll, "dic_gemm: k-major B too large for 32-bit buffer offsets");
    if (!a_km) DIC_REQUIRE((long long)BM * p.lda * es < 0x7FFFFFFFll, "dic_gemm: lda too large");
    if (epi != DIC_EPI_CE_PARTIAL && epi != DIC_EPI_CE_DLOGITS && epi != DIC_EPI_CE_EXP) DIC_REQUIRE(p.N % 4 == 0 && p.ldc % 4 == 0, "dic_gemm: N and ldc must be multiples of 4");
    if (dtype == DIC_BF16 && (epi == DIC_EPI_BIAS_GELU || epi == DIC_EPI_GELU_BWD || epi == DIC_EPI_BIAS_GELU_D || epi == DIC_EPI_MUL_AUX))
        DIC_REQUIRE(p.N % 8 == 0 && p.ldc % 8 == 0 && (p.aux == nullptr || p.ldaux % 8 == 0), "dic_gemm: bf16 GELU epilogues need N, ldc, ldaux multiples of 8");
    if (epi == DIC_EPI_GELU_BWD || epi == DIC_EPI_MUL_AUX) DIC_REQUIRE(p.aux != nullptr, "dic_gemm: GELU_BWD / MUL_AUX need their side input (aux)");
    if (epi == DIC_EPI_BIAS_GELU_D || epi == DIC_EPI_MUL_AUX)
        DIC_REQUIRE(dtype == DIC_BF16 && !bf16_on_v1() && p.split_k <= 1, "dic_gemm: BIAS_GELU_D / MUL_AUX are epilogues of the bf16 LDS-DMA kernels");
    if (dtype == DIC_BF16) DIC_REQUIRE(p.ldc % 4 == 0 && (p.R == nullptr || p.ldr % 4 == 0), "dic_gemm: ldc/ldr must be multiples of 4");
    if (p.tile == 256)
        DIC_REQUIRE(dtype == DIC_BF16 && !(epi == DIC_EPI_CE_PARTIAL && bf16_on_v1()) && (!a_km || p.M % 256 == 0) && (!b_km || p.N % 256 == 0 || p.N % 8 == 0),
                    "dic_gemm: tile=256 is a bf16 option of the LDS-DMA kernel; k-major A needs M % 256 == 0");
    if (p.colsum_out)
        DIC_REQUIRE(dtype == DIC_BF16 && a_km && b_km && epi == DIC_EPI_AFFINE && p.out_f32 && p.M % 4 == 0,
                    "dic_gemm: colsum_out (fused bias gradient) is available on bf16 weight-gradient GEMMs (k-major A and B, fp32 output)");
    if (p.split_k > 1)
        DIC_REQUIRE(epi == DIC_EPI_AFFINE && p.out_f32 && p.split_ws && !p.bias && !p.R && p.p_drop == 0.f && p.ldc == p.N && p.split_k <= 64,
                    "dic_gemm: split-K needs the plain fp32-output AFFINE epilogue, ldc == N and a workspace of split_k*(M*N [+M]) floats");
    if (p.out_f32 & DIC_RES_IS_F32)
        DIC_REQUIRE(dtype == DIC_BF16 && !bf16_on_v1() && !a_km && epi == DIC_EPI_AFFINE && (p.out_f32 & DIC_OUT_F32) && p.R != nullptr && p.N % 8 == 0 && !p.accumulate &&
                    p.split_k <= 1,
                    "dic_gemm: an fp32 residual (out_f32 = DIC_OUT_F32 | DIC_RES_IS_F32) is an option of the bf16 LDS-DMA kernels' AFFINE epilogue (row-major A, N % 8 == 0, fp32 C)");
    if (p.B2)
        DIC_REQUIRE(dtype == DIC_BF16 && !bf16_on_v1() && !a_km && !b_km && (epi == DIC_EPI_AFFINE || epi == DIC_EPI_BIAS_GELU || epi == DIC_EPI_BIAS_GELU_D) && p.split_k <= 1 &&
                    ((uintptr_t)p.B2 % 16) == 0 && p.b2_col0 >= 0 && p.b2_col0 % 256 == 0,
                    "dic_gemm: B2 (low-order weight half) is an option of the bf16 forward GEMMs (k-contiguous A and B, AFFINE / BIAS_GELU, no split-K)");
    if (p.bias2)
        DIC_REQUIRE(dtype == DIC_BF16 && !bf16_on_v1() && !a_km && epi == DIC_EPI_AFFINE && p.R != nullptr && p.p_drop > 0.f && !p.out_f32 && !p.accumulate && p.N % 8 == 0 &&
                    p.split_k <= 1 && ((uintptr_t)p.bias2 % 16) == 0 && (p.tile == 256 || p.tile == 0 || p.tile == 128),
                    "dic_gemm: bias2 (a bias row behind the dropout) is an option of the bf16 forward AFFINE epilogue with dropout and a bf16 residual (N % 8 == 0); "
                    "without dropout add it to `bias`");
    if (epi == DIC_EPI_CE_EXP)
        DIC_REQUIRE(dtype == DIC_BF16 && !bf16_on_v1() && p.C && p.lse && p.partial && p.tgt_logit && p.ldc % 8 == 0 && p.ldc >= p.N &&
                    p.ldc <= ((p.N + BN - 1) / BN) * BN && p.split_k <= 1,
                    "dic_gemm: CE_EXP is a bf16 epilogue (LDS-DMA kernels); needs C, lse (reference points), partial, tgt_logit and an ldc that covers N within the last tile");
    if (epi == DIC_EPI_CE_DLOGITS) DIC_REQUIRE(p.ldc % (dtype == DIC_BF16 ? 8 : 4) == 0 && p.ldc >= p.N && p.ldc <= ((p.N + BN - 1) / BN) * BN, "dic_gemm: dlogits ldc must cover N within the last tile");
    hipStream_t st = (hipStream_t)stream;
    if (w4a_mode() == 1) {
        if (w4n_ce_ok(dtype, a_km, b_km, epi, p)) {
            const int rc = launch_w4n_ce(p, st);
            if (rc) return rc;
            DIC_CHECK_LAUNCH();
            return 0;
        }
        const int w4nv = w4n_variant(dtype, a_km, b_km, epi, p);
        if (w4nv >= 0) {
            launch_w4n(p, b_km, w4nv, st);
            DIC_CHECK_LAUNCH();
            return 0;
        }
        const int w4v = w4a_variant(dtype, a_km, b_km, epi, p);
        if (w4v >= 0) {
            launch_w4a(p, b_km, w4v, st);
            DIC_CHECK_LAUNCH();
            return 0;
        }
    }
    if (dtype == DIC_BF16) return launch_layout<bf16_t>(p, a_km, b_km, epi, st);
    if (dtype == DIC_F32) return launch_layout<float>(p, a_km, b_km, epi, st);
    dic_set_error("dic_gemm: unknown dtype");
    return 1004;
}
