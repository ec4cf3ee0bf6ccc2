// LayerNorm family (K4, K7/K8 epilogue LN, K9) for D = 768: one 64-lane wave per token row, each lane owning
// 3 chunks of 4 consecutive features (cols c*256 + lane*4 ..), i.e. 16-byte (f32) / 8-byte (bf16) coalesced accesses,
// statistics by wave shuffles in fp32 (two-pass mean/variance in registers), eps = 1e-12 as hf:100,236,239,439.
// Backward kernels are persistent: each wave walks rows with a grid stride, keeps its dgamma/dbeta/bias-grad partial
// sums in registers, and the block writes ONE partial row; dic_colsum folds the partial rows (deterministic).
#include <cstdlib>
#include "common.h"
#include "../../include/dic_hip.h"

namespace {
constexpr int D = 768, NCH = 3;

template <typename T>
__device__ __forceinline__ void load_row(const T* p, int lane, f32x4 (&v)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) v[c] = Elem<T>::ld4(p + c * 256 + lane * 4);
}
template <typename T>
__device__ __forceinline__ void store_row(T* p, int lane, const f32x4 (&v)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) Elem<T>::st4(p + c * 256 + lane * 4, v[c]);
}
__device__ __forceinline__ void row_stats(const f32x4 (&v)[NCH], float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
    mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) { float d = v[c][k] - mean; q += d * d; }
    rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
}

// block-level fold of per-wave register partials: acc[K][NCH] f32x4 per lane -> partial[blockIdx][K*D]
template <int K, int NWV = 4>
__device__ __forceinline__ void fold_partials(f32x4 (&acc)[K][NCH], float* partial, float* lds) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int c = 0; c < NCH; ++c) *(f32x4*)(lds + ((w * K + k) * D) + c * 256 + lane * 4) = acc[k][c];
    __syncthreads();
    for (int i = threadIdx.x; i < K * D; i += 64 * NWV) {
        float s = lds[i];
#pragma unroll
        for (int q = 1; q < NWV; ++q) s += lds[q * K * D + i];
        partial[(size_t)blockIdx.x * K * D + i] = s;
    }
}

// Backward LayerNorm kernels run persistent blocks of 8 waves: with a few hundred blocks they are bound by memory LATENCY (every
// row iteration is one load round trip), so waves in flight per CU are what buy bandwidth (256 blocks x 4 waves: 35 us for 108 MB,
// 512 x 8: 20 us).
constexpr int LNB_WAVES = 8;

// ---------------------------------------------------------------------------------------------- fused input rows
// mode 0 concat (ref :299-300): rows t<L: x + seg0 + pos[t]; row L: img + seg1 + pos[L]; row L+1: txt + seg1 + pos[L+1]
// mode 1 add    (ref :306-307): rows t<L: x + img (+ txt if add_txt[n]) + pos[t]
// mode 2 concat without the text row (Tk = L+1): when no row of the batch is classifier-free-guided the text row is masked
//        as a key for every query and its own outputs are never read (ref :297, :323, :418), so it can be left out of the
//        sequence without changing any loss or any gradient (text_linear's gradient is exactly zero either way).
// Optional timestep embedding (BASELINE north_star names the operand; the reference has none -- its forward takes no t, ref :271 -- so it
// is off, temb == NULL, in every parity configuration): row temb[tidx[n]] of a learned [steps][D] table is added to every token row of
// sequence n before the LayerNorm (tidx[n] < 0: none for that sequence).
__device__ __forceinline__ void fused_row(int mode, const float* x, long long xs, const float* img, const float* txt, const uint8_t* add_txt,
                                          const float* seg, const float* pos, const float* temb, const int* tidx, int n, int t, int L, int lane,
                                          f32x4 (&v)[NCH]) {          // xs: elements between the first rows of consecutive sequences of x (L*D when packed)
    f32x4 p[NCH];
    load_row<float>(pos + (size_t)t * D, lane, p);
    if (temb) {
        const int ts = tidx[n];
        if (ts >= 0) {
            f32x4 e[NCH];
            load_row<float>(temb + (size_t)ts * D, lane, e);
#pragma unroll
            for (int c = 0; c < NCH; ++c) p[c] = p[c] + e[c];
        }
    }
    if (mode != 1) {
        const float* src = t < L ? x + (size_t)n * xs + (size_t)t * D : (t == L ? img + (size_t)n * D : txt + (size_t)n * D);
        f32x4 s[NCH];
        load_row<float>(src, lane, v);
        load_row<float>(seg + (t < L ? 0 : D), lane, s);
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[c] = (v[c] + s[c]) + p[c];     // hstack + segment (ref :300), then + position (hf:115)
    } else {
        f32x4 a[NCH];
        load_row<float>(x + (size_t)n * xs + (size_t)t * D, lane, v);
        load_row<float>(img + (size_t)n * D, lane, a);
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[c] = v[c] + a[c];
        if (add_txt && add_txt[n]) {
            load_row<float>(txt + (size_t)n * D, lane, a);
#pragma unroll
            for (int c = 0; c < NCH; ++c) v[c] = v[c] + a[c];
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[c] = v[c] + p[c];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fuse_ln_fwd_kernel(int mode, const float* x, long long xs, const float* img, const float* txt, const uint8_t* add_txt,
                                                           const float* seg, const float* pos, const float* temb, const int* tidx,
                                                           const float* gamma, const float* beta, T* h,
                                                           float* mean, float* rstd, int N, int L, int Tk, float eps, float p_drop,
                                                           SeedArg seed_) {
    const unsigned long long seed = seed_.resolve();
    const int lane = threadIdx.x & 63;
    const int rows = N * Tk;
    f32x4 g[NCH], b[NCH];
    load_row<float>(gamma, lane, g);
    load_row<float>(beta, lane, b);
    const float inv_keep = drop_inv_keep(p_drop);
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const int n = row / Tk, t = row - n * Tk;
        f32x4 v[NCH];
        fused_row(mode, x, xs, img, txt, add_txt, seg, pos, temb, tidx, n, t, L, lane, v);
        float mu, rs;
        row_stats(v, eps, mu, rs);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[c][k] = (v[c][k] - mu) * rs * g[c][k] + b[c][k];
            if (p_drop > 0.f) v[c] = dropout4(v[c], seed, (unsigned long long)row * D + c * 256 + lane * 4, p_drop, inv_keep);
        }
        store_row<T>(h + (size_t)row * D, lane, v);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

template <typename T>
__global__ __launch_bounds__(64 * LNB_WAVES) void fuse_ln_bwd_kernel(int mode, const float* x, const float* img, const float* txt, const uint8_t* add_txt,
                                                           const float* seg, const float* pos, const float* temb, const int* tidx,
                                                           const float* gamma, const T* dh,
                                                           const float* mean, const float* rstd, float* dy, float* partial, int N, int L, int Tk,
                                                           float p_drop, SeedArg seed_) {
    const unsigned long long seed = seed_.resolve();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int rows = N * Tk;
    f32x4 g[NCH];
    load_row<float>(gamma, lane, g);
    f32x4 acc[2][NCH];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[k][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float inv_keep = drop_inv_keep(p_drop);
    for (int row = blockIdx.x * LNB_WAVES + (threadIdx.x >> 6); row < rows; row += gridDim.x * LNB_WAVES) {
        const int n = row / Tk, t = row - n * Tk;
        f32x4 v[NCH], d[NCH];
        fused_row(mode, x, (long long)L * D, img, txt, add_txt, seg, pos, temb, tidx, n, t, L, lane, v);
        load_row<T>(dh + (size_t)row * D, lane, d);
        const float mu = mean[row], rs = rstd[row];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (p_drop > 0.f) d[c] = dropout4(d[c], seed, (unsigned long long)row * D + c * 256 + lane * 4, p_drop, inv_keep);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float xh = (v[c][k] - mu) * rs;
                float gg = d[c][k] * g[c][k];
                acc[0][c][k] += d[c][k] * xh;
                acc[1][c][k] += d[c][k];
                c1 += gg; c2 += gg * xh;
                v[c][k] = xh; d[c][k] = gg;
            }
        }
        c1 = wave_sum(c1) * (1.0f / D);
        c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) d[c][k] = rs * (d[c][k] - c1 - v[c][k] * c2);
        store_row<float>(dy + (size_t)row * D, lane, d);
    }
    fold_partials<2, LNB_WAVES>(acc, partial, lds);
}

// ---------------------------------------------------------------------------------------------- plain LayerNorm
// TY: type of the pre-LayerNorm sum y (fp32 in the fp32-residual-stream mode of the bf16 engines, dic_ln_fwd_r32); h32: optional fp32 copy of the
// output, the residual operand of the next GEMM (the bf16 copy h stays that GEMM's / the next Linear's MFMA operand)
template <typename T, typename TY = T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TY* y, const float* gamma, const float* beta, T* h, float* h32, float* mean, float* rstd, int rows, float eps) {
    const int lane = threadIdx.x & 63;
    f32x4 g[NCH], b[NCH];
    load_row<float>(gamma, lane, g);
    load_row<float>(beta, lane, b);
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        f32x4 v[NCH];
        load_row<TY>(y + (size_t)row * D, lane, v);
        float mu, rs;
        row_stats(v, eps, mu, rs);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[c][k] = (v[c][k] - mu) * rs * g[c][k] + b[c][k];
        store_row<T>(h + (size_t)row * D, lane, v);
        if (h32) store_row<float>(h32 + (size_t)row * D, lane, v);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// CENTRED RESIDUAL STREAM (round 5; replaces the fp32 copies of dic_ln_fwd_r32 in the parity mode): the pre-LayerNorm sum arrives as
// y_c = bf16(y - y_ref) with ONE fp32 reference row y_ref per tensor (predicted by dic_lin_prep before the GEMM ran), the LayerNorm runs on
// y_c + y_ref in fp32 and leaves two bf16 tensors: h, the next Linear's MFMA operand, and h_c = bf16(h - h_ref), the residual operand of
// the next residual GEMM, centred on h_ref = LayerNorm(y_ref) (every wave computes it; block 0 publishes it).  While a denoiser's rows are
// nearly equal (the first hundreds of training steps) the rounding error of a bf16 residual stream is the SAME vector for every token and
// does not average out of a batch-mean loss; stored relative to a row all tokens are close to, it is 2^-9 of the rows' differences instead
// of 2^-9 of the rows (profiles/r05_cen_probe.txt: the same loss distances as the fp32 residual stream, at bf16 bytes).
// nx_*: optional "next Linear" tail: nx_out[n] = nx_bias[n] + sum_k ([nx_hi] + [nx_lo])[n][k] h_ref[k], one output per wave, no cross-workgroup
// step -- every wave already holds h_ref.  nx_hi: the next Linear multiplies the CENTRED tensor h_c by bf16(W) = W_hi, so the reference row's
// share of that product belongs in its bias -- (h_c + 1 h_ref^T) W_hi^T = h_c W_hi^T + 1 (W_hi h_ref)^T, exact algebra.  nx_lo: the lo half's
// share W_lo h_ref of the mean-row correction W_lo abar, abar = h_ref + mean(h_c): the Linear's dic_lin_prep adds the measured rest W_lo
// mean(h_c), or -- q|k|v of layers >= 1, options.qkv_pred -- it is left at the prediction mean(h_c) = 0.  (FFN lin1 takes nx_hi only: it gets no
// lo correction at all, and a PREDICTED one would hurt it at the synthetic initial weights: profiles/r05_pred_probe.txt.)  Used for the q|k|v projection of layers >= 1 (options.qkv_pred): two launches fewer per layer on the
// critical path; what the prediction costs in accuracy is measured in profiles/r05_pred_probe.txt (nothing once the rows have begun to
// collapse, 1e-5 of the L1 terms at the synthetic initial weights -- FFN lin1, by contrast, must NOT be fed this way: 1.2e-4).
__global__ __launch_bounds__(256) void ln_fwd_cen_kernel(const bf16_t* y_c, const float* y_ref, const float* gamma, const float* beta, bf16_t* h, bf16_t* h_c,
                                                         float* h_ref, float* mean, float* rstd, int rows, float eps,
                                                         const bf16_t* nx_hi, const bf16_t* nx_lo, int nx_ldb, int nx_n, const float* nx_bias, float* nx_out) {
    const int lane = threadIdx.x & 63;
    f32x4 g[NCH], b[NCH], yr[NCH], hr[NCH];
    load_row<float>(gamma, lane, g);
    load_row<float>(beta, lane, b);
    load_row<float>(y_ref, lane, yr);
    {
        // h_ref: the LayerNorm of the reference row -- with the scale of a TYPICAL row, not its own.  While the rows are nearly equal the two are
        // the same; while they differ (initial weights: rms distance from the mean row ~ the rows' own spread) the mean row has a much smaller
        // variance than any row, LN(y_ref) would blow it up to unit scale, and a reference row that is far from every row doubles what the
        // centred tensor's bf16 rounding costs (measured at B = 16: 1.2e-4 of the loss).  mean over rows of (y_r - mu_r) rstd_r ~ (ybar - mu(ybar))
        // rstd* when rstd_r ~ rstd*: rstd* comes from four fixed sample rows (the same in every workgroup: deterministic), read by wave 0.
        __shared__ float s_var;
        if (threadIdx.x < 64) {
            float vs = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = (int)(((long long)rows * q) / 4);
                f32x4 t[NCH];
                load_row<bf16_t>(y_c + (size_t)r * D, lane, t);
                float sm = 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) { t[c] = t[c] + yr[c]; sm += t[c][0] + t[c][1] + t[c][2] + t[c][3]; }
                const float m = wave_sum(sm) * (1.0f / D);
                float qv = 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const float d = t[c][k] - m; qv += d * d; }
                vs += wave_sum(qv) * (1.0f / D);
            }
            if (lane == 0) s_var = vs * 0.25f;
        }
        __syncthreads();
        float mu, rs;
        row_stats(yr, eps, mu, rs);
        const float rs_typ = rsqrtf(s_var + eps);
        rs = rs_typ < rs ? rs_typ : rs;                    // (never sharper than the reference row's own normalisation)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) hr[c][k] = (yr[c][k] - mu) * rs * g[c][k] + b[c][k];
        if (h_ref && blockIdx.x == 0 && threadIdx.x < 64) store_row<float>(h_ref, lane, hr);
    }
    if (nx_out) {
        const int nw = gridDim.x * 4;
        for (int n = blockIdx.x * 4 + (threadIdx.x >> 6); n < nx_n; n += nw) {
            float a = 0.f;
            if (nx_lo) {
                f32x4 wv[NCH];
                load_row<bf16_t>(nx_lo + (size_t)n * nx_ldb, lane, wv);
#pragma unroll
                for (int c = 0; c < NCH; ++c) a += (wv[c][0] * hr[c][0] + wv[c][1] * hr[c][1]) + (wv[c][2] * hr[c][2] + wv[c][3] * hr[c][3]);
            }
            if (nx_hi) {
                f32x4 wv[NCH];
                load_row<bf16_t>(nx_hi + (size_t)n * nx_ldb, lane, wv);
                float a2 = 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) a2 += (wv[c][0] * hr[c][0] + wv[c][1] * hr[c][1]) + (wv[c][2] * hr[c][2] + wv[c][3] * hr[c][3]);
                a += a2;
            }
            a = wave_sum(a);
            if (lane == 0) nx_out[n] = (nx_bias ? nx_bias[n] : 0.f) + a;
        }
    }
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        f32x4 v[NCH];
        load_row<bf16_t>(y_c + (size_t)row * D, lane, v);
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[c] = v[c] + yr[c];
        float mu, rs;
        row_stats(v, eps, mu, rs);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[c][k] = (v[c][k] - mu) * rs * g[c][k] + b[c][k];
        if (h) store_row<bf16_t>(h + (size_t)row * D, lane, v);
        if (h_c) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) v[c] = v[c] - hr[c];
            store_row<bf16_t>(h_c + (size_t)row * D, lane, v);
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

template <typename T, typename TY = T>
__global__ __launch_bounds__(64 * LNB_WAVES) void ln_bwd_kernel(const T* dh, const TY* y, const float* gamma, const float* mean, const float* rstd, T* dx, T* dx_drop,
                                                      float p_drop, SeedArg seed_, float* partial, int rows, const float* y_ref = nullptr) {
    const unsigned long long seed = seed_.resolve();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    f32x4 g[NCH], yr[NCH];
    load_row<float>(gamma, lane, g);
    if (y_ref) load_row<float>(y_ref, lane, yr);         // centred residual stream: y holds bf16(y - y_ref) (ln_fwd_cen_kernel)
    f32x4 acc[3][NCH];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[k][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float inv_keep = drop_inv_keep(p_drop);
    // (A software-pipelined variant -- next row's loads issued before this row's stores -- was measured in round 3: 24.5 us instead of 19.4 us per
    // launch at 17 408 tokens, no gain at 48 756: the kernel alone already streams at 5.4 TB/s; what looked slow in a two-stream profile was CU
    // contention with the weight-gradient GEMMs.)
    for (int row = blockIdx.x * LNB_WAVES + (threadIdx.x >> 6); row < rows; row += gridDim.x * LNB_WAVES) {
        f32x4 v[NCH], d[NCH];
        load_row<TY>(y + (size_t)row * D, lane, v);
        load_row<T>(dh + (size_t)row * D, lane, d);
        const float mu = mean[row], rs = rstd[row];
        if (y_ref) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) v[c] = v[c] + yr[c];
        }
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float xh = (v[c][k] - mu) * rs;
                float gg = d[c][k] * g[c][k];
                acc[0][c][k] += d[c][k] * xh;
                acc[1][c][k] += d[c][k];
                c1 += gg; c2 += gg * xh;
                v[c][k] = xh; d[c][k] = gg;
            }
        c1 = wave_sum(c1) * (1.0f / D);
        c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) d[c][k] = rs * (d[c][k] - c1 - v[c][k] * c2);
        store_row<T>(dx + (size_t)row * D, lane, d);
        if (dx_drop) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) d[c] = dropout4(d[c], seed, (unsigned long long)row * D + c * 256 + lane * 4, p_drop, inv_keep);
            store_row<T>(dx_drop + (size_t)row * D, lane, d);
        }
        // bias gradient of the Linear that produced y: column sum of what flows into it
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[2][c] += d[c];
    }
    fold_partials<3, LNB_WAVES>(acc, partial, lds);
}

#ifdef DIC_LN_THIN      // measurement build only (scripts/build_variant.sh -DDIC_LN_THIN): the shipped library carries one LayerNorm backward
// "Thin" form (experiment, DIC_LN_BWD_ROWS = 2 / 4): a wave takes ROWS rows per iteration and issues all their loads before the first store, so
// the same bytes are in flight from 1/ROWS of the waves -- the kernel can then saturate HBM from a fraction of the CUs (fewer persistent blocks,
// DIC_LN_NPART) and leave the rest to a GEMM on another stream instead of time-slicing whole CUs with it (what comes next).
template <typename T, typename TY, int ROWS>
__global__ __launch_bounds__(64 * LNB_WAVES) void ln_bwd_thin_kernel(const T* dh, const TY* y, const float* gamma, const float* mean, const float* rstd, T* dx, T* dx_drop,
                                                           float p_drop, SeedArg seed_, float* partial, int rows) {
    const unsigned long long seed = seed_.resolve();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    f32x4 g[NCH];
    load_row<float>(gamma, lane, g);
    f32x4 acc[3][NCH];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[k][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float inv_keep = drop_inv_keep(p_drop);
    const int nw = gridDim.x * LNB_WAVES, w = blockIdx.x * LNB_WAVES + (threadIdx.x >> 6);
    for (int row0 = w; row0 < rows; row0 += nw * ROWS) {
        f32x4 v[ROWS][NCH], d[ROWS][NCH];
        float mu[ROWS], rs[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int row = row0 + j * nw;
            const int rr = row < rows ? row : rows - 1;              // (clamped: surplus rows are computed and dropped)
            load_row<TY>(y + (size_t)rr * D, lane, v[j]);
            load_row<T>(dh + (size_t)rr * D, lane, d[j]);
            mu[j] = mean[rr]; rs[j] = rstd[rr];
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int row = row0 + j * nw;
            if (row >= rows) break;
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float xh = (v[j][c][k] - mu[j]) * rs[j];
                    float gg = d[j][c][k] * g[c][k];
                    acc[0][c][k] += d[j][c][k] * xh;
                    acc[1][c][k] += d[j][c][k];
                    c1 += gg; c2 += gg * xh;
                    v[j][c][k] = xh; d[j][c][k] = gg;
                }
            c1 = wave_sum(c1) * (1.0f / D);
            c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) d[j][c][k] = rs[j] * (d[j][c][k] - c1 - v[j][c][k] * c2);
            store_row<T>(dx + (size_t)row * D, lane, d[j]);
            if (dx_drop) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) d[j][c] = dropout4(d[j][c], seed, (unsigned long long)row * D + c * 256 + lane * 4, p_drop, inv_keep);
                store_row<T>(dx_drop + (size_t)row * D, lane, d[j]);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[2][c] += d[j][c];
        }
    }
    fold_partials<3, LNB_WAVES>(acc, partial, lds);
}

#endif

// ---------------------------------------------------------------------------------------------- GELU + LayerNorm (hf:511-512)
template <typename T>
__global__ __launch_bounds__(256) void gelu_ln_fwd_kernel(const T* u, const float* gamma, const float* beta, float* x_out, float* mean, float* rstd, int rows, float eps) {
    const int lane = threadIdx.x & 63;
    f32x4 g[NCH], b[NCH];
    load_row<float>(gamma, lane, g);
    load_row<float>(beta, lane, b);
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        f32x4 v[NCH];
        load_row<T>(u + (size_t)row * D, lane, v);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[c][k] = gelu_f(v[c][k]);
        float mu, rs;
        row_stats(v, eps, mu, rs);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[c][k] = (v[c][k] - mu) * rs * g[c][k] + b[c][k];
        store_row<float>(x_out + (size_t)row * D, lane, v);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// TU: type of the stored pre-activation u (fp32 in the bf16 engines when DIC_U_F32 is set: see dic_gelu_ln_fwd), T: type of the gradient written
template <typename T, typename TU = T>
__global__ __launch_bounds__(64 * LNB_WAVES) void gelu_ln_bwd_kernel(const float* dx_out, const TU* u, const float* gamma, const float* mean, const float* rstd, T* du,
                                                           float* partial, int rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    f32x4 g[NCH];
    load_row<float>(gamma, lane, g);
    f32x4 acc[3][NCH];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[k][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int row = blockIdx.x * LNB_WAVES + (threadIdx.x >> 6); row < rows; row += gridDim.x * LNB_WAVES) {
        f32x4 uu[NCH], v[NCH], d[NCH];
        load_row<TU>(u + (size_t)row * D, lane, uu);
        load_row<float>(dx_out + (size_t)row * D, lane, d);
        const float mu = mean[row], rs = rstd[row];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float xh = (gelu_f(uu[c][k]) - mu) * rs;
                float gg = d[c][k] * g[c][k];
                acc[0][c][k] += d[c][k] * xh;
                acc[1][c][k] += d[c][k];
                c1 += gg; c2 += gg * xh;
                v[c][k] = xh; d[c][k] = gg;
            }
        c1 = wave_sum(c1) * (1.0f / D);
        c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int k = 0; k < 4; ++k) d[c][k] = rs * (d[c][k] - c1 - v[c][k] * c2) * gelu_grad_f(uu[c][k]);
            acc[2][c] += d[c];
        }
        store_row<T>(du + (size_t)row * D, lane, d);
    }
    fold_partials<3, LNB_WAVES>(acc, partial, lds);
}

inline int rows_grid(int rows, int cap) { int g = (rows + 3) / 4; return g < 1 ? 1 : (g > cap ? cap : g); }
}  // namespace

#define DISPATCH_T(dtype, CALL_BF, CALL_F32) do { if ((dtype) == DIC_BF16) { CALL_BF; } else { CALL_F32; } } while (0)

extern "C" int dic_fuse_ln_fwd_x(int dtype, int mode, const float* x, int64_t x_seq_stride, const float* img, const float* txt, const uint8_t* add_txt,
                                 const float* seg, const float* pos, const float* temb, const int32_t* tidx, const float* gamma, const float* beta,
                                 void* h, float* mean, float* rstd, int N, int L, int Dd, float eps, float p_drop, uint64_t seed, void* stream) {
    DIC_REQUIRE(temb == nullptr || tidx != nullptr, "dic_fuse_ln_fwd: a timestep-embedding table needs the per-sequence indices");
    DIC_REQUIRE(Dd == D, "dic_fuse_ln_fwd: D must be 768");
    DIC_REQUIRE(x_seq_stride >= (int64_t)L * D && x_seq_stride % 4 == 0, "dic_fuse_ln_fwd_x: the sequence stride of x must cover L rows (and keep 16-byte alignment)");
    const int Tk = mode == 0 ? L + 2 : (mode == 2 ? L + 1 : L);
    dim3 grid(rows_grid(N * Tk, 2048)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const long long xs = (long long)x_seq_stride;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(fuse_ln_fwd_kernel<bf16_t>, grid, block, 0, st, mode, x, xs, img, txt, add_txt, seg, pos, temb, tidx, gamma, beta, (bf16_t*)h, mean, rstd, N, L, Tk, eps, p_drop, make_seed(seed, DIC_STRIDE_DROP)),
               hipLaunchKernelGGL(fuse_ln_fwd_kernel<float>, grid, block, 0, st, mode, x, xs, img, txt, add_txt, seg, pos, temb, tidx, gamma, beta, (float*)h, mean, rstd, N, L, Tk, eps, p_drop, make_seed(seed, DIC_STRIDE_DROP)));
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_fuse_ln_fwd(int dtype, int mode, const float* x, const float* img, const float* txt, const uint8_t* add_txt,
                               const float* seg, const float* pos, const float* temb, const int32_t* tidx, const float* gamma, const float* beta,
                               void* h, float* mean, float* rstd, int N, int L, int Dd, float eps, float p_drop, uint64_t seed, void* stream) {
    return dic_fuse_ln_fwd_x(dtype, mode, x, (int64_t)L * D, img, txt, add_txt, seg, pos, temb, tidx, gamma, beta, h, mean, rstd, N, L, Dd, eps, p_drop, seed, stream);
}
extern "C" int dic_fuse_ln_bwd(int dtype, int mode, const float* x, const float* img, const float* txt, const uint8_t* add_txt,
                               const float* seg, const float* pos, const float* temb, const int32_t* tidx, const float* gamma, const void* dh,
                               const float* mean, const float* rstd, float* dy, float* partial, int n_partial_blocks, int N, int L, int Dd,
                               float p_drop, uint64_t seed, void* stream) {
    DIC_REQUIRE(Dd == D && n_partial_blocks > 0, "dic_fuse_ln_bwd: D must be 768");
    const int Tk = mode == 0 ? L + 2 : (mode == 2 ? L + 1 : L);
    dim3 grid(n_partial_blocks), block(64 * LNB_WAVES);
    const size_t lds = LNB_WAVES * 2 * D * sizeof(float);            // 48 KB
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(fuse_ln_bwd_kernel<bf16_t>, grid, block, lds, st, mode, x, img, txt, add_txt, seg, pos, temb, tidx, gamma, (const bf16_t*)dh, mean, rstd, dy, partial, N, L, Tk, p_drop, make_seed(seed, DIC_STRIDE_DROP)),
               hipLaunchKernelGGL(fuse_ln_bwd_kernel<float>, grid, block, lds, st, mode, x, img, txt, add_txt, seg, pos, temb, tidx, gamma, (const float*)dh, mean, rstd, dy, partial, N, L, Tk, p_drop, make_seed(seed, DIC_STRIDE_DROP)));
    DIC_CHECK_LAUNCH();
    return 0;
}
// d temb[s] = sum over the sequences n with tidx[n] == s, and over their Tk token rows, of dy (the gradient wrt the pre-LayerNorm rows that
// dic_fuse_ln_bwd wrote).  One block per (timestep, 256-column chunk), sequences visited in index order: deterministic, no atomics; a step
// uses at most S + 1 distinct timesteps, every other block finds no match and writes zeros.
__global__ __launch_bounds__(64) void temb_grad_kernel(const float* dy, const int* tidx, int N, int Tk, float* dtemb) {
    const int ts = blockIdx.x, col = blockIdx.y * 256 + threadIdx.x * 4;
    f32x4 acc{0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < N; ++n) {
        if (tidx[n] != ts) continue;
        const float* src = dy + (size_t)n * Tk * D + col;
        for (int t = 0; t < Tk; ++t) acc += *(const f32x4*)(src + (size_t)t * D);
    }
    *(f32x4*)(dtemb + (size_t)ts * D + col) = acc;
}
extern "C" int dic_temb_grad(const float* dy, const int32_t* tidx, int N, int Tk, int Dd, int steps, float* dtemb, void* stream) {
    DIC_REQUIRE(Dd == D && steps > 0 && N > 0, "dic_temb_grad: D must be 768");
    hipLaunchKernelGGL(temb_grad_kernel, dim3(steps, D / 256), dim3(64), 0, (hipStream_t)stream, dy, tidx, N, Tk, dtemb);
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_ln_fwd(int dtype, const void* y, const float* gamma, const float* beta, void* h, float* mean, float* rstd, int T,
                          int Dd, float eps, void* stream) {
    DIC_REQUIRE(Dd == D && T > 0, "dic_ln_fwd: D must be 768");
    dim3 grid(rows_grid(T, 2048)), block(256);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)y, gamma, beta, (bf16_t*)h, (float*)nullptr, mean, rstd, T, eps),
               hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, block, 0, st, (const float*)y, gamma, beta, (float*)h, (float*)nullptr, mean, rstd, T, eps));
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_ln_fwd_r32(const float* y32, const float* gamma, const float* beta, void* h_bf16, float* h32, float* mean, float* rstd, int T,
                              int Dd, float eps, void* stream) {
    DIC_REQUIRE(Dd == D && T > 0 && h_bf16 != nullptr, "dic_ln_fwd_r32: D must be 768");
    dim3 grid(rows_grid(T, 2048)), block(256);
    hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, float>), grid, block, 0, (hipStream_t)stream, y32, gamma, beta, (bf16_t*)h_bf16, h32, mean, rstd, T, eps);
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_ln_fwd_cen(const void* y_c, const float* y_ref, const float* gamma, const float* beta, void* h, void* h_c, float* h_ref, float* mean,
                              float* rstd, int T, int Dd, float eps, const void* next_hi, const void* next_lo, int next_ldb, int next_n, const float* next_bias,
                              float* next_bias_out, void* stream) {
    DIC_REQUIRE(Dd == D && T > 0 && y_c && y_ref && (h || h_c), "dic_ln_fwd_cen: D must be 768; y_c (bf16), y_ref (fp32 [768]) and one of h / h_c are required");
    DIC_REQUIRE((next_lo == nullptr && next_hi == nullptr) || (next_bias_out != nullptr && next_n > 0 && next_ldb >= D && next_ldb % 4 == 0),
                "dic_ln_fwd_cen: the next-Linear tail needs next_bias_out, next_n > 0 and next_ldb >= 768 (a multiple of 4)");
    if (next_lo == nullptr && next_hi == nullptr) next_bias_out = nullptr;
    dim3 grid(rows_grid(T, 1024)), block(256);
    hipLaunchKernelGGL(ln_fwd_cen_kernel, grid, block, 0, (hipStream_t)stream, (const bf16_t*)y_c, y_ref, gamma, beta, (bf16_t*)h, (bf16_t*)h_c, h_ref, mean, rstd, T, eps,
                       (const bf16_t*)next_hi, (const bf16_t*)next_lo, next_ldb, next_n, next_bias, next_bias_out);
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_ln_bwd_cen(const void* dh, const void* y_c, const float* y_ref, const float* gamma, const float* mean, const float* rstd, void* dx,
                              void* dx_drop, float p_drop, uint64_t seed, float* partial, int n_partial_blocks, int T, int Dd, void* stream) {
    DIC_REQUIRE(Dd == D && T > 0 && n_partial_blocks > 0 && y_ref, "dic_ln_bwd_cen: D must be 768; y_ref is the reference row of dic_ln_fwd_cen's input");
    dim3 grid(n_partial_blocks), block(64 * LNB_WAVES);
    const size_t lds = LNB_WAVES * 3 * D * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)ln_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, grid, block, lds, (hipStream_t)stream, (const bf16_t*)dh, (const bf16_t*)y_c, gamma, mean, rstd, (bf16_t*)dx,
                       (bf16_t*)dx_drop, p_drop, make_seed(seed, DIC_STRIDE_DROP), partial, T, y_ref);
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_ln_bwd(int dtype, const void* dh, const void* y, const float* gamma, const float* mean, const float* rstd, void* dx,
                          void* dx_drop, float p_drop, uint64_t seed, float* partial, int n_partial_blocks, int T, int Dd, void* stream) {
    DIC_REQUIRE(Dd == D && T > 0 && n_partial_blocks > 0, "dic_ln_bwd: D must be 768");
    dim3 grid(n_partial_blocks), block(64 * LNB_WAVES);
    const size_t lds = LNB_WAVES * 3 * D * sizeof(float);              // 72 KB: above the 64 KB default cap of dynamic LDS
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)ln_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)ln_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)ln_bwd_kernel<bf16_t, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
#ifdef DIC_LN_THIN
    static const int thin_rows = [] { const char* e = getenv("DIC_LN_BWD_ROWS"); return e ? atoi(e) : 1; }();
    if (dtype == DIC_BF16 && (thin_rows == 2 || thin_rows == 4)) {          // experiment: the thin form (bf16 engine only)
        static bool thin_attr = false;
        if (!thin_attr) {
            (void)hipFuncSetAttribute((const void*)ln_bwd_thin_kernel<bf16_t, bf16_t, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)ln_bwd_thin_kernel<bf16_t, bf16_t, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            thin_attr = true;
        }
        if (thin_rows == 2)
            hipLaunchKernelGGL((ln_bwd_thin_kernel<bf16_t, bf16_t, 2>), grid, block, lds, st, (const bf16_t*)dh, (const bf16_t*)y, gamma, mean, rstd, (bf16_t*)dx, (bf16_t*)dx_drop, p_drop, make_seed(seed, DIC_STRIDE_DROP), partial, T);
        else
            hipLaunchKernelGGL((ln_bwd_thin_kernel<bf16_t, bf16_t, 4>), grid, block, lds, st, (const bf16_t*)dh, (const bf16_t*)y, gamma, mean, rstd, (bf16_t*)dx, (bf16_t*)dx_drop, p_drop, make_seed(seed, DIC_STRIDE_DROP), partial, T);
        DIC_CHECK_LAUNCH();
        return 0;
    }
#endif
    if (dtype == (DIC_BF16 | DIC_RES_F32)) {           // fp32 residual stream: y is fp32, the gradients stay bf16
        hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, float>), grid, block, lds, st, (const bf16_t*)dh, (const float*)y, gamma, mean, rstd, (bf16_t*)dx, (bf16_t*)dx_drop, p_drop, make_seed(seed, DIC_STRIDE_DROP), partial, T);
        DIC_CHECK_LAUNCH();
        return 0;
    }
    dtype &= ~DIC_RES_F32;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, grid, block, lds, st, (const bf16_t*)dh, (const bf16_t*)y, gamma, mean, rstd, (bf16_t*)dx, (bf16_t*)dx_drop, p_drop, make_seed(seed, DIC_STRIDE_DROP), partial, T),
               hipLaunchKernelGGL(ln_bwd_kernel<float>, grid, block, lds, st, (const float*)dh, (const float*)y, gamma, mean, rstd, (float*)dx, (float*)dx_drop, p_drop, make_seed(seed, DIC_STRIDE_DROP), partial, T));
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_gelu_ln_fwd(int dtype, const void* u, const float* gamma, const float* beta, float* x_out, float* mean, float* rstd,
                               int T, int Dd, float eps, void* stream) {
    DIC_REQUIRE(Dd == D && T > 0, "dic_gelu_ln_fwd: D must be 768");
    dim3 grid(rows_grid(T, 2048)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype & DIC_U_F32) dtype = DIC_F32;            // u was stored in fp32 (x_out always is): the fp32 kernel is the whole forward
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(gelu_ln_fwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)u, gamma, beta, x_out, mean, rstd, T, eps),
               hipLaunchKernelGGL(gelu_ln_fwd_kernel<float>, grid, block, 0, st, (const float*)u, gamma, beta, x_out, mean, rstd, T, eps));
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_gelu_ln_bwd(int dtype, const float* dx_out, const void* u, const float* gamma, const float* mean, const float* rstd,
                               void* du, float* partial, int n_partial_blocks, int T, int Dd, void* stream) {
    DIC_REQUIRE(Dd == D && T > 0 && n_partial_blocks > 0, "dic_gelu_ln_bwd: D must be 768");
    dim3 grid(n_partial_blocks), block(64 * LNB_WAVES);
    const size_t lds = LNB_WAVES * 3 * D * sizeof(float);            // 72 KB: above the 64 KB default cap of dynamic LDS
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gelu_ln_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gelu_ln_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gelu_ln_bwd_kernel<bf16_t, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == (DIC_BF16 | DIC_U_F32)) {             // fp32 u, bf16 gradient (the bf16 engines' default)
        hipLaunchKernelGGL((gelu_ln_bwd_kernel<bf16_t, float>), grid, block, lds, st, dx_out, (const float*)u, gamma, mean, rstd, (bf16_t*)du, partial, T);
        DIC_CHECK_LAUNCH();
        return 0;
    }
    dtype &= ~DIC_U_F32;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(gelu_ln_bwd_kernel<bf16_t>, grid, block, lds, st, dx_out, (const bf16_t*)u, gamma, mean, rstd, (bf16_t*)du, partial, T),
               hipLaunchKernelGGL(gelu_ln_bwd_kernel<float>, grid, block, lds, st, dx_out, (const float*)u, gamma, mean, rstd, (float*)du, partial, T));
    DIC_CHECK_LAUNCH();
    return 0;
}
