// HBM-bound helpers of the hot path: embedding gather (K1), q_sample (K2), embedding losses (K13), the
// rounding-loss combine (K12), CFG mix (K10), column reductions for bias / LayerNorm / embedding gradients,
// and the fused AdamW (K15).  All are streaming kernels: 16-byte accesses per lane, grid-stride, no reuse.
#include "common.h"
#include "../../include/dic_hip.h"
#include <string.h>

static thread_local char g_err[256] = "";
extern "C" __attribute__((visibility("hidden"))) void dic_set_error(const char* msg) { strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1); g_err[sizeof(g_err) - 1] = 0; }
extern "C" const char* dic_last_error(void) { return g_err; }
extern "C" int dic_version(void) { return DIC_HIP_VERSION; }

// ---- step context (see common.h): process-global, set by the code that captures a training step into a hipGraph --------------------
static DicStepCtx g_step_ctx = {nullptr, 0, 0, nullptr};
DicStepCtx dic_step_ctx() { return g_step_ctx; }
extern "C" int dic_step_ctx_set(const int64_t* ctr, int64_t ctr0, uint64_t stride_noise, const float* adam_table) {
    g_step_ctx = DicStepCtx{(const long long*)ctr, (long long)ctr0, (unsigned long long)stride_noise, adam_table};
    return 0;
}
__global__ void step_advance_kernel(long long* ctr) { if (threadIdx.x == 0) ctr[0] += 1; }
extern "C" int dic_step_advance(int64_t* ctr, void* stream) {
    DIC_REQUIRE(ctr != nullptr, "dic_step_advance: counter missing");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)ctr);
    DIC_CHECK_LAUNCH();
    return 0;
}

static inline int grid_for(long long work_items, int per_block, int cap = 4096) {
    long long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ------------------------------------------------------------------------------------------------ K1 embedding gather
// ref CLIP-DDPM.py:459  x_0 = model.embedding(ids).  One wave per token row, 16 B per lane.
// nn.Embedding raises on an id outside [0, V); a kernel cannot raise, so it REPORTS: the row is zero-filled and `err` (optional, two
// ints the caller zeroed: device memory or device-visible pinned host memory) gets err[0] += 1 and err[1] = max(err[1], position + 1).
// The caller turns a non-zero err[0] into its IndexError at its next synchronisation point.  Nothing is clamped.
__global__ void embed_gather_kernel(const int64_t* ids, const float* E, float* out, int n_tokens, int D, int V, int* err) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < n_tokens; row += gridDim.x * wpb) {
        const long long id = ids[row];
        const bool bad = id < 0 || id >= V;
        if (bad && err && lane == 0) {
            __hip_atomic_fetch_add(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_fetch_max(err + 1, row + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const f32x4* src = (const f32x4*)(E + (size_t)(bad ? 0 : id) * D);
        f32x4* dst = (f32x4*)(out + (size_t)row * D);
        for (int c = lane; c < D / 4; c += 64) dst[c] = bad ? f32x4{0.f, 0.f, 0.f, 0.f} : src[c];
    }
}
extern "C" int dic_embed_gather(const int64_t* ids, const float* E, float* out, int n_tokens, int D, int V, int* err, void* stream) {
    DIC_REQUIRE(D % 4 == 0 && n_tokens > 0, "dic_embed_gather: D must be a multiple of 4");
    hipLaunchKernelGGL(embed_gather_kernel, dim3(grid_for(n_tokens, 4)), dim3(256), 0, (hipStream_t)stream, ids, E, out, n_tokens, D, V, err);
    DIC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ K2 q_sample
// ref CLIP-DDPM.py:356-362.  Each thread owns 4 consecutive elements of one [B][LD] slab position, draws (or
// reads) its noise ONCE and writes the S noised copies (s-major output), so x0/eps are read once, not S times.
__global__ void qsample_kernel(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac, float* out,
                               float* noise_out, int S, long long BLD, int step_tot, SeedArg seed_) {
#pragma clang fp contract(off)   // the reference rounds a*x, eps*b and their sum separately: no FMA contraction here
    const unsigned long long seed = seed_.resolve();
    const long long n4 = BLD >> 2;
    for (long long i4 = blockIdx.x * (long long)blockDim.x + threadIdx.x; i4 < n4; i4 += (long long)gridDim.x * blockDim.x) {
        f32x4 x = *(const f32x4*)(x0 + i4 * 4);
        f32x4 e;
        if (noise) {
            e = *(const f32x4*)(noise + i4 * 4);
        } else {
            uint4 r = rng4(seed, (unsigned long long)i4);
            // Box-Muller on two uniform pairs
            float u1 = fmaxf(u01(r.x), 5.9604645e-8f), u2 = u01(r.y), u3 = fmaxf(u01(r.z), 5.9604645e-8f), u4 = u01(r.w);
            float r1 = sqrtf(-2.f * __logf(u1)), r2 = sqrtf(-2.f * __logf(u3));
            float s1, c1, s2, c2;
            __sincosf(6.283185307179586f * u2, &s1, &c1);
            __sincosf(6.283185307179586f * u4, &s2, &c2);
            e = f32x4{r1 * c1, r1 * s1, r2 * c2, r2 * s2};
        }
        if (noise_out) *(f32x4*)(noise_out + i4 * 4) = e;
        for (int s = 0; s < S; ++s) {
            long long ts = t[s];
            ts = ts < 0 ? 0 : (ts >= step_tot ? step_tot - 1 : ts);
            const float ca = sqrt_ac[ts], cb = sqrt_1mac[ts];    // tables built on the host exactly as ref :360-361 builds them
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __fadd_rn(__fmul_rn(ca, x[k]), __fmul_rn(e[k], cb));   // un-fused, same op order as ref :360-362 => bit-exact
            *(f32x4*)(out + (size_t)s * BLD + i4 * 4) = o;
        }
    }
}
extern "C" int dic_qsample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac, float* out,
                           float* noise_out, int S, int B, int LD, int step_tot, uint64_t seed, void* stream) {
    long long BLD = (long long)B * LD;
    DIC_REQUIRE(BLD % 4 == 0 && S > 0, "dic_qsample: B*L*D must be a multiple of 4");
    hipLaunchKernelGGL(qsample_kernel, dim3(grid_for(BLD / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, x0, noise, t,
                       sqrt_ac, sqrt_1mac, out, noise_out, S, BLD, step_tot, make_seed(seed, dic_step_ctx().stride_noise));
    DIC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ K13 embedding losses
// ref CLIP-DDPM.py:77-87, 418, 428.  One workgroup per sequence: reduce |d| or d^2 over the L*D elements of rows
// t<L, then (training) write the gradient  grad_scale[n] * (sign(d) | d / norm)  into dx_out and zero the CLIP rows.
template <typename T>
__global__ __launch_bounds__(256) void emb_loss_kernel(int kind, const float* x_out, const float* target, int tgt_rows,
                                                        float* per_seq, float* dx_out, const float* grad_scale, T* xr,
                                                        int L, int Tk, int D) {
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* xo = x_out + (size_t)n * Tk * D;
    const float* tg = target + (size_t)(n % tgt_rows) * L * D;
    const int n4 = L * D / 4;
    const bool l2 = kind >= 2;
    float acc = 0.f;
    for (int i = tid; i < n4; i += 256) {
        f32x4 a = *(const f32x4*)(xo + i * 4), b = *(const f32x4*)(tg + i * 4);
        if (xr) Elem<T>::st4(xr + (size_t)n * L * D + i * 4, a);
#pragma unroll
        for (int k = 0; k < 4; ++k) { float d = a[k] - b[k]; acc += l2 ? d * d : fabsf(d); }
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    float tot = red[0] + red[1] + red[2] + red[3];
    float val = l2 ? sqrtf(tot) : tot;
    if (tid == 0) per_seq[n] = val;
    if (dx_out) {
        const float gs = grad_scale[n];
        const float inv = l2 ? (val > 0.f ? gs / val : 0.f) : gs;
        float* dx = dx_out + (size_t)n * Tk * D;
        for (int i = tid; i < n4; i += 256) {
            f32x4 a = *(const f32x4*)(xo + i * 4), b = *(const f32x4*)(tg + i * 4), g;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float d = a[k] - b[k];
                g[k] = l2 ? d * inv : (d > 0.f ? inv : (d < 0.f ? -inv : 0.f));
            }
            *(f32x4*)(dx + i * 4) = g;
        }
        for (int i = n4 + tid; i < Tk * D / 4; i += 256) *(f32x4*)(dx + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
extern "C" int dic_emb_loss(int dtype, int kind, const float* x_out, const float* target, int tgt_rows, float* per_seq,
                            float* dx_out, const float* grad_scale, void* xr, int N, int L, int Tk, int D, void* stream) {
    DIC_REQUIRE(D % 4 == 0 && N > 0 && tgt_rows > 0 && kind >= 0 && kind < 4, "dic_emb_loss: bad arguments");
    if (dtype == DIC_BF16)
        hipLaunchKernelGGL(emb_loss_kernel<bf16_t>, dim3(N), dim3(256), 0, (hipStream_t)stream, kind, x_out, target, tgt_rows, per_seq, dx_out, grad_scale, (bf16_t*)xr, L, Tk, D);
    else
        hipLaunchKernelGGL(emb_loss_kernel<float>, dim3(N), dim3(256), 0, (hipStream_t)stream, kind, x_out, target, tgt_rows, per_seq, dx_out, grad_scale, (float*)xr, L, Tk, D);
    DIC_CHECK_LAUNCH();
    return 0;
}

__global__ void add_rows_kernel(float* dx_out, const float* dxr, long long n4_total, int LD4, int TkD4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4_total; i += (long long)gridDim.x * blockDim.x) {
        long long n = i / LD4, r = i - n * LD4;
        f32x4* d = (f32x4*)dx_out + n * TkD4 + r;
        *d = *d + ((const f32x4*)dxr)[i];
    }
}
__global__ void add_rows_scaled_kernel(float* dx_out, const float* dxr, const float* inv_z, float scale, long long n4_total, int LD4, int TkD4, int D4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4_total; i += (long long)gridDim.x * blockDim.x) {
        long long n = i / LD4, r = i - n * LD4;
        const float f = inv_z[i / D4] * scale;
        f32x4* d = (f32x4*)dx_out + n * TkD4 + r;
        const f32x4 v = ((const f32x4*)dxr)[i];
        *d = *d + f32x4{v[0] * f, v[1] * f, v[2] * f, v[3] * f};
    }
}
extern "C" int dic_add_rows_scaled(float* dx_out, const float* dxr, const float* inv_z, float scale, int N, int L, int Tk, int D, void* stream) {
    DIC_REQUIRE(D % 4 == 0 && N > 0 && inv_z, "dic_add_rows_scaled: bad arguments");
    long long n4 = (long long)N * L * D / 4;
    hipLaunchKernelGGL(add_rows_scaled_kernel, dim3(grid_for(n4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, dx_out, dxr, inv_z, scale, n4, L * D / 4, Tk * D / 4, D / 4);
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_add_rows(float* dx_out, const float* dxr, int N, int L, int Tk, int D, void* stream) {
    long long n4 = (long long)N * L * D / 4;
    hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for(n4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, dx_out, dxr, n4, L * D / 4, Tk * D / 4);
    DIC_CHECK_LAUNCH();
    return 0;
}

// out[0] = scale_a * sum in[0:n_a], out[1] = scale_b * sum in[n_a:n], out[2] = out[0]+out[1]  (one workgroup, fixed order)
__global__ __launch_bounds__(256) void seg_sum_kernel(const float* in, int n, int n_a, float sa, float sb, float* out2, const float* carry,
                                                      const long long* ctr, long long ctr0) {
    // replayed step number k writes result slot k of the ring the captured step's slot starts (8 floats per slot, diffusion.LOSS_RING)
    const long long slot = step_delta(ctr, ctr0) * 8;
    out2 += slot;
    if (carry) carry += slot;
    __shared__ double red[2][4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { float v = in[i]; if (i < n_a) a += v; else b += v; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a_ = (float)((red[0][0] + red[0][1] + red[0][2] + red[0][3]) * sa);
        float b_ = (float)((red[1][0] + red[1][1] + red[1][2] + red[1][3]) * sb);
        out2[0] = a_; out2[1] = b_; out2[2] = a_ + b_;
        out2[3] = (a_ + b_) + (carry ? *carry : 0.f);        // running total over several calls (the step's l = x_t + x_1 + prob, ref :481)
    }
}
extern "C" int dic_seg_sum(const float* in, int n, int n_a, float scale_a, float scale_b, float* out2, const float* carry, void* stream) {
    const DicStepCtx c = dic_step_ctx();
    hipLaunchKernelGGL(seg_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, in, n, n_a, scale_a, scale_b, out2, carry, c.ctr, c.ctr0);
    DIC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ step inputs of the stacked encoder batch
// What `loss` (ref :406-415, 426, 434-437) assembles with repeat / hstack / cat every step -- the CLIP rows repeated over the S timesteps,
// the key-padding masks with their constant CLIP columns, the repeated target ids, the per-sequence loss scales -- written by ONE launch
// straight into the encoder's and the rounding head's input buffers.  Batch layout: rows [0, S*B) = x_t (s-major: row s*B + b), then B rows x_1.
__global__ __launch_bounds__(256) void step_prep_kernel(const float* img, const float* txt, const int64_t* mask, const int64_t* ids, int S, int B, int L,
                                                        int Tk, float* img_in, float* txt_in, uint8_t* kmask, uint8_t* addtxt, int64_t* tgt,
                                                        float* gscale, float sa, float sb) {
    const int N = S * B + B, Nt = S * B;
    const long long n_clip = (long long)N * 128;                 // float4 items per CLIP buffer
    const long long total = n_clip + (long long)N * Tk + (long long)N * L + N;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        if (i < n_clip) {
            const int n = (int)(i >> 7), c = (int)(i & 127), b = n < Nt ? n % B : n - Nt;
            ((f32x4*)img_in)[i] = ((const f32x4*)img)[(size_t)b * 128 + c];
            if (txt_in) ((f32x4*)txt_in)[i] = ((const f32x4*)txt)[(size_t)b * 128 + c];
            continue;
        }
        long long j = i - n_clip;
        if (j < (long long)N * Tk) {
            const int n = (int)(j / Tk), c = (int)(j - (long long)n * Tk), b = n < Nt ? n % B : n - Nt;
            kmask[j] = c < L ? (mask[(size_t)b * L + c] != 0) : (c == L ? 1 : 0);      // token keys | image row visible | text row masked
            continue;
        }
        j -= (long long)N * Tk;
        if (j < (long long)N * L) {
            if (tgt) { const int n = (int)(j / L), c = (int)(j - (long long)n * L), b = n < Nt ? n % B : n - Nt; tgt[j] = ids[(size_t)b * L + c]; }
            continue;
        }
        j -= (long long)N * L;
        addtxt[j] = 0;
        if (gscale) gscale[j] = j < Nt ? sa : sb;
    }
}
extern "C" int dic_step_prep(const float* img, const float* txt, const int64_t* mask, const int64_t* ids, int S, int B, int L, int Tk,
                             float* img_in, float* txt_in, uint8_t* kmask, uint8_t* addtxt, int64_t* tgt, float* gscale, float scale_a,
                             float scale_b, void* stream) {
    DIC_REQUIRE(S > 0 && B > 0 && L > 0 && Tk >= L && Tk <= L + 2, "dic_step_prep: Tk must be L, L+1 or L+2");
    const long long total = (long long)(S * B + B) * (128 + Tk + L + 1);
    hipLaunchKernelGGL(step_prep_kernel, dim3(grid_for(total, 256, 2048)), dim3(256), 0, (hipStream_t)stream, img, txt, mask, ids, S, B, L, Tk, img_in,
                       txt_in, kmask, addtxt, tgt, gscale, scale_a, scale_b);
    DIC_CHECK_LAUNCH();
    return 0;
}
// The same for a step WITH classifier-free guidance (ref :406-415, 313-317): stacked batch [S*B x_t rows | Ng guided copies | B x_1 rows], where guided
// copy i repeats x_t row gi[i] with the text key unmasked (concat fusion) / the text projection added ("add" fusion).  The host draws the guidance
// mask (it needs Ng for every launch shape), uploads the row list once, and this launch does everything the reference's cat / repeat / index ops do:
// CLIP rows, key masks, add_txt flags, target ids and loss scales (the rounding head and the embedding losses see only the x_t and x_1 rows), the
// copy of the guided rows' noisy inputs inside xin, and the zero fill of their dx_out rows (cfg_mix_bwd accumulates into them).
__global__ __launch_bounds__(256) void cfg_prep_kernel(const float* img, const float* txt, const int64_t* mask, const int64_t* ids, const int64_t* gi, int S,
                                                       int B, int L, int Tk, int Ng, float* img_in, float* txt_in, uint8_t* kmask, uint8_t* addtxt,
                                                       int64_t* tgt, float* gscale, float sa, float sb, float* xin, float* dx, int D4) {
    const int Nt = S * B, N = Nt + Ng + B;
    const long long n_clip = (long long)N * 128, n_km = (long long)N * Tk, n_tgt = tgt ? (long long)(Nt + B) * L : 0, n_seq = N;
    const long long n_x = (long long)Ng * L * D4, n_dx = dx ? (long long)Ng * Tk * D4 : 0;
    const long long total = n_clip + n_km + n_tgt + n_seq + n_x + n_dx;
    auto src_b = [&](int n, bool& guided) {          // batch item that row n of the stacked batch belongs to
        guided = n >= Nt && n < Nt + Ng;
        return n < Nt ? n % B : (guided ? (int)(gi[n - Nt] % B) : n - Nt - Ng);
    };
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long j = i;
        bool gd;
        if (j < n_clip) {
            const int n = (int)(j >> 7), c = (int)(j & 127), b = src_b(n, gd);
            ((f32x4*)img_in)[j] = ((const f32x4*)img)[(size_t)b * 128 + c];
            ((f32x4*)txt_in)[j] = ((const f32x4*)txt)[(size_t)b * 128 + c];
            continue;
        }
        j -= n_clip;
        if (j < n_km) {
            const int n = (int)(j / Tk), c = (int)(j - (long long)n * Tk), b = src_b(n, gd);
            kmask[j] = c < L ? (mask[(size_t)b * L + c] != 0) : (c == L ? 1 : (gd ? 1 : 0));   // token keys | image row | text row: guided copies only
            continue;
        }
        j -= n_km;
        if (j < n_tgt) {
            const int n = (int)(j / L), c = (int)(j - (long long)n * L), b = n < Nt ? n % B : n - Nt;
            tgt[j] = ids[(size_t)b * L + c];
            continue;
        }
        j -= n_tgt;
        if (j < n_seq) {
            const int n = (int)j;
            addtxt[n] = n >= Nt && n < Nt + Ng;
            if (gscale && n < Nt + B) gscale[n] = n < Nt ? sa : sb;
            continue;
        }
        j -= n_seq;
        if (j < n_x) {
            const long long row4 = (long long)L * D4;
            const int g = (int)(j / row4);
            ((f32x4*)xin)[(size_t)(Nt + g) * row4 + (j - g * row4)] = ((const f32x4*)xin)[(size_t)gi[g] * row4 + (j - g * row4)];
            continue;
        }
        j -= n_x;
        ((f32x4*)dx)[(size_t)Nt * Tk * D4 + j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
extern "C" int dic_cfg_prep(const float* img, const float* txt, const int64_t* mask, const int64_t* ids, const int64_t* gi, int S, int B, int L, int Tk,
                            int Ng, int D, float* img_in, float* txt_in, uint8_t* kmask, uint8_t* addtxt, int64_t* tgt, float* gscale, float scale_a,
                            float scale_b, float* xin, float* dx, void* stream) {
    DIC_REQUIRE(S > 0 && B > 0 && L > 0 && (Tk == L || Tk == L + 2) && Ng >= 0 && Ng <= S * B && D % 4 == 0, "dic_cfg_prep: Tk must be L or L+2, 0 <= Ng <= S*B");
    DIC_REQUIRE(Ng == 0 || gi != nullptr, "dic_cfg_prep: guided row list missing");
    const long long N = (long long)S * B + Ng + B;
    const long long total = N * (128 + Tk + 1) + (long long)(S * B + B) * L + (long long)Ng * (L + Tk) * (D / 4);
    hipLaunchKernelGGL(cfg_prep_kernel, dim3(grid_for(total, 256, 2048)), dim3(256), 0, (hipStream_t)stream, img, txt, mask, ids, gi, S, B, L, Tk, Ng, img_in,
                       txt_in, kmask, addtxt, tgt, gscale, scale_a, scale_b, xin, dx, D / 4);
    DIC_CHECK_LAUNCH();
    return 0;
}
// out[i] = uniform integer in [0, hi) from Philox4x32-10 keyed by (seed, i): the step's timestep vector (ref :460-461 torch.randint)
__global__ void randint_kernel(int64_t* out, int n, unsigned hi, SeedArg seed_) {
    const unsigned long long seed = seed_.resolve();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int64_t)(((unsigned long long)rng4(seed, (unsigned long long)i).x * hi) >> 32);
}
extern "C" int dic_randint(int64_t* out, int n, int hi, uint64_t seed, void* stream) {
    DIC_REQUIRE(n > 0 && hi > 0, "dic_randint: empty range");
    hipLaunchKernelGGL(randint_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, n, (unsigned)hi, make_seed(seed, DIC_STRIDE_T));
    DIC_CHECK_LAUNCH();
    return 0;
}
__global__ void zero_kernel(f32x4* p4, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) p4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}
extern "C" int dic_zero(void* p, int64_t nbytes, void* stream) {
    DIC_REQUIRE(((uintptr_t)p % 16) == 0 && nbytes % 16 == 0 && nbytes >= 0, "dic_zero: 16-byte aligned ranges only");
    if (nbytes == 0) return 0;
    hipLaunchKernelGGL(zero_kernel, dim3(grid_for(nbytes / 16, 256, 2048)), dim3(256), 0, (hipStream_t)stream, (f32x4*)p, (long long)(nbytes / 16));
    DIC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ K12 rounding combine
// One wave per row: merge the per-half-tile {max, sumexp, argmax} partials (ordered by column range, so the first
// partial holding the global max carries the lowest index -- torch.argmax's tie rule).
__global__ void ce_combine_kernel(const float* partial, const float* tgt_logit, int M, int np, float* lse, int64_t* argmax, float* nll) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int m = blockIdx.x * wpb + (threadIdx.x >> 6); m < M; m += gridDim.x * wpb) {
        const float4* pr = (const float4*)partial + (size_t)m * np;
        float mx = -INFINITY, sm = 0.f;
        int ix = 0x7fffffff;
        for (int i = lane; i < np; i += 64) {
            float4 q = pr[i];
            int qi = __float_as_int(q.z);
            if (q.x > mx) { sm = sm * __expf(mx - q.x) + q.y; mx = q.x; ix = qi; }
            else if (q.x > -INFINITY) { sm += q.y * __expf(q.x - mx); if (q.x == mx) ix = min(ix, qi); }
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            float mx2 = __shfl_xor(mx, o, 64), sm2 = __shfl_xor(sm, o, 64);
            int ix2 = __shfl_xor(ix, o, 64);
            float Mx = fmaxf(mx, mx2);
            float s1 = (mx == -INFINITY) ? 0.f : sm * __expf(mx - Mx);
            float s2 = (mx2 == -INFINITY) ? 0.f : sm2 * __expf(mx2 - Mx);
            ix = (mx > mx2) ? ix : (mx2 > mx) ? ix2 : min(ix, ix2);
            mx = Mx; sm = s1 + s2;
        }
        if (lane == 0) {
            float l = mx + logf(sm);
            lse[m] = l;
            argmax[m] = ix;
            if (nll) nll[m] = l - tgt_logit[m];
        }
    }
}
// ---- mean-centred rounding-head input (dic_head_center).  logits = x W^T is evaluated as (x - xbar) W^T + xbar W^T: the row-common part in fp32
// (one 768 x V matrix-vector product per step, handed to the head GEMM as its bias), only the deviations through bf16.  Why: an early-training
// denoiser predicts nearly the same vector for every row (measured after 200 steps: |xbar| = 27, rms |x - xbar| = 0.009), so the bf16 rounding
// error of x is the SAME for every row and the batch-mean loss does not average it out (-2.9e-4 on the rounding loss against the fp32 head on
// identical encoder outputs; centred: 7e-10 -- profiles/r04_trained_gap_split.txt).  The deviations' rounding errors are independent again.
constexpr int HC_BLOCKS = 1024;
__device__ __forceinline__ const float* head_row(const float* xa, int na, const float* xb, int L, int Tk, int D, long long r) {
    const long long n = r / L;
    const int t = (int)(r - n * L);
    return (n < na ? xa + n * (long long)Tk * D : xb + (n - na) * (long long)Tk * D) + (long long)t * D;
}
// stage 1: 1024 blocks x 192 lanes (one float4 column each), four independent row loads in flight; stage 2 is colsum_small over the 1024 partial rows
__global__ __launch_bounds__(192) void head_colsum_kernel(const float* xa, int na, const float* xb, int nb, int L, int Tk, int D, float* ws) {
    const long long rows = (long long)(na + nb) * L;
    const long long per = (rows + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    f32x4 a{0.f, 0.f, 0.f, 0.f};
    long long r = r0;
    for (; r + 4 <= r1; r += 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(head_row(xa, na, xb, L, Tk, D, r + u) + threadIdx.x * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) a += v[u];
    }
    for (; r < r1; ++r) a += *(const f32x4*)(head_row(xa, na, xb, L, Tk, D, r) + threadIdx.x * 4);
    *(f32x4*)(ws + (size_t)blockIdx.x * D + threadIdx.x * 4) = a;
}
__global__ void head_bias_kernel(const float* W, int Vpad, int D, const float* xsum, float inv_rows, float* cvec) {    // one wave per vocabulary row
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (v >= Vpad) return;
    float a = 0.f;
#pragma unroll
    for (int d = lane * 4; d < 768; d += 256) {
        const f32x4 w = *(const f32x4*)(W + (size_t)v * D + d), x = *(const f32x4*)(xsum + d) * inv_rows;
        a = __builtin_fmaf(w[0], x[0], a); a = __builtin_fmaf(w[1], x[1], a); a = __builtin_fmaf(w[2], x[2], a); a = __builtin_fmaf(w[3], x[3], a);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) cvec[v] = a;
}
__global__ void head_xr_kernel(const float* xa, int na, const float* xb, int nb, int L, int Tk, int D, const float* xsum, float inv_rows, bf16_t* xr,
                               float* xbar) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const long long rows = (long long)(na + nb) * L;
    f32x4 m[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) m[c] = *(const f32x4*)(xsum + c * 256 + lane * 4) * inv_rows;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
#pragma unroll
        for (int c = 0; c < 3; ++c) *(f32x4*)(xbar + c * 256 + lane * 4) = m[c];
    }
    for (long long r = (long long)blockIdx.x * wpb + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * wpb) {
        const float* x = head_row(xa, na, xb, L, Tk, D, r);
        f32x4 v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = *(const f32x4*)(x + c * 256 + lane * 4);
#pragma unroll
        for (int c = 0; c < 3; ++c) Elem<bf16_t>::st4(xr + (size_t)r * D + c * 256 + lane * 4, v[c] - m[c]);
    }
}
__global__ __launch_bounds__(64 * 16) void colsum_small(const float* in, int rows, int cols, int ld, float* out, int accumulate, const float* in2, float* out2);
extern "C" size_t dic_head_center_ws_bytes(int D) { return ((size_t)HC_BLOCKS * D + D) * sizeof(float); }
extern "C" int dic_head_center(const float* x_a, int n_a, const float* x_b, int n_b, int L, int Tk, int D, const float* W32, int Vpad, float* ws,
                               float* xbar, float* cvec, void* xr, void* stream) {
    DIC_REQUIRE(D == 768 && n_a >= 0 && n_b >= 0 && n_a + n_b > 0 && L > 0 && Tk >= L && x_a && W32 && ws && xbar && cvec && xr && (n_b == 0 || x_b),
                "dic_head_center: D must be 768; x_a / W32 / ws / xbar / cvec / xr must be given");
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)(n_a + n_b) * L;
    float* xsum = ws + (size_t)HC_BLOCKS * D;
    const float inv = 1.0f / (float)rows;
    hipLaunchKernelGGL(head_colsum_kernel, dim3(HC_BLOCKS), dim3(192), 0, st, x_a, n_a, x_b, n_b, L, Tk, D, ws);
    hipLaunchKernelGGL(colsum_small, dim3((D + 255) / 256), dim3(64 * 16), 0, st, (const float*)ws, HC_BLOCKS, D, D, xsum, 0, (const float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(head_bias_kernel, dim3((Vpad + 3) / 4), dim3(256), 0, st, W32, Vpad, D, (const float*)xsum, inv, cvec);
    hipLaunchKernelGGL(head_xr_kernel, dim3(grid_for(rows, 4)), dim3(256), 0, st, x_a, n_a, x_b, n_b, L, Tk, D, (const float*)xsum, inv, (bf16_t*)xr, xbar);
    DIC_CHECK_LAUNCH();
    return 0;
}

// ---- rounding loss, training form (include/dic_hip.h: dic_ce_target_logit -> dic_gemm(CE_EXP) -> dic_ce_exp_combine)
// one wave per row: t = <xr[m], W[tgt[m]]>, c = t + shift
__global__ void ce_target_logit_kernel(const bf16_t* xr, const bf16_t* W, const int64_t* tgt, int M, int V, int D, float shift, float* t_out, float* c_out,
                                       const float* col_bias) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int m = blockIdx.x * wpb + (threadIdx.x >> 6); m < M; m += gridDim.x * wpb) {
        const long long tg = tgt[m];
        float a = 0.f;
        if (tg >= 0 && tg < V) {
            const bf16_t* x = xr + (size_t)m * D;
            const bf16_t* w = W + (size_t)tg * D;
            for (int d = lane * 8; d < D; d += 512) {
                const bf16x8 xv = *(const bf16x8*)(x + d), wv = *(const bf16x8*)(w + d);
#pragma unroll
                for (int k = 0; k < 8; ++k) a = __builtin_fmaf((float)xv[k], (float)wv[k], a);
            }
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) {
            if (col_bias && tg >= 0 && tg < V) a += col_bias[tg];          // (mean-centred head input: the row-common part of the logit)
            t_out[m] = a; c_out[m] = a + shift;
        }
    }
}
extern "C" int dic_ce_target_logit(const void* xr, const void* W, const int64_t* tgt, int M, int V, int D, float shift, float* t, float* c,
                                   const float* col_bias, void* stream) {
    DIC_REQUIRE(M > 0 && D % 8 == 0 && xr && W && tgt && t && c, "dic_ce_target_logit: bad arguments");
    hipLaunchKernelGGL(ce_target_logit_kernel, dim3(grid_for(M, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)xr, (const bf16_t*)W, tgt, M, V, D, shift, t, c,
                       col_bias);
    DIC_CHECK_LAUNCH();
    return 0;
}
// one wave per row: Z = sum of the slab sums (lane-strided, then a fixed xor tree: deterministic), the row's statistics, and the target entry of E
__global__ void ce_exp_combine_kernel(const float* partial, int np, const float* c, const float* tgt_logit, const int64_t* tgt, int M, int V, bf16_t* E,
                                      int ldE, float* lse, float* nll, float* inv_z) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int m = blockIdx.x * wpb + (threadIdx.x >> 6); m < M; m += gridDim.x * wpb) {
        const float* pr = partial + (size_t)m * np;
        float z = 0.f;
        for (int i = lane; i < np; i += 64) z += pr[i];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) z += __shfl_xor(z, o, 64);
        if (lane == 0) {
            const float cm = c[m], l = logf(z) + cm;
            const long long tg = tgt[m];
            const bool ok = tg >= 0 && tg < V;
            const float tl = ok ? tgt_logit[m] : 0.f;
            lse[m] = l;
            if (nll) nll[m] = l - tl;
            inv_z[m] = 1.0f / z;
            // (the same fma + v_exp the epilogue applied to this logit: bit-identical to the term inside z, so the difference is exactly the
            //  other columns' sum)
            if (ok) E[(size_t)m * ldE + tg] = f2bf(__builtin_amdgcn_exp2f(__builtin_fmaf(tl, 1.4426950408889634f, -(cm * 1.4426950408889634f))) - z);
        }
    }
}
extern "C" int dic_ce_exp_combine(const float* partial, int n_partials, const float* c, const float* tgt_logit, const int64_t* tgt, int M, int V,
                                  void* E, int ldE, float* lse, float* nll, float* inv_z, void* stream) {
    DIC_REQUIRE(M > 0 && n_partials > 0 && partial && c && tgt_logit && tgt && E && lse && inv_z && ldE >= V, "dic_ce_exp_combine: bad arguments");
    hipLaunchKernelGGL(ce_exp_combine_kernel, dim3(grid_for(M, 4)), dim3(256), 0, (hipStream_t)stream, partial, n_partials, c, tgt_logit, tgt, M, V, (bf16_t*)E, ldE, lse, nll, inv_z);
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_ce_combine(const float* partial, const float* tgt_logit, int M, int n_partials, float* lse, int64_t* argmax,
                              float* nll, void* stream) {
    hipLaunchKernelGGL(ce_combine_kernel, dim3(grid_for(M, 4)), dim3(256), 0, (hipStream_t)stream, partial, tgt_logit, M, n_partials, lse, argmax, nll);
    DIC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ K10 classifier-free-guidance mix
__global__ void cfg_mix_fwd_kernel(float* x_out, const float* g_out, const int64_t* idx, long long n4, int row4, float w) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        long long r = i / row4, c = i - r * row4;
        f32x4* x = (f32x4*)x_out + idx[r] * row4 + c;
        f32x4 g = ((const f32x4*)g_out)[i];
        *x = (1.0f + w) * g - w * (*x);            // ref :315-317
    }
}
__global__ void cfg_mix_bwd_kernel(float* dx_out, float* dg_out, const int64_t* idx, long long n4, int row4, float w) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        long long r = i / row4, c = i - r * row4;
        f32x4* x = (f32x4*)dx_out + idx[r] * row4 + c;
        f32x4 d = *x;
        ((f32x4*)dg_out)[i] = (1.0f + w) * d;
        *x = -w * d;
    }
}
extern "C" int dic_cfg_mix_fwd(float* x_out, const float* g_out, const int64_t* idx, int n_g, int row_elems, float w, void* stream) {
    long long n4 = (long long)n_g * row_elems / 4;
    hipLaunchKernelGGL(cfg_mix_fwd_kernel, dim3(grid_for(n4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, x_out, g_out, idx, n4, row_elems / 4, w);
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_cfg_mix_bwd(float* dx_out, float* dg_out, const int64_t* idx, int n_g, int row_elems, float w, void* stream) {
    long long n4 = (long long)n_g * row_elems / 4;
    hipLaunchKernelGGL(cfg_mix_bwd_kernel, dim3(grid_for(n4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, dx_out, dg_out, idx, n4, row_elems / 4, w);
    DIC_CHECK_LAUNCH();
    return 0;
}

// "add" fusion backward (ref :306-307): the projected CLIP row was broadcast over the L sequence rows, so its gradient
// is the sum over t; out_txt gets the same sum only where the text row was added (classifier-free-guided rows).
__global__ void seq_sum_kernel(const float* in, const uint8_t* flags, float* out_all, float* out_flag, int N, int L, int D4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (long long)N * D4; i += (long long)gridDim.x * blockDim.x) {
        int n = (int)(i / D4), c = (int)(i - (long long)n * D4);
        f32x4 acc{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < L; ++t) acc += ((const f32x4*)in)[((size_t)n * L + t) * D4 + c];
        ((f32x4*)out_all)[i] = acc;
        if (out_flag) ((f32x4*)out_flag)[i] = (flags && flags[n]) ? acc : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
extern "C" int dic_seq_sum(const float* in, const uint8_t* flags, float* out_all, float* out_flag, int N, int L, int D, void* stream) {
    DIC_REQUIRE(D % 4 == 0 && N > 0, "dic_seq_sum: D must be a multiple of 4");
    hipLaunchKernelGGL(seq_sum_kernel, dim3(grid_for((long long)N * D / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, in, flags, out_all, out_flag, N, L, D / 4);
    DIC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ column sums
// Stage 1: grid (col groups of 256, slabs); the 4 waves of a block take rows slab*4+w, +4*nslab, ...; LDS-reduce the
// 4 waves; write ws[slab][cols].  Stage 2: sum the slabs in fixed order.  Deterministic.
template <typename T>
__global__ __launch_bounds__(256) void colsum_stage1(const T* in, int rows, int cols, int ld, float* ws, int nslab) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    f32x4 acc{0.f, 0.f, 0.f, 0.f};
    if (c < cols)
        for (int r = blockIdx.y * 4 + w; r < rows; r += nslab * 4) acc += Elem<T>::ld4(in + (size_t)r * ld + c);
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0 && c < cols) *(f32x4*)(ws + (size_t)blockIdx.y * cols + c) = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}
__global__ void colsum_stage2(const float* ws, int nslab, int cols, float* out, int accumulate) {
    int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (c >= cols) return;
    f32x4 acc{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nslab; ++s) acc += *(const f32x4*)(ws + (size_t)s * cols + c);
    if (accumulate) acc += *(const f32x4*)(out + c);
    *(f32x4*)(out + c) = acc;
}
// few rows (the per-block partial rows of the LayerNorm backward kernels, split-K style folds): one launch, 16 waves per
// 256 columns split the rows (the kernel is a chain of memory round trips: rows / (16 waves x 8 loads in flight) of them),
// LDS fold in fixed order.
constexpr int CS_WAVES = 16;
__global__ __launch_bounds__(64 * CS_WAVES) void colsum_small(const float* in, int rows, int cols, int ld, float* out, int accumulate,
                                                              const float* in2 = nullptr, float* out2 = nullptr) {
    if (blockIdx.y == 1) { in = in2; out = out2; }          // dic_colsum_pair: two independent (in, out) problems of one shape in one launch
    __shared__ f32x4 red[CS_WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    f32x4 acc{0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
        int r = w;
        for (; r + 7 * CS_WAVES < rows; r += 8 * CS_WAVES) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(in + (size_t)(r + CS_WAVES * u) * ld + c);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; r < rows; r += CS_WAVES) acc += *(const f32x4*)(in + (size_t)r * ld + c);
    }
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0 && c < cols) {
        f32x4 v = red[0][lane];
#pragma unroll
        for (int q = 1; q < CS_WAVES; ++q) v += red[q][lane];
        if (accumulate) v += *(const f32x4*)(out + c);
        *(f32x4*)(out + c) = v;
    }
}
// ------------------------------------------------------------------------------------------------ TRAIN_EMBEDDING ablation
// ref :98-102, 238-243, 459-468: with a learned embedding x_0 = E[ids] carries gradient -- through q_sample into every noised copy
// (x_t[s] = sqrt_ac[t_s] x_0 + ..., x_1 = sqrt_ac[1] x_0 + ...) and as the TARGET of both embedding losses.
//   dx0[b][l][c] = sum_s ( sqrt_ac[t_s] * dxin[s*B+b] - k_s * g[s*B+b] ) + sqrt_ac[1] * dxin[x1_row0+b] - g[x1_row0+b]     ([l][c] implied)
// dxin = gradient wrt the stacked 16-d encoder input, g = gradient of the embedding losses wrt the model's 16-d output
// (both [rows][Tk][C], rows t < L used; the x_1 sequences start at row x1_row0 -- guided copies may sit in between).
// k_s = 1 when the x_t loss targets x_0 (x_0 prediction), sqrt_ac[t_next_s] when it targets the noised x_{t_next} (ref :364-380).
__global__ void te_dx0_kernel(const float* dxin, const float* g, const float* sqrt_ac, const int64_t* t, const int64_t* t_next, int S, int B,
                              int L, int Tk, int C, int step_tot, int x1_row0, float* dx0) {
    const int n = B * L * C;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c = i % C, l = (i / C) % L, b = i / (C * L);
        float acc = 0.f;
        for (int s_ = 0; s_ <= S; ++s_) {
            long long ts = s_ < S ? t[s_] : 1;
            ts = ts < 0 ? 0 : (ts >= step_tot ? step_tot - 1 : ts);
            float kt = 1.0f;
            if (s_ < S && t_next) {
                long long tn = t_next[s_];
                tn = tn < 0 ? 0 : (tn >= step_tot ? step_tot - 1 : tn);
                kt = sqrt_ac[tn];
            }
            const size_t row = ((size_t)((s_ < S ? s_ * B : x1_row0) + b) * Tk + l) * C + c;
            acc += sqrt_ac[ts] * dxin[row] - kt * g[row];
        }
        dx0[i] = acc;
    }
}
extern "C" int dic_te_dx0(const float* dxin, const float* g, const float* sqrt_ac, const int64_t* t, const int64_t* t_next, int S, int B,
                          int L, int Tk, int C, int step_tot, int x1_row0, float* dx0, void* stream) {
    DIC_REQUIRE(S > 0 && B > 0 && L > 0 && Tk >= L && C > 0 && x1_row0 >= S * B, "dic_te_dx0: bad arguments");
    hipLaunchKernelGGL(te_dx0_kernel, dim3(grid_for((long long)B * L * C, 256, 1024)), dim3(256), 0, (hipStream_t)stream, dxin, g, sqrt_ac, t,
                       t_next, S, B, L, Tk, C, step_tot, x1_row0, dx0);
    DIC_CHECK_LAUNCH();
    return 0;
}
// dE[id][:] = sum of dx0 rows of the token positions holding `id` (nn.Embedding backward), in ascending position order: the
// caller passes the ids sorted (stable) with the permutation, so the sum has a fixed order and needs no atomics.  dE is
// zeroed by the caller; ids outside [0, V) are ignored.
__global__ void embed_scatter_kernel(const int64_t* sorted_ids, const int64_t* order, const float* dx0, int n_tokens, int C, int V, float* dE) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_tokens * C; i += gridDim.x * blockDim.x) {
        const int r = i / C, c = i - r * C;
        const int64_t id = sorted_ids[r];
        if (id < 0 || id >= V || (r > 0 && sorted_ids[r - 1] == id)) continue;        // only the first position of a run works
        float acc = 0.f;
        for (int q = r; q < n_tokens && sorted_ids[q] == id; ++q) acc += dx0[(size_t)order[q] * C + c];
        dE[(size_t)id * C + c] = acc;
    }
}
extern "C" int dic_embed_scatter(const int64_t* sorted_ids, const int64_t* order, const float* dx0, int n_tokens, int C, int V, float* dE,
                                 void* stream) {
    DIC_REQUIRE(n_tokens > 0 && C > 0 && V > 0, "dic_embed_scatter: bad arguments");
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(grid_for((long long)n_tokens * C, 256, 1024)), dim3(256), 0, (hipStream_t)stream, sorted_ids,
                       order, dx0, n_tokens, C, V, dE);
    DIC_CHECK_LAUNCH();
    return 0;
}

// ---- workspace-size queries (host only) ---------------------------------------------------------------------------------
extern "C" size_t dic_gemm_split_ws_bytes(int M, int N, int split_k, int with_colsum) {
    if (split_k <= 1) return 0;
    return (size_t)split_k * ((size_t)M * N + (with_colsum ? M : 0)) * sizeof(float);
}
extern "C" int dic_ce_n_partials(int N, int tile) { return tile == 256 ? 4 * ((N + 255) / 256) : 2 * ((N + 127) / 128); }
extern "C" size_t dic_ce_partial_bytes(int M, int N, int tile) { return (size_t)M * dic_ce_n_partials(N, tile) * 4 * sizeof(float); }
extern "C" size_t dic_colsum_ws_bytes(int in_dtype, int rows, int cols) {
    if (in_dtype == DIC_F32 && rows <= 1024) return 0;
    int nslab = (rows + 3) / 4;
    if (nslab > 64) nslab = 64;
    return (size_t)nslab * cols * sizeof(float);
}
extern "C" size_t dic_ln_partial_bytes(int n_partial_blocks, int n_vectors, int D) { return (size_t)n_partial_blocks * n_vectors * D * sizeof(float); }

// Two fp32 column sums of the same shape (rows <= 1024) in ONE launch: the two LayerNorm-backward partial buffers of an encoder layer
// ([gamma | beta | bias] gradients of sa_layer_norm + out_lin and of output_layer_norm + lin2) are folded together.
extern "C" int dic_colsum_pair(const float* in0, float* out0, const float* in1, float* out1, int rows, int cols, int ld, void* stream) {
    DIC_REQUIRE(cols % 4 == 0 && rows > 0 && rows <= 1024 && in0 && in1 && out0 && out1, "dic_colsum_pair: two fp32 problems of <= 1024 rows, cols a multiple of 4");
    hipLaunchKernelGGL(colsum_small, dim3((cols + 255) / 256, 2), dim3(64 * CS_WAVES), 0, (hipStream_t)stream, in0, rows, cols, ld, out0, 0, in1, out1);
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_colsum(int in_dtype, const void* in, int rows, int cols, int ld, float* out, int accumulate, float* ws, void* stream) {
    DIC_REQUIRE(cols % 4 == 0 && rows > 0, "dic_colsum: cols must be a multiple of 4");
    if (in_dtype == DIC_F32 && rows <= 1024) {
        hipLaunchKernelGGL(colsum_small, dim3((cols + 255) / 256), dim3(64 * CS_WAVES), 0, (hipStream_t)stream, (const float*)in, rows, cols, ld, out, accumulate,
                           (const float*)nullptr, (float*)nullptr);
        DIC_CHECK_LAUNCH();
        return 0;
    }
    int nslab = (rows + 3) / 4;
    if (nslab > 64) nslab = 64;
    dim3 grid((cols + 255) / 256, nslab);
    if (in_dtype == DIC_BF16)
        hipLaunchKernelGGL(colsum_stage1<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, rows, cols, ld, ws, nslab);
    else
        hipLaunchKernelGGL(colsum_stage1<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)in, rows, cols, ld, ws, nslab);
    hipLaunchKernelGGL(colsum_stage2, dim3((cols / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, nslab, cols, out, accumulate);
    DIC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ mean-row correction of a bf16-rounded weight
// bias_eff[n] = bias[n] + sum_k lo[n][k] * abar[k],  abar = column mean of (a sample of) the rows of the Linear's input A, lo = bf16(W - bf16(W)).
// A (W_hi + W_lo)^T = A W_hi^T + A W_lo^T: the second product is 2^-9 of the first, and what it contributes to a batch-mean loss is almost
// entirely its row-common part abar W_lo^T (the token-specific part is independent from row to row and averages out, like the activations'
// roundings: DESIGN.md section 4).  That part is ONE vector per Linear: a GEMV, handed to the GEMM as its bias, instead of a second pass of
// the K loop over every row.  Stage 1 = column sums of the sampled rows in LMB_SLABS slabs, summed in fixed order by stage 2: deterministic.
constexpr int LMB_OUT = 8, LMB_SLABS = 4, LMB_WAVES = 16;    // outputs per block of stage 2; row slabs of stage 1 and waves per slab
// stage 1: grid (K / 256, LMB_SLABS), 1024 threads; wave w of slab s sums the sampled rows s * LMB_WAVES + w, + LMB_WAVES * LMB_SLABS, ... (all of
// them in flight at once); LDS fold of the 16 waves.  (Round 5: 4 slabs of 16 waves instead of 16 slabs of 4 -- every block of stage 2 reads all
// the slabs, 196 KB each at K = 3072: 20 -> 10 us for the FFN lin2 preparation.)
__global__ __launch_bounds__(1024) void lo_mean_rows_kernel(const bf16_t* in, int rows, int cols, long long ld, float* ws) {
    __shared__ f32x4 red[LMB_WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    f32x4 a0{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (c < cols) {
        const int step = LMB_WAVES * LMB_SLABS;
        constexpr int DEPTH = 20;                         // (17 408 tokens / 16 = 1 088 sampled rows = 17 per wave: all in flight at once)
        for (int r = blockIdx.y * LMB_WAVES + w; r < rows; r += DEPTH * step) {
            uint2 v[DEPTH];
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                const int rj = r + j * step;
                v[j] = *(const uint2*)(in + (size_t)(rj < rows ? rj : r) * ld + c);          // (clamped: surplus rows are loaded and dropped)
            }
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                if (r + j * step < rows) {
                    f32x4 x;
                    x[0] = __uint_as_float(v[j].x << 16); x[1] = __uint_as_float(v[j].x & 0xffff0000u);
                    x[2] = __uint_as_float(v[j].y << 16); x[3] = __uint_as_float(v[j].y & 0xffff0000u);
                    if ((j & 3) == 0) a0 += x; else if ((j & 3) == 1) a1 += x; else if ((j & 3) == 2) a2 += x; else a3 += x;
                }
            }
        }
    }
    red[w][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (w == 0 && c < cols) {
        f32x4 t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = (red[4 * q][lane] + red[4 * q + 1][lane]) + (red[4 * q + 2][lane] + red[4 * q + 3][lane]);
        *(f32x4*)(ws + (size_t)blockIdx.y * cols + c) = (t[0] + t[1]) + (t[2] + t[3]);
    }
}
// stage 2: a block owns LMB_OUT outputs and splits K over its 256 threads (every weight row of the block is in flight at once)
__global__ __launch_bounds__(256) void lo_mean_bias_kernel(const float* slabs, float inv_rows, const bf16_t* lo, int ldb, int K, int N,
                                                           const float* bias, float* out) {
    extern __shared__ __attribute__((aligned(16))) float abar[];            // [K], then [4][LMB_OUT] partial sums
    float* red = abar + K;
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
        f32x4 a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < LMB_SLABS; ++sl) a += *(const f32x4*)(slabs + (size_t)sl * K + k);
        *(f32x4*)(abar + k) = a * inv_rows;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n0 = blockIdx.x * LMB_OUT;
    float acc[LMB_OUT];
#pragma unroll
    for (int q = 0; q < LMB_OUT; ++q) acc[q] = 0.f;
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
        const f32x4 sv = *(const f32x4*)(abar + k);
#pragma unroll
        for (int q = 0; q < LMB_OUT; ++q) {
            const int n = n0 + q < N ? n0 + q : N - 1;                      // (clamped: the surplus sums are not stored)
            const f32x4 wv = Elem<bf16_t>::ld4(lo + (size_t)n * ldb + k);
            acc[q] += (wv[0] * sv[0] + wv[1] * sv[1]) + (wv[2] * sv[2] + wv[3] * sv[3]);
        }
    }
#pragma unroll
    for (int q = 0; q < LMB_OUT; ++q) {
        const float t = wave_sum(acc[q]);
        if (lane == 0) red[w * LMB_OUT + q] = t;
    }
    __syncthreads();
    if (threadIdx.x < LMB_OUT && n0 + (int)threadIdx.x < N) {
        const int q = threadIdx.x;
        out[n0 + q] = (bias ? bias[n0 + q] : 0.f) + ((red[q] + red[LMB_OUT + q]) + (red[2 * LMB_OUT + q] + red[3 * LMB_OUT + q]));
    }
}
extern "C" size_t dic_lo_mean_bias_ws_bytes(int K) { return (size_t)LMB_SLABS * K * sizeof(float); }
extern "C" int dic_lo_mean_bias(const void* A, int T, int lda, int row_stride, int K, const void* lo, int ldb, int N, const float* bias, float* bias_eff,
                                float* ws, void* stream) {
    DIC_REQUIRE(A && lo && bias_eff && ws && T > 0 && row_stride > 0 && K % 4 == 0 && K > 0 && N > 0 && lda % 4 == 0 && ldb % 4 == 0 && K <= 8192,
                "dic_lo_mean_bias: bf16 A [T][lda] and lo [N][ldb], K a multiple of 4 (<= 8192), ws of dic_lo_mean_bias_ws_bytes(K)");
    const int rows = (T + row_stride - 1) / row_stride;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(lo_mean_rows_kernel, dim3((K + 255) / 256, LMB_SLABS), dim3(64 * LMB_WAVES), 0, st, (const bf16_t*)A, rows, K, (long long)lda * row_stride, ws);
    hipLaunchKernelGGL(lo_mean_bias_kernel, dim3((N + LMB_OUT - 1) / LMB_OUT), dim3(256), (size_t)(K + 4 * LMB_OUT) * sizeof(float), st, ws, 1.0f / (float)rows,
                       (const bf16_t*)lo, ldb, K, N, bias, bias_eff);
    DIC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ Linear preparation of the parity mode (round 5)
// dic_lo_mean_bias extended by the reference rows of the CENTRED residual stream, in front of a forward Linear y = drop(A W^T + b) + R
// (hf:183-185, 201, 221-223, 236, 253, 510):
//   abar      = column mean of every row_stride-th row of A                                   (stage 1: lo_mean_rows_kernel, LMB_SLABS slabs)
//   s_lo, s_hi = W_lo abar, W_hi abar                                                          (stage 2: GEMVs, LMB_OUT outputs per block)
//   bias_in   = b + s_lo                          the mean-row lo-weight correction: what the GEMM adds in front of the dropout
//   y_ref     = b + s_lo + s_hi + r_ref           (residual Linears) the PREDICTED mean row of the sum: the stored sum is bf16(y - y_ref)
//   bias_post = r_ref - y_ref                     what the GEMM adds behind the dropout so that acc + bias_in + bias_post + R_c = y - y_ref
//               (R_c = bf16(R - r_ref), the centred residual copy); fold_post: no dropout in between, bias_in += bias_post (= -s_hi)
// Two launches.  ONE launch with a grid barrier between the stages was built and measured three ways this round (slab stores + agent-scope
// release / acquire fences around a counter: 15-20 us; column sums and arrival counts as 64-bit fixed-point atomics polled by the readers:
// 18-27 us; an empty kernel back to back: 4.5 us): on this chip anything that crosses workgroups inside a kernel goes through memory
// (eight XCDs, eight L2s) at ~2 us per dependent round trip, and a kernel boundary is the cheapest such crossing there is
// (profiles/r05_lin_prep_probe.txt).
template <bool HI>
__global__ __launch_bounds__(256) void lin_prep_bias_kernel(const float* slabs, float inv_rows, const bf16_t* w_hi, const bf16_t* w_lo, int ldb, int K, int N,
                                                            const float* bias, const float* r_ref, int fold_post, float* bias_in, float* bias_post, float* y_ref) {
    extern __shared__ __attribute__((aligned(16))) float abar[];            // [K], then [4][2 * LMB_OUT] partial sums
    float* red = abar + K;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n0 = blockIdx.x * LMB_OUT;
    // the block's weight rows do not depend on stage 1: requested first, their latency overlaps the slab reads (K <= 3072: all in registers)
    constexpr int KPT = 3;
    uint2 pre_lo[KPT][LMB_OUT], pre_hi[HI ? KPT : 1][HI ? LMB_OUT : 1];
    const bool pre_ok = K <= 1024 * KPT;
    if (pre_ok) {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int k = threadIdx.x * 4 + j * 1024;
            if (k < K) {
#pragma unroll
                for (int q = 0; q < LMB_OUT; ++q) {
                    const int n = n0 + q < N ? n0 + q : N - 1;              // (clamped: the surplus sums are not stored)
                    pre_lo[j][q] = *(const uint2*)(w_lo + (size_t)n * ldb + k);
                    if constexpr (HI) pre_hi[j][q] = *(const uint2*)(w_hi + (size_t)n * ldb + k);
                }
            }
        }
    }
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
        f32x4 a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < LMB_SLABS; ++sl) a += *(const f32x4*)(slabs + (size_t)sl * K + k);
        *(f32x4*)(abar + k) = a * inv_rows;
    }
    __syncthreads();
    auto bfx4 = [](const uint2& u) {
        f32x4 x;
        x[0] = __uint_as_float(u.x << 16); x[1] = __uint_as_float(u.x & 0xffff0000u);
        x[2] = __uint_as_float(u.y << 16); x[3] = __uint_as_float(u.y & 0xffff0000u);
        return x;
    };
    float alo[LMB_OUT], ahi[LMB_OUT];
#pragma unroll
    for (int q = 0; q < LMB_OUT; ++q) alo[q] = ahi[q] = 0.f;
    if (pre_ok) {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int k = threadIdx.x * 4 + j * 1024;
            if (k < K) {
                const f32x4 sv = *(const f32x4*)(abar + k);
#pragma unroll
                for (int q = 0; q < LMB_OUT; ++q) {
                    const f32x4 wl = bfx4(pre_lo[j][q]);
                    alo[q] += (wl[0] * sv[0] + wl[1] * sv[1]) + (wl[2] * sv[2] + wl[3] * sv[3]);
                    if constexpr (HI) {
                        const f32x4 wh = bfx4(pre_hi[j][q]);
                        ahi[q] += (wh[0] * sv[0] + wh[1] * sv[1]) + (wh[2] * sv[2] + wh[3] * sv[3]);
                    }
                }
            }
        }
    } else {
        for (int k = threadIdx.x * 4; k < K; k += 1024) {
            const f32x4 sv = *(const f32x4*)(abar + k);
#pragma unroll
            for (int q = 0; q < LMB_OUT; ++q) {
                const int n = n0 + q < N ? n0 + q : N - 1;
                const f32x4 wl = Elem<bf16_t>::ld4(w_lo + (size_t)n * ldb + k);
                alo[q] += (wl[0] * sv[0] + wl[1] * sv[1]) + (wl[2] * sv[2] + wl[3] * sv[3]);
                if constexpr (HI) {
                    const f32x4 wh = Elem<bf16_t>::ld4(w_hi + (size_t)n * ldb + k);
                    ahi[q] += (wh[0] * sv[0] + wh[1] * sv[1]) + (wh[2] * sv[2] + wh[3] * sv[3]);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < LMB_OUT; ++q) {
        const float tl = wave_sum(alo[q]);
        const float th = HI ? wave_sum(ahi[q]) : 0.f;
        if (lane == 0) { red[(w * 2) * LMB_OUT + q] = tl; red[(w * 2 + 1) * LMB_OUT + q] = th; }
    }
    __syncthreads();
    if (threadIdx.x < LMB_OUT && n0 + (int)threadIdx.x < N) {
        const int q = threadIdx.x, n = n0 + q;
        const float s_lo = (red[q] + red[2 * LMB_OUT + q]) + (red[4 * LMB_OUT + q] + red[6 * LMB_OUT + q]);
        const float b_in = (bias ? bias[n] : 0.f) + s_lo;
        if constexpr (HI) {
            const float s_hi = (red[LMB_OUT + q] + red[3 * LMB_OUT + q]) + (red[5 * LMB_OUT + q] + red[7 * LMB_OUT + q]);
            const float rr = r_ref ? r_ref[n] : 0.f;
            const float yr = (b_in + s_hi) + rr;
            const float post = rr - yr;
            y_ref[n] = yr;
            if (bias_post) bias_post[n] = post;
            bias_in[n] = fold_post ? b_in + post : b_in;
        } else {
            bias_in[n] = b_in;
        }
    }
}
extern "C" size_t dic_lin_prep_ws_bytes(int K) { return (size_t)LMB_SLABS * K * sizeof(float); }
extern "C" int dic_lin_prep(const void* A, int T, int lda, int row_stride, int K, const void* w_hi, const void* w_lo, int ldb, int N, const float* bias,
                            const float* r_ref, int fold_post, float* bias_in, float* bias_post, float* y_ref, void* ws, void* stream) {
    DIC_REQUIRE(A && w_lo && bias_in && ws && T > 0 && row_stride > 0 && K % 4 == 0 && K > 0 && N > 0 && lda % 4 == 0 && ldb % 4 == 0 && K <= 8192,
                "dic_lin_prep: bf16 A [T][lda], lo [N][ldb], K a multiple of 4 (<= 8192), a ws of dic_lin_prep_ws_bytes(K)");
    DIC_REQUIRE((y_ref != nullptr) == (w_hi != nullptr) && (y_ref || (!bias_post && !r_ref && !fold_post)),
                "dic_lin_prep: the reference row y_ref needs the hi weights; bias_post / r_ref / fold_post only exist with it");
    const int rows = (T + row_stride - 1) / row_stride;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(lo_mean_rows_kernel, dim3((K + 255) / 256, LMB_SLABS), dim3(64 * LMB_WAVES), 0, st, (const bf16_t*)A, rows, K, (long long)lda * row_stride, (float*)ws);
    const dim3 grid((N + LMB_OUT - 1) / LMB_OUT);
    const size_t lds = (size_t)(K + 8 * LMB_OUT) * sizeof(float);
    if (y_ref)
        hipLaunchKernelGGL(lin_prep_bias_kernel<true>, grid, dim3(256), lds, st, (const float*)ws, 1.0f / (float)rows, (const bf16_t*)w_hi, (const bf16_t*)w_lo, ldb, K, N, bias,
                           r_ref, fold_post, bias_in, bias_post, y_ref);
    else
        hipLaunchKernelGGL(lin_prep_bias_kernel<false>, grid, dim3(256), lds, st, (const float*)ws, 1.0f / (float)rows, (const bf16_t*)nullptr, (const bf16_t*)w_lo, ldb, K, N, bias,
                           (const float*)nullptr, 0, bias_in, (float*)nullptr, (float*)nullptr);
    DIC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ rank-one completion of a weight gradient
// dW[m][n] += db[m] * x_ref[n]: the Linear's input is stored centred, X = X_c + 1 x_ref^T (dic_ln_fwd_cen), the weight-gradient GEMM contracts
// dY with X_c, and dY^T (1 x_ref^T) = colsum(dY) x_ref^T = db x_ref^T is what it leaves out (hf:183-185, 221, 510 backward; exact algebra, no
// approximation).  Streaming: 8 B per element of dW.
__global__ __launch_bounds__(256) void rank1_add_kernel(float* dW, const float* db, const float* x_ref, int M, int N) {
    const int n4 = N >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (long long)M * n4; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), c = (int)(i - (long long)m * n4) * 4;
        const float s = db[m];
        const f32x4 r = *(const f32x4*)(x_ref + c);
        f32x4* p = (f32x4*)(dW + (size_t)m * N + c);
        f32x4 v = *p;
        v[0] += s * r[0]; v[1] += s * r[1]; v[2] += s * r[2]; v[3] += s * r[3];
        *p = v;
    }
}
extern "C" int dic_rank1_add(float* dW, const float* db, const float* x_ref, int M, int N, void* stream) {
    DIC_REQUIRE(dW && db && x_ref && M > 0 && N > 0 && N % 4 == 0, "dic_rank1_add: dW [M][N] fp32 with N a multiple of 4, db [M], x_ref [N]");
    long long n = (long long)M * (N / 4);
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(rank1_add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dW, db, x_ref, M, N);
    DIC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ K15 AdamW
// torch.optim.AdamW semantics (ref :335): p *= 1-lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).  One pass over the flat buffers: 16 B/param read, 12(+2) B written.
// (shadow_lo, optional: bf16(p - bf16(p)), the low-order half the split-weight forward GEMMs add back -- DicGemmParams.B2)
__device__ __forceinline__ void store_bf16_hi_lo(uint16_t* hi, uint16_t* lo, long long i4, const f32x4& P) {
    uint2 u;
    u.x = pack2bf(P[0], P[1]);
    u.y = pack2bf(P[2], P[3]);
    *(uint2*)(hi + i4 * 4) = u;
    if (lo) {
        f32x4 H;                                                  // the rounded values, back in fp32
        H[0] = __uint_as_float(u.x << 16); H[1] = __uint_as_float(u.x & 0xffff0000u);
        H[2] = __uint_as_float(u.y << 16); H[3] = __uint_as_float(u.y & 0xffff0000u);
        Elem<bf16_t>::st4(lo + i4 * 4, P - H);
    }
}
__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, uint16_t* shadow, uint16_t* shadow_lo, long long n4, float lr, float b1,
                             float b2, float eps, float wd, float bc1, float rsqrt_bc2, float gscale, const float* table, const long long* ctr,
                             long long ctr0) {
    if (table && ctr) {          // replayed step: this step's bias corrections from the table the capture code filled (host arithmetic, bit for bit)
        const long long k = ctr[0] - ctr0;
        bc1 = table[2 * k]; rsqrt_bc2 = table[2 * k + 1];
    }
    const float step = lr / bc1, decay = 1.0f - lr * wd;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 P = ((f32x4*)p)[i], G = ((const f32x4*)g)[i] * gscale, Mo = ((f32x4*)m)[i], Vo = ((f32x4*)v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = P[k] * decay;
            float mk = Mo[k] * b1 + (1.0f - b1) * G[k];
            float vk = Vo[k] * b2 + (1.0f - b2) * G[k] * G[k];
            float denom = sqrtf(vk) * rsqrt_bc2 + eps;
            P[k] = pk - step * (mk / denom);
            Mo[k] = mk; Vo[k] = vk;
        }
        ((f32x4*)p)[i] = P; ((f32x4*)m)[i] = Mo; ((f32x4*)v)[i] = Vo;
        if (shadow) store_bf16_hi_lo(shadow, shadow_lo, i, P);
    }
}
extern "C" int dic_adamw_hl(float* p, const float* g, float* m, float* v, uint16_t* shadow, uint16_t* shadow_lo, int64_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, float grad_scale, void* stream) {
    DIC_REQUIRE(n % 4 == 0 && n > 0, "dic_adamw: flat length must be a multiple of 4");
    DIC_REQUIRE(shadow != nullptr || shadow_lo == nullptr, "dic_adamw_hl: a low-order shadow needs the bf16 shadow it is the remainder of");
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, shadow, shadow_lo, (long long)(n / 4),
                       lr, beta1, beta2, eps, weight_decay, bias_corr1, 1.0f / sqrtf(bias_corr2), grad_scale, dic_step_ctx().adam_table, dic_step_ctx().ctr,
                       dic_step_ctx().ctr0);
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_adamw(float* p, const float* g, float* m, float* v, uint16_t* shadow, int64_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, float bias_corr1, float bias_corr2, float grad_scale, void* stream) {
    return dic_adamw_hl(p, g, m, v, shadow, nullptr, n, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, grad_scale, stream);
}
__global__ void cast_bf16_kernel(const float* in, uint16_t* out, uint16_t* out_lo, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        store_bf16_hi_lo(out, out_lo, i, ((const f32x4*)in)[i]);
}
extern "C" int dic_cast_bf16_hl(const float* in, uint16_t* out, uint16_t* out_lo, int64_t n, void* stream) {
    DIC_REQUIRE(n % 4 == 0 && n > 0, "dic_cast_bf16: length must be a multiple of 4");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, in, out, out_lo, (long long)(n / 4));
    DIC_CHECK_LAUNCH();
    return 0;
}
extern "C" int dic_cast_bf16(const float* in, uint16_t* out, int64_t n, void* stream) { return dic_cast_bf16_hl(in, out, nullptr, n, stream); }

// ------------------------------------------------------------------------------------------------ layout probe
// out[lane*4+j] = element j returned to `lane` by ds_read_b64_tr_b16 when lane supplies address lds + lane*8 over
// in[0..255].  The GPU test asserts the mapping gemm.hip/attn.hip rely on:  out[l*4+j] == in[(l&15) + 16*j + 64*(l>>4)].
__global__ void probe_tr16_kernel(const uint16_t* in, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[256];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) lds[i] = in[i];
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + l * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
extern "C" int dic_probe_tr16(const uint16_t* in, uint16_t* out, void* stream) {
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
    DIC_CHECK_LAUNCH();
    return 0;
}
