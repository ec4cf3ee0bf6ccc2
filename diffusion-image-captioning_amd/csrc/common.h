// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned short bf16_t;   // raw bf16 bits

#define DIC_F32 0
#define DIC_BF16 1

#define LDS_PTR(T) __attribute__((address_space(3))) T*

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) ---------------------------------------
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, RNE, NaN quieted): one instruction per PAIR instead of ~10 VALU ops and an
// exec-masked NaN branch per element
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 hbf16x2;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, hbf16x2));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int dtype = DIC_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    // 4 consecutive elements
    __device__ static __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
    __device__ static __forceinline__ void st4(float* p, f32x4 v) { *(f32x4*)p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int dtype = DIC_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ f32x4 ld4(const bf16_t* p) {
        uint2 u = *(const uint2*)p;
        f32x4 r;
        r[0] = __uint_as_float(u.x << 16); r[1] = __uint_as_float(u.x & 0xffff0000u);
        r[2] = __uint_as_float(u.y << 16); r[3] = __uint_as_float(u.y & 0xffff0000u);
        return r;
    }
    __device__ static __forceinline__ void st4(bf16_t* p, f32x4 v) {
        uint2 u;
        u.x = pack2bf(v[0], v[1]);
        u.y = pack2bf(v[2], v[3]);
        *(uint2*)p = u;
    }
};

// ---- wave / block reductions ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- fast erf-GELU for the bf16 GEMM epilogues ---------------------------------------------------------
// Abramowitz-Stegun 7.1.26: erf(z) = 1 - (a1 t + ... + a5 t^5) e^{-z^2}, t = 1/(1 + p z), |error| <= 1.5e-7 -- three
// orders below bf16 resolution, ~14 VALU ops instead of ocml erff's ~45 (the epilogue of the two FFN GEMMs evaluates it
// 2.4 M times per workgroup wave).  e^{-z^2} with z = x/sqrt(2) is also the Gaussian factor of GELU', so the
// derivative costs the same single v_exp_f32.  The fp32 parity path keeps ocml erff.
__device__ __forceinline__ void gelu_fast_parts(float x, float& cdf, float& gauss) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    gauss = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);            // e^{-x^2/2}
    float poly = fmaf(fmaf(fmaf(fmaf(1.061405429f, t, -1.453152027f), t, 1.421413741f), t, -0.284496736f), t, 0.254829592f) * t;
    const float e = 1.0f - poly * gauss;                                        // erf(|x|/sqrt2)
    cdf = 0.5f * (1.0f + copysignf(e, x));
}
// The same arithmetic two elements per instruction (v_pk_fma_f32 / v_pk_mul_f32): the GEMM epilogues that apply GELU / GELU' to a whole
// 224 x 256 tile are VALU-bound (s_memtime: 12 k cycles per tile and wave), and 22 of the ~30 operations per element pair are fma / mul.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t pk_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2_t pk_splat(float v) { return f32x2_t{v, v}; }
__device__ __forceinline__ void gelu_fast_parts2(f32x2_t x, f32x2_t& cdf, f32x2_t& gauss) {
    const f32x2_t d = pk_fma(__builtin_elementwise_abs(x), pk_splat(0.3275911f * 0.70710678118654752f), pk_splat(1.0f));
    const f32x2_t t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2_t a = x * x * pk_splat(-0.72134752044448170f);
    gauss = f32x2_t{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};          // e^{-x^2/2}
    f32x2_t p = pk_fma(pk_splat(1.061405429f), t, pk_splat(-1.453152027f));
    p = pk_fma(p, t, pk_splat(1.421413741f));
    p = pk_fma(p, t, pk_splat(-0.284496736f));
    p = pk_fma(p, t, pk_splat(0.254829592f));
    p = p * t;
    const f32x2_t e = pk_fma(-p, gauss, pk_splat(1.0f));                                  // erf(|x|/sqrt2)
    const f32x2_t es{copysignf(e[0], x[0]), copysignf(e[1], x[1])};
    cdf = pk_fma(es, pk_splat(0.5f), pk_splat(0.5f));
}
template <class V4> __device__ __forceinline__ void gelu_fast4(V4& v) {                  // v <- gelu(v), four elements
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2_t x{v[2 * h], v[2 * h + 1]};
        f32x2_t c, g; gelu_fast_parts2(x, c, g);
        const f32x2_t y = x * c;
        v[2 * h] = y[0]; v[2 * h + 1] = y[1];
    }
}
template <class V4> __device__ __forceinline__ void gelu_fast_with_grad4(V4& v, V4& d) {  // v <- gelu(v), d <- gelu'(v): one erf / exp evaluation for both
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2_t x{v[2 * h], v[2 * h + 1]};
        f32x2_t c, g; gelu_fast_parts2(x, c, g);
        const f32x2_t y = x * c;
        const f32x2_t dd = pk_fma(x * pk_splat(0.3989422804014327f), g, c);
        v[2 * h] = y[0]; v[2 * h + 1] = y[1];
        d[2 * h] = dd[0]; d[2 * h + 1] = dd[1];
    }
}
template <class V4> __device__ __forceinline__ void gelu_grad_mul4(V4& v, const V4& u) { // v <- v * gelu'(u), four elements
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2_t x{u[2 * h], u[2 * h + 1]};
        f32x2_t c, g; gelu_fast_parts2(x, c, g);
        const f32x2_t d = pk_fma(x * pk_splat(0.3989422804014327f), g, c);
        v[2 * h] *= d[0]; v[2 * h + 1] *= d[1];
    }
}
__device__ __forceinline__ float gelu_fast(float x) { float c, g; gelu_fast_parts(x, c, g); return x * c; }
__device__ __forceinline__ float gelu_grad_fast(float x) { float c, g; gelu_fast_parts(x, c, g); return fmaf(x * 0.3989422804014327f, g, c); }

// ---- exact (erf) GELU, hf get_activation("gelu") -------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// ---- counter-based RNG ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011): used for q_sample noise and for dropout keep-masks, keyed by
// (seed, element index) so a backward kernel regenerates exactly the mask its forward used.
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        unsigned hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
        unsigned hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += 0x9E3779B9u; key.y += 0xBB67AE85u;
    }
    return ctr;
}
__device__ __forceinline__ uint4 rng4(unsigned long long seed, unsigned long long e4) {
    return philox4x32(make_uint4((unsigned)e4, (unsigned)(e4 >> 32), 0x0d1c5eedu, 0u),
                      make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
}
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// ---- dropout keep-masks ----------------------------------------------------------------------------------
// Philox (above) is kept for the Gaussian noise of q_sample, drawn once per step.  Dropout decisions are drawn ~10^9 times
// per step inside GEMM epilogues, LayerNorm and attention kernels, where ten Philox rounds (40 quarter-rate integer multiplies
// per 4 elements) made HBM-bound kernels ALU-bound; they use a 2-multiply avalanche hash (lowbias32, bias 0.17) of
// (seed, element pair): one 32-bit hash decides two consecutive elements with 16-bit thresholds, so the drop probability is
// p rounded to 1/65536 (0.1 -> 0.100006) and the rescale uses that exact value.  Backward kernels regenerate the same mask
// from the same (seed, index).
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned pair_hash(unsigned long long seed, unsigned long long pair) {
    const unsigned hi = (unsigned)(pair >> 32) * 0x9E3779B9u + (unsigned)(seed >> 32);
    return hash32((unsigned)pair ^ (unsigned)seed ^ hash32(hi));
}
__device__ __forceinline__ unsigned drop_thr(float p) { return (unsigned)(p * 65536.0f + 0.5f); }
__device__ __forceinline__ float drop_inv_keep(float p) { return p > 0.f ? 65536.0f / (65536.0f - (float)drop_thr(p)) : 1.0f; }
// dropout on 4 consecutive values whose flat index starts at idx (idx % 4 == 0)
__device__ __forceinline__ f32x4 dropout4(f32x4 v, unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    const unsigned thr = drop_thr(p);
    const unsigned h0 = pair_hash(seed, idx >> 1), h1 = pair_hash(seed, (idx >> 1) + 1);
    v[0] = ((h0 & 0xffffu) >= thr) ? v[0] * inv_keep : 0.f;
    v[1] = ((h0 >> 16) >= thr) ? v[1] * inv_keep : 0.f;
    v[2] = ((h1 & 0xffffu) >= thr) ? v[2] * inv_keep : 0.f;
    v[3] = ((h1 >> 16) >= thr) ? v[3] * inv_keep : 0.f;
    return v;
}
__device__ __forceinline__ float dropout1(float v, unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    const unsigned h = pair_hash(seed, idx >> 1);
    const unsigned w = (idx & 1) ? (h >> 16) : (h & 0xffffu);
    return (w >= drop_thr(p)) ? v * inv_keep : 0.f;
}

// ---- per-step seed indirection (hipGraph replay of the training step) ---------------------------------------------
// A captured launch freezes its arguments, but the dropout / noise / timestep seeds, AdamW's bias corrections and the slot the step's
// losses are written to change every step.  While a step context is set (dic_step_ctx_set, called by the capture code), every seeded
// kernel receives the address of a device step counter and its value at capture time and shifts its seed by
// (counter - counter_at_capture) * stride -- exactly what the host adds between two eager steps, so a replayed step draws the very
// masks / noise / timesteps the eager step with the same number would.  Without a context (eager code): ctr == NULL, zero shift.
struct DicStepCtx { const long long* ctr; long long ctr0; unsigned long long stride_noise; const float* adam_table; };
DicStepCtx dic_step_ctx();
struct SeedArg {
    unsigned long long base, stride;
    const long long* ctr;
    long long ctr0;
    __device__ __forceinline__ unsigned long long resolve() const { return ctr ? base + (unsigned long long)(ctr[0] - ctr0) * stride : base; }
};
enum { DIC_STRIDE_DROP = 64, DIC_STRIDE_T = 1 };          // what engine.encode / diffusion._next_t_seed add per step on the host
inline SeedArg make_seed(unsigned long long seed, unsigned long long stride) {
    const DicStepCtx c = dic_step_ctx();
    return SeedArg{seed, stride, c.ctr, c.ctr0};
}
__device__ __forceinline__ long long step_delta(const long long* ctr, long long ctr0) { return ctr ? ctr[0] - ctr0 : 0; }

// ---- error plumbing for the C-ABI ----------------------------------------------------------------
// internal to the library (hidden visibility: not part of the C-ABI include/dic_hip.h declares; callers read dic_last_error())
extern "C" __attribute__((visibility("hidden"))) void dic_set_error(const char* msg);
#define DIC_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) {                            \
            dic_set_error(hipGetErrorString(e__));          \
            return (int)e__;                                \
        }                                                   \
    } while (0)
#define DIC_REQUIRE(cond, msg)                              \
    do {                                                    \
        if (!(cond)) {                                      \
            dic_set_error(msg);                             \
            return 1001;                                    \
        }                                                   \
    } while (0)
