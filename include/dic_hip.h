/* dic_hip.h -- C-ABI of the MI355X (gfx950) hot path of CLIP-Diffusion-LM captioning.
 *
 * The reference (xu-shitong/diffusion-image-captioning) has no FFI: its hot path is Python calling
 * ATen/HuggingFace ops.  This header is therefore the boundary a maintainer would bind INSTEAD of those op
 * groups; each entry point cites the reference lines whose arithmetic it replaces ("ref" =
 * /root/reference/CLIP-DDPM.py, "hf" = transformers/models/distilbert/modeling_distilbert.py 5.15.0).
 * INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory owned by the caller
 *   - kernels never allocate, never synchronise, launch on `stream` (a hipStream_t passed as void*); no per-call state is kept (the
 *     process-global exceptions -- last-error text, the bench timing hook, two measurement / replay switches -- say so where declared)
 *   - return value: 0 = ok, otherwise a hipError_t (or >= 1000 for argument errors); dic_last_error()
 *     gives the message
 *   - dtype: DIC_F32 (0) or DIC_BF16 (1) selects the activation/operand type `T`; statistics, losses,
 *     gradients of parameters, optimizer state and biases are always float32
 *   - row-major tensors; "T" below = number of tokens = sequences x tokens-per-sequence
 */
#ifndef DIC_HIP_H
#define DIC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIC_F32 0
#define DIC_BF16 1
/* flag on the `dtype` of dic_gelu_ln_fwd / dic_gelu_ln_bwd: the MLM-head pre-activation u is stored in fp32 whatever the engine's type */
#define DIC_U_F32 0x100
/* FP32 RESIDUAL STREAM of the bf16 engines (hf:236, 253: sa_layer_norm(attn + x), output_layer_norm(ffn + sa) with the sums and the residual
 * operands in fp32, the MFMA operands in bf16).  The pre-LayerNorm sums y1 / y2 and the residual reads are the rounding points whose error is
 * common to all tokens while the denoiser's rows are nearly equal (profiles/r04_collapse_probe.txt, "residual stream"):
 *   - DicGemmParams.out_f32 = DIC_OUT_F32 | DIC_RES_IS_F32: C is fp32 AND the residual R is fp32 (AFFINE, bf16 forward layouts, N % 8 == 0);
 *   - dic_ln_fwd_r32: LayerNorm of an fp32 y, writing the bf16 operand copy h and (optionally) the fp32 residual copy h32;
 *   - dic_ln_bwd with dtype DIC_BF16 | DIC_RES_F32: y is fp32, dh / dx bf16.                                                                  */
#define DIC_RES_F32 0x200
#define DIC_OUT_F32 1
#define DIC_RES_IS_F32 2

/* ABI version: bumped whenever a struct layout or a signature in this header changes; a binding must refuse a library whose dic_version()
 * differs from the DIC_HIP_VERSION it was written against (diffusion-image-captioning_amd/_lib.py does).                          */
#define DIC_HIP_VERSION 19
int dic_version(void);
const char* dic_last_error(void);

/* ---------------------------------------------------------------- GEMM (hf:183-185,201,221-223,510; ref:299,323)
 * C[m][n] = sum_k A(m,k) B(n,k), A stored [M][lda] (a_km=0) or [K][lda] (a_km=1); B likewise.
 * Epilogues:
 *   AFFINE      C = dropout(acc + bias[n]) + R[m][n]   (bias, R optional; p_drop 0 = none); out T or f32,
 *               accumulate=1 adds the previous C (f32 output only)         -- nn.Linear fwd/bwd, residual adds
 *   BIAS_GELU   aux = acc + bias ; C = gelu(aux)                            -- hf:221-222 (ffn.lin1 + GELU); aux == NULL: the
 *               pre-activation is not kept (forward-only calls: sampling, validation)
 *   GELU_BWD    C = acc * gelu'(aux)                                        -- backward of the above
 *   BIAS_GELU_D u = acc + bias ; C = gelu(u) ; aux = gelu'(u)  (bf16 LDS-DMA kernels only) -- the same forward, but what it leaves behind for
 *               the backward is the DERIVATIVE (evaluated on the unrounded fp32 u, where erf / exp are already at hand) instead of the
 *               pre-activation: same bytes, and the backward's epilogue becomes one multiply (MUL_AUX) instead of erf + exp per element
 *   MUL_AUX     C = acc * aux                                              -- backward of BIAS_GELU_D (bf16 LDS-DMA kernels only)
 *   CE_PARTIAL  per (row, 64-col half tile): {max, sum exp(x-max), first argmax}; tgt_logit[m] = acc[m][tgt[m]]
 *               -- streaming form of softmax/gather/argmax over the 30522-wide logits (ref:323,436-437,620)
 *   CE_DLOGITS  C = (exp(acc - lse[m]) - [n == tgt[m]]) * (m < ce_rows_a ? ce_scale_a : ce_scale_b),
 *               zero for N <= n < ldc                                        -- backward of the rounding loss (recompute form)
 *   CE_EXP      C = exp(acc - lse[m]) (bf16; `lse` holds the caller's per-row REFERENCE POINT c[m], see dic_ce_target_logit), zero for
 *               N <= n < ldc; partial[m][n/64] = sum of the unrounded values over that 64-column slab (np floats per row, np =
 *               dic_ce_n_partials(N, tile)); tgt_logit[m] = acc[m][tgt[m]].  The training forward of the rounding loss: dic_ce_exp_combine
 *               then turns C into the gradient operand, so the backward is ONE GEMM (C x W) and no second pass over the vocabulary
 *               (bf16 LDS-DMA kernels only)
 */
#define DIC_EPI_AFFINE 0
#define DIC_EPI_BIAS_GELU 1
#define DIC_EPI_GELU_BWD 2
#define DIC_EPI_CE_PARTIAL 3
#define DIC_EPI_CE_DLOGITS 4
#define DIC_EPI_CE_EXP 5
#define DIC_EPI_BIAS_GELU_D 6
#define DIC_EPI_MUL_AUX 7

typedef struct DicGemmParams {
    const void* A; const void* B; void* C;
    int M, N, K;
    int lda, ldb, ldc;
    const float* bias;          /* [N] or NULL */
    const void* R; int ldr;     /* residual, dtype T, or NULL */
    void* aux; int ldaux;       /* BIAS_GELU: out pre-activation; GELU_BWD: in pre-activation */
    float p_drop; uint64_t seed;/* dropout on (acc+bias), mask keyed by (seed, m*N+n) */
    int out_f32; int accumulate;/* out_f32: 0 = C in T, DIC_OUT_F32 = fp32 C, | DIC_RES_IS_F32 = and R is fp32 (see DIC_RES_F32 above) */
    const int64_t* tgt;         /* [M] target ids (CE) */
    const float* lse;           /* [M] logsumexp (CE_DLOGITS) */
    float* partial;             /* [M][np][4] (CE_PARTIAL); np = 2*ceil(N/128) for 128-tiles, 4*ceil(N/256) for tile=256 */
    float* tgt_logit;           /* [M] (CE_PARTIAL) */
    int ce_rows_a; float ce_scale_a, ce_scale_b;
    int split_k; void* split_ws;/* >1: K is cut into split_k slices (fills the chip when M*N has few tiles -- the dW GEMMs);
                                   slices write fp32 partial tiles to split_ws [split_k][M*N (+M)], then folded into C in fixed order */
    int tile;                   /* 0/128: 128x128 workgroup tiles; 256: 256x256 tiles, 8 waves (bf16 kernel) -- half the
                                   L2->LDS traffic per flop; worth it when M*N/65536 tiles still fill the 256 CUs */
    int cu_cap;                 /* >0: cap the persistent grid at this many CUs' worth of workgroups (bf16 kernel), leaving the rest of the chip
                                   to a kernel on another stream */
    float* colsum_out;          /* bf16 (k-major,k-major) fp32-output GEMMs only: out[m] = sum_k A(m,k) -- the bias gradient that
                                   goes with a weight gradient dW = dY^T X (hf nn.Linear backward), taken from the LDS-resident A tile */
    const void* B2;             /* bf16 forward GEMMs (k-contiguous A and B, AFFINE / BIAS_GELU, no split-K) or NULL: the LOW-ORDER half of a
                                   split fp32 weight, same shape / ldb as B: C = A (B + B2)^T, computed as two passes of the K loop into one
                                   accumulator (B = bf16(W), B2 = bf16(W - B): 16 mantissa bits; dic_adamw_hl / dic_cast_bf16_hl produce the pair).
                                   Replaces nn.Linear's fp32 weight in hf:183-185, 201, 221-223, 510 at bf16 MFMA rate x 1/2 */
    const int64_t* step_ctr;    /* RESERVED, leave 0: dic_gemm fills these two from the step context (dic_step_ctx_set) so that a launch */
    int64_t step_ctr0;          /* replayed inside a hipGraph shifts `seed` by 64 x (steps since capture), as the host does between eager steps */
    int b2_col0;                /* with B2: only output columns >= b2_col0 (a multiple of 256) take the second pass, the others use B alone -- the
                                   fused q|k|v projection with the low-order half on its value third only (b2_col0 = 2 D); 0: every column */
    const float* bias2;         /* [N] or NULL (bf16 forward AFFINE with dropout and a bf16 residual): C = dropout(acc + bias) + bias2 + R -- the row the
                                   centred residual stream adds BEHIND the dropout of hf:223 (dic_lin_prep's bias_post); without dropout it is part of `bias` */
} DicGemmParams;

int dic_gemm(int dtype, int a_km, int b_km, int epi, const DicGemmParams* p, void* stream);

/* Weight gradients of several nn.Linear layers in ONE launch (bf16 operands, fp32 results): dW_i [M_i][N_i] = dY_i^T X_i over the T tokens,
 * db_i [M_i] = column sums of dY_i (optional) -- what autograd computes for the q/k/v, out-proj and FFN Linears of a transformer block
 * (hf:183-185, 201, 221-223 backward).  dY_i is [T][ldy] and X_i [T][ldx] (both k-major for this contraction).  Every 256 x 256 tile of
 * every problem is cut into the same number of K-slices (chosen for the whole group), the (slice, tile) units are walked slice-slowest by one
 * persistent grid, partial tiles land in `ws` and a second launch folds them in slice order (deterministic).  M_i must be a multiple of 256, N_i / ldy / ldx of 8; at most 8 problems; `items` is a HOST array.          */
typedef struct DicWgradItem {
    const void* dY; int ldy;
    const void* X; int ldx;
    float* dW;                  /* [M][N], overwritten */
    float* db;                  /* [M] or NULL */
    int M, N;
} DicWgradItem;
size_t dic_wgrad_group_ws_bytes(const DicWgradItem* items, int n, int T, int cu_cap);
int dic_wgrad_group(const DicWgradItem* items, int n, int T, void* ws, size_t ws_bytes, int cu_cap, void* stream);

/* Workspace sizes (bytes) the caller must provide -- kernels never allocate.
 *   dic_gemm_split_ws_bytes : split_ws of a split-K launch (split_k slabs of M*N fp32, + M when colsum_out is used)
 *   dic_ce_n_partials       : records per row written to `partial` by CE_PARTIAL for this N and tile (pass to dic_ce_combine)
 *   dic_ce_partial_bytes    : size of `partial` ([M][n_partials][4] fp32)
 *   dic_colsum_ws_bytes     : `ws` of dic_colsum (0 when the single-launch path is taken)
 *   dic_ln_partial_bytes    : `partial` of the LayerNorm backward kernels (n_blocks rows of n_vectors*D fp32)             */
size_t dic_gemm_split_ws_bytes(int M, int N, int split_k, int with_colsum);
int    dic_ce_n_partials(int N, int tile);
size_t dic_ce_partial_bytes(int M, int N, int tile);
size_t dic_colsum_ws_bytes(int in_dtype, int rows, int cols);
size_t dic_ln_partial_bytes(int n_partial_blocks, int n_vectors, int D);

/* Kept for ABI stability.  The library carries ONE 8-wave K loop (+ the generated four-wave asm kernel); the round-3 alternatives it once
 * selected are no longer built.  0 is accepted, anything else returns 1006.                                                          */
int dic_gemm_set_variant(int pp);
/* Every process-global switch of the library by name (the library reads no environment variable): "gemm_w4a" (0 / 1, default 0: the four-wave asm
 * GEMM where eligible), "gemm_w4a_mask" (default 0x3FF; bit 4 * b_km + v allows epilogue form v = 0 plain, 1 + residual, 2 x aux, 3 dropout + residual; bit 8: BIAS_GELU / BIAS_GELU_D without aux, bit 9: BIAS_GELU_D with aux -- k-contiguous B only), "gemm_w4a_rows" (default 0: the asm kernel's tile height per launch -- 256 or 224 rows, whichever fills the rounds of resident workgroups better; 224 / 256 force one), "gemm_two_heights" (default 0), "gemm_rows" (default 1: per-launch tile heights), "gemm_persist" (default 1: persistent grids),
 * "gemm_v1" (default 0: bf16 on the register-staged fp32-style kernel), "gemm_w4n" (0 / 1, default 0; needs "gemm_w4a" = 1: launches inside the four-wave asm GEMM's scope -- with its K condition replaced by "a multiple of 192 in
 * [576, gemm_w4n_kmax]", "gemm_w4n_kmax" default 1024 -- run on its NARROW-tile form -- 256 x 128 tiles, the finished tile's epilogue drained under the next tile's K loop,
 * csrc/gemm_w4n.h; results are bit-identical to the wide bodies'), "gemm_w4n_mask" (default 0x740 = the heavy-epilogue forms; the bits of "gemm_w4a_mask" + bit 10, the CE_EXP
 * launch of the rounding head -- ldc = N rounded up to 128, tile 256),
 * "gemm_w4n_flat" (default 1: K = 768 launches take the loop-free narrow bodies, 0: the loop form everywhere).  Unknown name: 1007.                */
int dic_set_option(const char* name, int value);
/* Measurement / test switch (PROCESS-GLOBAL): 1 (default 0; env DIC_GEMM_TWO_HEIGHTS=1 turns it on everywhere) lets a forward GEMM of the 256-column
 * geometry run whole rounds of tall tiles followed by ONE round of shorter tiles over the remaining rows, when its units would otherwise end
 * in a partly filled round of the persistent grid; 0 = one tile height per launch.  Results are identical (a tile's arithmetic does not
 * depend on its height).  Returns the previous setting.  Pays on single-stream forward passes (the sampling loop turns it on), costs on the
 * two-stream training step, where another stream's kernels use the CUs a partial round leaves idle.                                 */
int dic_gemm_set_two_heights(int on);
/* Measurement / test switch (PROCESS-GLOBAL; env DIC_GEMM_W4A): 1 = a dic_gemm call inside its scope (bf16, k-contiguous A and B, AFFINE with optional
 * bias, bf16 output, M and N multiples of 256, K a multiple of 128, tile = 256) runs on the hand-scheduled four-wave kernel (csrc/gemm_w4a.h: one wave per
 * SIMD, accumulators in the AGPR file, tile loop + K loop + epilogue as one generated assembly block) instead of the 8-wave kernel.  Same tile order,
 * LDS image and MFMA order; the bias is added after the K loop instead of before it (last-bit differences).  Returns the previous setting.       */
int dic_gemm_set_w4a(int on);
/* host-only: the plan for an M x N x K forward problem on this device -- out[0] / out[1] = 16-row fragments per wave of the tall / the
 * last-round tiles (out[0] == 0: one height), out[2] = rows covered by the tall tiles */
int dic_gemm_two_heights_plan(int M, int N, int K, int cu_cap, int* out3);
/* host-only, needs no device: the tile height (256 or 224 rows) the four-wave asm kernel takes for an M x N x K problem on `cus` compute units under the
 * current "gemm_w4a_rows" option -- rounds of resident workgroups x (16-row fragments per wave + a fixed per-tile cost); -1 on bad arguments            */
int dic_gemm_w4a_rows_plan(int M, int N, int K, int cus);

/* hipGraph support for the training step (PROCESS-GLOBAL state: one capture at a time).  Kernel arguments are frozen by a capture, but the
 * dropout / noise / timestep seeds, AdamW's bias corrections and the slot the step's losses go to change every step.  While a context is set,
 * every seeded launch additionally receives `ctr` (a device int64 step counter) and `ctr0` (its value during the captured step) and shifts
 * its seed by (ctr[0] - ctr0) x the amount the host adds per eager step (dropout 64, timestep draw 1, q_sample `stride_noise`); dic_seg_sum
 * writes result slot (ctr[0] - ctr0); dic_adamw takes its two bias-correction factors from adam_table[2k], [2k+1] (k = ctr[0] - ctr0; floats
 * {1 - beta1^t, 1/sqrt(1 - beta2^t)} for the steps after the captured one).  dic_step_advance(ctr) is the graph's first node: ctr[0] += 1.
 * dic_step_ctx_set(NULL, 0, 0, NULL) switches the indirection off (eager launches: plain seeds).                                            */
int dic_step_ctx_set(const int64_t* ctr, int64_t ctr0, uint64_t stride_noise, const float* adam_table);
int dic_step_advance(int64_t* ctr, void* stream);

/* Measurement hooks for bench.py: between begin/end every dic_gemm launch is bracketed by hipEvents recorded on its own
 * stream; end() (after the caller synchronised) returns the summed kernel time, algorithmic flops (2*M*N*K) and count.
 * PROCESS-GLOBAL state (one record table for the whole process, armed per calling thread): a measurement aid, not for concurrent
 * use.  The only other shared state of the library: dic_last_error's message buffer, dic_gemm_set_variant, dic_step_ctx_set. */
int dic_prof_begin(int max_launches);
int dic_prof_end(double* total_ms, double* total_flops, int* n_launches);
/* ALGORITHMIC bytes of the launches recorded since dic_prof_begin (call before dic_prof_end): each operand, side input and output once. */
double dic_prof_algorithmic_bytes(void);
/* record i of the armed window (after a stream synchronise, before dic_prof_end): kernel milliseconds (GEMM + its slab fold), flops, algorithmic bytes; 0 = past the end */
int dic_prof_get(int i, double* ms, double* flops, double* bytes);

/* Reduce CE_PARTIAL output: lse[m], argmax[m] (first index of the max, as torch.argmax), nll[m] = lse - tgt_logit.
 * ref:436-437 (-log softmax gathered at the true id) and ref:620 (softmax.argmax).                               */
int dic_ce_combine(const float* partial, const float* tgt_logit, int M, int n_partials,
                   float* lse, int64_t* argmax, float* nll, void* stream);

/* Rounding loss, training form (ref:323, 436-437 forward AND the backward autograd derives from them), three steps around two GEMMs:
 *  1. dic_ce_target_logit: t[m] = <xr[m], W[tgt[m]]> (fp32 accumulate over the bf16 operands), c[m] = t[m] + shift -- the reference point of
 *     row m's exponentials.  With shift = 40 everything up to a per-token loss of ~109 nats is exact to fp32/bf16 rounding; beyond, CE_EXP
 *     saturates the exponential at 2^100 (the loss of such a row reads ~109, its gradient stays finite) instead of overflowing; what
 *     underflows is < e^-47 of the row's sum.  A target outside [0, V) gives t = 0.
 *  2. dic_gemm(CE_EXP) with lse = c: E [M][ldE] bf16, the slab sums, tgt_logit.
 *  3. dic_ce_exp_combine: Z[m] = sum of the slab sums (fixed order); lse[m] = log Z + c[m]; nll[m] = lse[m] - tgt_logit[m]; inv_z[m] = 1/Z;
 *     and E[m][tgt[m]] = exp(tgt_logit[m] - c[m]) - Z[m], so that  inv_z[m] * E[m][:]  ==  softmax(logits[m]) - onehot(tgt[m])  to bf16
 *     rounding per element -- what CE_DLOGITS writes after a second GEMM.
 *  The gradient w.r.t. xr is then  row_scale[m] * inv_z[m] * (E @ W)[m]:  dic_gemm (k-contiguous A = E, k-major B = W, fp32 out) followed by
 *  dic_add_rows_scaled.                                                                                                            */
int dic_ce_target_logit(const void* xr_bf16, const void* W_bf16, const int64_t* tgt, int M, int V, int D, float shift, float* t, float* c,
                        const float* col_bias /* [V] or NULL: added to t (the bias the head GEMM is given, dic_head_center) */, void* stream);

/* Mean-centred input of the rounding head (ref:323 `lm_head(x_out[:, :L])` evaluated as (x - xbar) W^T + xbar W^T; bf16 engines).  The head rows are
 * rows t < L of the n_a sequences at x_a and the n_b sequences at x_b (each [n][Tk][D] fp32; x_b optional).  Writes xbar [D] = their mean,
 * cvec [Vpad] = W32 xbar (fp32; W32 is [Vpad][D], rows >= V zero) -- pass it to the head's dic_gemm (CE_PARTIAL / CE_EXP / CE_DLOGITS; bf16 LDS-DMA kernels) as `bias` and to
 * dic_ce_target_logit as col_bias -- and xr [(n_a + n_b) L][D] = bf16(x - xbar).  The logits are the same function of x as before (the backward is
 * unchanged); what changes is that the part every row shares is no longer rounded to bf16: when the rows are nearly equal, as an early-training
 * denoiser's are, that rounding error is the same for every row and a batch-mean loss does not average it out.  ws: dic_head_center_ws_bytes(D). */
size_t dic_head_center_ws_bytes(int D);
int dic_head_center(const float* x_a, int n_a, const float* x_b, int n_b, int L, int Tk, int D, const float* W32, int Vpad, float* ws,
                    float* xbar, float* cvec, void* xr_bf16, void* stream);
int dic_ce_exp_combine(const float* partial, int n_partials, const float* c, const float* tgt_logit, const int64_t* tgt, int M, int V,
                       void* E_bf16, int ldE, float* lse, float* nll, float* inv_z, void* stream);

/* ---------------------------------------------------------------- embedding + q_sample (ref:459, 347-362) */
/* out[i] = E[ids[i]].  An id outside [0, V) (where nn.Embedding raises) zero-fills its row and is REPORTED through `err` (optional;
 * two ints zeroed by the caller, device memory or device-visible pinned host memory): err[0] += 1, err[1] = max(err[1], i + 1).      */
int dic_embed_gather(const int64_t* ids, const float* E, float* out, int n_tokens, int D, int V, int* err, void* stream);
/* out[(s*B+b)*LD + i] = sqrt_ac[t[s]]*x0[b*LD+i] + eps[b*LD+i]*sqrt_1mac[t[s]] (un-fused mul/mul/add: bit-exact with
 * ref:360-362); sqrt_ac = sqrt(alpha_cumprod), sqrt_1mac = sqrt(1-alpha_cumprod), tables of length step_tot;
 * eps = `noise` when non-NULL, else N(0,1) from Philox4x32-10 keyed by (seed, b*LD+i) -- ONE draw per element shared
 * by all S, as ref:359.  noise_out (optional) receives eps.                                                          */
int dic_qsample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac,
                float* out, float* noise_out, int S, int B, int LD, int step_tot, uint64_t seed, void* stream);

/* ---------------------------------------------------------------- fusion + embeddings LayerNorm (ref:299-307, hf:113-117)
 * mode 0 "concat": row t<L = x[n][t]; row L = img[n]; row L+1 = txt[n]; + seg[t>=L] + pos[t]; LayerNorm(eps);
 * mode 1 "add":    row t = x[n][t] + img[n] (+ txt[n] when add_txt[n]) + pos[t]; LayerNorm.   Tk = L+2 / L.
 * mode 2 "concat, text row dropped" (Tk = L+1): legal when no sequence is guided -- the text row is then masked as a key
 *        and its outputs are unused, so losses and gradients are unchanged.
 * Optional timestep embedding (named by BASELINE.json's north_star; the reference's forward takes no t, ref:271, so every parity
 * configuration passes temb = NULL): temb [steps][D] f32, tidx [N] int32 -- row temb[tidx[n]] is added to every token row of sequence n
 * before the LayerNorm (tidx[n] < 0: none).
 * Writes h [N][Tk][D] (dtype T, dropout p applied) and mean/rstd [N*Tk].                                          */
int dic_fuse_ln_fwd(int dtype, int mode, const float* x, const float* img, const float* txt, const uint8_t* add_txt,
                    const float* seg, const float* pos, const float* temb, const int32_t* tidx, const float* gamma, const float* beta,
                    void* h, float* mean, float* rstd, int N, int L, int D, float eps,
                    float p_drop, uint64_t seed, void* stream);
/* The same with x read in place from a larger tensor: sequence n's L rows start at x + n * x_seq_stride floats (the sampling loop feeds
 * rows [:, :L] of the previous pass's x_out [N][Tk][D] straight back in, ref:613-620, without a compaction copy).                      */
int dic_fuse_ln_fwd_x(int dtype, int mode, const float* x, int64_t x_seq_stride, const float* img, const float* txt, const uint8_t* add_txt,
                      const float* seg, const float* pos, const float* temb, const int32_t* tidx, const float* gamma, const float* beta,
                      void* h, float* mean, float* rstd, int N, int L, int D, float eps,
                      float p_drop, uint64_t seed, void* stream);
/* Backward: dh (T) -> dy [N][Tk][D] f32 (gradient wrt the pre-LN fused rows) and per-block partial sums of
 * dgamma/dbeta in `partial` [nblocks][2*D] (reduce with dic_colsum).                                              */
int dic_fuse_ln_bwd(int dtype, int mode, const float* x, const float* img, const float* txt, const uint8_t* add_txt,
                    const float* seg, const float* pos, const float* temb, const int32_t* tidx, const float* gamma,
                    const void* dh, const float* mean, const float* rstd, float* dy, float* partial, int n_partial_blocks,
                    int N, int L, int D, float p_drop, uint64_t seed, void* stream);
/* Gradient of the timestep-embedding table: dtemb[s] = sum of dy over every token row of the sequences with tidx[n] == s (all `steps`
 * rows of dtemb are written, zeros where no sequence carries that timestep).                                        */
int dic_temb_grad(const float* dy, const int32_t* tidx, int N, int Tk, int D, int steps, float* dtemb, void* stream);

/* ---------------------------------------------------------------- LayerNorm (hf:236,239,253,257; eps 1e-12) */
int dic_ln_fwd(int dtype, const void* y, const float* gamma, const float* beta, void* h, float* mean, float* rstd,
               int T, int D, float eps, void* stream);
int dic_ln_fwd_r32(const float* y32, const float* gamma, const float* beta, void* h_bf16, float* h32 /* or NULL */, float* mean, float* rstd,
                   int T, int D, float eps, void* stream);
/* dx (T) = LN backward; dx_drop (T, optional) = dx with the dropout mask of the producing GEMM epilogue applied
 * (mask keyed by (seed, row*D+col)); partial [nblocks][3*D] = {dgamma, dbeta, colsum(dx_drop if given else dx)}.       */
int dic_ln_bwd(int dtype, const void* dh, const void* y, const float* gamma, const float* mean, const float* rstd,
               void* dx, void* dx_drop, float p_drop, uint64_t seed, float* partial, int n_partial_blocks,
               int T, int D, void* stream);

/* ---------------------------------------------------------------- MLM-head GELU + LayerNorm (hf:511-512)
 * dtype | DIC_U_F32: u is fp32 (the vocab_transform GEMM wrote it with out_f32) while du stays the engine's type.  The bf16 engines set it:
 * in a denoiser whose output has (half-)collapsed onto one row -- the first few hundred steps of training -- the bf16 rounding of u is the
 * same for every token, does not average out of the batch-mean L1 terms and alone moves them by 2-4e-4 (profiles/r04_collapse_probe.txt). */
int dic_gelu_ln_fwd(int dtype, const void* u, const float* gamma, const float* beta, float* x_out,
                    float* mean, float* rstd, int T, int D, float eps, void* stream);
/* du (T) from dx_out (f32); partial [nblocks][3*D] = {dgamma, dbeta, colsum(du)}                                   */
int dic_gelu_ln_bwd(int dtype, const float* dx_out, const void* u, const float* gamma, const float* mean,
                    const float* rstd, void* du, float* partial, int n_partial_blocks, int T, int D, void* stream);

/* ---------------------------------------------------------------- attention (hf:136-147, 183-185; ref:296-297)
 * qkv [N][Tk][3*D] (q | k | v), key_mask [N][Tk] (1 = attend), ctx [N][Tk][D].  softmax(QK^T/sqrt(dh)+mask) V,
 * attention-prob dropout p keyed by (seed, ((n*H+h)*Tk+i)*Tk+j).  bf16: MFMA 32x32x16; f32: exact VALU path.     */
int dic_attn_fwd(int dtype, const void* qkv, const uint8_t* key_mask, void* ctx, int N, int Tk, int H, int dh,
                 float p_drop, uint64_t seed, void* stream);
int dic_attn_bwd(int dtype, const void* qkv, const uint8_t* key_mask, const void* dctx, void* dqkv,
                 int N, int Tk, int H, int dh, float p_drop, uint64_t seed, void* stream);

/* ---------------------------------------------------------------- TRAIN_EMBEDDING ablation (ref:98-102, 238-243, 459-468)
 * dic_te_dx0: gradient wrt the learned embedding rows x_0 [B][L][C] gathered for this batch: through q_sample into the S noised
 *   copies and the x_1 copy (dxin [S*B+B][Tk][C], rows t<L), and as the target of both embedding losses (g, same shape).
 * dic_embed_scatter: nn.Embedding backward, dE[id] = sum over the positions holding id, fixed order (ids pre-sorted, stable). */
int dic_te_dx0(const float* dxin, const float* g, const float* sqrt_ac, const int64_t* t, const int64_t* t_next /* or NULL */, int S,
               int B, int L, int Tk, int C, int step_tot, int x1_row0, float* dx0, void* stream);
int dic_embed_scatter(const int64_t* sorted_ids, const int64_t* order, const float* dx0, int n_tokens, int C, int V, float* dE,
                      void* stream);

/* ---------------------------------------------------------------- losses (ref:77-87, 418, 428)
 * kind 0 series_sum_sample_mean, 1 series_sum, 2 mse_series_mean, 3 mse_series_sum.  x_out [N][Tk][D] f32 (rows
 * t<L used), target [N or B][L][D] f32 (index n % tgt_rows).  per_seq[n] = sum|d| (kinds 0,1) or sqrt(sum d^2);
 * grad_scale[n]-weighted gradient written to dx_out rows t<L (rows t>=L zeroed) when dx_out != NULL;
 * xr [N*L][D] (dtype T) = compact copy of the rows that feed the rounding head.                                    */
int dic_emb_loss(int dtype, int kind, const float* x_out, const float* target, int tgt_rows, float* per_seq,
                 float* dx_out, const float* grad_scale, void* xr, int N, int L, int Tk, int D, void* stream);
/* dx_out[n][t<L][:] += dxr[n*L+t][:]   (adds the rounding-loss gradient)                                           */
int dic_add_rows(float* dx_out, const float* dxr, int N, int L, int Tk, int D, void* stream);
/* the same with a per-row factor: dx_out row (n, l) += inv_z[n*L + l] * scale * dxr row (n*L + l)   (inv_z as written by dic_ce_exp_combine) */
int dic_add_rows_scaled(float* dx_out, const float* dxr, const float* inv_z, float scale, int N, int L, int Tk, int D, void* stream);
/* out4[0] = scale_a*sum in[0:n_a), out4[1] = scale_b*sum in[n_a:n), out4[2] = out4[0]+out4[1] (fp64 accumulate)      */
int dic_seg_sum(const float* in, int n, int n_a, float scale_a, float scale_b, float* out4, const float* carry, void* stream);
/* (out4[0..2] as above; out4[3] = out4[2] + *carry (carry optional): the step's total l = x_t_loss + x_1_loss + prob_loss, ref:481) */

/* Step inputs of the stacked encoder batch [S*B x_t rows (s-major) | B x_1 rows] without classifier-free guidance, in one launch:
 * img_in/txt_in [N][512] = the CLIP rows repeated over S (ref:415 `.repeat((SAMPLE_SIZE,1,1))`; txt_in optional), kmask [N][Tk] =
 * [mask != 0 | 1 | 0] (ref:296-297 hstack([mask, [1,0]]); Tk = L+1 drops the text column, Tk = L is "add" fusion), addtxt [N] = 0,
 * tgt [(N)*L] = ids repeated (ref:434-437; optional), gscale [N] = scale_a for the x_t rows, scale_b for the x_1 rows (optional).      */
int dic_step_prep(const float* img, const float* txt, const int64_t* mask, const int64_t* ids, int S, int B, int L, int Tk,
                  float* img_in, float* txt_in, uint8_t* kmask, uint8_t* addtxt, int64_t* tgt, float* gscale, float scale_a, float scale_b,
                  void* stream);
/* The same for a step WITH classifier-free guidance (ref :406-415, 313-317): stacked batch [S*B x_t rows | Ng guided copies | B x_1 rows]; guided copy i
 * repeats x_t row gi[i] (device list, ascending) with the text key unmasked (Tk = L+2) / add_txt set (Tk = L).  Also copies the guided rows' noisy inputs
 * inside xin [N][L][D] (rows S*B+i <- rows gi[i]) and zero-fills rows [S*B, S*B+Ng) of dx [N][Tk][D] (optional).  tgt / gscale cover the S*B + B rows the
 * losses see.  The host draws the guidance mask (it needs Ng for the launch shapes), so a guided step has no device->host sync and no ATen kernel.   */
int dic_cfg_prep(const float* img, const float* txt, const int64_t* mask, const int64_t* ids, const int64_t* gi, int S, int B, int L, int Tk, int Ng, int D,
                 float* img_in, float* txt_in, uint8_t* kmask, uint8_t* addtxt, int64_t* tgt, float* gscale, float scale_a, float scale_b,
                 float* xin, float* dx, void* stream);
/* out[i] ~ U{0..hi-1}, Philox4x32-10 keyed by (seed, i) -- the step's shared timestep vector (ref:460-461).                          */
int dic_randint(int64_t* out, int n, int hi, uint64_t seed, void* stream);
/* zero a 16-byte aligned device range (gradient slots a backward does not write)                                                      */
int dic_zero(void* p, int64_t nbytes, void* stream);

/* ---------------------------------------------------------------- classifier-free-guidance mix (ref:313-317)
 * x_out[idx[i]] = (1+w)*g_out[i] - w*x_out[idx[i]]  (rows of Tk*D floats); backward splits the gradient.          */
int dic_cfg_mix_fwd(float* x_out, const float* g_out, const int64_t* idx, int n_g, int row_elems, float w, void* stream);
int dic_cfg_mix_bwd(float* dx_out, float* dg_out, const int64_t* idx, int n_g, int row_elems, float w, void* stream);

/* "add" fusion backward (ref:306-307): out_all[n] = sum_t in[n][t]; out_flag[n] = flags[n] ? out_all[n] : 0        */
int dic_seq_sum(const float* in, const uint8_t* flags, float* out_all, float* out_flag, int N, int L, int D, void* stream);

/* ---------------------------------------------------------------- reductions for bias / LN / embedding grads
 * out[c] (+)= sum_r in[r][c]; in dtype f32 (in_dtype 0) or bf16 (1); two-stage, deterministic; ws >= 64*cols f32 */
int dic_colsum(int in_dtype, const void* in, int rows, int cols, int ld, float* out, int accumulate, float* ws, void* stream);
/* two fp32 problems of one shape (rows <= 1024) in one launch: out0[c] = sum_r in0[r][c], out1[c] = sum_r in1[r][c] -- the two LayerNorm-backward
 * partial buffers of an encoder layer (hf:236, 253 backward: gamma / beta / bias gradients) */
int dic_colsum_pair(const float* in0, float* out0, const float* in1, float* out1, int rows, int cols, int ld, void* stream);

/* ---------------------------------------------------------------- mean-row correction of a bf16-rounded weight (hf:183-185, 201, 221-223, 510)
 * bias_eff [N] = bias [N] (or 0) + lo [N][K] . abar [K], abar = column mean of the rows 0, row_stride, 2 row_stride, ... of the bf16 input A [T][lda]
 * of an nn.Linear, lo = bf16(W - bf16(W)) (dic_adamw_hl / dic_cast_bf16_hl).  Passed to dic_gemm as `bias`, it gives the Linear the part of
 * A W_lo^T that all rows share -- the part of the weights' rounding a batch-mean loss does not average out -- for one GEMV instead of a second
 * pass of the K loop (DicGemmParams.B2).  Deterministic; ws: dic_lo_mean_bias_ws_bytes(K).                                                    */
size_t dic_lo_mean_bias_ws_bytes(int K);
int dic_lo_mean_bias(const void* A, int T, int lda, int row_stride, int K, const void* lo, int ldb, int N, const float* bias, float* bias_eff,
                     float* ws, void* stream);

/* ---------------------------------------------------------------- one-launch Linear preparation + centred residual stream (round 5)
 * The parity mode of the bf16 engine (hf:183-185, 201, 221-223, 236, 253, 510 evaluated so that no rounding is COMMON to all token rows):
 *   dic_lin_prep, in front of a forward Linear y = dropout(A W^T + b) + R (two small launches, like dic_lo_mean_bias, which it extends):
 *     abar = column mean of every row_stride-th row of A (bf16 [T][lda]); s_lo = W_lo abar, s_hi = W_hi abar (W_hi / W_lo: the bf16 halves of the fp32
 *     weight, [N][ldb]);
 *     bias_in[n]   = bias[n] + s_lo[n]                 -> DicGemmParams.bias: dic_lo_mean_bias's correction
 *     and, when y_ref is given (the residual Linears; needs w_hi):
 *     y_ref[n]     = bias[n] + s_lo[n] + s_hi[n] + r_ref[n]    the predicted mean row of the sum; the GEMM stores bf16(y - y_ref)
 *     bias_post[n] = r_ref[n] - y_ref[n]               -> DicGemmParams.bias2 (behind the dropout); fold_post = 1 (no dropout): added to bias_in instead
 *     r_ref: the reference row of the centred residual operand R_c = bf16(R - r_ref) (NULL: R is stored as it is).
 *     ws: dic_lin_prep_ws_bytes(K) bytes.  Deterministic (slabs of column sums added in a fixed order).
 *   dic_ln_fwd_cen: LayerNorm of y = y_c + y_ref (y_c bf16 [T][768], y_ref fp32 [768]) -> h = bf16(LN(y)) (optional) and / or h_c =
 *     bf16(LN(y) - h_ref) (optional) with h_ref = LN(y_ref) (fp32 [768], optional output), mean / rstd as dic_ln_fwd.  The shipped engine writes
 *     ONLY h_c: it is the residual operand of the next residual GEMM AND the MFMA operand of the next Linear, whose bias then carries W h_ref:
 *     next_lo / next_hi (optional, bf16 [next_n][next_ldb], the halves of that Linear's fp32 weight): next_bias_out[n] = next_bias[n] +
 *     (next_lo [+ next_hi])[n] . h_ref, computed inside this launch (one output per wave).  Its weight gradient needs dic_rank1_add.
 *   dic_ln_bwd_cen: dic_ln_bwd (bf16) with y given as (y_c, y_ref).                                                                             */
size_t dic_lin_prep_ws_bytes(int K);
int dic_lin_prep(const void* A, int T, int lda, int row_stride, int K, const void* w_hi, const void* w_lo, int ldb, int N, const float* bias,
                 const float* r_ref, int fold_post, float* bias_in, float* bias_post, float* y_ref, void* ws, void* stream);
int dic_ln_fwd_cen(const void* y_c, const float* y_ref, const float* gamma, const float* beta, void* h, void* h_c, float* h_ref, float* mean,
                   float* rstd, int T, int D, float eps, const void* next_hi, const void* next_lo, int next_ldb, int next_n, const float* next_bias,
                   float* next_bias_out, void* stream);
/* dW[m][n] += db[m] * x_ref[n] (fp32): completes a weight gradient whose GEMM contracted dY with the CENTRED input X_c = X - 1 x_ref^T. */
int dic_rank1_add(float* dW, const float* db, const float* x_ref, int M, int N, void* stream);
int dic_ln_bwd_cen(const void* dh, const void* y_c, const float* y_ref, const float* gamma, const float* mean, const float* rstd, void* dx,
                   void* dx_drop, float p_drop, uint64_t seed, float* partial, int n_partial_blocks, int T, int D, void* stream);

/* ---------------------------------------------------------------- AdamW (ref:335 -- torch defaults, decoupled wd on every tensor)
 * p,g,m,v flat f32 [n]; g is multiplied by grad_scale first (1/world_size after the RCCL sum);
 * shadow (bf16, optional) receives the updated parameters for the bf16 GEMM operands.                              */
int dic_adamw(float* p, const float* g, float* m, float* v, uint16_t* shadow, int64_t n, float lr, float beta1,
              float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, float grad_scale,
              void* stream);
int dic_cast_bf16(const float* in, uint16_t* out, int64_t n, void* stream);
/* Split-weight forms (DicGemmParams.B2): shadow_lo / out_lo (bf16, optional) additionally receives bf16(p - bf16(p)), so that
 * shadow + shadow_lo carries 16 mantissa bits of the fp32 master weight the reference's nn.Linear multiplies by (hf:183-185, 201, 221-223, 510). */
int dic_adamw_hl(float* p, const float* g, float* m, float* v, uint16_t* shadow, uint16_t* shadow_lo, int64_t n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, float grad_scale, void* stream);
int dic_cast_bf16_hl(const float* in, uint16_t* out, uint16_t* out_lo, int64_t n, void* stream);

/* ---------------------------------------------------------------- probe (layout self-test used by the GPU tests) */
int dic_probe_tr16(const uint16_t* in, uint16_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
