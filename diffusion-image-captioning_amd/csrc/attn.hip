// Masked multi-head self-attention of the denoiser (K6), forward and backward, for the tiny sequences of this path
// (Tk = 16+2 CLIP rows; 12 heads of dh = 64).  hf:136-147 (eager_attention_forward), hf:183-185 head split,
// ref CLIP-DDPM.py:296-297 key-padding mask.
//
// bf16 path: ONE 64-lane wave per (sequence, head), MFMA 32x32x16 with the sequence padded to one 32x32 tile.
//   * scores are computed TRANSPOSED (S^T = K.Q^T) so each lane owns one query column and 16 of its 32 keys in
//     registers: the row softmax is 16 in-register ops + one __shfl_xor(32); no LDS round trip for P.
//   * the P registers are already in the A-operand k-slot order of the following P.V MFMA if V's B-operand is read with
//     the matching key order -- which is what two ds_read_b64_tr_b16 transpose-reads of the LDS-staged V tile give.
//   * Q and K fragments are loaded straight from HBM into MFMA operand registers (16 B per lane, each byte read once);
//     only V (fwd) / K,Q,dO (bwd) are staged through LDS because they are consumed k-major.
//   * backward recomputes P from Q,K (cheaper than storing 12x18x18 probabilities per sequence), runs the score / dP
//     tiles in both orientations (query-major for dQ, key-major for dK,dV) and never leaves registers in between.
//   * 33..64 tokens (seq_len 32 + CLIP rows): the same scheme on a 2 x 2 grid of 32-token tiles (attn_*_bf16_t64).
// f32 path: exact-fp32 VALU kernel (one wave per (sequence, head), LDS-resident Q,K,V): the parity path.
#include "common.h"
#include "../../include/dic_hip.h"

namespace {

constexpr int DH = 64;
constexpr int VSTRIDE = 192;            // bytes per LDS row of a [32][64] bf16 tile (128 data + 64 pad): tr reads conflict-free
constexpr int TILE = 32 * VSTRIDE;      // 6 KB

__device__ __forceinline__ bf16x8 ld_frag_global(const bf16_t* p, bool valid) {
    i32x4 v = valid ? *(const i32x4*)p : i32x4{0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8, v);
}
// stage a [Tk][64] bf16 head slice (row stride `ld` elements) into LDS rows of VSTRIDE bytes, zero-filling rows >= Tk
__device__ __forceinline__ void stage_tile(char* lds, const bf16_t* src, int ld, int Tk, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int c = lane + 64 * j, row = c >> 3, ch = c & 7;
        i32x4 v = row < Tk ? *(const i32x4*)(src + (size_t)row * ld + ch * 8) : i32x4{0, 0, 0, 0};
        *(i32x4*)(lds + row * VSTRIDE + ch * 16) = v;
    }
}
// B-operand (k-major) fragment for MFMA step s (16 tokens) and 32-column block db, token order matching the
// register order of a transposed-score accumulator: slots 0-3 <-> tokens 16s+4hi+{0..3}, slots 4-7 <-> +8.
__device__ __forceinline__ bf16x8 tr_frag(const char* lds, int s, int db, int lane) {
    const int hi = lane >> 5, half = (lane >> 4) & 1, t = lane & 15;
    const char* p = lds + (16 * s + 4 * hi + (t >> 2)) * VSTRIDE + (db * 32 + half * 16 + 4 * (t & 3)) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(p));
    s16x4 hh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(p + 8 * VSTRIDE));
    s16x8 v = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {
    s16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (short)f2bf(a[8 * s + e]);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ int reg_tok(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }   // 32x32 C/D row of register r
// Output tiles leave through LDS.  With the token-major product as the MFMA *B* operand the accumulator is out^T[d][token]: lane =
// token, and registers 4a..4a+3 hold FOUR CONSECUTIVE d (8a + 4hi + 0..3), i.e. one 8-byte bf16 run.  The runs go into a wave-
// private LDS tile ([token][64] bf16, VSTRIDE pitch) and come back as 16-byte row chunks, so the global store is 3 fully
// coalesced instructions per [Tk][64] output instead of 32 two-byte-per-lane ones (the kernels were VMEM-issue bound).
__device__ __forceinline__ void stage_out(char* tile, const f32x16& o0, const f32x16& o1, int lane, int Tk) {
    const int q = lane & 31, hi = lane >> 5;
    if (q < Tk) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            uint2 u0, u1;
            u0.x = pack2bf(o0[4 * a], o0[4 * a + 1]); u0.y = pack2bf(o0[4 * a + 2], o0[4 * a + 3]);
            u1.x = pack2bf(o1[4 * a], o1[4 * a + 1]); u1.y = pack2bf(o1[4 * a + 2], o1[4 * a + 3]);
            *(uint2*)(tile + q * VSTRIDE + (8 * a + 4 * hi) * 2) = u0;
            *(uint2*)(tile + q * VSTRIDE + (32 + 8 * a + 4 * hi) * 2) = u1;
        }
    }
}
__device__ __forceinline__ void store_out(const char* tile, bf16_t* dst, int ld, int Tk, int lane) {
    for (int e = lane; e < Tk * 8; e += 64) {
        const int row = e >> 3, ch = e & 7;
        *(i32x4*)(dst + (size_t)row * ld + ch * 8) = *(const i32x4*)(tile + row * VSTRIDE + ch * 16);
    }
}
// Attention-probability dropout of the MFMA kernels: 16 mask bits per (head, query, key), two keys (2j, 2j+1) per 32-bit hash of the
// 32-bit index ((pair*32 + query)*16 + j) -- a lane that holds one query's keys in registers (forward, backward pass 1) hashes
// once per TWO probabilities, and nothing is 64-bit.  Forward and backward share the definition; the VALU kernels keep dropout1.
__device__ __forceinline__ unsigned attn_seed_mix(unsigned long long seed) { return hash32((unsigned)seed) ^ (unsigned)(seed >> 32) * 0x9E3779B9u; }
__device__ __forceinline__ unsigned attn_drop_hash(unsigned mix, unsigned pair, int q, int key) {
    return hash32((((pair << 5) + (unsigned)q) << 4) + (unsigned)(key >> 1) ^ mix);
}
__device__ __forceinline__ float attn_drop(float v, unsigned h, int key, unsigned thr, float inv_keep) {
    const unsigned w = (key & 1) ? (h >> 16) : (h & 0xffffu);
    return w >= thr ? v * inv_keep : 0.f;
}
__device__ __forceinline__ f32x16 zero16() { f32x16 z; for (int i = 0; i < 16; ++i) z[i] = 0.f; return z; }

// D[i][j] = sum_d X[i][d] * Y[j][d] over dh=64 with X rows as MFMA "A" (row index in registers after the MFMA) and
// Y rows as "B" (column index = lane&31).  Both loaded from HBM; rows >= Tk read as zero.
__device__ __forceinline__ f32x16 rowdot(const bf16_t* X, int ldx, const bf16_t* Y, int ldy, int Tk, int lane) {
    const int r = lane & 31, hi = lane >> 5;
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        bf16x8 a = ld_frag_global(X + (size_t)r * ldx + 16 * s + 8 * hi, r < Tk);
        bf16x8 b = ld_frag_global(Y + (size_t)r * ldy + 16 * s + 8 * hi, r < Tk);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    return acc;
}

// ------------------------------------------------------------------------------------------------ bf16 forward
__global__ __launch_bounds__(256) void attn_fwd_bf16(const bf16_t* qkv, const uint8_t* key_mask, bf16_t* ctx, int N, int Tk, int H,
                                                      float scale, float p_drop, SeedArg seed_) {
    const unsigned long long seed = seed_.resolve();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= N * H) return;
    const int n = pair / H, h = pair - n * H;
    const int ld = 3 * H * DH, Dm = H * DH;
    const bf16_t* Q = qkv + (size_t)n * Tk * ld + h * DH;
    const bf16_t* K = Q + Dm;
    const bf16_t* V = Q + 2 * Dm;
    char* vt = smem + wave * TILE;
    stage_tile(vt, V, ld, Tk, lane);

    const int q = lane & 31, hi = lane >> 5;
    // S^T[key][query]: lane = query column, register r = key reg_tok(r,hi)
    // key-padding mask: one byte load per lane, shared as a 64-bit ballot (not 16 dependent byte loads per lane)
    const unsigned long long mbits = __ballot(lane < Tk && key_mask[(size_t)n * Tk + (lane < Tk ? lane : 0)] != 0);
    f32x16 st = rowdot(K, ld, Q, ld, Tk, lane);
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = reg_tok(r, hi);
        const bool ok = (mbits >> key) & 1ull;
        st[r] = ok ? st[r] * scale : -INFINITY;
        mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = (mx == -INFINITY) ? 0.f : __expf(st[r] - mx); sum += st[r]; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    const float inv_keep = drop_inv_keep(p_drop);
    const unsigned thr = drop_thr(p_drop), mix = attn_seed_mix(seed);
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        st[r] *= inv; st[r + 1] *= inv;
        if (p_drop > 0.f) {
            const unsigned hsh = attn_drop_hash(mix, (unsigned)pair, q, reg_tok(r, hi));      // keys reg_tok(r), reg_tok(r)+1
            st[r] = (hsh & 0xffffu) >= thr ? st[r] * inv_keep : 0.f;
            st[r + 1] = (hsh >> 16) >= thr ? st[r + 1] * inv_keep : 0.f;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);   // V tile stores by this wave are complete (wave-private LDS region, no barrier needed)
    __builtin_amdgcn_wave_barrier();
    // O^T[d][query] = sum_key V[key][d] P[query][key]
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        o[db] = zero16();
#pragma unroll
        for (int s = 0; s < 2; ++s) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(vt, s, db, lane), pack8(st, s), o[db], 0, 0, 0);
    }
    stage_out(vt, o[0], o[1], lane, Tk);          // (the V tile is dead: every transpose read above has been consumed)
    store_out(vt, ctx + (size_t)n * Tk * Dm + h * DH, Dm, Tk, lane);
}

// ------------------------------------------------------------------------------------------------ bf16 backward
// Every byte of Q, K, V, dO is read from HBM exactly once, straight into MFMA operand registers (16 x 16-byte loads per
// lane, all issued before the first use): the four score-shaped products S^T = K.Q^T, S = Q.K^T, dP^T = V.dO^T and
// dP = dO.V^T are the same register fragments with the A/B roles swapped.  K, Q and dO are additionally written from those
// registers into padded LDS tiles ((Tk+1 rounded up to 4) rows, the last one zero) for the three products that consume
// them k-major (dQ = dS.K, dK = dS^T.Q, dV = P^T.dO) through ds_read_b64_tr_b16; token rows beyond Tk are redirected to the
// zero row, so a wave needs 3 x 3.8 KB of LDS at Tk = 18 and three workgroups fit a CU.
__device__ __forceinline__ bf16x8 tr_frag_clamped(const char* lds, int s, int db, int lane, int zero_row) {
    const int hi = lane >> 5, half = (lane >> 4) & 1, t = lane & 15;
    const int r0 = min(16 * s + 4 * hi + (t >> 2), zero_row), r1 = min(16 * s + 8 + 4 * hi + (t >> 2), zero_row);
    const int col = (db * 32 + half * 16 + 4 * (t & 3)) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + r0 * VSTRIDE + col));
    s16x4 hh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + r1 * VSTRIDE + col));
    s16x8 v = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
struct RowFrags { bf16x8 f[4]; };
__device__ __forceinline__ RowFrags load_rows(const bf16_t* X, int ld, int Tk, int lane) {
    const int r = lane & 31, hi = lane >> 5;
    RowFrags o;
#pragma unroll
    for (int s = 0; s < 4; ++s) o.f[s] = ld_frag_global(X + (size_t)r * ld + 16 * s + 8 * hi, r < Tk);
    return o;
}
__device__ __forceinline__ void store_rows(char* tile, const RowFrags& x, int Tk, int lane) {
    const int r = lane & 31, hi = lane >> 5;
    if (r <= Tk && r < 32) {                 // row Tk holds zeros (its fragment was loaded as zero): the redirect target
#pragma unroll
        for (int s = 0; s < 4; ++s) *(i32x4*)(tile + r * VSTRIDE + 32 * s + 16 * hi) = __builtin_bit_cast(i32x4, x.f[s]);
    }
}
__device__ __forceinline__ f32x16 rowdot_reg(const RowFrags& a, const RowFrags& b) {   // D[i][j] = sum_d a_row_i . b_row_j
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.f[s], b.f[s], acc, 0, 0, 0);
    return acc;
}

__global__ __launch_bounds__(256, 3) void attn_bwd_bf16(const bf16_t* qkv, const uint8_t* key_mask, const bf16_t* dctx, bf16_t* dqkv, int N, int Tk,
                                                      int H, float scale, float p_drop, SeedArg seed_, int tile_bytes) {
    const unsigned long long seed = seed_.resolve();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= N * H) return;
    const int n = pair / H, h = pair - n * H;
    const int ld = 3 * H * DH, Dm = H * DH;
    const bf16_t* Q = qkv + (size_t)n * Tk * ld + h * DH;
    const bf16_t* dO = dctx + (size_t)n * Tk * Dm + h * DH;
    bf16_t* dQ = dqkv + (size_t)n * Tk * ld + h * DH;
    bf16_t* dK = dQ + Dm;
    bf16_t* dV = dQ + 2 * Dm;
    char* base = smem + wave * (3 * tile_bytes + 512);
    char* kt = base;
    char* qt = base + tile_bytes;
    char* dot = base + 2 * tile_bytes;
    float* stats = (float*)(base + 3 * tile_bytes);       // [0..31] row max, [32..63] 1/rowsum, [64..95] delta
    const int zero_row = Tk < 32 ? Tk : 31;

    const RowFrags fq = load_rows(Q, ld, Tk, lane), fk = load_rows(Q + Dm, ld, Tk, lane), fv = load_rows(Q + 2 * Dm, ld, Tk, lane),
                   fo = load_rows(dO, Dm, Tk, lane);
    store_rows(kt, fk, Tk, lane);
    store_rows(qt, fq, Tk, lane);
    store_rows(dot, fo, Tk, lane);

    const int c = lane & 31, hi = lane >> 5;
    const float inv_keep = drop_inv_keep(p_drop);
    const unsigned thr = drop_thr(p_drop), mix = attn_seed_mix(seed);
    const unsigned long long mbits = __ballot(lane < Tk && key_mask[(size_t)n * Tk + (lane < Tk ? lane : 0)] != 0);

    // ---- pass 1 (query-major): lane = query c, registers = keys.  P, delta, dS -> dQ
    {
        f32x16 st = rowdot_reg(fk, fq);
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = reg_tok(r, hi);
            const bool ok = (mbits >> key) & 1ull;
            st[r] = ok ? st[r] * scale : -INFINITY;
            mx = fmaxf(mx, st[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = (mx == -INFINITY) ? 0.f : __expf(st[r] - mx); sum += st[r]; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        f32x16 dpt = rowdot_reg(fv, fo);                                           // dPd^T[key][query]
        float delta = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            st[r] *= inv; st[r + 1] *= inv;                                        // P
            if (p_drop > 0.f) {
                const unsigned hsh = attn_drop_hash(mix, (unsigned)pair, c, reg_tok(r, hi));
                dpt[r] = (hsh & 0xffffu) >= thr ? dpt[r] * inv_keep : 0.f;
                dpt[r + 1] = (hsh >> 16) >= thr ? dpt[r + 1] * inv_keep : 0.f;
            }
            delta += dpt[r] * st[r] + dpt[r + 1] * st[r + 1];
        }
        delta += __shfl_xor(delta, 32, 64);
        if (hi == 0) { stats[c] = mx; stats[32 + c] = inv; stats[64 + c] = delta; }
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = st[r] * (dpt[r] - delta) * scale;      // dS[query][key] (scaled for dQ/dK)
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {                                            // dQ^T[d][query] = sum_key K[key][d] dS[query][key]
            o[db] = zero16();
#pragma unroll
            for (int s = 0; s < 2; ++s) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_clamped(kt, s, db, lane, zero_row), pack8(st, s), o[db], 0, 0, 0);
        }
        stage_out(kt, o[0], o[1], lane, Tk);                                        // K's tile is not read again
        store_out(kt, dQ, ld, Tk, lane);
    }
    // ---- pass 2 (key-major): lane = key c, registers = queries.  Pd -> dV, dS -> dK
    {
        f32x16 s2 = rowdot_reg(fq, fk);                                             // S[query(reg)][key(lane)]
        f32x16 dp2 = rowdot_reg(fo, fv);                                            // dPd[query][key]
        const bool kok = (mbits >> c) & 1ull;
        f32x16 pd, ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = reg_tok(r, hi);
            const float m_ = stats[qi], inv = stats[32 + qi], delta = stats[64 + qi];
            float pr = (kok && m_ > -INFINITY) ? __expf(s2[r] * scale - m_) * inv : 0.f;
            float dpr = dp2[r], pdr = pr;
            if (p_drop > 0.f) {
                const unsigned hsh = attn_drop_hash(mix, (unsigned)pair, qi, c);
                dpr = attn_drop(dpr, hsh, c, thr, inv_keep);
                pdr = attn_drop(pr, hsh, c, thr, inv_keep);
            }
            pd[r] = pdr;
            ds[r] = pr * (dpr - delta) * scale;
        }
        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {                                            // dV^T[d][key] = sum_query dO[query][d] Pd[query][key]
            o[db] = zero16();
#pragma unroll
            for (int s = 0; s < 2; ++s) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_clamped(dot, s, db, lane, zero_row), pack8(pd, s), o[db], 0, 0, 0);
        }
        stage_out(dot, o[0], o[1], lane, Tk);
        store_out(dot, dV, ld, Tk, lane);
#pragma unroll
        for (int db = 0; db < 2; ++db) {                                            // dK^T[d][key] = sum_query Q[query][d] dS[query][key]
            o[db] = zero16();
#pragma unroll
            for (int s = 0; s < 2; ++s) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_clamped(qt, s, db, lane, zero_row), pack8(ds, s), o[db], 0, 0, 0);
        }
        stage_out(qt, o[0], o[1], lane, Tk);
        store_out(qt, dK, ld, Tk, lane);
    }
}

// ------------------------------------------------------------------------------------------------ bf16, 33..64 tokens (seq_len 32 + CLIP rows)
// The same scheme on a 2 x 2 grid of 32-token tiles: ONE wave per (sequence, head), scores transposed so that a lane owns one query and
// its keys are registers (two accumulator tiles = 32 keys per half-wave), softmax in registers + one __shfl_xor(32), P never stored,
// P.V through ds_read_b64_tr_b16 fragments of the LDS-staged V.  Queries are processed one 32-token tile at a time; K fragments stay in
// registers across both, Q (and in the backward dO) tiles are re-read per use (8 KB per operand per head: L1/L2 hits).
// Dropout index: ((pair * 64 + query) * 32 + key/2) -- 32 bits up to 2^21 (sequence, head) pairs.
__device__ __forceinline__ unsigned attn_drop_hash64(unsigned mix, unsigned pair, int q, int key) {
    return hash32((((pair << 6) + (unsigned)q) << 5) + (unsigned)(key >> 1) ^ mix);
}
__device__ __forceinline__ RowFrags load_rows_at(const bf16_t* X, int ld, int row0, int Tk, int lane) {   // rows row0 .. row0+31 of X, zero beyond Tk
    const int r = row0 + (lane & 31), hi = lane >> 5;
    RowFrags o;
#pragma unroll
    for (int s = 0; s < 4; ++s) o.f[s] = ld_frag_global(X + (size_t)r * ld + 16 * s + 8 * hi, r < Tk);
    return o;
}
__device__ __forceinline__ void store_rows_at(char* tile, const RowFrags& x, int row0, int Tk, int lane) {   // rows <= Tk (row Tk = the zero row)
    const int r = row0 + (lane & 31), hi = lane >> 5;
    if (r <= Tk && r < 64) {
#pragma unroll
        for (int s = 0; s < 4; ++s) *(i32x4*)(tile + r * VSTRIDE + 32 * s + 16 * hi) = __builtin_bit_cast(i32x4, x.f[s]);
    }
}
// one 32-token output tile (rows row0 ..) through a wave-private LDS tile, as stage_out / store_out
__device__ __forceinline__ void put_out(char* tile, const f32x16& o0, const f32x16& o1, bf16_t* dst, int ld, int row0, int Tk, int lane) {
    __builtin_amdgcn_s_waitcnt(0);            // the previous contents of `tile` have been read (wave-private: program order + this wait)
    __builtin_amdgcn_wave_barrier();
    stage_out(tile, o0, o1, lane, Tk - row0 < 32 ? Tk - row0 : 32);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    store_out(tile, dst + (size_t)row0 * ld, ld, Tk - row0 < 32 ? Tk - row0 : 32, lane);
}

__global__ __launch_bounds__(128) void attn_fwd_bf16_t64(const bf16_t* qkv, const uint8_t* key_mask, bf16_t* ctx, int N, int Tk, int H,
                                                         float scale, float p_drop, SeedArg seed_) {
    const unsigned long long seed = seed_.resolve();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 2 + wave;
    if (pair >= N * H) return;
    const int n = pair / H, h = pair - n * H;
    const int ld = 3 * H * DH, Dm = H * DH;
    const bf16_t* Q = qkv + (size_t)n * Tk * ld + h * DH;
    const bf16_t* K = Q + Dm;
    const bf16_t* V = Q + 2 * Dm;
    char* vt = smem + wave * (3 * TILE);          // [64][VSTRIDE] V tile + one 32-row output tile
    char* ot = vt + 2 * TILE;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = lane + 64 * j, row = c >> 3, ch = c & 7;
        *(i32x4*)(vt + row * VSTRIDE + ch * 16) = row < Tk ? *(const i32x4*)(V + (size_t)row * ld + ch * 8) : i32x4{0, 0, 0, 0};
    }
    const int ql = lane & 31, hi = lane >> 5;
    const unsigned long long mbits = __ballot(lane < Tk && key_mask[(size_t)n * Tk + (lane < Tk ? lane : 0)] != 0);
    const RowFrags fk0 = load_rows_at(K, ld, 0, Tk, lane), fk1 = load_rows_at(K, ld, 32, Tk, lane);
    const float inv_keep = drop_inv_keep(p_drop);
    const unsigned thr = drop_thr(p_drop), mix = attn_seed_mix(seed);
    const int nqt = Tk > 32 ? 2 : 1;
    for (int qt = 0; qt < nqt; ++qt) {
        const RowFrags fq = load_rows_at(Q, ld, 32 * qt, Tk, lane);
        f32x16 st[2] = {rowdot_reg(fk0, fq), rowdot_reg(fk1, fq)};       // S^T[key][query]: lane = query, register r of tile kt = key 32 kt + reg_tok(r, hi)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = (mbits >> (32 * kt + reg_tok(r, hi))) & 1ull;
                st[kt][r] = ok ? st[kt][r] * scale : -INFINITY;
                mx = fmaxf(mx, st[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[kt][r] = (mx == -INFINITY) ? 0.f : __expf(st[kt][r] - mx); sum += st[kt][r]; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                st[kt][r] *= inv; st[kt][r + 1] *= inv;
                if (p_drop > 0.f) {
                    const unsigned hsh = attn_drop_hash64(mix, (unsigned)pair, 32 * qt + ql, 32 * kt + reg_tok(r, hi));
                    st[kt][r] = (hsh & 0xffffu) >= thr ? st[kt][r] * inv_keep : 0.f;
                    st[kt][r + 1] = (hsh >> 16) >= thr ? st[kt][r + 1] * inv_keep : 0.f;
                }
            }
        __builtin_amdgcn_s_waitcnt(0);   // V tile stores by this wave are complete (wave-private LDS region, no barrier needed)
        __builtin_amdgcn_wave_barrier();
        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {                                 // O^T[d][query] = sum_key V[key][d] P[query][key], 4 steps of 16 keys
            o[db] = zero16();
#pragma unroll
            for (int s = 0; s < 4; ++s) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(vt, s, db, lane), pack8(st[s >> 1], s & 1), o[db], 0, 0, 0);
        }
        put_out(ot, o[0], o[1], ctx + (size_t)n * Tk * Dm + h * DH, Dm, 32 * qt, Tk, lane);
    }
}

// Backward, 33..64 tokens.  K, Q, dO are staged in LDS (Tk + 1 rows, the last one zero: the redirect target of token rows >= Tk) for the
// k-major products; the score-shaped products take row fragments -- K's and V's stay in registers from the one global read, Q's and dO's come
// from the tiles.  Pass 1 per query tile (lane = query): P, delta, dS -> dQ.  Pass 2 per key tile (lane = key), accumulating over the query
// tiles: Pd -> dV, dS -> dK.  LDS is what bounds residency (one wave per workgroup): three tiles + the row statistics = 21.5 KB at 34 tokens,
// seven waves per CU (a first version also staged V and kept a separate output tile: 34.5 KB, four waves per CU, and the kernel is a chain of
// dependent load / LDS / MFMA latencies that only other waves can hide).  Outputs are staged through K's tile once its last transposed read is done.
__device__ __forceinline__ bf16x8 tr_frag_clamped64(const char* lds, int s, int db, int lane, int zero_row) {
    const int hi = lane >> 5, half = (lane >> 4) & 1, t = lane & 15;
    const int r0 = min(16 * s + 4 * hi + (t >> 2), zero_row), r1 = min(16 * s + 8 + 4 * hi + (t >> 2), zero_row);
    const int col = (db * 32 + half * 16 + 4 * (t & 3)) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + r0 * VSTRIDE + col));
    s16x4 hh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + r1 * VSTRIDE + col));
    s16x8 v = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// R1: registers of the SECOND 32-token tile that can hold a valid token (register r of a lane half covers tokens (r&3) + 8 (r>>2) + 4 hi): 4 up to
// 40 tokens, 8 up to 48, 16 beyond.  Everything past them is padding whose probabilities are exactly zero, so the softmax / dropout / dS arithmetic
// on those registers and the MFMA k-steps made only of them are skipped at compile time (34 tokens = config 5: 20 of 32 score registers per lane
// and 3 of 4 k-steps remain; the kernel is VALU-bound on exactly that arithmetic).
template <int R1>
__global__ __launch_bounds__(64, 2) void attn_bwd_bf16_t64(const bf16_t* qkv, const uint8_t* key_mask, const bf16_t* dctx, bf16_t* dqkv, int N, int Tk,
                                                         int H, float scale, float p_drop, SeedArg seed_, int tile_bytes) {
    const unsigned long long seed = seed_.resolve();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x;       // one wave per workgroup: LDS (27 KB at 34 tokens) is what bounds residency, five of these fit a CU
    if (pair >= N * H) return;
    const int n = pair / H, h = pair - n * H;
    const int ld = 3 * H * DH, Dm = H * DH;
    const bf16_t* Q = qkv + (size_t)n * Tk * ld + h * DH;
    const bf16_t* K = Q + Dm;
    const bf16_t* V = Q + 2 * Dm;
    const bf16_t* dO = dctx + (size_t)n * Tk * Dm + h * DH;
    bf16_t* dQ = dqkv + (size_t)n * Tk * ld + h * DH;
    bf16_t* dK = dQ + Dm;
    bf16_t* dV = dQ + 2 * Dm;
    char* base = smem;
    char* kt_ = base;
    char* qt_ = base + tile_bytes;
    char* dot = base + 2 * tile_bytes;
    char* ot = kt_;                                        // 32-row output staging: K's tile, after pass 1's last transposed read of it
    float* stats = (float*)(base + 3 * tile_bytes);        // [0..63] row max, [64..127] 1/rowsum, [128..191] delta
    const int zero_row = Tk < 64 ? Tk : 63;
    const int c = lane & 31, hi = lane >> 5;
    const float inv_keep = drop_inv_keep(p_drop);
    const unsigned thr = drop_thr(p_drop), mix = attn_seed_mix(seed);
    const unsigned long long mbits = __ballot(lane < Tk && key_mask[(size_t)n * Tk + (lane < Tk ? lane : 0)] != 0);
    // the only HBM reads of the kernel, all in flight together: K and V row fragments stay in registers, K / Q / dO go to their tiles (rows
    // 0 .. Tk, row Tk zero).  Every later operand -- Q / dO row fragments for the score-shaped products, transpose-read fragments for the
    // k-major ones -- comes from these tiles (a first version re-read row fragments from HBM/L2 in every phase: ~26 dependent load phases
    // per head, 1.3 TB/s).
    const RowFrags fk0 = load_rows_at(K, ld, 0, Tk, lane), fk1 = load_rows_at(K, ld, 32, Tk, lane);
    const RowFrags fv0 = load_rows_at(V, ld, 0, Tk, lane), fv1 = load_rows_at(V, ld, 32, Tk, lane);
    {
        const RowFrags b0 = load_rows_at(Q, ld, 0, Tk, lane), b1 = load_rows_at(Q, ld, 32, Tk, lane), c0 = load_rows_at(dO, Dm, 0, Tk, lane),
                       c1 = load_rows_at(dO, Dm, 32, Tk, lane);
        store_rows_at(kt_, fk0, 0, Tk, lane); store_rows_at(kt_, fk1, 32, Tk, lane);
        store_rows_at(qt_, b0, 0, Tk, lane); store_rows_at(qt_, b1, 32, Tk, lane);
        store_rows_at(dot, c0, 0, Tk, lane); store_rows_at(dot, c1, 32, Tk, lane);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
    }
    auto rows_lds = [&](const char* tile, int row0) {       // row fragments of rows row0 .. row0+31 (rows >= Tk: the zero row)
        const int r = min(row0 + (lane & 31), zero_row);
        RowFrags o;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) o.f[s2] = __builtin_bit_cast(bf16x8, *(const i32x4*)(tile + r * VSTRIDE + 32 * s2 + 16 * hi));
        return o;
    };
    // ---- pass 1 (query-major), one query tile at a time
    {
        f32x16 oq[2][2];                               // dQ of both query tiles: staged through K's tile after the last read of it
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {               // (this kernel only runs beyond 32 tokens: always two tiles)
            const RowFrags fq = rows_lds(qt_, 32 * qt), fo = rows_lds(dot, 32 * qt);
            f32x16 st[2] = {rowdot_reg(fk0, fq), rowdot_reg(fk1, fq)};
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (kt == 1 && r >= R1) { st[kt][r] = 0.f; continue; }
                    const bool ok = (mbits >> (32 * kt + reg_tok(r, hi))) & 1ull;
                    st[kt][r] = ok ? st[kt][r] * scale : -INFINITY;
                    mx = fmaxf(mx, st[kt][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { if (kt == 1 && r >= R1) continue; st[kt][r] = (mx == -INFINITY) ? 0.f : __expf(st[kt][r] - mx); sum += st[kt][r]; }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = sum > 0.f ? 1.0f / sum : 0.f;
            f32x16 dpt[2] = {rowdot_reg(fv0, fo), rowdot_reg(fv1, fo)};                 // dPd^T[key][query]
            float delta = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    if (kt == 1 && r >= R1) continue;
                    st[kt][r] *= inv; st[kt][r + 1] *= inv;                             // P
                    if (p_drop > 0.f) {
                        const unsigned hsh = attn_drop_hash64(mix, (unsigned)pair, 32 * qt + c, 32 * kt + reg_tok(r, hi));
                        dpt[kt][r] = (hsh & 0xffffu) >= thr ? dpt[kt][r] * inv_keep : 0.f;
                        dpt[kt][r + 1] = (hsh >> 16) >= thr ? dpt[kt][r + 1] * inv_keep : 0.f;
                    }
                    delta += dpt[kt][r] * st[kt][r] + dpt[kt][r + 1] * st[kt][r + 1];
                }
            delta += __shfl_xor(delta, 32, 64);
            if (hi == 0) { stats[32 * qt + c] = mx; stats[64 + 32 * qt + c] = inv; stats[128 + 32 * qt + c] = delta; }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { if (kt == 1 && r >= R1) continue; st[kt][r] = st[kt][r] * (dpt[kt][r] - delta) * scale; }      // dS[query][key]
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int db = 0; db < 2; ++db) {                                            // dQ^T[d][query] = sum_key K[key][d] dS[query][key]
                oq[qt][db] = zero16();
#pragma unroll
                for (int s = 0; s < (R1 > 8 ? 4 : 3); ++s)          // 16-key steps; the last one is all padding up to 48 tokens
                    oq[qt][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_clamped64(kt_, s, db, lane, zero_row), pack8(st[s >> 1], s & 1), oq[qt][db], 0, 0, 0);
            }
        }
        put_out(ot, oq[0][0], oq[0][1], dQ, ld, 0, Tk, lane);
        put_out(ot, oq[1][0], oq[1][1], dQ, ld, 32, Tk, lane);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // ---- pass 2 (key-major), one key tile at a time, accumulating over the query tiles
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const RowFrags& fk = kt ? fk1 : fk0;
        const RowFrags& fv = kt ? fv1 : fv0;
        const int key = 32 * kt + c;
        const bool kok = (mbits >> key) & 1ull;
        f32x16 ov[2] = {zero16(), zero16()}, ok_[2] = {zero16(), zero16()};
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {              // (this kernel only runs beyond 32 tokens: always two query tiles; unrolled so that the
                                                      //  padding tests below are compile-time)
            const RowFrags fq = rows_lds(qt_, 32 * qt), fo = rows_lds(dot, 32 * qt);
            f32x16 s2 = rowdot_reg(fq, fk);                                             // S[query(reg)][key(lane)]
            f32x16 dp2 = rowdot_reg(fo, fv);                                            // dPd[query][key]
            f32x16 pd, ds;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (qt == 1 && r >= R1) { pd[r] = 0.f; ds[r] = 0.f; continue; }
                const int qi = 32 * qt + reg_tok(r, hi);
                const float m_ = stats[qi], inv = stats[64 + qi], delta = stats[128 + qi];
                float pr = (kok && m_ > -INFINITY) ? __expf(s2[r] * scale - m_) * inv : 0.f;
                float dpr = dp2[r], pdr = pr;
                if (p_drop > 0.f) {
                    const unsigned hsh = attn_drop_hash64(mix, (unsigned)pair, qi, key);
                    dpr = attn_drop(dpr, hsh, key, thr, inv_keep);
                    pdr = attn_drop(pr, hsh, key, thr, inv_keep);
                }
                pd[r] = pdr;
                ds[r] = pr * (dpr - delta) * scale;
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int s = 0; s < ((qt == 1 && R1 <= 8) ? 1 : 2); ++s) {       // queries 32 qt + 16 s ..: dV^T[d][key] += dO[q][d] Pd[q][key];  dK^T[d][key] += Q[q][d] dS[q][key]
                    ov[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_clamped64(dot, 2 * qt + s, db, lane, zero_row), pack8(pd, s), ov[db], 0, 0, 0);
                    ok_[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_clamped64(qt_, 2 * qt + s, db, lane, zero_row), pack8(ds, s), ok_[db], 0, 0, 0);
                }
        }
        put_out(ot, ov[0], ov[1], dV, ld, 32 * kt, Tk, lane);
        put_out(ot, ok_[0], ok_[1], dK, ld, 32 * kt, Tk, lane);
    }
}

// ------------------------------------------------------------------------------------------------ f32 VALU path
constexpr int TMAX = 64, PADW = 65;
template <typename T>
__global__ __launch_bounds__(64) void attn_fwd_f32(const T* qkv, const uint8_t* key_mask, T* ctx, int N, int Tk, int H, float scale,
                                                    float p_drop, SeedArg seed_) {
    const unsigned long long seed = seed_.resolve();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* q = (float*)smem;
    float* k = q + Tk * PADW;
    float* v = k + Tk * PADW;
    float* p = v + Tk * PADW;          // [Tk][Tk+1]
    const int lane = threadIdx.x, pair = blockIdx.x, n = pair / H, h = pair - n * H;
    const int ld = 3 * H * DH, Dm = H * DH, PW = Tk + 1;
    const T* src = qkv + (size_t)n * Tk * ld + h * DH;
    for (int i = lane; i < Tk * DH; i += 64) {
        int r = i >> 6, d = i & 63;
        q[r * PADW + d] = Elem<T>::ld(src + (size_t)r * ld + d);
        k[r * PADW + d] = Elem<T>::ld(src + (size_t)r * ld + Dm + d);
        v[r * PADW + d] = Elem<T>::ld(src + (size_t)r * ld + 2 * Dm + d);
    }
    __syncthreads();
    for (int e = lane; e < Tk * Tk; e += 64) {
        int i = e / Tk, j = e - i * Tk;
        float s = 0.f;
        for (int d = 0; d < DH; ++d) s = fmaf(q[i * PADW + d], k[j * PADW + d], s);
        p[i * PW + j] = key_mask[(size_t)n * Tk + j] ? s * scale : -INFINITY;
    }
    __syncthreads();
    const float inv_keep = drop_inv_keep(p_drop);
    for (int i = lane; i < Tk; i += 64) {
        float mx = -INFINITY;
        for (int j = 0; j < Tk; ++j) mx = fmaxf(mx, p[i * PW + j]);
        float sum = 0.f;
        for (int j = 0; j < Tk; ++j) { float e = (mx == -INFINITY) ? 0.f : expf(p[i * PW + j] - mx); p[i * PW + j] = e; sum += e; }
        float inv = sum > 0.f ? 1.f / sum : 0.f;
        for (int j = 0; j < Tk; ++j) {
            float pr = p[i * PW + j] * inv;
            if (p_drop > 0.f) pr = dropout1(pr, seed, ((unsigned long long)pair * Tk + i) * Tk + j, p_drop, inv_keep);
            p[i * PW + j] = pr;
        }
    }
    __syncthreads();
    for (int i = 0; i < Tk; ++i) {
        float o = 0.f;
        for (int j = 0; j < Tk; ++j) o = fmaf(p[i * PW + j], v[j * PADW + lane], o);
        Elem<T>::st(ctx + ((size_t)n * Tk + i) * Dm + h * DH + lane, o);
    }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_f32(const T* qkv, const uint8_t* key_mask, const T* dctx, T* dqkv, int N, int Tk, int H,
                                                    float scale, float p_drop, SeedArg seed_) {
    const unsigned long long seed = seed_.resolve();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* q = (float*)smem;
    float* k = q + Tk * PADW;
    float* v = k + Tk * PADW;
    float* go = v + Tk * PADW;
    float* p = go + Tk * PADW;         // P (no dropout)      [Tk][Tk+1]
    float* ds = p + Tk * (Tk + 1);     // dPd then dS         [Tk][Tk+1]
    const int lane = threadIdx.x, pair = blockIdx.x, n = pair / H, h = pair - n * H;
    const int ld = 3 * H * DH, Dm = H * DH, PW = Tk + 1;
    const T* src = qkv + (size_t)n * Tk * ld + h * DH;
    const T* gsrc = dctx + (size_t)n * Tk * Dm + h * DH;
    for (int i = lane; i < Tk * DH; i += 64) {
        int r = i >> 6, d = i & 63;
        q[r * PADW + d] = Elem<T>::ld(src + (size_t)r * ld + d);
        k[r * PADW + d] = Elem<T>::ld(src + (size_t)r * ld + Dm + d);
        v[r * PADW + d] = Elem<T>::ld(src + (size_t)r * ld + 2 * Dm + d);
        go[r * PADW + d] = Elem<T>::ld(gsrc + (size_t)r * Dm + d);
    }
    __syncthreads();
    for (int e = lane; e < Tk * Tk; e += 64) {
        int i = e / Tk, j = e - i * Tk;
        float s = 0.f, g = 0.f;
        for (int d = 0; d < DH; ++d) { s = fmaf(q[i * PADW + d], k[j * PADW + d], s); g = fmaf(go[i * PADW + d], v[j * PADW + d], g); }
        p[i * PW + j] = key_mask[(size_t)n * Tk + j] ? s * scale : -INFINITY;
        ds[i * PW + j] = g;
    }
    __syncthreads();
    const float inv_keep = drop_inv_keep(p_drop);
    for (int i = lane; i < Tk; i += 64) {
        float mx = -INFINITY;
        for (int j = 0; j < Tk; ++j) mx = fmaxf(mx, p[i * PW + j]);
        float sum = 0.f;
        for (int j = 0; j < Tk; ++j) { float e = (mx == -INFINITY) ? 0.f : expf(p[i * PW + j] - mx); p[i * PW + j] = e; sum += e; }
        float inv = sum > 0.f ? 1.f / sum : 0.f;
        float delta = 0.f;
        for (int j = 0; j < Tk; ++j) {
            float pr = p[i * PW + j] * inv;
            float dp = ds[i * PW + j];
            if (p_drop > 0.f) dp = dropout1(dp, seed, ((unsigned long long)pair * Tk + i) * Tk + j, p_drop, inv_keep);
            p[i * PW + j] = pr;
            ds[i * PW + j] = dp;
            delta += dp * pr;
        }
        for (int j = 0; j < Tk; ++j) ds[i * PW + j] = p[i * PW + j] * (ds[i * PW + j] - delta) * scale;
    }
    __syncthreads();
    T* dst = dqkv + (size_t)n * Tk * ld + h * DH;
    for (int i = 0; i < Tk; ++i) {       // dQ[i][lane]
        float o = 0.f;
        for (int j = 0; j < Tk; ++j) o = fmaf(ds[i * PW + j], k[j * PADW + lane], o);
        Elem<T>::st(dst + (size_t)i * ld + lane, o);
    }
    for (int j = 0; j < Tk; ++j) {       // dK[j][lane], dV[j][lane]
        float ok_ = 0.f, ov = 0.f;
        for (int i = 0; i < Tk; ++i) {
            ok_ = fmaf(ds[i * PW + j], q[i * PADW + lane], ok_);
            float pd = p[i * PW + j];
            if (p_drop > 0.f) pd = dropout1(pd, seed, ((unsigned long long)pair * Tk + i) * Tk + j, p_drop, inv_keep);
            ov = fmaf(pd, go[i * PADW + lane], ov);
        }
        Elem<T>::st(dst + (size_t)j * ld + Dm + lane, ok_);
        Elem<T>::st(dst + (size_t)j * ld + 2 * Dm + lane, ov);
    }
}

}  // namespace

extern "C" int dic_attn_fwd(int dtype, const void* qkv, const uint8_t* key_mask, void* ctx, int N, int Tk, int H, int dh, float p_drop,
                            uint64_t seed, void* stream) {
    DIC_REQUIRE(dh == DH && N > 0 && H > 0, "dic_attn: head dim must be 64");
    const float scale = 0.125f;
    hipStream_t st = (hipStream_t)stream;
    DIC_REQUIRE(Tk <= TMAX, "dic_attn: at most 64 tokens per sequence");
    DIC_REQUIRE((long long)N * H < (1ll << 23), "dic_attn: at most 2^23 (sequence, head) pairs per launch (32-bit dropout index)");
    if (dtype == DIC_BF16 && Tk > 32) {      // seq_len 32 + CLIP rows: the 2 x 2-tile MFMA kernel, two (sequence, head) pairs per workgroup
        DIC_REQUIRE((long long)N * H < (1ll << 21), "dic_attn: at most 2^21 (sequence, head) pairs per launch beyond 32 tokens (32-bit dropout index)");
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)attn_fwd_bf16_t64, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 3 * TILE); attr = true; }
        hipLaunchKernelGGL(attn_fwd_bf16_t64, dim3((N * H + 1) / 2), dim3(128), 2 * 3 * TILE, st, (const bf16_t*)qkv, key_mask, (bf16_t*)ctx, N, Tk, H, scale, p_drop, make_seed(seed, DIC_STRIDE_DROP));
    } else if (dtype == DIC_BF16) {
        hipLaunchKernelGGL(attn_fwd_bf16, dim3((N * H + 3) / 4), dim3(256), 4 * TILE, st, (const bf16_t*)qkv, key_mask, (bf16_t*)ctx, N, Tk, H, scale, p_drop, make_seed(seed, DIC_STRIDE_DROP));
    } else {
        size_t lds = (size_t)(3 * Tk * PADW + Tk * (Tk + 1)) * sizeof(float);
        hipLaunchKernelGGL(attn_fwd_f32<float>, dim3(N * H), dim3(64), lds, st, (const float*)qkv, key_mask, (float*)ctx, N, Tk, H, scale, p_drop, make_seed(seed, DIC_STRIDE_DROP));
    }
    DIC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dic_attn_bwd(int dtype, const void* qkv, const uint8_t* key_mask, const void* dctx, void* dqkv, int N, int Tk, int H, int dh,
                            float p_drop, uint64_t seed, void* stream) {
    DIC_REQUIRE(dh == DH && N > 0 && H > 0, "dic_attn: head dim must be 64");
    const float scale = 0.125f;
    hipStream_t st = (hipStream_t)stream;
    DIC_REQUIRE(Tk <= TMAX, "dic_attn: at most 64 tokens per sequence");
    DIC_REQUIRE((long long)N * H < (1ll << 23), "dic_attn: at most 2^23 (sequence, head) pairs per launch (32-bit dropout index)");
    if (dtype == DIC_BF16 && Tk > 32) {
        DIC_REQUIRE((long long)N * H < (1ll << 21), "dic_attn: at most 2^21 (sequence, head) pairs per launch beyond 32 tokens (32-bit dropout index)");
        const int rows = Tk < 64 ? ((Tk + 1 + 3) & ~3) : 64;          // valid rows + one zero row
        const int tile_bytes = rows * VSTRIDE;
        const size_t lds = (size_t)(3 * tile_bytes + 768);
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(N * H), dim3(64), lds, st, (const bf16_t*)qkv, key_mask, (const bf16_t*)dctx, (bf16_t*)dqkv, N, Tk, H, scale, p_drop, make_seed(seed, DIC_STRIDE_DROP), tile_bytes);
        };
        if (Tk <= 40) go(attn_bwd_bf16_t64<4>); else if (Tk <= 48) go(attn_bwd_bf16_t64<8>); else go(attn_bwd_bf16_t64<16>);
    } else if (dtype == DIC_BF16) {
        const int rows = Tk < 32 ? ((Tk + 1 + 3) & ~3) : 32;          // valid rows + one zero row
        const int tile_bytes = rows * VSTRIDE;
        size_t lds = 4 * (size_t)(3 * tile_bytes + 512);
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)attn_bwd_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (3 * TILE + 512)); attr = true; }
        hipLaunchKernelGGL(attn_bwd_bf16, dim3((N * H + 3) / 4), dim3(256), lds, st, (const bf16_t*)qkv, key_mask, (const bf16_t*)dctx, (bf16_t*)dqkv, N, Tk, H, scale, p_drop, make_seed(seed, DIC_STRIDE_DROP), tile_bytes);
    } else {
        size_t lds = (size_t)(4 * Tk * PADW + 2 * Tk * (Tk + 1)) * sizeof(float);
        static bool attr3 = false;
        if (!attr3) { (void)hipFuncSetAttribute((const void*)attn_bwd_f32<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (4 * TMAX * PADW + 2 * TMAX * (TMAX + 1))); attr3 = true; }
        hipLaunchKernelGGL(attn_bwd_f32<float>, dim3(N * H), dim3(64), lds, st, (const float*)qkv, key_mask, (const float*)dctx, (float*)dqkv, N, Tk, H, scale, p_drop, make_seed(seed, DIC_STRIDE_DROP));
    }
    DIC_CHECK_LAUNCH();
    return 0;
}
