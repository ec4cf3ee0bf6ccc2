/* CPU oracle for the rounding head (TEST INFRASTRUCTURE -- see oracle/ref_model.py header).
 *
 * Restates ref CLIP-DDPM.py:323 (logits = x_out[:, :L] @ W^T, bias 0), :620 (softmax(...).argmax(-1)) and the
 * log-softmax gather of :436-437 for ONE fixed summation order: a k-ascending fp32 fmaf chain starting from 0.
 * The gfx950 kernel (csrc/gemm.hip, fp32 path: v_mfma_f32_16x16x4_f32 issued with ascending k) produces bit-identical
 * logits, so token ids can be compared bit-for-bit, including ties (first index wins, as torch.argmax).
 * softmax is monotone, so argmax(softmax(l)) == argmax(l).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -o oracle/librounding.so oracle/rounding.c -lm
 */
#include <math.h>
#include <stdint.h>

void rounding_ref(const float* x, const float* W, int M, int V, int K, const int64_t* tgt,
                  int64_t* argmax, float* maxlogit, double* lse, float* tgt_logit, float* logits_out) {
    for (int m = 0; m < M; ++m) {
        const float* xr = x + (long)m * K;
        float best = -INFINITY;
        int64_t bi = 0;
        double sum = 0.0;
        /* pass 1: logits, max/argmax */
        for (int v = 0; v < V; ++v) {
            const float* w = W + (long)v * K;
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc = fmaf(xr[k], w[k], acc);
            if (logits_out) logits_out[(long)m * V + v] = acc;
            if (acc > best) { best = acc; bi = v; }
            if (tgt && tgt[m] == v) tgt_logit[m] = acc;
        }
        /* pass 2: logsumexp in double around the max */
        for (int v = 0; v < V; ++v) {
            const float* w = W + (long)v * K;
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc = fmaf(xr[k], w[k], acc);
            sum += exp((double)acc - (double)best);
        }
        argmax[m] = bi;
        maxlogit[m] = best;
        lse[m] = (double)best + log(sum);
    }
}
