// losses.hip -- SURVEY.md 8f row 2: the sampled-gather + loss tails of the two training branches,
// each as ONE kernel that also leaves the gradient with respect to the dense head outputs.
//   RPN   RPN/rpn.py:122-137      gather 4 tensors by random_index, BCE-with-logits (mean),
//                                 utils.smooth_l1_loss (utils.py:315-336, beta 1/9, mean over ALL
//                                 gathered elements, negatives included: quirk Q7)
//   RoI   ROIHead/roi_head.py:141-152   cross-entropy (mean), per-class regression rows of the
//                                 positives, smooth-L1 (mean; NaN when there is no positive, as torch)
// The reference runs ~10 tiny torch ops forward and as many backward per branch; the data is a few
// thousand rows, so one workgroup does the whole reduction in a fixed order (deterministic).
#include "common.h"

#define LOSS_THREADS 1024

// Sum over the block of up to three floats; result valid in every thread.  `red` is 3*16 floats.
__device__ __forceinline__ void block_sum3(float &a, float &b, float &c, float *red)
{
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o);
        b += __shfl_down(b, o);
        c += __shfl_down(c, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
        red[wave] = a;
        red[16 + wave] = b;
        red[32 + wave] = c;
    }
    __syncthreads();
    float sa = 0.0f, sb = 0.0f, sc = 0.0f;
    for (int w = 0; w < LOSS_THREADS / 64; ++w) {
        sa += red[w];
        sb += red[16 + w];
        sc += red[32 + w];
    }
    a = sa;
    b = sb;
    c = sc;
}

// utils.py:331-333: x = |d|; x < beta ? 0.5 x^2 / beta : x - 0.5 beta.  *g = d loss / d y_hat.
__device__ __forceinline__ float smooth_l1(float y_hat, float y, float beta, float *g)
{
    const float d = y_hat - y;
    const float x = fabsf(d);
    if (x < beta) {
        *g = d / beta;
        return 0.5f * x * x / beta;
    }
    *g = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    return x - 0.5f * beta;
}

__global__ void __launch_bounds__(LOSS_THREADS)
rpn_loss_kernel(const float *__restrict__ scores, const float4 *__restrict__ regs,
                const float *__restrict__ labels, const float4 *__restrict__ reg_labels,
                const int64_t *__restrict__ index, int S, long long rows, float beta,
                float *__restrict__ loss_out, float *__restrict__ grad_scores,
                float4 *__restrict__ grad_regs)
{
    __shared__ float red[48];
    const float inv_s = 1.0f / (float)S, inv_4s = 1.0f / (4.0f * (float)S);
    float cls = 0.0f, reg = 0.0f, bad = 0.0f;
    for (int s = threadIdx.x; s < S; s += LOSS_THREADS) {
        const long long i = index[s];
        if (i < 0 || i >= rows) { // torch would raise an index error: poison the result instead
            bad += 1.0f;
            continue;
        }
        const float x = scores[i], y = labels[i];
        // F.binary_cross_entropy_with_logits: max(x,0) - x*y + log(1 + exp(-|x|))
        cls += (x > 0.0f ? x : 0.0f) - x * y + log1pf(expf(-fabsf(x)));
        const float4 p = regs[i], t = reg_labels[i];
        float4 g;
        reg += smooth_l1(p.x, t.x, beta, &g.x) + smooth_l1(p.y, t.y, beta, &g.y) +
               smooth_l1(p.z, t.z, beta, &g.z) + smooth_l1(p.w, t.w, beta, &g.w);
        if (grad_scores) atomicAdd(&grad_scores[i], (1.0f / (1.0f + expf(-x)) - y) * inv_s);
        if (grad_regs) {
            float *gr = (float *)&grad_regs[i];
            atomicAdd(gr + 0, g.x * inv_4s);
            atomicAdd(gr + 1, g.y * inv_4s);
            atomicAdd(gr + 2, g.z * inv_4s);
            atomicAdd(gr + 3, g.w * inv_4s);
        }
    }
    block_sum3(cls, reg, bad, red);
    if (threadIdx.x == 0) {
        const float poison = bad > 0.0f ? __int_as_float(0x7fc00000) : 0.0f;
        loss_out[0] = cls * inv_s + poison;
        loss_out[1] = reg * inv_4s + poison;
    }
}

extern "C" int frcnn_rpn_sampled_loss_f32(const float *scores, const float *regs, const float *labels,
                                          const float *reg_labels, const int64_t *index, int S, int64_t rows,
                                          float beta, float *loss_out, float *grad_scores, float *grad_regs,
                                          void *stream)
{
    FRCNN_REQUIRE(S >= 0 && rows >= 0 && beta > 0.0f, "rpn_sampled_loss: bad argument");
    FRCNN_REQUIRE(loss_out, "rpn_sampled_loss: null loss_out");
    FRCNN_REQUIRE(S == 0 || (scores && regs && labels && reg_labels && index), "rpn_sampled_loss: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (grad_scores) {
        hipError_t e = hipMemsetAsync(grad_scores, 0, sizeof(float) * (size_t)rows, st);
        FRCNN_REQUIRE(e == hipSuccess, "rpn_sampled_loss: memset failed: %s", hipGetErrorString(e));
    }
    if (grad_regs) {
        hipError_t e = hipMemsetAsync(grad_regs, 0, sizeof(float) * 4 * (size_t)rows, st);
        FRCNN_REQUIRE(e == hipSuccess, "rpn_sampled_loss: memset failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(rpn_loss_kernel, dim3(1), dim3(LOSS_THREADS), 0, st, scores, (const float4 *)regs, labels,
                       (const float4 *)reg_labels, index, S, (long long)rows, beta, loss_out, grad_scores,
                       (float4 *)grad_regs);
    FRCNN_CHECK_LAUNCH("rpn_sampled_loss");
    return FRCNN_OK;
}

// One thread per proposal row (strided): cross-entropy over Kc logits and, for label > 0, smooth-L1 of
// pred[n, label, :] against reg_label[n, :].  Pass 1 counts the positives (the regression mean's
// denominator), pass 2 accumulates the losses and writes every gradient element exactly once.
__global__ void __launch_bounds__(LOSS_THREADS)
roi_loss_kernel(const float *__restrict__ logits, const float4 *__restrict__ pred,
                const float *__restrict__ cls_label, const float4 *__restrict__ reg_label, int N, int Kc,
                float beta, float *__restrict__ loss_out, float *__restrict__ grad_logits,
                float4 *__restrict__ grad_pred)
{
    __shared__ float red[48];
    float npos = 0.0f, bad = 0.0f, zero = 0.0f;
    for (int n = threadIdx.x; n < N; n += LOSS_THREADS) {
        const float l = cls_label[n];
        const int li = (int)l; // .long(): truncation
        if (!(li >= 0 && li < Kc)) bad += 1.0f;
        else if (l > 0.0f) npos += 1.0f;
    }
    block_sum3(npos, bad, zero, red);
    const float inv_n = 1.0f / (float)N;
    const float inv_4p = 1.0f / (4.0f * npos); // inf when there is no positive: 0 * inf = NaN below, as torch
    float ce = 0.0f, reg = 0.0f;
    zero = 0.0f;
    for (int n = threadIdx.x; n < N; n += LOSS_THREADS) {
        const float l = cls_label[n];
        int li = (int)l;
        li = (li >= 0 && li < Kc) ? li : 0;
        const float *x = logits + (long long)n * Kc;
        float m = x[0];
        for (int k = 1; k < Kc; ++k) m = x[k] > m ? x[k] : m;
        float sum = 0.0f;
        for (int k = 0; k < Kc; ++k) sum += expf(x[k] - m);
        const float lse = m + logf(sum);
        ce += lse - x[li];
        if (grad_logits) {
            float *g = grad_logits + (long long)n * Kc;
            for (int k = 0; k < Kc; ++k) g[k] = (expf(x[k] - lse) - (k == li ? 1.0f : 0.0f)) * inv_n;
        }
        const bool posr = l > 0.0f;
        if (posr) {
            const float4 p = pred[(long long)n * Kc + li], t = reg_label[n];
            float4 g;
            reg += smooth_l1(p.x, t.x, beta, &g.x) + smooth_l1(p.y, t.y, beta, &g.y) +
                   smooth_l1(p.z, t.z, beta, &g.z) + smooth_l1(p.w, t.w, beta, &g.w);
            if (grad_pred) {
                for (int k = 0; k < Kc; ++k)
                    grad_pred[(long long)n * Kc + k] =
                        k == li ? make_float4(g.x * inv_4p, g.y * inv_4p, g.z * inv_4p, g.w * inv_4p)
                                : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        } else if (grad_pred) {
            for (int k = 0; k < Kc; ++k) grad_pred[(long long)n * Kc + k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    block_sum3(ce, reg, zero, red);
    if (threadIdx.x == 0) {
        const float poison = bad > 0.0f ? __int_as_float(0x7fc00000) : 0.0f;
        loss_out[0] = ce * inv_n + poison;
        loss_out[1] = (npos > 0.0f ? reg * inv_4p : __int_as_float(0x7fc00000)) + poison;
    }
}

extern "C" int frcnn_roi_loss_f32(const float *logits, const float *pred, const float *cls_label,
                                  const float *reg_label, int N, int Kc, float beta, float *loss_out,
                                  float *grad_logits, float *grad_pred, void *stream)
{
    FRCNN_REQUIRE(N >= 0 && Kc > 0 && beta > 0.0f, "roi_loss: bad argument");
    FRCNN_REQUIRE(loss_out, "roi_loss: null loss_out");
    FRCNN_REQUIRE(N == 0 || (logits && pred && cls_label && reg_label), "roi_loss: null pointer");
    hipLaunchKernelGGL(roi_loss_kernel, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, logits,
                       (const float4 *)pred, cls_label, (const float4 *)reg_label, N, Kc, beta, loss_out,
                       grad_logits, (float4 *)grad_pred);
    FRCNN_CHECK_LAUNCH("roi_loss");
    return FRCNN_OK;
}
