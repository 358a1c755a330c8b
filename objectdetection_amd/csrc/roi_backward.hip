// roi_backward.hip -- backward of the batched MultiScaleRoIAlign level pass with respect to the feature map.
// The reference pools with torchvision's differentiable MultiScaleRoIAlign (ROIHead/roi_pooling.py:51-53, call
// site ROIHead/roi_head.py:127-130), so its RoI losses train the FPN / backbone through it; this is that
// gradient: every sample of a bin sends grad * w_k / count to its four taps (torchvision's
// roi_align_backward), samples outside [-1, H] x [-1, W] send nothing, boxes get no gradient.
//
// torchvision scatters with one global float atomic per tap (16 per output element).  Here a workgroup owns
// (image, channel chunk, RoI split) and keeps the chunk's GRADIENT PLANES in LDS: the scatter is LDS float
// atomics (ds_add_f32, no return), and HBM sees grad_out once plus one write of the planes.  Planes that do
// not fit LDS (FPN P2 at 800 x 1344) fall back to global atomics.  The sum order over RoIs is not fixed
// (atomics), as in torchvision: results agree to rounding, not bit for bit, between runs.
#include <math.h>

#include "common.h"
#include "roi_common.h"

#define RAB_THREADS 1024
#define RAB_LDS_BUDGET (150 * 1024)

#ifndef RAB_EXP
#define RAB_EXP 0
#endif
__device__ __forceinline__ void lds_add(float *p, float v)
{
#if RAB_EXP == 1 // A/B: plain read-modify-write (wrong sums, speed probe only)
    *p += v;
#elif RAB_EXP == 2 // A/B: atomics, but every lane on its own address
    __hip_atomic_fetch_add(p - (p - (float *)0) % 64 + (threadIdx.x & 63), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // ds_add_f32, result unused
#endif
}

// The four samples of bin (ph, pw) of a RoI record (start_w, start_h, bin_w, bin_h in feature pixels): the
// expressions of roi_align's forward, so that forward and backward place every sample in the same cell.
__device__ __forceinline__ void bin_samples(const float4 rec, int ph, int pw, int H, int W, Sample (&s)[4])
{
#pragma unroll
    for (int iy = 0; iy < 2; ++iy)
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
            const float y = rec.y + (float)ph * rec.w + ((float)iy + 0.5f) * rec.w / 2.0f;
            const float x = rec.x + (float)pw * rec.z + ((float)ix + 0.5f) * rec.z / 2.0f;
            s[iy * 2 + ix] = make_sample(y, x, H, W);
        }
}

__device__ __forceinline__ float4 rab_record(float4 box, float scale)
{
    float start_w = box.x * scale, start_h = box.y * scale;
    float roi_w = box.z * scale - start_w, roi_h = box.w * scale - start_h;
    roi_w = roi_w > 1.0f ? roi_w : 1.0f;
    roi_h = roi_h > 1.0f ? roi_h : 1.0f;
    return make_float4(start_w, start_h, roi_w / 7.0f, roi_h / 7.0f);
}

// Workgroup = (image, CC-channel chunk, RoI split).  A wave takes one RoI per iteration, lane = bin (as the
// forward plane kernel): the taps and weights of the bin's four samples are computed once and used for every
// channel of the chunk; grad_out is read in 196-byte runs.  splits == 1: the workgroup owns its planes and
// writes them with plain stores; splits > 1 (small batches): grad_feat was zeroed by the host entry and the
// planes are added to it with global float atomics.
__global__ void __launch_bounds__(RAB_THREADS)
roi_align_plane_backward_kernel(const float *__restrict__ grad_out, int C, int H, int W,
                                const float4 *__restrict__ props, const int32_t *__restrict__ count, int P,
                                int CC, float scale, int level, int n_levels, int k_min, int k_max, int chunks,
                                int splits, int B, float *__restrict__ grad_feat)
{
    extern __shared__ float s_g[]; // [CC][H * W]
    __shared__ int s_off, s_cnt;
    const int chunk = blockIdx.x % chunks;
    const int grp = blockIdx.x / chunks;
    const int split = grp % splits;
    const int b = grp / splits;
    const int c0 = chunk * CC;
    const int nc = min(CC, C - c0);
    const int HW = H * W;
    if (threadIdx.x == 0) {
        int off = 0;
        for (int i = 0; i < B; ++i) {
            int c = count ? count[i] : P;
            c = c < 0 ? 0 : (c > P ? P : c);
            if (i < b) off += c;
            if (i == b) s_cnt = c;
        }
        s_off = off;
    }
    for (int t = threadIdx.x; t < nc * HW; t += RAB_THREADS) s_g[t] = 0.0f;
    __syncthreads();
    const int cnt = s_cnt;
    const int per = (cnt + splits - 1) / splits;
    const int j1 = min(cnt, (split + 1) * per);
    const float4 *pb = props + (long long)b * P;
    const int bin = threadIdx.x & 63;
    const int ph = bin / 7, pw = bin - ph * 7;
    const float *gb = grad_out + ((long long)s_off * C + c0) * 49;
    for (int j = split * per + (int)(threadIdx.x >> 6); j < j1; j += RAB_THREADS / 64) {
        const float4 box = pb[j];
        if (n_levels > 1 && map_level(box, k_min, k_max) != level) continue; // wave-uniform
        if (bin >= 49) continue;
        Sample s[4];
        bin_samples(rab_record(box, scale), ph, pw, H, W, s);
        const float *g = gb + (long long)j * C * 49 + bin;
        for (int c = 0; c < nc; ++c) {
            const float gv = g[c * 49] * 0.25f; // grad / count, count = 2 x 2 (exact)
            float *pl = s_g + c * HW;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // a sample outside the map has all four weights zero and sends nothing (torchvision skips it)
                if (s[q].w1 != 0.0f || s[q].w2 != 0.0f || s[q].w3 != 0.0f || s[q].w4 != 0.0f) {
                    lds_add(pl + s[q].p1, gv * s[q].w1);
                    lds_add(pl + s[q].p2, gv * s[q].w2);
                    lds_add(pl + s[q].p3, gv * s[q].w3);
                    lds_add(pl + s[q].p4, gv * s[q].w4);
                }
            }
        }
    }
    __syncthreads();
    float *out = grad_feat + ((long long)b * C + c0) * HW;
    if (splits == 1) {
        for (int t = threadIdx.x; t < nc * HW; t += RAB_THREADS) out[t] = s_g[t];
    } else {
        for (int t = threadIdx.x; t < nc * HW; t += RAB_THREADS) {
            const float v = s_g[t];
            if (v != 0.0f) unsafeAtomicAdd(out + t, v);
        }
    }
}

// Any plane size / pooled size / sampling ratio: block = (RoI split, 32-channel chunk, image), global float
// atomics per tap (grad_feat zeroed by the host entry).
__global__ void __launch_bounds__(256)
roi_align_backward_atomic_kernel(const float *__restrict__ grad_out, int C, int H, int W,
                                 const float4 *__restrict__ props, const int32_t *__restrict__ count, int P,
                                 int pooled_h, int pooled_w, float scale, int sampling_ratio, int level,
                                 int n_levels, int k_min, int k_max, float *__restrict__ grad_feat)
{
    __shared__ int s_off, s_cnt;
    const int b = blockIdx.z;
    const int B = gridDim.z;
    if (threadIdx.x == 0) {
        int off = 0;
        for (int i = 0; i < B; ++i) {
            int c = count ? count[i] : P;
            c = c < 0 ? 0 : (c > P ? P : c);
            if (i < b) off += c;
            if (i == b) s_cnt = c;
        }
        s_off = off;
    }
    __syncthreads();
    const int cnt = s_cnt;
    const int bins = pooled_h * pooled_w;
    const int c0 = blockIdx.y * 32;
    const int nc = min(32, C - c0);
    const float inv = 1.0f / (float)(sampling_ratio * sampling_ratio);
    for (int j = blockIdx.x; j < cnt; j += gridDim.x) {
        const float4 box = props[(long long)b * P + j];
        if (n_levels > 1 && map_level(box, k_min, k_max) != level) continue;
        float start_w = box.x * scale, start_h = box.y * scale;
        float roi_w = box.z * scale - start_w, roi_h = box.w * scale - start_h;
        roi_w = roi_w > 1.0f ? roi_w : 1.0f;
        roi_h = roi_h > 1.0f ? roi_h : 1.0f;
        const float bin_h = roi_h / (float)pooled_h, bin_w = roi_w / (float)pooled_w;
        const float *g = grad_out + ((long long)(s_off + j) * C + c0) * bins;
        for (int t = threadIdx.x; t < nc * bins; t += blockDim.x) {
            const int c = t / bins;
            const int bin = t - c * bins;
            const int ph = bin / pooled_w, pw = bin - ph * pooled_w;
            const float gv = g[t] * inv;
            float *pl = grad_feat + ((long long)b * C + c0 + c) * H * W;
            for (int iy = 0; iy < sampling_ratio; ++iy)
                for (int ix = 0; ix < sampling_ratio; ++ix) {
                    const float y = start_h + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)sampling_ratio;
                    const float x = start_w + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)sampling_ratio;
                    const Sample s = make_sample(y, x, H, W);
                    if (s.w1 != 0.0f || s.w2 != 0.0f || s.w3 != 0.0f || s.w4 != 0.0f) {
                        unsafeAtomicAdd(pl + s.p1, gv * s.w1);
                        unsafeAtomicAdd(pl + s.p2, gv * s.w2);
                        unsafeAtomicAdd(pl + s.p3, gv * s.w3);
                        unsafeAtomicAdd(pl + s.p4, gv * s.w4);
                    }
                }
        }
    }
}

extern "C" int frcnn_roi_align_batched_backward_f32(const float *grad_out, int B, int C, int H, int W,
                                                    const float *props, const int32_t *count, int P,
                                                    int pooled_h, int pooled_w, float spatial_scale,
                                                    int sampling_ratio, int level, int n_levels, int k_min,
                                                    int k_max, float *grad_feat, void *stream)
{
    FRCNN_REQUIRE(B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && P >= 0 && pooled_h > 0 && pooled_w > 0,
                  "roi_align_backward: bad shape");
    FRCNN_REQUIRE(n_levels >= 1 && level >= 0 && level < n_levels, "roi_align_backward: bad level");
    FRCNN_REQUIRE(grad_feat, "roi_align_backward: null grad_feat");
    FRCNN_REQUIRE(sampling_ratio >= 1, "roi_align_backward: sampling_ratio must be >= 1 (got %d)", sampling_ratio);
    FRCNN_REQUIRE((long long)H * W < (1ll << 30), "roi_align_backward: feature plane too large");
    hipStream_t st = (hipStream_t)stream;
    const size_t plane_elems = (size_t)B * C * H * W;
    if (P == 0) {
        hipError_t e = hipMemsetAsync(grad_feat, 0, plane_elems * sizeof(float), st);
        FRCNN_REQUIRE(e == hipSuccess, "roi_align_backward: memset failed: %s", hipGetErrorString(e));
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(grad_out && props, "roi_align_backward: null pointer");
    const long long plane_bytes = (long long)H * W * 4;
    if (pooled_h == 7 && pooled_w == 7 && sampling_ratio == 2 && plane_bytes <= RAB_LDS_BUDGET &&
        (long long)P * 49 < (1ll << 30)) {
        int CC = (int)(RAB_LDS_BUDGET / plane_bytes);
        if (CC > 16) CC = 16;
        if (CC > C) CC = C;
        const int chunks = (C + CC - 1) / CC;
        // one workgroup per (image, chunk) when that fills the chip; else split the RoIs and add the planes up
        int splits = 1;
        if ((long long)B * chunks < 256) {
            splits = (int)((512 + (long long)B * chunks - 1) / ((long long)B * chunks));
            if (splits > 16) splits = 16;
        }
        if (splits > 1) {
            hipError_t e = hipMemsetAsync(grad_feat, 0, plane_elems * sizeof(float), st);
            FRCNN_REQUIRE(e == hipSuccess, "roi_align_backward: memset failed: %s", hipGetErrorString(e));
        }
        const size_t lds = (size_t)plane_bytes * (size_t)CC;
        static size_t granted[FRCNN_MAX_DEVICES];
        hipError_t e = frcnn_reserve_dynamic_lds((const void *)roi_align_plane_backward_kernel, lds, granted);
        FRCNN_REQUIRE(e == hipSuccess, "roi_align_backward: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        FRCNN_REQUIRE((long long)B * chunks * splits < (1ll << 31), "roi_align_backward: grid too large");
        hipLaunchKernelGGL(roi_align_plane_backward_kernel, dim3((unsigned)(B * chunks * splits)), dim3(RAB_THREADS),
                           lds, st, grad_out, C, H, W, (const float4 *)props, count, P, CC, spatial_scale, level,
                           n_levels, k_min, k_max, chunks, splits, B, grad_feat);
        FRCNN_CHECK_LAUNCH("roi_align_plane_backward");
        return FRCNN_OK;
    }
    hipError_t e = hipMemsetAsync(grad_feat, 0, plane_elems * sizeof(float), st);
    FRCNN_REQUIRE(e == hipSuccess, "roi_align_backward: memset failed: %s", hipGetErrorString(e));
    const int chunks = (C + 31) / 32;
    FRCNN_REQUIRE(chunks <= 65535, "roi_align_backward: too many channels");
    const int splits = P < 256 ? P : 256;
    hipLaunchKernelGGL(roi_align_backward_atomic_kernel, dim3(splits, chunks, B), dim3(256), 0, st, grad_out, C, H, W,
                       (const float4 *)props, count, P, pooled_h, pooled_w, spatial_scale, sampling_ratio, level,
                       n_levels, k_min, k_max, grad_feat);
    FRCNN_CHECK_LAUNCH("roi_align_backward_atomic");
    return FRCNN_OK;
}
