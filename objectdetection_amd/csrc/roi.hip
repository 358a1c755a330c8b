// roi.hip -- K6 / K7: RoI list assembly, FPN level mapping, RoIAlign (bilinear average) and the
// max RoIPool variant.  Replaces torchvision.ops.MultiScaleRoIAlign as used by
// ROIHead/roi_pooling.py:44-53 (call site ROIHead/roi_head.py:127-130).
//
// The output (K, C, 7, 7) fp32 is 97 % of the hot path's HBM bytes: the kernels are
// organised around writing it in long contiguous runs.
#include <math.h>

#include "common.h"

// ---- roi list ---------------------------------------------------------------------------------
// rois[off[b] + j] = [b, props[b, j, :]] for j < count[b]; one 1024-thread block, B <= 4096.
__global__ void __launch_bounds__(1024)
build_rois_kernel(const float4 *__restrict__ props, const int32_t *__restrict__ count, int B, int P,
                  float *__restrict__ rois, int32_t *__restrict__ total_out)
{
    extern __shared__ int s_off[]; // B + 1
    if (threadIdx.x == 0) {
        int run = 0;
        for (int b = 0; b < B; ++b) {
            s_off[b] = run;
            int c = count ? count[b] : P;
            run += c < 0 ? 0 : (c > P ? P : c);
        }
        s_off[B] = run;
        *total_out = run;
    }
    __syncthreads();
    for (long long t = threadIdx.x; t < (long long)B * P; t += blockDim.x) {
        int b = (int)(t / P);
        int j = (int)(t - (long long)b * P);
        int c = s_off[b + 1] - s_off[b];
        if (j < c) {
            float4 p = props[t];
            float *o = rois + (long long)(s_off[b] + j) * 5;
            o[0] = (float)b;
            o[1] = p.x;
            o[2] = p.y;
            o[3] = p.z;
            o[4] = p.w;
        }
    }
}

extern "C" int frcnn_build_rois_f32(const float *props, const int32_t *count, int B, int P,
                                    float *rois_out, int32_t *total_out, void *stream)
{
    FRCNN_REQUIRE(B > 0 && B <= 4096 && P >= 0, "build_rois: bad shape");
    FRCNN_REQUIRE(total_out && (P == 0 || (props && rois_out)), "build_rois: null pointer");
    hipLaunchKernelGGL(build_rois_kernel, dim3(1), dim3(1024), (size_t)(B + 1) * sizeof(int),
                       (hipStream_t)stream, (const float4 *)props, count, B, P, rois_out, total_out);
    FRCNN_CHECK_LAUNCH("build_rois");
    return FRCNN_OK;
}

// ---- level mapper -----------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
level_map_kernel(const float *__restrict__ rois, const int32_t *__restrict__ n_rois, int K,
                 int k_min, int k_max, int32_t *__restrict__ lvl)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n = n_rois ? min(*n_rois, K) : K;
    if (i >= n) return;
    const float *r = rois + (long long)i * 5;
    float area = (r[3] - r[1]) * (r[4] - r[2]);
    float s = sqrtf(area);
    float t = floorf((4.0f + log2f(s / 224.0f)) + 1e-6f);
    int out = -1;
    if (t == t) {
        t = t < (float)k_min ? (float)k_min : t; // torch.clamp: min first, then max
        t = t > (float)k_max ? (float)k_max : t;
        out = (int)t - k_min;
    }
    lvl[i] = out;
}

extern "C" int frcnn_level_map_f32(const float *rois, const int32_t *n_rois, int K, int k_min,
                                   int k_max, int32_t *lvl_out, void *stream)
{
    FRCNN_REQUIRE(K >= 0, "level_map: bad shape");
    if (K == 0) return FRCNN_OK;
    FRCNN_REQUIRE(rois && lvl_out, "level_map: null pointer");
    hipLaunchKernelGGL(level_map_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       rois, n_rois, K, k_min, k_max, lvl_out);
    FRCNN_CHECK_LAUNCH("level_map");
    return FRCNN_OK;
}

// ---- RoIAlign ---------------------------------------------------------------------------------
// One bilinear sample: 4 tap offsets inside the channel plane + 4 weights (zero when the sample
// falls outside [-1, H] x [-1, W]).  Mirrors torchvision's pre_calc_for_bilinear_interpolate.
struct __align__(16) Sample {
    int p1, p2, p3, p4;
    float w1, w2, w3, w4;
};

__device__ __forceinline__ Sample make_sample(float y, float x, int H, int W)
{
    Sample s;
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
        s.p1 = s.p2 = s.p3 = s.p4 = 0;
        s.w1 = s.w2 = s.w3 = s.w4 = 0.0f;
        return s;
    }
    if (y <= 0.0f) y = 0.0f;
    if (x <= 0.0f) x = 0.0f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) {
        y_high = y_low = H - 1;
        y = (float)y_low;
    } else {
        y_high = y_low + 1;
    }
    if (x_low >= W - 1) {
        x_high = x_low = W - 1;
        x = (float)x_low;
    } else {
        x_high = x_low + 1;
    }
    float ly = y - (float)y_low, lx = x - (float)x_low;
    float hy = 1.0f - ly, hx = 1.0f - lx;
    s.p1 = y_low * W + x_low;
    s.p2 = y_low * W + x_high;
    s.p3 = y_high * W + x_low;
    s.p4 = y_high * W + x_high;
    s.w1 = hy * hx;
    s.w2 = hy * lx;
    s.w3 = ly * hx;
    s.w4 = ly * lx;
    return s;
}

#define RA_MAX_SAMPLES 1024 // pooled_h*pooled_w*grid_h*grid_w staged per RoI in LDS (32 KB)
#define RA_CHUNK 32         // channels per block

// Generic direct-gather RoIAlign: block = (roi, 32-channel chunk).  The RoI's sample table is
// built once in LDS; threads then sweep the chunk's (channel, bin) outputs in memory order so the
// block writes one contiguous 32*49*4-byte run.
__global__ void __launch_bounds__(256)
roi_align_direct_kernel(const float *__restrict__ feat, int B, int C, int H, int W,
                        const float *__restrict__ rois, const int32_t *__restrict__ n_rois,
                        int pooled_h, int pooled_w, float scale, int sampling_ratio,
                        const int32_t *__restrict__ lvl, int lvl_value, float *__restrict__ out)
{
    __shared__ Sample s_samp[RA_MAX_SAMPLES];
    const int r = blockIdx.x;
    if (n_rois && r >= *n_rois) return;
    if (lvl && lvl[r] != lvl_value) return;
    const float *roi = rois + (long long)r * 5;
    const int b = (int)roi[0];
    if (b < 0 || b >= B) return; // a bad batch index must not become an out-of-bounds read
    const float start_w = roi[1] * scale;
    const float start_h = roi[2] * scale;
    const float end_w = roi[3] * scale;
    const float end_h = roi[4] * scale;
    float roi_w = end_w - start_w;
    float roi_h = end_h - start_h;
    roi_w = roi_w > 1.0f ? roi_w : 1.0f;
    roi_h = roi_h > 1.0f ? roi_h : 1.0f;
    const float bin_h = roi_h / (float)pooled_h;
    const float bin_w = roi_w / (float)pooled_w;
    const int grid_h = sampling_ratio, grid_w = sampling_ratio; // host guarantees >= 1
    const int per_bin = grid_h * grid_w;
    const float count = (float)per_bin;
    const int bins = pooled_h * pooled_w;
    const int nsamp = bins * per_bin; // host guarantees <= RA_MAX_SAMPLES
    for (int t = threadIdx.x; t < nsamp; t += blockDim.x) {
        int bin = t / per_bin;
        int si = t - bin * per_bin;
        int iy = si / grid_w, ix = si - iy * grid_w;
        int ph = bin / pooled_w, pw = bin - ph * pooled_w;
        float yy = start_h + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)grid_h;
        float xx = start_w + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)grid_w;
        s_samp[t] = make_sample(yy, xx, H, W);
    }
    __syncthreads();
    const int c0 = blockIdx.y * RA_CHUNK;
    const int nc = min(RA_CHUNK, C - c0);
    const float *fb = feat + ((long long)b * C + c0) * H * W;
    float *ob = out + ((long long)r * C + c0) * bins;
    for (int t = threadIdx.x; t < nc * bins; t += blockDim.x) {
        int c = t / bins;
        int bin = t - c * bins;
        const float *f = fb + (long long)c * H * W;
        float acc = 0.0f;
        const Sample *sp = s_samp + bin * per_bin;
        for (int si = 0; si < per_bin; ++si) {
            Sample s = sp[si];
            acc += s.w1 * f[s.p1] + s.w2 * f[s.p2] + s.w3 * f[s.p3] + s.w4 * f[s.p4];
        }
        ob[t] = acc / count;
    }
}

extern "C" int frcnn_roi_align_f32(const float *feat, int B, int C, int H, int W, const float *rois,
                                   const int32_t *n_rois, int K, int pooled_h, int pooled_w,
                                   float spatial_scale, int sampling_ratio, const int32_t *lvl,
                                   int lvl_value, float *out, void *stream)
{
    FRCNN_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && K >= 0 && pooled_h > 0 && pooled_w > 0,
                  "roi_align: bad shape");
    if (K == 0) return FRCNN_OK;
    FRCNN_REQUIRE(feat && rois && out, "roi_align: null pointer");
    FRCNN_REQUIRE((long long)H * W < (1ll << 30), "roi_align: feature plane too large");
    // torchvision's adaptive grid (sampling_ratio <= 0) is not offered: the reference always
    // passes sampling_ratio = 2 (ROIHead/roi_head.py:259) and an unbounded per-RoI grid has no
    // place in a batched kernel.
    FRCNN_REQUIRE(sampling_ratio >= 1, "roi_align: sampling_ratio must be >= 1 (got %d)", sampling_ratio);
    FRCNN_REQUIRE((long long)pooled_h * pooled_w * sampling_ratio * sampling_ratio <= RA_MAX_SAMPLES,
                  "roi_align: pooled_h*pooled_w*sampling_ratio^2 exceeds %d", RA_MAX_SAMPLES);
    int chunks = (C + RA_CHUNK - 1) / RA_CHUNK;
    FRCNN_REQUIRE(chunks <= 65535, "roi_align: too many channels");
    hipLaunchKernelGGL(roi_align_direct_kernel, dim3(K, chunks), dim3(256), 0, (hipStream_t)stream,
                       feat, B, C, H, W, rois, n_rois, pooled_h, pooled_w, spatial_scale,
                       sampling_ratio, lvl, lvl_value, out);
    FRCNN_CHECK_LAUNCH("roi_align");
    return FRCNN_OK;
}

// ---- max RoIPool (torchvision roi_pool semantics; not called by the reference) -----------------
__global__ void __launch_bounds__(256)
roi_pool_kernel(const float *__restrict__ feat, int B, int C, int H, int W,
                const float *__restrict__ rois,
                int pooled_h, int pooled_w, float scale, float *__restrict__ out,
                int32_t *__restrict__ argmax)
{
    const int r = blockIdx.x;
    const float *roi = rois + (long long)r * 5;
    const int b = (int)roi[0];
    if (b < 0 || b >= B) return;
    const int rsw = (int)roundf(roi[1] * scale);
    const int rsh = (int)roundf(roi[2] * scale);
    const int rew = (int)roundf(roi[3] * scale);
    const int reh = (int)roundf(roi[4] * scale);
    const int rw = max(rew - rsw + 1, 1);
    const int rh = max(reh - rsh + 1, 1);
    const float bin_h = (float)rh / (float)pooled_h;
    const float bin_w = (float)rw / (float)pooled_w;
    const int bins = pooled_h * pooled_w;
    const int c0 = blockIdx.y * RA_CHUNK;
    const int nc = min(RA_CHUNK, C - c0);
    const float *fb = feat + ((long long)b * C + c0) * H * W;
    const long long o0 = ((long long)r * C + c0) * bins;
    for (int t = threadIdx.x; t < nc * bins; t += blockDim.x) {
        int c = t / bins;
        int bin = t - c * bins;
        int ph = bin / pooled_w, pw = bin - ph * pooled_w;
        int hs = (int)floorf((float)ph * bin_h);
        int ws = (int)floorf((float)pw * bin_w);
        int he = (int)ceilf((float)(ph + 1) * bin_h);
        int we = (int)ceilf((float)(pw + 1) * bin_w);
        hs = min(max(hs + rsh, 0), H);
        he = min(max(he + rsh, 0), H);
        ws = min(max(ws + rsw, 0), W);
        we = min(max(we + rsw, 0), W);
        bool empty = (he <= hs) || (we <= ws);
        const float *f = fb + (long long)c * H * W;
        float mx = empty ? 0.0f : -3.402823466e+38f;
        int mi = -1;
        for (int h = hs; h < he; ++h)
            for (int w = ws; w < we; ++w) {
                float v = f[h * W + w];
                if (v > mx) {
                    mx = v;
                    mi = h * W + w;
                }
            }
        out[o0 + t] = mx;
        argmax[o0 + t] = mi;
    }
}

extern "C" int frcnn_roi_pool_f32(const float *feat, int B, int C, int H, int W, const float *rois,
                                  int K, int pooled_h, int pooled_w, float spatial_scale, float *out,
                                  int32_t *argmax, void *stream)
{
    FRCNN_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && K >= 0 && pooled_h > 0 && pooled_w > 0,
                  "roi_pool: bad shape");
    if (K == 0) return FRCNN_OK;
    FRCNN_REQUIRE(feat && rois && out && argmax, "roi_pool: null pointer");
    int chunks = (C + RA_CHUNK - 1) / RA_CHUNK;
    hipLaunchKernelGGL(roi_pool_kernel, dim3(K, chunks), dim3(256), 0, (hipStream_t)stream, feat, B,
                       C, H, W, rois, pooled_h, pooled_w, spatial_scale, out, argmax);
    FRCNN_CHECK_LAUNCH("roi_pool");
    return FRCNN_OK;
}
