// roi_backward.hip -- backward of the batched MultiScaleRoIAlign level pass with respect to the feature map.
// The reference pools with torchvision's differentiable MultiScaleRoIAlign (ROIHead/roi_pooling.py:51-53, call
// site ROIHead/roi_head.py:127-130), so its RoI losses train the FPN / backbone through it; this is that
// gradient: every sample of a bin sends grad * w_k / count to its four taps (torchvision's
// roi_align_backward), samples outside [-1, H] x [-1, W] send nothing, boxes get no gradient.
//
// torchvision scatters with one global float atomic per tap (16 per output element).  Here a workgroup owns
// (image, channel chunk, RoI split) and keeps the chunk's GRADIENT PLANES in LDS, one wave per channel, and the
// scatter needs no atomics (see roi_align_plane_backward_kernel); HBM sees grad_out once plus one write of the
// planes, and with one workgroup per (image, chunk) the result is deterministic.  Planes too large for four
// channels of them to fit LDS (FPN P3 / P2 at 800 x 1344) fall back to global atomics, whose sum order is not
// fixed (as in torchvision).
#include <math.h>

#include "common.h"
#include "roi_common.h"

#define RAB_LDS_BUDGET (150 * 1024)

__device__ __forceinline__ float4 rab_record(float4 box, float scale)
{
    float start_w = box.x * scale, start_h = box.y * scale;
    float roi_w = box.z * scale - start_w, roi_h = box.w * scale - start_h;
    roi_w = roi_w > 1.0f ? roi_w : 1.0f;
    roi_h = roi_h > 1.0f ? roi_h : 1.0f;
    return make_float4(start_w, start_h, roi_w / 7.0f, roi_h / 7.0f);
}

__device__ __forceinline__ float lane_get(float v, int src_lane) // src_lane the same for every lane
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// Weight that the two samples of an axis record put on cell `cell` (0 when it is none of their taps).  A sample's
// taps are lo and lo + 1 (a sample clamped to the last cell has all its weight on lo and 0 on "lo + 1", so the
// out-of-range cell never receives anything); idx holds lo of the two samples, 0x7FFF0000-ish sentinels for an
// invalid one.
__device__ __forceinline__ float axis_weight(const int2 lo, const float4 w, int cell)
{
    const int d0 = cell - lo.x, d1 = cell - lo.y;
    return (d0 == 0 ? w.x : (d0 == 1 ? w.y : 0.0f)) + (d1 == 0 ? w.z : (d1 == 1 ? w.w : 0.0f));
}

// Workgroup = (image, chunk of nw channels, RoI split), nw waves: WAVE w OWNS CHANNEL c0 + w and its gradient
// plane in LDS, so no two waves ever add into the same plane, and inside a wave the lanes are the CELLS of a row
// segment -- distinct addresses -- so the scatter needs no atomics at all: ds_read, add, ds_write.  (LDS float
// atomics are what the first version used: ds_add_f32 costs ~130 cycles per wave instruction on gfx950 whether or
// not lanes collide -- 68 ms at the headline shape, 62 ms with every lane on its own address, 4.6 ms with plain
// read-modify-write.)  RoIAlign's weights are separable, grad_feat[y][x] += sum_ph Wy[ph][y] * (sum_pw G[ph][pw] *
// Wx[pw][x]), so a wave first contracts the RoI's 7 x 7 gradients of its channel over the bins' columns
// (T[ph] per lane = column x; G broadcast with v_readlane) and then walks the RoI's rows (Wy of the row broadcast the
// same way; rows no sample touches are skipped).  The axis records of a batch of RoIs (taps and weights of the two
// samples of each of the 7 + 7 bins, the forward's expressions) are built once per workgroup in LDS, one RoI per wave.
__global__ void __launch_bounds__(1024)
roi_align_plane_backward_kernel(const float *__restrict__ grad_out, int C, int H, int W,
                                const float4 *__restrict__ props, const int32_t *__restrict__ count, int P,
                                float scale, int level, int n_levels, int k_min, int k_max, int chunks,
                                int splits, int B, float *__restrict__ grad_feat)
{
    extern __shared__ float s_g[]; // [nw][H * W]
    __shared__ int2 s_ri[16 * 16];    // per RoI of the batch: records 0..6 rows, 8..14 columns: lo of the two samples
    __shared__ float4 s_rw[16 * 16];  // their weights (rows carry the 1/4 of the 2 x 2 average)
    __shared__ int4 s_rng[16];        // ymin, ymax, xmin, xmax of the RoI's taps (ymin > ymax: nothing to do)
    __shared__ int s_off, s_cnt;
    const int nw = blockDim.x >> 6;
    const int chunk = blockIdx.x % chunks;
    const int grp = blockIdx.x / chunks;
    const int split = grp % splits;
    const int b = grp / splits;
    const int c0 = chunk * nw;
    const int nc = min(nw, C - c0);
    const int HW = H * W;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    for (int t = threadIdx.x; t < nw * HW; t += blockDim.x) s_g[t] = 0.0f;
    __syncthreads();
    const int cnt = s_cnt;
    const int per = (cnt + splits - 1) / splits;
    const int j_end = min(cnt, (split + 1) * per);
    const float4 *pb = props + (long long)b * P;
    float *plane = s_g + wave * HW;
    const float *gb = grad_out + ((long long)s_off * C + c0 + wave) * 49;
    for (int j0 = split * per; j0 < j_end; j0 += nw) {
        __syncthreads(); // the previous batch's records have been used
        {
            // ---- records of RoI j0 + wave: lanes 0..6 its bin rows, lanes 8..14 its bin columns
            const int j = j0 + wave;
            bool live = j < j_end;
            float4 rec = make_float4(0.0f, 0.0f, 1.0f, 1.0f);
            if (live) {
                const float4 box = pb[j];
                if (n_levels > 1 && map_level(box, k_min, k_max) != level) live = false;
                rec = rab_record(box, scale);
            }
            if (lane < 16) {
                const int axis = lane >> 3, p = lane & 7;
                const float start = axis ? rec.x : rec.y, bin = axis ? rec.z : rec.w;
                const int n = axis ? W : H;
                const AxisSample a0 = axis_sample(start + (float)p * bin + (0.0f + 0.5f) * bin / 2.0f, n);
                const AxisSample a1 = axis_sample(start + (float)p * bin + (1.0f + 0.5f) * bin / 2.0f, n);
                const float k = axis ? 1.0f : 0.25f;
                const bool rec_ok = live && p < 7;
                // (an invalid sample: a lo that no cell is near)
                s_ri[wave * 16 + lane] = make_int2(a0.ok ? a0.lo : -0x40000000, a1.ok ? a1.lo : -0x40000000);
                s_rw[wave * 16 + lane] = make_float4(a0.wlo * k, a0.whi * k, a1.wlo * k, a1.whi * k);
                // first and last tap of the axis: samples grow with (bin, sample), so a min / max over the 7 bins
                int tmin = rec_ok ? (a0.ok ? a0.lo : (a1.ok ? a1.lo : 0x7FFFFFFF)) : 0x7FFFFFFF;
                int tmax = rec_ok ? (a1.ok ? a1.hi : (a0.ok ? a0.hi : -1)) : -1;
#pragma unroll
                for (int o = 4; o >= 1; o >>= 1) { // lanes 0..7 and 8..15 reduce among themselves
                    tmin = min(tmin, __shfl_xor(tmin, o));
                    tmax = max(tmax, __shfl_xor(tmax, o));
                }
                const int xmin = __shfl(tmin, 8), xmax = __shfl(tmax, 8);
                if (lane == 0) {
                    const bool any = tmin <= tmax && xmin <= xmax;
                    s_rng[wave] = any ? make_int4(tmin, tmax, xmin, xmax) : make_int4(1, 0, 1, 0);
                }
            }
        }
        __syncthreads();
        if (wave >= nc) continue; // (a chunk's tail: this wave has no channel, it only helped with the records)
        const int nb = min(nw, j_end - j0);
        for (int e = 0; e < nb; ++e) {
            const int4 rng = s_rng[e];
            if (rng.x > rng.y) continue; // not this level's RoI, or every sample outside the map
            const float g = lane < 49 ? gb[(long long)(j0 + e) * C * 49 + lane] : 0.0f;
            for (int yb = rng.x; yb <= rng.y; yb += 64) {
                // this lane's row weights: row yb + lane against the 7 row records
                float wy[7];
#pragma unroll
                for (int ph = 0; ph < 7; ++ph) wy[ph] = axis_weight(s_ri[e * 16 + ph], s_rw[e * 16 + ph], yb + lane);
                for (int xb = rng.z; xb <= rng.w; xb += 64) {
                    const int x = xb + lane;
                    float wx[7];
#pragma unroll
                    for (int pw = 0; pw < 7; ++pw) wx[pw] = axis_weight(s_ri[e * 16 + 8 + pw], s_rw[e * 16 + 8 + pw], x);
                    float T[7];
#pragma unroll
                    for (int ph = 0; ph < 7; ++ph) {
                        float acc = 0.0f;
#pragma unroll
                        for (int pw = 0; pw < 7; ++pw) acc += lane_get(g, ph * 7 + pw) * wx[pw];
                        T[ph] = acc;
                    }
                    // rows that some sample touches (lane = row): the others are not visited at all
                    unsigned long long rows = __ballot(wy[0] != 0.0f || wy[1] != 0.0f || wy[2] != 0.0f || wy[3] != 0.0f ||
                                                       wy[4] != 0.0f || wy[5] != 0.0f || wy[6] != 0.0f);
                    while (rows != 0) {
                        const int r = __ffsll((long long)rows) - 1;
                        rows &= rows - 1;
                        float d = 0.0f;
#pragma unroll
                        for (int ph = 0; ph < 7; ++ph) d += lane_get(wy[ph], r) * T[ph];
                        if (x <= rng.w) plane[(yb + r) * W + x] += d;
                    }
                }
            }
        }
    }
    __syncthreads();
    float *out = grad_feat + ((long long)b * C + c0) * HW;
    if (splits == 1) {
        for (int t = threadIdx.x; t < nc * HW; t += blockDim.x) out[t] = s_g[t];
    } else {
        for (int t = threadIdx.x; t < nc * HW; t += blockDim.x) {
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
    // planes in LDS, one wave per channel: worth it from four channels (waves) per workgroup on
    if (pooled_h == 7 && pooled_w == 7 && sampling_ratio == 2 && plane_bytes * 4 <= RAB_LDS_BUDGET &&
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
        hipLaunchKernelGGL(roi_align_plane_backward_kernel, dim3((unsigned)(B * chunks * splits)), dim3(64 * CC),
                           lds, st, grad_out, C, H, W, (const float4 *)props, count, P, spatial_scale, level,
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
