// roi.hip -- K6 / K7: RoI list assembly, FPN level mapping, RoIAlign (bilinear average) and the
// max RoIPool variant.  Replaces torchvision.ops.MultiScaleRoIAlign as used by
// ROIHead/roi_pooling.py:44-53 (call site ROIHead/roi_head.py:127-130).
//
// The output (K, C, 7, 7) fp32 is 97 % of the hot path's HBM bytes: the kernels are
// organised around writing it in long contiguous runs.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "roi_common.h"

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
// One bilinear sample from a plane in global memory.  The low / high taps of a row are neighbours,
// so each row is ONE 8-byte load (the kernels that gather through L1 are bound by the number of
// tap loads); at the right edge, where both taps are the last pixel, the pair starts one pixel
// earlier.  Same four products, same order of additions as the scalar form.  Needs W >= 2.
__device__ __forceinline__ float sample_value(const float *__restrict__ f, const Sample &s)
{
    const int cl = (s.p2 == s.p1 && s.p1 > 0) ? 1 : 0;
    float2 r0, r1;
    __builtin_memcpy(&r0, f + (s.p1 - cl), 8);
    __builtin_memcpy(&r1, f + (s.p3 - cl), 8);
    const float v1 = cl ? r0.y : r0.x, v3 = cl ? r1.y : r1.x;
    return s.w1 * v1 + s.w2 * r0.y + s.w3 * v3 + s.w4 * r1.y;
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

// ---- batched RoIAlign over padded proposals ---------------------------------------------------
// The fast path of the headline workload.  One workgroup owns (image b, CC-channel chunk, RoI
// split): it first copies its CC feature planes into LDS, channel-interleaved in groups of four
// ([group][pixel] of float4), then sweeps its RoIs.  A lane produces one (RoI, 4 channels, bin)
// item: 2x2 bilinear samples x 4 taps, each tap ONE ds_read_b128 that serves 4 channels, i.e. the
// 16 taps per output come from LDS (256 B/clk/CU) instead of the vector L1 (64 B/clk/CU), and HBM
// sees the feature map once plus the output stream.  A wave takes one whole RoI at a time with
// lane = bin, so each channel store is one 196-byte run and the 16 stores of an iteration fill a
// contiguous 3136-byte piece of the (K, C, 7, 7) output.

// Staging of a chunk's planes (shared by the RoIAlign and max-RoIPool plane kernels): 4 coalesced plane reads ->
// one 16-byte LDS write per pixel/group.  Four pixels per thread are fetched before any is written, so a
// workgroup pays about three memory round trips for its 150 KB instead of nine.
__device__ __forceinline__ void rap_stage_planes(float4 *s_plane, const float *__restrict__ fb, int G, int nc, int HW,
                                                 int W, int WS)
{
    for (int t0 = threadIdx.x; t0 < G * HW; t0 += 4 * RAP_THREADS) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u * RAP_THREADS;
            v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (t < G * HW) {
                const int g = t / HW;
                const int p = t - g * HW;
                const float *f = fb + (long long)(4 * g) * HW + p;
                v[u].x = f[0];
                if (4 * g + 1 < nc) v[u].y = f[HW];
                if (4 * g + 2 < nc) v[u].z = f[2 * HW];
                if (4 * g + 3 < nc) v[u].w = f[3 * HW];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u * RAP_THREADS;
            if (t < G * HW) {
                const int g = t / HW;
                const int p = t - g * HW;
                const int y = p / W;
                const int x = p - y * W;
                s_plane[(y * G + g) * WS + (x ^ rap_swz(y))] = v[u];
            }
        }
    }
}

// What bounds this kernel (A/B builds on MI355X, batch 16 x 2000 RoIs, C = 512; DESIGN.md has the
// table): the OUTPUT STORE STREAM.  With the arithmetic removed the stores alone take 0.77 ms of
// the 0.91 ms; the arithmetic alone (LDS taps + packed FMAs + per-item set-up) takes 0.66 ms and
// overlaps with them only in part.  How the 3.2 GB are handed to the memory system matters more
// than anything computed here: the same bytes written as 64-item slices that straddle RoIs took
// 1.07 ms, as whole (RoI, chunk) pieces per wave 0.77 ms (tools/store_probe.hip), and with the
// XCD-grouped block order below 0.62 ms (the kernel: 0.66 ms).
// GT  = channel groups per chunk known at compile time (4 = a full 16-channel chunk) or 0.
// WST = compile-time LDS row stride in pixels (64 covers W <= 64) or 0 for a runtime stride.
template <int GT, int WST>
__global__ void __launch_bounds__(RAP_THREADS)
roi_align_plane_kernel(const float *__restrict__ feat, int C, int H, int W,
                       const float4 *__restrict__ props, const int32_t *__restrict__ count, int P,
                       int CC, float scale, int level, int n_levels, int k_min, int k_max,
                       int chunks, int splits, int B, float *__restrict__ out, int32_t *__restrict__ total_out)
{
    extern __shared__ float4 s_plane[];
    __shared__ float4 s_rec[RAP_SEG]; // per RoI: start_w, start_h, bin_w, bin_h (feature pixels)
    __shared__ signed char s_lvl[RAP_SEG]; // per RoI: pyramid level that pools it, -1 = none
    __shared__ int s_off, s_cnt;
    // Block order: the workgroups of all channel chunks of one (image, RoI split) sit 8 apart in the 1-D
    // grid.  Blocks i and i + 8 are dealt to the same XCD (observed dispatch order, used for speed only),
    // so the neighbouring 3,136-byte pieces that the chunks of a RoI row write at about the same time
    // meet in ONE L2 and leave it as whole lines: tools/store_probe.hip, the same stores 0.73 -> 0.62 ms.
    const int slot = blockIdx.x >> 3;
    const int chunk = slot % chunks;
    const int grp = (slot / chunks) * 8 + (blockIdx.x & 7); // (image, split); padded to a multiple of 8
    if (grp >= splits * B) return;
    const int split = grp % splits;
    const int b = grp / splits;
    const int c0 = chunk * CC;
    const int nc = min(CC, C - c0);
    const int G = GT ? GT : (nc + 3) >> 2;
    const int HW = H * W;
    const int WS = WST ? WST : (W + 15) & ~15;
    if (threadIdx.x == 0) {
        int off = 0, tot = 0;
        for (int i = 0; i < B; ++i) {
            int c = count ? count[i] : P;
            c = c < 0 ? 0 : (c > P ? P : c);
            if (i < b) off += c;
            if (i == b) s_cnt = c;
            tot += c;
        }
        s_off = off;
        if (total_out && chunk == 0 && grp == 0) *total_out = tot;
    }
    const float4 *pb = props + (long long)b * P;
    if (n_levels > 1) {
        // a pyramid level none of this split's RoIs maps to (the coarsest ones, mostly) is not staged at all
        __syncthreads();
        const int per_ = (s_cnt + splits - 1) / splits;
        int mine = 0;
        for (int i = split * per_ + (int)threadIdx.x; i < min(s_cnt, (split + 1) * per_); i += RAP_THREADS) {
            const int lv = map_level(pb[i], k_min, k_max);
            mine |= (lv == level) || ((lv < 0 || lv >= n_levels) && level == 0);
        }
        if (!__syncthreads_or(mine)) return;
    }
    rap_stage_planes(s_plane, feat + ((long long)b * C + c0) * HW, G, nc, HW, W, WS);
    __syncthreads();
    const int cnt = s_cnt;
    const int per = (cnt + splits - 1) / splits;
    const int j0 = split * per;
    const int j1 = min(cnt, j0 + per);
    const int row_stride = G * WS;
    float *ob = out + ((long long)s_off * C + c0) * 49;
    // RoIs are taken RAP_SEG at a time: their geometry goes to LDS first, so that the item loop
    // below issues NO global loads.  The memory counter retires loads and stores in order; a box
    // load inside the loop made every iteration wait for the previous iteration's 16 stores to
    // reach L2 (a full store round trip per item -- that stall, not LDS or ALU work, bounded the
    // kernel).  With the loop load-free the stores are fire-and-forget.
    for (int seg = j0; seg < j1; seg += RAP_SEG) {
        const int nseg = min(RAP_SEG, j1 - seg);
        __syncthreads(); // readers of the previous segment's records are done
        for (int i = threadIdx.x; i < nseg; i += RAP_THREADS) {
            float4 box = pb[seg + i];
            int lv = 0;
            if (n_levels > 1) {
                lv = map_level(box, k_min, k_max);
                lv = (lv < 0 || lv >= n_levels) ? -1 : lv;
            }
            float start_w = box.x * scale, start_h = box.y * scale;
            float roi_w = box.z * scale - start_w, roi_h = box.w * scale - start_h;
            roi_w = roi_w > 1.0f ? roi_w : 1.0f;
            roi_h = roi_h > 1.0f ? roi_h : 1.0f;
            s_rec[i] = make_float4(start_w, start_h, roi_w / 7.0f, roi_h / 7.0f);
            s_lvl[i] = (signed char)lv;
        }
        __syncthreads();
        // item = (RoI, bin): its tap addresses and weights are computed once and reused for every
        // channel group of the chunk.  A wave takes ONE whole RoI per iteration (lane = bin, 49 of
        // 64 lanes busy): its 16 channel stores then fill one
        // 3136-byte (RoI, chunk) piece of the output back to back, which the memory system takes
        // 1.5x faster than the same bytes written as 64-item slices that straddle two RoIs and are
        // completed by another wave (tools/store_probe.hip, modes 16/20/23) -- worth more here than
        // the idle quarter of the lanes.  It also makes the level test, the zero fill and the
        // 3-tap choice below uniform per RoI.
        // Wave w takes RoIs w, w+16, ...: at any time the 16 waves of the workgroup -- and, in step with
        // them, the workgroups of the image's other channel chunks -- write 16 neighbouring output rows.
        const int bin = threadIdx.x & 63;
        const int ph = bin / 7, pw = bin - ph * 7;
        for (int jl = (int)(threadIdx.x >> 6); jl < (bin < 49 ? nseg : 0); jl += RAP_THREADS / 64) { // lanes 49..63 idle
            const float4 rec = s_rec[jl];
            float *o = ob + (long long)(seg + jl) * C * 49 + bin;
            if (n_levels > 1) {
                const int lv = s_lvl[jl];
                if (lv < 0 && level == 0) { // no level takes it: level 0's pass writes zeros
                    for (int c = 0; c < nc; ++c) o[c * 49] = 0.0f;
                }
                if (lv != level) continue;
            }
            const float start_w = rec.x, start_h = rec.y, bin_w = rec.z, bin_h = rec.w;
            AxisSample ys[2], xs[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ys[i] = axis_sample(start_h + (float)ph * bin_h + ((float)i + 0.5f) * bin_h / 2.0f, H);
                xs[i] = axis_sample(start_w + (float)pw * bin_w + ((float)i + 0.5f) * bin_w / 2.0f, W);
            }
            // the 1/4 of the 2x2 average is folded into the row weights (exact: power of two);
            // the 3-tap forms are chosen per wave so that the branch is uniform
            const bool y3 = __all(axis_mergeable(ys[0], ys[1]));
            const bool x3 = __all(axis_mergeable(xs[0], xs[1]));
            if (y3) {
                const AxisTaps ty = axis_taps3(ys[0], ys[1], 0.25f, H);
                if (x3)
                    rap_item<3, 3, GT>(s_plane, WS, row_stride, G, nc, ty, axis_taps3(xs[0], xs[1], 1.0f, W), o);
                else
                    rap_item<3, 4, GT>(s_plane, WS, row_stride, G, nc, ty, axis_taps4(xs[0], xs[1], 1.0f), o);
            } else {
                const AxisTaps ty = axis_taps4(ys[0], ys[1], 0.25f);
                if (x3)
                    rap_item<4, 3, GT>(s_plane, WS, row_stride, G, nc, ty, axis_taps3(xs[0], xs[1], 1.0f, W), o);
                else
                    rap_item<4, 4, GT>(s_plane, WS, row_stride, G, nc, ty, axis_taps4(xs[0], xs[1], 1.0f), o);
            }
        }
    }
}

// ---- planes too large for LDS (FPN P3 100x168, P2 200x336): windows of the plane -----------------
// A RoI that the level mapper sends to a fine level is small in feature pixels, so the plane kernel's
// scheme still works on a WINDOW of the plane: RAW_WIN x RAW_WIN pixels x 16 channels in LDS, windows
// every RAW_STEP pixels.  A RoI belongs to the window that holds its first tap, and fits when its last
// tap is inside too -- always the case when it spans at most RAW_WIN - RAW_STEP + 1 = 33 cells per
// axis; the rare larger ones are left to the direct kernel, which is then launched with the same test
// inverted.  Workgroup = (16-channel chunk, window, image); a window no RoI of the image falls into
// costs one pass over the boxes and no staging.
#ifndef RAW_WG_TARGET
#define RAW_WG_TARGET 1024
#endif

// taps of one axis relative to the window origin; zero-weight cells are pulled inside the window
__device__ __forceinline__ void taps_to_window(AxisTaps &t, int n_taps, int origin)
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n_taps) t.idx[i] = min(max(t.idx[i] - origin, 0), RAW_WIN - 1);
}

// Staging of a window: thread t owns pixel column t % RAW_WIN of rows t / RAW_WIN + 16 k, for every
// channel group -- offsets between a thread's slots are compile-time multiples of the plane and row
// strides, and the LDS swizzle (low four bits of the row) is the same for all of them.
#define RAW_ROWS (RAP_THREADS / RAW_WIN)           // rows per pass of the workgroup
#define RAW_PASSES (RAW_WIN / RAW_ROWS)            // passes per channel group
#define RAW_SLOTS (RAW_G * RAW_PASSES)             // float4 registers per thread
static_assert(RAP_THREADS % RAW_WIN == 0 && RAW_WIN % RAW_ROWS == 0 && RAW_ROWS % 16 == 0, "window staging layout");

__device__ __forceinline__ void raw_fetch(float4 (&v)[RAW_SLOTS], const float *__restrict__ feat, int b, int C,
                                          int HW, int W, int ch, int wy0, int wx0, int wh, int ww)
{
    const int c0 = ch * (4 * RAW_G);
    const int nc = min(4 * RAW_G, C - c0);
    const int lx = threadIdx.x % RAW_WIN, ly0 = threadIdx.x / RAW_WIN;
    const float *f0 = feat + ((long long)b * C + c0) * HW + (long long)(wy0 + ly0) * W + wx0 + lx;
#pragma unroll
    for (int u = 0; u < RAW_SLOTS; ++u) {
        const int g = u / RAW_PASSES, k = u % RAW_PASSES;
        v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (lx < ww && ly0 + RAW_ROWS * k < wh && 4 * g < nc) {
            const float *f = f0 + (long long)(4 * g) * HW + (long long)(RAW_ROWS * k) * W;
            v[u].x = f[0];
            if (4 * g + 1 < nc) v[u].y = f[HW];
            if (4 * g + 2 < nc) v[u].z = f[2 * HW];
            if (4 * g + 3 < nc) v[u].w = f[3 * HW];
        }
    }
}

__device__ __forceinline__ void raw_store(float4 *s_plane, const float4 (&v)[RAW_SLOTS], int G, int wh, int ww)
{
    const int lx = threadIdx.x % RAW_WIN, ly0 = threadIdx.x / RAW_WIN;
    const int col = lx ^ rap_swz(ly0); // rows RAW_ROWS apart share their low four bits
#pragma unroll
    for (int u = 0; u < RAW_SLOTS; ++u) {
        const int g = u / RAW_PASSES, ly = ly0 + RAW_ROWS * (u % RAW_PASSES);
        if (lx < ww && ly < wh && g < G) s_plane[(ly * G + g) * RAW_WIN + col] = v[u];
    }
}

#define RAW_LIST 1024 // RoIs of one window gathered before its chunks are staged
#define RAW_SCAN 512  // boxes examined between two looks at the list's fill

template <int GT>
__global__ void __launch_bounds__(RAP_THREADS)
roi_align_window_kernel(const float *__restrict__ feat, int C, int H, int W,
                        const float4 *__restrict__ props, const int32_t *__restrict__ count, int P,
                        float scale, int level, int n_levels, int k_min, int k_max, int nwy, int nwx,
                        int csplit, int B, float *__restrict__ out, int32_t *__restrict__ total_out)
{
    extern __shared__ float4 s_plane[];
    __shared__ float4 s_rec[RAW_LIST];
    __shared__ int s_idx[RAW_LIST];
    __shared__ int s_off, s_cnt, s_n;
    // Block order.  Blocks i and i + 8 share an XCD (and its L2), and windows overlap by half: an XCD is given
    // whole IMAGES (batches of 8 or more) or whole window rows of an image, neighbouring windows in consecutive
    // blocks, so that a pixel staged for one window is still in that L2 when the next window asks for it --
    // the 4x re-read of the maps then stays on the XCD instead of crossing the fabric.  The channel splits of
    // a (window, image) are consecutive too (they share the output lines of their RoIs).
    const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int cs = slot % csplit;
    const int r = slot / csplit;
    int b, wid;
    if (B >= 8) {
        wid = r % (nwy * nwx);
        b = (r / (nwy * nwx)) * 8 + xcd;
        if (b >= B) return;
    } else {
        const int u = (r / nwx) * 8 + xcd; // (image, window row)
        if (u >= B * nwy) return;
        b = u / nwy;
        wid = (u % nwy) * nwx + r % nwx;
    }
    const int chunks = (C + 4 * RAW_G - 1) / (4 * RAW_G);
    const int HW = H * W;
    const int wy0 = (wid / nwx) * RAW_STEP, wx0 = (wid % nwx) * RAW_STEP;
    if (threadIdx.x == 0) {
        int off = 0, tot = 0;
        for (int i = 0; i < B; ++i) {
            int c = count ? count[i] : P;
            c = c < 0 ? 0 : (c > P ? P : c);
            if (i < b) off += c;
            if (i == b) s_cnt = c;
            tot += c;
        }
        s_off = off;
        s_n = 0;
        if (total_out && cs == 0 && b == 0 && wid == 0) *total_out = tot;
    }
    __syncthreads();
    const int cnt = s_cnt;
    const float4 *pb = props + (long long)b * P;
    const int wh = min(RAW_WIN, H - wy0), ww = min(RAW_WIN, W - wx0);
    const int bin = threadIdx.x & 63;
    const int ph = bin / 7, pw = bin - ph * 7;
    // The boxes are examined ONCE per workgroup: the window's RoIs collect in a list, and the list is
    // worked off for every channel chunk of this workgroup (cs, + csplit, ...) before it
    // can overflow.  A window no RoI of the image falls into costs that one pass and no staging.
    for (int seg = 0; seg < cnt; seg += RAW_SCAN) {
        const int nseg = min(RAW_SCAN, cnt - seg);
        for (int i = threadIdx.x; i < nseg; i += RAP_THREADS) {
            const float4 box = pb[seg + i];
            int lv = 0;
            if (n_levels > 1) lv = map_level(box, k_min, k_max);
            const float4 rec = roi_record(box, scale);
            const RoiWindow rw = roi_window(rec, H, W, nwy, nwx);
            if (lv == level && rw.fits && rw.win == wid) {
                const int pos = atomicAdd(&s_n, 1);
                s_rec[pos] = rec;
                s_idx[pos] = seg + i;
            }
        }
        __syncthreads();
        const int n_list = s_n;
        __syncthreads(); // everyone has read the fill before the next scan adds to it
        if (n_list <= RAW_LIST - RAW_SCAN && seg + RAW_SCAN < cnt) continue; // room for another scan
        for (int ch = cs; ch < (n_list ? chunks : 0); ch += csplit) {
            const int c0 = ch * (4 * RAW_G);
            const int nc = min(4 * RAW_G, C - c0);
            const int G = GT ? GT : (nc + 3) >> 2;
            __syncthreads(); // the previous chunk's readers are done with the window
            {
                // (fetching the next chunk's window into registers while this one's items run was tried:
                // the item code needs the registers, the fetched window spilled, 0.67 -> 0.88 ms)
                float4 v[RAW_SLOTS];
                raw_fetch(v, feat, b, C, HW, W, ch, wy0, wx0, wh, ww);
                raw_store(s_plane, v, G, wh, ww);
            }
            __syncthreads();
            const int row_stride = G * RAW_WIN;
            float *ob = out + ((long long)s_off * C + c0) * 49;
            // one whole RoI per wave iteration, lane = bin (see roi_align_plane_kernel)
            for (int e = (int)(threadIdx.x >> 6); e < (bin < 49 ? n_list : 0); e += RAP_THREADS / 64) {
                const float4 rec = s_rec[e];
                float *o = ob + (long long)s_idx[e] * C * 49 + bin;
                const float start_w = rec.x, start_h = rec.y, bin_w = rec.z, bin_h = rec.w;
                AxisSample ys[2], xs[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ys[i] = axis_sample(start_h + (float)ph * bin_h + ((float)i + 0.5f) * bin_h / 2.0f, H);
                    xs[i] = axis_sample(start_w + (float)pw * bin_w + ((float)i + 0.5f) * bin_w / 2.0f, W);
                }
                const bool y3 = __all(axis_mergeable(ys[0], ys[1]));
                const bool x3 = __all(axis_mergeable(xs[0], xs[1]));
                AxisTaps ty = y3 ? axis_taps3(ys[0], ys[1], 0.25f, H) : axis_taps4(ys[0], ys[1], 0.25f);
                AxisTaps tx = x3 ? axis_taps3(xs[0], xs[1], 1.0f, W) : axis_taps4(xs[0], xs[1], 1.0f);
                taps_to_window(ty, y3 ? 3 : 4, wy0);
                taps_to_window(tx, x3 ? 3 : 4, wx0);
                if (y3) {
                    if (x3)
                        rap_item<3, 3, GT>(s_plane, RAW_WIN, row_stride, G, nc, ty, tx, o);
                    else
                        rap_item<3, 4, GT>(s_plane, RAW_WIN, row_stride, G, nc, ty, tx, o);
                } else {
                    if (x3)
                        rap_item<4, 3, GT>(s_plane, RAW_WIN, row_stride, G, nc, ty, tx, o);
                    else
                        rap_item<4, 4, GT>(s_plane, RAW_WIN, row_stride, G, nc, ty, tx, o);
                }
            }
        }
        __syncthreads(); // every reader of the list is done
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
    }
}

// Generic batched fallback (any pooled size / sampling ratio / plane size): block = (RoI split,
// 32-channel chunk, image); per RoI the sample table is rebuilt in LDS, taps come through L1.
__global__ void __launch_bounds__(256)
roi_align_batched_direct_kernel(const float *__restrict__ feat, int C, int H, int W,
                                const float4 *__restrict__ props, const int32_t *__restrict__ count,
                                int P, int pooled_h, int pooled_w, float scale, int sampling_ratio,
                                int level, int n_levels, int k_min, int k_max, int nwy, int nwx,
                                float *__restrict__ out, int32_t *__restrict__ total_out)
{
    __shared__ Sample s_samp[RA_MAX_SAMPLES];
    __shared__ int s_list[256];
    __shared__ int s_off, s_cnt, s_nl;
    const int b = blockIdx.z;
    const int B = gridDim.z;
    if (threadIdx.x == 0) {
        int off = 0, tot = 0;
        for (int i = 0; i < B; ++i) {
            int c = count ? count[i] : P;
            c = c < 0 ? 0 : (c > P ? P : c);
            if (i < b) off += c;
            if (i == b) s_cnt = c;
            tot += c;
        }
        s_off = off;
        if (total_out && blockIdx.x == 0 && blockIdx.y == 0 && b == 0) *total_out = tot;
    }
    __syncthreads();
    const int cnt = s_cnt;
    const int bins = pooled_h * pooled_w;
    const int per_bin = sampling_ratio * sampling_ratio;
    const int nsamp = bins * per_bin;
    const int c0 = blockIdx.y * RA_CHUNK;
    const int nc = min(RA_CHUNK, C - c0);
    const float *fb = feat + ((long long)b * C + c0) * H * W;
    // Which of this workgroup's RoIs (blockIdx.x, + gridDim.x, ...) need work is decided by all threads
    // at once -- one box per thread, one memory latency -- instead of one box after the other.
    for (int base = 0; base < cnt; base += gridDim.x * blockDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) s_nl = 0;
        __syncthreads();
        {
            const int j = base + blockIdx.x + gridDim.x * threadIdx.x;
            int code = -1;
            if (j < cnt) {
                const float4 box = props[(long long)b * P + j];
                bool todo = true, zero = false;
                if (n_levels > 1) {
                    const int lv = map_level(box, k_min, k_max);
                    zero = (lv < 0 || lv >= n_levels) && level == 0; // no level pools it: zeros, written once
                    todo = zero || lv == level;
                }
                // behind the window kernel (nwy > 0): only the RoIs that fit no window are left to do
                if (todo && !zero && nwy > 0 && roi_window(roi_record(box, scale), H, W, nwy, nwx).fits) todo = false;
                // behind the tile kernel (nwy < 0: -nwy x -nwx tiles), which has also written the zeros
                if (nwy < 0 && (zero || (todo && tile_roi_fits(roi_record(box, scale), H, W, -nwy, -nwx)))) todo = false;
                if (todo) code = zero ? (j | 0x40000000) : j;
            }
            if (code >= 0) s_list[atomicAdd(&s_nl, 1)] = code;
        }
        __syncthreads();
        const int n_list = s_nl;
        for (int e = 0; e < n_list; ++e) {
            const int code = s_list[e];
            const int j = code & 0x3FFFFFFF;
            float4 box = props[(long long)b * P + j];
            float *ob = out + ((long long)(s_off + j) * C + c0) * bins;
            if (code & 0x40000000) {
                for (int t = threadIdx.x; t < nc * bins; t += blockDim.x) ob[t] = 0.0f;
                continue;
            }
            float start_w = box.x * scale, start_h = box.y * scale;
            float roi_w = box.z * scale - start_w, roi_h = box.w * scale - start_h;
            roi_w = roi_w > 1.0f ? roi_w : 1.0f;
            roi_h = roi_h > 1.0f ? roi_h : 1.0f;
            float bin_h = roi_h / (float)pooled_h, bin_w = roi_w / (float)pooled_w;
            __syncthreads(); // previous RoI's readers are done with s_samp
            for (int t = threadIdx.x; t < nsamp; t += blockDim.x) {
                int bin = t / per_bin;
                int si = t - bin * per_bin;
                int iy = si / sampling_ratio, ix = si - iy * sampling_ratio;
                int ph = bin / pooled_w, pw = bin - ph * pooled_w;
                float yy = start_h + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)sampling_ratio;
                float xx = start_w + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)sampling_ratio;
                s_samp[t] = make_sample(yy, xx, H, W);
            }
            __syncthreads();
#pragma unroll 3
            for (int t = threadIdx.x; t < nc * bins; t += blockDim.x) {
                int c = t / bins;
                int bin = t - c * bins;
                const float *f = fb + (long long)c * H * W;
                float acc = 0.0f;
                const Sample *sp = s_samp + bin * per_bin;
                if (W >= 2) {
#pragma unroll 4
                    for (int si = 0; si < per_bin; ++si) acc += sample_value(f, sp[si]);
                } else {
                    for (int si = 0; si < per_bin; ++si) {
                        Sample s = sp[si];
                        acc += s.w1 * f[s.p1] + s.w2 * f[s.p2] + s.w3 * f[s.p3] + s.w4 * f[s.p4];
                    }
                }
                ob[t] = acc / (float)per_bin;
            }
        }
    }
}

#ifndef RAP_WG_TARGET
#define RAP_WG_TARGET (4 * 256)
#endif
#ifndef RAP_WG_TARGET_ML
#define RAP_WG_TARGET_ML 256 // config 5: P4 191 -> 167 us, P5 + P6 96 -> 62 us against 1024
#endif
#ifndef RAD_SPLITS_BEHIND
#define RAD_SPLITS_BEHIND 16 // config 5: 128 -> 86 us against 128 splits (32: 88, 64: 101)
#endif

extern "C" int frcnn_roi_align_batched_f32(const float *feat, int B, int C, int H, int W,
                                           const float *props, const int32_t *count, int P,
                                           int pooled_h, int pooled_w, float spatial_scale,
                                           int sampling_ratio, int level, int n_levels, int k_min,
                                           int k_max, float *out, int32_t *total_out, void *stream)
{
    FRCNN_REQUIRE(B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && P >= 0 && pooled_h > 0 && pooled_w > 0,
                  "roi_align_batched: bad shape");
    FRCNN_REQUIRE(n_levels >= 1 && level >= 0 && level < n_levels, "roi_align_batched: bad level");
    if (P == 0) {
        if (total_out) {
            hipError_t e = hipMemsetAsync(total_out, 0, sizeof(int32_t), (hipStream_t)stream);
            FRCNN_REQUIRE(e == hipSuccess, "roi_align_batched: memset failed");
        }
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(feat && props && out, "roi_align_batched: null pointer");
    FRCNN_REQUIRE((long long)H * W < (1ll << 30), "roi_align_batched: feature plane too large");
    FRCNN_REQUIRE(sampling_ratio >= 1, "roi_align_batched: sampling_ratio must be >= 1 (got %d)", sampling_ratio);
    FRCNN_REQUIRE((long long)pooled_h * pooled_w * sampling_ratio * sampling_ratio <= RA_MAX_SAMPLES,
                  "roi_align_batched: pooled_h*pooled_w*sampling_ratio^2 exceeds %d", RA_MAX_SAMPLES);
    // LDS row stride of the plane kernel: the compile-time 64 for every map at most 64 pixels wide whose 16
    // channels still fit (coarse pyramid levels included), else W rounded up to 16
    int WS = (W + 15) & ~15;
    if (W <= 64 && (long long)H * 64 * 4 * 16 <= RAP_LDS_BUDGET) WS = 64;
    const long long plane_bytes = (long long)H * WS * 4; // LDS bytes per channel
    if (pooled_h == 7 && pooled_w == 7 && sampling_ratio == 2 && plane_bytes * 4 <= RAP_LDS_BUDGET &&
        (long long)P * 49 < (1ll << 30) && (long long)H * WS * 4 < (1 << 23)) {
        int CC = 16;
        while (CC > 4 && plane_bytes * CC > RAP_LDS_BUDGET) CC >>= 1;
        if (CC > C) CC = (C + 3) / 4 * 4;
        const int chunks = (C + CC - 1) / CC;
        // RoI splits: about four workgroups per CU.  Fewer, longer workgroups stage fewer planes
        // (0.745 ms at 2 per CU, 0.748 at 4, 0.770 at 8, 0.838 at 16); shorter ones let the small
        // kernels of the next batch on another stream find a free CU sooner (bench.py pipelining).
        // (one level of several: only a share of the RoIs is pooled here, and every RoI split stages the planes
        // again -- fewer, longer workgroups)
        const int wg_target = n_levels > 1 ? RAP_WG_TARGET_ML : RAP_WG_TARGET;
        int splits = (wg_target + chunks * B - 1) / (chunks * B);
        if (splits < 1) splits = 1;
        if (splits > 16) splits = 16;
        size_t lds = (size_t)plane_bytes * (size_t)CC;
        // fast instantiation: full 16-channel chunks on a map at most 64 pixels wide
        const bool full = (CC == 16) && (C % 16 == 0) && WS == 64;
        // wider maps whose chunks are 8 whole channels (FPN P4 at 800 x 1344): channel groups known, stride not
        const bool half = !full && (CC == 8) && (C % 8 == 0);
        {
            static size_t granted4[FRCNN_MAX_DEVICES], granted2[FRCNN_MAX_DEVICES], granted0[FRCNN_MAX_DEVICES];
            hipError_t e = full   ? frcnn_reserve_dynamic_lds((const void *)roi_align_plane_kernel<4, 64>, lds, granted4)
                           : half ? frcnn_reserve_dynamic_lds((const void *)roi_align_plane_kernel<2, 0>, lds, granted2)
                                  : frcnn_reserve_dynamic_lds((const void *)roi_align_plane_kernel<0, 0>, lds, granted0);
            FRCNN_REQUIRE(e == hipSuccess, "roi_align_batched: cannot reserve %zu B of LDS: %s", lds,
                          hipGetErrorString(e));
        }
        const long long groups8 = ((long long)splits * B + 7) / 8 * 8; // (image, split) groups, padded (see kernel)
        FRCNN_REQUIRE(groups8 * chunks < (1ll << 31), "roi_align_batched: grid too large");
        const dim3 grid((unsigned)(groups8 * chunks));
        if (full)
            hipLaunchKernelGGL((roi_align_plane_kernel<4, 64>), grid, dim3(RAP_THREADS), lds,
                               (hipStream_t)stream, feat, C, H, W, (const float4 *)props, count, P, CC,
                               spatial_scale, level, n_levels, k_min, k_max, chunks, splits, B, out, total_out);
        else if (half)
            hipLaunchKernelGGL((roi_align_plane_kernel<2, 0>), grid, dim3(RAP_THREADS), lds,
                               (hipStream_t)stream, feat, C, H, W, (const float4 *)props, count, P, CC,
                               spatial_scale, level, n_levels, k_min, k_max, chunks, splits, B, out, total_out);
        else
            hipLaunchKernelGGL((roi_align_plane_kernel<0, 0>), grid, dim3(RAP_THREADS), lds,
                               (hipStream_t)stream, feat, C, H, W, (const float4 *)props, count, P, CC,
                               spatial_scale, level, n_levels, k_min, k_max, chunks, splits, B, out, total_out);
        FRCNN_CHECK_LAUNCH("roi_align_plane");
        return FRCNN_OK;
    }
    int nwy = 0, nwx = 0;
    if (pooled_h == 7 && pooled_w == 7 && sampling_ratio == 2 && (long long)P * 49 < (1ll << 30) &&
        (long long)H * W < (1ll << 28)) {
        // plane too large for LDS: windows of it (fine pyramid levels, where RoIs are small)
        nwy = H <= RAW_WIN ? 1 : (H - RAW_WIN + RAW_STEP - 1) / RAW_STEP + 1;
        nwx = W <= RAW_WIN ? 1 : (W - RAW_WIN + RAW_STEP - 1) / RAW_STEP + 1;
        if ((long long)nwy * nwx > 65535) {
            nwy = nwx = 0;
        } else {
            // a workgroup examines the image's boxes once and then works through several channel chunks
            // of its window; as many channel splits as give the chip about four workgroups per CU
            const int chunks = (C + 4 * RAW_G - 1) / (4 * RAW_G);
            int chunks16 = (RAW_WG_TARGET + nwy * nwx * B - 1) / (nwy * nwx * B);
            chunks16 = chunks16 < 1 ? 1 : (chunks16 > chunks ? chunks : chunks16);
            const size_t lds = (size_t)RAW_WIN * RAW_WIN * 16 * RAW_G;
            const bool full = C % (4 * RAW_G) == 0;
            {
                static size_t grantedg[FRCNN_MAX_DEVICES], granted0[FRCNN_MAX_DEVICES];
                hipError_t e =
                    full ? frcnn_reserve_dynamic_lds((const void *)roi_align_window_kernel<RAW_G>, lds, grantedg)
                         : frcnn_reserve_dynamic_lds((const void *)roi_align_window_kernel<0>, lds, granted0);
                FRCNN_REQUIRE(e == hipSuccess, "roi_align_batched: cannot reserve %zu B of LDS: %s", lds,
                              hipGetErrorString(e));
            }
            // (the kernel's block order: whole images per XCD from 8 images on, else whole window rows)
            const long long units = B >= 8 ? B : (long long)B * nwy, per_unit = B >= 8 ? (long long)nwy * nwx : nwx;
            const long long groups8 = (units + 7) / 8 * per_unit * 8;
            FRCNN_REQUIRE(groups8 * chunks16 < (1ll << 31), "roi_align_batched: grid too large");
            const dim3 grid((unsigned)(groups8 * chunks16));
            if (full)
                hipLaunchKernelGGL((roi_align_window_kernel<RAW_G>), grid, dim3(RAP_THREADS), lds,
                                   (hipStream_t)stream, feat, C, H, W, (const float4 *)props, count, P,
                                   spatial_scale, level, n_levels, k_min, k_max, nwy, nwx, chunks16, B, out,
                                   total_out);
            else
                hipLaunchKernelGGL((roi_align_window_kernel<0>), grid, dim3(RAP_THREADS), lds,
                                   (hipStream_t)stream, feat, C, H, W, (const float4 *)props, count, P,
                                   spatial_scale, level, n_levels, k_min, k_max, nwy, nwx, chunks16, B, out,
                                   total_out);
            FRCNN_CHECK_LAUNCH("roi_align_window");
        }
    }
    int chunks = (C + RA_CHUNK - 1) / RA_CHUNK;
    // behind the window kernel only the few oversize RoIs are left: fewer workgroups, several RoIs each
    const int want = nwy > 0 ? RAD_SPLITS_BEHIND : 128;
    int splits = P < want ? P : want;
    hipLaunchKernelGGL(roi_align_batched_direct_kernel, dim3(splits, chunks, B), dim3(256), 0,
                       (hipStream_t)stream, feat, C, H, W, (const float4 *)props, count, P, pooled_h,
                       pooled_w, spatial_scale, sampling_ratio, level, n_levels, k_min, k_max, nwy, nwx, out,
                       total_out);
    FRCNN_CHECK_LAUNCH("roi_align_batched_direct");
    return FRCNN_OK;
}

int frcnn_roi_align_direct_behind_tiles(const float *feat, int B, int C, int H, int W, const float *props,
                                        const int32_t *count, int P, float spatial_scale, int level, int n_levels,
                                        int k_min, int k_max, float *out, void *stream)
{
    const int chunks = (C + RA_CHUNK - 1) / RA_CHUNK;
    FRCNN_REQUIRE(chunks <= 65535, "roi_align_levels: too many channels");
    const int splits = P < RAD_SPLITS_BEHIND ? P : RAD_SPLITS_BEHIND;
    hipLaunchKernelGGL(roi_align_batched_direct_kernel, dim3(splits, chunks, B), dim3(256), 0, (hipStream_t)stream, feat,
                       C, H, W, (const float4 *)props, count, P, 7, 7, spatial_scale, 2, level, n_levels, k_min, k_max,
                       -tile_count(H), -tile_count(W), out, (int32_t *)nullptr);
    FRCNN_CHECK_LAUNCH("roi_align_batched_direct (behind tiles)");
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

// ---- max RoIPool over padded proposals, on the plane scheme ------------------------------------------
// The batched form (K7 at the headline shape: 16 x 2,000 RoIs, C = 512).  Same organisation as
// roi_align_plane_kernel: a workgroup owns (image, 16-channel chunk, RoI split), keeps the chunk's planes in
// LDS as 4-channel pixels, and a wave takes one whole RoI per iteration with lane = bin, so that every
// channel's 49 values -- and its 49 argmax indices -- leave as one 196-byte run each.  A bin's window is
// scanned in row-major order with a strict `>`, which is torchvision's first-maximum rule; one LDS read
// serves four channels.  Output rows are compacted in image order like RoIAlign's.
struct PoolRec { // integer geometry of a RoI: round(coord * scale), extents max(end - start + 1, 1)
    int rsw, rsh, rw, rh;
};

template <int GT, int WST>
__global__ void __launch_bounds__(RAP_THREADS)
roi_pool_plane_kernel(const float *__restrict__ feat, int C, int H, int W, const float4 *__restrict__ props,
                      const int32_t *__restrict__ count, int P, int CC, float scale, int chunks, int splits, int B,
                      float *__restrict__ out, int32_t *__restrict__ argmax, int32_t *__restrict__ total_out)
{
    extern __shared__ float4 s_plane[];
    __shared__ int4 s_rec[RAP_SEG];
    __shared__ int s_off, s_cnt;
    const int slot = blockIdx.x >> 3; // block order as in roi_align_plane_kernel
    const int chunk = slot % chunks;
    const int grp = (slot / chunks) * 8 + (blockIdx.x & 7);
    if (grp >= splits * B) return;
    const int split = grp % splits;
    const int b = grp / splits;
    const int c0 = chunk * CC;
    const int nc = min(CC, C - c0);
    const int G = GT ? GT : (nc + 3) >> 2;
    const int HW = H * W;
    const int WS = WST ? WST : (W + 15) & ~15;
    if (threadIdx.x == 0) {
        int off = 0, tot = 0;
        for (int i = 0; i < B; ++i) {
            int c = count ? count[i] : P;
            c = c < 0 ? 0 : (c > P ? P : c);
            if (i < b) off += c;
            if (i == b) s_cnt = c;
            tot += c;
        }
        s_off = off;
        if (total_out && chunk == 0 && grp == 0) *total_out = tot;
    }
    rap_stage_planes(s_plane, feat + ((long long)b * C + c0) * HW, G, nc, HW, W, WS);
    __syncthreads();
    const int cnt = s_cnt;
    const int per = (cnt + splits - 1) / splits;
    const int j0 = split * per;
    const int j1 = min(cnt, j0 + per);
    const int row_stride = G * WS;
    const float4 *pb = props + (long long)b * P;
    const long long o0 = ((long long)s_off * C + c0) * 49;
    const int bin = threadIdx.x & 63;
    const int ph = bin / 7, pw = bin - ph * 7;
    for (int seg = j0; seg < j1; seg += RAP_SEG) {
        const int nseg = min(RAP_SEG, j1 - seg);
        __syncthreads();
        for (int i = threadIdx.x; i < nseg; i += RAP_THREADS) {
            const float4 box = pb[seg + i];
            int4 r;
            r.x = (int)roundf(box.x * scale);
            r.y = (int)roundf(box.y * scale);
            r.z = max((int)roundf(box.z * scale) - r.x + 1, 1);
            r.w = max((int)roundf(box.w * scale) - r.y + 1, 1);
            s_rec[i] = r;
        }
        __syncthreads();
        for (int jl = (int)(threadIdx.x >> 6); jl < (bin < 49 ? nseg : 0); jl += RAP_THREADS / 64) {
            const int4 r = s_rec[jl];
            const float bin_h = (float)r.w / 7.0f, bin_w = (float)r.z / 7.0f;
            int hs = (int)floorf((float)ph * bin_h), he = (int)ceilf((float)(ph + 1) * bin_h);
            int ws = (int)floorf((float)pw * bin_w), we = (int)ceilf((float)(pw + 1) * bin_w);
            hs = min(max(hs + r.y, 0), H);
            he = min(max(he + r.y, 0), H);
            ws = min(max(ws + r.x, 0), W);
            we = min(max(we + r.x, 0), W);
            const bool empty = (he <= hs) || (we <= ws);
            const long long o = o0 + (long long)(seg + jl) * C * 49 + bin;
            // all channel groups of the chunk per window cell: 16 running maxima and their indices
            float mx[4 * (GT ? GT : 4)];
            int mi[4 * (GT ? GT : 4)];
#pragma unroll
            for (int q = 0; q < 4 * (GT ? GT : 4); ++q) {
                mx[q] = empty ? 0.0f : -3.402823466e+38f;
                mi[q] = -1;
            }
            int h = hs, w = ws;
            for (int left = empty ? 0 : (he - hs) * (we - ws); left > 0; --left) {
                const float4 *cell = s_plane + __mul24(h, row_stride) + (w ^ rap_swz(h));
                const int idx = h * W + w;
#pragma unroll
                for (int g = 0; g < (GT ? GT : 4); ++g) {
                    if (GT || g < G) {
                        const float4 v = cell[g * WS];
                        if (v.x > mx[4 * g]) { mx[4 * g] = v.x; mi[4 * g] = idx; }
                        if (v.y > mx[4 * g + 1]) { mx[4 * g + 1] = v.y; mi[4 * g + 1] = idx; }
                        if (v.z > mx[4 * g + 2]) { mx[4 * g + 2] = v.z; mi[4 * g + 2] = idx; }
                        if (v.w > mx[4 * g + 3]) { mx[4 * g + 3] = v.w; mi[4 * g + 3] = idx; }
                    }
                }
                if (++w == we) {
                    w = ws;
                    ++h;
                }
            }
#pragma unroll
            for (int q = 0; q < 4 * (GT ? GT : 4); ++q) {
                if (GT || q < nc) {
                    out[o + q * 49] = mx[q];
                    argmax[o + q * 49] = mi[q];
                }
            }
        }
    }
}

extern "C" int frcnn_roi_pool_batched_f32(const float *feat, int B, int C, int H, int W, const float *props,
                                          const int32_t *count, int P, int pooled_h, int pooled_w,
                                          float spatial_scale, float *out, int32_t *argmax, int32_t *total_out,
                                          void *stream)
{
    FRCNN_REQUIRE(B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && P >= 0, "roi_pool_batched: bad shape");
    FRCNN_REQUIRE(pooled_h == 7 && pooled_w == 7, "roi_pool_batched: only 7 x 7 bins (got %d x %d); use frcnn_roi_pool_f32",
                  pooled_h, pooled_w);
    if (P == 0) {
        if (total_out) {
            hipError_t e = hipMemsetAsync(total_out, 0, sizeof(int32_t), (hipStream_t)stream);
            FRCNN_REQUIRE(e == hipSuccess, "roi_pool_batched: memset failed");
        }
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(feat && props && out && argmax, "roi_pool_batched: null pointer");
    int WS = (W + 15) & ~15;
    if (W <= 64 && (long long)H * 64 * 4 * 16 <= RAP_LDS_BUDGET) WS = 64;
    const long long plane_bytes = (long long)H * WS * 4;
    FRCNN_REQUIRE(plane_bytes * 4 <= RAP_LDS_BUDGET && (long long)P * 49 < (1ll << 30) && (long long)H * WS * 4 < (1 << 23),
                  "roi_pool_batched: a %d x %d plane does not fit LDS; use frcnn_roi_pool_f32", H, W);
    int CC = 16;
    while (CC > 4 && plane_bytes * CC > RAP_LDS_BUDGET) CC >>= 1;
    if (CC > C) CC = (C + 3) / 4 * 4;
    const int chunks = (C + CC - 1) / CC;
    int splits = (RAP_WG_TARGET + chunks * B - 1) / (chunks * B);
    if (splits < 1) splits = 1;
    if (splits > 16) splits = 16;
    const size_t lds = (size_t)plane_bytes * (size_t)CC;
    const bool full = (CC == 16) && (C % 16 == 0) && WS == 64;
    {
        static size_t granted4[FRCNN_MAX_DEVICES], granted0[FRCNN_MAX_DEVICES];
        hipError_t e = full ? frcnn_reserve_dynamic_lds((const void *)roi_pool_plane_kernel<4, 64>, lds, granted4)
                            : frcnn_reserve_dynamic_lds((const void *)roi_pool_plane_kernel<0, 0>, lds, granted0);
        FRCNN_REQUIRE(e == hipSuccess, "roi_pool_batched: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    }
    const long long groups8 = ((long long)splits * B + 7) / 8 * 8;
    FRCNN_REQUIRE(groups8 * chunks < (1ll << 31), "roi_pool_batched: grid too large");
    const dim3 grid((unsigned)(groups8 * chunks));
    if (full)
        hipLaunchKernelGGL((roi_pool_plane_kernel<4, 64>), grid, dim3(RAP_THREADS), lds, (hipStream_t)stream, feat, C, H,
                           W, (const float4 *)props, count, P, CC, spatial_scale, chunks, splits, B, out, argmax,
                           total_out);
    else
        hipLaunchKernelGGL((roi_pool_plane_kernel<0, 0>), grid, dim3(RAP_THREADS), lds, (hipStream_t)stream, feat, C, H,
                           W, (const float4 *)props, count, P, CC, spatial_scale, chunks, splits, B, out, argmax,
                           total_out);
    FRCNN_CHECK_LAUNCH("roi_pool_plane");
    return FRCNN_OK;
}
