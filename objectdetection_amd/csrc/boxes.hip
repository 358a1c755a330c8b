// boxes.hip -- anchors (K1), decode / clip / filter (K3), gather, IoU and target labelling (K5).
// All of these are HBM-/latency-bound elementwise or row-wise kernels: one 16-B box per lane,
// float4 loads and stores, batch dimension in the grid so B images are one launch.
#include <stdarg.h>
#include <math.h>

#include "common.h"

// ---- error plumbing ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void frcnn_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *frcnn_last_error_string(void) { return g_err; }
extern "C" int frcnn_abi_version(void) { return 1; }

// ---- K1 anchors -------------------------------------------------------------------------------
extern "C" int frcnn_base_anchors_f32(const float *size_host, const float *ratio_host, int na,
                                      float *out_host)
{
    FRCNN_REQUIRE(size_host && ratio_host && out_host && na > 0, "base_anchors: bad arguments");
    for (int a = 0; a < na; ++a) {
        volatile float h = size_host[a];
        volatile float w = size_host[a] * ratio_host[a];
        out_host[4 * a + 0] = -w / 2.0f;
        out_host[4 * a + 1] = -h / 2.0f;
        out_host[4 * a + 2] = w / 2.0f;
        out_host[4 * a + 3] = h / 2.0f;
    }
    return FRCNN_OK;
}

// One lane per anchor (16-B float4 store, fully coalesced: out is (cells, na, 4) row-major).
__global__ void __launch_bounds__(256) anchor_grid_kernel(const float4 *__restrict__ base, int na,
                                                          int fw, int total, float sy, float sx,
                                                          float4 *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int cell = i / na;
    int a = i - cell * na;
    int y = cell / fw;
    int x = cell - y * fw;
    float gx = (float)x * sx;
    float gy = (float)y * sy;
    float4 b = base[a];
    out[i] = make_float4(gx + b.x, gy + b.y, gx + b.z, gy + b.w);
}

extern "C" int frcnn_anchor_grid_f32(const float *base, int na, int fh, int fw, float stride_y,
                                     float stride_x, float *out, void *stream)
{
    FRCNN_REQUIRE(base && out && na > 0 && fh > 0 && fw > 0, "anchor_grid: bad arguments");
    long long total = (long long)fh * fw * na;
    FRCNN_REQUIRE(total < (1ll << 31), "anchor_grid: too many anchors");
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(anchor_grid_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)base, na, fw, (int)total, stride_y, stride_x, (float4 *)out);
    FRCNN_CHECK_LAUNCH("anchor_grid");
    return FRCNN_OK;
}

// ---- K3 elementwise pieces --------------------------------------------------------------------
__device__ __forceinline__ float4 decode_one(float4 box, float4 d, float clampv)
{
    // utils.py:186-204.  Quirk Q1: the centre is (w/2, h/2); the box origin is not added.
    float bh = box.w - box.y;
    float bw = box.z - box.x;
    float cx = bw / 2.0f;
    float cy = bh / 2.0f;
    float dw = d.z > clampv ? clampv : d.z;
    float dh = d.w > clampv ? clampv : d.w;
    float px = cx + d.x * bw;
    float py = cy + d.y * bh;
    float ph = expf(dh) * bh;
    float pw = expf(dw) * bw;
    return make_float4(px - pw / 2.0f, py - ph / 2.0f, px + pw / 2.0f, py + ph / 2.0f);
}

__device__ __forceinline__ float clip1(float v, float hi)
{
    v = v < 0.0f ? 0.0f : v;
    v = v > hi ? hi : v;
    return v;
}

__device__ __forceinline__ float4 clip_box(float4 b, float image_h, float image_w)
{
    return make_float4(clip1(b.x, image_w), clip1(b.y, image_h), clip1(b.z, image_w),
                       clip1(b.w, image_h));
}

#define FRCNN_DW_CLAMP 4.135166556742356 /* math.log(1000. / 16), utils.py:193 */

__global__ void __launch_bounds__(256) box_regression_kernel(const float4 *__restrict__ box,
                                                             const float4 *__restrict__ regs,
                                                             int n, int K,
                                                             float4 *__restrict__ out)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * K) return;
    int row = (int)(i / K);
    out[i] = decode_one(box[row], regs[i], (float)FRCNN_DW_CLAMP);
}

extern "C" int frcnn_box_regression_f32(const float *box, const float *regs, int n, int K,
                                        float *out, void *stream)
{
    FRCNN_REQUIRE(n >= 0 && K > 0, "box_regression: bad shape");
    if (n == 0) return FRCNN_OK;
    FRCNN_REQUIRE(box && regs && out, "box_regression: null pointer");
    long long total = (long long)n * K;
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(box_regression_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)box, (const float4 *)regs, n, K, (float4 *)out);
    FRCNN_CHECK_LAUNCH("box_regression");
    return FRCNN_OK;
}

// Backward of decode (utils.py:173-208 under torch autograd; the reference never detaches its
// proposals, rpn.py:75,85).  One thread per box row, looping over its K delta groups so that the
// box gradient (a sum over K) needs no atomics.  torch.clamp(max=) passes the gradient where
// dw <= max (equality included).
__global__ void __launch_bounds__(256) box_regression_backward_kernel(
    const float4 *__restrict__ box, const float4 *__restrict__ regs, const float4 *__restrict__ grad_out, int n,
    int K, float4 *__restrict__ grad_regs, float4 *__restrict__ grad_box)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float clampv = (float)FRCNN_DW_CLAMP;
    const float4 b = box[i];
    const float bw = b.z - b.x, bh = b.w - b.y;
    float gbw = 0.0f, gbh = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float4 r = regs[(long long)i * K + k];
        const float4 g = grad_out[(long long)i * K + k];
        const float dwc = r.z > clampv ? clampv : r.z;
        const float dhc = r.w > clampv ? clampv : r.w;
        const float ew = expf(dwc), eh = expf(dhc);
        const float gpx = g.x + g.z, gpy = g.y + g.w;               // through box_x / box_y
        const float gpw = g.z / 2.0f - g.x / 2.0f, gph = g.w / 2.0f - g.y / 2.0f; // through box_w/2, box_h/2
        float4 gr;
        gr.x = gpx * bw;
        gr.y = gpy * bh;
        gr.z = r.z <= clampv ? (gpw * bw) * ew : 0.0f;
        gr.w = r.w <= clampv ? (gph * bh) * eh : 0.0f;
        grad_regs[(long long)i * K + k] = gr;
        gbw += gpx / 2.0f + gpx * r.x + gpw * ew;
        gbh += gpy / 2.0f + gpy * r.y + gph * eh;
    }
    if (grad_box) grad_box[i] = make_float4(-gbw, -gbh, gbw, gbh);
}

extern "C" int frcnn_box_regression_backward_f32(const float *box, const float *regs, const float *grad_out,
                                                 int n, int K, float *grad_regs, float *grad_box, void *stream)
{
    FRCNN_REQUIRE(n >= 0 && K > 0, "box_regression_backward: bad shape");
    if (n == 0) return FRCNN_OK;
    FRCNN_REQUIRE(box && regs && grad_out && grad_regs, "box_regression_backward: null pointer");
    hipLaunchKernelGGL(box_regression_backward_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)box, (const float4 *)regs, (const float4 *)grad_out, n, K,
                       (float4 *)grad_regs, (float4 *)grad_box);
    FRCNN_CHECK_LAUNCH("box_regression_backward");
    return FRCNN_OK;
}

__global__ void __launch_bounds__(256) box_clip_kernel(const float4 *__restrict__ box, int n,
                                                       float image_h, float image_w,
                                                       float4 *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = clip_box(box[i], image_h, image_w);
}

extern "C" int frcnn_box_clip_f32(const float *box, int n, float image_h, float image_w, float *out,
                                  void *stream)
{
    FRCNN_REQUIRE(n >= 0, "box_clip: bad shape");
    if (n == 0) return FRCNN_OK;
    FRCNN_REQUIRE(box && out, "box_clip: null pointer");
    hipLaunchKernelGGL(box_clip_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)box, n, image_h, image_w, (float4 *)out);
    FRCNN_CHECK_LAUNCH("box_clip");
    return FRCNN_OK;
}

// Order-preserving compaction by one 1024-thread block that sweeps the rows in tiles.
// MODE 0: keep boxes with w > t && h > t (writes box_out + index); MODE 1: keep value > t.
template <int MODE>
__global__ void __launch_bounds__(1024) compact_kernel(const float4 *__restrict__ box,
                                                       const float *__restrict__ value, int n,
                                                       float thresh, float4 *__restrict__ box_out,
                                                       int64_t *__restrict__ index_out,
                                                       int32_t *__restrict__ count_out)
{
    __shared__ int wave_sums[16];
    int base = 0;
    for (int start = 0; start < n; start += 1024) {
        int i = start + threadIdx.x;
        bool keep = false;
        float4 b = make_float4(0, 0, 0, 0);
        if (i < n) {
            if (MODE == 0) {
                b = box[i];
                float w = b.z - b.x;
                float h = b.w - b.y;
                keep = (w > thresh) && (h > thresh);
            } else {
                keep = value[i] > thresh;
            }
        }
        int tot;
        int pos = base + frcnn_block_excl_count(keep, wave_sums, &tot);
        if (keep) {
            if (MODE == 0) box_out[pos] = b;
            index_out[pos] = i;
        }
        base += tot;
    }
    if (threadIdx.x == 0) *count_out = base;
}

extern "C" int frcnn_remove_small_box_f32(const float *box, int n, float size_thresh,
                                          float *box_out, int64_t *index_out, int32_t *count_out,
                                          void *stream)
{
    FRCNN_REQUIRE(n >= 0 && count_out, "remove_small_box: bad arguments");
    FRCNN_REQUIRE(n == 0 || (box && box_out && index_out), "remove_small_box: null pointer");
    hipLaunchKernelGGL(compact_kernel<0>, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                       (const float4 *)box, (const float *)nullptr, n, size_thresh,
                       (float4 *)box_out, index_out, count_out);
    FRCNN_CHECK_LAUNCH("remove_small_box");
    return FRCNN_OK;
}

extern "C" int frcnn_select_gt_f32(const float *value, int n, float thresh, int64_t *index_out,
                                   int32_t *count_out, void *stream)
{
    FRCNN_REQUIRE(n >= 0 && count_out, "select_gt: bad arguments");
    FRCNN_REQUIRE(n == 0 || (value && index_out), "select_gt: null pointer");
    hipLaunchKernelGGL(compact_kernel<1>, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                       (const float4 *)nullptr, value, n, thresh, (float4 *)nullptr, index_out,
                       count_out);
    FRCNN_CHECK_LAUNCH("select_gt");
    return FRCNN_OK;
}

// Fused gather + decode + clip + small-box compaction.  Workgroup = (tile of 1024 candidates, image).
// The compaction is order-preserving over the whole image, so a tile needs the number of rows kept in
// the tiles before it: it simply re-evaluates their keep flags (a decode is ~40 instructions; the loads
// of all earlier tiles are independent and overlap) and reduces once, instead of the image's tiles
// being walked one after the other by a single workgroup (12 dependent gather round trips and 24
// barriers at k = 12,000).
__device__ __forceinline__ bool decode_row(const float4 *__restrict__ anc, const float4 *__restrict__ del,
                                           const int32_t *__restrict__ idx, int i, float image_h,
                                           float image_w, float min_size, float4 *p_out)
{
    const int a = idx[i];
    float4 p = decode_one(anc[a], del[a], (float)FRCNN_DW_CLAMP);
    p = clip_box(p, image_h, image_w);
    *p_out = p;
    const float w = p.z - p.x;
    const float h = p.w - p.y;
    return (w > min_size) && (h > min_size);
}

// Two launches.  decode_clip_filter_kernel: workgroup = (tile of 1024 ranks, image); it decodes its rows
// ONCE, compacts the kept ones to the front of the tile's own 1024 output slots and marks the unused slots
// with src = -1 -- i.e. it writes the final list as it is when every earlier tile keeps all its rows (the
// RPN's min_size of 1e-3 removes next to nothing).  decode_close_gaps_kernel (one workgroup per image)
// counts each tile's kept rows from those marks, moves the tiles down over any gaps in ascending order and
// writes the image's count; with no gap it only counts.  (The single-launch form had every tile re-derive
// the keep count of the tiles before it: 6.5x the anchor / delta gathers, 30 us against 8 + 4.)
__global__ void __launch_bounds__(1024)
decode_clip_filter_kernel(const float4 *__restrict__ anchors, long long anchor_batch_stride4,
                          const float4 *__restrict__ deltas, const int32_t *__restrict__ top_idx,
                          const float *__restrict__ top_score, int A, int k, float image_h,
                          float image_w, float min_size, float4 *__restrict__ boxes_out,
                          float *__restrict__ scores_out, int32_t *__restrict__ src_out,
                          int32_t *__restrict__ count_out)
{
    __shared__ int wave_sums[16];
    const int b = blockIdx.y;
    const int tile = blockIdx.x;
    const float4 *anc = anchors + (long long)b * anchor_batch_stride4;
    const float4 *del = deltas + (long long)b * A;
    const int32_t *idx = top_idx + (long long)b * k;
    const float *sc = top_score + (long long)b * k;
    float4 *bo = boxes_out + (long long)b * k + tile * 1024;
    float *so = scores_out + (long long)b * k + tile * 1024;
    int32_t *ro = src_out + (long long)b * k + tile * 1024;
    const int i = tile * 1024 + threadIdx.x;
    bool keep = false;
    float4 p = make_float4(0, 0, 0, 0);
    float s = 0.0f;
    if (i < k) {
        s = sc[i];
        keep = decode_row(anc, del, idx, i, image_h, image_w, min_size, &p);
    }
    int tot;
    const int pos = frcnn_block_excl_count(keep, wave_sums, &tot);
    if (keep) {
        bo[pos] = p;
        so[pos] = s;
        ro[pos] = i;
    }
    if ((int)threadIdx.x >= tot && i < k) ro[threadIdx.x] = -1; // unused slot of this tile
    if (gridDim.x == 1 && threadIdx.x == 0) count_out[b] = tot;
}

// the same with the deltas read from the head's NCHW outputs (common.h: RpnLevels)
__global__ void __launch_bounds__(1024)
decode_clip_filter_nchw_kernel(const float4 *__restrict__ anchors, long long anchor_batch_stride4, RpnLevels L,
                               const int32_t *__restrict__ top_idx, const float *__restrict__ top_score, int k,
                               float image_h, float image_w, float min_size, float4 *__restrict__ boxes_out,
                               float *__restrict__ scores_out, int32_t *__restrict__ src_out,
                               int32_t *__restrict__ count_out)
{
    __shared__ int wave_sums[16];
    const int b = blockIdx.y;
    const int tile = blockIdx.x;
    const float4 *anc = anchors + (long long)b * anchor_batch_stride4;
    float4 *bo = boxes_out + (long long)b * k + tile * 1024;
    float *so = scores_out + (long long)b * k + tile * 1024;
    int32_t *ro = src_out + (long long)b * k + tile * 1024;
    const int i = tile * 1024 + threadIdx.x;
    bool keep = false;
    float4 p = make_float4(0, 0, 0, 0);
    float s = 0.0f;
    if (i < k) {
        s = top_score[(long long)b * k + i];
        const int a = top_idx[(long long)b * k + i];
        p = clip_box(decode_one(anc[a], rpn_delta(L, b, a), (float)FRCNN_DW_CLAMP), image_h, image_w);
        keep = (p.z - p.x > min_size) && (p.w - p.y > min_size);
    }
    int tot;
    const int pos = frcnn_block_excl_count(keep, wave_sums, &tot);
    if (keep) {
        bo[pos] = p;
        so[pos] = s;
        ro[pos] = i;
    }
    if ((int)threadIdx.x >= tot && i < k) ro[threadIdx.x] = -1;
    if (gridDim.x == 1 && threadIdx.x == 0) count_out[b] = tot;
}

__global__ void __launch_bounds__(1024)
decode_close_gaps_kernel(int k, float4 *__restrict__ boxes_out, float *__restrict__ scores_out,
                         int32_t *__restrict__ src_out, int32_t *__restrict__ count_out)
{
    __shared__ int s_cnt[64];
    const int b = blockIdx.x;
    const int tiles = (k + 1023) / 1024;
    float4 *bo = boxes_out + (long long)b * k;
    float *so = scores_out + (long long)b * k;
    int32_t *ro = src_out + (long long)b * k;
    const int t = threadIdx.x;
    int base = 0; // kept rows of the tiles already passed
    bool gap = false;
    for (int t0 = 0; t0 < tiles; t0 += 64) { // counts of up to 64 tiles at a time
        const int nt = min(64, tiles - t0);
        if (t < 64) s_cnt[t] = 0;
        __syncthreads();
        for (int u0 = 0; u0 < nt; u0 += 8) { // eight tiles' marks in flight at a time
            int r[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int i = (t0 + u0 + v) * 1024 + t;
                r[v] = (u0 + v < nt && i < k) ? ro[i] : -1;
            }
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const unsigned long long m = __ballot(r[v] >= 0);
                if ((t & 63) == 0 && m) atomicAdd(&s_cnt[u0 + v], __popcll(m));
            }
        }
        __syncthreads();
        for (int u = 0; u < nt; ++u) {
            const int c = s_cnt[u];
            const int from = (t0 + u) * 1024;
            if (gap || base != from) { // (uniform) move this tile's c rows down to `base`
                float4 bx = make_float4(0, 0, 0, 0);
                float sv = 0.0f;
                int rv = 0;
                if (t < c) {
                    bx = bo[from + t];
                    sv = so[from + t];
                    rv = ro[from + t];
                }
                __syncthreads();
                if (t < c) {
                    bo[base + t] = bx;
                    so[base + t] = sv;
                    ro[base + t] = rv;
                }
                __syncthreads();
                gap = true;
            }
            base += c;
        }
        __syncthreads();
    }
    if (t == 0) count_out[b] = base;
}

extern "C" int frcnn_decode_clip_filter_f32(const float *anchors, int64_t anchor_batch_stride,
                                            const float *deltas, const int32_t *top_idx,
                                            const float *top_score, int B, int A, int k,
                                            float image_h, float image_w, float min_size,
                                            float *boxes_out, float *scores_out, int32_t *src_out,
                                            int32_t *count_out, void *stream)
{
    FRCNN_REQUIRE(B > 0 && A > 0 && k > 0 && k <= A, "decode_clip_filter: bad shape");
    FRCNN_REQUIRE(anchors && deltas && top_idx && top_score && boxes_out && scores_out && src_out &&
                      count_out,
                  "decode_clip_filter: null pointer");
    FRCNN_REQUIRE(anchor_batch_stride % 4 == 0, "decode_clip_filter: anchor stride not a box multiple");
    FRCNN_REQUIRE(B <= 65535, "decode_clip_filter: batch too large");
    const int tiles = (k + 1023) / 1024;
    hipLaunchKernelGGL(decode_clip_filter_kernel, dim3(tiles, B), dim3(1024), 0, (hipStream_t)stream,
                       (const float4 *)anchors, (long long)(anchor_batch_stride / 4),
                       (const float4 *)deltas, top_idx, top_score, A, k, image_h, image_w, min_size,
                       (float4 *)boxes_out, scores_out, src_out, count_out);
    FRCNN_CHECK_LAUNCH("decode_clip_filter");
    if (tiles > 1) {
        hipLaunchKernelGGL(decode_close_gaps_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, k,
                           (float4 *)boxes_out, scores_out, src_out, count_out);
        FRCNN_CHECK_LAUNCH("decode_close_gaps");
    }
    return FRCNN_OK;
}

extern "C" int frcnn_decode_clip_filter_nchw_f32(const float *anchors, int64_t anchor_batch_stride,
                                                 const float *const *reg_host, const int32_t *hw_host, int n_levels,
                                                 int na, const int32_t *top_idx, const float *top_score, int B, int k,
                                                 float image_h, float image_w, float min_size, float *boxes_out,
                                                 float *scores_out, int32_t *src_out, int32_t *count_out,
                                                 void *stream)
{
    FRCNN_REQUIRE(B > 0 && B <= 65535 && k > 0 && reg_host, "decode_clip_filter_nchw: bad shape");
    RpnLevels L;
    const long long A = frcnn_make_levels(&L, nullptr, reg_host, hw_host, n_levels, na);
    FRCNN_REQUIRE(A > 0 && k <= A, "decode_clip_filter_nchw: bad level table or k > A");
    for (int l = 0; l < n_levels; ++l) FRCNN_REQUIRE(reg_host[l], "decode_clip_filter_nchw: null level %d", l);
    FRCNN_REQUIRE(anchors && top_idx && top_score && boxes_out && scores_out && src_out && count_out,
                  "decode_clip_filter_nchw: null pointer");
    FRCNN_REQUIRE(anchor_batch_stride % 4 == 0, "decode_clip_filter_nchw: anchor stride not a box multiple");
    const int tiles = (k + 1023) / 1024;
    hipLaunchKernelGGL(decode_clip_filter_nchw_kernel, dim3(tiles, B), dim3(1024), 0, (hipStream_t)stream,
                       (const float4 *)anchors, (long long)(anchor_batch_stride / 4), L, top_idx, top_score, k, image_h,
                       image_w, min_size, (float4 *)boxes_out, scores_out, src_out, count_out);
    FRCNN_CHECK_LAUNCH("decode_clip_filter_nchw");
    if (tiles > 1) {
        hipLaunchKernelGGL(decode_close_gaps_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, k,
                           (float4 *)boxes_out, scores_out, src_out, count_out);
        FRCNN_CHECK_LAUNCH("decode_close_gaps");
    }
    return FRCNN_OK;
}

// ---- ROIHead eval post-process, batched (ROIHead/roi_head.py:178-209) ---------------------------
// For every image at once: per (proposal j, class c) candidate, in the reference's flat order j*K + c:
// decode the class's deltas against the proposal (utils.box_regression), clip to the image's own
// size, and keep it when score > score_thresh (:197), w > min_size and h > min_size (:203) and c > 0
// (:209).  The three filters of the reference are order-preserving and each looks at one candidate
// only, so applying them as one predicate gives the same list.  Workgroup = (tile of 1024
// candidates, image).  Fallback kernel (K > 1024): a tile re-derives the keep count of the tiles before it.
__device__ __forceinline__ bool detect_keep(const float4 *__restrict__ props, const float4 *__restrict__ regs,
                                            const float *__restrict__ scores, long long row0, int m, int K,
                                            float image_h, float image_w, float score_thresh, float min_size,
                                            float4 *box_out, float *score_out)
{
    const int j = m / K, c = m - j * K;
    const long long e = (row0 + j) * K + c;
    const float sc = scores[e];
    *score_out = sc;
    if (!(c > 0 && sc > score_thresh)) return false;
    float4 p = clip_box(decode_one(props[row0 + j], regs[e], (float)FRCNN_DW_CLAMP), image_h, image_w);
    *box_out = p;
    return (p.z - p.x > min_size) && (p.w - p.y > min_size);
}

__global__ void __launch_bounds__(1024)
roi_detect_filter_rederive_kernel(const float4 *__restrict__ props, const float4 *__restrict__ regs,
                         const float *__restrict__ scores, const int32_t *__restrict__ row_off,
                         const float *__restrict__ image_sizes, int K, float score_thresh, float min_size,
                         int cap, float4 *__restrict__ boxes_out, float *__restrict__ scores_out,
                         int32_t *__restrict__ labels_out, int32_t *__restrict__ count_out)
{
    __shared__ int wave_sums[16];
    __shared__ int s_prior[16];
    const int b = blockIdx.y;
    const int tile = blockIdx.x;
    const long long row0 = row_off[b];
    const int M = (row_off[b + 1] - row_off[b]) * K; // candidates of this image
    if (tile * 1024 >= M && tile != 0) return;       // (tile 0 still reports a zero count)
    const float image_h = image_sizes[2 * b], image_w = image_sizes[2 * b + 1];
    int prior = 0;
    for (int u = 0; u < tile; ++u) {
        float4 q;
        float sc;
        prior += detect_keep(props, regs, scores, row0, u * 1024 + (int)threadIdx.x, K, image_h, image_w,
                             score_thresh, min_size, &q, &sc) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) prior += __shfl_down(prior, o);
    if ((threadIdx.x & 63) == 0) s_prior[threadIdx.x >> 6] = prior;
    const int m = tile * 1024 + threadIdx.x;
    float4 p = make_float4(0, 0, 0, 0);
    float sc = 0.0f;
    const bool keep = m < M && detect_keep(props, regs, scores, row0, m, K, image_h, image_w, score_thresh,
                                           min_size, &p, &sc);
    int tot;
    int pos = frcnn_block_excl_count(keep, wave_sums, &tot); // (its barriers also publish s_prior)
    int base = 0;
    for (int w = 0; w < 16; ++w) base += s_prior[w];
    pos += base;
    if (keep) {
        boxes_out[(long long)b * cap + pos] = p;
        scores_out[(long long)b * cap + pos] = sc;
        labels_out[(long long)b * cap + pos] = m % K;
    }
    if (threadIdx.x == 0 && (tile + 1) * 1024 >= M) count_out[b] = base + tot; // the image's last tile
}

// Two launches for K <= 1024 classes.  Class 0 is never kept, so an image with r rows keeps at most r (K - 1)
// of its M = r K candidates: slots [M - r, M) of its labels_out row are never an output row, and an image has
// at most r tiles of 1024 candidates when K <= 1024.  The first launch leaves every tile's keep count in slot
// M - 1 - tile; the second adds up the counts of the tiles before its own instead of re-deriving them.
__global__ void __launch_bounds__(1024)
roi_detect_count_kernel(const float4 *__restrict__ props, const float4 *__restrict__ regs,
                        const float *__restrict__ scores, const int32_t *__restrict__ row_off,
                        const float *__restrict__ image_sizes, int K, float score_thresh, float min_size, int cap,
                        int32_t *__restrict__ labels_out)
{
    __shared__ int wave_sums[16];
    const int b = blockIdx.y;
    const int tile = blockIdx.x;
    const long long row0 = row_off[b];
    const int M = (row_off[b + 1] - row_off[b]) * K;
    if (tile * 1024 >= M) return;
    const float image_h = image_sizes[2 * b], image_w = image_sizes[2 * b + 1];
    const int m = tile * 1024 + threadIdx.x;
    float4 p;
    float sc;
    const bool keep = m < M && detect_keep(props, regs, scores, row0, m, K, image_h, image_w, score_thresh,
                                           min_size, &p, &sc);
    int tot;
    frcnn_block_excl_count(keep, wave_sums, &tot);
    if (threadIdx.x == 0) labels_out[(long long)b * cap + M - 1 - tile] = tot;
}

__global__ void __launch_bounds__(1024)
roi_detect_filter_kernel(const float4 *__restrict__ props, const float4 *__restrict__ regs,
                         const float *__restrict__ scores, const int32_t *__restrict__ row_off,
                         const float *__restrict__ image_sizes, int K, float score_thresh, float min_size,
                         int cap, float4 *__restrict__ boxes_out, float *__restrict__ scores_out,
                         int32_t *__restrict__ labels_out, int32_t *__restrict__ count_out)
{
    __shared__ int wave_sums[16];
    __shared__ int s_prior[16];
    const int b = blockIdx.y;
    const int tile = blockIdx.x;
    const long long row0 = row_off[b];
    const int M = (row_off[b + 1] - row_off[b]) * K;
    if (tile * 1024 >= M && tile != 0) return; // (tile 0 still reports a zero count)
    const float image_h = image_sizes[2 * b], image_w = image_sizes[2 * b + 1];
    int prior = 0;
    for (int u = threadIdx.x; u < tile; u += 1024) prior += labels_out[(long long)b * cap + M - 1 - u];
    for (int o = 32; o > 0; o >>= 1) prior += __shfl_down(prior, o);
    if ((threadIdx.x & 63) == 0) s_prior[threadIdx.x >> 6] = prior;
    const int m = tile * 1024 + threadIdx.x;
    float4 p = make_float4(0, 0, 0, 0);
    float sc = 0.0f;
    const bool keep = m < M && detect_keep(props, regs, scores, row0, m, K, image_h, image_w, score_thresh,
                                           min_size, &p, &sc);
    int tot;
    int pos = frcnn_block_excl_count(keep, wave_sums, &tot); // (its barriers also publish s_prior)
    int base = 0;
    for (int w = 0; w < 16; ++w) base += s_prior[w];
    pos += base;
    if (keep) { // pos < r (K - 1) <= M - r: never one of the count slots
        boxes_out[(long long)b * cap + pos] = p;
        scores_out[(long long)b * cap + pos] = sc;
        labels_out[(long long)b * cap + pos] = m % K;
    }
    if (threadIdx.x == 0 && (tile + 1) * 1024 >= M) count_out[b] = base + tot; // the image's last tile
}

extern "C" int frcnn_roi_detect_filter_f32(const float *props, const float *regs, const float *scores,
                                           const int32_t *row_off, const float *image_sizes, int B, int K,
                                           int max_rows, float score_thresh, float min_size, float *boxes_out,
                                           float *scores_out, int32_t *labels_out, int32_t *count_out,
                                           void *stream)
{
    FRCNN_REQUIRE(B > 0 && B <= 65535 && K > 0 && max_rows >= 0, "roi_detect_filter: bad shape");
    FRCNN_REQUIRE((long long)max_rows * K < (1ll << 30), "roi_detect_filter: too many candidates per image");
    FRCNN_REQUIRE(row_off && image_sizes && count_out, "roi_detect_filter: null pointer");
    const int cap = max_rows * K;
    if (cap == 0) {
        hipError_t e = hipMemsetAsync(count_out, 0, sizeof(int32_t) * (size_t)B, (hipStream_t)stream);
        FRCNN_REQUIRE(e == hipSuccess, "roi_detect_filter: memset failed: %s", hipGetErrorString(e));
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(props && regs && scores && boxes_out && scores_out && labels_out, "roi_detect_filter: null pointer");
    const dim3 grid((cap + 1023) / 1024, B);
    if (K > 1024 || grid.x == 1) { // (a single tile has nothing before it)
        hipLaunchKernelGGL(roi_detect_filter_rederive_kernel, grid, dim3(1024), 0, (hipStream_t)stream,
                           (const float4 *)props, (const float4 *)regs, scores, row_off, image_sizes, K, score_thresh,
                           min_size, cap, (float4 *)boxes_out, scores_out, labels_out, count_out);
        FRCNN_CHECK_LAUNCH("roi_detect_filter");
        return FRCNN_OK;
    }
    hipLaunchKernelGGL(roi_detect_count_kernel, grid, dim3(1024), 0, (hipStream_t)stream, (const float4 *)props,
                       (const float4 *)regs, scores, row_off, image_sizes, K, score_thresh, min_size, cap, labels_out);
    FRCNN_CHECK_LAUNCH("roi_detect_count");
    hipLaunchKernelGGL(roi_detect_filter_kernel, grid, dim3(1024), 0, (hipStream_t)stream,
                       (const float4 *)props, (const float4 *)regs, scores, row_off, image_sizes, K, score_thresh,
                       min_size, cap, (float4 *)boxes_out, scores_out, labels_out, count_out);
    FRCNN_CHECK_LAUNCH("roi_detect_filter");
    return FRCNN_OK;
}

// ---- gather -----------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gather_boxes_kernel(const float4 *__restrict__ src,
                                                           const int32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ count, int n,
                                                           int m, float4 *__restrict__ out)
{
    int b = blockIdx.y;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    int c = count ? count[b] : m;
    float4 v = make_float4(0, 0, 0, 0);
    if (j < c) {
        int s = idx[(long long)b * m + j];
        if (s >= 0 && s < n) v = src[(long long)b * n + s];
    }
    out[(long long)b * m + j] = v;
}

extern "C" int frcnn_gather_boxes_f32(const float *src, const int32_t *idx, const int32_t *count,
                                      int B, int n, int m, float *out, void *stream)
{
    FRCNN_REQUIRE(B >= 0 && n >= 0 && m >= 0, "gather_boxes: bad shape");
    if (B == 0 || m == 0) return FRCNN_OK;
    FRCNN_REQUIRE(idx && out && (n == 0 || src), "gather_boxes: null pointer");
    hipLaunchKernelGGL(gather_boxes_kernel, dim3((m + 255) / 256, B), dim3(256), 0,
                       (hipStream_t)stream, (const float4 *)src, idx, count, n, m, (float4 *)out);
    FRCNN_CHECK_LAUNCH("gather_boxes");
    return FRCNN_OK;
}

// ---- K5 IoU matrix + fused label/encode -------------------------------------------------------
#define IOU_TILE_M 256

// (n, m) materialised IoU: a block owns 256 rows (one per lane) and sweeps box_2 in LDS tiles of
// 256; each lane writes its row segment.  Only used for API parity with utils.box_iou.
__global__ void __launch_bounds__(256) box_iou_kernel(const float4 *__restrict__ box1, int n,
                                                      const float4 *__restrict__ box2, int m,
                                                      float *__restrict__ out)
{
    __shared__ float4 s_box[IOU_TILE_M];
    __shared__ float s_area[IOU_TILE_M];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float4 a = make_float4(0, 0, 0, 0);
    if (i < n) a = box1[i];
    float aarea = frcnn_area(a.x, a.y, a.z, a.w);
    for (int j0 = 0; j0 < m; j0 += IOU_TILE_M) {
        int jn = min(IOU_TILE_M, m - j0);
        __syncthreads();
        if ((int)threadIdx.x < jn) {
            float4 g = box2[j0 + threadIdx.x];
            s_box[threadIdx.x] = g;
            s_area[threadIdx.x] = frcnn_area(g.x, g.y, g.z, g.w);
        }
        __syncthreads();
        if (i < n) {
            float *o = out + (long long)i * m + j0;
            for (int j = 0; j < jn; ++j) {
                float4 g = s_box[j];
                o[j] = frcnn_iou(a.x, a.y, a.z, a.w, aarea, g.x, g.y, g.z, g.w, s_area[j]);
            }
        }
    }
}

extern "C" int frcnn_box_iou_f32(const float *box1, int n, const float *box2, int m, float *out,
                                 void *stream)
{
    FRCNN_REQUIRE(n >= 0 && m >= 0, "box_iou: bad shape");
    if (n == 0 || m == 0) return FRCNN_OK;
    FRCNN_REQUIRE(box1 && box2 && out, "box_iou: null pointer");
    hipLaunchKernelGGL(box_iou_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)box1, n, (const float4 *)box2, m, out);
    FRCNN_CHECK_LAUNCH("box_iou");
    return FRCNN_OK;
}

struct Weights4 {
    float x, y, w, h;
};

// One lane per box_1 row; the image's ground-truth boxes (+ areas) are staged in LDS tiles and
// swept by every lane (wave-uniform LDS address => broadcast reads).  Running max follows
// torch.max(dim=1): replace when !(v <= best); the first NaN wins and freezes the row.
//
// Round 3: no division per pair.  While every box of the wave and of the tile is "regular" (area in
// (1e-6, 1e12): finite, no 0/0, no product that under- or overflows) the running max is kept as the PAIR
// (inter, union) of the best box, and "is fl(i/u) > fl(bi/bu)" is decided by cross-multiplication,
// d = i*bu - bi*u: when |d| > 2^-20 (i*bu + bi*u) the exact quotients differ by more than 2^-21 relative -- four
// units in the last place -- so their roundings are ordered like d's sign; when both products are zero both
// quotients are +0 (no replacement); anything else (ties between equal boxes, one pair in ~2^20 otherwise) is
// redone with the two IEEE divisions, which IS the definition.  The decision is taken for FOUR boxes at a
// time (one wave-uniform branch per group).  One division per row at the end gives the maximum itself.
// Irregular boxes (degenerate, NaN, huge) take the loop of rounds 1-2, division per pair, unchanged.
#define IOU_REG_LO 1e-6f
#define IOU_REG_HI 1e12f

struct IouBest { // running max of one row in the regular loop
    float bi, bu; // intersection and union of the best box so far
    int arg;
};

// The operations of frcnn_iou, in its order (utils.py:114-119), for REGULAR boxes: v_max_f32 / v_min_f32 return what
// the reference's compare-and-select does whenever neither operand is NaN and the two are not zeros of opposite sign
// (regular boxes hold no NaN and no -0: iou_regular), and a clamped difference of such operands is never -0.
__device__ __forceinline__ void iou_terms(const float4 &a, float aarea, const float4 &g, float garea, float &inter,
                                          float &uni)
{
    const float ltx = frcnn_vmax(a.x, g.x), lty = frcnn_vmax(a.y, g.y);
    const float rbx = frcnn_vmin(a.z, g.z), rby = frcnn_vmin(a.w, g.w);
    const float jw = frcnn_vmax(rbx - ltx, 0.0f), jh = frcnn_vmax(rby - lty, 0.0f);
    inter = jw * jh;
    uni = (aarea + garea) - inter;
}

__device__ __forceinline__ bool iou_regular(const float4 &b, float area)
{
    const bool neg_zero = __float_as_uint(b.x) == 0x80000000u || __float_as_uint(b.y) == 0x80000000u ||
                          __float_as_uint(b.z) == 0x80000000u || __float_as_uint(b.w) == 0x80000000u;
    return area > IOU_REG_LO && area < IOU_REG_HI && !neg_zero;
}

__device__ __forceinline__ void iou_best_exact(IouBest &s, float inter, float uni, int j)
{
    const bool take = (inter / uni) > (s.bi / s.bu); // regular boxes: no NaN, so !(v <= best) is v > best
    s.bi = take ? inter : s.bi;
    s.bu = take ? uni : s.bu;
    s.arg = take ? j : s.arg;
}

__global__ void __launch_bounds__(256)
iou_label_encode_kernel(const float4 *__restrict__ box1, long long box1_batch_stride4,
                        const float4 *__restrict__ box2, const int32_t *__restrict__ m_per_image,
                        const float *__restrict__ labels, int N, int m_max, float neg_thr,
                        float pos_thr, Weights4 wt, float *__restrict__ label_out,
                        int32_t *__restrict__ argmax_out, float *__restrict__ max_iou_out,
                        float4 *__restrict__ dev_out)
{
    __shared__ float4 s_box[IOU_TILE_M];
    __shared__ float s_area[IOU_TILE_M];
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = m_per_image ? m_per_image[b] : m_max;
    const float4 *gt = box2 + (long long)b * m_max;
    float4 a = make_float4(0.0f, 0.0f, 1.0f, 1.0f); // lanes past the end: a regular box, never written back
    if (i < N) a = box1[(long long)b * box1_batch_stride4 + i];
    float aarea = frcnn_area(a.x, a.y, a.z, a.w);
    float best = 0.0f;
    int arg = 0;
    bool frozen = false;
    IouBest rb = {0.0f, 1.0f, 0};
    // wave-uniform: does this wave still run the division-free loop?  (Once a tile sends it to the generic loop it
    // stays there: the generic state is the value, not the pair.)
    bool regular = __all(iou_regular(a, aarea));
    for (int j0 = 0; j0 < m; j0 += IOU_TILE_M) {
        int jn = min(IOU_TILE_M, m - j0);
        __syncthreads();
        bool g_reg = true;
        if ((int)threadIdx.x < jn) {
            float4 g = gt[j0 + threadIdx.x];
            const float ga = frcnn_area(g.x, g.y, g.z, g.w);
            s_box[threadIdx.x] = g;
            s_area[threadIdx.x] = ga;
            g_reg = iou_regular(g, ga);
        }
        const bool tile_reg = __syncthreads_and(g_reg) != 0;
        if (regular && !tile_reg) { // leave the pair form for good: the value of the best box so far
            regular = false;
            if (j0 > 0) {
                best = rb.bi / rb.bu;
                arg = rb.arg;
            }
        }
        if (regular) {
            int j = 0;
            if (j0 == 0) { // the first box is taken whatever its value
                iou_terms(a, aarea, s_box[0], s_area[0], rb.bi, rb.bu);
                rb.arg = 0;
                j = 1;
            }
            for (; j + 4 <= jn; j += 4) {
                const IouBest saved = rb;
                float in4[4], un4[4];
                unsigned long long unsure = 0; // lanes whose decision for one of the four boxes is inside the band
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    iou_terms(a, aarea, s_box[j + u], s_area[j + u], in4[u], un4[u]);
                    const float p1 = in4[u] * rb.bu, p2 = rb.bi * un4[u];
                    const float d = p1 - p2, sum = p1 + p2;
                    unsure |= __ballot(fabsf(d) <= sum * 9.5367431640625e-07f) & __ballot(sum > 0.0f);
                    const bool take = d > 0.0f;
                    rb.bi = take ? in4[u] : rb.bi;
                    rb.bu = take ? un4[u] : rb.bu;
                    rb.arg = take ? j0 + j + u : rb.arg;
                }
                if (__builtin_expect(unsure != 0, 0)) { // wave-uniform: the group again, by the definition
                    rb = saved;
#pragma unroll
                    for (int u = 0; u < 4; ++u) iou_best_exact(rb, in4[u], un4[u], j0 + j + u);
                }
            }
            for (; j < jn; ++j) {
                float inter, uni;
                iou_terms(a, aarea, s_box[j], s_area[j], inter, uni);
                iou_best_exact(rb, inter, uni, j0 + j);
            }
            continue;
        }
        for (int j = 0; j < jn; ++j) {
            float4 g = s_box[j];
            float v = frcnn_iou(a.x, a.y, a.z, a.w, aarea, g.x, g.y, g.z, g.w, s_area[j]);
            bool first = (j0 + j) == 0;
            bool take = !frozen && (first || !(v <= best));
            if (take) {
                best = v;
                arg = j0 + j;
                frozen = (v != v);
            }
        }
    }
    if (regular && m > 0) {
        best = rb.bi / rb.bu;
        arg = rb.arg;
    }
    if (i >= N) return;
    long long o = (long long)b * N + i;
    if (max_iou_out) max_iou_out[o] = best;
    if (argmax_out) argmax_out[o] = arg;
    if (label_out) {
        // utils.py:275-290: zeros; pos -> 1 / labels[arg]; then neg -> 0; then mid -> -1
        float lab = 0.0f;
        if (best > pos_thr) lab = labels ? labels[(long long)b * m_max + arg] : 1.0f;
        if (best < neg_thr) lab = 0.0f;
        if (best >= neg_thr && best <= pos_thr) lab = -1.0f;
        label_out[o] = lab;
    }
    if (dev_out) {
        // utils.py:147-161 on the argmax ground truth only (the reference builds all (N,M,4))
        float4 g = gt[arg];
        float w1 = a.z - a.x, h1 = a.w - a.y;
        float cx1 = a.x + w1 / 2.0f, cy1 = a.y + h1 / 2.0f;
        float w2 = g.z - g.x, h2 = g.w - g.y;
        float cx2 = g.x + w2 / 2.0f, cy2 = g.y + h2 / 2.0f;
        dev_out[o] = make_float4(wt.x * ((cx2 - cx1) / w1), wt.y * ((cy2 - cy1) / h1),
                                 wt.w * logf(w2 / w1), wt.h * logf(h2 / h1));
    }
}

extern "C" int frcnn_iou_label_encode_f32(const float *box1, int64_t box1_batch_stride,
                                          const float *box2, const int32_t *m_per_image,
                                          const float *labels, int B, int N, int m_max,
                                          float neg_threshold, float pos_threshold,
                                          const float *weights_host, float *label_out,
                                          int32_t *argmax_out, float *max_iou_out, float *dev_out,
                                          void *stream)
{
    FRCNN_REQUIRE(B > 0 && N >= 0 && m_max > 0, "iou_label_encode: bad shape (every image needs >= 1 box_2 row)");
    if (N == 0) return FRCNN_OK;
    FRCNN_REQUIRE(box1 && box2, "iou_label_encode: null pointer");
    FRCNN_REQUIRE(box1_batch_stride % 4 == 0, "iou_label_encode: box1 stride not a box multiple");
    Weights4 wt = {1.0f, 1.0f, 1.0f, 1.0f};
    if (weights_host) wt = {weights_host[0], weights_host[1], weights_host[2], weights_host[3]};
    hipLaunchKernelGGL(iou_label_encode_kernel, dim3((N + 255) / 256, B), dim3(256), 0,
                       (hipStream_t)stream, (const float4 *)box1,
                       (long long)(box1_batch_stride / 4), (const float4 *)box2, m_per_image, labels,
                       N, m_max, neg_threshold, pos_threshold, wt, label_out, argmax_out,
                       max_iou_out, (float4 *)dev_out);
    FRCNN_CHECK_LAUNCH("iou_label_encode");
    return FRCNN_OK;
}
