// nms.hip -- K4: batched greedy NMS (torchvision.ops.nms semantics; call sites RPN/rpn.py:84 and
// ROIHead/roi_head.py:215).
//
// Two kernels, both batched over images:
//   nms_mask_kernel : 64x64 suppression tiles.  Lane = column box, the 64 row boxes of the tile
//                     are broadcast from LDS; `iou > thr` is evaluated for 64 columns at once and
//                     the wave's compare mask (the ballot) IS the 64-bit suppression word of the
//                     row.  Only the upper triangle (column block >= row block) is computed.
//                     Tiles are stored tile-major, [row block][col block][row in block], so every
//                     wave writes one coalesced 512-B line set.
//   nms_scan_kernel : one workgroup per image walks the row blocks in order; wave 0 resolves the
//                     64 candidates of a block against the running `removed` word (scalar loop over
//                     surviving candidates only: ffs / readlane / or), then all 16 waves fold the
//                     kept rows' tiles into `removed` for the later column blocks with a wave-wide
//                     OR reduction.  Stops as soon as max_keep boxes are kept.
//
// IoU is the reference's fp32 expression with IEEE division; the decision `(double)iou > thr` is
// made with the float threshold t = RD(thr) (largest fp32 <= thr), which is equivalent for every
// fp32 iou.  NaN never suppresses.
#include <math.h>

#include "common.h"

#define NMS_SCAN_THREADS 1024
#define NMS_SCAN_WAVES 16
#define NMS_COL_WAVES 4 // waves per mask block; each wave owns one column block at a time

__global__ void __launch_bounds__(256)
nms_mask_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ n_per_image,
                int max_n, int W, float thr, unsigned long long *__restrict__ mask)
{
    __shared__ float4 s_row[64];
    __shared__ float s_row_area[64];
    const int b = blockIdx.z;
    const int n = n_per_image ? min(n_per_image[b], max_n) : max_n;
    const int rb = blockIdx.x;
    if (rb * 64 >= n) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int Wn = (n + 63) >> 6;
    const float4 *bx = boxes + (long long)b * max_n;
    if (threadIdx.x < 64) {
        int r = rb * 64 + threadIdx.x;
        float4 v = r < n ? bx[r] : make_float4(0, 0, 0, 0);
        s_row[threadIdx.x] = v;
        s_row_area[threadIdx.x] = frcnn_area(v.x, v.y, v.z, v.w);
    }
    __syncthreads();
    unsigned long long *mimg = mask + (long long)b * W * W * 64;
    const int stride = gridDim.y * NMS_COL_WAVES;
    for (int cb = rb + blockIdx.y * NMS_COL_WAVES + wave; cb < Wn; cb += stride) {
        int c = cb * 64 + lane;
        bool cvalid = c < n;
        float4 cbx = cvalid ? bx[c] : make_float4(0, 0, 0, 0);
        float carea = frcnn_area(cbx.x, cbx.y, cbx.z, cbx.w);
        unsigned long long word = 0;
        const bool diag = cb == rb;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) {
            float4 r = s_row[i];
            float iou = frcnn_iou(r.x, r.y, r.z, r.w, s_row_area[i], cbx.x, cbx.y, cbx.z, cbx.w,
                                  carea);
            bool sup = cvalid && (iou > thr) && (!diag || lane > i);
            unsigned long long m = __ballot(sup);
            if (lane == i) word = m;
        }
        mimg[((long long)rb * W + cb) * 64 + lane] = word;
    }
}

__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo |= __shfl_xor(lo, o);
        hi |= __shfl_xor(hi, o);
    }
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v)
{
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

__global__ void __launch_bounds__(NMS_SCAN_THREADS)
nms_scan_kernel(const unsigned long long *__restrict__ mask, const int32_t *__restrict__ n_per_image,
                int max_n, int W, int max_keep, int32_t *__restrict__ keep_out,
                int32_t *__restrict__ n_keep_out)
{
    extern __shared__ unsigned long long s_removed[]; // W words
    __shared__ unsigned long long s_keep;
    __shared__ int s_nkeep;
    const int b = blockIdx.x;
    const int n = n_per_image ? min(n_per_image[b], max_n) : max_n;
    const int Wn = (n + 63) >> 6;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned long long *mimg = mask + (long long)b * W * W * 64;
    int32_t *keep = keep_out + (long long)b * max_keep;
    for (int t = threadIdx.x; t < Wn; t += NMS_SCAN_THREADS) s_removed[t] = 0;
    if (threadIdx.x == 0) s_nkeep = 0;
    __syncthreads();
    for (int c = 0; c < Wn; ++c) {
        if (wave == 0) {
            int nkeep = s_nkeep;
            unsigned long long d = mimg[((long long)c * W + c) * 64 + lane]; // diagonal tile
            int rows = min(64, n - c * 64);
            unsigned long long valid = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
            unsigned long long rem = uniform64(s_removed[c]);
            unsigned long long K = 0;
            unsigned long long cand = ~rem & valid;
            int room = max_keep - nkeep;
            while (cand != 0 && room > 0) {
                int bit = __ffsll((long long)cand) - 1;
                K |= 1ull << bit;
                --room;
                unsigned lo = __builtin_amdgcn_readlane((unsigned)d, bit);
                unsigned hi = __builtin_amdgcn_readlane((unsigned)(d >> 32), bit);
                rem |= ((unsigned long long)hi << 32) | lo;
                unsigned long long above = bit >= 63 ? 0ull : (~0ull << (bit + 1));
                cand = uniform64(~rem & valid & above);
            }
            if ((K >> lane) & 1ull) {
                int pos = nkeep + __popcll(K & ((1ull << lane) - 1ull));
                keep[pos] = c * 64 + lane;
            }
            if (lane == 0) {
                s_keep = K;
                s_nkeep = nkeep + __popcll(K);
            }
        }
        __syncthreads();
        const unsigned long long K = s_keep;
        const bool done = s_nkeep >= max_keep;
        if (!done && K != 0) {
            for (int cb = c + 1 + wave; cb < Wn; cb += NMS_SCAN_WAVES) {
                unsigned long long v = mimg[((long long)c * W + cb) * 64 + lane];
                v = ((K >> lane) & 1ull) ? v : 0ull;
                v = wave_or64(v);
                if (lane == 0) s_removed[cb] |= v;
            }
        }
        __syncthreads();
        if (done) break;
    }
    if (threadIdx.x == 0) n_keep_out[b] = s_nkeep;
}

// ---- lazy ("pull") NMS ------------------------------------------------------------------------
// One 1024-thread workgroup per image, no suppression matrix at all.  The kept boxes live in LDS.
// For each block of 64 candidates (lane = candidate):
//   1. pull   : all 16 waves test the 64 candidates against the boxes kept so far (wave w takes
//               kept rows w, w+16, ...; kept box broadcast from LDS) -> ballot -> LDS atomicOr;
//   2. diag   : for the candidates still alive, the 16 waves compute their 64-bit intra-block
//               suppression rows in parallel (alive candidate #a goes to wave a % 16);
//   3. resolve: wave 0 walks the alive candidates in order with scalar ffs / readlane / or, and
//               appends the survivors to the kept list.
// Work = (kept so far) x 64 pairs per block, and the loop stops at max_keep keeps, so only the
// part of the IoU matrix that greedy NMS actually consults is ever evaluated (a few million pairs
// instead of n^2/2 = 72 M at n = 12,000).  Same fp32 IoU expression, same threshold rule.
#define NMS_LAZY_THREADS 1024
#define NMS_LAZY_WAVES 16
#define NMS_LAZY_LDS_LIMIT (120 * 1024)

__device__ __forceinline__ float lane_bcast(float v, int src_lane)
{
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), src_lane));
}

__global__ void __launch_bounds__(NMS_LAZY_THREADS)
nms_lazy_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ n_per_image, int max_n,
                float thr, int max_keep, int cap, int32_t *__restrict__ keep_out,
                int32_t *__restrict__ n_keep_out)
{
    // dynamic LDS: [cap] kept boxes (float4), [cap] kept areas, [cap] kept positions
    extern __shared__ float4 s_kbox[];
    float *s_karea = (float *)(s_kbox + cap);
    int *s_kidx = (int *)(s_karea + cap);
    __shared__ float4 s_cand[NMS_LAZY_THREADS]; // the next 16 blocks of candidates, staged per refill
    __shared__ unsigned long long s_row[64];
    __shared__ unsigned s_rem[2];
    __shared__ int s_nkeep;
    const int b = blockIdx.x;
    const int n = n_per_image ? min(n_per_image[b], max_n) : max_n;
    const int Wn = (n + 63) >> 6;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const float4 *bx = boxes + (long long)b * max_n;
    if (threadIdx.x == 0) {
        s_nkeep = 0;
        s_rem[0] = 0;
        s_rem[1] = 0;
    }
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int nkeep_final = 0;
    for (int c0 = 0; c0 < Wn; c0 += NMS_LAZY_WAVES) {
        // refill: one coalesced 16-KB read serves the next 16 candidate blocks; no global memory
        // operation is issued inside the per-block loop below, so its barriers are cheap
        {
            int idx = c0 * 64 + threadIdx.x;
            s_cand[threadIdx.x] = idx < n ? bx[idx] : zero4;
        }
        __syncthreads();
        const int cend = min(Wn, c0 + NMS_LAZY_WAVES);
        bool done = false;
        for (int c = c0; c < cend; ++c) {
            const float4 cb = s_cand[(c - c0) * 64 + lane];
            const float carea = frcnn_area(cb.x, cb.y, cb.z, cb.w);
            const int nk = s_nkeep;
            // 1. pull
            bool sup = false;
            for (int r = wave; r < nk; r += NMS_LAZY_WAVES) {
                float4 k = s_kbox[r];
                float iou = frcnn_iou(k.x, k.y, k.z, k.w, s_karea[r], cb.x, cb.y, cb.z, cb.w, carea);
                sup = sup || (iou > thr);
            }
            unsigned long long m = __ballot(sup);
            if (lane == 0 && m != 0) {
                atomicOr(&s_rem[0], (unsigned)m);
                atomicOr(&s_rem[1], (unsigned)(m >> 32));
            }
            __syncthreads();
            const int rows = min(64, n - c * 64);
            const unsigned long long valid = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
            unsigned long long rem =
                uniform64((((unsigned long long)s_rem[1] << 32) | s_rem[0]) | ~valid);
            const unsigned long long alive = ~rem;
            // 2. diag rows of the alive candidates, spread over the waves
            {
                unsigned long long rest = alive;
                int a = 0;
                while (rest != 0) {
                    int i = __ffsll((long long)rest) - 1;
                    rest &= rest - 1;
                    if ((a & (NMS_LAZY_WAVES - 1)) == wave) {
                        float rx1 = lane_bcast(cb.x, i), ry1 = lane_bcast(cb.y, i);
                        float rx2 = lane_bcast(cb.z, i), ry2 = lane_bcast(cb.w, i);
                        float rarea = lane_bcast(carea, i);
                        float iou = frcnn_iou(rx1, ry1, rx2, ry2, rarea, cb.x, cb.y, cb.z, cb.w, carea);
                        unsigned long long w = __ballot(lane > i && iou > thr);
                        if (lane == 0) s_row[i] = w;
                    }
                    ++a;
                }
            }
            __syncthreads();
            // 3. resolve
            if (wave == 0) {
                const unsigned long long myrow = s_row[lane];
                unsigned long long K = 0;
                unsigned long long cand = alive;
                int room = max_keep - nk;
                while (cand != 0 && room > 0) {
                    int i = __ffsll((long long)cand) - 1;
                    K |= 1ull << i;
                    --room;
                    unsigned lo = __builtin_amdgcn_readlane((unsigned)myrow, i);
                    unsigned hi = __builtin_amdgcn_readlane((unsigned)(myrow >> 32), i);
                    rem |= ((unsigned long long)hi << 32) | lo;
                    unsigned long long above = i >= 63 ? 0ull : (~0ull << (i + 1));
                    cand = uniform64(~rem & above);
                }
                if ((K >> lane) & 1ull) {
                    int pos = nk + __popcll(K & ((1ull << lane) - 1ull));
                    s_kbox[pos] = cb;
                    s_karea[pos] = carea;
                    s_kidx[pos] = c * 64 + lane;
                }
                if (lane == 0) {
                    s_nkeep = nk + __popcll(K);
                    s_rem[0] = 0;
                    s_rem[1] = 0;
                }
            }
            __syncthreads();
            nkeep_final = s_nkeep;
            if (nkeep_final >= max_keep) {
                done = true;
                break;
            }
        }
        if (done) break;
    }
    __syncthreads();
    nkeep_final = s_nkeep;
    int32_t *keep = keep_out + (long long)b * max_keep;
    for (int t = threadIdx.x; t < nkeep_final; t += NMS_LAZY_THREADS) keep[t] = s_kidx[t];
    if (threadIdx.x == 0) n_keep_out[b] = nkeep_final;
}

extern "C" size_t frcnn_nms_workspace_bytes(int B, int max_n)
{
    if (B <= 0 || max_n <= 0) return 0;
    size_t W = ((size_t)max_n + 63) / 64;
    return (size_t)B * W * W * 64 * sizeof(unsigned long long);
}

// Largest fp32 t with (double)t <= thr:  (double)x > thr  <=>  x > t  for every fp32 x.
static float threshold_round_down(double thr)
{
    float t = (float)thr;
    if ((double)t > thr) t = nextafterf(t, -INFINITY);
    return t;
}

extern "C" int frcnn_nms_sorted_f32(const float *boxes, const int32_t *n_per_image, int B, int max_n,
                                    double iou_threshold, int max_keep, int32_t *keep_out,
                                    int32_t *n_keep_out, void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    FRCNN_REQUIRE(B >= 0 && max_n >= 0 && max_keep >= 0, "nms: bad shape");
    if (B == 0) return FRCNN_OK;
    FRCNN_REQUIRE(n_keep_out, "nms: null n_keep_out");
    if (max_n == 0 || max_keep == 0) {
        hipError_t e = hipMemsetAsync(n_keep_out, 0, sizeof(int32_t) * (size_t)B, (hipStream_t)stream);
        FRCNN_REQUIRE(e == hipSuccess, "nms: memset failed: %s", hipGetErrorString(e));
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(boxes && keep_out, "nms: null pointer");
    FRCNN_REQUIRE(iou_threshold == iou_threshold, "nms: NaN threshold");
    FRCNN_REQUIRE(B <= 65535, "nms: batch too large");
    const float thr = threshold_round_down(iou_threshold);

    // Lazy path: the kept list (20 B per keep) must fit in LDS.  keep_out has max_keep columns,
    // but no more than max_n boxes can ever be kept.
    const int cap = max_keep < max_n ? max_keep : max_n;
    const size_t lazy_lds = (size_t)cap * 24;
    if (lazy_lds <= NMS_LAZY_LDS_LIMIT) {
        static size_t configured = 48 * 1024;
        if (lazy_lds > configured) {
            hipError_t e = hipFuncSetAttribute((const void *)nms_lazy_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               NMS_LAZY_LDS_LIMIT);
            FRCNN_REQUIRE(e == hipSuccess, "nms: cannot reserve LDS: %s", hipGetErrorString(e));
            configured = NMS_LAZY_LDS_LIMIT;
        }
        hipLaunchKernelGGL(nms_lazy_kernel, dim3(B), dim3(NMS_LAZY_THREADS), lazy_lds,
                           (hipStream_t)stream, (const float4 *)boxes, n_per_image, max_n, thr,
                           max_keep, cap, keep_out, n_keep_out);
        FRCNN_CHECK_LAUNCH("nms_lazy");
        return FRCNN_OK;
    }

    // Bitmask path (very large keep budgets): full upper-triangular suppression matrix + scan.
    FRCNN_REQUIRE(workspace, "nms: null workspace");
    if (workspace_bytes < frcnn_nms_workspace_bytes(B, max_n)) {
        frcnn_set_error("nms: workspace too small (%zu < %zu)", workspace_bytes,
                        frcnn_nms_workspace_bytes(B, max_n));
        return FRCNN_EWORKSPACE;
    }
    const int W = (max_n + 63) / 64;
    FRCNN_REQUIRE((size_t)W * 8 <= 60 * 1024, "nms: max_n too large for the scan's LDS bitmap");
    int col_groups = (W + 4 * NMS_COL_WAVES - 1) / (4 * NMS_COL_WAVES); // ~4 column blocks per wave
    if (col_groups < 1) col_groups = 1;
    if (col_groups > 64) col_groups = 64;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(W, col_groups, B), dim3(64 * NMS_COL_WAVES), 0,
                       (hipStream_t)stream, (const float4 *)boxes, n_per_image, max_n, W, thr,
                       (unsigned long long *)workspace);
    FRCNN_CHECK_LAUNCH("nms_mask");
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(NMS_SCAN_THREADS), (size_t)W * 8,
                       (hipStream_t)stream, (const unsigned long long *)workspace, n_per_image,
                       max_n, W, max_keep, keep_out, n_keep_out);
    FRCNN_CHECK_LAUNCH("nms_scan");
    return FRCNN_OK;
}
