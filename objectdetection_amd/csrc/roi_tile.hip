// roi_tile.hip -- K6 for feature pyramids: every level of MultiScaleRoIAlign in ONE launch.
// Replaces torchvision.ops.MultiScaleRoIAlign as used by ROIHead/roi_pooling.py:44-53 (call site
// ROIHead/roi_head.py:127-130) on the FPN of BackBone/fpn.py:91-96.
//
// The plane kernel of roi.hip keeps 16 channels of a whole map in LDS, which stops at maps of about
// 37 x 64 cells.  Here every level is cut into TILES of RT_WIN x RT_WIN cells, one every RT_STEP cells, 16
// channels of a tile resident (147 KB), and the unit of ownership is the BIN, not the RoI: a bin belongs
// to the RoI's base tile when its taps lie inside it -- then the whole RoI is one visit -- and otherwise
// to the tile that holds the bin's first tap.  A bin's taps span at most RT_WIN - RT_STEP + 1 = 17 cells
// unless the RoI is wider than 105 cells on that level, so there is no RoI size limit worth the name
// (the rare exception goes to the direct kernel of roi.hip), the tiles overlap by a third instead of a
// half, and an item amortises its tap set-up over 16 channels as on the headline path.
// A workgroup = (level, tile, image, channel split): it examines the image's boxes once, lists the RoIs
// with a bin of its tile, then stages and sweeps one 16-channel chunk after the other.  Levels are block
// roles of one grid, finest (most RoIs) first.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#define RAP_STORE_NT 1 // see rap_out (roi_common.h)
#include "roi_common.h"

struct TileLevel {
    const float *feat; // (B, C, H, W)
    int H, W;
    float scale;
    int nwy, nwx;
    int first; // first unit (tile) of this level in the launch's unit order
};
struct TileLevels {
    TileLevel l[FRCNN_MAX_LEVELS];
    int n;     // levels
    int units; // tiles over all levels
};

// Staging of one 16-channel chunk of a tile: pixel p = thread + 1024 k of the tile (row-major), four plane reads
// -> one 16-byte LDS write per pixel and channel group; everything is fetched before anything is written (one
// exposed round trip per chunk).
#ifndef RT_THREADS
#define RT_THREADS 1024
#endif
#define RT_PIX (RT_WIN * RT_WIN)
#define RT_PASSES ((RT_PIX + RT_THREADS - 1) / RT_THREADS)
template <int GT>
__device__ __forceinline__ void tile_stage(float4 *s_plane, const float *__restrict__ fb, int G, int nc, int HW, int W,
                                           int wy0, int wx0, int wh, int ww)
{
    float4 v[RT_PASSES * RT_G];
#pragma unroll
    for (int k = 0; k < RT_PASSES; ++k) {
        const int p = (int)threadIdx.x + k * RT_THREADS;
        const int y = p / RT_WIN, x = p - y * RT_WIN;
        const bool in = p < RT_PIX && y < wh && x < ww;
        const float *f0 = fb + (long long)(wy0 + y) * W + wx0 + x;
#pragma unroll
        for (int g = 0; g < RT_G; ++g) {
            float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (in && (GT || g < G)) {
                const float *f = f0 + (long long)(4 * g) * HW;
                t.x = f[0];
                if (GT || 4 * g + 1 < nc) t.y = f[HW];
                if (GT || 4 * g + 2 < nc) t.z = f[2 * HW];
                if (GT || 4 * g + 3 < nc) t.w = f[3 * HW];
            }
            v[k * RT_G + g] = t;
        }
    }
#pragma unroll
    for (int k = 0; k < RT_PASSES; ++k) {
        const int p = (int)threadIdx.x + k * RT_THREADS;
        const int y = p / RT_WIN, x = p - y * RT_WIN;
        if (p < RT_PIX && y < wh && x < ww) {
#pragma unroll
            for (int g = 0; g < RT_G; ++g)
                if (GT || g < G) s_plane[(y * G + g) * RT_WIN + (x ^ rap_swz(y))] = v[k * RT_G + g];
        }
    }
}

// The same staging with 16-byte loads: four neighbouring pixels of a row per load and channel, transposed in
// registers into four 4-channel pixels.  Needs W % 4 == 0 and a 16-byte aligned map (tiles start on multiples of
// RT_STEP pixels); a quarter of the load instructions.
#define RT_QPR (RT_WIN / 4)                 // quads per tile row
#define RT_NQ (RT_WIN * RT_QPR * RT_G)      // (group, row, quad) triples
#define RT_QPASSES ((RT_NQ + RT_THREADS - 1) / RT_THREADS)
static_assert(RT_STEP % 4 == 0, "tiles start on quads");
template <int GT>
__device__ __forceinline__ void tile_fetch_quads(float4 (&v)[RT_QPASSES][4], const float *__restrict__ fb, int G, int nc,
                                                 int HW, int W, int wy0, int wx0, int wh, int ww)
{
#pragma unroll
    for (int k = 0; k < RT_QPASSES; ++k) {
        const int t = (int)threadIdx.x + k * RT_THREADS;
        const int g = t / (RT_WIN * RT_QPR), r = t - g * (RT_WIN * RT_QPR);
        const int y = r / RT_QPR, q = r - y * RT_QPR;
        const bool in = t < RT_NQ && y < wh && 4 * q < ww && (GT || g < G);
        const float *f = fb + (long long)(4 * g) * HW + (long long)(wy0 + y) * W + wx0 + 4 * q;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            v[k][c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (in && (GT || 4 * g + c < nc)) v[k][c] = *reinterpret_cast<const float4 *>(f + (long long)c * HW);
        }
    }
}

template <int GT>
__device__ __forceinline__ void tile_store_quads(float4 *s_plane, const float4 (&v)[RT_QPASSES][4], int G, int wh, int ww)
{
#pragma unroll
    for (int k = 0; k < RT_QPASSES; ++k) {
        const int t = (int)threadIdx.x + k * RT_THREADS;
        const int g = t / (RT_WIN * RT_QPR), r = t - g * (RT_WIN * RT_QPR);
        const int y = r / RT_QPR, q = r - y * RT_QPR;
        if (t < RT_NQ && y < wh && 4 * q < ww && (GT || g < G)) {
            float4 *row = s_plane + (y * G + g) * RT_WIN;
            const int sw = rap_swz(y);
            row[(4 * q) ^ sw] = make_float4(v[k][0].x, v[k][1].x, v[k][2].x, v[k][3].x);
            row[(4 * q + 1) ^ sw] = make_float4(v[k][0].y, v[k][1].y, v[k][2].y, v[k][3].y);
            row[(4 * q + 2) ^ sw] = make_float4(v[k][0].z, v[k][1].z, v[k][2].z, v[k][3].z);
            row[(4 * q + 3) ^ sw] = make_float4(v[k][0].w, v[k][1].w, v[k][2].w, v[k][3].w);
        }
    }
}

#define RT_ZERO 0x40000000 // list entry: no level takes the RoI, its rows are zeros (written by unit 0)

template <int GT>
__global__ void __launch_bounds__(RT_THREADS, 4)
roi_align_tile_kernel(TileLevels L, int C, const float4 *__restrict__ props, const int32_t *__restrict__ count, int P,
                      int k_min, int k_max, int csplit, int B, float *__restrict__ out,
                      int32_t *__restrict__ total_out)
{
    extern __shared__ float4 s_plane[];
    __shared__ float4 s_rec[RT_LIST];
    __shared__ int s_idx[RT_LIST]; // RoI index | base tile row << 20 | base tile column << 25 | RT_ZERO
    __shared__ int s_off, s_cnt, s_n;
    // Block order.  Blocks i and i + 8 share an XCD (observed dispatch order; speed only).  An XCD is given whole
    // images (batches of 8 or more), units in the outer loop -- finest level first, so the long workgroups start
    // first -- and neighbouring tiles of an image close together in time, so that the cells two tiles share are
    // still in that XCD's L2 when the second one asks.  The channel splits of a (unit, image) are consecutive.
    const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int cs = slot % csplit;
    const int r = slot / csplit;
    int b, u;
    if (B >= 8) {
        const int per_xcd = (B + 7) >> 3;
        u = r / per_xcd;
        b = (r % per_xcd) * 8 + xcd;
        if (b >= B) return;
    } else {
        const int q = (r >> 2) * 32 + xcd * 4 + (r & 3); // runs of four consecutive (image, unit) pairs per XCD
        if (q >= B * L.units) return;
        b = q / L.units;
        u = q - b * L.units;
    }
    // the unit's level, by compare-and-select over statically indexed entries (scalar registers); levels own
    // ranges of units in an order of the host's choosing (finest first)
    const float *feat = L.l[0].feat;
    int H = L.l[0].H, W = L.l[0].W, nwy = L.l[0].nwy, nwx = L.l[0].nwx, first = L.l[0].first, level = 0;
    float scale = L.l[0].scale;
#pragma unroll
    for (int l = 1; l < FRCNN_MAX_LEVELS; ++l) {
        const bool in = l < L.n && u >= L.l[l].first && u < L.l[l].first + L.l[l].nwy * L.l[l].nwx;
        feat = in ? L.l[l].feat : feat;
        H = in ? L.l[l].H : H;
        W = in ? L.l[l].W : W;
        nwy = in ? L.l[l].nwy : nwy;
        nwx = in ? L.l[l].nwx : nwx;
        scale = in ? L.l[l].scale : scale;
        first = in ? L.l[l].first : first;
        level = in ? l : level;
    }
    const int wid = u - first;
    const int wy = wid / nwx, wx = wid - wy * nwx;
    const int wy0 = wy * RT_STEP, wx0 = wx * RT_STEP;
    const int chunks = (C + 4 * RT_G - 1) / (4 * RT_G);
    const int HW = H * W;
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
        if (total_out && cs == 0 && b == 0 && u == 0) *total_out = tot;
    }
    __syncthreads();
    const int cnt = s_cnt;
    const float4 *pb = props + (long long)b * P;
    const int wh = min(RT_WIN, H - wy0), ww = min(RT_WIN, W - wx0);
    const int bin = threadIdx.x & 63;
    const int ph = bin / 7, pw = bin - ph * 7;
    const bool quads = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0;
    for (int seg = 0; seg < cnt; seg += RT_THREADS) {
        // One box per thread.  A RoI is listed when this tile lies in the range of tiles that can own one of its bins;
        // which bins it does own is decided per lane by the items.  The list is worked off when it is full (the
        // threads left over then look at their box again) and at the end.
        bool pending = seg + (int)threadIdx.x < cnt;
        const bool last_seg = seg + RT_THREADS >= cnt;
      for (;;) {
        if (pending) {
            const int i = seg + (int)threadIdx.x;
            const float4 box = pb[i];
            float4 rec = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            int code = -1;
            int lv = 0;
            if (L.n > 1) lv = map_level(box, k_min, k_max);
            if ((lv < 0 || lv >= L.n) && u == 0) {
                code = i | RT_ZERO;
            } else if (lv == level) {
                rec = roi_record(box, scale);
                int by, ly, bx, lx;
                bool fits = true;
                tile_axis_range(rec.y, rec.w, H, nwy, by, ly, fits);
                tile_axis_range(rec.x, rec.z, W, nwx, bx, lx, fits);
                if (fits && wy >= by && wy <= ly && wx >= bx && wx <= lx) code = i | (by << 20) | (bx << 25);
            }
            pending = false;
            if (code >= 0) {
                const int pos = atomicAdd(&s_n, 1);
                if (pos < RT_LIST) {
                    s_rec[pos] = rec;
                    s_idx[pos] = code;
                } else {
                    pending = true;
                }
            }
        }
        __syncthreads();
        const int filled = s_n;
        __syncthreads(); // everyone has read the fill before anyone changes it
        const bool overflow = filled > RT_LIST;
        if (!overflow && !last_seg) break; // room left: examine more boxes first
        const int n_list = min(filled, RT_LIST);
        for (int ch = cs; ch < (n_list ? chunks : 0); ch += csplit) {
            const int c0 = ch * (4 * RT_G);
            const int nc = min(4 * RT_G, C - c0);
            const int G = GT ? GT : (nc + 3) >> 2;
            // The loads of a chunk are issued by every wave as soon as IT is done with the previous chunk's items --
            // before the barrier that waits for the slowest wave -- and land while that barrier is waited for.
            const float *fb = feat + ((long long)b * C + c0) * HW;
#ifndef RT_EXP_NOSTAGE
            if (quads) {
                float4 v[RT_QPASSES][4];
                tile_fetch_quads<GT>(v, fb, G, nc, HW, W, wy0, wx0, wh, ww);
                __syncthreads(); // the previous chunk's readers are done with the tile
                tile_store_quads<GT>(s_plane, v, G, wh, ww);
            } else {
                __syncthreads();
                tile_stage<GT>(s_plane, fb, G, nc, HW, W, wy0, wx0, wh, ww);
            }
#endif
            __syncthreads();
            const int row_stride = G * RT_WIN;
            float *ob = out + ((long long)s_off * C + c0) * 49;
            // one whole RoI per wave iteration, lane = bin (see roi_align_plane_kernel); the lanes whose bin
            // another tile owns sit the item out
#ifdef RT_EXP_NOITEMS
            if (n_list >= 0) continue;
#endif
            for (int e = (int)(threadIdx.x >> 6); e < (bin < 49 ? n_list : 0); e += RT_THREADS / 64) {
                const float4 rec = s_rec[e];
                const int code = s_idx[e];
                float *o = ob + (long long)(code & 0xFFFFF) * C * 49 + bin;
                if (code & RT_ZERO) {
                    for (int c = 0; c < nc; ++c) o[c * 49] = 0.0f;
                    continue;
                }
                AxisSample ys[2], xs[2];
                tile_bin_samples(rec.y, rec.w, ph, H, ys);
                tile_bin_samples(rec.x, rec.z, pw, W, xs);
                bool fits = true;
                const bool active = tile_owner(ys, (code >> 20) & 31, nwy, fits) == wy &&
                                    tile_owner(xs, (code >> 25) & 31, nwx, fits) == wx;
                const bool y3 = __all(axis_mergeable(ys[0], ys[1]));
                const bool x3 = __all(axis_mergeable(xs[0], xs[1]));
                AxisTaps ty = y3 ? axis_taps3(ys[0], ys[1], 0.25f, H) : axis_taps4(ys[0], ys[1], 0.25f);
                AxisTaps tx = x3 ? axis_taps3(xs[0], xs[1], 1.0f, W) : axis_taps4(xs[0], xs[1], 1.0f);
#pragma unroll
                for (int i = 0; i < 4; ++i) { // relative to the tile; weightless cells are pulled inside
                    ty.idx[i] = min(max(ty.idx[i] - wy0, 0), RT_WIN - 1);
                    tx.idx[i] = min(max(tx.idx[i] - wx0, 0), RT_WIN - 1);
                }
                if (active) {
                    if (y3) {
                        if (x3)
                            rap_item<3, 3, GT>(s_plane, RT_WIN, row_stride, G, nc, ty, tx, o);
                        else
                            rap_item<3, 4, GT>(s_plane, RT_WIN, row_stride, G, nc, ty, tx, o);
                    } else {
                        if (x3)
                            rap_item<4, 3, GT>(s_plane, RT_WIN, row_stride, G, nc, ty, tx, o);
                        else
                            rap_item<4, 4, GT>(s_plane, RT_WIN, row_stride, G, nc, ty, tx, o);
                    }
                }
            }
        }
        __syncthreads(); // every reader of the list is done
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        if (!overflow) break;
      }
    }
}

static int tiles_supported(const int32_t *hw, int n_levels, int P, int pooled_h, int pooled_w, int sampling_ratio)
{
    if (n_levels < 1 || n_levels > FRCNN_MAX_LEVELS) return 0;
    if (pooled_h != 7 || pooled_w != 7 || sampling_ratio != 2 || P >= (1 << 20)) return 0;
    for (int l = 0; l < n_levels; ++l)
        if (tile_count(hw[2 * l]) > RT_MAX_TILES || tile_count(hw[2 * l + 1]) > RT_MAX_TILES) return 0;
    return 1;
}

extern "C" int frcnn_roi_align_levels_f32(const float *const *feats, const int32_t *hw, const float *scales,
                                          int n_levels, int B, int C, const float *props, const int32_t *count, int P,
                                          int pooled_h, int pooled_w, int sampling_ratio, int k_min, int k_max,
                                          float *out, int32_t *total_out, void *stream)
{
    FRCNN_REQUIRE(n_levels >= 1 && n_levels <= FRCNN_MAX_LEVELS, "roi_align_levels: 1..%d levels (got %d)",
                  FRCNN_MAX_LEVELS, n_levels);
    FRCNN_REQUIRE(feats && hw && scales, "roi_align_levels: null level table");
    FRCNN_REQUIRE(B > 0 && B <= 65535 && C > 0 && P >= 0 && pooled_h > 0 && pooled_w > 0, "roi_align_levels: bad shape");
    for (int l = 0; l < n_levels; ++l)
        FRCNN_REQUIRE(hw[2 * l] > 0 && hw[2 * l + 1] > 0 && (long long)hw[2 * l] * hw[2 * l + 1] < (1ll << 30),
                      "roi_align_levels: bad plane size on level %d", l);
    // A single map that the plane kernel holds with 16 channels (the headline shape), and everything the tiles do
    // not cover (other bin counts or sampling ratios): level by level through frcnn_roi_align_batched_f32.
    const bool plane16 = n_levels == 1 && (long long)hw[0] * ((hw[1] + 15) & ~15) * 4 * 16 <= RAP_LDS_BUDGET;
#ifdef RT_EXP_LEGACY
    const bool legacy = true; // experiment: the level-by-level launches of round 2
#else
    const bool legacy = false;
#endif
    if (legacy || P == 0 || plane16 || !tiles_supported(hw, n_levels, P, pooled_h, pooled_w, sampling_ratio)) {
        for (int l = 0; l < n_levels; ++l) {
            const int rc = frcnn_roi_align_batched_f32(feats[l], B, C, hw[2 * l], hw[2 * l + 1], props, count, P, pooled_h,
                                                       pooled_w, scales[l], sampling_ratio, l, n_levels, k_min, k_max, out,
                                                       total_out, stream);
            if (rc != FRCNN_OK) return rc;
        }
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(props && out, "roi_align_levels: null pointer");
    for (int l = 0; l < n_levels; ++l) FRCNN_REQUIRE(feats[l], "roi_align_levels: null feature map %d", l);
    TileLevels L;
    // unit order: the level with the most cells (the finest: most tiles, most RoIs) first.  The level INDEX a RoI
    // maps to stays the position in `feats`.
    int order[FRCNN_MAX_LEVELS];
    for (int l = 0; l < n_levels; ++l) order[l] = l;
    for (int i = 1; i < n_levels; ++i)
        for (int j = i; j > 0 && (long long)hw[2 * order[j]] * hw[2 * order[j] + 1] >
                                     (long long)hw[2 * order[j - 1]] * hw[2 * order[j - 1] + 1]; --j) {
            const int t = order[j];
            order[j] = order[j - 1];
            order[j - 1] = t;
        }
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
        TileLevel &t = L.l[l];
        t.feat = l < n_levels ? feats[l] : nullptr;
        t.H = l < n_levels ? hw[2 * l] : 1;
        t.W = l < n_levels ? hw[2 * l + 1] : 1;
        t.scale = l < n_levels ? scales[l] : 1.0f;
        t.nwy = tile_count(t.H);
        t.nwx = tile_count(t.W);
        t.first = 0;
    }
    int units = 0;
    for (int i = 0; i < n_levels; ++i) {
        L.l[order[i]].first = units;
        units += L.l[order[i]].nwy * L.l[order[i]].nwx;
    }
    L.n = n_levels;
    L.units = units;
    const int chunks = (C + 4 * RT_G - 1) / (4 * RT_G);
    // channel splits: about five workgroups per CU; a divisor of the chunk count when one is near
    const long long pairs = (long long)units * B;
    int csplit = (int)((RT_WG_TARGET + pairs - 1) / pairs);
    csplit = csplit < 1 ? 1 : (csplit > chunks ? chunks : csplit);
    for (int c = csplit; c <= chunks && c <= 2 * csplit; ++c)
        if (chunks % c == 0) {
            csplit = c;
            break;
        }
    const size_t lds = (size_t)RT_PIX * 16 * RT_G;
    const bool full = C % (4 * RT_G) == 0;
    {
        static size_t grantedg[FRCNN_MAX_DEVICES], granted0[FRCNN_MAX_DEVICES];
        hipError_t e = full ? frcnn_reserve_dynamic_lds((const void *)roi_align_tile_kernel<RT_G>, lds, grantedg)
                            : frcnn_reserve_dynamic_lds((const void *)roi_align_tile_kernel<0>, lds, granted0);
        FRCNN_REQUIRE(e == hipSuccess, "roi_align_levels: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    }
    const long long groups = B >= 8 ? (long long)((B + 7) / 8) * units * 8 : (pairs + 31) / 32 * 32;
    FRCNN_REQUIRE(groups * csplit < (1ll << 31), "roi_align_levels: grid too large");
    const dim3 grid((unsigned)(groups * csplit));
    if (full)
        hipLaunchKernelGGL((roi_align_tile_kernel<RT_G>), grid, dim3(RT_THREADS), lds, (hipStream_t)stream, L, C,
                           (const float4 *)props, count, P, k_min, k_max, csplit, B, out, total_out);
    else
        hipLaunchKernelGGL((roi_align_tile_kernel<0>), grid, dim3(RT_THREADS), lds, (hipStream_t)stream, L, C,
                           (const float4 *)props, count, P, k_min, k_max, csplit, B, out, total_out);
    FRCNN_CHECK_LAUNCH("roi_align_tile");
    // RoIs with a bin wider than a tile's overlap on a level of several tiles (wider than 105 cells): the direct kernel
    for (int l = 0; l < n_levels; ++l)
        if (L.l[l].nwy * L.l[l].nwx > 1) {
            const int rc = frcnn_roi_align_direct_behind_tiles(feats[l], B, C, hw[2 * l], hw[2 * l + 1], props, count, P,
                                                               scales[l], l, n_levels, k_min, k_max, out, stream);
            if (rc != FRCNN_OK) return rc;
        }
    return FRCNN_OK;
}
