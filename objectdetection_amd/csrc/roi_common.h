// roi_common.h -- pieces shared by the RoI pooling kernels (roi.hip) and their backward (roi_backward.hip).
#pragma once
#include "common.h"

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

__device__ __forceinline__ int map_level(float4 p, int k_min, int k_max)
{
    float area = (p.z - p.x) * (p.w - p.y);
    float s = sqrtf(area);
    float t = floorf((4.0f + log2f(s / 224.0f)) + 1e-6f);
    if (!(t == t)) return -1;
    t = t < (float)k_min ? (float)k_min : t;
    t = t > (float)k_max ? (float)k_max : t;
    return (int)t - k_min;
}


struct AxisSample { // one 1-D bilinear sample: taps lo/hi with weights (zero when out of range)
    int lo, hi;
    float wlo, whi;
    bool ok;
};

__device__ __forceinline__ AxisSample axis_sample(float v, int n)
{
    AxisSample a;
    bool ok = !(v < -1.0f || v > (float)n);
    if (v <= 0.0f) v = 0.0f;
    int lo = (int)v, hi;
    if (lo >= n - 1) {
        hi = lo = n - 1;
        v = (float)lo;
    } else {
        hi = lo + 1;
    }
    float l = v - (float)lo;
    float h = 1.0f - l;
    a.lo = ok ? lo : 0;
    a.hi = ok ? hi : 0;
    a.wlo = ok ? h : 0.0f;
    a.whi = ok ? l : 0.0f;
    a.ok = ok;
    return a;
}

// ---- windows of a plane too large for LDS (forward: roi_align_window_kernel; backward: its mirror) --------------
#ifndef RAW_WIN
#define RAW_WIN 64
#define RAW_STEP 32
#define RAW_G 2 // float4 channel groups per workgroup: RAW_WIN^2 * 16 B * RAW_G of LDS
#endif
struct RoiWindow {
    int win;   // iy * nwx + ix
    bool fits; // every tap of the RoI lies inside that window
};

__device__ __forceinline__ float4 roi_record(float4 box, float scale)
{
    float start_w = box.x * scale, start_h = box.y * scale;
    float roi_w = box.z * scale - start_w, roi_h = box.w * scale - start_h;
    roi_w = roi_w > 1.0f ? roi_w : 1.0f;
    roi_h = roi_h > 1.0f ? roi_h : 1.0f;
    return make_float4(start_w, start_h, roi_w / 7.0f, roi_h / 7.0f);
}

// Same expressions as the item loops use for bin 0 / sample 0 and bin 6 / sample 1: sample
// coordinates grow with (bin, sample), so every tap lies between the first sample's low cell and the
// last sample's high cell (samples outside the map have zero weight and are relocated by the caller).
__device__ __forceinline__ RoiWindow roi_window(float4 rec, int H, int W, int nwy, int nwx)
{
    const AxisSample yf = axis_sample(rec.y + 0.0f * rec.w + (0.0f + 0.5f) * rec.w / 2.0f, H);
    const AxisSample yl = axis_sample(rec.y + 6.0f * rec.w + (1.0f + 0.5f) * rec.w / 2.0f, H);
    const AxisSample xf = axis_sample(rec.x + 0.0f * rec.z + (0.0f + 0.5f) * rec.z / 2.0f, W);
    const AxisSample xl = axis_sample(rec.x + 6.0f * rec.z + (1.0f + 0.5f) * rec.z / 2.0f, W);
    const int y_lo = yf.ok ? yf.lo : 0, y_hi = yl.ok ? yl.hi : H - 1;
    const int x_lo = xf.ok ? xf.lo : 0, x_hi = xl.ok ? xl.hi : W - 1;
    const int iy = min(y_lo / RAW_STEP, nwy - 1), ix = min(x_lo / RAW_STEP, nwx - 1);
    RoiWindow r;
    r.win = iy * nwx + ix;
    r.fits = y_hi < iy * RAW_STEP + RAW_WIN && x_hi < ix * RAW_STEP + RAW_WIN;
    return r;
}

// ---- the LDS-resident plane scheme: pieces shared by the plane, window and tile kernels (roi.hip, roi_tile.hip) ----
#define RAP_THREADS 1024
#define RAP_LDS_BUDGET (150 * 1024) // dynamic LDS of the plane kernels (8 channels of a 50 x 96 plane); 8.7 KB static on top
#define RAP_SEG 512 // RoIs whose geometry is staged in LDS at a time (10 KB)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// acc += w * f for four channels as two packed fp32 FMAs (v_pk_fma_f32).
__device__ __forceinline__ void pk_axpy4(f32x2 &lo, f32x2 &hi, float w, const float4 &f)
{
    f32x2 w2 = {w, w};
    f32x2 a = {f.x, f.y}, c = {f.z, f.w};
    lo = __builtin_elementwise_fma(w2, a, lo);
    hi = __builtin_elementwise_fma(w2, c, hi);
}

// LDS image of a chunk: index (in float4 = 4-channel pixels) = (y * G + g) * WS + (x ^ swz(y)),
// WS = W rounded up to 16.  The G channel groups of a row sit WS pixels apart, so with a
// compile-time WS the per-group tap address is an IMMEDIATE ds_read offset (g * WS * 16 bytes) and
// the 16 tap addresses of an item are computed once for all its channels.  A pixel is one 16-byte
// slot, 16 slots span the 64 banks, and every row starts on a multiple of 16 slots, so the slot of
// (y, x) is (x ^ swz(y)) & 15; the 4-bit bit reversal swz(y) keeps bins of neighbouring rows that
// share a column off the same banks.
__device__ __forceinline__ int rap_swz(int y)
{
    return ((y & 1) << 3) | ((y & 2) << 1) | ((y & 4) >> 1) | ((y & 8) >> 3);
}

// The two samples of a bin along one axis touch 4 cells (lo0, hi0, lo1, hi1).  When the bin is at
// most 2 pixels wide the samples are at most one cell apart, so the 4 taps fall on 3 consecutive
// cells a, a+1, a+2 and their weights can be added up per cell: 3 LDS reads and 3 FMAs per row
// instead of 4 (9 instead of 16 per bin when both axes merge).  Same sum, different rounding order
// (a few ulp; the parity tests allow 1e-5, tests/test_gpu_parity.py).
struct AxisTaps {
    int idx[4];
    float w[4];
};

__device__ __forceinline__ bool axis_mergeable(const AxisSample &s0, const AxisSample &s1)
{
    return s0.ok && s1.ok && (unsigned)(s1.lo - s0.lo) <= 1u;
}

__device__ __forceinline__ AxisTaps axis_taps4(const AxisSample &s0, const AxisSample &s1, float k)
{
    AxisTaps t;
    t.idx[0] = s0.lo; t.idx[1] = s0.hi; t.idx[2] = s1.lo; t.idx[3] = s1.hi;
    t.w[0] = s0.wlo * k; t.w[1] = s0.whi * k; t.w[2] = s1.wlo * k; t.w[3] = s1.whi * k;
    return t;
}

__device__ __forceinline__ AxisTaps axis_taps3(const AxisSample &s0, const AxisSample &s1, float k, int n)
{
    AxisTaps t;
    const bool same = s1.lo == s0.lo;
    const int a = s0.lo;
    t.idx[0] = a;
    t.idx[1] = min(a + 1, n - 1);
    t.idx[2] = same ? a : min(a + 2, n - 1); // weight 0 when `same`: stay on a cell that is read anyway
    t.idx[3] = 0;
    // a clamped sample (lo == n-1) has whi == 0, so treating its hi tap as cell lo+1 is exact
    t.w[0] = (s0.wlo + (same ? s1.wlo : 0.0f)) * k;
    t.w[1] = (s0.whi + (same ? s1.whi : s1.wlo)) * k;
    t.w[2] = (same ? 0.0f : s1.whi) * k;
    t.w[3] = 0.0f;
    return t;
}

// The pooled output is written once and not read again by these kernels.  roi_tile.hip defines RAP_STORE_NT: its
// config-5 launch runs 0.856 -> 0.790 ms with `nt` stores; the plane kernel of roi.hip (headline shape) runs 1 %
// slower with them and keeps plain stores.
__device__ __forceinline__ void rap_out(float *p, float v)
{
#ifdef RAP_STORE_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// All channel groups of one item: NY x NX taps, each ONE ds_read_b128 serving 4 channels.
template <int NY, int NX, int GT>
__device__ __forceinline__ void rap_item(const float4 *s_plane, int WS, int row_stride, int G, int nc,
                                         const AxisTaps &ty, const AxisTaps &tx, float *o)
{
    int off[NY * NX];
    float wt[NY * NX];
#pragma unroll
    for (int r = 0; r < NY; ++r) {
        const int rb = __mul24(ty.idx[r], row_stride);
        const int sw = rap_swz(ty.idx[r]);
#pragma unroll
        for (int c = 0; c < NX; ++c) {
            off[r * NX + c] = (tx.idx[c] ^ sw) + rb;
            wt[r * NX + c] = ty.w[r] * tx.w[c];
        }
    }
    if (GT) {
#pragma unroll
        for (int g = 0; g < (GT ? GT : 1); ++g) {
            const float4 *pl = s_plane + g * WS; // immediate offset when WS is a compile-time constant
            f32x2 lo = {0.0f, 0.0f}, hi = {0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < NY * NX; ++q) pk_axpy4(lo, hi, wt[q], pl[off[q]]);
            float *og = o + (4 * g) * 49; // immediate store offsets
#ifdef RT_EXP_NOSTORE
            if (lo.x + lo.y + hi.x + hi.y != 1234.5f) continue; // experiment: arithmetic without the store stream
#endif
            rap_out(og, lo.x);
            rap_out(og + 49, lo.y);
            rap_out(og + 98, hi.x);
            rap_out(og + 147, hi.y);
        }
    } else {
        for (int g = 0; g < G; ++g) {
            const float4 *pl = s_plane + g * WS;
            f32x2 lo = {0.0f, 0.0f}, hi = {0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < NY * NX; ++q) pk_axpy4(lo, hi, wt[q], pl[off[q]]);
            float *og = o + (4 * g) * 49;
            rap_out(og, lo.x);
            if (4 * g + 1 < nc) rap_out(og + 49, lo.y);
            if (4 * g + 2 < nc) rap_out(og + 98, hi.x);
            if (4 * g + 3 < nc) rap_out(og + 147, hi.y);
        }
    }
}

// ---- tiles of a pyramid level (roi_tile.hip; the direct kernel of roi.hip takes the RoIs that fit none) ----
#ifndef RT_WIN
#define RT_WIN 48
#define RT_STEP 32
#endif
#ifndef RT_G
#define RT_G 4          // float4 channel groups per chunk: RT_WIN^2 * 16 B * RT_G = 147,456 B of LDS
#endif
#ifndef RT_LIST
#define RT_LIST 768     // RoIs of one tile gathered before its chunks are staged (20 B each)
#endif
#define RT_MAX_TILES 32 // per axis: owners travel as bit masks
#ifndef RT_WG_TARGET
#define RT_WG_TARGET 1280
#endif
static_assert(RT_WIN % 16 == 0 && RT_STEP <= RT_WIN, "tile layout");

// The two samples of bin p along one axis (torchvision roi_align, sampling_ratio = 2): the expression of the
// item loops of roi.hip, shared by the box scan and the items so that both see the same cells.
__device__ __forceinline__ void tile_bin_samples(float start, float bin, int p, int n, AxisSample (&s)[2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) s[i] = axis_sample(start + (float)p * bin + ((float)i + 0.5f) * bin / 2.0f, n);
}

// Base tile of a RoI along one axis: the tile that holds its first cell (roi_common.h roi_window's rule).
__device__ __forceinline__ int tile_base(float start, float bin, int n, int nw)
{
    AxisSample s[2];
    tile_bin_samples(start, bin, 0, n, s);
    const int lo = s[0].ok ? s[0].lo : 0;
    return min(lo / RT_STEP, nw - 1);
}

// Owner tile of a bin along one axis, from its two samples.  Cells that carry weight: [first, last]; samples
// outside the map carry none.  `fits` = the owner tile holds every such cell.
__device__ __forceinline__ int tile_owner(const AxisSample (&s)[2], int base, int nw, bool &fits)
{
    if (!s[0].ok && !s[1].ok) return base; // all-zero bin: written by the base tile
    const int first = s[0].ok ? s[0].lo : s[1].lo;
    const int last = s[1].ok ? s[1].hi : s[0].hi;
    if (last < base * RT_STEP + RT_WIN) return base; // first >= the RoI's first cell >= base * RT_STEP
    const int o = min(first / RT_STEP, nw - 1);
    // the last tile of an axis reaches the end of the map (host: (nw - 1) * RT_STEP + RT_WIN >= n)
    fits = fits && (o == nw - 1 || last < o * RT_STEP + RT_WIN);
    return o;
}

// Owners of the seven bins of one axis as a bit mask; `fits` as above over all of them.
__device__ __forceinline__ unsigned tile_axis_owners(float start, float bin, int n, int nw, int base, bool &fits)
{
    unsigned m = 0;
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        AxisSample s[2];
        tile_bin_samples(start, bin, p, n, s);
        m |= 1u << tile_owner(s, base, nw, fits);
    }
    return m;
}

// What the box scan of the tile kernel needs per axis: the RoI's base tile, the last tile that can own one of its
// bins (base .. last is a superset of tile_axis_owners' mask, and exact when the RoI lies inside its base tile), and
// `fits`.  A bin's weighted cells span at most bin / 2 + 3 cells, so only bins wider than 2 (RT_WIN - RT_STEP) - 4
// cells can fail to fit their owner: the seven bins are looked at one by one for those alone.
__device__ __forceinline__ void tile_axis_range(float start, float bin, int n, int nw, int &base, int &last_tile,
                                                bool &fits)
{
    AxisSample f[2], l[2];
    tile_bin_samples(start, bin, 0, n, f);
    tile_bin_samples(start, bin, 6, n, l);
    const int lo = f[0].ok ? f[0].lo : 0;
    const int hi = l[1].ok ? l[1].hi : n - 1;
    base = min(lo / RT_STEP, nw - 1);
    last_tile = hi < base * RT_STEP + RT_WIN ? base : min(hi / RT_STEP, nw - 1);
    if (bin > (float)(2 * (RT_WIN - RT_STEP) - 4)) tile_axis_owners(start, bin, n, nw, base, fits);
}

// Does the tile kernel pool this RoI?  (The direct kernel behind it takes the others: roi.hip.)
__device__ __forceinline__ bool tile_roi_fits(float4 rec, int H, int W, int nwy, int nwx)
{
    bool fits = true;
    tile_axis_owners(rec.y, rec.w, H, nwy, tile_base(rec.y, rec.w, H, nwy), fits);
    tile_axis_owners(rec.x, rec.z, W, nwx, tile_base(rec.x, rec.z, W, nwx), fits);
    return fits;
}

// nw tiles of RT_WIN cells, one every RT_STEP, the last one reaching the end of an axis of n cells
static inline int tile_count(int n) { return n <= RT_WIN ? 1 : (n - RT_WIN + RT_STEP - 1) / RT_STEP + 1; }

// roi.hip: the generic direct kernel on the RoIs of `level` that no tile pools (internal, not part of the C ABI)
int frcnn_roi_align_direct_behind_tiles(const float *feat, int B, int C, int H, int W, const float *props,
                                        const int32_t *count, int P, float spatial_scale, int level, int n_levels,
                                        int k_min, int k_max, float *out, void *stream);
