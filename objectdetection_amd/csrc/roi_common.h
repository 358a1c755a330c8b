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

