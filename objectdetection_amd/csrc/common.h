// common.h -- shared helpers for the gfx950 kernels of libfrcnn_hip.so.
// Built with -ffp-contract=off: every fp32 mul/add/sub/div below rounds separately, which is
// what bit-exact parity with the reference's torch ops needs (hard part 1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/frcnn_hip.h"

#define FRCNN_WAVE 64

void frcnn_set_error(const char *fmt, ...);

#define FRCNN_REQUIRE(cond, ...)                                                                    \
    do {                                                                                            \
        if (!(cond)) {                                                                              \
            frcnn_set_error(__VA_ARGS__);                                                           \
            return FRCNN_EINVAL;                                                                    \
        }                                                                                           \
    } while (0)

// Called after every launch: reports a bad launch configuration without synchronising.
#define FRCNN_CHECK_LAUNCH(name)                                                                    \
    do {                                                                                            \
        hipError_t e_ = hipGetLastError();                                                          \
        if (e_ != hipSuccess) {                                                                     \
            frcnn_set_error("%s: %s", name, hipGetErrorString(e_));                                 \
            return FRCNN_ELAUNCH;                                                                   \
        }                                                                                           \
    } while (0)

static inline size_t frcnn_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- device helpers -------------------------------------------------------------------------

__device__ __forceinline__ float frcnn_area(float x1, float y1, float x2, float y2)
{
    float w = x2 - x1;
    float h = y2 - y1;
    return w * h;
}

// utils.py:114-119 / torchvision nms: inter / ((area_a + area_b) - inter), 0/0 -> NaN.
__device__ __forceinline__ float frcnn_iou(float ax1, float ay1, float ax2, float ay2, float aarea,
                                           float bx1, float by1, float bx2, float by2, float barea)
{
    float ltx = ax1 > bx1 ? ax1 : bx1;
    float lty = ay1 > by1 ? ay1 : by1;
    float rbx = ax2 < bx2 ? ax2 : bx2;
    float rby = ay2 < by2 ? ay2 : by2;
    float jw = rbx - ltx;
    float jh = rby - lty;
    jw = jw < 0.0f ? 0.0f : jw;
    jh = jh < 0.0f ? 0.0f : jh;
    float inter = jw * jh;
    return inter / ((aarea + barea) - inter);
}

__device__ __forceinline__ float frcnn_vmax(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float frcnn_vmin(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Intersection and union terms of the IoU above, for callers that only need a threshold decision.
__device__ __forceinline__ void frcnn_inter_union(float ax1, float ay1, float ax2, float ay2,
                                                  float aarea, float bx1, float by1, float bx2,
                                                  float by2, float barea, float *inter, float *uni)
{
    // v_max_f32 / v_min_f32 (one instruction each instead of compare + select).  They drop a NaN operand
    // where torch.max / torch.min would propagate it, which cannot change a decision: a box with a NaN
    // coordinate has a NaN area, so `uni` is NaN and frcnn_iou_gt answers "not suppressed" either way.
    // The instruction itself rather than fmaxf / fminf: for an operand that was loaded, not computed, the
    // compiler puts a canonicalising v_max_f32 x, x in front of each (four more VALU operations per pair).
    float ltx = frcnn_vmax(ax1, bx1);
    float lty = frcnn_vmax(ay1, by1);
    float rbx = frcnn_vmin(ax2, bx2);
    float rby = frcnn_vmin(ay2, by2);
    float jw = __builtin_fmaxf(rbx - ltx, 0.0f);
    float jh = __builtin_fmaxf(rby - lty, 0.0f);
    *inter = jw * jh;
    *uni = (aarea + barea) - *inter;
}

// Threshold test of NMS, bit-identical to `RN(inter / uni) > t` (t = largest fp32 <= the double
// threshold) but without a division in the common case.
//   With t2 = the next fp32 above t, the test is RN(inter/uni) >= t2.  p = RN(t2 * uni) has a
//   relative error <= 2^-24, so when inter differs from p by more than 2^-20 |p| the exact quotient
//   is at least 4 ulps away from t2 on that side and its rounding cannot cross t2.  Only inside
//   that band (about one pair in 2^20), or when uni is not a positive normal number (degenerate or
//   NaN boxes) or the threshold itself is not a positive normal, is the IEEE division evaluated.
struct FrcnnThr {
    float t;    // largest fp32 <= threshold (double)
    float t2;   // next fp32 above t
    int fast;   // t2 is a positive normal number: the multiply test is valid
};

__device__ __forceinline__ bool frcnn_iou_gt(float inter, float uni, const FrcnnThr &th)
{
    const float p = th.t2 * uni;
    const float d = inter - p;
    // p = +inf or NaN (huge or non-finite union) fails the last comparison by itself: inf > inf and
    // NaN > x are false
    const bool sure = (p > 1e-30f) && (fabsf(d) > p * 9.5367431640625e-07f);
    bool r = d > 0.0f;
    // wave-uniform, taken about once per 2^14 wave tests.  (`th.fast` stays out of the per-lane mask: OR-ed
    // into it, the compiler rebuilds the ballot with two more VALU operations per pair.)
    if (__builtin_expect(!th.fast || __ballot(sure) != __builtin_amdgcn_read_exec(), 0)) {
        const bool slow = (inter / uni) > th.t;
        r = (th.fast && sure) ? r : slow;
    }
    return r;
}

// The same test for loops over many pairs, in two halves: the multiply decision, and -- AND-ed into `sure` over
// a GROUP of pairs -- whether it is safe.  A group with an unsafe pair (frcnn_group_unsure, wave-uniform, about
// one pair in 2^20) is redone with the division for every lane, which is the definition of the test; the
// per-pair cost is then the decision alone, and the mask bookkeeping (two VALU operations) is paid per group.
__device__ __forceinline__ bool frcnn_iou_gt_fast(float inter, float uni, const FrcnnThr &th, bool &sure)
{
    const float p = th.t2 * uni;
    const float d = inter - p;
    sure = sure && (p > 1e-30f) && (fabsf(d) > p * 9.5367431640625e-07f);
    return d > 0.0f;
}

__device__ __forceinline__ bool frcnn_iou_gt_exact(float inter, float uni, const FrcnnThr &th)
{
    return (inter / uni) > th.t;
}

__device__ __forceinline__ bool frcnn_group_unsure(bool sure, const FrcnnThr &th)
{
    return !th.fast || __ballot(sure) != __builtin_amdgcn_read_exec();
}

// Kernel attributes are per device: remember, per device, the largest dynamic-LDS size already granted to a
// kernel.  `granted` is a static array of the call site, one per kernel.
#define FRCNN_MAX_DEVICES 64
static inline hipError_t frcnn_reserve_dynamic_lds(const void *kernel, size_t bytes, size_t (&granted)[FRCNN_MAX_DEVICES])
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool tracked = dev >= 0 && dev < FRCNN_MAX_DEVICES;
    if (tracked && bytes <= granted[dev]) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && tracked) granted[dev] = bytes;
    return e;
}

// Order-preserving unsigned key of an fp32 score: larger key = ranks first.
// NaN -> 0xFFFFFFFF (above +inf, as torch.topk treats NaN), -0.0 below +0.0.
__device__ __forceinline__ uint32_t frcnn_score_key(float s)
{
    uint32_t u = __float_as_uint(s);
    if (s != s) return 0xFFFFFFFFu;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// The score a key was made from (bit-exact, except that every NaN comes back as one quiet NaN).
__device__ __forceinline__ float frcnn_key_score(uint32_t key)
{
    return __uint_as_float((key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
}

// Exclusive prefix count of `flag` over a 1024-thread (16-wave) block; returns the block total
// through *total.  `wave_sums` is 16 ints of LDS.  Two barriers.
__device__ __forceinline__ int frcnn_block_excl_count(bool flag, int *wave_sums, int *total)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwave = blockDim.x >> 6;
    unsigned long long m = __ballot(flag);
    int in_wave = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_sums[wave] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nwave; ++w) {
        int s = wave_sums[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + in_wave;
}

// ---- the RPN head's outputs read in place ------------------------------------------------------------
// RPN/rpn.py:56-57 flattens the per-level (B, na, h, w) scores and (B, 4 na, h, w) deltas to one vector per
// image, flat index e = level offset + cell * na + a.  Kernels that take a level table read element e
// where it lies instead of from a flattened copy.  The table travels by value (kernel argument); the level
// of e is found by compare-and-select over statically indexed entries so that it stays in scalar
// registers.  Unused levels start at A, so no index selects them.
#define FRCNN_MAX_LEVELS 6
struct RpnLevels {
    const float *cls[FRCNN_MAX_LEVELS]; // (B, na, h, w)
    const float *reg[FRCNN_MAX_LEVELS]; // (B, 4 na, h, w)
    int hw[FRCNN_MAX_LEVELS];
    int off[FRCNN_MAX_LEVELS]; // first flat index of each level (A for unused levels)
    int na;
};

struct RpnElem { // where flat element e lives
    const float *cls, *reg;
    int hw, a, cell;
};

__device__ __forceinline__ RpnElem rpn_locate(const RpnLevels &L, int e)
{
    RpnElem r;
    r.cls = L.cls[0];
    r.reg = L.reg[0];
    r.hw = L.hw[0];
    int off = 0;
#pragma unroll
    for (int l = 1; l < FRCNN_MAX_LEVELS; ++l) {
        const bool in = e >= L.off[l];
        r.cls = in ? L.cls[l] : r.cls;
        r.reg = in ? L.reg[l] : r.reg;
        r.hw = in ? L.hw[l] : r.hw;
        off = in ? L.off[l] : off;
    }
    const int rem = e - off;
    r.cell = rem / L.na;
    r.a = rem - r.cell * L.na;
    return r;
}

__device__ __forceinline__ float rpn_score(const RpnLevels &L, int b, int e)
{
    const RpnElem r = rpn_locate(L, e);
    return r.cls[((long long)b * L.na + r.a) * r.hw + r.cell];
}

__device__ __forceinline__ float4 rpn_delta(const RpnLevels &L, int b, int e)
{
    const RpnElem r = rpn_locate(L, e);
    const float *p = r.reg + ((long long)b * 4 * L.na + 4 * r.a) * r.hw + r.cell;
    return make_float4(p[0], p[r.hw], p[2 * r.hw], p[3 * r.hw]);
}

// Fills a table from host arrays of per-level device pointers and (h, w) pairs; returns A or -1.
static inline long long frcnn_make_levels(RpnLevels *L, const float *const *cls_host, const float *const *reg_host,
                                          const int32_t *hw_host, int n_levels, int na)
{
    if (n_levels <= 0 || n_levels > FRCNN_MAX_LEVELS || na <= 0 || !hw_host) return -1;
    long long A = 0;
    L->na = na;
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
        const bool used = l < n_levels;
        if (used && (hw_host[2 * l] <= 0 || hw_host[2 * l + 1] <= 0)) return -1;
        L->cls[l] = used && cls_host ? cls_host[l] : nullptr;
        L->reg[l] = used && reg_host ? reg_host[l] : nullptr;
        L->hw[l] = used ? hw_host[2 * l] * hw_host[2 * l + 1] : 0;
        L->off[l] = (int)A;
        if (used) A += (long long)L->hw[l] * na;
        if (A >= (1ll << 31)) return -1;
    }
    for (int l = n_levels; l < FRCNN_MAX_LEVELS; ++l) L->off[l] = (int)A;
    return A;
}
