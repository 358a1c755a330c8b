// roi_cl.hip -- K6 for feature pyramids with lanes = channels: MultiScaleRoIAlign on channel-innermost maps.
// Replaces torchvision.ops.MultiScaleRoIAlign as used by ROIHead/roi_pooling.py:44-53 (call site
// ROIHead/roi_head.py:127-130) on the FPN of BackBone/fpn.py:91-96.
//
// The plane / window / tile kernels (roi.hip, roi_tile.hip) stage NCHW cells in LDS and give a lane one BIN;
// on planes beyond LDS the staging cannot overlap the items.  Here a lane is a CHANNEL and no window is staged:
//   out[c, ph, pw] = sum_y Wy[ph, y] * sum_x Wx[pw, x] * F[y, x, c]      (RoIAlign's bilinear weights are separable)
// On a map laid out (B, H, W, C) a wave takes (RoI, 64 channels): every weight and every cell address is wave-uniform
// (scalar registers), the 49 sums of a channel live in the lane's registers, and the RoI's 64 x 49 block leaves through
// a wave-private LDS transposition as one contiguous 12.5 KB run of 16-byte stores.  Rows are walked once: the 14 row
// samples of a RoI ascend, so each cell row the samples touch is fetched and contracted along x exactly once (into 7
// registers) whatever the number of bins that use it, and added to the bin rows in which it has weight.
// A row arrives in one of two ways: when the x taps span at most 16 cells, as those 16 cells by four
// `buffer_load_dwordx4 ... lds` into a wave-private two-slot LDS ring, from which the 21-28 taps are ds_read_b32 (lane =
// channel: no bank conflicts; every cell fetched once although neighbouring bins share cells); otherwise as 21-28
// coalesced 256-byte loads into registers (no RoI size limit).
// Work order: a preparation launch sorts every image's RoIs by (level, band of rows) -- a band's 64-channel slice
// is at most ~1.5 MB -- and the main launch hands consecutive workgroups of one XCD consecutive RoIs of one
// (image, channel group), so that the cells a band's RoIs share are served by that XCD's L2.
// NCHW maps (what the reference's FPN produces) are first copied to a channel-innermost layout blocked by 64 channels
// in the caller's workspace (one pass); maps that already are channels-last are read in place.
// Measurements, what bounds the kernel and what was tried and dropped: DESIGN.md section 4.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "roi_common.h"

#ifndef RCL_THREADS
#define RCL_THREADS 64 // one wave per workgroup: a wave that has finished its RoIs frees its registers and LDS at once
#endif
#define RCL_CH 64 // channels per wave item: a lane is a channel
#define RCL_WAVES (RCL_THREADS / 64)
// RoIs per workgroup (<= 64: a chunk's codes and boxes travel one per lane).  One: vector memory operations retire in
// order, so the first row of a wave's next RoI would wait for the 14 stores of the one before it (config 5: 0.761 /
// 0.772 / 0.794 / 0.820 / 0.996 ms at 1 / 2 / 3 / 4 / 8 RoIs per wave).
#define RCL_CHUNK_DEFAULT 1
#define RCL_MAX_UNITS 16
#ifndef RCL_BAND_BYTES
#define RCL_BAND_BYTES (1536 * 1024)
#endif
#define RCL_ZERO_LV 7 // level code of a RoI that no level takes (NaN area): its rows are zeros
#define RCL_PREP_THREADS 1024
#ifndef RCL_WAVES_PER_EU
#define RCL_WAVES_PER_EU 4 // 128 registers per lane
#endif

struct ClLevel {
    const float *feat; // (B, H, W, C)
    int H, W;
    float scale;
    int unit0, nbands, band_rows;
};
struct ClLevels {
    ClLevel l[FRCNN_MAX_LEVELS];
    int n;
    int units; // bands over all levels + 1 (the last unit collects the RoIs no level takes)
};

// ---- NCHW -> (B, H, W, C) ---------------------------------------------------------------------------------------
struct ClCopy {
    const float *src[FRCNN_MAX_LEVELS];
    float *dst[FRCNN_MAX_LEVELS];
    int hw[FRCNN_MAX_LEVELS];
    int tile0[FRCNN_MAX_LEVELS + 1]; // first 64-pixel tile of each level in blockIdx.x order
    int n;
};

__device__ __forceinline__ void cl_copy_tile(const ClCopy &T, int C, int tile, int cg, int b)
{
    __shared__ float s_t[64][65];
    const float *src = T.src[0];
    float *dst = T.dst[0];
    int HW = T.hw[0], t0 = 0;
#pragma unroll
    for (int l = 1; l < FRCNN_MAX_LEVELS; ++l) {
        const bool in = l < T.n && tile >= T.tile0[l];
        src = in ? T.src[l] : src;
        dst = in ? T.dst[l] : dst;
        HW = in ? T.hw[l] : HW;
        t0 = in ? T.tile0[l] : t0;
    }
    const int px0 = (tile - t0) * 64, c0 = cg * 64;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const float *s = src + ((long long)b * C + c0) * HW + px0;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int ch = k * 4 + q;
        v[k] = (px0 + lane < HW && c0 + ch < C) ? s[(long long)ch * HW + lane] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) s_t[k * 4 + q][lane] = v[k];
    __syncthreads();
    // destination: (B, C / RCL_CH, H * W, RCL_CH) -- the slice of an image that a wave item reads is one contiguous block
    float *d = dst + (((long long)b * (C / RCL_CH) + cg) * HW + px0) * RCL_CH;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int px = k * 4 + q;
        if (px0 + px < HW) d[(long long)px * RCL_CH + lane] = s_t[lane][px];
    }
}

// ---- preparation: every image's RoIs sorted by (level, band), one record per RoI -------------------------------
// Records live at PADDED positions: every unit starts on a multiple of the chunk size, so that chunk c of an image is
// the records [c * chunk, (c + 1) * chunk) and a workgroup of the main kernel finds its work with one load.
struct __align__(16) ClRec {
    float4 box;
    int code;    // RoI index within the image | level << 24; -1 = padding
    int out_row; // row of the output
    int pad0, pad1;
};

__device__ __forceinline__ int cl_unit_of(const ClLevels &L, float4 box, int k_min, int k_max, int &lv_out)
{
    int lv = 0;
    if (L.n > 1) lv = map_level(box, k_min, k_max);
    if (lv < 0 || lv >= L.n) {
        lv_out = RCL_ZERO_LV;
        return L.units - 1;
    }
    int H = L.l[0].H, unit0 = L.l[0].unit0, nbands = L.l[0].nbands, band_rows = L.l[0].band_rows;
    float scale = L.l[0].scale;
#pragma unroll
    for (int l = 1; l < FRCNN_MAX_LEVELS; ++l) {
        const bool in = l == lv;
        H = in ? L.l[l].H : H;
        unit0 = in ? L.l[l].unit0 : unit0;
        nbands = in ? L.l[l].nbands : nbands;
        band_rows = in ? L.l[l].band_rows : band_rows;
        scale = in ? L.l[l].scale : scale;
    }
    const float4 rec = roi_record(box, scale);
    float cy = rec.y + 3.5f * rec.w;
    cy = fminf(fmaxf(cy, 0.0f), (float)(H - 1)); // NaN -> 0
    const int band = min((int)cy / band_rows, nbands - 1);
    lv_out = lv;
    return unit0 + band;
}

__global__ void __launch_bounds__(RCL_PREP_THREADS)
roi_cl_sort_kernel(ClLevels L, const float4 *__restrict__ props, const int32_t *__restrict__ count, int P, int B,
                   int k_min, int k_max, int chunk, int padded, ClRec *__restrict__ recs,
                   int32_t *__restrict__ total_out)
{
    __shared__ int s_hist[RCL_MAX_UNITS], s_base[RCL_MAX_UNITS], s_cnt, s_off;
    const int b = blockIdx.x;
    if (threadIdx.x < RCL_MAX_UNITS) s_hist[threadIdx.x] = 0;
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
        if (total_out && b == 0) *total_out = tot;
    }
    __syncthreads();
    const int cnt = s_cnt, off = s_off;
    const float4 *pb = props + (long long)b * P;
    ClRec *rb = recs + (long long)b * padded;
    for (int q = threadIdx.x; q < padded; q += blockDim.x) rb[q].code = -1;
    for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
        int lv;
        atomicAdd(&s_hist[cl_unit_of(L, pb[j], k_min, k_max, lv)], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int u = 0; u < RCL_MAX_UNITS; ++u) {
            s_base[u] = run;
            run += (s_hist[u] + chunk - 1) / chunk * chunk;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
        int lv;
        const float4 box = pb[j];
        const int u = cl_unit_of(L, box, k_min, k_max, lv);
        const int pos = atomicAdd(&s_base[u], 1);
        ClRec r;
        r.box = box;
        r.code = j | (lv << 24);
        r.out_row = off + j;
        r.pad0 = r.pad1 = 0;
        rb[pos] = r;
    }
}

__device__ __forceinline__ float rcl_readlane(float v, int lane)
{
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}

// ---- per-RoI set-up, by the lanes of the wave in parallel ---------------------------------------------------------
// Lanes 0..27 take the 7 x 4 x taps (weight, cell), lanes 32..45 the 14 row samples; out come the ascending list of
// the <= 28 rows the samples touch (one row per lane) and the weight of each list row in each of the 7 bin rows.
// (Computing these once per RoI in a launch of its own and loading them here -- 1,152 bytes per RoI, x taps by scalar
// loads -- was built and measured: 0.986 against 0.944 ms at config 5; the table's load latency at the head of every
// item costs more than the ~450 vector instructions it replaces.)
#ifndef RCL_RING_CELLS
#define RCL_RING_CELLS 16
#endif
struct ClSetup {
    int nrows;   // rows in the list (wave-uniform)
    bool x3;     // every bin's two x samples merge into three taps (wave-uniform)
    int c0, c1;  // first and last cell the x taps read (wave-uniform)
    int v_cell;  // lane j < 28: the cell of tap j = (bin j / 4, tap j % 4)
    float v_xw;  //              and its weight
    int v_rows;  // lane k < nrows: row k of the list
    float v_wy[7]; //            and its weight in every bin row (zeros from lane nrows on)
};

__device__ __forceinline__ ClSetup rcl_setup(float4 box, float scale, int H, int W, int lane)
{
    ClSetup S;
    const float4 rec = roi_record(box, scale);
    const bool is_y = lane >= 32;
    const int idx = lane & 31;
    const int p = min(is_y ? idx >> 1 : idx >> 2, 6);
    AxisSample sm[2];
    tile_bin_samples(is_y ? rec.y : rec.x, is_y ? rec.w : rec.z, p, is_y ? H : W, sm);
    const bool x3 = (__ballot(axis_mergeable(sm[0], sm[1]) || is_y || idx >= 28) == ~0ull);
    const AxisTaps tx = x3 ? axis_taps3(sm[0], sm[1], 1.0f, W) : axis_taps4(sm[0], sm[1], 1.0f);
    const AxisSample sy = (idx & 1) ? sm[1] : sm[0];
    // Row samples (lanes 32..45): the sample's last row b (its high row, or the clamped row H - 1 alone) with
    // weight wh, and the row above it with weight wl when it takes part.
    const bool valid = is_y && idx < 14 && sy.ok;
    const bool use_lo = sy.hi != sy.lo;
    const int b_row = valid ? sy.hi : -8;
    const float s_wl = (valid && use_lo) ? sy.wlo * 0.25f : 0.0f;
    const float s_wh = valid ? (use_lo ? sy.whi : sy.wlo) * 0.25f : 0.0f;
    // the ascending list of rows the samples touch: a sample appends the rows the one before it did not
    const int prev_b = __shfl_up(b_row, 1);
    const int pb = idx > 0 ? prev_b : -8; // (the valid samples are consecutive lanes)
    const bool new_a = valid && use_lo && b_row - 1 > pb;
    const bool new_b = valid && b_row > pb;
    const unsigned long long mask_a = __ballot(new_a), mask_b = __ballot(new_b);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int pos_a = __popcll(mask_a & below) + __popcll(mask_b & below);
    const int pos_b = pos_a + (new_a ? 1 : 0);
#ifdef RCL_EXP_NOROWS
    S.nrows = 0 * (__popcll(mask_a) + __popcll(mask_b)); // experiment: set-up and epilogue alone
#else
    S.nrows = __popcll(mask_a) + __popcll(mask_b);
#endif
    const int ra = __builtin_amdgcn_ds_permute((new_a ? pos_a : 63) * 4, b_row); // (row + 1) of b_row - 1
    const int rb = __builtin_amdgcn_ds_permute((new_b ? pos_b : 63) * 4, b_row + 1);
#ifdef RCL_EXP_ONEROW
    S.v_rows = 0 * max(max(ra, rb) - 1, 0); // experiment: every row fetch hits the same cells
#else
    S.v_rows = max(max(ra, rb) - 1, 0);
#endif
#pragma unroll
    for (int ph = 0; ph < 7; ++ph) S.v_wy[ph] = 0.0f;
#pragma unroll
    for (int sidx = 0; sidx < 14; ++sidx) {
        const int sb = __builtin_amdgcn_readlane(b_row, 32 + sidx);
        const float wl = rcl_readlane(s_wl, 32 + sidx), wh = rcl_readlane(s_wh, 32 + sidx);
        const float w = (S.v_rows == sb ? wh : 0.0f) + (S.v_rows == sb - 1 ? wl : 0.0f);
        S.v_wy[sidx >> 1] += lane < S.nrows ? w : 0.0f;
    }
    const int t = idx & 3;
    S.v_cell = t == 0 ? tx.idx[0] : t == 1 ? tx.idx[1] : t == 2 ? tx.idx[2] : tx.idx[3];
    S.v_xw = t == 0 ? tx.w[0] : t == 1 ? tx.w[1] : t == 2 ? tx.w[2] : tx.w[3];
    // the span of the cells the x taps read (a merged bin's fourth tap is not read)
    const bool tap_read = lane < 28 && (t < 3 || !x3);
    int cmin = tap_read ? S.v_cell : 0x7fffffff, cmax = tap_read ? S.v_cell : -1;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        cmin = min(cmin, __shfl_xor(cmin, d));
        cmax = max(cmax, __shfl_xor(cmax, d));
    }
    S.c0 = __builtin_amdgcn_readfirstlane(cmin);
    S.c1 = __builtin_amdgcn_readfirstlane(cmax);
    S.x3 = x3;
    return S;
}

// One 64-channel x 64-pixel tile per workgroup.
__global__ void __launch_bounds__(256)
nchw_to_cl_kernel(ClCopy T, int C, int tiles)
{
    cl_copy_tile(T, C, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- the main kernel -------------------------------------------------------------------------------------------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float rcl_vec;

// A lane's channels of one cell, one dword load of 256 contiguous bytes per 64 channels.  (An 8-byte load per lane
// moves the same bytes through the vector cache at half the rate: 17 cache accesses per 512-byte wave load against 4
// per 256-byte one, rocprofv3 TCP_TOTAL_CACHE_ACCESSES.)
__device__ __forceinline__ rcl_vec rcl_loadv(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff)
{
#ifdef RCL_EXP_NOLOAD
    return rcl_vec(__uint_as_float(voff + soff)); // experiment: arithmetic and stores without the loads
#else
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
#endif
}

typedef float rcl_f32x4 __attribute__((ext_vector_type(4)));
// The output is written once and never read here: `nt` stores (global_store_dwordx4 ... nt), 0.887 against 0.911 ms at
// config 5 with plain ones; `sc1` / `sc0 sc1` (write-through, line dropped from L2) measured like plain.
__device__ __forceinline__ void rcl_store(float4 *p, float4 v)
{
    const rcl_f32x4 q = {v.x, v.y, v.z, v.w};
#ifdef RCL_EXP_NOSTORE
    if (v.x == 1234.5f) *reinterpret_cast<rcl_f32x4 *>(p) = q; // experiment: everything but the store stream
#elif defined(RCL_EXP_STORE_PLAIN)
    *reinterpret_cast<rcl_f32x4 *>(p) = q;
#else
    __builtin_nontemporal_store(q, reinterpret_cast<rcl_f32x4 *>(p));
#endif
}

// acc += w * t and w * t with a wave-uniform weight (the LOW word of an SGPR pair), IN PLACE: written as instructions
// because the compiler, given the same expressions, gives every result a new register and copies it back at each
// join of the guarded blocks below.
__device__ __forceinline__ void rcl_fma(rcl_vec &acc, uint64_t w, rcl_vec t)
{
    const uint32_t ws = (uint32_t)w;
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "s"(ws), "v"(t));
}

__device__ __forceinline__ rcl_vec rcl_mul(uint64_t w, rcl_vec t)
{
    const uint32_t ws = (uint32_t)w;
    rcl_vec r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "s"(ws), "v"(t));
    return r;
}

// One cell row: the 7 x NX loads of its taps (lane = two channels), and their contraction along x,
// T[pw] = sum_t w[pw][t] * F[row][cell(pw, t)][channels].  wx[j] = tap j's weight (low word) and the byte offset of
// its cell inside a row (high word): one SGPR pair serves the packed multiply and the address.
template <int NX>
__device__ __forceinline__ void rcl_issue(rcl_vec (&v)[7 * NX], __amdgpu_buffer_rsrc_t rsrc, int lane8, int row_off,
                                          const uint64_t (&wx)[28])
{
#pragma unroll
    for (int p = 0; p < 7; ++p)
#pragma unroll
        for (int t = 0; t < NX; ++t) v[p * NX + t] = rcl_loadv(rsrc, lane8, row_off + (int)(wx[p * 4 + t] >> 32));
}

template <int NX>
__device__ __forceinline__ void rcl_reduce(rcl_vec (&T)[7], const rcl_vec (&v)[7 * NX], const uint64_t (&wx)[28])
{
    // tap-major: seven independent chains side by side, no instruction waits for the one before it
#pragma unroll
    for (int p = 0; p < 7; ++p) T[p] = rcl_mul(wx[p * 4], v[p * NX]);
#pragma unroll
    for (int t = 1; t < NX; ++t)
#pragma unroll
        for (int p = 0; p < 7; ++p) rcl_fma(T[p], wx[p * 4 + t], v[p * NX + t]);
}

// One bin row: acc[0..6] += w * T[0..6] unless w is +-0 (a row never multiplies a zero: 0 x Inf stays out of bins
// the row has no part in).  The test, the branch around the sums and the sums are ONE asm statement: seen as a
// branch by the compiler, every guarded block ends in a join at which it keeps the old and the new sums in two sets
// of registers and copies between them (49 moves per row and 60 registers more, measured).
__device__ __forceinline__ void rcl_add_row(rcl_vec *a, uint32_t w, const rcl_vec (&T)[7])
{
    uint32_t tmp;
    asm("s_and_b32 %7, %8, 0x7fffffff\n\t"
        "s_cbranch_scc0 .Lrcl_skip%=\n\t"
        "v_fmac_f32 %0, %8, %9\n\tv_fmac_f32 %1, %8, %10\n\tv_fmac_f32 %2, %8, %11\n\tv_fmac_f32 %3, %8, %12\n\t"
        "v_fmac_f32 %4, %8, %13\n\tv_fmac_f32 %5, %8, %14\n\tv_fmac_f32 %6, %8, %15\n"
        ".Lrcl_skip%=:"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "=&s"(tmp)
        : "s"(w), "v"(T[0]), "v"(T[1]), "v"(T[2]), "v"(T[3]), "v"(T[4]), "v"(T[5]), "v"(T[6])
        : "scc");
}

#ifndef RCL_DEPTH
#define RCL_DEPTH 1
#endif
#define RCL_OOB 0x40000000 // added to a lane's byte offset: beyond every map, so the buffer load returns 0 unread

// One row of the list: contract the buffer that holds it, refill the buffer with the row two further on, add the
// row's sums to the bin rows in which it has weight.
template <int NX>
__device__ __forceinline__ void rcl_row(rcl_vec (&acc)[49], rcl_vec (&Lb)[7 * NX], __amdgpu_buffer_rsrc_t rsrc, int lane8,
                                        int row_bytes, const uint64_t (&wx)[28], int v_rows, const float (&v_wy)[7],
                                        int k, int nrows)
{
    uint32_t wy[7];
#pragma unroll
    for (int ph = 0; ph < 7; ++ph) wy[ph] = __builtin_amdgcn_readlane(__float_as_uint(v_wy[ph]), min(k, 63));
    rcl_vec T[7];
    rcl_reduce<NX>(T, Lb, wx);
    const int kn = k + RCL_DEPTH;
    rcl_issue<NX>(Lb, rsrc, lane8 + (kn < nrows ? 0 : RCL_OOB),
                  __builtin_amdgcn_readlane(v_rows, min(kn, 63)) * row_bytes, wx);
#pragma unroll
    for (int ph = 0; ph < 7; ++ph) rcl_add_row(&acc[ph * 7], wy[ph], T);
}

template <int NX>
__device__ __forceinline__ void rcl_item(rcl_vec (&acc)[49], __amdgpu_buffer_rsrc_t rsrc, int lane8, int row_bytes,
                                         const uint64_t (&wx)[28], int v_rows, const float (&v_wy)[7], int nrows)
{
    // RCL_DEPTH rows in flight: a row's loads are issued that many rows ahead, always (past the list they go out of
    // range and return zeros unread), so that the number of loads behind the one a contraction waits for is known
    // statically.  Rows in pairs with one exit: behind an odd list the second row of the last pair is a null row (its
    // loads were out of range, its weights are zeros, every bin row skips it).
    rcl_vec LA[7 * NX];
    rcl_issue<NX>(LA, rsrc, lane8, __builtin_amdgcn_readlane(v_rows, 0) * row_bytes, wx);
#if RCL_DEPTH == 2
    rcl_vec LB[7 * NX];
    rcl_issue<NX>(LB, rsrc, lane8 + (1 < nrows ? 0 : RCL_OOB), __builtin_amdgcn_readlane(v_rows, 1) * row_bytes, wx);
    for (int k = 0; k < nrows; k += 2) {
        rcl_row<NX>(acc, LA, rsrc, lane8, row_bytes, wx, v_rows, v_wy, k, nrows);
        rcl_row<NX>(acc, LB, rsrc, lane8, row_bytes, wx, v_rows, v_wy, k + 1, nrows);
    }
#else
    for (int k = 0; k < nrows; ++k) rcl_row<NX>(acc, LA, rsrc, lane8, row_bytes, wx, v_rows, v_wy, k, nrows);
#endif
}

// ---- rows through LDS: 16-byte fills, every cell of a row fetched once ------------------------------------------
// The vector cache serves a wave's dword loads at ~35 bytes per clock and CU and its dwordx4 loads at ~72
// (tools/ubench/vmem_valu_overlap.hip), and the taps of neighbouring bins overlap (21-28 taps on 9-16 distinct cells).
// When the cells a RoI touches in a row span at most RCL_RING_CELLS, the row is therefore fetched as that many
// CONTIGUOUS cells by four `buffer_load_dwordx4 ... lds` (LDS-DMA: 1 KiB = four 64-channel cells of the blocked copy per
// instruction, no registers) into a wave-private two-slot ring, and the 21-28 taps are ds_read_b32 from it (lane =
// channel).  The fills of row k + 1 are in flight while row k is contracted; `s_waitcnt vmcnt(4)` is placed by hand
// (fills, like every vector memory operation, retire in order).  68 cycles per row and CU in tools/ubench/lds_dma.hip
// against 21 x 7.7 = 162 for the register loads.
#define RCL_SLOT_FLOATS (RCL_RING_CELLS * 64)
typedef __attribute__((address_space(3))) void rcl_lds_void;

// One fill = four cells: lane l brings bytes [16 (l % 16), +16) of cell l / 16 (vlane = that cell's and piece's byte
// offset: cells are cell_bytes apart, 256 in the blocked copy where a fill is one contiguous KiB); in LDS the four
// cells are contiguous whatever the layout.  A ring row is always 16 cells from the first tap's cell on, which can run
// past the end of the map's row: inside the image those are the next row's first cells (fetched, never read by a
// tap); past the end of the image's slice -- at the last row of the last image, past the end of the tensor -- the
// buffer's range check, which on gfx950 covers per-lane offset + scalar offset (tools/ubench/buffer_range.hip), returns
// zeros without touching memory: the descriptor's size is exactly the rest of the image's slice.
__device__ __forceinline__ void rcl_fill_row(__amdgpu_buffer_rsrc_t rsrc, float *slot, int vlane, int off, int cell_bytes)
{
#pragma unroll
    for (int d = 0; d < RCL_RING_CELLS / 4; ++d)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (rcl_lds_void *)(slot + d * 256), 16, vlane, off + d * 4 * cell_bytes, 0, 0);
}

template <int NX>
__device__ __forceinline__ void rcl_ring_row(float (&acc)[49], const float *slot, float *refill, __amdgpu_buffer_rsrc_t rsrc,
                                             int vlane, int row_bytes, int cell_bytes, int c0_bytes, const int (&vaddr)[7 * NX],
                                             const uint64_t (&wx)[28], int v_rows, const float (&v_wy)[7], int k, int nrows)
{
    uint32_t wy[7];
#pragma unroll
    for (int ph = 0; ph < 7; ++ph) wy[ph] = __builtin_amdgcn_readlane(__float_as_uint(v_wy[ph]), min(k, 63));
    // this row's four fills have landed once at most the four of the next row are outstanding
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    float v[7 * NX];
#pragma unroll
    for (int j = 0; j < 7 * NX; ++j) v[j] = *(const float *)((const char *)slot + vaddr[j]);
    float T[7];
#pragma unroll
    for (int p = 0; p < 7; ++p) T[p] = rcl_mul(wx[p * 4], v[p * NX]);
#pragma unroll
    for (int t = 1; t < NX; ++t)
#pragma unroll
        for (int p = 0; p < 7; ++p) rcl_fma(T[p], wx[p * 4 + t], v[p * NX + t]);
    // the taps are in registers (T depends on them; the explicit wait makes that independent of how the compiler
    // orders the contraction against the fills): the slot may be refilled with the row two further on
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int kn = k + 2;
    rcl_fill_row(rsrc, refill, vlane + (kn < nrows ? 0 : RCL_OOB),
                 __builtin_amdgcn_readlane(v_rows, min(kn, 63)) * row_bytes + c0_bytes, cell_bytes);
#pragma unroll
    for (int ph = 0; ph < 7; ++ph) rcl_add_row(&acc[ph * 7], wy[ph], T);
}

template <int NX>
__device__ __forceinline__ void rcl_item_ring(float (&acc)[49], __amdgpu_buffer_rsrc_t rsrc, int lane, int row_bytes,
                                              int cell_bytes, int c0, const uint64_t (&wx)[28], int v_rows,
                                              const float (&v_wy)[7], int nrows, float *ring)
{
    // here the high word of wx[j] is 256 x (tap j's cell - c0): where the cell sits in a ring slot
    int vaddr[7 * NX];
    // (Computing them per row from the scalars instead -- one v_add per tap -- frees 24 registers, but the register
    // path beside this one needs them anyway: 120 registers either way, and forced to 96 for five waves per SIMD
    // the kernel spills and runs 1.00 against 0.77 ms.  With the register path compiled out as well (every RoI through
    // the ring: wrong for the 31 % of config 5's RoIs that are wider than it, timing only), taps read in rounds of seven
    // and nothing spilled inside the row loop, five waves per SIMD run 0.833 ms against 0.746 for four: more waves
    // are not what this kernel lacks, and the ring for every RoI would be worth 2.4 %.)
#pragma unroll
    for (int p = 0; p < 7; ++p)
#pragma unroll
        for (int t = 0; t < NX; ++t) vaddr[p * NX + t] = (int)(wx[p * 4 + t] >> 32) + lane * 4;
    const int vlane = (lane >> 4) * cell_bytes + (lane & 15) * 16;
    const int c0_bytes = c0 * cell_bytes;
    float *s0 = ring, *s1 = ring + RCL_SLOT_FLOATS;
    rcl_fill_row(rsrc, s0, vlane, __builtin_amdgcn_readlane(v_rows, 0) * row_bytes + c0_bytes, cell_bytes);
    rcl_fill_row(rsrc, s1, vlane + (1 < nrows ? 0 : RCL_OOB), __builtin_amdgcn_readlane(v_rows, 1) * row_bytes + c0_bytes,
                 cell_bytes);
    for (int k = 0; k < nrows; k += 2) {
        rcl_ring_row<NX>(acc, s0, s0, rsrc, vlane, row_bytes, cell_bytes, c0_bytes, vaddr, wx, v_rows, v_wy, k, nrows);
        rcl_ring_row<NX>(acc, s1, s1, rsrc, vlane, row_bytes, cell_bytes, c0_bytes, vaddr, wx, v_rows, v_wy, k + 1, nrows);
    }
    // the (out-of-range) fills issued behind the list still write the ring: it is reused by the epilogue
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void __launch_bounds__(RCL_THREADS) __attribute__((amdgpu_waves_per_eu(RCL_WAVES_PER_EU, RCL_WAVES_PER_EU)))
roi_align_cl_kernel(ClLevels L, int C, int cgroups, int blocked, int chunks, int chunk_rois, int padded,
                    const ClRec *__restrict__ recs, int B, float *__restrict__ out)
{
    __shared__ __attribute__((aligned(1024))) float s_out[RCL_WAVES][2 * RCL_SLOT_FLOATS]; // row ring; the epilogue reuses it
    static_assert(2 * RCL_SLOT_FLOATS >= 32 * 49, "the epilogue's 32 x 49 floats fit the ring");
    // Blocks i and i + 8 share an XCD (observed dispatch order; speed only).  An XCD takes whole (image, channel
    // group) pairs, one after the other, and inside a pair the chunks in unit order.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int su = (slot / chunks) * 8 + xcd;
    if (su >= B * cgroups) return;
    const int chunk = slot % chunks;
    const int b = su / cgroups, g = su - b * cgroups;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane8 = lane * 4; // byte offset of the lane's first channel inside a cell
    // the chunk's records, one per lane
    ClRec r;
    r.code = -1;
    r.out_row = 0;
    r.box = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (lane < chunk_rois) r = recs[(long long)b * padded + chunk * chunk_rois + lane];
    const int n = __popcll(__ballot(r.code >= 0)); // the valid records are the chunk's first ones
    float *sw = s_out[wave];
    for (int i = wave; i < n; i += RCL_WAVES) {
        const int out_row = __builtin_amdgcn_readlane(r.out_row, i);
        float *o = out + ((long long)out_row * C + g * RCL_CH) * 49;
        const int lv = __builtin_amdgcn_readlane(r.code, i) >> 24;
        const float4 box = make_float4(rcl_readlane(r.box.x, i), rcl_readlane(r.box.y, i), rcl_readlane(r.box.z, i),
                                       rcl_readlane(r.box.w, i));
        // the sums start at zero here and are only ever updated in place (rcl_fma): one set of registers on every path
        rcl_vec acc[49];
#pragma unroll
        for (int k = 0; k < 49; ++k) acc[k] = rcl_vec(0.0f);
        if (lv != RCL_ZERO_LV) {
            const float *feat = L.l[0].feat;
            int H = L.l[0].H, W = L.l[0].W;
            float scale = L.l[0].scale;
#pragma unroll
            for (int l = 1; l < FRCNN_MAX_LEVELS; ++l) {
                const bool in = l == lv;
                feat = in ? L.l[l].feat : feat;
                H = in ? L.l[l].H : H;
                W = in ? L.l[l].W : W;
                scale = in ? L.l[l].scale : scale;
            }
            const ClSetup S = rcl_setup(box, scale, H, W, lane);
            const int nrows = S.nrows, c0 = S.c0;
            const bool x3 = S.x3;
            // blocked: (B, C / 64, H, W, 64), the copy this library makes; else (B, H, W, C) read in place
            const int cell_bytes = blocked ? RCL_CH * 4 : C * 4;
#ifdef RCL_EXP_NORING
            const bool ring = false;
#else
            const bool ring = S.c1 - c0 < RCL_RING_CELLS;
#endif
            // tap j, wave-uniform: weight (low word) and byte offset of its cell inside a row -- of the map, or of
            // the ring's image of the row, where cells are 256 bytes apart (high word)
            const int tap_bytes = ring ? RCL_CH * 4 : cell_bytes, tap_c0 = ring ? c0 : 0;
            uint64_t wx[28];
#pragma unroll
            for (int j = 0; j < 28; ++j)
                wx[j] = ((uint64_t)(uint32_t)((__builtin_amdgcn_readlane(S.v_cell, j) - tap_c0) * tap_bytes) << 32) |
                        (uint32_t)__builtin_amdgcn_readlane(__float_as_uint(S.v_xw), j);
            const long long slice_bytes = (long long)H * W * cell_bytes;
            const char *base = blocked ? (const char *)feat + ((long long)b * cgroups + g) * slice_bytes
                                       : (const char *)feat + (long long)b * slice_bytes + g * (RCL_CH * 4);
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void *)base, 0, (int)(blocked ? slice_bytes : slice_bytes - g * (RCL_CH * 4)), 0x00020000);
            if (ring) {
                if (x3)
                    rcl_item_ring<3>(acc, rsrc, lane, W * cell_bytes, cell_bytes, c0, wx, S.v_rows, S.v_wy, nrows, sw);
                else
                    rcl_item_ring<4>(acc, rsrc, lane, W * cell_bytes, cell_bytes, c0, wx, S.v_rows, S.v_wy, nrows, sw);
            } else if (x3) {
                rcl_item<3>(acc, rsrc, lane8, W * cell_bytes, wx, S.v_rows, S.v_wy, nrows);
            } else {
                rcl_item<4>(acc, rsrc, lane8, W * cell_bytes, wx, S.v_rows, S.v_wy, nrows);
            }
        }
        // Through LDS, so that the run of RCL_CH x 49 floats leaves as 16-byte stores.
        const float4 *s4 = reinterpret_cast<const float4 *>(sw);
        float4 *o4 = reinterpret_cast<float4 *>(o);
        // a lane holds out[lane][0..48]: 32 channels (6,272 bytes) at a time -- half the LDS per wave, which is what
        // lets sixteen and more waves share a CU
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if ((lane >> 5) == h) {
                float *d = sw + (lane & 31) * 49;
#pragma unroll
                for (int k = 0; k < 49; ++k) d[k] = acc[k];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 6; ++k) rcl_store(o4 + h * 392 + k * 64 + lane, s4[k * 64 + lane]);
            if (lane < 8) rcl_store(o4 + h * 392 + 6 * 64 + lane, s4[6 * 64 + lane]);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------------------
static int cl_supported(const int32_t *hw, int n_levels, int B, int C, int P, int pooled_h, int pooled_w,
                        int sampling_ratio)
{
    if (n_levels < 1 || n_levels > FRCNN_MAX_LEVELS || !hw) return 0;
    if (pooled_h != 7 || pooled_w != 7 || sampling_ratio != 2) return 0;
    if (B <= 0 || C <= 0 || C % RCL_CH != 0 || P < 0 || P >= (1 << 24)) return 0;
    for (int l = 0; l < n_levels; ++l) {
        if (hw[2 * l] <= 0 || hw[2 * l + 1] <= 0) return 0;
        if ((long long)hw[2 * l] * hw[2 * l + 1] * C * 4 >= (1ll << 31)) return 0;
    }
    return 1;
}

#define RCL_CHUNK_MAX 64
struct ClLayout {
    size_t recs, maps[FRCNN_MAX_LEVELS], total;
};

// records per image: every unit padded to a multiple of the chunk size (whatever chunk size a call picks)
static long long cl_padded(int P) { return (long long)(P + RCL_CHUNK_MAX - 1) / RCL_CHUNK_MAX * RCL_CHUNK_MAX + (long long)RCL_MAX_UNITS * RCL_CHUNK_MAX; }

static ClLayout cl_layout(const int32_t *hw, int n_levels, int B, int C, int P, int maps_channels_last)
{
    ClLayout w;
    size_t off = 0;
    w.recs = off;
    off = frcnn_align_up(off + (size_t)B * cl_padded(P) * sizeof(ClRec), 256);
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
        w.maps[l] = off;
        if (l < n_levels && !maps_channels_last)
            off = frcnn_align_up(off + (size_t)B * hw[2 * l] * hw[2 * l + 1] * C * 4, 256);
    }
    w.total = off;
    return w;
}

extern "C" size_t frcnn_roi_align_levels_cl_workspace_bytes(const int32_t *hw, int n_levels, int B, int C, int P,
                                                            int maps_channels_last)
{
    if (!cl_supported(hw, n_levels, B, C, P, 7, 7, 2)) return 0;
    return cl_layout(hw, n_levels, B, C, P, maps_channels_last).total;
}

extern "C" int frcnn_roi_align_levels_cl_f32(const float *const *feats, const int32_t *hw, const float *scales,
                                             int n_levels, int B, int C, int maps_channels_last, const float *props,
                                             const int32_t *count, int P, int pooled_h, int pooled_w,
                                             int sampling_ratio, int k_min, int k_max, float *out, int32_t *total_out,
                                             void *workspace, size_t workspace_bytes, void *stream)
{
    FRCNN_REQUIRE(cl_supported(hw, n_levels, B, C, P, pooled_h, pooled_w, sampling_ratio),
                  "roi_align_levels_cl: needs 1..%d levels, 7 x 7 bins at sampling ratio 2, C a multiple of %d, "
                  "P < 2^24 and maps below 2 GiB per image (use frcnn_roi_align_levels_f32 otherwise)",
                  FRCNN_MAX_LEVELS, RCL_CH);
    FRCNN_REQUIRE(feats && scales && props && out, "roi_align_levels_cl: null pointer");
    for (int l = 0; l < n_levels; ++l) FRCNN_REQUIRE(feats[l], "roi_align_levels_cl: null feature map %d", l);
    FRCNN_REQUIRE(B <= 65535, "roi_align_levels_cl: batch too large");
    const ClLayout w = cl_layout(hw, n_levels, B, C, P, maps_channels_last);
    FRCNN_REQUIRE(workspace && workspace_bytes >= w.total, "roi_align_levels_cl: workspace of %zu bytes needed, got %zu",
                  w.total, workspace_bytes);
    FRCNN_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "roi_align_levels_cl: out must be 16-byte aligned");
    if (P == 0) {
        if (total_out) {
            hipError_t e = hipMemsetAsync(total_out, 0, 4, (hipStream_t)stream);
            FRCNN_REQUIRE(e == hipSuccess, "roi_align_levels_cl: %s", hipGetErrorString(e));
        }
        return FRCNN_OK;
    }
    char *ws = (char *)workspace;
    ClRec *recs = (ClRec *)(ws + w.recs);
    const int cgroups = C / RCL_CH;
    ClLevels L;
    ClCopy T;
    int units = 0, tiles = 0;
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
        ClLevel &t = L.l[l];
        const bool used = l < n_levels;
        t.H = used ? hw[2 * l] : 1;
        t.W = used ? hw[2 * l + 1] : 1;
        t.scale = used ? scales[l] : 1.0f;
        t.feat = used ? (maps_channels_last ? feats[l] : (const float *)(ws + w.maps[l])) : nullptr;
        const long long row_bytes = (long long)t.W * RCL_CH * 4;
        long long band_bytes = RCL_BAND_BYTES;
        if (const char *e = getenv("FRCNN_RCL_BAND_KB")) band_bytes = 1024ll * atoi(e); // experiments
        int band_rows = (int)(band_bytes / row_bytes);
        band_rows = band_rows < 1 ? 1 : band_rows;
        int nbands = used ? (t.H + band_rows - 1) / band_rows : 0;
        // the units of all levels and the zero unit share RCL_MAX_UNITS slots: coarser bands when they do not fit
        const int room = RCL_MAX_UNITS - 1 - units - (n_levels - 1 - l);
        if (used && nbands > room) {
            nbands = room < 1 ? 1 : room;
            band_rows = (t.H + nbands - 1) / nbands;
            nbands = (t.H + band_rows - 1) / band_rows;
        }
        t.unit0 = units;
        t.nbands = nbands;
        t.band_rows = band_rows;
        units += nbands;
        T.src[l] = used ? feats[l] : nullptr;
        T.dst[l] = used ? (float *)(ws + w.maps[l]) : nullptr;
        T.hw[l] = t.H * t.W;
        T.tile0[l] = tiles;
        if (used) tiles += (t.H * t.W + 63) / 64;
    }
    T.tile0[FRCNN_MAX_LEVELS] = tiles;
    T.n = n_levels;
    L.n = n_levels;
    L.units = units + 1;
    FRCNN_REQUIRE(L.units <= RCL_MAX_UNITS, "roi_align_levels_cl: too many bands");
    hipStream_t st = (hipStream_t)stream;
    int chunk_rois = RCL_CHUNK_DEFAULT;
    if (const char *e = getenv("FRCNN_RCL_CHUNK")) chunk_rois = atoi(e); // experiments
    chunk_rois = chunk_rois < 1 ? 1 : (chunk_rois > RCL_CHUNK_MAX ? RCL_CHUNK_MAX : chunk_rois);
    const int chunks = (P + chunk_rois - 1) / chunk_rois + L.units;
    const int padded = chunks * chunk_rois; // <= cl_padded(P)
    hipLaunchKernelGGL(roi_cl_sort_kernel, dim3(B), dim3(RCL_PREP_THREADS), 0, st, L, (const float4 *)props, count, P,
                       B, k_min, k_max, chunk_rois, padded, recs, total_out);
    FRCNN_CHECK_LAUNCH("roi_cl_sort");
    if (!maps_channels_last) {
        hipLaunchKernelGGL(nchw_to_cl_kernel, dim3(tiles, C / 64, B), dim3(256), 0, st, T, C, tiles);
        FRCNN_CHECK_LAUNCH("nchw_to_cl");
    }
    const long long sus = ((long long)B * cgroups + 7) / 8;
    const long long grid = sus * chunks * 8;
    FRCNN_REQUIRE(grid < (1ll << 31), "roi_align_levels_cl: grid too large");
    hipLaunchKernelGGL(roi_align_cl_kernel, dim3((unsigned)grid), dim3(RCL_THREADS), 0, st, L, C, cgroups,
                       maps_channels_last ? 0 : 1, chunks, chunk_rois, padded, (const ClRec *)recs, B, out);
    FRCNN_CHECK_LAUNCH("roi_align_cl");
    return FRCNN_OK;
}
