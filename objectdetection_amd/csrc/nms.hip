// nms.hip -- K4: batched greedy NMS (torchvision.ops.nms semantics; call sites RPN/rpn.py:84 and
// ROIHead/roi_head.py:215).
//
// Greedy NMS only ever consults the IoU of a candidate with the boxes KEPT before it, so neither
// path below builds the n x n suppression matrix:
//
//   lazy kernel  (n <= 2048): one workgroup per image, kept list in LDS, 64 candidates per step:
//                pull against the kept list -> intra-block rows of the survivors -> scalar resolve.
//   round kernels (larger n): candidates are taken 1024 at a time.  Per round, kernel PM spreads
//                over the whole chip (a) the PULL of the round's 1024 candidates against every box
//                kept so far (wave = 64 candidates x a slice of the kept rows, ballot -> atomicOr)
//                and (b) the 136 upper-triangular 64x64 suppression TILES of the round's own
//                candidates; kernel L (one workgroup per image) then resolves the 16 blocks in
//                order from LDS: scalar ffs / readlane / or over survivors, wave-wide OR folds of
//                the kept rows' tiles.  Rounds after max_keep is reached exit at once.
//
// In both, `iou > thr` for 64 columns is one compare whose lane mask (the ballot) IS the 64-bit
// suppression word.  IoU is the reference's fp32 expression with IEEE division; the decision
// `(double)iou > thr` is made with the float threshold t = RD(thr) (largest fp32 <= thr), which is
// equivalent for every fp32 iou.  NaN never suppresses.
#include <math.h>

#include <mutex>

#include "common.h"

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
#define NMS_LAZY_MAX_N 2048 // candidates (and so keeps) whose kept list fits LDS: 24 B per keep
#define NMS_LAZY_WAVES 16
#define NMS_RT_BYTES (136 * 64 * 8) // one round's tiles in the round path's workspace (NMS_RT * 64 words, below)

__device__ __forceinline__ float lane_bcast(float v, int src_lane)
{
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), src_lane));
}

// The walk of nms_lazy_kernel: candidates from block `first_block` on, 64 at a time, against the kept list in LDS (nk0
// entries on entry; the keeps found are appended: boxes, areas, positions).  Every thread of a 1024-thread workgroup
// calls it (it contains barriers).  Returns the number of keeps; *blocks_out = blocks that had work.
__device__ __forceinline__ int nms_lazy_walk(const float4 *__restrict__ bx, int n, int first_block, int nk0,
                                             const FrcnnThr &thr, int max_keep, float4 *s_kbox, float *s_karea,
                                             int *s_kidx, float4 *s_cand, unsigned long long *s_row, unsigned *s_rem,
                                             int *s_nkeep_p, int *blocks_out)
{
    int &s_nkeep = *s_nkeep_p;
    const int Wn = (n + 63) >> 6;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        s_nkeep = nk0;
        s_rem[0] = 0;
        s_rem[1] = 0;
    }
    __syncthreads();
    int nkeep_final = nk0;
    int blocks_done = first_block; // 64-candidate blocks that had work (the tail reports them)
    for (int c0 = first_block; c0 < Wn; c0 += NMS_LAZY_WAVES) {
        // refill: one coalesced 16-KB read serves the next 16 candidate blocks; no global memory
        // operation is issued inside the per-block loop below, so its barriers are cheap
        {
            int idx = c0 * 64 + threadIdx.x;
            s_cand[threadIdx.x] = bx[min(idx, n - 1)]; // (beyond n: marked removed below)
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
                float inter, uni;
                frcnn_inter_union(k.x, k.y, k.z, k.w, s_karea[r], cb.x, cb.y, cb.z, cb.w, carea, &inter, &uni);
                sup = sup | frcnn_iou_gt(inter, uni, thr);
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
                        float inter, uni;
                        frcnn_inter_union(rx1, ry1, rx2, ry2, rarea, cb.x, cb.y, cb.z, cb.w, carea, &inter, &uni);
                        unsigned long long w = __ballot(frcnn_iou_gt(inter, uni, thr) && lane > i);
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
            blocks_done = c + 1;
            if (nkeep_final >= max_keep) {
                done = true;
                break;
            }
        }
        if (done) break;
    }
    __syncthreads();
    nkeep_final = s_nkeep;
    *blocks_out = blocks_done;
    return nkeep_final;
}

__global__ void __launch_bounds__(NMS_LAZY_THREADS)
nms_lazy_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ n_per_image, int max_n,
                FrcnnThr thr, int max_keep, int cap, int32_t *__restrict__ keep_out,
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
    int blocks_done = 0;
    const int nkeep_final = nms_lazy_walk(boxes + (long long)b * max_n, n, 0, 0, thr, max_keep, s_kbox, s_karea, s_kidx,
                                          s_cand, s_row, s_rem, &s_nkeep, &blocks_done);
    int32_t *keep = keep_out + (long long)b * max_keep;
    for (int t = threadIdx.x; t < nkeep_final; t += NMS_LAZY_THREADS) keep[t] = s_kidx[t];
    if (threadIdx.x == 0) n_keep_out[b] = nkeep_final;
}

// ---- round-based NMS --------------------------------------------------------------------------
#define NMS_RS 1024                       // candidates per round
#define NMS_RC (NMS_RS / 64)              // 64-candidate blocks per round (16)
#define NMS_RT (NMS_RC * (NMS_RC + 1) / 2) // upper-triangular tiles per round (136)
#ifndef NMS_PULL_WGS
#define NMS_PULL_WGS 1024
#endif
#define NMS_DENSE 512                     // survivors per round from which chip-wide tiles pay off
#define NMS_TILE_WORDS ((NMS_RT * 64 + 1023) / 1024) // tile words per thread of the scan kernel (9)
static_assert(NMS_RT_BYTES == NMS_RT * 64 * 8, "NMS_RT_BYTES");
static_assert(NMS_RT_BYTES >= NMS_LAZY_MAX_N * 24 && NMS_RS == NMS_LAZY_THREADS, "the tail walk's kept list (NMS_LAZY_MAX_N entries) lies over the scan kernel's tiles, its candidates over the list");
#define NMS_TILE_GROUPS ((NMS_RT + NMS_RC * 4 - 1) / (NMS_RC * 4)) // grid.y slices of tile waves (3)

// Per-image workspace of the round path (all offsets 16-byte aligned).
struct NmsWs {
    float4 *kbox;              // [cap] kept boxes, in keep order
    float *karea;              // [cap]
    unsigned long long *tiles; // [NMS_RT][64] this round's suppression tiles, tile-major
    unsigned long long *sup;   // [NMS_RC] this round's pull result (zero between rounds)
    int *nkeep;                // [1]
    int *dense;                // [1] last round left >= NMS_DENSE survivors of the pull: precompute tiles
    int *rounds;               // [1] 64-candidate blocks that had work to do so far (16 per round)
};

static size_t nms_ws_stride(int cap)
{
    return frcnn_align_up((size_t)cap * 16, 256) + frcnn_align_up((size_t)cap * 4, 256) +
           (size_t)NMS_RT * 64 * 8 + 256 /* sup */ + 256 /* nkeep */;
}

__device__ __forceinline__ NmsWs nms_ws(char *base, size_t stride, int b, int cap)
{
    char *p = base + (size_t)b * stride;
    NmsWs w;
    w.kbox = (float4 *)p;
    p += ((size_t)cap * 16 + 255) / 256 * 256;
    w.karea = (float *)p;
    p += ((size_t)cap * 4 + 255) / 256 * 256;
    w.tiles = (unsigned long long *)p;
    p += (size_t)NMS_RT * 64 * 8;
    w.sup = (unsigned long long *)p;
    p += 256;
    w.nkeep = (int *)p;
    w.dense = w.nkeep + 1;
    w.rounds = w.nkeep + 2;
    return w;
}

// tile id -> (row block, col block), row-major over the upper triangle
__device__ __forceinline__ void nms_tile_rc(int id, int *rb, int *cb)
{
    int r = 0, left = id;
    while (left >= NMS_RC - r) {
        left -= NMS_RC - r;
        ++r;
    }
    *rb = r;
    *cb = r + left;
}

__device__ __forceinline__ int nms_tile_id(int rb, int cb)
{
    return rb * NMS_RC - rb * (rb - 1) / 2 + (cb - rb);
}

// One 64x64 tile of COLUMN words: lane = column candidate, the 64 row candidates (boxes and areas) are
// broadcast from LDS.  Bit i of a lane's word = "row i suppresses me".  32 rows at a time, highest first, so
// that a word grows as word * 2 + bit (one add-with-carry); the threshold test is the grouped form of
// common.h: a half tile with an unsafe pair is redone with the division.
template <bool EXACT>
__device__ __forceinline__ unsigned nms_tile_rows32(const float4 *rows, const float *rarea, const float4 &c,
                                                    float carea, const FrcnnThr &thr, bool &sure)
{
    unsigned w = 0;
#pragma unroll 8
    for (int i = 31; i >= 0; --i) {
        const float4 r = rows[i];
        float inter, uni;
        frcnn_inter_union(r.x, r.y, r.z, r.w, rarea[i], c.x, c.y, c.z, c.w, carea, &inter, &uni);
        const bool s = EXACT ? frcnn_iou_gt_exact(inter, uni, thr) : frcnn_iou_gt_fast(inter, uni, thr, sure);
        w = w + w + (s ? 1u : 0u);
    }
    return w;
}

__device__ __forceinline__ unsigned long long nms_tile_words(const float4 *rows, const float *rarea,
                                                             const float4 &c, float carea, const FrcnnThr &thr)
{
    unsigned h[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        bool sure = true;
        unsigned w = nms_tile_rows32<false>(rows + 32 * half, rarea + 32 * half, c, carea, thr, sure);
        if (__builtin_expect(frcnn_group_unsure(sure, thr), 0))
            w = nms_tile_rows32<true>(rows + 32 * half, rarea + 32 * half, c, carea, thr, sure);
        h[half] = w;
    }
    return ((unsigned long long)h[1] << 32) | h[0];
}

// A tile of the candidate list held in LDS (the scan kernel's own tiles).  Rows or columns beyond `count`
// hold stale boxes: they are marked removed, so their bits are never looked at.
__device__ __forceinline__ unsigned long long nms_tile_from_lds(const float4 *s_box, const float *s_area, int rb,
                                                                int cb, int lane, const FrcnnThr &thr)
{
    const float4 c = s_box[cb * 64 + lane];
    unsigned long long word = nms_tile_words(s_box + rb * 64, s_area + rb * 64, c, s_area[cb * 64 + lane], thr);
    if (rb == cb) word &= (1ull << lane) - 1ull; // only earlier candidates of the same block
    return word;
}

// Kernel PM of a round.  grid = (16 candidate blocks, RSPLIT pull slices + tile slices, B).
__global__ void __launch_bounds__(256)
nms_round_pm_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ n_per_image,
                    int max_n, int n_cap, FrcnnThr thr, int max_keep, int cap, int round, int rsplit, char *ws,
                    size_t ws_stride)
{
    __shared__ float4 s_rows[4][64]; // a tile wave's row block
    __shared__ float s_rarea[4][64];
    const int b = blockIdx.z;
    // (n_cap: the candidates this call's rounds cover; the last scan's tail takes the rest, see nms_sorted_run)
    const int n = min(n_per_image ? min(n_per_image[b], max_n) : max_n, n_cap);
    const int base = round * NMS_RS;
    if (base >= n) return;
    const NmsWs w = nms_ws(ws, ws_stride, b, cap);
    const int nk = round == 0 ? 0 : *w.nkeep; // (round 0 reads nothing of the workspace: it need not be cleared)
    if (nk >= max_keep) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float4 *bx = boxes + (long long)b * max_n;
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if ((int)blockIdx.y < rsplit) {
        // ---- pull: 64 candidates (lanes) x kept rows wave*..; the kept box is wave-uniform
        const int c = blockIdx.x;
        const int idx = base + c * 64 + lane;
        if (base + c * 64 >= n || nk == 0) return;
        const float4 cb = idx < n ? bx[idx] : zero4;
        const float carea = frcnn_area(cb.x, cb.y, cb.z, cb.w);
        bool sup = false;
        // The workgroup's slice of the kept list goes through LDS 256 rows at a time, one row per thread, the
        // next 256 requested before these are tested: two or three memory round trips per workgroup instead of
        // one per group of pairs (four waves per SIMD do not hide those).  Wave w tests rows w, w + 4, ... of
        // the 256, four per group, with the grouped threshold test of common.h.
        float4 *s_k = &s_rows[0][0];
        float *s_ka = &s_rarea[0][0];
        const int per = ((nk + rsplit - 1) / rsplit + 3) & ~3;
        const int r_begin = blockIdx.y * per, r_end = min(nk, r_begin + per);
        if (r_begin >= r_end) return;
        // (rows beyond the slice: the slice's last row again -- never tested, see `row < m` below)
        float4 pk = w.kbox[min(r_begin + (int)threadIdx.x, r_end - 1)];
        float pa = w.karea[min(r_begin + (int)threadIdx.x, r_end - 1)];
        for (int r0 = r_begin; r0 < r_end; r0 += 256) {
            __syncthreads(); // the previous 256 have been tested
            s_k[threadIdx.x] = pk;
            s_ka[threadIdx.x] = pa;
            __syncthreads();
            pk = w.kbox[min(r0 + 256 + (int)threadIdx.x, r_end - 1)];
            pa = w.karea[min(r0 + 256 + (int)threadIdx.x, r_end - 1)];
            const int m = min(256, r_end - r0); // rows of this piece
            for (int g = 0; g * 4 + wave < m; g += 4) {
                float inter[4], uni[4];
                bool sure = true, s = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = (g + j) * 4 + wave; // < 256 + 12: the arrays end at 256, so clamp
                    const float4 k = s_k[min(row, 255)];
                    frcnn_inter_union(k.x, k.y, k.z, k.w, s_ka[min(row, 255)], cb.x, cb.y, cb.z, cb.w, carea,
                                      &inter[j], &uni[j]);
                    const bool sj = frcnn_iou_gt_fast(inter[j], uni[j], thr, sure);
                    s = s | (sj && row < m);
                }
                if (__builtin_expect(frcnn_group_unsure(sure, thr), 0)) {
                    s = false;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        s = s | (frcnn_iou_gt_exact(inter[j], uni[j], thr) && (g + j) * 4 + wave < m);
                }
                sup = sup | s;
            }
        }
        const unsigned long long m = __ballot(sup);
        if (lane == 0 && m != 0) atomicOr(&w.sup[c], m);
    } else {
        // ---- tiles: wave = one 64x64 tile of the round's own candidates
        const int id = (((int)blockIdx.y - rsplit) * NMS_RC + blockIdx.x) * 4 + wave;
        if (id >= NMS_RT) return;
        // chip-wide tiles only pay off when most of the round survives the pull; the scan kernel
        // builds the few tiles of a sparse round itself (it records which case the last round was)
        if (round != 0 && *w.dense == 0) return;
        int rb, cbk;
        nms_tile_rc(id, &rb, &cbk);
        if (base + cbk * 64 >= n) return;
        const int ri = base + rb * 64 + lane, ci = base + cbk * 64 + lane;
        const float4 rbx = ri < n ? bx[ri] : zero4;
        const float4 cbx = ci < n ? bx[ci] : zero4;
        const float rarea = frcnn_area(rbx.x, rbx.y, rbx.z, rbx.w);
        const float carea = frcnn_area(cbx.x, cbx.y, cbx.z, cbx.w);
        const bool diag = rb == cbk;
        // COLUMN words: bit i of lane c's word = "row candidate i suppresses column candidate c".
        // With the suppressors of a candidate in its own lane, the scan kernel needs no cross-lane
        // reduction: `(word & kept_mask) != 0` is lane-local and a ballot collects the block.
        // The row block goes through a wave-private piece of LDS: one broadcast read per row instead of five
        // cross-lane register reads.  (Candidates beyond n are marked removed by the scan: their bits are unused.)
        s_rows[wave][lane] = rbx;
        s_rarea[wave][lane] = rarea;
        __builtin_amdgcn_wave_barrier();
        unsigned long long word = nms_tile_words(s_rows[wave], s_rarea[wave], cbx, carea, thr);
        if (diag) word &= (1ull << lane) - 1ull;
        if (ci >= n) word = 0;
        w.tiles[(long long)id * 64 + lane] = word;
    }
}

// Kernel L of a round: one workgroup per image resolves the round's candidates in order.
// It works on a LIST of candidates in LDS: the identity list with the pull's suppression words when
// most of the round survived the pull (dense: tiles come from kernel PM), or the compacted list of
// survivors when few did (sparse: the handful of tiles is built here by the 16 waves).
__global__ void __launch_bounds__(1024)
nms_round_scan_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ n_per_image,
                      int max_n, int n_cap, FrcnnThr thr, int max_keep, int cap, int round, char *ws,
                      size_t ws_stride, int32_t *__restrict__ keep_out,
                      int32_t *__restrict__ n_keep_out, int32_t *rounds_used, int tail_block)
{
    __shared__ unsigned long long s_tiles[NMS_RT * 64];
    __shared__ unsigned long long s_trow[64]; // the tail walk's (tail_block > 0)
    __shared__ unsigned s_trem[2];
    __shared__ float4 s_box[NMS_RS];   // the list's boxes
    __shared__ float s_area[NMS_RS];   // and their areas
    __shared__ int s_pos[NMS_RS];      // the list's positions inside the round
    __shared__ unsigned long long s_removed[NMS_RC];
    __shared__ unsigned long long s_K[NMS_RC];
    __shared__ int s_nkeep;
    const int b = blockIdx.x;
    const int n = min(n_per_image ? min(n_per_image[b], max_n) : max_n, n_cap);
    const int base = round * NMS_RS;
    const NmsWs w = nms_ws(ws, ws_stride, b, cap);
    // Round 0 reads nothing of the workspace and leaves count, pull words and flag behind for the later rounds,
    // so the caller's buffer needs no clearing.
    const int nk0 = round == 0 ? 0 : *w.nkeep;
    if (base >= n || nk0 >= max_keep) {
        if (threadIdx.x == 0) {
            n_keep_out[b] = nk0;
            if (round == 0) {
                *w.nkeep = 0; // (an empty image: what the later rounds report)
                *w.rounds = 0;
            }
            if (rounds_used) rounds_used[b] = round == 0 ? 0 : *w.rounds; // (the call's last round publishes)
        }
        return;
    }
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int cnt = min(NMS_RS, n - base);
    const bool pm_tiles = round == 0 || *w.dense != 0; // what kernel PM decided for this image
    // Everything this workgroup reads from memory is requested here, in one round trip: its candidate, and
    // kernel PM's tiles (from all over the chip, so from beyond this XCD's L2) if there are any -- whether
    // they are used is only known once the pull words are in.
    const float4 mybox = boxes[(long long)b * max_n + base + min(t, cnt - 1)]; // (beyond cnt: marked removed below)
    unsigned long long tv[NMS_TILE_WORDS];
#pragma unroll
    for (int u = 0; u < NMS_TILE_WORDS; ++u)
        tv[u] = (pm_tiles && t + u * 1024 < NMS_RT * 64) ? w.tiles[t + u * 1024] : 0ull;
    // survivors of the pull
    unsigned long long pulled = 0;
    if (t < NMS_RC) {
        const int rows = min(64, cnt - t * 64);
        const unsigned long long valid = rows >= 64 ? ~0ull : (rows <= 0 ? 0ull : ((1ull << rows) - 1ull));
        pulled = (round == 0 ? 0ull : w.sup[t]) | ~valid;
        s_removed[t] = pulled;
        s_K[t] = 0;
        w.sup[t] = 0; // ready for the next round's atomicOr
    }
    if (t == 0) s_nkeep = nk0;
    __syncthreads();
    int alive = 0;
    for (int c = 0; c < NMS_RC; ++c) alive += 64 - __popcll(s_removed[c]);
    const bool dense = alive >= min(NMS_DENSE, (cnt + 1) / 2); // (a round cut short by the caller's bound: half of what it has)
    int count; // length of the list
    if (dense) {
        count = cnt;
        s_box[t] = mybox;
        s_area[t] = frcnn_area(mybox.x, mybox.y, mybox.z, mybox.w);
        s_pos[t] = t;
        __syncthreads();
    } else {
        count = alive;
        const unsigned long long rm = s_removed[wave];
        int rank = 0;
        for (int c = 0; c < wave; ++c) rank += 64 - __popcll(s_removed[c]);
        rank += lane - __popcll(rm & ((1ull << lane) - 1ull));
        __syncthreads(); // everyone has read the pull words
        if (!((rm >> lane) & 1ull)) {
            s_box[rank] = mybox;
            s_area[rank] = frcnn_area(mybox.x, mybox.y, mybox.z, mybox.w);
            s_pos[rank] = t;
        }
        if (t < NMS_RC) { // the compacted list starts with nothing removed but its tail
            const int rows = min(64, count - t * 64);
            s_removed[t] = rows >= 64 ? 0ull : (rows <= 0 ? ~0ull : ~((1ull << rows) - 1ull));
        }
        __syncthreads();
    }
    const int nC = (count + 63) >> 6;
    if (dense && pm_tiles) {
#pragma unroll
        for (int u = 0; u < NMS_TILE_WORDS; ++u)
            if (t + u * 1024 < NMS_RT * 64) s_tiles[t + u * 1024] = tv[u];
    } else {
        // the nC (nC + 1) / 2 tiles of the list, dealt to the waves in turn: every SIMD gets its share
        const int nT = nC * (nC + 1) / 2;
        for (int q = wave; q < nT; q += 16) {
            int rb = 0, left = q;
            while (left >= nC - rb) {
                left -= nC - rb;
                ++rb;
            }
            const int cb = rb + left;
            s_tiles[nms_tile_id(rb, cb) * 64 + lane] = nms_tile_from_lds(s_box, s_area, rb, cb, lane, thr);
        }
    }
    __syncthreads();
    // One wave walks the blocks of the list with no barrier.  Lane j of block c holds col = the
    // earlier candidates of the block that would suppress j.  Resolve = fixed-point iteration on
    // uniform masks: a candidate is removed once a KEPT one suppresses it, and kept once no kept or
    // still undecided one does; the lowest undecided candidate is always decidable, so the loop
    // ends after (dependency-chain depth) <= 64 passes of ~10 instructions, typically 1-3.  Folding
    // the block's keeps into a later block is `ballot((col & K) != 0)` on that block's tile.
    if (wave == 0) {
        unsigned long long removed[NMS_RC];
#pragma unroll
        for (int cb = 0; cb < NMS_RC; ++cb) removed[cb] = uniform64(s_removed[cb]);
        int nk = nk0;
#pragma unroll
        for (int c = 0; c < NMS_RC; ++c) {
            if (c < nC && nk < max_keep) {
                const int tb = nms_tile_id(c, c);
                const unsigned long long col = s_tiles[tb * 64 + lane];
                // the block's words in the later blocks' tiles: requested together, ahead of the resolve loop
                // (one LDS latency per block instead of one per later block; beyond nC they are never used)
                unsigned long long cw[NMS_RC];
#pragma unroll
                for (int cb = c + 1; cb < NMS_RC; ++cb) cw[cb] = s_tiles[(tb + cb - c) * 64 + lane];
                unsigned long long U = ~removed[c]; // undecided
                unsigned long long K = 0;           // decided: kept
                // (all 64 lanes are active, and a ballot of a bare comparison is its mask: the rest is scalar)
                while (U != 0) {
                    const unsigned long long killed = __ballot((col & K) != 0);
                    const unsigned long long busy = __ballot((col & (K | U)) != 0); // killed, or waiting for an undecided one
                    const unsigned long long keep_now = U & ~busy;
                    K |= keep_now;
                    U &= ~(killed | keep_now);
                }
                // only the first (max_keep - nk) keeps count: drop the highest surplus bits
                int surplus = __popcll(K) - (max_keep - nk);
                while (surplus > 0) {
                    K &= ~(1ull << (63 - __clzll((long long)K)));
                    --surplus;
                }
                if (lane == 0) s_K[c] = K;
                nk += __popcll(K);
                if (K != 0 && nk < max_keep) {
#pragma unroll
                    for (int cb = c + 1; cb < NMS_RC; ++cb)
                        if (cb < nC) removed[cb] |= __ballot((cw[cb] & K) != 0);
                }
            }
        }
        if (lane == 0) s_nkeep = nk;
    }
    __syncthreads();
    // append the round's keeps to the kept list (list order = candidate order)
    {
        const unsigned long long K = s_K[wave];
        if (wave < nC && ((K >> lane) & 1ull)) {
            int pos = nk0 + __popcll(K & ((1ull << lane) - 1ull));
            for (int cc = 0; cc < wave; ++cc) pos += __popcll(s_K[cc]);
            const float4 bxv = s_box[t];
            w.kbox[pos] = bxv;
            w.karea[pos] = s_area[t];
            keep_out[(long long)b * max_keep + pos] = base + s_pos[t];
            if (tail_block > 0 && pos < NMS_LAZY_MAX_N) { // (the tiles are done with: the tail's kept list lies over them)
                ((float4 *)s_tiles)[pos] = bxv;
                ((float *)((float4 *)s_tiles + NMS_LAZY_MAX_N))[pos] = s_area[t];
            }
        }
    }
    if (t == 0) {
        *w.nkeep = s_nkeep;
        *w.dense = dense ? 1 : 0;
        // 64-candidate blocks that had work so far: where in the round max_keep was reached
        // how far into the candidates this image's work reached: the block of the keep that filled max_keep, or --
        // not filled yet -- every candidate of the round.  (A function of the data alone: the same whatever bound
        // the call ran under, whether the list was the round itself or a compacted one.)
        int in_round = (cnt + 63) >> 6;
        if (s_nkeep >= max_keep) {
            for (int c = NMS_RC - 1; c >= 0; --c) {
                const unsigned long long K = s_K[c];
                if (K != 0) {
                    in_round = s_pos[c * 64 + 63 - __clzll((long long)K)] / 64 + 1;
                    break;
                }
            }
        }
        const int blocks = round * NMS_RC + in_round;
        *w.rounds = blocks;
        n_keep_out[b] = s_nkeep;
        if (rounds_used) rounds_used[b] = blocks;
    }
    // The call's last round under a bound (tail_block = the first 64-candidate block the rounds did not cover): an
    // image that is neither full nor out of candidates -- the bound was too tight for it -- is finished here, by this
    // workgroup, 64 candidates at a time against its kept list in LDS (nms_lazy_walk: slow next to the rounds, exact all
    // the same).  The normal case costs the test.  Kept list: the earlier rounds' entries from the workspace, this round's
    // were put into LDS as they were appended.
    if (tail_block > 0) {
        const int n_img = n_per_image ? min(n_per_image[b], max_n) : max_n;
        __syncthreads();
        const int nk1 = s_nkeep;
        if (nk1 < max_keep && tail_block * 64 < n_img) {
            float4 *l_kbox = (float4 *)s_tiles;
            float *l_karea = (float *)(l_kbox + NMS_LAZY_MAX_N);
            int *l_kidx = (int *)(l_karea + NMS_LAZY_MAX_N);
            for (int i = t; i < nk0; i += 1024) {
                l_kbox[i] = w.kbox[i];
                l_karea[i] = w.karea[i];
            }
            int blocks_done = tail_block;
            const int nk2 = nms_lazy_walk(boxes + (long long)b * max_n, n_img, tail_block, nk1, thr, max_keep, l_kbox,
                                          l_karea, l_kidx, s_box, s_trow, s_trem, &s_nkeep, &blocks_done);
            for (int i = nk1 + t; i < nk2; i += 1024) keep_out[(long long)b * max_keep + i] = l_kidx[i];
            if (t == 0) {
                n_keep_out[b] = nk2;
                if (rounds_used) rounds_used[b] = blocks_done;
            }
        }
    }
}


extern "C" size_t frcnn_nms_workspace_bytes(int B, int max_n)
{
    if (B <= 0 || max_n <= NMS_LAZY_MAX_N) return 0;
    return (size_t)B * nms_ws_stride(max_n);
}

// t = largest fp32 with (double)t <= thr:  (double)x > thr  <=>  x > t  for every fp32 x.
static FrcnnThr make_threshold(double thr)
{
    FrcnnThr th;
    th.t = (float)thr;
    if ((double)th.t > thr) th.t = nextafterf(th.t, -INFINITY);
    th.t2 = nextafterf(th.t, INFINITY);
    th.fast = (th.t2 >= 1e-6f && th.t2 < 1e6f) ? 1 : 0;
    return th;
}

// max_blocks > 0: the rounds cover at most the first max_blocks 64-candidate blocks of every image (the last round
// possibly only part of its 16), and the last round's scan workgroup finishes whatever image is not done by itself
// (nms_lazy_walk, the tail) -- for callers who know from the previous batch how far into the candidates the work reaches: rounds after max_keep is reached exit at once, but still cost two launches each (nine of twelve at
// 12,000 -> 2,000).  The result never depends on max_blocks.  rounds_used (B ints, may be host-mapped memory, may
// be null): per image, the 64-candidate BLOCKS that had work (16 per round: how far into its last round an image
// got tells the caller how much slack its bound has) -- written once, by the call's last launch.
static int nms_sorted_run(const float *boxes, const int32_t *n_per_image, int B, int max_n,
                          double iou_threshold, int max_keep, int32_t *keep_out, int32_t *n_keep_out,
                          void *workspace, size_t workspace_bytes, int max_blocks, int32_t *rounds_used,
                          void *stream)
{
    FRCNN_REQUIRE(B >= 0 && max_n >= 0 && max_keep >= 0, "nms: bad shape");
    if (B == 0) return FRCNN_OK;
    FRCNN_REQUIRE(n_keep_out, "nms: null n_keep_out");
    hipStream_t st = (hipStream_t)stream;
    if (max_n == 0 || max_keep == 0) {
        hipError_t e = hipMemsetAsync(n_keep_out, 0, sizeof(int32_t) * (size_t)B, st);
        FRCNN_REQUIRE(e == hipSuccess, "nms: memset failed: %s", hipGetErrorString(e));
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(boxes && keep_out, "nms: null pointer");
    FRCNN_REQUIRE(iou_threshold == iou_threshold, "nms: NaN threshold");
    FRCNN_REQUIRE(B <= 65535, "nms: batch too large");
    const FrcnnThr thr = make_threshold(iou_threshold);
    const int cap = max_keep < max_n ? max_keep : max_n; // no more than max_n boxes can be kept

    if (max_n <= NMS_LAZY_MAX_N) { // kept list (24 B per keep) fits LDS: one launch
        const size_t lazy_lds = (size_t)cap * 24;
        if (lazy_lds > 48 * 1024) {
            static size_t granted[FRCNN_MAX_DEVICES];
            hipError_t e = frcnn_reserve_dynamic_lds((const void *)nms_lazy_kernel, NMS_LAZY_MAX_N * 24, granted);
            FRCNN_REQUIRE(e == hipSuccess, "nms: cannot reserve LDS: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(nms_lazy_kernel, dim3(B), dim3(NMS_LAZY_THREADS), lazy_lds, st,
                           (const float4 *)boxes, n_per_image, max_n, thr, max_keep, cap, keep_out, n_keep_out);
        FRCNN_CHECK_LAUNCH("nms_lazy");
        return FRCNN_OK;
    }

    FRCNN_REQUIRE(workspace, "nms: null workspace");
    const size_t need = frcnn_nms_workspace_bytes(B, max_n);
    if (workspace_bytes < need) {
        frcnn_set_error("nms: workspace too small (%zu < %zu)", workspace_bytes, need);
        return FRCNN_EWORKSPACE;
    }
    // the kept-list capacity of the workspace is max_n whatever max_keep is
    const size_t stride = nms_ws_stride(max_n);
    int rsplit = (NMS_PULL_WGS + NMS_RC * B - 1) / (NMS_RC * B); // pull slices: ~NMS_PULL_WGS workgroups in flight
    if (rsplit < 1) rsplit = 1;
    if (rsplit > 16) rsplit = 16;
    // the tail keeps its kept list in LDS: only when that fits (RPN: 2,000 keeps) can candidates be left to it
    const int all_blocks = (max_n + 63) / 64;
    int blocks = all_blocks;
    if (max_blocks > 0 && max_blocks < all_blocks && cap <= NMS_LAZY_MAX_N) blocks = max_blocks;
    const int run = (blocks + NMS_RC - 1) / NMS_RC; // rounds; the last one may cover only part of its candidates
    const int n_cap = blocks < all_blocks ? blocks * 64 : max_n;
    for (int r = 0; r < run; ++r) {
        // a last round cut short has fewer candidate blocks: slice its pull finer, so that it still fills the chip
        int rs = r == 0 ? 0 : rsplit; // (round 0 has no kept boxes to pull against: tile slices only)
        const int in_round = blocks - r * NMS_RC;
        if (in_round < NMS_RC && r > 0) {
            rs = rsplit * NMS_RC / in_round;
            if (rs > 16) rs = 16;
        }
        hipLaunchKernelGGL(nms_round_pm_kernel, dim3(NMS_RC, rs + NMS_TILE_GROUPS, B), dim3(256), 0, st,
                           (const float4 *)boxes, n_per_image, max_n, n_cap, thr, max_keep, max_n, r, rs,
                           (char *)workspace, stride);
        FRCNN_CHECK_LAUNCH("nms_round_pm");
        hipLaunchKernelGGL(nms_round_scan_kernel, dim3(B), dim3(1024), 0, st, (const float4 *)boxes,
                           n_per_image, max_n, n_cap, thr, max_keep, max_n, r, (char *)workspace, stride,
                           keep_out, n_keep_out, r == run - 1 ? rounds_used : (int32_t *)nullptr,
                           (r == run - 1 && blocks < all_blocks) ? blocks : 0);
        FRCNN_CHECK_LAUNCH("nms_round_scan");
    }
    return FRCNN_OK;
}

extern "C" int frcnn_nms_sorted_f32(const float *boxes, const int32_t *n_per_image, int B, int max_n,
                                    double iou_threshold, int max_keep, int32_t *keep_out,
                                    int32_t *n_keep_out, void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    return nms_sorted_run(boxes, n_per_image, B, max_n, iou_threshold, max_keep, keep_out, n_keep_out, workspace,
                          workspace_bytes, 0, nullptr, stream);
}

extern "C" int frcnn_nms_sorted_rounds_f32(const float *boxes, const int32_t *n_per_image, int B, int max_n,
                                           double iou_threshold, int max_keep, int32_t *keep_out,
                                           int32_t *n_keep_out, void *workspace, size_t workspace_bytes,
                                           int max_blocks, int32_t *rounds_used, void *stream)
{
    return nms_sorted_run(boxes, n_per_image, B, max_n, iou_threshold, max_keep, keep_out, n_keep_out, workspace,
                          workspace_bytes, max_blocks, rounds_used, stream);
}

// ---- the entry point that allocates nothing ---------------------------------------------------------
// For callers that cannot allocate device memory (HIP graph capture, tight serving loops, harnesses that
// forbid it): the round path's workspace (kept list, one round's tiles, pull words, counters) lives in a
// file-scope device array instead of a caller buffer.  Every GPU holds its own copy of the array at its own
// address; `scratch` below is only the host-side stand-in of the symbol, so host code never passes it to a
// kernel: the device address is looked up once per device (hipGetSymbolAddress, with that device current)
// and kept in g_scratch_ptr[device].  The NMS kernels need nothing of the workspace cleared (round 0 reads
// none of it), so a call starts on whatever the previous call left there.  All calls on one GPU share the
// array: calls that overlap in time on one GPU (two streams, two threads) race.
#define NMS_STATIC_SCRATCH_BYTES ((size_t)16 << 20)
static __device__ __align__(256) unsigned char scratch[NMS_STATIC_SCRATCH_BYTES];
static void *g_scratch_ptr[FRCNN_MAX_DEVICES];
static std::mutex g_scratch_lock;

static int nms_scratch_address(void **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    FRCNN_REQUIRE(e == hipSuccess, "nms_static: hipGetDevice failed: %s", hipGetErrorString(e));
    FRCNN_REQUIRE(dev >= 0 && dev < FRCNN_MAX_DEVICES, "nms_static: device %d not supported (max %d)", dev,
                  FRCNN_MAX_DEVICES);
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        FRCNN_REQUIRE(e == hipSuccess && p, "nms_static: hipGetSymbolAddress failed on device %d: %s", dev,
                      hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return FRCNN_OK;
}

extern "C" size_t frcnn_nms_static_scratch_bytes(void) { return NMS_STATIC_SCRATCH_BYTES; }

extern "C" int frcnn_nms_sorted_static_f32(const float *boxes, const int32_t *n_per_image, int B, int max_n,
                                           double iou_threshold, int max_keep, int32_t *keep_out,
                                           int32_t *n_keep_out, void *stream)
{
    FRCNN_REQUIRE(B >= 0 && max_n >= 0 && max_keep >= 0, "nms_static: bad shape");
    const size_t need = frcnn_nms_workspace_bytes(B, max_n);
    FRCNN_REQUIRE(need <= NMS_STATIC_SCRATCH_BYTES,
                  "nms_static: B = %d images of max_n = %d boxes need %zu B of scratch, the built-in array holds %zu B; "
                  "use frcnn_nms_sorted_f32 with a caller workspace",
                  B, max_n, need, (size_t)NMS_STATIC_SCRATCH_BYTES);
    void *ws = nullptr;
    if (need > 0) {
        const int rc = nms_scratch_address(&ws);
        if (rc != FRCNN_OK) return rc;
    }
    return nms_sorted_run(boxes, n_per_image, B, max_n, iou_threshold, max_keep, keep_out, n_keep_out, ws, need,
                          0, nullptr, stream);
}
