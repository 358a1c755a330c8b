// roi_backward.hip -- backward of the batched MultiScaleRoIAlign level pass with respect to the feature map.
// The reference pools with torchvision's differentiable MultiScaleRoIAlign (ROIHead/roi_pooling.py:51-53, call
// site ROIHead/roi_head.py:127-130), so its RoI losses train the FPN / backbone through it; this is that
// gradient: every sample of a bin sends grad * w_k / count to its four taps (torchvision's
// roi_align_backward), samples outside [-1, H] x [-1, W] send nothing, boxes get no gradient.
//
// torchvision scatters with one global float atomic per tap (16 per output element).  Here a workgroup owns
// (image, channel chunk, RoI split) and keeps the chunk's GRADIENT PLANES in LDS, one wave per channel pair, and the
// scatter needs no atomics: RoIAlign's weights are separable, so per RoI the update of a channel is Wx^T (G^T Wy), two
// small matrix products that run on the fp32 MFMA (see roi_align_plane_backward_kernel); HBM sees grad_out once plus
// one write of the planes, and with one workgroup per (image, chunk) the result is deterministic.  Planes too large
// for four channels of them to fit LDS (FPN P3 / P2 at 800 x 1344) are walked window by window
// (roi_align_window_backward_kernel) and added up with global atomics, whose sum order is not fixed (as in
// torchvision); shapes outside all of that (other bin counts / sampling ratios) scatter per tap with global atomics.
#include <math.h>

#include "common.h"
#include "roi_common.h"

#ifndef RAB_DEPTH
#define RAB_DEPTH 6 // RoIs whose grad_out is in flight per wave
#endif
#define RAB_LDS_BUDGET (160 * 1024 - 1024) // dynamic LDS of the plane kernel: planes + tables

__device__ __forceinline__ float4 rab_record(float4 box, float scale)
{
    float start_w = box.x * scale, start_h = box.y * scale;
    float roi_w = box.z * scale - start_w, roi_h = box.w * scale - start_h;
    roi_w = roi_w > 1.0f ? roi_w : 1.0f;
    roi_h = roi_h > 1.0f ? roi_h : 1.0f;
    return make_float4(start_w, start_h, roi_w / 7.0f, roi_h / 7.0f);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The cells that the 14 samples of an axis touch: samples grow with (bin, sample), so the first tap of the first
// valid sample and the last tap of the last valid one bound them.  False when every sample is outside [-1, n].
__device__ __forceinline__ bool axis_range(float start, float bin, int n, int &cmin, int &cmax)
{
    float vfirst = 0.0f, vlast = 0.0f;
    bool any = false;
#pragma unroll
    for (int p = 0; p < 7; ++p)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float v = start + (float)p * bin + ((float)s + 0.5f) * bin / 2.0f; // the forward's expression
            const bool ok = !(v < -1.0f || v > (float)n);
            vfirst = (ok && !any) ? v : vfirst;
            vlast = ok ? v : vlast;
            any = any || ok;
        }
    cmin = axis_sample(vfirst, n).lo;
    cmax = axis_sample(vlast, n).hi;
    return any;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// grad_out of one RoI as the A operand of the ROW contraction (v_mfma_f32_16x16x4_f32: A[i][k] in lane i + 16 k):
// row i = 4 g + v stands for (channel v / 2, bin column pw = g + 4 (v % 2)), k for bin row ph = k (lo) and k + 4 (hi);
// rows and columns that do not exist (pw = 7, ph = 7) are zero.  Two dword loads per lane and RoI, each wave
// instruction a gather inside two 196-byte runs.
struct RabG {
    float lo, hi;
};

struct RabLane { // per-lane constants of the gather
    int off_lo, off_hi;
    bool ok_lo, ok_hi;
};

__device__ __forceinline__ RabLane rab_lane(int lane, int ch1_off)
{
    const int i = lane & 15, k = lane >> 4;
    const int v = i & 3, pw = (i >> 2) + 4 * (v & 1);
    RabLane L;
    L.ok_lo = pw < 7;
    L.ok_hi = pw < 7 && k < 3;
    const int col = (v >> 1) * ch1_off + (L.ok_lo ? pw : 0);
    L.off_lo = col + k * 7;
    L.off_hi = col + (L.ok_hi ? k + 4 : k) * 7;
    return L;
}

__device__ __forceinline__ void rab_load(RabG &g, const float *__restrict__ gb, int j, int C, const RabLane &L)
{
#ifdef RAB_EXP_NOLOAD
    g.lo = g.hi = 1.0f; // experiment: the walk without grad_out
    return;
#endif
    const float *gp = gb + (long long)j * C * 49;
    // Issued by hand and waited for by hand (rab_wait): the compiler's own counter bookkeeping gives up on a rotating
    // set of loads in a loop with exits and waits for ALL of them before every second RoI.
    asm volatile("global_load_dword %0, %1, %2" : "=v"(g.lo) : "v"(L.off_lo * 4), "s"(gp));
    asm volatile("global_load_dword %0, %1, %2" : "=v"(g.hi) : "v"(L.off_hi * 4), "s"(gp));
}

// The loads of `g` are followed by those of RAB_DEPTH - 1 other RoIs, two each, and loads return in order.
__device__ __forceinline__ void rab_wait(RabG &g)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(g.lo), "+v"(g.hi) : "n"(2 * (RAB_DEPTH - 1)));
}

// Leaving the walk: the compiler knows nothing of the loads issued by hand, so the registers of G[] must stay
// allocated until every one of them (the dummy re-loads behind the last RoI included) has landed -- a load landing
// later would overwrite whatever the allocator had put there meanwhile.  All of G[] is an operand of the drain.
__device__ __forceinline__ void rab_drain(RabG (&G)[RAB_DEPTH])
{
    static_assert(RAB_DEPTH == 6, "operand list of the drain");
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(G[0].lo), "+v"(G[0].hi), "+v"(G[1].lo), "+v"(G[1].hi), "+v"(G[2].lo), "+v"(G[2].hi),
                   "+v"(G[3].lo), "+v"(G[3].hi), "+v"(G[4].lo), "+v"(G[4].hi), "+v"(G[5].lo), "+v"(G[5].hi));
}

// Where the planes of a workgroup live in LDS: [first channels of the nw waves: nw planes][dummy: 16 B per lane]
// [second channels: nw planes][dummy], a plane = rows of SX floats (SX a multiple of 4, so that the four cells a lane
// owns in a block are one 16-byte access).  Lanes li = 0..7 of an access are 8 consecutive rows: they fall into 8
// different 16-byte bank groups either because SX / 4 is odd or, when SX is a multiple of 32 (`swz`), because the
// 16-byte group index of a cell is XORed with (row & 7).
struct RabPlanes {
    int SX;        // floats per plane row
    int swz;       // 1: byte offset inside a row ^= (row & 7) << 4
    unsigned pb;   // bytes per plane
    unsigned ch1;  // bytes from a wave's first plane to its second (= from the first dummy area to the second)
    int zero_ent;  // table entry whose 8 floats are zero
};

__device__ __forceinline__ unsigned rab_cell(const RabPlanes &PL, int y, int x)
{
    return (unsigned)(y * PL.SX * 4) + ((unsigned)(x * 4) ^ (PL.swz ? (unsigned)(y & 7) << 4 : 0u));
}

// One RoI, two channels, lane = (q = lane / 16, li = lane % 16).  Per block of 16 ROWS: B = Wy (lane (q, li) reads
// Wy[q][yb + li] and Wy[q + 4][yb + li] from the row table; entry 7 is zero), two MFMAs leave
// T[pw][y] = sum_ph G[ph][pw] Wy[ph][y] as D rows, i.e. lane (q, li) holds T[q][y], T[q + 4][y] of channel 0 and of
// channel 1 -- which is the B operand layout of the column contraction (B[k][j] in lane j + 16 k).  Per block of 16
// COLUMNS: A = Wx^T (lane (q, li) reads Wx[q][x], Wx[q + 4][x] of column x = xb + li from the column table, the zero
// entry when x is outside the footprint), two MFMAs per channel leave D[x][y] = sum_pw Wx[pw][x] T[pw][y] as the four
// columns xb + 4 q + v of row yb + li per lane: four neighbouring cells of one plane row = ONE 16-byte LDS access per
// channel to read them, one to write them back (blocks start on a multiple of 4 columns).  Lanes whose row or column
// group lies outside the footprint go to a dummy; columns outside it inside a group add the zero the zero entry gives.
__device__ __forceinline__ void rab_scatter(const RabG &g, const RabLane &L, const float *s_tab, const RabPlanes &PL,
                                            int e_off, int e_ymin, int e_ny, int e_xmin, int e_nx, int q, int li,
                                            char *lds, unsigned wave_base, unsigned dummy)
{
#ifdef RAB_EXP_NOSCATTER
    asm volatile("" ::"v"(g.lo), "v"(g.hi)); // experiment: the walk's skeleton (loads, waits, RoI bookkeeping) alone
    return;
#endif
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    const float glo = L.ok_lo ? g.lo : 0.0f, ghi = L.ok_hi ? g.hi : 0.0f;
    const int shift = e_xmin & 3, x0 = e_xmin - shift, nxe = e_nx + shift; // blocks start at column x0 of the plane
    for (int yb = 0; yb < e_ny; yb += 16) {
        const int yr = yb + li;
        const float *wp = s_tab + (e_off + min(yr, e_ny - 1)) * 8 + q;
        f32x4 t = __builtin_amdgcn_mfma_f32_16x16x4f32(glo, wp[0], z, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x4f32(ghi, wp[4], t, 0, 0, 0);
        const int y = e_ymin + yr;
        const unsigned rowb = wave_base + (unsigned)(y * PL.SX * 4);
        const unsigned sw = PL.swz ? (unsigned)(y & 7) << 4 : 0u;
        const int groups_here = yr < e_ny ? nxe - 4 * q : 0; // > xb: this lane's four columns take part
        for (int xb = 0; xb < nxe; xb += 16) {
            // (reading the column entry of block xb + 16 here, one block ahead of its use, was measured: 3.38 against
            // 3.25 ms -- the extra read behind the last block costs more than the latency the others hide)
            const int xi = xb + li - shift;
            const float *ap = s_tab + ((unsigned)xi < (unsigned)e_nx ? e_off + e_ny + xi : PL.zero_ent) * 8 + q;
            const float alo = ap[0], ahi = ap[4];
            f32x4 acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(alo, t.x, z, 0, 0, 0);
            f32x4 acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(alo, t.z, z, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ahi, t.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ahi, t.w, acc1, 0, 0, 0);
#ifdef RAB_EXP_NORMW
            asm volatile("" ::"v"(acc0), "v"(acc1)); // experiment: both contractions without the LDS read-modify-write
            continue;
#endif
            const unsigned col = (unsigned)((x0 + xb + 4 * q) * 4);
            const unsigned a0 = xb < groups_here ? rowb + (col ^ sw) : dummy;
            f32x4 *p0 = (f32x4 *)(lds + a0), *p1 = (f32x4 *)(lds + a0 + PL.ch1);
            const f32x4 o0 = *p0, o1 = *p1;
            *p0 = o0 + acc0;
            *p1 = o1 + acc1;
        }
    }
}

// 64 candidate RoIs, one per lane (live: to be walked; rec: its record; jidx: its row among the image's RoIs; every
// wave of the workgroup holds the same values): footprints and table sizes, then batch after batch -- as many RoIs as
// the table space holds -- the tables are built by all waves and every wave walks the batch for its two channels.
// The planes (layout PL) start at cell (oy, ox) of the H x W map.  Contains workgroup barriers.
__device__ __forceinline__ void rab_walk64(bool live, const float4 rec, int jidx, int n_cand, int H, int W,
                                           const RabPlanes &PL, int oy,
                                           int ox, float *s_tab, int cap, int nw, int wave, int lane, bool has0,
                                           const float *__restrict__ gb, int C, const RabLane &GL, char *lds,
                                           unsigned wave_base, unsigned dummy)
{
    const int q = lane >> 4, li = lane & 15;
    int ymin = 0, ymax = 0, xmin = 0, xmax = 0;
    live = axis_range(rec.y, rec.w, H, ymin, ymax) && live;
    live = axis_range(rec.x, rec.z, W, xmin, xmax) && live;
    const int ny = ymax - ymin + 1, nx = xmax - xmin + 1;
    const int ent = live ? ny + nx : 0; // <= H + W <= cap
    int pre = ent; // inclusive prefix over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(pre, o);
        pre += lane >= o ? t : 0;
    }
    int done = 0, base = 0;
    while (done < n_cand) {
        // ---- the batch: candidates done .. done + m - 1, as many as the table space holds (at least one)
        const unsigned long long fit = __ballot(lane >= done && pre - base <= cap);
        const unsigned long long miss = ~fit >> done;
        int m = miss ? __ffsll((long long)miss) - 1 : 64 - done;
        m = min(m, n_cand - done);
        const int my_off = pre - ent - base; // first entry of this lane's RoI
        __syncthreads(); // the previous batch's tables have been used
#ifdef RAB_EXP_NOTABLES
        if (false) // experiment: the walk on whatever the table space holds
#endif
        for (int g = wave * 4; g < m; g += nw * 4) {
            // ---- tables of RoIs done + g .. + 3: 16 lanes per RoI, lane = (axis, bin) for the scatter
            const int src = min(done + g + (lane >> 4), 63);
            const bool mine = g + (lane >> 4) < m;
            const int src_ent = __shfl(ent, src); // (every lane takes part: the source lanes are anywhere in the wave)
            const int e_ent = mine ? src_ent : 0;
            const int e_off = __shfl(my_off, src);
            const int e_ymin = __shfl(ymin, src), e_ny = __shfl(ny, src), e_xmin = __shfl(xmin, src);
            const float e_sx = __shfl(rec.x, src), e_sy = __shfl(rec.y, src);
            const float e_bw = __shfl(rec.z, src), e_bh = __shfl(rec.w, src);
            float4 *t4 = (float4 *)s_tab + e_off * 2;
            for (int i = lane & 15; i < e_ent * 2; i += 16) t4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int axis = (lane >> 3) & 1, p = lane & 7;
            if (e_ent > 0 && p < 7) {
                const float start = axis ? e_sx : e_sy, bin = axis ? e_bw : e_bh;
                const float k = axis ? 1.0f : 0.25f;
                float *col = s_tab + (e_off + (axis ? e_ny - e_xmin : -e_ymin)) * 8 + p;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const AxisSample a = axis_sample(start + (float)p * bin + ((float)s + 0.5f) * bin / 2.0f, axis ? W : H);
                    if (a.ok) { // (a sample outside the map sends nothing)
                        col[a.lo * 8] += a.wlo * k;
                        col[a.hi * 8] += a.whi * k;
                    }
                }
            }
        }
        __syncthreads();
        if (has0) {
            // ---- this wave's two channels over the batch's RoIs, grad_out of the next RAB_DEPTH RoIs in flight
            unsigned long long todo = __ballot(ent > 0) & (m >= 64 ? ~0ull : (((1ull << m) - 1ull) << done));
            auto next = [&]() -> int {
                if (!todo) return -1;
                const int e = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                return e;
            };
            const int e_first = next();
            if (e_first >= 0) {
                RabG G[RAB_DEPTH];
                int E[RAB_DEPTH];
                E[0] = e_first;
#pragma unroll
                for (int d = 1; d < RAB_DEPTH; ++d) E[d] = next();
#pragma unroll
                for (int d = 0; d < RAB_DEPTH; ++d) rab_load(G[d], gb, __builtin_amdgcn_readlane(jidx, E[d] >= 0 ? E[d] : e_first), C, GL);
                bool more = true;
                while (more) {
#pragma unroll
                    for (int d = 0; d < RAB_DEPTH; ++d) {
                        if (E[d] < 0) {
                            more = false;
                            break;
                        }
                        const int e = E[d];
                        const int e_off = __builtin_amdgcn_readlane(my_off, e);
                        const int e_ymin = __builtin_amdgcn_readlane(ymin, e), e_ny = __builtin_amdgcn_readlane(ny, e);
                        const int e_xmin = __builtin_amdgcn_readlane(xmin, e), e_nx = __builtin_amdgcn_readlane(nx, e);
                        rab_wait(G[d]);
                        rab_scatter(G[d], GL, s_tab, PL, e_off, e_ymin - oy, e_ny, e_xmin - ox, e_nx, q, li, lds, wave_base, dummy);
                        E[d] = next();
                        rab_load(G[d], gb, __builtin_amdgcn_readlane(jidx, E[d] >= 0 ? E[d] : e_first), C, GL);
                    }
                }
                rab_drain(G);
            }
        }
        base = __builtin_amdgcn_readlane(pre, done + m - 1);
        done += m;
    }
}

// Workgroup = (image, chunk of 2 nw channels, RoI split), nw waves: WAVE w OWNS CHANNELS c0 + 2w, c0 + 2w + 1 and
// their gradient planes in LDS (two planes, layout RabPlanes), so no two waves ever add into the same plane, and
// inside a wave the lanes are distinct groups of four neighbouring CELLS of the RoI's footprint: the scatter needs no
// atomics, it is ds_read_b128, add, ds_write_b128 per channel.  (LDS float atomics are what the first version used: ds_add_f32 costs ~130 cycles per wave
// instruction on gfx950 whether or not lanes collide -- 68 ms at the headline shape, against 3.4 ms for this form.)
//
// RoIAlign's weights are separable: grad_feat[y][x] += sum_pw Wx[pw][x] * (sum_ph G[ph][pw] * Wy[ph][y]), Wy / Wx =
// the weight that the two samples of bin row ph / column pw put on cell row y / column x.  Per RoI the workgroup
// builds both as TABLES in LDS once (an entry = the 7 weights of one cell row or column; rows carry the 1/4 of the
// 2 x 2 average), for as many RoIs at a time as the table space holds; every wave then walks those RoIs for its two
// channels (rab_scatter: both contractions as 16 x 16 x 4 fp32 MFMAs, the row one for the two channels at once;
// grad_out of the next RAB_DEPTH RoIs in flight, because a wave's RoIs are a serial chain of dependent loads
// otherwise).
// HBM sees grad_out once plus one write of the planes; one workgroup per (image, chunk) -> deterministic sums.
#define RAB_TAB_MAX_ENTRIES 2048
__global__ void __launch_bounds__(512)
roi_align_plane_backward_kernel(const float *__restrict__ grad_out, int C, int H, int W,
                                const float4 *__restrict__ props, const int32_t *__restrict__ count, int P,
                                float scale, int level, int n_levels, int k_min, int k_max, int chunks,
                                int splits, int B, int SX, int swz, int cap, float *__restrict__ grad_feat)
{
    // planes and dummies (RabPlanes), then the tables: cap entries of 8 floats and the zero entry
    extern __shared__ __align__(16) float s_dyn[];
    __shared__ int s_off, s_cnt;
    const int nw = blockDim.x >> 6;
    RabPlanes PL;
    PL.SX = SX;
    PL.swz = swz;
    PL.pb = (unsigned)(H * SX * 4);
    PL.ch1 = (unsigned)nw * PL.pb + 1024u;
    PL.zero_ent = cap;
    const unsigned planes_bytes = 2u * PL.ch1;
    float *const s_tab = s_dyn + planes_bytes / 4;
    const int chunk = blockIdx.x % chunks;
    const int grp = blockIdx.x / chunks;
    const int split = grp % splits;
    const int b = grp / splits;
    const int c0 = chunk * 2 * nw;
    const int nc = min(2 * nw, C - c0);
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
    for (unsigned t = threadIdx.x; t < planes_bytes / 16; t += blockDim.x)
        reinterpret_cast<float4 *>(s_dyn)[t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (threadIdx.x < 8) s_tab[cap * 8 + threadIdx.x] = 0.0f;
    __syncthreads();
    const int cnt = s_cnt;
    const int per = (cnt + splits - 1) / splits;
    const int j_end = min(cnt, (split + 1) * per);
    const float4 *pb = props + (long long)b * P;
    const int ch0 = c0 + 2 * wave;
    const bool has0 = ch0 < C, has1 = ch0 + 1 < C;
    const float *gb = grad_out + ((long long)s_off * C + min(ch0, C - 1)) * 49;
    const RabLane GL = rab_lane(lane, has1 ? 49 : 0);
    char *const lds = (char *)s_dyn;
    const unsigned dummy = (unsigned)nw * PL.pb + (unsigned)lane * 16u; // masked lanes add here (each its own 16 bytes)
    const unsigned wave_base = (unsigned)wave * PL.pb;
    for (int jw = split * per; jw < j_end; jw += 64) {
        // ---- lane = candidate RoI jw + lane: footprint and table size (every wave computes the same values)
        bool live = jw + lane < j_end;
        float4 rec = make_float4(0.0f, 0.0f, 1.0f, 1.0f);
        if (live) {
            const float4 box = pb[jw + lane];
            if (n_levels > 1 && map_level(box, k_min, k_max) != level) live = false;
            rec = rab_record(box, scale);
        }
        rab_walk64(live, rec, jw + lane, min(64, j_end - jw), H, W, PL, 0, 0, s_tab, cap, nw, wave, lane, has0, gb, C, GL,
                   lds, wave_base, dummy);
    }
    __syncthreads();
    // ---- the planes leave row by row: a wave takes (channel, row) pairs, lanes = cells of the row
    float *out = grad_feat + ((long long)b * C + c0) * HW;
    for (int r = wave; r < nc * H; r += nw) {
        const int c = r / H, y = r - c * H;
        const char *pl = lds + ((c & 1) ? PL.ch1 : 0u) + (unsigned)(c >> 1) * PL.pb;
        float *o = out + (long long)c * HW + (long long)y * W;
        for (int x = lane; x < W; x += 64) {
            const float v = *(const float *)(pl + rab_cell(PL, y, x));
            if (splits == 1)
                o[x] = v;
            else if (v != 0.0f)
                unsafeAtomicAdd(o + x, v);
        }
    }
}

// ---- planes too large for LDS (FPN P3 100 x 168, P2 200 x 336): windows of the gradient plane -----------------------
// The forward's windows (roi_common.h: RAW_WIN x RAW_WIN cells every RAW_STEP; a RoI belongs to the window that holds
// its first tap and fits when its last tap is inside too) with the scheme above run on each: workgroup = (window,
// image, channel split), RABW_PAIRS waves = channel pairs whose 64 x 64 gradient WINDOWS live in LDS.  The window's RoIs
// are listed once (in index order), then chunk after chunk of 2 RABW_PAIRS channels is zeroed, walked (rab_walk64) and
// added to grad_feat with global float atomics -- windows overlap, so a cell collects up to four partial sums (and
// whatever the oversize RoIs send through roi_align_backward_atomic_kernel, launched with the same test inverted);
// the host entry zeroes grad_feat first.  HBM sees grad_out once; the atomics are coalesced 256-byte rows.
#define RABW_PAIRS 4
#define RABW_THREADS (64 * RABW_PAIRS)
#define RABW_LIST 2048 // RoIs of one image examined per pass (a window's list cannot be longer)
__global__ void __launch_bounds__(RABW_THREADS)
roi_align_window_backward_kernel(const float *__restrict__ grad_out, int C, int H, int W,
                                 const float4 *__restrict__ props, const int32_t *__restrict__ count, int P,
                                 float scale, int level, int n_levels, int k_min, int k_max, int nwy, int nwx,
                                 int csplit, int B, int cap, float *__restrict__ grad_feat)
{
    // windows of RAW_WIN x RAW_WIN cells as planes (RabPlanes, rows of RAW_WIN floats, swizzled) and dummies, then the
    // tables and the zero entry
    extern __shared__ __align__(16) float s_dyn[];
    __shared__ int s_list[RABW_LIST];
    __shared__ int s_wave_sums[RABW_PAIRS];
    __shared__ int s_off, s_cnt;
    static_assert(RAW_WIN % 32 == 0, "swizzled rows are whole 32-float blocks");
    RabPlanes PL;
    PL.SX = RAW_WIN;
    PL.swz = 1;
    PL.pb = RAW_WIN * RAW_WIN * 4u;
    PL.ch1 = RABW_PAIRS * PL.pb + 1024u;
    PL.zero_ent = cap;
    const unsigned planes_bytes = 2u * PL.ch1;
    float *const s_tab = s_dyn + planes_bytes / 4;
    const int cs = blockIdx.x % csplit;
    const int grp = blockIdx.x / csplit;
    const int wid = grp % (nwy * nwx);
    const int b = grp / (nwy * nwx);
    const int wy0 = (wid / nwx) * RAW_STEP, wx0 = (wid % nwx) * RAW_STEP;
    const int wh = min(RAW_WIN, H - wy0), ww = min(RAW_WIN, W - wx0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int HW = H * W;
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
    if (threadIdx.x < 8) s_tab[cap * 8 + threadIdx.x] = 0.0f;
    __syncthreads();
    const int cnt = s_cnt;
    const float4 *pb = props + (long long)b * P;
    char *const lds = (char *)s_dyn;
    const unsigned dummy = RABW_PAIRS * PL.pb + (unsigned)lane * 16u;
    const unsigned wave_base = (unsigned)wave * PL.pb;
    const int chunks = (C + 2 * RABW_PAIRS - 1) / (2 * RABW_PAIRS);
    for (int seg = 0; seg < cnt; seg += RABW_LIST) {
        // ---- the window's RoIs among seg .. seg + RABW_LIST - 1, in index order
        int n_list = 0;
        for (int i0 = seg; i0 < min(cnt, seg + RABW_LIST); i0 += RABW_THREADS) {
            const int i = i0 + threadIdx.x;
            bool mine = false;
            if (i < min(cnt, seg + RABW_LIST)) {
                const float4 box = pb[i];
                const int lv = n_levels > 1 ? map_level(box, k_min, k_max) : 0;
                const RoiWindow rw = roi_window(roi_record(box, scale), H, W, nwy, nwx);
                mine = lv == level && rw.fits && rw.win == wid;
            }
            int total;
            const int pos = n_list + frcnn_block_excl_count(mine, s_wave_sums, &total);
            if (mine) s_list[pos] = i;
            n_list += total;
        }
        __syncthreads();
        for (int ch = cs; ch < (n_list ? chunks : 0); ch += csplit) {
            const int c0 = ch * 2 * RABW_PAIRS;
            const int nc = min(2 * RABW_PAIRS, C - c0);
            const int ch0 = c0 + 2 * wave;
            const bool has0 = ch0 < C, has1 = ch0 + 1 < C;
            const float *gb = grad_out + ((long long)s_off * C + min(ch0, C - 1)) * 49;
            const RabLane GL = rab_lane(lane, has1 ? 49 : 0);
            for (unsigned t = threadIdx.x; t < planes_bytes / 16; t += RABW_THREADS)
                reinterpret_cast<float4 *>(s_dyn)[t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            __syncthreads();
            for (int lw = 0; lw < n_list; lw += 64) {
                const bool live = lw + lane < n_list;
                const int j = live ? s_list[lw + lane] : 0;
                const float4 rec = live ? roi_record(pb[j], scale) : make_float4(0.0f, 0.0f, 1.0f, 1.0f);
                rab_walk64(live, rec, j, min(64, n_list - lw), H, W, PL, wy0, wx0, s_tab, cap, RABW_PAIRS, wave, lane,
                           has0, gb, C, GL, lds, wave_base, dummy);
            }
            __syncthreads();
            // ---- add the windows to the map: rows of ww cells, coalesced
            for (int t = threadIdx.x; t < nc * wh * RAW_WIN; t += RABW_THREADS) {
                const int c = t / (wh * RAW_WIN), rem = t - c * (wh * RAW_WIN);
                const int y = rem / RAW_WIN, x = rem - y * RAW_WIN;
                const float v = *(const float *)(lds + ((c & 1) ? PL.ch1 : 0u) + (unsigned)(c >> 1) * PL.pb + rab_cell(PL, y, x));
                if (x < ww && v != 0.0f)
                    unsafeAtomicAdd(grad_feat + ((long long)b * C + c0 + c) * HW + (long long)(wy0 + y) * W + wx0 + x, v);
            }
            __syncthreads();
        }
    }
}

// Any plane size / pooled size / sampling ratio: block = (RoI split, 32-channel chunk, image), global float
// atomics per tap (grad_feat zeroed by the host entry).
__global__ void __launch_bounds__(256)
roi_align_backward_atomic_kernel(const float *__restrict__ grad_out, int C, int H, int W,
                                 const float4 *__restrict__ props, const int32_t *__restrict__ count, int P,
                                 int pooled_h, int pooled_w, float scale, int sampling_ratio, int level,
                                 int n_levels, int k_min, int k_max, int nwy, int nwx, float *__restrict__ grad_feat)
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
        // behind the window kernel: only the RoIs that fit no window (the same test, inverted)
        if (nwy > 0 && roi_window(roi_record(box, scale), H, W, nwy, nwx).fits) continue;
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
    // planes in LDS, one wave per channel pair: worth it from two waves (four channels) per workgroup on.
    // Plane rows (RabPlanes): W rounded up to 4 floats with an odd number of 16-byte groups, or to a multiple of 32
    // with the groups swizzled by the row -- whichever is shorter.
    int sx_odd = (W + 3) / 4 * 4;
    if ((sx_odd / 4) % 2 == 0) sx_odd += 4;
    const int sx_swz = (W + 31) / 32 * 32;
    const int swz = sx_swz < sx_odd ? 1 : 0;
    const int SX = swz ? sx_swz : sx_odd;
    const long long pair_bytes = (long long)H * SX * 8;
    const long long tab_min = ((long long)(H + W) * 32 > 8192 ? (long long)(H + W) * 32 : 8192) + 32; // + the zero entry
    int nw = 8;
    while (nw >= 2 && nw * pair_bytes + 2048 + tab_min > RAB_LDS_BUDGET) --nw;
    if (pooled_h == 7 && pooled_w == 7 && sampling_ratio == 2 && nw >= 2 && (long long)P * 49 < (1ll << 30)) {
        if (2 * nw > C) nw = (C + 1) / 2;
        const int CC = 2 * nw;
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
        const long long planes_bytes = nw * pair_bytes + 2048; // + the two dummy areas
        long long cap = (RAB_LDS_BUDGET - planes_bytes - 32) / 32;
        if (cap > RAB_TAB_MAX_ENTRIES) cap = RAB_TAB_MAX_ENTRIES;
        FRCNN_REQUIRE(cap >= H + W, "roi_align_backward: internal: table space %lld < %d entries", cap, H + W);
        const size_t lds = (size_t)planes_bytes + (size_t)(cap + 1) * 32;
        static size_t granted[FRCNN_MAX_DEVICES];
        hipError_t e = frcnn_reserve_dynamic_lds((const void *)roi_align_plane_backward_kernel, lds, granted);
        FRCNN_REQUIRE(e == hipSuccess, "roi_align_backward: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        FRCNN_REQUIRE((long long)B * chunks * splits < (1ll << 31), "roi_align_backward: grid too large");
        hipLaunchKernelGGL(roi_align_plane_backward_kernel, dim3((unsigned)(B * chunks * splits)), dim3(64 * nw),
                           lds, st, grad_out, C, H, W, (const float4 *)props, count, P, spatial_scale, level,
                           n_levels, k_min, k_max, chunks, splits, B, SX, swz, (int)cap, grad_feat);
        FRCNN_CHECK_LAUNCH("roi_align_plane_backward");
        return FRCNN_OK;
    }
    hipError_t e = hipMemsetAsync(grad_feat, 0, plane_elems * sizeof(float), st);
    FRCNN_REQUIRE(e == hipSuccess, "roi_align_backward: memset failed: %s", hipGetErrorString(e));
    int nwy = 0, nwx = 0;
    if (pooled_h == 7 && pooled_w == 7 && sampling_ratio == 2 && (long long)P * 49 < (1ll << 30) &&
        (long long)H * W < (1ll << 28)) {
        // plane too large for LDS: windows of it (fine pyramid levels, where RoIs are small)
        nwy = H <= RAW_WIN ? 1 : (H - RAW_WIN + RAW_STEP - 1) / RAW_STEP + 1;
        nwx = W <= RAW_WIN ? 1 : (W - RAW_WIN + RAW_STEP - 1) / RAW_STEP + 1;
        if ((long long)nwy * nwx * B >= (1ll << 24)) {
            nwy = nwx = 0;
        } else {
            const int chunks = (C + 2 * RABW_PAIRS - 1) / (2 * RABW_PAIRS);
            int csplit = (int)((1024 + (long long)nwy * nwx * B - 1) / ((long long)nwy * nwx * B));
            csplit = csplit < 1 ? 1 : (csplit > chunks ? chunks : csplit);
            const long long planes_bytes = 2ll * (RABW_PAIRS * RAW_WIN * RAW_WIN * 4 + 1024); // windows + dummy areas
            // beside the windows: the RoI list and the dummies (static), and tables for at least one RoI of a window
            long long cap = (RAB_LDS_BUDGET - planes_bytes - RABW_LIST * 4 - 64 - 32) / 32;
            if (cap > RAB_TAB_MAX_ENTRIES) cap = RAB_TAB_MAX_ENTRIES;
            FRCNN_REQUIRE(cap >= 2 * RAW_WIN, "roi_align_backward: internal: table space %lld entries", cap);
            const size_t lds = (size_t)planes_bytes + (size_t)(cap + 1) * 32;
            static size_t grantedw[FRCNN_MAX_DEVICES];
            e = frcnn_reserve_dynamic_lds((const void *)roi_align_window_backward_kernel, lds, grantedw);
            FRCNN_REQUIRE(e == hipSuccess, "roi_align_backward: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
            hipLaunchKernelGGL(roi_align_window_backward_kernel, dim3((unsigned)(nwy * nwx * B * csplit)),
                               dim3(RABW_THREADS), lds, st, grad_out, C, H, W, (const float4 *)props, count, P,
                               spatial_scale, level, n_levels, k_min, k_max, nwy, nwx, csplit, B, (int)cap,
                               grad_feat);
            FRCNN_CHECK_LAUNCH("roi_align_window_backward");
        }
    }
    const int chunks = (C + 31) / 32;
    FRCNN_REQUIRE(chunks <= 65535, "roi_align_backward: too many channels");
    const int splits = P < 256 ? P : 256;
    hipLaunchKernelGGL(roi_align_backward_atomic_kernel, dim3(splits, chunks, B), dim3(256), 0, st, grad_out, C, H, W,
                       (const float4 *)props, count, P, pooled_h, pooled_w, spatial_scale, sampling_ratio, level,
                       n_levels, k_min, k_max, nwy, nwx, grad_feat);
    FRCNN_CHECK_LAUNCH("roi_align_backward_atomic");
    return FRCNN_OK;
}
