// topk.hip -- K2: RPN output flattening and per-image sorted top-k.
//
// Replaces RPN/rpn.py:56-57 (permute/flatten/cat) and RPN/rpn.py:65-66 (torch.topk per image).
//
// Top-k = radix SELECT of the k-th largest key (4 x 8-bit MSB-first histogram passes over the
// image's scores, LDS histograms, wave-aggregated LDS atomics), an index-ordered compaction of
// the k winners, then a stable 4-pass LSD radix SORT of the winners (wave-private digit counters
// in LDS, 64-wide ballot "match" ranking, ping-pong buffers that stay L2-resident).  One
// 1024-thread workgroup (16 wave64) owns one image; the batch is the grid.  Everything is
// integer/key work: the result is bit-exact by construction and the ordering rule is
// "descending key, ties by ascending index" (the stable sort of an index-ordered list).
#include "common.h"

// ---- flatten ----------------------------------------------------------------------------------
// cls (B,na,h,w) -> scores[b, off + cell*na + a]; reg (B,4na,h,w) -> deltas[b, off + cell*na + a].
// A block handles 64 cells: NCHW reads are coalesced along the cell index (x fastest), the
// transposed writes go through LDS so that they are coalesced along (cell, a).
#ifndef FLAT_CELLS
#define FLAT_CELLS 64
#endif

__global__ void __launch_bounds__(256)
rpn_flatten_kernel(const float *__restrict__ cls, const float *__restrict__ reg, int na, int hw,
                   int level_offset, int A_total, float *__restrict__ scores,
                   float4 *__restrict__ deltas)
{
    extern __shared__ float s_tile[]; // [5*na][FLAT_CELLS + 1]: rows 0..na-1 cls, then 4*na reg
    const int b = blockIdx.y;
    const int cell0 = blockIdx.x * FLAT_CELLS;
    const int ncell = min(FLAT_CELLS, hw - cell0);
    const int rows = 5 * na;
    const float *cls_b = cls + (long long)b * na * hw;
    const float *reg_b = reg + (long long)b * 4 * na * hw;
    for (int t = threadIdx.x; t < rows * FLAT_CELLS; t += blockDim.x) {
        int r = t / FLAT_CELLS;
        int c = t - r * FLAT_CELLS;
        if (c < ncell) {
            float v = r < na ? cls_b[(long long)r * hw + cell0 + c]
                             : reg_b[(long long)(r - na) * hw + cell0 + c];
            s_tile[r * (FLAT_CELLS + 1) + c] = v;
        }
    }
    __syncthreads();
    const long long out0 = (long long)b * A_total + level_offset + (long long)cell0 * na;
    for (int t = threadIdx.x; t < ncell * na; t += blockDim.x) {
        int c = t / na;
        int a = t - c * na;
        scores[out0 + t] = s_tile[a * (FLAT_CELLS + 1) + c];
        float4 d;
        d.x = s_tile[(na + 4 * a + 0) * (FLAT_CELLS + 1) + c];
        d.y = s_tile[(na + 4 * a + 1) * (FLAT_CELLS + 1) + c];
        d.z = s_tile[(na + 4 * a + 2) * (FLAT_CELLS + 1) + c];
        d.w = s_tile[(na + 4 * a + 3) * (FLAT_CELLS + 1) + c];
        deltas[out0 + t] = d;
    }
}

extern "C" int frcnn_rpn_flatten_f32(const float *cls, const float *reg, int B, int na, int h, int w,
                                     int level_offset, int A_total, float *scores, float *deltas,
                                     void *stream)
{
    FRCNN_REQUIRE(B > 0 && na > 0 && h > 0 && w > 0, "rpn_flatten: bad shape");
    FRCNN_REQUIRE(cls && reg && scores && deltas, "rpn_flatten: null pointer");
    long long hw = (long long)h * w;
    FRCNN_REQUIRE(level_offset >= 0 && level_offset + hw * na <= A_total,
                  "rpn_flatten: level does not fit in A_total");
    size_t lds = (size_t)5 * na * (FLAT_CELLS + 1) * sizeof(float);
    FRCNN_REQUIRE(lds <= 64 * 1024, "rpn_flatten: too many anchors per cell (%d)", na);
    int blocks = (int)((hw + FLAT_CELLS - 1) / FLAT_CELLS);
    hipLaunchKernelGGL(rpn_flatten_kernel, dim3(blocks, B), dim3(256), lds, (hipStream_t)stream, cls,
                       reg, na, (int)hw, level_offset, A_total, scores, (float4 *)deltas);
    FRCNN_CHECK_LAUNCH("rpn_flatten");
    return FRCNN_OK;
}

// ---- top-k ------------------------------------------------------------------------------------
#define TOPK_THREADS 1024
#define TOPK_WAVES 16

// Lanes of the wave that are active and hold the same 8-bit digit as this lane ("match any"),
// built from 8 ballots: constant cost whatever the digit distribution.
__device__ __forceinline__ unsigned long long wave_match8(unsigned digit, bool active)
{
    unsigned long long peers = __ballot(active);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
        bool one = (digit >> bit) & 1u;
        unsigned long long bm = __ballot(one);
        peers &= one ? bm : ~bm;
    }
    return peers;
}

// Adds 1 to hist[digit] for every active lane: one LDS atomic per distinct digit in the wave.
__device__ __forceinline__ void wave_hist_add(unsigned *hist, unsigned digit, bool active)
{
    unsigned long long peers = wave_match8(digit, active);
    int lane = threadIdx.x & 63;
    if (active && (peers & ((1ull << lane) - 1ull)) == 0) atomicAdd(&hist[digit], (unsigned)__popcll(peers));
}

__global__ void __launch_bounds__(TOPK_THREADS)
topk_kernel(const float *__restrict__ scores, int n, int k, float *__restrict__ val_out,
            int32_t *__restrict__ idx_out, uint32_t *__restrict__ ws)
{
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_cnt[TOPK_WAVES][256];
    __shared__ unsigned s_digit_base[256];
    __shared__ unsigned s_prefix, s_remaining;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = blockIdx.x;
    const float *sc = scores + (long long)b * n;
    uint32_t *keyA = ws + (long long)b * 4 * k;
    uint32_t *idxA = keyA + k;
    uint32_t *keyB = idxA + k;
    uint32_t *idxB = keyB + k;

    // ---- 1. radix select: key of the k-th largest element, and how many of its equals to take
    unsigned prefix = 0, mask = 0, remaining = (unsigned)k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += 4 * TOPK_THREADS) { // four rounds of loads in flight
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * TOPK_THREADS + tid;
                v[u] = i < n ? sc[i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * TOPK_THREADS + tid;
                const unsigned key = frcnn_score_key(v[u]);
                const bool act = i < n && (key & mask) == prefix;
                const unsigned dg = (key >> shift) & 255u;
                if (__any(act)) wave_hist_add(s_hist, dg, act); // after the first pass few waves hold a candidate
            }
        }
        __syncthreads();
        if (wave == 0) {
            // suffix sums over the 256 bins, 4 bins per lane: S(d) = sum_{d' >= d} hist[d'].
            // The wanted digit is the largest d with S(d) >= remaining.
            unsigned h0 = s_hist[4 * lane], h1 = s_hist[4 * lane + 1], h2 = s_hist[4 * lane + 2],
                     h3 = s_hist[4 * lane + 3];
            unsigned sum = h0 + h1 + h2 + h3;
            unsigned incl = sum; // inclusive suffix sum over lanes (lane 63 first)
            for (int o = 1; o < 64; o <<= 1) {
                unsigned t = __shfl_down(incl, o);
                if (lane + o < 64) incl += t;
            }
            unsigned above = incl - sum; // bins of higher lanes
            unsigned s3 = above + h3, s2 = s3 + h2, s1 = s2 + h1, s0 = s1 + h0;
            // S is non-increasing in d; exactly one d has S(d) >= remaining > S(d+1)
            int d = -1;
            unsigned cum = 0;
            if (s3 >= remaining && above < remaining) { d = 4 * lane + 3; cum = above; }
            else if (s2 >= remaining && s3 < remaining) { d = 4 * lane + 2; cum = s3; }
            else if (s1 >= remaining && s2 < remaining) { d = 4 * lane + 1; cum = s2; }
            else if (s0 >= remaining && s1 < remaining) { d = 4 * lane; cum = s1; }
            if (d >= 0) {
                s_prefix = prefix | ((unsigned)d << shift);
                s_remaining = remaining - cum;
            }
        }
        __syncthreads();
        prefix = s_prefix;
        remaining = s_remaining;
        mask |= 255u << shift;
    }
    const unsigned T = prefix;        // key of the k-th largest
    const unsigned need_eq = remaining; // how many elements with key == T are selected (lowest idx)

    // ---- 2. index-ordered compaction of the k winners into (keyA, idxA).  A wave owns a contiguous
    // index range: it first counts its winners, one barrier publishes the 16 counts, then it writes
    // its range with ballot ranks -- no barrier per round of 1024 elements.
    {
        __shared__ int s_wgt[TOPK_WAVES], s_weq[TOPK_WAVES];
        const int wseg = ((n + TOPK_WAVES - 1) / TOPK_WAVES + 63) / 64 * 64;
        const int lo = min(wave * wseg, n), hi = min(lo + wseg, n);
        const unsigned long long lt = (1ull << lane) - 1ull;
        int cg = 0, ce = 0;
        for (int e0 = lo; e0 < hi; e0 += 256) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = e0 + u * 64 + lane < hi ? sc[e0 + u * 64 + lane] : 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = e0 + u * 64 + lane;
                const unsigned key = frcnn_score_key(v[u]);
                cg += __popcll(__ballot(i < hi && key > T));
                ce += __popcll(__ballot(i < hi && key == T));
            }
        }
        if (lane == 0) {
            s_wgt[wave] = cg;
            s_weq[wave] = ce;
        }
        __syncthreads();
        int gt_before = 0, eq_before = 0;
        for (int w = 0; w < wave; ++w) {
            gt_before += s_wgt[w];
            eq_before += s_weq[w];
        }
        int base = gt_before + min(eq_before, (int)need_eq); // winners at lower indices
        for (int e0 = lo; e0 < hi; e0 += 256) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = e0 + u * 64 + lane < hi ? sc[e0 + u * 64 + lane] : 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = e0 + u * 64 + lane;
                const unsigned key = frcnn_score_key(v[u]);
                const bool gt = i < hi && key > T;
                const bool eq = i < hi && key == T;
                const unsigned long long meq = __ballot(eq);
                const int eq_rank = eq_before + __popcll(meq & lt);
                const bool sel = gt || (eq && (unsigned)eq_rank < need_eq);
                const unsigned long long msel = __ballot(sel);
                if (sel) {
                    const int pos = base + __popcll(msel & lt);
                    keyA[pos] = key;
                    idxA[pos] = (uint32_t)i;
                }
                base += __popcll(msel);
                eq_before += __popcll(meq);
            }
        }
    }
    __syncthreads();

    // ---- 3. stable LSD radix sort, descending key (digit' = 255 - digit), 4 x 8 bits
    const int seg = ((k + TOPK_WAVES - 1) / TOPK_WAVES + 63) / 64 * 64; // per-wave contiguous segment
    const int seg_lo = min(wave * seg, k);
    const int seg_hi = min(seg_lo + seg, k);
    uint32_t *kin = keyA, *iin = idxA, *kout = keyB, *iout = idxB;
    for (int shift = 0; shift < 32; shift += 8) {
        for (int t = tid; t < TOPK_WAVES * 256; t += TOPK_THREADS) (&s_cnt[0][0])[t] = 0;
        __syncthreads();
        for (int e0 = seg_lo; e0 < seg_hi; e0 += 256) {
            unsigned kk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) kk[u] = e0 + u * 64 + lane < seg_hi ? kin[e0 + u * 64 + lane] : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (e0 + u * 64 < seg_hi) { // wave-uniform
                    const bool act = e0 + u * 64 + lane < seg_hi;
                    wave_hist_add(s_cnt[wave], 255u - ((kk[u] >> shift) & 255u), act);
                }
            }
        }
        __syncthreads();
        // per digit: exclusive prefix over waves, digit totals -> s_hist
        if (tid < 256) {
            unsigned run = 0;
            for (int w = 0; w < TOPK_WAVES; ++w) {
                unsigned c = s_cnt[w][tid];
                s_cnt[w][tid] = run;
                run += c;
            }
            s_hist[tid] = run;
        }
        __syncthreads();
        if (wave == 0) { // exclusive scan of the 256 digit totals: 4 digits per lane + wave scan
            unsigned v0 = s_hist[4 * lane], v1 = s_hist[4 * lane + 1], v2 = s_hist[4 * lane + 2],
                     v3 = s_hist[4 * lane + 3];
            unsigned sum = v0 + v1 + v2 + v3;
            unsigned incl = sum;
            for (int o = 1; o < 64; o <<= 1) {
                unsigned t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            unsigned excl = incl - sum;
            s_digit_base[4 * lane] = excl;
            s_digit_base[4 * lane + 1] = excl + v0;
            s_digit_base[4 * lane + 2] = excl + v0 + v1;
            s_digit_base[4 * lane + 3] = excl + v0 + v1 + v2;
        }
        __syncthreads();
        for (int e0 = seg_lo; e0 < seg_hi; e0 += 256) {
          unsigned kk[4], ii[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
              const bool a = e0 + u * 64 + lane < seg_hi;
              kk[u] = a ? kin[e0 + u * 64 + lane] : 0u;
              ii[u] = a ? iin[e0 + u * 64 + lane] : 0u;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (e0 + u * 64 >= seg_hi) break; // wave-uniform
            bool act = e0 + u * 64 + lane < seg_hi;
            unsigned key = kk[u];
            unsigned idx = ii[u];
            unsigned dg = 255u - ((key >> shift) & 255u);
            unsigned long long peers = wave_match8(dg, act);
            if (act) {
                // wave-private row: all lanes read the running offset of their digit, then the
                // first peer advances it (volatile: the next round must re-read it)
                volatile unsigned *wcnt = s_cnt[wave];
                unsigned rank = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
                unsigned off = wcnt[dg];
                unsigned pos = s_digit_base[dg] + off + rank;
                kout[pos] = key;
                iout[pos] = idx;
                if (rank == 0) wcnt[dg] = off + (unsigned)__popcll(peers);
            }
          }
        }
        __syncthreads();
        uint32_t *tk = kin; kin = kout; kout = tk;
        uint32_t *ti = iin; iin = iout; iout = ti;
    }
    // 4 passes: the sorted list is back in (keyA, idxA) == (kin, iin)
    for (int j = tid; j < k; j += TOPK_THREADS) {
        unsigned idx = iin[j];
        idx_out[(long long)b * k + j] = (int32_t)idx;
        if (val_out) val_out[(long long)b * k + j] = sc[idx];
    }
}

// ---- register-resident variant ----------------------------------------------------------------
// Same algorithm, but every phase works on registers: the image's keys are loaded ONCE (NR per
// thread) for the four select passes and the compaction, and each sort pass loads a wave's whole
// segment (KR rounds of 64) up front, ranks it from registers and scatters with fire-and-forget
// stores.  The streaming kernel above pays one exposed memory latency per round of 64/1024
// elements; this one pays one per pass.  Used when n <= NR*1024 and k <= KR*1024.
__device__ __forceinline__ int2 block_excl_count2(int a, int b, int *wave_sums, int2 *total)
{
    // exclusive prefix counts of two flags at once, packed as (b << 16) | a (each <= 1024 per tile)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    unsigned long long ma = __ballot(a != 0), mb = __ballot(b != 0);
    unsigned long long lt = (1ull << lane) - 1ull;
    int in_a = __popcll(ma & lt), in_b = __popcll(mb & lt);
    if (lane == 0) wave_sums[wave] = (__popcll(mb) << 16) | __popcll(ma);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < TOPK_WAVES; ++w) {
        int v = wave_sums[w];
        if (w < wave) base += v;
        tot += v;
    }
    __syncthreads();
    *total = make_int2(tot & 0xFFFF, tot >> 16);
    return make_int2((base & 0xFFFF) + in_a, (base >> 16) + in_b);
}

template <int NR, int KR>
__global__ void __launch_bounds__(TOPK_THREADS)
topk_reg_kernel(const float *__restrict__ scores, int n, int k, float *__restrict__ val_out,
                int32_t *__restrict__ idx_out, uint32_t *__restrict__ ws)
{
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_cnt[TOPK_WAVES][256];
    __shared__ unsigned s_digit_base[256];
    __shared__ int s_wave_sums[16];
    __shared__ unsigned s_prefix, s_remaining;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = blockIdx.x;
    const float *sc = scores + (long long)b * n;
    uint32_t *keyA = ws + (long long)b * 4 * k;
    uint32_t *idxA = keyA + k;
    uint32_t *keyB = idxA + k;
    uint32_t *idxB = keyB + k;

    // ---- 0. one pass over HBM: element i = r*1024 + tid
    unsigned key[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        int i = r * TOPK_THREADS + tid;
        key[r] = i < n ? frcnn_score_key(sc[i]) : 0u;
    }

    // ---- 1. radix select
    unsigned prefix = 0, mask = 0, remaining = (unsigned)k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r * TOPK_THREADS < n) { // block-uniform
                int i = r * TOPK_THREADS + tid;
                bool act = i < n && (key[r] & mask) == prefix;
                wave_hist_add(s_hist, (key[r] >> shift) & 255u, act);
            }
        }
        __syncthreads();
        if (wave == 0) {
            unsigned h0 = s_hist[4 * lane], h1 = s_hist[4 * lane + 1], h2 = s_hist[4 * lane + 2],
                     h3 = s_hist[4 * lane + 3];
            unsigned sum = h0 + h1 + h2 + h3;
            unsigned incl = sum;
            for (int o = 1; o < 64; o <<= 1) {
                unsigned t = __shfl_down(incl, o);
                if (lane + o < 64) incl += t;
            }
            unsigned above = incl - sum;
            unsigned s3 = above + h3, s2 = s3 + h2, s1 = s2 + h1, s0 = s1 + h0;
            int d = -1;
            unsigned cum = 0;
            if (s3 >= remaining && above < remaining) { d = 4 * lane + 3; cum = above; }
            else if (s2 >= remaining && s3 < remaining) { d = 4 * lane + 2; cum = s3; }
            else if (s1 >= remaining && s2 < remaining) { d = 4 * lane + 1; cum = s2; }
            else if (s0 >= remaining && s1 < remaining) { d = 4 * lane; cum = s1; }
            if (d >= 0) {
                s_prefix = prefix | ((unsigned)d << shift);
                s_remaining = remaining - cum;
            }
        }
        __syncthreads();
        prefix = s_prefix;
        remaining = s_remaining;
        mask |= 255u << shift;
    }
    const unsigned T = prefix;
    const int need_eq = (int)remaining;

    // ---- 2. index-ordered compaction: #selected before me = #gt before + min(#eq before, need_eq)
    {
        int gt_base = 0, eq_base = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r * TOPK_THREADS < n) {
                int i = r * TOPK_THREADS + tid;
                bool gt = i < n && key[r] > T;
                bool eq = i < n && key[r] == T;
                int2 tot;
                int2 ex = block_excl_count2(gt, eq, s_wave_sums, &tot);
                int eq_before = eq_base + ex.y;
                if (gt || (eq && eq_before < need_eq)) {
                    int pos = gt_base + ex.x + min(eq_before, need_eq);
                    keyA[pos] = key[r];
                    idxA[pos] = (uint32_t)i;
                }
                gt_base += tot.x;
                eq_base += tot.y;
            }
        }
    }
    __syncthreads();

    // ---- 3. stable LSD radix sort from registers
    const int seg = ((k + TOPK_WAVES - 1) / TOPK_WAVES + 63) / 64 * 64;
    const int seg_lo = min(wave * seg, k);
    const int seg_hi = min(seg_lo + seg, k);
    uint32_t *kin = keyA, *iin = idxA, *kout = keyB, *iout = idxB;
    for (int shift = 0; shift < 32; shift += 8) {
        unsigned kreg[KR], ireg[KR], rk[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            int e = seg_lo + r * 64 + lane;
            bool act = e < seg_hi;
            kreg[r] = act ? kin[e] : 0u;
            ireg[r] = act ? iin[e] : 0u;
        }
        for (int t = tid; t < TOPK_WAVES * 256; t += TOPK_THREADS) (&s_cnt[0][0])[t] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            if (seg_lo + r * 64 < seg_hi) { // wave-uniform
                bool act = seg_lo + r * 64 + lane < seg_hi;
                unsigned dg = 255u - ((kreg[r] >> shift) & 255u);
                unsigned long long peers = wave_match8(dg, act);
                unsigned rank = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
                unsigned cnt = (unsigned)__popcll(peers);
                rk[r] = rank | (cnt << 8);
                if (act && rank == 0) atomicAdd(&s_cnt[wave][dg], cnt);
            }
        }
        __syncthreads();
        if (tid < 256) {
            unsigned run = 0;
#pragma unroll
            for (int w = 0; w < TOPK_WAVES; ++w) {
                unsigned c = s_cnt[w][tid];
                s_cnt[w][tid] = run;
                run += c;
            }
            s_hist[tid] = run;
        }
        __syncthreads();
        if (wave == 0) {
            unsigned v0 = s_hist[4 * lane], v1 = s_hist[4 * lane + 1], v2 = s_hist[4 * lane + 2],
                     v3 = s_hist[4 * lane + 3];
            unsigned sum = v0 + v1 + v2 + v3;
            unsigned incl = sum;
            for (int o = 1; o < 64; o <<= 1) {
                unsigned t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            unsigned excl = incl - sum;
            s_digit_base[4 * lane] = excl;
            s_digit_base[4 * lane + 1] = excl + v0;
            s_digit_base[4 * lane + 2] = excl + v0 + v1;
            s_digit_base[4 * lane + 3] = excl + v0 + v1 + v2;
        }
        __syncthreads();
        volatile unsigned *wcnt = s_cnt[wave];
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            if (seg_lo + r * 64 < seg_hi) {
                bool act = seg_lo + r * 64 + lane < seg_hi;
                if (act) {
                    unsigned dg = 255u - ((kreg[r] >> shift) & 255u);
                    unsigned rank = rk[r] & 255u, cnt = rk[r] >> 8;
                    unsigned off = wcnt[dg];
                    unsigned pos = s_digit_base[dg] + off + rank;
                    kout[pos] = kreg[r];
                    iout[pos] = ireg[r];
                    if (rank == 0) wcnt[dg] = off + cnt;
                }
            }
        }
        __syncthreads();
        uint32_t *tk = kin; kin = kout; kout = tk;
        uint32_t *ti = iin; iin = iout; iout = ti;
    }
    for (int j = tid; j < k; j += TOPK_THREADS) {
        unsigned idx = iin[j];
        idx_out[(long long)b * k + j] = (int32_t)idx;
        if (val_out) val_out[(long long)b * k + j] = sc[idx];
    }
}

// ---- full-sort variant (k is a large part of n) ---------------------------------------------------
// When most of the image is kept (the RPN keeps 12,000 of 20,646 anchors) selecting first saves
// little: 4 select passes over n, a compaction and 4 sort passes over k cost more than SORTING all
// n keys with wider digits.  This kernel is a stable LSD radix sort of the whole image in 3 passes
// of 11-bit digits (descending key; the input is in index order, so ties come out by ascending
// index) whose last pass scatters only the first k positions, straight into the outputs.
// Per pass a wave owns a contiguous segment, held in registers (KR rounds of 64): rank inside the
// wave by 11-ballot matching, wave-private digit counters in LDS (16 x 2048 words = 128 KB),
// one cross-wave scan, fire-and-forget scatter to the ping-pong buffers (L2-resident).
#define TOPK_SORT_BITS 11
#define TOPK_SORT_BINS (1 << TOPK_SORT_BITS)

__device__ __forceinline__ unsigned long long wave_match11(unsigned digit, bool active)
{
    unsigned long long peers = __ballot(active);
#pragma unroll
    for (int bit = 0; bit < TOPK_SORT_BITS; ++bit) {
        bool one = (digit >> bit) & 1u;
        unsigned long long bm = __ballot(one);
        peers &= one ? bm : ~bm;
    }
    return peers;
}

// sc: the n scores to sort; the first k positions of the result go to idx_out[0..k) (index into sc) and,
// when val_out is given, val_out[0..k) (the score, or its raw sort key when RAWKEY); ws: 4 n words.
// Where the keys of pass 0 come from: scores in a flat array, or keys made beforehand.
struct FlatScores {
    const float *sc;
    __device__ __forceinline__ uint32_t key(int e) const { return frcnn_score_key(sc[e]); }
    __device__ __forceinline__ unsigned idx(int e) const { return (unsigned)e; }
};

struct ReadyKeys {
    const uint32_t *keys;
    __device__ __forceinline__ uint32_t key(int e) const { return keys[e]; }
    __device__ __forceinline__ unsigned idx(int e) const { return (unsigned)e; }
};

// keys that survived a filter, in index order, with the place each one came from (< 2^15)
struct CompactKeys {
    const uint32_t *keys, *orig;
    __device__ __forceinline__ uint32_t key(int e) const { return keys[e]; }
    __device__ __forceinline__ unsigned idx(int e) const { return orig[e]; }
};

template <int KR, bool RAWKEY, typename Src>
__device__ __forceinline__ void topk_sort_body(const Src src, int n, int k, uint32_t *__restrict__ val_out,
                                               int32_t *__restrict__ idx_out, uint32_t *__restrict__ ws)
{
    __shared__ unsigned s_cnt[TOPK_WAVES][TOPK_SORT_BINS];
    __shared__ unsigned s_base[TOPK_SORT_BINS];
    __shared__ unsigned s_wsum[2][TOPK_WAVES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    uint32_t *keyA = ws;
    uint32_t *idxA = keyA + n;
    uint32_t *keyB = idxA + n;
    uint32_t *idxB = keyB + n;
    const int seg = ((n + TOPK_WAVES - 1) / TOPK_WAVES + 63) / 64 * 64;
    const int seg_lo = min(wave * seg, n);
    const int seg_hi = min(seg_lo + seg, n);
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t *kin = keyA, *iin = idxA, *kout = keyB, *iout = idxB;
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass * TOPK_SORT_BITS;
        // ireg: index in bits 0..14 (n < 2^15), then the element's rank among its digit peers of the
        // wave (bits 16..21) and their number (bits 24..30) once known
        unsigned kreg[KR], ireg[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int e = seg_lo + r * 64 + lane;
            const bool act = e < seg_hi;
            if (pass == 0) {
                kreg[r] = act ? src.key(e) : 0u;
                ireg[r] = act ? src.idx(e) : 0u;
            } else {
                kreg[r] = act ? kin[e] : 0u;
                ireg[r] = act ? iin[e] : 0u;
            }
        }
        for (int t = tid; t < TOPK_WAVES * TOPK_SORT_BINS; t += TOPK_THREADS) (&s_cnt[0][0])[t] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            if (seg_lo + r * 64 < seg_hi) { // wave-uniform
                const bool act = seg_lo + r * 64 + lane < seg_hi;
                const unsigned dg = (TOPK_SORT_BINS - 1) - ((kreg[r] >> shift) & (TOPK_SORT_BINS - 1));
                const unsigned long long peers = wave_match11(dg, act);
                const unsigned rank = (unsigned)__popcll(peers & lt);
                const unsigned cnt = (unsigned)__popcll(peers);
                ireg[r] |= (rank << 16) | (cnt << 24);
                if (act && rank == 0) atomicAdd(&s_cnt[wave][dg], cnt);
            }
        }
        __syncthreads();
        // per digit: exclusive scan over the waves; thread t owns digits 2t and 2t+1 (adjacent, so the
        // scan over digits below is a scan over threads)
        unsigned tot0 = 0, tot1 = 0;
#pragma unroll
        for (int w = 0; w < TOPK_WAVES; ++w) {
            const unsigned c0 = s_cnt[w][2 * tid], c1 = s_cnt[w][2 * tid + 1];
            s_cnt[w][2 * tid] = tot0;
            s_cnt[w][2 * tid + 1] = tot1;
            tot0 += c0;
            tot1 += c1;
        }
        {
            const unsigned mine = tot0 + tot1;
            unsigned incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            if (lane == 63) s_wsum[pass & 1][wave] = incl;
            __syncthreads();
            unsigned wbase = 0;
#pragma unroll
            for (int w = 0; w < TOPK_WAVES; ++w) wbase += w < wave ? s_wsum[pass & 1][w] : 0u;
            const unsigned excl = wbase + incl - mine;
            s_base[2 * tid] = excl;
            s_base[2 * tid + 1] = excl + tot0;
        }
        __syncthreads();
        volatile unsigned *wcnt = s_cnt[wave];
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            if (seg_lo + r * 64 < seg_hi) {
                const bool act = seg_lo + r * 64 + lane < seg_hi;
                if (act) {
                    const unsigned dg = (TOPK_SORT_BINS - 1) - ((kreg[r] >> shift) & (TOPK_SORT_BINS - 1));
                    const unsigned rank = (ireg[r] >> 16) & 63u, cnt = ireg[r] >> 24;
                    const unsigned idx = ireg[r] & 0x7FFFu;
                    const unsigned off = wcnt[dg];
                    const unsigned pos = s_base[dg] + off + rank;
                    if (pass == 2) {
                        if (pos < (unsigned)k) {
                            idx_out[pos] = (int32_t)idx;
                            if (val_out) val_out[pos] = RAWKEY ? kreg[r] : __float_as_uint(frcnn_key_score(kreg[r]));
                        }
                    } else {
                        kout[pos] = kreg[r];
                        iout[pos] = idx;
                    }
                    if (rank == 0) wcnt[dg] = off + cnt;
                }
            }
        }
        __syncthreads();
        uint32_t *tk = kin; kin = kout; kout = tk;
        uint32_t *ti = iin; iin = iout; iout = ti;
    }
}

template <int KR>
__global__ void __launch_bounds__(TOPK_THREADS)
topk_sort_kernel(const float *__restrict__ scores, int n, int k, float *__restrict__ val_out,
                 int32_t *__restrict__ idx_out, uint32_t *__restrict__ ws)
{
    const long long b = blockIdx.x;
    FlatScores src;
    src.sc = scores + b * n;
    topk_sort_body<KR, false>(src, n, k, val_out ? (uint32_t *)val_out + b * k : nullptr, idx_out + b * k, ws + b * 4 * n);
}

// ---- split variant: several workgroups per image ---------------------------------------------------
// One workgroup per image leaves 240 of the 256 CUs idle at batch 16 and all but one at batch 1, and an
// image of 200,000 scores (FPN) does not fit one workgroup's registers at all.  Here
// an image is cut into S contiguous slices; slice s is SORTED by its own workgroup (the kernel above, on
// slice_len scores) which leaves its best min(k, slice length) (key, local index) pairs; then every
// listed element finds its place in the image's order -- its place in its own list plus, for every other
// list, the number of elements that go before it -- and, when that is below k, writes itself there.
// "Before" is the total order of the whole kernel family: larger key first, equal keys by ascending
// index; slices are index ranges, so against an earlier slice equal keys go first and against a later
// one they do not.  No atomics, no data-dependent fallback: any distribution costs the same.
// The merge works a block of 256 consecutive list places at a time: two binary searches per other list
// (for the block's first and last key) bound a window of that list outside of which nothing changes
// sides within the block; the windows (about 256 elements each) are staged in LDS and each thread
// searches them there.  A block whose first element already ranks >= k has nothing to write.
#define TOPK_MERGE_THREADS 256
#define TOPK_MERGE_WIN 5120  // words of LDS for the windows (15 x 256 typical); larger windows are searched in global memory
#define TOPK_MAX_SLICES 64

template <int KR>
__global__ void __launch_bounds__(TOPK_THREADS)
topk_slice_sort_kernel(const float *__restrict__ scores, int n, int k, int S, int slice_len, int kcap,
                       uint32_t *__restrict__ ws, uint32_t *__restrict__ skeys, int32_t *__restrict__ sidx)
{
    const long long b = blockIdx.x / S;
    const int s = blockIdx.x % S;
    const int start = s * slice_len;
    const int n_s = min(slice_len, n - start);
    const long long list = (b * S + s) * kcap;
    FlatScores src;
    src.sc = scores + b * n + start;
    topk_sort_body<KR, true>(src, n_s, min(k, n_s), skeys + list, sidx + list, ws + b * 4 * n + 4ll * start);
}

// The same on the RPN head's outputs where they lie (common.h: RpnLevels): the slice's keys are made
// first, in flat order, into the one buffer of its workspace that the sort's first pass does not write --
// the level lookup stays out of the sort's unrolled, register-resident loops.
template <int KR>
__global__ void __launch_bounds__(TOPK_THREADS)
topk_slice_sort_nchw_kernel(RpnLevels L, int n, int k, int S, int slice_len, int kcap, uint32_t *__restrict__ ws,
                            uint32_t *__restrict__ skeys, int32_t *__restrict__ sidx)
{
    const long long b = blockIdx.x / S;
    const int s = blockIdx.x % S;
    const int start = s * slice_len;
    const int n_s = min(slice_len, n - start);
    const long long list = (b * S + s) * kcap;
    uint32_t *wss = ws + b * 4 * n + 4ll * start;
#pragma unroll 4
    for (int e = threadIdx.x; e < n_s; e += TOPK_THREADS) wss[e] = frcnn_score_key(rpn_score(L, (int)b, start + e));
    __syncthreads();
    ReadyKeys src;
    src.keys = wss;
    topk_sort_body<KR, true>(src, n_s, min(k, n_s), skeys + list, sidx + list, wss);
}

// ---- selective variant: k is a small part of n (FPN: 30,000 of 200,907 or of 805,707) --------------------
// Sorting every slice in full is then mostly wasted: a slice's sort costs ~10 us per 1,000 scores, and the merge
// looks at every listed element.  Three cheap steps first decide which scores can matter at all:
//   topk_slice_hist_kernel: every slice makes its keys (kept for the sort) and a 4096-bin histogram of their
//                           top 12 bits, written to its own row of the workspace (no atomics, nothing to clear);
//   topk_pick_kernel:       per image, the bin in which the k-th largest key lies -> threshold T = that bin's
//                           lowest key: at least k keys are >= T, and every key of the top k is;
//   topk_slice_sort_sel_kernel: a slice compacts its keys >= T in place (index order, so ties keep their rule),
//                           sorts only those, and records how many it listed for the merge.
// Nothing is approximate: the merge ranks exactly as before over shorter lists, and a degenerate input (all
// scores in one bin) just falls back to sorting everything, at the old cost.
#define TOPK_SEL_BITS 12
#define TOPK_SEL_BINS (1 << TOPK_SEL_BITS)

template <bool NCHW>
__global__ void __launch_bounds__(TOPK_THREADS)
topk_slice_hist_kernel(const float *__restrict__ scores, RpnLevels L, int n, int S, int slice_len,
                       uint32_t *__restrict__ ws, unsigned *__restrict__ hist)
{
    __shared__ unsigned s_hist[TOPK_SEL_BINS];
    const long long b = blockIdx.x / S;
    const int s = blockIdx.x % S;
    const int start = s * slice_len;
    const int n_s = min(slice_len, n - start);
    uint32_t *wss = ws + b * 4 * n + 4ll * start;
    for (int t = threadIdx.x; t < TOPK_SEL_BINS; t += TOPK_THREADS) s_hist[t] = 0;
    __syncthreads();
#pragma unroll 4
    for (int e = threadIdx.x; e < n_s; e += TOPK_THREADS) {
        const uint32_t key = frcnn_score_key(NCHW ? rpn_score(L, (int)b, start + e) : scores[b * n + start + e]);
        wss[e] = key;
        atomicAdd(&s_hist[key >> (32 - TOPK_SEL_BITS)], 1u);
    }
    __syncthreads();
    unsigned *h = hist + (long long)blockIdx.x * TOPK_SEL_BINS;
    for (int t = threadIdx.x; t < TOPK_SEL_BINS; t += TOPK_THREADS) h[t] = s_hist[t];
}

__global__ void __launch_bounds__(TOPK_THREADS)
topk_pick_kernel(const unsigned *__restrict__ hist, int S, int k, uint32_t *__restrict__ thr_key)
{
    __shared__ unsigned s_wsum[TOPK_WAVES];
    const long long b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // thread t owns the four bins that come (4t)-th .. (4t+3)-th from the TOP
    unsigned c[4] = {0, 0, 0, 0};
    for (int s = 0; s < S; ++s) {
        const unsigned *h = hist + (b * S + s) * TOPK_SEL_BINS + (TOPK_SEL_BINS - 4 - 4 * tid);
        const uint4 v = *(const uint4 *)h; // bins ascending: v.w is the highest of the four
        c[0] += v.w; c[1] += v.z; c[2] += v.y; c[3] += v.x;
    }
    const unsigned mine = c[0] + c[1] + c[2] + c[3];
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    unsigned before = incl - mine;
    for (int w = 0; w < wave; ++w) before += s_wsum[w];
    // the thread whose bins take the count of keys from below k to at least k holds the k-th largest key
    if (before < (unsigned)k && before + mine >= (unsigned)k) {
        unsigned cum = before;
        int j = 0;
        for (; j < 3; ++j) {
            cum += c[j];
            if (cum >= (unsigned)k) break;
        }
        const unsigned bin = (unsigned)(TOPK_SEL_BINS - 1 - (4 * tid + j));
        thr_key[b] = bin << (32 - TOPK_SEL_BITS);
    }
}

template <int KR>
__global__ void __launch_bounds__(TOPK_THREADS)
topk_slice_sort_sel_kernel(int n, int k, int S, int slice_len, int kcap, uint32_t *__restrict__ ws,
                           const uint32_t *__restrict__ thr_key, uint32_t *__restrict__ skeys,
                           int32_t *__restrict__ sidx, int32_t *__restrict__ slen)
{
    __shared__ int s_wave_sums[TOPK_WAVES];
    const long long b = blockIdx.x / S;
    const int s = blockIdx.x % S;
    const int start = s * slice_len;
    const int n_s = min(slice_len, n - start);
    const long long list = (b * S + s) * kcap;
    uint32_t *wss = ws + b * 4 * n + 4ll * start; // [0, n_s): the slice's keys; [n_s, 2 n_s): where a survivor came from
    const uint32_t T = thr_key[b];
    // In-place compaction, 4096 places at a time (four consecutive ones per thread): a round's keys are all in
    // registers before anything is written, and a survivor never moves up.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int kept = 0;
    for (int c0 = 0; c0 < n_s; c0 += 4 * TOPK_THREADS) {
        const int e = c0 + 4 * threadIdx.x;
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (e + 3 < n_s) {
            q = *(const uint4 *)(wss + e); // (a slice's buffer starts on a multiple of four words)
        } else {
            if (e < n_s) q.x = wss[e];
            if (e + 1 < n_s) q.y = wss[e + 1];
            if (e + 2 < n_s) q.z = wss[e + 2];
        }
        const bool k0 = e < n_s && q.x >= T, k1 = e + 1 < n_s && q.y >= T, k2 = e + 2 < n_s && q.z >= T,
                   k3 = e + 3 < n_s && q.w >= T;
        const int mine = (int)k0 + (int)k1 + (int)k2 + (int)k3;
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_wave_sums[wave] = incl;
        __syncthreads(); // (also: every key of the round is in registers)
        int pos = kept + incl - mine, total = 0;
#pragma unroll
        for (int w = 0; w < TOPK_WAVES; ++w) {
            const int v = s_wave_sums[w];
            pos += w < wave ? v : 0;
            total += v;
        }
        __syncthreads(); // the sums have been read
        if (k0) { wss[pos] = q.x; wss[n_s + pos] = (uint32_t)e; ++pos; }
        if (k1) { wss[pos] = q.y; wss[n_s + pos] = (uint32_t)(e + 1); ++pos; }
        if (k2) { wss[pos] = q.z; wss[n_s + pos] = (uint32_t)(e + 2); ++pos; }
        if (k3) { wss[pos] = q.w; wss[n_s + pos] = (uint32_t)(e + 3); ++pos; }
        kept += total;
    }
    __syncthreads();
    const int k_s = min(k, kept);
    CompactKeys src;
    src.keys = wss;
    src.orig = wss + n_s;
    // (the sort's first pass has every key and index in registers before it writes; its buffers may overlap them)
    topk_sort_body<KR, true>(src, kept, k_s, skeys + list, sidx + list, wss);
    if (threadIdx.x == 0) slen[b * S + s] = k_s;
}

// number of leading elements of a descending list that go before key K: those >= K (or_equal) or > K
__device__ __forceinline__ int topk_count_before(const uint32_t *list, int len, uint32_t K, bool or_equal)
{
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const uint32_t v = list[mid];
        const bool before = or_equal ? v >= K : v > K;
        lo = before ? mid + 1 : lo;
        hi = before ? hi : mid;
    }
    return lo;
}

// Sum over the other lists o of posF[o] + (number of elements of window o that go before K).  Window o is
// base[start(o) .. start(o) + posL[o] - posF[o]) with start(o) = woff[o] (windows staged back to back) or
// o * stride + posF[o] (the lists themselves).  "v goes before K" is v >= K against an earlier slice and
// v > K against a later one, i.e. v > K - 1 and v > K (a key is never 0: -inf maps to 0x007FFFFF).
__device__ __forceinline__ int topk_merge_count(const uint32_t *__restrict__ base, const int *woff, const int *lstart,
                                                int stride, const int *posF, const int *posL, int S, int s, uint32_t K,
                                                int top)
{
    int sum = 0;
    for (int o0 = 0; o0 < S; o0 += 4) {
        int cnt[4], len[4];
        const uint32_t *win[4];
        uint32_t thr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int o = o0 + u;
            const bool valid = o < S && o != s;
            const int oc = o < S ? o : 0;
            len[u] = valid ? posL[oc] - posF[oc] : 0;
            win[u] = base + (woff ? (long long)woff[oc] : (long long)oc * stride + lstart[oc]) - 1; // win[h] = element h - 1
            thr[u] = o < s ? K - 1u : K;
            cnt[u] = 0;
            sum += valid ? posF[oc] : 0;
        }
        for (int h = top; h >= 1; h >>= 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pos = cnt[u] + h;
                if (pos <= len[u]) cnt[u] = win[u][pos] > thr[u] ? pos : cnt[u];
            }
        }
        sum += cnt[0] + cnt[1] + cnt[2] + cnt[3];
    }
    return sum;
}

__global__ void __launch_bounds__(TOPK_MERGE_THREADS)
topk_merge_kernel(const uint32_t *__restrict__ skeys, const int32_t *__restrict__ sidx, int n, int k, int S,
                  int slice_len, int kcap, const int32_t *__restrict__ slen, float *__restrict__ val_out,
                  int32_t *__restrict__ idx_out)
{
    __shared__ uint32_t s_win[TOPK_MERGE_WIN];
    __shared__ int s_posF[TOPK_MAX_SLICES], s_posL[TOPK_MAX_SLICES], s_woff[TOPK_MAX_SLICES + 1];
    const long long b = blockIdx.z;
    const int s = blockIdx.y;
    const int i0 = blockIdx.x * TOPK_MERGE_THREADS;
    // list lengths: what the slice could hold, or (selective variant) what it listed
    const int len_s = slen ? slen[b * S + s] : min(k, min(slice_len, n - s * slice_len));
    if (i0 >= len_s) return;
    const int iend = min(i0 + TOPK_MERGE_THREADS, len_s);
    const uint32_t *lists = skeys + b * S * kcap;
    const uint32_t *mine = lists + (long long)s * kcap;
    const int t = threadIdx.x;
    // windows: places of the block's first and last key in every other list.  A search is a chain of
    // dependent loads, so 16 lanes share one and test 16 places per step (3-4 steps instead of 11-15).
    // this thread's own element first: its loads are in flight while the windows are found
    const int i = i0 + t;
    const uint32_t K = i < iend ? mine[i] : 0u;
    const int my_idx = i < iend ? sidx[(b * S + s) * kcap + i] : 0;
    __shared__ uint32_t s_kfl[2];
    if (t == 0) s_kfl[0] = K;
    if (i == iend - 1) s_kfl[1] = K;
    __syncthreads();
    {
        const uint32_t Kf = s_kfl[0], Kl = s_kfl[1];
        const int grp = t >> 4, j = t & 15;
        for (int q0 = 0; q0 < 2 * S; q0 += TOPK_MERGE_THREADS / 16) {
            const int q = q0 + grp;
            const bool live = q < 2 * S;
            const int o = live ? (q < S ? q : q - S) : s;
            const uint32_t K = q < S ? Kf : Kl;
            const uint32_t *list = lists + (long long)o * kcap;
            const bool or_equal = o < s;
            int lo = 0, hi = o == s ? 0 : (slen ? slen[b * S + o] : min(k, min(slice_len, n - o * slice_len)));
            while (__any(hi > lo)) {
                const int w = hi - lo;
                const int qj = lo + (int)(((long long)(j + 1) * w) >> 4); // a boundary in [lo, hi]
                bool before = true;                                       // boundary lo is known
                if (qj > lo) {
                    const uint32_t v = list[qj - 1];
                    before = or_equal ? v >= K : v > K;
                }
                const unsigned m16 = (unsigned)(__ballot(before) >> (t & 48)) & 0xFFFFu; // true for a prefix of lanes
                const int m = __popc(m16);
                const int lo_n = __shfl(qj, m > 0 ? m - 1 : 0, 16), hi_n = __shfl(qj - 1, m < 16 ? m : 15, 16);
                lo = m > 0 ? lo_n : lo;
                hi = m < 16 ? hi_n : hi;
            }
            if (live && j == 0) {
                if (q < S) s_posF[o] = lo; else s_posL[o] = lo;
            }
        }
    }
    __syncthreads();
    int rank0 = i0, total = 0;
    for (int o = 0; o < S; ++o) {
        rank0 += s_posF[o];
        total += s_posL[o] - s_posF[o];
    }
    if (rank0 >= k) return; // the block's best element is already outside the top k (uniform)
    const bool fits = total <= TOPK_MERGE_WIN;
    if (t == 0) {
        int off = 0;
        for (int o = 0; o < S; ++o) {
            s_woff[o] = off;
            off += s_posL[o] - s_posF[o];
        }
        s_woff[S] = off;
    }
    __syncthreads();
    if (fits) {
        // one flat loop over all windows: the loads of different lists are independent and in flight together
        // (a loop per list cost one memory round trip per list)
        const int total_w = s_woff[S];
#pragma unroll 4
        for (int f = t; f < total_w; f += TOPK_MERGE_THREADS) {
            int o = 0;
            while (f >= s_woff[o + 1]) ++o;
            s_win[f] = lists[(long long)o * kcap + s_posF[o] + (f - s_woff[o])];
        }
        __syncthreads();
    }
    if (i >= iend) return;
    int rank = i, top = 1;
    for (int o = 0; o < S; ++o)
        while (2 * top <= s_posL[o] - s_posF[o]) top *= 2;
    // four lists at a time, their (dependent) reads interleaved; count = sum of accepted powers of two.
    // Two instantiations so that the LDS case compiles to ds_read, not to flat loads through a pointer
    // that may be either.
    if (fits)
        rank += topk_merge_count(s_win, s_woff, nullptr, 0, s_posF, s_posL, S, s, K, top);
    else
        rank += topk_merge_count(lists, nullptr, s_posF, kcap, s_posF, s_posL, S, s, K, top);
    if (rank < k) {
        idx_out[b * k + rank] = my_idx + s * slice_len;
        if (val_out) val_out[b * k + rank] = frcnn_key_score(K);
    }
}

#define TOPK_SORT_KR 21
static bool topk_use_sort(int n, int k) { return n <= TOPK_SORT_KR * TOPK_THREADS && 2LL * k >= n && n > 8 * TOPK_THREADS; }

// Split plan: slices of at most 21,504 scores, enough of them to give the chip about one workgroup per
// CU (at most 8 by choice -- the merge does one search per other slice -- more only when the image needs them).  Images of up to 8,192 scores stay
// with one workgroup (register-resident kernels); more than 64 slices (1.37 M scores) with the
// streaming kernel.
struct TopkSplit {
    int S, slice_len, kcap;
};

static bool topk_split_plan(int B, int n, int k, int max_slices, TopkSplit *p)
{
    if (n <= 8 * TOPK_THREADS || B <= 0) return false;
    const int cap = TOPK_SORT_KR * TOPK_THREADS;
    int S = (n + cap - 1) / cap; // what the image needs
    int want = (256 + B - 1) / B;
    if (want > 8) want = 8; // measured: 0.116 / 0.078 / 0.054 / 0.048 / 0.059 ms at 1 / 2 / 4 / 8 / 16 slices (16 x 20,646)
    if (max_slices > 0 && want > max_slices) want = max_slices; // the caller's cap on workgroups per image
    if (S < want) S = want;
    if (S < 2) return false;
    if (S > TOPK_MAX_SLICES) return false;
    int slice_len = (n + S - 1) / S;
    slice_len = (slice_len + 63) / 64 * 64; // whole wave rounds; the last slice takes what is left
    S = (n + slice_len - 1) / slice_len;
    if (S < 2) return false;
    p->S = S;
    p->slice_len = slice_len;
    p->kcap = k < slice_len ? k : slice_len;
    return true;
}

extern "C" size_t frcnn_topk_workspace_bytes(int B, int n, int k)
{
    if (B <= 0 || k <= 0) return 0;
    // enough for every plan frcnn_topk_desc_ex_f32 can choose: one workgroup per image needs 4 n or 4 k words,
    // the split path 4 n for the slice sorts plus the slices' lists (keys and indices, at most n rounded up
    // to whole wave rounds per slice)
    // (+ the selective variant's per-slice histograms, list lengths and threshold, where it applies: k <= n / 4)
    if (n > 8 * TOPK_THREADS) {
        size_t words = 6 * (size_t)n + 2 * 64 * TOPK_MAX_SLICES;
        if (4LL * k <= n) words += (size_t)TOPK_MAX_SLICES * (TOPK_SEL_BINS + 1) + 64;
        return (size_t)B * words * sizeof(uint32_t) + 16;
    }
    const size_t per = topk_use_sort(n, k) ? (size_t)n : (size_t)k;
    return (size_t)B * 4 * per * sizeof(uint32_t);
}

template <int KR>
static void topk_launch_slices(const float *scores, const RpnLevels *levels, int B, int n, int k, const TopkSplit &sp,
                               uint32_t *ws, uint32_t *skeys, int32_t *sidx, hipStream_t stream)
{
    if (levels)
        hipLaunchKernelGGL((topk_slice_sort_nchw_kernel<KR>), dim3(B * sp.S), dim3(TOPK_THREADS), 0, stream, *levels, n,
                           k, sp.S, sp.slice_len, sp.kcap, ws, skeys, sidx);
    else
        hipLaunchKernelGGL((topk_slice_sort_kernel<KR>), dim3(B * sp.S), dim3(TOPK_THREADS), 0, stream, scores, n, k,
                           sp.S, sp.slice_len, sp.kcap, ws, skeys, sidx);
}

template <int KR>
static void topk_launch_slices_sel(int B, int n, int k, const TopkSplit &sp, uint32_t *ws, const uint32_t *thr,
                                   uint32_t *skeys, int32_t *sidx, int32_t *slen, hipStream_t stream)
{
    hipLaunchKernelGGL((topk_slice_sort_sel_kernel<KR>), dim3(B * sp.S), dim3(TOPK_THREADS), 0, stream, n, k, sp.S,
                       sp.slice_len, sp.kcap, ws, thr, skeys, sidx, slen);
}

// k at most a quarter of n: filter before sorting (see topk_slice_hist_kernel)
static bool topk_selective(int n, int k) { return 4LL * k <= n; }

// slice sorts + merge; scores == nullptr: read the level table instead
static int topk_run_split(const float *scores, const RpnLevels *levels, int B, int n, int k, const TopkSplit &sp,
                          float *val_out, int32_t *idx_out, void *workspace, hipStream_t stream)
{
    uint32_t *ws = (uint32_t *)workspace;
    uint32_t *skeys = ws + (size_t)B * 4 * n;
    int32_t *sidx = (int32_t *)(skeys + (size_t)B * sp.S * sp.kcap);
    // behind the lists' largest extent (B * (2 n + 2 * 64 * TOPK_MAX_SLICES) words): histograms, lengths, thresholds
    unsigned *hist = (unsigned *)(ws + ((size_t)B * (6 * (size_t)n + 2 * 64 * TOPK_MAX_SLICES) + 3) / 4 * 4); // 16-byte aligned rows
    int32_t *slen = (int32_t *)(hist + (size_t)B * TOPK_MAX_SLICES * TOPK_SEL_BINS);
    uint32_t *thr = (uint32_t *)(slen + (size_t)B * TOPK_MAX_SLICES);
    const int rounds = (sp.slice_len + TOPK_THREADS - 1) / TOPK_THREADS;
    const bool sel = topk_selective(n, k);
    if (sel) {
        if (levels)
            hipLaunchKernelGGL((topk_slice_hist_kernel<true>), dim3(B * sp.S), dim3(TOPK_THREADS), 0, stream,
                               (const float *)nullptr, *levels, n, sp.S, sp.slice_len, ws, hist);
        else {
            RpnLevels none = {};
            hipLaunchKernelGGL((topk_slice_hist_kernel<false>), dim3(B * sp.S), dim3(TOPK_THREADS), 0, stream, scores,
                               none, n, sp.S, sp.slice_len, ws, hist);
        }
        FRCNN_CHECK_LAUNCH("topk_slice_hist");
        hipLaunchKernelGGL(topk_pick_kernel, dim3(B), dim3(TOPK_THREADS), 0, stream, hist, sp.S, k, thr);
        FRCNN_CHECK_LAUNCH("topk_pick");
        if (rounds <= 2)
            topk_launch_slices_sel<2>(B, n, k, sp, ws, thr, skeys, sidx, slen, stream);
        else if (rounds <= 4)
            topk_launch_slices_sel<4>(B, n, k, sp, ws, thr, skeys, sidx, slen, stream);
        else if (rounds <= 8)
            topk_launch_slices_sel<8>(B, n, k, sp, ws, thr, skeys, sidx, slen, stream);
        else if (rounds <= 13)
            topk_launch_slices_sel<13>(B, n, k, sp, ws, thr, skeys, sidx, slen, stream);
        else
            topk_launch_slices_sel<TOPK_SORT_KR>(B, n, k, sp, ws, thr, skeys, sidx, slen, stream);
    } else if (rounds <= 2)
        topk_launch_slices<2>(scores, levels, B, n, k, sp, ws, skeys, sidx, stream);
    else if (rounds <= 4)
        topk_launch_slices<4>(scores, levels, B, n, k, sp, ws, skeys, sidx, stream);
    else if (rounds <= 8)
        topk_launch_slices<8>(scores, levels, B, n, k, sp, ws, skeys, sidx, stream);
    else if (rounds <= 13)
        topk_launch_slices<13>(scores, levels, B, n, k, sp, ws, skeys, sidx, stream);
    else
        topk_launch_slices<TOPK_SORT_KR>(scores, levels, B, n, k, sp, ws, skeys, sidx, stream);
    FRCNN_CHECK_LAUNCH("topk_slice_sort");
    hipLaunchKernelGGL(topk_merge_kernel, dim3((sp.kcap + TOPK_MERGE_THREADS - 1) / TOPK_MERGE_THREADS, sp.S, B),
                       dim3(TOPK_MERGE_THREADS), 0, stream, skeys, sidx, n, k, sp.S, sp.slice_len, sp.kcap,
                       sel ? slen : (const int32_t *)nullptr, val_out, idx_out);
    FRCNN_CHECK_LAUNCH("topk_merge");
    return FRCNN_OK;
}

extern "C" int frcnn_topk_desc_ex_f32(const float *scores, int B, int n, int k, float *val_out, int32_t *idx_out,
                                      void *workspace, size_t workspace_bytes, int max_slices, void *stream);

extern "C" int frcnn_topk_desc_f32(const float *scores, int B, int n, int k, float *val_out,
                                   int32_t *idx_out, void *workspace, size_t workspace_bytes,
                                   void *stream)
{
    return frcnn_topk_desc_ex_f32(scores, B, n, k, val_out, idx_out, workspace, workspace_bytes, 0, stream);
}

extern "C" int frcnn_topk_desc_ex_f32(const float *scores, int B, int n, int k, float *val_out, int32_t *idx_out,
                                      void *workspace, size_t workspace_bytes, int max_slices, void *stream)
{
    FRCNN_REQUIRE(B >= 0 && n >= 0 && k >= 0 && k <= n, "topk: bad shape (B=%d n=%d k=%d)", B, n, k);
    if (B == 0 || k == 0) return FRCNN_OK;
    FRCNN_REQUIRE(scores && idx_out && workspace, "topk: null pointer");
    if (workspace_bytes < frcnn_topk_workspace_bytes(B, n, k)) {
        frcnn_set_error("topk: workspace too small (%zu < %zu)", workspace_bytes,
                        frcnn_topk_workspace_bytes(B, n, k));
        return FRCNN_EWORKSPACE;
    }
    TopkSplit sp;
    if (topk_split_plan(B, n, k, max_slices, &sp))
        return topk_run_split(scores, nullptr, B, n, k, sp, val_out, idx_out, workspace, (hipStream_t)stream);
    if (topk_use_sort(n, k))
        hipLaunchKernelGGL((topk_sort_kernel<TOPK_SORT_KR>), dim3(B), dim3(TOPK_THREADS), 0, (hipStream_t)stream,
                           scores, n, k, val_out, idx_out, (uint32_t *)workspace);
    else if (n <= 8 * TOPK_THREADS && k <= 8 * TOPK_THREADS)
        hipLaunchKernelGGL((topk_reg_kernel<8, 8>), dim3(B), dim3(TOPK_THREADS), 0, (hipStream_t)stream,
                           scores, n, k, val_out, idx_out, (uint32_t *)workspace);
    else if (n <= 24 * TOPK_THREADS && k <= 12 * TOPK_THREADS)
        hipLaunchKernelGGL((topk_reg_kernel<24, 12>), dim3(B), dim3(TOPK_THREADS), 0, (hipStream_t)stream,
                           scores, n, k, val_out, idx_out, (uint32_t *)workspace);
    else if (n <= 32 * TOPK_THREADS && k <= 16 * TOPK_THREADS)
        hipLaunchKernelGGL((topk_reg_kernel<32, 16>), dim3(B), dim3(TOPK_THREADS), 0, (hipStream_t)stream,
                           scores, n, k, val_out, idx_out, (uint32_t *)workspace);
    else
        hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(TOPK_THREADS), 0, (hipStream_t)stream, scores, n,
                           k, val_out, idx_out, (uint32_t *)workspace);
    FRCNN_CHECK_LAUNCH("topk");
    return FRCNN_OK;
}

extern "C" int frcnn_topk_desc_nchw_supported(int B, int A, int k, int max_slices)
{
    TopkSplit sp;
    return (B > 0 && k > 0 && k <= A && topk_split_plan(B, A, k, max_slices, &sp)) ? 1 : 0;
}

extern "C" int frcnn_topk_desc_nchw_f32(const float *const *cls_host, const int32_t *hw_host, int n_levels, int na,
                                        int B, int k, float *val_out, int32_t *idx_out, void *workspace,
                                        size_t workspace_bytes, int max_slices, void *stream)
{
    FRCNN_REQUIRE(cls_host && B > 0 && k > 0, "topk_nchw: bad arguments");
    RpnLevels L;
    const long long A = frcnn_make_levels(&L, cls_host, nullptr, hw_host, n_levels, na);
    FRCNN_REQUIRE(A > 0, "topk_nchw: bad level table (1..%d levels, positive sizes)", FRCNN_MAX_LEVELS);
    for (int l = 0; l < n_levels; ++l) FRCNN_REQUIRE(cls_host[l], "topk_nchw: null level %d", l);
    FRCNN_REQUIRE(k <= A, "topk_nchw: k (%d) > A (%lld)", k, A);
    TopkSplit sp;
    FRCNN_REQUIRE(topk_split_plan(B, (int)A, k, max_slices, &sp),
                  "topk_nchw: A=%lld is outside the split path's range (flatten and use frcnn_topk_desc_f32)", A);
    FRCNN_REQUIRE(idx_out && workspace, "topk_nchw: null pointer");
    if (workspace_bytes < frcnn_topk_workspace_bytes(B, (int)A, k)) {
        frcnn_set_error("topk_nchw: workspace too small (%zu < %zu)", workspace_bytes,
                        frcnn_topk_workspace_bytes(B, (int)A, k));
        return FRCNN_EWORKSPACE;
    }
    return topk_run_split(nullptr, &L, B, (int)A, k, sp, val_out, idx_out, workspace, (hipStream_t)stream);
}
