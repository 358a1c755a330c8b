// store_probe: how fast can 3.2 GB of RoI features be WRITTEN with the block decomposition of the
// RoIAlign plane kernel, compared with a plain streaming fill?  (experiment, not product code)
//   hipcc --offload-arch=gfx950 -O3 -o store_probe store_probe.hip && ./store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                         \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

// mode 0: plain grid-stride float4 fill.
__global__ void __launch_bounds__(1024) fill_kernel(float4 *out, long long n4)
{
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (long long)gridDim.x * 1024)
        out[i] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
}

// mode 1: grid (chunks, splits, B); block writes, for each of its RoIs, the 3136-byte piece of its
// channel chunk (196 float4), RoI row stride C*49 floats -- the plane kernel's pattern.
// piece4 = float4 per piece, cpr = pieces per RoI row.
__global__ void __launch_bounds__(1024) piece_kernel(float4 *out, int P, int piece4, int cpr)
{
    const int chunk = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const int per = (P + gridDim.y - 1) / gridDim.y;
    const int j0 = split * per, j1 = min(P, j0 + per);
    const int n = (j1 - j0) * piece4;
    for (int idx = threadIdx.x; idx < n; idx += blockDim.x) {
        int jl = idx / piece4, f = idx - jl * piece4;
        long long row = (long long)b * P + j0 + jl;
        out[(row * cpr + chunk) * piece4 + f] = make_float4(1.0f, (float)f, 3.0f, 4.0f);
    }
}

__global__ void __launch_bounds__(1024) piece_nt_kernel(float4 *out, int P, int piece4, int cpr)
{
    const int chunk = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const int per = (P + gridDim.y - 1) / gridDim.y;
    const int j0 = split * per, j1 = min(P, j0 + per);
    const int n = (j1 - j0) * piece4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int idx = threadIdx.x; idx < n; idx += blockDim.x) {
        int jl = idx / piece4, f = idx - jl * piece4;
        long long row = (long long)b * P + j0 + jl;
        f4 v = {1.0f, (float)f, 3.0f, 4.0f};
        __builtin_nontemporal_store(v, (f4 *)&out[(row * cpr + chunk) * piece4 + f]);
    }
}

// mode 12/13: 1-D grid; the 32 chunk blocks of one (image, split) share an XCD (block id % 8), nt optional.
template <bool NT>
__global__ void __launch_bounds__(1024) piece_xcd_kernel(float4 *out, int P, int piece4, int cpr, int splits)
{
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;    // slot-th block on this XCD
    const int chunk = slot % cpr;
    const int grp = (slot / cpr) * 8 + xcd;      // (image, split) group
    const int split = grp % splits, b = grp / splits;
    const int per = (P + splits - 1) / splits;
    const int j0 = split * per, j1 = min(P, j0 + per);
    const int n = (j1 - j0) * piece4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int idx = threadIdx.x; idx < n; idx += blockDim.x) {
        int jl = idx / piece4, f = idx - jl * piece4;
        long long row = (long long)b * P + j0 + jl;
        f4 v = {1.0f, (float)f, 3.0f, 4.0f};
        f4 *p = (f4 *)&out[(row * cpr + chunk) * piece4 + f];
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
}

// mode 16: the plane kernel's CURRENT stores: lane = (RoI, bin) item, 16 dword stores (one per channel).
// mode 17: same items; per 4-channel group ONE aligned 16-byte store per lane (the lane that starts an
// aligned quad of 4 consecutive bins of one channel inside this wave) plus dword stores for the ragged ends.
template <int MODE>
__global__ void __launch_bounds__(1024) item_kernel(float *out, int P, int C)
{
    const int chunk = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const int per = (P + gridDim.y - 1) / gridDim.y;
    const int j0 = split * per, j1 = min(P, j0 + per);
    const int items = (j1 - j0) * 49;
    const int lane = threadIdx.x & 63;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int t0 = threadIdx.x - lane; t0 < items; t0 += 1024) {
        const int t = t0 + lane;
        if (t >= items) continue;
        const int jl = t / 49, bin = t - jl * 49;
        float *o = out + (((long long)b * P + j0 + jl) * C + chunk * 16) * 49 + bin;
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) o[c * 49] = (float)c;
        } else {
            const int last = min(63, items - 1 - t0);
            const int La = max(0, lane - bin), Lb = min(last, lane + 48 - bin);
            const int kown = (4 - (bin & 3)) & 3;
            const bool own = lane + 3 <= Lb;
            bool single[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int r = (k + bin) & 3, Lq = lane - r;
                single[k] = !(Lq >= La && Lq + 3 <= Lb);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (own) {
                    f4 v = {1.0f, (float)g, 3.0f, 4.0f};
                    if (MODE == 2) __builtin_nontemporal_store(v, (f4 *)(o + (4 * g + kown) * 49));
                    else *(f4 *)(o + (4 * g + kown) * 49) = v;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (single[k]) o[(4 * g + k) * 49] = (float)k;
            }
        }
    }
}

// modes 19-21: same 3136-byte pieces as mode 1, different instruction shapes.
//  SHAPE 0: dword stores, 64 consecutive lanes = 256 B per instruction
//  SHAPE 1: dword stores, one 196-byte channel run (49 active lanes) per instruction
//  SHAPE 2: 16-byte stores, one 784-byte group span (49 active lanes) per instruction
template <int SHAPE>
__global__ void __launch_bounds__(1024) shape_kernel(float *out, int P, int C)
{
    const int chunk = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const int per = (P + gridDim.y - 1) / gridDim.y;
    const int j0 = split * per, j1 = min(P, j0 + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (SHAPE == 0) {
        const int n = (j1 - j0) * 784;
        for (int idx = threadIdx.x; idx < n; idx += 1024) {
            int jl = idx / 784, e = idx - jl * 784;
            out[(((long long)b * P + j0 + jl) * C + chunk * 16) * 49 + e] = (float)e;
        }
    } else {
        for (int jl = wave; jl < j1 - j0; jl += (int)(blockDim.x >> 6)) { // one RoI per wave iteration
            float *o = out + (((long long)b * P + j0 + jl) * C + chunk * 16) * 49;
            if (lane < 49) {
                if (SHAPE == 1) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) o[c * 49 + lane] = (float)c;
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f4 v = {1.0f, (float)g, 3.0f, 4.0f};
                        if (SHAPE == 3) __builtin_nontemporal_store(v, (f4 *)o + g * 49 + lane);
                        else ((f4 *)o)[g * 49 + lane] = v;
                    }
                }
            }
        }
    }
}

// modes 29-31: shape_kernel's per-wave pieces with the XCD-grouped block order of mode 12.
template <int SHAPE>
__global__ void __launch_bounds__(1024) shape_xcd_kernel(float *out, int P, int C, int splits)
{
    const int cpr = C / 16;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int chunk = slot % cpr;
    const int grp = (slot / cpr) * 8 + xcd;
    const int split = grp % splits, b = grp / splits;
    const int per = (P + splits - 1) / splits;
    const int j0 = split * per, j1 = min(P, j0 + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int jl = wave; jl < j1 - j0; jl += (int)(blockDim.x >> 6)) {
        float *o = out + (((long long)b * P + j0 + jl) * C + chunk * 16) * 49;
        if (lane < 49) {
            if (SHAPE == 1) {
#pragma unroll
                for (int c = 0; c < 16; ++c) o[c * 49 + lane] = (float)c;
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f4 v = {1.0f, (float)g, 3.0f, 4.0f};
                    if (SHAPE == 3) __builtin_nontemporal_store(v, (f4 *)o + g * 49 + lane);
                    else ((f4 *)o)[g * 49 + lane] = v;
                }
            }
        }
    }
}

// modes 35-37: as mode 29 (XCD-grouped, dword runs) but WPR waves share one RoI (each writes 16 / WPR of
// its 16 channels), i.e. fewer output rows in flight per workgroup.
template <int WPR>
__global__ void __launch_bounds__(1024) shape_xcd_share_kernel(float *out, int P, int C, int splits)
{
    const int cpr = C / 16;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int chunk = slot % cpr;
    const int grp = (slot / cpr) * 8 + xcd;
    const int split = grp % splits, b = grp / splits;
    const int per = (P + splits - 1) / splits;
    const int j0 = split * per, j1 = min(P, j0 + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = wave % WPR, team = wave / WPR;
    for (int jl = team; jl < j1 - j0; jl += 16 / WPR) {
        float *o = out + (((long long)b * P + j0 + jl) * C + chunk * 16) * 49;
        if (lane < 49) {
#pragma unroll
            for (int c = 0; c < 16 / WPR; ++c) o[(part * (16 / WPR) + c) * 49 + lane] = (float)c;
        }
    }
}

// mode 23: the item pattern of mode 16, but each WAVE walks a contiguous range of whole RoIs (64 items per
// iteration), so the two parts of a RoI that straddles an iteration are written by the same wave back to back.
__global__ void __launch_bounds__(1024) item_seq_kernel(float *out, int P, int C, int rois_per_wave)
{
    const int chunk = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
    const int per = (P + gridDim.y - 1) / gridDim.y;
    const int j0 = split * per, j1 = min(P, j0 + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = j0; base < j1; base += 16 * rois_per_wave) {
        const int ja = min(j1, base + wave * rois_per_wave), jb = min(j1, ja + rois_per_wave);
        const int items = (jb - ja) * 49;
        for (int t = lane; t < items; t += 64) {
            const int jl = t / 49, bin = t - jl * 49;
            float *o = out + (((long long)b * P + ja + jl) * C + chunk * 16) * 49 + bin;
#pragma unroll
            for (int c = 0; c < 16; ++c) o[c * 49] = (float)c;
        }
    }
}

// contiguous-per-block fill (each 256-thread block owns 4 KB x iters contiguous)
__global__ void __launch_bounds__(256) fill_block_kernel(float4 *out, long long n4, int iters)
{
    long long base = (long long)blockIdx.x * 256 * iters;
    for (int i = 0; i < iters; ++i) {
        long long k = base + (long long)i * 256 + threadIdx.x;
        if (k < n4) out[k] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
    }
}

// chunk-major order inside a block group: grid (splits*B, chunks) so that blocks sharing a RoI range are NOT adjacent
// mode 2: same bytes, but a block owns whole RoI rows (contiguous cpr*piece4 float4 each).
__global__ void __launch_bounds__(1024) row_kernel(float4 *out, long long rows, int row4)
{
    for (long long r = blockIdx.x; r < rows; r += gridDim.x)
        for (int f = threadIdx.x; f < row4; f += 1024) out[r * row4 + f] = make_float4(1.0f, (float)f, 3.0f, 4.0f);
}

int main()
{
    const int B = 16, P = 2000, C = 512;
    const long long n4 = (long long)B * P * C * 49 / 4;
    float4 *out;
    CHECK(hipMalloc(&out, n4 * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 10;
    for (int mode = 0; mode < 38; ++mode) {
        float ms = 0;
        for (int r = -2; r < reps; ++r) {
            if (r == 0) CHECK(hipEventRecord(e0, 0));
            switch (mode) {
            case 0: hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(1024), 0, 0, out, n4); break;
            case 1: hipLaunchKernelGGL(piece_kernel, dim3(32, 2, B), dim3(1024), 0, 0, out, P, 196, 32); break;
            case 2: hipLaunchKernelGGL(piece_kernel, dim3(32, 8, B), dim3(1024), 0, 0, out, P, 196, 32); break;
            case 3: hipLaunchKernelGGL(piece_kernel, dim3(8, 8, B), dim3(1024), 0, 0, out, P, 784, 8); break;
            case 4: hipLaunchKernelGGL(piece_kernel, dim3(64, 1, B), dim3(1024), 0, 0, out, P, 98, 64); break;
            case 5: hipLaunchKernelGGL(row_kernel, dim3(1024), dim3(1024), 0, 0, out, (long long)B * P, 6272); break;
            case 6: hipLaunchKernelGGL(piece_nt_kernel, dim3(32, 2, B), dim3(1024), 0, 0, out, P, 196, 32); break;
            case 8: hipLaunchKernelGGL(fill_block_kernel, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, out, n4, 4); break;
            case 9: hipLaunchKernelGGL(fill_block_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, out, n4, 1); break;
            case 10: hipLaunchKernelGGL(piece_kernel, dim3(32, 16, B), dim3(1024), 0, 0, out, P, 196, 32); break;
            case 12: hipLaunchKernelGGL((piece_xcd_kernel<false>), dim3(32 * 2 * B), dim3(1024), 0, 0, out, P, 196, 32, 2); break;
            case 13: hipLaunchKernelGGL((piece_xcd_kernel<true>), dim3(32 * 2 * B), dim3(1024), 0, 0, out, P, 196, 32, 2); break;
            case 14: hipLaunchKernelGGL((piece_xcd_kernel<false>), dim3(32 * 4 * B), dim3(1024), 0, 0, out, P, 196, 32, 4); break;
            case 15: hipLaunchKernelGGL((piece_xcd_kernel<true>), dim3(32 * 4 * B), dim3(1024), 0, 0, out, P, 196, 32, 4); break;
            case 16: hipLaunchKernelGGL((item_kernel<0>), dim3(32, 2, B), dim3(1024), 0, 0, (float *)out, P, C); break;
            case 17: hipLaunchKernelGGL((item_kernel<1>), dim3(32, 2, B), dim3(1024), 0, 0, (float *)out, P, C); break;
            case 18: hipLaunchKernelGGL((item_kernel<2>), dim3(32, 2, B), dim3(1024), 0, 0, (float *)out, P, C); break;
            case 19: hipLaunchKernelGGL((shape_kernel<0>), dim3(32, 2, B), dim3(1024), 0, 0, (float *)out, P, C); break;
            case 20: hipLaunchKernelGGL((shape_kernel<1>), dim3(32, 2, B), dim3(1024), 0, 0, (float *)out, P, C); break;
            case 21: hipLaunchKernelGGL((shape_kernel<2>), dim3(32, 2, B), dim3(1024), 0, 0, (float *)out, P, C); break;
            case 22: hipLaunchKernelGGL((shape_kernel<3>), dim3(32, 2, B), dim3(1024), 0, 0, (float *)out, P, C); break;
            case 23: hipLaunchKernelGGL(item_seq_kernel, dim3(32, 2, B), dim3(1024), 0, 0, (float *)out, P, C, 32); break;
            case 24: hipLaunchKernelGGL(item_seq_kernel, dim3(32, 2, B), dim3(1024), 0, 0, (float *)out, P, C, 64); break;
            case 25: hipLaunchKernelGGL(item_seq_kernel, dim3(32, 2, B), dim3(1024), 0, 0, (float *)out, P, C, 4); break;
            case 11: hipLaunchKernelGGL(piece_kernel, dim3(32, 1, B), dim3(1024), 0, 0, out, P, 196, 32); break;
            case 35: hipLaunchKernelGGL((shape_xcd_share_kernel<2>), dim3(32 * 2 * B), dim3(1024), 0, 0, (float *)out, P, C, 2); break;
            case 36: hipLaunchKernelGGL((shape_xcd_share_kernel<4>), dim3(32 * 2 * B), dim3(1024), 0, 0, (float *)out, P, C, 2); break;
            case 37: hipLaunchKernelGGL((shape_xcd_share_kernel<16>), dim3(32 * 2 * B), dim3(1024), 0, 0, (float *)out, P, C, 2); break;
            case 29: hipLaunchKernelGGL((shape_xcd_kernel<1>), dim3(32 * 2 * B), dim3(1024), 0, 0, (float *)out, P, C, 2); break;
            case 30: hipLaunchKernelGGL((shape_xcd_kernel<2>), dim3(32 * 2 * B), dim3(1024), 0, 0, (float *)out, P, C, 2); break;
            case 31: hipLaunchKernelGGL((shape_xcd_kernel<3>), dim3(32 * 2 * B), dim3(1024), 0, 0, (float *)out, P, C, 2); break;
            case 32: hipLaunchKernelGGL((shape_xcd_kernel<1>), dim3(32 * 4 * B), dim3(1024), 0, 0, (float *)out, P, C, 4); break;
            case 33: hipLaunchKernelGGL((shape_xcd_kernel<3>), dim3(32 * 4 * B), dim3(1024), 0, 0, (float *)out, P, C, 4); break;
            case 26: hipLaunchKernelGGL((shape_kernel<1>), dim3(32, 2, B), dim3(768), 0, 0, (float *)out, P, C); break;
            case 27: hipLaunchKernelGGL((shape_kernel<2>), dim3(32, 2, B), dim3(768), 0, 0, (float *)out, P, C); break;
            case 28: hipLaunchKernelGGL((shape_kernel<3>), dim3(32, 2, B), dim3(768), 0, 0, (float *)out, P, C); break;
            case 7: hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(1024), 0, 0, out, n4); break;
            }
        }
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        printf("mode %d: %.3f ms  %.2f TB/s\n", mode, ms, (double)n4 * 16 / ms / 1e9);
    }
    return 0;
}
