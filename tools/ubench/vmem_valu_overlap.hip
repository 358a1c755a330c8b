// Calibration: do L1-hitting vector loads and VALU work overlap on a CU (gfx950)?
//   mode 0: every wave does FMAs only            mode 1: every wave does loads only (28 x 256 B per step, L1 hits)
//   mode 2: even waves load, odd waves do FMAs    mode 3: every wave does both per step (loads of step k+1 in flight
//                                                          under the FMAs of step k)
//   hipcc --offload-arch=gfx950 -O3 -o vmem_valu_overlap vmem_valu_overlap.hip && ./vmem_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define NL 28

__device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

template <int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
k(float *out, const float *in, int iters, int fmas)
{
    // every workgroup reads its own 28 x 256 B = 7 KB: stays in the CU's 32 KB L1 with 4 workgroups... (one per wave)
    const int lane = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (blockIdx.x % 64) * 2048), 0, 8192, 0x00020000);
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = (float)lane * i;
    unsigned w = __builtin_amdgcn_readfirstlane(__float_as_uint(in[0]) | 0x3f800000u);
    const bool loader = MODE == 1 || MODE == 3 || (MODE == 2 && (blockIdx.x & 1) == 0);
    const bool worker = MODE == 0 || MODE == 3 || (MODE == 2 && (blockIdx.x & 1) == 1);
    float v[NL];
    for (int j = 0; j < NL; ++j) v[j] = 0.0f;
    if (loader)
        for (int j = 0; j < NL; ++j) v[j] = ld(r, lane * 4, j * 256);
    for (int it = 0; it < iters; ++it) {
        float s = 0.0f;
        if (loader) {
            for (int j = 0; j < NL; ++j) s += v[j];              // consume (28 adds) ...
            for (int j = 0; j < NL; ++j) v[j] = ld(r, lane * 4, ((j + it) % NL) * 256); // ... and refill
        }
        if (worker) {
            for (int f = 0; f < fmas; ++f)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(w), "v"(a[(i + 1) & 15]));
        }
        a[0] += s;
    }
    float t = 0;
    for (int i = 0; i < 16; ++i) t += a[i];
    for (int j = 0; j < NL; ++j) t += v[j];
    out[blockIdx.x * 64 + lane] = t;
}

template <int MODE>
static float run(const char *name, int waves_per_cu, int iters, int fmas)
{
    float *in, *out;
    const int grid = 256 * waves_per_cu;
    CHECK(hipMalloc(&in, 64 * 2048 * 4 + 65536));
    CHECK(hipMalloc(&out, grid * 64 * 4));
    CHECK(hipMemset(in, 0, 64 * 2048 * 4 + 65536));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, out, in, iters, fmas);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%-44s waves/CU %2d  %.3f ms\n", name, waves_per_cu, ms);
    CHECK(hipFree(in)); CHECK(hipFree(out));
    return ms;
}

// loads only, by width: W dwords per lane (wave load of W x 256 bytes, contiguous), the same bytes per step
template <int W>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
kw(float *out, const float *in, int iters)
{
    const int lane = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (blockIdx.x % 64) * 2048), 0, 8192, 0x00020000);
    typedef float fw __attribute__((ext_vector_type(W)));
    typedef unsigned uw __attribute__((ext_vector_type(W)));
    constexpr int N = NL / W;
    fw v[N];
    float t = 0;
    for (int it = 0; it < iters; ++it) {
        for (int j = 0; j < N; ++j) {
            const int so = ((j + it) % N) * 256 * W;
            if constexpr (W == 1) v[j][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, so, 0));
            if constexpr (W == 2) v[j] = __builtin_bit_cast(fw, __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, so, 0));
            if constexpr (W == 4) v[j] = __builtin_bit_cast(fw, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, so, 0));
        }
        for (int j = 0; j < N; ++j) {
            t += v[j][0] + v[j][W - 1];
        }
    }
    out[blockIdx.x * 64 + lane] = t;
}

template <int W>
static void runw(int waves_per_cu, int iters)
{
    float *in, *out;
    const int grid = 256 * waves_per_cu;
    CHECK(hipMalloc(&in, 64 * 2048 * 4 + 65536));
    CHECK(hipMalloc(&out, grid * 64 * 4));
    CHECK(hipMemset(in, 0, 64 * 2048 * 4 + 65536));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((kw<W>), dim3(grid), dim3(64), 0, 0, out, in, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double bytes = (double)grid * iters * (NL / W) * W * 256;
    printf("loads of %d dword(s) per lane, %2d waves/CU: %.3f ms, %.1f B/clk/CU at 2.1 GHz, %.1f TB/s\n", W, waves_per_cu, ms,
           bytes / 256 / (ms * 1e-3 * 2.1e9), bytes / ms / 1e9);
    CHECK(hipFree(in)); CHECK(hipFree(out));
}

int main()
{
    runw<1>(16, 4000); runw<2>(16, 4000); runw<4>(16, 4000);
    runw<1>(8, 4000); runw<4>(8, 4000); runw<4>(4, 4000);
    const int iters = 4000;
    // 3 x 16 = 48 FMAs per step beside 28 loads + 28 adds: about the row loop's mix
    run<0>("FMAs only (48 per step)", 16, iters, 3);
    run<1>("loads only (28 x 256 B + 28 adds per step)", 16, iters, 3);
    run<3>("both in every wave", 16, iters, 3);
    run<0>("FMAs only, 8 waves", 8, iters, 3);
    run<1>("loads only, 8 waves", 8, iters, 3);
    run<2>("8 loader waves + 8 FMA waves", 16, iters, 3);
    run<3>("both in every wave, 8 waves", 8, iters, 3);
    return 0;
}
