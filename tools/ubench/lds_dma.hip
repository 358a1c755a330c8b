// Calibration + semantics check: buffer_load_dwordx4 ... lds (LDS-DMA, 1 KiB per wave instruction) on gfx950.
//   1. the LDS image of one instruction = the 1 KiB global segment its lanes address (lane i: 16 bytes at i * 16)
//   2. rate of L1-hitting DMA fills per CU, alone and with ds_read_b32 taps + FMAs consuming the rows
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma lds_dma.hip && ./lds_dma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, lds_void *dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
}

__global__ void __launch_bounds__(64) check_kernel(const float *in, float *out)
{
    __shared__ __attribute__((aligned(1024))) float s[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) s[i] = -1.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, 4096 * 4, 0x00020000);
    dma16(r, (lds_void *)(s + 256), lane * 16, 1024);       // global bytes 1024..2047 -> LDS floats 256..511
    dma16(r, (lds_void *)(s + 1024), lane * 16, 4096);      // global bytes 4096..5119 -> LDS floats 1024..1279
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 2048; i += 64) out[i] = s[i];
}

// rows of NCELL cells x 256 B: DMA fills of 4 cells each into a two-slot ring, 21 ds_read_b32 taps per row
template <int MODE, int NDMA>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
rate_kernel(const float *in, float *out, int iters)
{
    __shared__ __attribute__((aligned(1024))) float s[2 * NDMA * 256]; // two slots of NDMA KiB
    const int lane = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (blockIdx.x % 32) * 4096), 0, 16384, 0x00020000);
    float acc[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < NDMA; ++d) dma16(r, (lds_void *)(s + d * 256), lane * 16, d * 1024);
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        float *nxt = s + (cur ^ 1) * NDMA * 256;
#pragma unroll
        for (int d = 0; d < NDMA; ++d) dma16(r, (lds_void *)(nxt + d * 256), lane * 16, ((d + it) % NDMA) * 1024);
        if (MODE == 1) {
            // wait for the current slot only: the NDMA fills just issued stay in flight
            if (NDMA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (NDMA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            const float *row = s + cur * NDMA * 256;
            float v[21];
#pragma unroll
            for (int j = 0; j < 21; ++j) v[j] = row[((j * 5 + it) % (NDMA * 4)) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 21; ++j) acc[j % 7] = __builtin_fmaf(v[j], 1.0001f, acc[j % 7]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = s[lane];
    for (int p = 0; p < 7; ++p) t += acc[p];
    out[blockIdx.x * 64 + lane] = t;
}

template <int MODE, int NDMA>
static void run(const char *name, int waves_per_cu, int iters)
{
    float *in, *out;
    const int grid = 256 * waves_per_cu;
    CHECK(hipMalloc(&in, 32 * 4096 * 4 + 65536));
    CHECK(hipMalloc(&out, grid * 64 * 4));
    CHECK(hipMemset(in, 0, 32 * 4096 * 4 + 65536));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((rate_kernel<MODE, NDMA>), dim3(grid), dim3(64), 0, 0, out, in, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double bytes = (double)grid * iters * NDMA * 1024;
    printf("%-40s %2d waves/CU: %.3f ms, %.1f fill B/clk/CU at 2.1 GHz, %.0f cycles per row and CU\n", name, waves_per_cu, ms,
           bytes / 256 / (ms * 1e-3 * 2.1e9), ms * 1e-3 * 2.1e9 / ((double)waves_per_cu * iters));
    CHECK(hipFree(in)); CHECK(hipFree(out));
}

int main()
{
    float *in, *out, h[4096], o[2048];
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    CHECK(hipMalloc(&in, sizeof h)); CHECK(hipMalloc(&out, sizeof o));
    CHECK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, in, out);
    CHECK(hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 2048; ++i) {
        float want = -1.0f;
        if (i >= 256 && i < 512) want = (float)(256 + (i - 256));
        if (i >= 1024 && i < 1280) want = (float)(1024 + (i - 1024));
        if (o[i] != want) { if (bad < 8) printf("LDS[%d] = %g, expected %g\n", i, o[i], want); ++bad; }
    }
    printf("semantics: %s (%d mismatches)\n", bad ? "DIFFERENT" : "LDS image = the addressed 1 KiB segments", bad);
    if (bad) return 1;
    run<0, 4>("fills only, 4 KiB rows", 12, 4000);
    run<0, 4>("fills only, 4 KiB rows", 6, 4000);
    run<1, 4>("fills + 21 taps + 21 FMAs, 4 KiB rows", 12, 4000);
    run<1, 6>("fills + 21 taps + 21 FMAs, 6 KiB rows", 12, 4000);
    run<1, 4>("fills + 21 taps + 21 FMAs, 4 KiB rows", 6, 4000);
    return 0;
}
