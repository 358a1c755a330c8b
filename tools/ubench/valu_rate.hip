// Calibration: what a wave64 VALU / SALU / branch instruction costs a SIMD at a given occupancy (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
k(float *out, const float *in, int iters, unsigned mask)
{
    float a[16];
    float t[16];
    for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x + i * 256]; t[i] = in[threadIdx.x + 4096 + i * 256]; }
    unsigned w = __builtin_amdgcn_readfirstlane(__float_as_uint(in[0]));
    float vw = in[threadIdx.x & 63];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { // 16 independent in-place FMAs with a scalar weight
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(w), "v"(t[i]));
        } else if (MODE == 1) { // the same behind a scalar guard per pair (taken / not taken by mask)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                if ((mask >> (i >> 1)) & 1) {
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(w), "v"(t[i]));
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i + 1]) : "s"(w), "v"(t[i + 1]));
                }
            }
        } else if (MODE == 2) { // weights fetched by v_readlane with a run-time lane, then the guard
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                unsigned ww = __builtin_amdgcn_readlane(__float_as_uint(vw), (it + i) & 63);
                if ((ww << 1) != 0u) {
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(ww), "v"(t[i]));
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i + 1]) : "s"(ww), "v"(t[i + 1]));
                }
            }
        } else if (MODE == 3) { // packed
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 acc = {a[i], a[i + 1]}, tt = {t[i], t[i + 1]};
                unsigned long long w2 = w;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(w2), "v"(tt));
                a[i] = acc.x; a[i + 1] = acc.y;
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int WPE>
static void run(const char *name, int wgs_per_cu, unsigned mask, int per_iter)
{
    float *in, *out;
    const int grid = 256 * wgs_per_cu, iters = 20000;
    CHECK(hipMalloc(&in, 65536 * 4));
    CHECK(hipMalloc(&out, grid * 256 * 4));
    CHECK(hipMemset(in, 0, 65536 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE, WPE>), dim3(grid), dim3(256), 0, 0, out, in, iters, mask);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: wgs_per_cu waves; instructions per wave = iters * per_iter
    const double cyc = ms * 1e-3 * 2.1e9;
    printf("%-34s waves/SIMD %d  %.3f ms  %.2f cycles per instruction per SIMD (at 2.1 GHz), %.2f per wave-instruction\n", name,
           wgs_per_cu, ms, cyc / ((double)iters * per_iter * wgs_per_cu), cyc / ((double)iters * per_iter));
    CHECK(hipFree(in)); CHECK(hipFree(out));
}

int main()
{
    run<0, 1>("16 v_fmac", 1, 0, 16);
    run<0, 2>("16 v_fmac", 2, 0, 16);
    run<0, 4>("16 v_fmac", 4, 0, 16);
    run<3, 2>("8 v_pk_fma", 2, 0, 8);
    run<3, 4>("8 v_pk_fma", 4, 0, 8);
    run<1, 2>("guards all taken (16 fmac + 8 br)", 2, 0xFF, 16);
    run<1, 2>("guards none taken (8 skipped)", 2, 0x00, 8);
    run<1, 2>("guards alternate (8 fmac + 8 br)", 2, 0x55, 8);
    run<2, 2>("readlane + guard, w = 0 (8 skipped)", 2, 0, 8);
    return 0;
}
