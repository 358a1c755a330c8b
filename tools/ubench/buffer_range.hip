// Which parts of a raw buffer address does the range check of gfx950 look at?  A buffer of 8 KiB holds its own dword
// indices; the descriptor says num_records = 1 KiB.  Loads at (per-lane offset, scalar offset) = (in, in), (in, out),
// (out, in): a checked out-of-range load returns 0, an unchecked one the data.  Measured on MI355X: all three
// out-of-range forms return 0, i.e. the check covers per-lane offset + scalar offset.  roi_cl.hip relies on it: its ring
// fills may run past the end of an image's slice (rows and cells travel in the scalar offset), and "do not fetch"
// travels in the per-lane offset.
//   hipcc --offload-arch=gfx950 -O2 -o buffer_range buffer_range.hip && ./buffer_range
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(const unsigned *buf, unsigned *out)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, 1024, 0x00020000);
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, 0, 0);             // in, in
    out[64 + lane] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, 4096, 0);     // lane offset in, scalar offset out
    out[128 + lane] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, 4096 + lane * 4, 0, 0); // lane offset out
    out[192 + lane] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, 0x40000000 + lane * 4, 0, 0);
}

int main()
{
    unsigned *buf, *out, h[2048], r[256];
    for (int i = 0; i < 2048; ++i) h[i] = 1000 + i;
    if (hipMalloc(&buf, 8192) != hipSuccess || hipMalloc(&out, 1024) != hipSuccess) return 1;
    if (hipMemcpy(buf, h, 8192, hipMemcpyHostToDevice) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, out);
    if (hipMemcpy(r, out, 1024, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("lane offset in,  scalar offset 0    : %u (data = 1005)\n", r[5]);
    printf("lane offset in,  scalar offset 4096 : %u (data = %u; 0 = range-checked)\n", r[64 + 5], 1000 + 1024 + 5);
    printf("lane offset 4096, scalar offset 0   : %u (0 = range-checked)\n", r[128 + 5]);
    printf("lane offset 2^30, scalar offset 0   : %u (0 = range-checked)\n", r[192 + 5]);
    return 0;
}
