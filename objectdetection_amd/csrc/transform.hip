// transform.hip -- SURVEY.md 8f row 3: batch assembly in front of the path.
// generalized_transform.py:58-64 (F.interpolate bilinear, align_corners=False, scale recomputed from
// the integer sizes) and utils.py:339-372 (copy into a zero canvas) fused: one launch writes one
// image's whole (C, canvas_h, canvas_w) slice of the batch -- resized pixels inside (out_h, out_w),
// zeros outside -- so the resized image is never materialised on its own.  HBM-bound:
// reads C*in_h*in_w*4 B once (taps hit L2), writes C*canvas_h*canvas_w*4 B.
#include "common.h"

__global__ void __launch_bounds__(256)
resize_pad_kernel(const float *__restrict__ src, int C, int in_h, int in_w, int out_h, int out_w,
                  float scale_h, float scale_w, float *__restrict__ dst, int canvas_h, int canvas_w)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= canvas_w) return;
    const long long cstride = (long long)canvas_h * canvas_w;
    float *d = dst + (long long)y * canvas_w + x;
    if (y >= out_h || x >= out_w) {
        for (int c = 0; c < C; ++c) d[c * cstride] = 0.0f;
        return;
    }
    // torch upsample_bilinear2d, align_corners = false: src = scale * (dst + 0.5) - 0.5, clamped at 0
    float sy = scale_h * ((float)y + 0.5f) - 0.5f;
    float sx = scale_w * ((float)x + 0.5f) - 0.5f;
    sy = sy < 0.0f ? 0.0f : sy;
    sx = sx < 0.0f ? 0.0f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = y0 > in_h - 1 ? in_h - 1 : y0;
    x0 = x0 > in_w - 1 ? in_w - 1 : x0;
    const int y1 = y0 < in_h - 1 ? y0 + 1 : y0;
    const int x1 = x0 < in_w - 1 ? x0 + 1 : x0;
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const long long sstride = (long long)in_h * in_w;
    const float *r0 = src + (long long)y0 * in_w, *r1 = src + (long long)y1 * in_w;
    for (int c = 0; c < C; ++c) {
        const float p00 = r0[c * sstride + x0], p01 = r0[c * sstride + x1];
        const float p10 = r1[c * sstride + x0], p11 = r1[c * sstride + x1];
        d[c * cstride] = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
    }
}

// The whole batch in one launch: the images are separate tensors of different sizes, so their pointers and
// shapes travel in a by-value table (kernel argument memory, indexed by blockIdx.z: scalar loads), up to
// RESIZE_TABLE images per launch.  16 launches of ~12 us each were launch-bound.
#define RESIZE_TABLE 32
struct ResizeTable {
    const float *src[RESIZE_TABLE];
    int in_h[RESIZE_TABLE], in_w[RESIZE_TABLE], out_h[RESIZE_TABLE], out_w[RESIZE_TABLE];
    float scale_h[RESIZE_TABLE], scale_w[RESIZE_TABLE]; // input / output in fp32, computed on the host as in frcnn_resize_pad_f32
};

__global__ void __launch_bounds__(256)
resize_pad_batch_kernel(ResizeTable t, int C, float *__restrict__ dst, int canvas_h, int canvas_w)
{
    const int i = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= canvas_w) return;
    const int in_h = t.in_h[i], in_w = t.in_w[i], out_h = t.out_h[i], out_w = t.out_w[i];
    const float *__restrict__ src = t.src[i];
    const long long cstride = (long long)canvas_h * canvas_w;
    float *d = dst + (long long)i * C * cstride + (long long)y * canvas_w + x;
    if (y >= out_h || x >= out_w) {
        for (int c = 0; c < C; ++c) d[c * cstride] = 0.0f;
        return;
    }
    const float scale_h = t.scale_h[i], scale_w = t.scale_w[i];
    float sy = scale_h * ((float)y + 0.5f) - 0.5f;
    float sx = scale_w * ((float)x + 0.5f) - 0.5f;
    sy = sy < 0.0f ? 0.0f : sy;
    sx = sx < 0.0f ? 0.0f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = y0 > in_h - 1 ? in_h - 1 : y0;
    x0 = x0 > in_w - 1 ? in_w - 1 : x0;
    const int y1 = y0 < in_h - 1 ? y0 + 1 : y0;
    const int x1 = x0 < in_w - 1 ? x0 + 1 : x0;
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const long long sstride = (long long)in_h * in_w;
    const float *r0 = src + (long long)y0 * in_w, *r1 = src + (long long)y1 * in_w;
    for (int c = 0; c < C; ++c) {
        const float p00 = r0[c * sstride + x0], p01 = r0[c * sstride + x1];
        const float p10 = r1[c * sstride + x0], p11 = r1[c * sstride + x1];
        d[c * cstride] = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
    }
}

extern "C" int frcnn_resize_pad_batch_f32(const float *const *src_host, const int32_t *shape_host, int n, int C,
                                          float *dst, int canvas_h, int canvas_w, void *stream)
{
    FRCNN_REQUIRE(n > 0 && C > 0 && canvas_h > 0 && canvas_w > 0 && src_host && shape_host && dst,
                  "resize_pad_batch: bad arguments");
    FRCNN_REQUIRE(canvas_h <= 65535, "resize_pad_batch: canvas too tall");
    for (int i0 = 0; i0 < n; i0 += RESIZE_TABLE) {
        const int m = n - i0 < RESIZE_TABLE ? n - i0 : RESIZE_TABLE;
        ResizeTable t;
        for (int j = 0; j < RESIZE_TABLE; ++j) {
            const int i = i0 + (j < m ? j : 0);
            const int32_t *sh = shape_host + 4 * i;
            FRCNN_REQUIRE(src_host[i] && sh[0] > 0 && sh[1] > 0 && sh[2] > 0 && sh[3] > 0,
                          "resize_pad_batch: bad image %d", i);
            FRCNN_REQUIRE(sh[2] <= canvas_h && sh[3] <= canvas_w,
                          "resize_pad_batch: a %d x %d image does not fit the %d x %d canvas", sh[2], sh[3], canvas_h,
                          canvas_w);
            t.src[j] = src_host[i];
            t.in_h[j] = sh[0];
            t.in_w[j] = sh[1];
            t.out_h[j] = sh[2];
            t.out_w[j] = sh[3];
            t.scale_h[j] = (float)sh[0] / (float)sh[2];
            t.scale_w[j] = (float)sh[1] / (float)sh[3];
        }
        hipLaunchKernelGGL(resize_pad_batch_kernel, dim3((canvas_w + 255) / 256, canvas_h, m), dim3(256), 0,
                           (hipStream_t)stream, t, C, dst + (long long)i0 * C * canvas_h * canvas_w, canvas_h, canvas_w);
        FRCNN_CHECK_LAUNCH("resize_pad_batch");
    }
    return FRCNN_OK;
}

extern "C" int frcnn_resize_pad_f32(const float *src, int C, int in_h, int in_w, int out_h, int out_w,
                                    float *dst, int canvas_h, int canvas_w, void *stream)
{
    FRCNN_REQUIRE(C > 0 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0 && canvas_h > 0 && canvas_w > 0,
                  "resize_pad: bad shape");
    FRCNN_REQUIRE(out_h <= canvas_h && out_w <= canvas_w,
                  "resize_pad: a %d x %d image does not fit the %d x %d canvas", out_h, out_w, canvas_h, canvas_w);
    FRCNN_REQUIRE(canvas_h <= 65535, "resize_pad: canvas too tall");
    FRCNN_REQUIRE(src && dst, "resize_pad: null pointer");
    // area_pixel_compute_scale with no explicit scale: input / output in fp32
    const float scale_h = (float)in_h / (float)out_h, scale_w = (float)in_w / (float)out_w;
    hipLaunchKernelGGL(resize_pad_kernel, dim3((canvas_w + 255) / 256, canvas_h), dim3(256), 0, (hipStream_t)stream,
                       src, C, in_h, in_w, out_h, out_w, scale_h, scale_w, dst, canvas_h, canvas_w);
    FRCNN_CHECK_LAUNCH("resize_pad");
    return FRCNN_OK;
}
