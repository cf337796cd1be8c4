// Small bandwidth-bound helpers of the OSAdapt mask branch and the input padding (gfx950).
#include "common.hpp"

namespace savsr {

// nn.AvgPool2d(2), savsr_arch.py:193
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int c, int h, int w) {
    const int ho = h / 2, wo = w / 2;
    const long long n = (long long)c * ho * wo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % wo);
        const int y = (int)((i / wo) % ho);
        const int ch = (int)(i / ((long long)wo * ho));
        const float* p = in + ((long long)ch * h + 2 * y) * w + 2 * x;
        out[i] = (p[0] + p[1] + p[w] + p[w + 1]) * 0.25f;
    }
}

// F.interpolate(bilinear, align_corners=False) source index/lambda as ATen computes them
// (area_pixel_compute_source_index + guard_index_and_lambda), scale = in/out.
__device__ __forceinline__ void bilinear_src(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = fminf(fmaxf(s - (float)i0, 0.f), 1.f);
}

// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False), savsr_arch.py:202
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ in, float* __restrict__ out, int c, int h, int w) {
    const int ho = 2 * h, wo = 2 * w;
    const long long n = (long long)c * ho * wo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % wo);
        const int y = (int)((i / wo) % ho);
        const int ch = (int)(i / ((long long)wo * ho));
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_src(y, 0.5f, h, y0, y1, ly);
        bilinear_src(x, 0.5f, w, x0, x1, lx);
        const float* p = in + (long long)ch * h * w;
        const float top = (1.f - lx) * p[y0 * w + x0] + lx * p[y0 * w + x1];
        const float bot = (1.f - lx) * p[y1 * w + x0] + lx * p[y1 * w + x1];
        out[i] = (1.f - ly) * top + ly * bot;
    }
}

// F.pad(..., [0, pw, 0, ph], mode='reflect'), savsr_arch.py:681-690
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int h, int w, int hp, int wp) {
    const long long tot = (long long)n * hp * wp;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % wp);
        const int y = (int)((i / wp) % hp);
        const int pl = (int)(i / ((long long)wp * hp));
        const int sx = x < w ? x : 2 * (w - 1) - x;
        const int sy = y < h ? y : 2 * (h - 1) - y;
        out[i] = in[((long long)pl * h + sy) * w + sx];
    }
}

static inline int grid_for(long long n) {
    long long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace savsr

using namespace savsr;

extern "C" int savsr_avgpool2(const float* in, float* out, int c, int h, int w, void* stream) {
    if (!in || !out) return fail_arg("avgpool2: null pointer");
    if (c < 1 || h < 2 || w < 2 || (h & 1) || (w & 1)) return fail_arg("avgpool2: h and w must be even");
    hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for((long long)c * (h / 2) * (w / 2))), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, c, h, w);
    return check_launch("avgpool2_kernel");
}

extern "C" int savsr_upsample2x(const float* in, float* out, int c, int h, int w, void* stream) {
    if (!in || !out) return fail_arg("upsample2x: null pointer");
    if (c < 1 || h < 1 || w < 1) return fail_arg("upsample2x: shape");
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for((long long)c * h * w * 4)), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, c, h, w);
    return check_launch("upsample2x_kernel");
}

extern "C" int savsr_reflect_pad(const float* in, float* out, int n, int h, int w, int hp, int wp, void* stream) {
    if (!in || !out) return fail_arg("reflect_pad: null pointer");
    if (n < 1 || h < 2 || w < 2 || hp < h || wp < w || hp > h + 1 || wp > w + 1) return fail_arg("reflect_pad: shape");
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(grid_for((long long)n * hp * wp)), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, n, h, w, hp, wp);
    return check_launch("reflect_pad_kernel");
}
