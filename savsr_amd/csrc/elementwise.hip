// Small bandwidth-bound helpers on channel-last maps (gfx950): the OSAdapt mask branch's
// pool / upsample and the input-window packing (with the reflect padding folded in).
#include "common.hpp"

namespace savsr {

// nn.AvgPool2d(2), savsr_arch.py:193;  [h][w][c] -> [h/2][w/2][c]
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int c, int h, int w) {
    const int ho = h / 2, wo = w / 2;
    const long long n = (long long)c * ho * wo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % c);
        const int x = (int)((i / c) % wo);
        const int y = (int)(i / ((long long)c * wo));
        const float* p = in + (((long long)(2 * y) * w + 2 * x) * c) + ch;
        out[i] = (p[0] + p[c] + p[(long long)w * c] + p[(long long)w * c + c]) * 0.25f;
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
        const int ch = (int)(i % c);
        const int x = (int)((i / c) % wo);
        const int y = (int)(i / ((long long)c * wo));
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_src(y, 0.5f, h, y0, y1, ly);
        bilinear_src(x, 0.5f, w, x0, x1, lx);
        const float* p = in + ch;
        const float top = (1.f - lx) * p[((long long)y0 * w + x0) * c] + lx * p[((long long)y0 * w + x1) * c];
        const float bot = (1.f - lx) * p[((long long)y1 * w + x0) * c] + lx * p[((long long)y1 * w + x1) * c];
        out[i] = (1.f - ly) * top + ly * bot;
    }
}

// Input windows for WindowUnit_l1 (savsr_arch.py:448-454, generate_it :661-668) with
// pad_spatial's reflect padding (:681-690) folded in.  lq: [T][3][h][w] planar;
// out: [T-2][hp][wp][16] channel-last, position q <-> window centre t = q + 1:
//   ch 0-2 = frame t (x_c), 3-5 = frame t-1, 6-8 = frame t+1 (x_sup), 9-15 = 0.
__global__ __launch_bounds__(256) void pack_windows_kernel(const float* __restrict__ lq, float* __restrict__ out, int T, int h, int w, int hp, int wp) {
    const long long n = (long long)(T - 2) * hp * wp * 4;        // float4 units
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int q4 = (int)(i & 3);
        const long long pi = i >> 2;
        const int x = (int)(pi % wp);
        const int y = (int)((pi / wp) % hp);
        const int q = (int)(pi / ((long long)wp * hp));
        const int sx = x < w ? x : 2 * (w - 1) - x;
        const int sy = y < h ? y : 2 * (h - 1) - y;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = 4 * q4 + j;
            if (ch < 9) {
                const int which = ch / 3, comp = ch - 3 * which;
                const int frame = (q + 1) + (which == 0 ? 0 : (which == 1 ? -1 : 1));
                v[j] = lq[(((long long)frame * 3 + comp) * h + sy) * w + sx];
            }
        }
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}

// bench.py's clock probe: one wave, no LDS; per window: shader clock = (s_memtime delta) / (s_memrealtime delta) x 100 MHz
__global__ __launch_bounds__(64) void clock_probe_kernel(long long* out, int window_ticks, int windows) {
    for (int wi = 0; wi < windows; ++wi) {
        const long long r0 = (long long)__builtin_amdgcn_s_memrealtime();
        const long long c0 = (long long)__builtin_amdgcn_s_memtime();
        long long r1 = r0;
        while (r1 - r0 < window_ticks) {
            __builtin_amdgcn_s_sleep(32);
            r1 = (long long)__builtin_amdgcn_s_memrealtime();
        }
        const long long c1 = (long long)__builtin_amdgcn_s_memtime();
        r1 = (long long)__builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) { out[2 * wi] = c1 - c0; out[2 * wi + 1] = r1 - r0; }
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

extern "C" int savsr_clock_probe(int64_t* out, int window_ticks, int windows, void* stream) {
    if (!out || window_ticks < 1 || windows < 1 || windows > 64 || (long long)window_ticks * windows > 100000000ll) return fail_arg("clock_probe: null pointer / ticks or windows out of range");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), reinterpret_cast<long long*>(out), window_ticks, windows);
    return check_launch("clock_probe_kernel");
}

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

extern "C" int savsr_pack_windows(const float* lq, float* out, int T, int h, int w, int hp, int wp, void* stream) {
    if (!lq || !out) return fail_arg("pack_windows: null pointer");
    if (T < 3 || h < 2 || w < 2 || hp < h || wp < w || hp > h + 1 || wp > w + 1) return fail_arg("pack_windows: shape");
    if (reinterpret_cast<uintptr_t>(out) & 15) { set_error("pack_windows: out must be 16-byte aligned"); return SAVSR_E_ALIGN; }
    hipLaunchKernelGGL(pack_windows_kernel, dim3(grid_for((long long)(T - 2) * hp * wp * 4)), dim3(256), 0, static_cast<hipStream_t>(stream), lq, out, T, h, w, hp, wp);
    return check_launch("pack_windows_kernel");
}
