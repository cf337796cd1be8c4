// HR tail: 3x3 conv 64 -> 3 (+bias) on the SATU output, plus the bilinear residual of the
// unpadded centre LR frame (savsr_arch.py:738-739).  N = 3 output channels is far too narrow for
// MFMA (a 32-wide tile would waste >90 %), so this is a VALU kernel over an LDS-staged tile;
// it is bound by the 236 MB read of the feature map.
#include "common.hpp"

namespace savsr {

constexpr int TL_TH = 8, TL_TW = 32, TL_CH = 16;
constexpr int TL_R = TL_TH + 2, TL_C = TL_TW + 2;

__device__ __forceinline__ void bil_src(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;          // area_pixel_compute_source_index
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = fminf(fmaxf(s - (float)i0, 0.f), 1.f);
}

__global__ __launch_bounds__(256) void tail_kernel(const float* __restrict__ feat, const float* __restrict__ wgt, const float* __restrict__ bias,
                                                   const float* __restrict__ center, int h, int w, int H, int W, float* __restrict__ out) {
    __shared__ float tile[TL_CH * TL_R * TL_C];
    __shared__ float wl[3 * 64 * 9];
    const int tid = threadIdx.x;
    const int tx = tid & 31, ty = tid >> 5;
    const int X0 = blockIdx.x * TL_TW, Y0 = blockIdx.y * TL_TH;
    for (int i = tid; i < 3 * 64 * 9; i += 256) wl[i] = wgt[i];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const long long HW = (long long)H * W;
    for (int c0 = 0; c0 < 64; c0 += TL_CH) {
        __syncthreads();
        for (int e = tid; e < TL_CH * TL_R * TL_C; e += 256) {
            const int ch = e / (TL_R * TL_C);
            const int rem = e - ch * (TL_R * TL_C);
            const int r = rem / TL_C, c = rem - r * TL_C;
            const int gy = Y0 - 1 + r, gx = X0 - 1 + c;
            float v = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = feat[(long long)(c0 + ch) * HW + (long long)gy * W + gx];
            tile[e] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int ch = 0; ch < TL_CH; ++ch) {
            const float* tp = tile + ch * (TL_R * TL_C) + ty * TL_C + tx;
            const float* w0 = wl + (0 * 64 + c0 + ch) * 9;
            const float* w1 = wl + (1 * 64 + c0 + ch) * 9;
            const float* w2 = wl + (2 * 64 + c0 + ch) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float v = tp[(k / 3) * TL_C + (k % 3)];
                a0 += w0[k] * v; a1 += w1[k] * v; a2 += w2[k] * v;
            }
        }
    }
    const int X = X0 + tx, Y = Y0 + ty;
    if (X >= W || Y >= H) return;
    // F.interpolate(x_center, size=(H, W), mode='bilinear', align_corners=False), :739
    int y0, y1, x0, x1;
    float ly, lx;
    bil_src(Y, (float)h / (float)H, h, y0, y1, ly);
    bil_src(X, (float)w / (float)W, w, x0, x1, lx);
    float acc[3] = {a0 + bias[0], a1 + bias[1], a2 + bias[2]};
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const float* c = center + (long long)o * h * w;
        const float top = (1.f - lx) * c[y0 * w + x0] + lx * c[y0 * w + x1];
        const float bot = (1.f - lx) * c[y1 * w + x0] + lx * c[y1 * w + x1];
        out[(long long)o * HW + (long long)Y * W + X] = acc[o] + ((1.f - ly) * top + ly * bot);
    }
}

}  // namespace savsr

using namespace savsr;

extern "C" int savsr_tail_residual(const float* feat, const float* w, const float* b, const float* center, int h, int wd, int H, int W,
                                   float* out, void* stream) {
    if (!feat || !w || !b || !center || !out) return fail_arg("tail_residual: null pointer");
    if (h < 1 || wd < 1 || H < 1 || W < 1) return fail_arg("tail_residual: shape");
    dim3 grid((W + TL_TW - 1) / TL_TW, (H + TL_TH - 1) / TL_TH);
    hipLaunchKernelGGL(tail_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), feat, w, b, center, h, wd, H, W, out);
    return check_launch("tail_kernel");
}
