// Anti-aliased bicubic resize of planar fp32 images: the reference's LR synthesis, the step BEFORE the hot path
// (SURVEY section 8 row f2).  lbasicsr/data/data_util.py:371-420 `arbitrary_scale_downsample` (degradation 'BI', mode
// 'torch') calls torchvision T.Resize(size, BICUBIC, antialias=True) on float tensors, i.e. ATen's separable
// _upsample_bicubic2d_aa (align_corners=False): per output index a window [xmin, xmin + xsize) of the input axis and
// normalised cubic (a = -0.5) weights stretched by the scale; width pass first, then height, fp32 accumulation in tap
// order.  The index / weight tables are built on the host exactly as ATen builds them (savsr_amd/resize_gpu.py); this
// file is the weighted gather of one axis.  HBM-bound: 7 x 3 x 720 x 1280 -> 180 x 320 reads 77 MB once.
#include "common.hpp"

namespace savsr {

struct ResizeParams {
    const float* in;
    float* out;
    int planes, h, w;          // input planes of [h][w]
    int out_size, max_taps;    // output extent of the resized axis; row pitch of the weight table
    const int* xmin;
    const int* xsize;
    const float* wt;           // [out_size][max_taps]
};

// axis = width: out[plane][y][xo] = sum_j in[plane][y][xmin[xo] + j] * wt[xo][j]
__global__ __launch_bounds__(256) void resize_w_kernel(const ResizeParams p) {
    const long long total = (long long)p.planes * p.h * p.out_size;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int xo = (int)(e % p.out_size);
        const long long row = e / p.out_size;                      // plane * h + y
        const float* src = p.in + row * p.w + p.xmin[xo];
        const float* wv = p.wt + (long long)xo * p.max_taps;
        const int n = p.xsize[xo];
        float acc = 0.f;
        for (int j = 0; j < n; ++j) acc += src[j] * wv[j];
        p.out[e] = acc;
    }
}

// The same through LDS: a workgroup stages RS_ROWS whole input rows with coalesced 16-B loads, then every thread gathers
// its outputs from LDS (the direct kernel issues ~17 overlapping strided dword loads per output: 128 us for the 7-frame
// 720x1280 clip, all address-path time).  Used when the rows fit (RS_ROWS * w floats of LDS) and w % 4 == 0.
constexpr int RS_ROWS = 4;
__global__ __launch_bounds__(256) void resize_w_lds_kernel(const ResizeParams p) {
    extern __shared__ __attribute__((aligned(16))) float rows[];          // [RS_ROWS][w]
    const long long row0 = (long long)blockIdx.x * RS_ROWS;                // rows are (plane, y) pairs: planes * h of them
    const long long nrows = (long long)p.planes * p.h;
    const int w4 = p.w / 4;
    for (int e = threadIdx.x; e < RS_ROWS * w4; e += 256) {
        const int r = e / w4, c = e - r * w4;
        if (row0 + r < nrows) reinterpret_cast<f32x4*>(rows)[e] = reinterpret_cast<const f32x4*>(p.in + (row0 + r) * p.w)[c];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < RS_ROWS * p.out_size; e += 256) {
        const int r = e / p.out_size, xo = e - r * p.out_size;
        if (row0 + r >= nrows) continue;
        const float* src = rows + r * p.w + p.xmin[xo];
        const float* wv = p.wt + (long long)xo * p.max_taps;
        const int n = p.xsize[xo];
        float acc = 0.f;
        for (int j = 0; j < n; ++j) acc += src[j] * wv[j];                 // tap order, like the direct kernel (bit-identical)
        p.out[(row0 + r) * p.out_size + xo] = acc;
    }
}

// axis = height: out[plane][yo][x] = sum_j in[plane][ymin[yo] + j][x] * wt[yo][j]
__global__ __launch_bounds__(256) void resize_h_kernel(const ResizeParams p) {
    const long long total = (long long)p.planes * p.out_size * p.w;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int x = (int)(e % p.w);
        const long long t = e / p.w;
        const int yo = (int)(t % p.out_size);
        const long long plane = t / p.out_size;
        const float* src = p.in + (plane * p.h + p.xmin[yo]) * p.w + x;
        const float* wv = p.wt + (long long)yo * p.max_taps;
        const int n = p.xsize[yo];
        float acc = 0.f;
        for (int j = 0; j < n; ++j) acc += src[(long long)j * p.w] * wv[j];
        p.out[e] = acc;
    }
}

}  // namespace savsr

using namespace savsr;

extern "C" int savsr_resize_aa_axis(const float* in, int planes, int h, int w, int axis, int out_size, const int32_t* xmin,
                                    const int32_t* xsize, const float* weights, int max_taps, float* out, void* stream) {
    if (!in || !out || !xmin || !xsize || !weights) return fail_arg("resize_aa_axis: null pointer");
    if (planes < 1 || h < 1 || w < 1 || out_size < 1 || max_taps < 1 || (axis != 0 && axis != 1)) return fail_arg("resize_aa_axis: shape / axis (0 = width, 1 = height)");
    ResizeParams p;
    p.in = in; p.out = out; p.planes = planes; p.h = h; p.w = w; p.out_size = out_size; p.max_taps = max_taps;
    p.xmin = xmin; p.xsize = xsize; p.wt = weights;
    const long long total = axis == 0 ? (long long)planes * h * out_size : (long long)planes * out_size * w;
    long long g = (total + 255) / 256;
    if (g > 65535 * 16) g = 65535 * 16;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)RS_ROWS * w * sizeof(float);
    const bool staged = axis == 0 && (w % 4) == 0 && lds <= 64 * 1024 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    if (staged) {
        const long long nrows = (long long)planes * h;
        hipLaunchKernelGGL(resize_w_lds_kernel, dim3((unsigned)((nrows + RS_ROWS - 1) / RS_ROWS)), dim3(256), lds, st, p);
    } else if (axis == 0) hipLaunchKernelGGL(resize_w_kernel, dim3((unsigned)g), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)g), dim3(256), 0, st, p);
    return check_launch("resize_aa_axis");
}
