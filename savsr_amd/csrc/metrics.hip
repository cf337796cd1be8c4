// PSNR-Y / SSIM-Y of one output frame on the GPU with the reference's numerics (the step
// AFTER the hot path; the reference does it on the CPU per frame, lbasicsr/models/video_base_model.py:86-100):
//   tensor2img      clamp(0,1) * 255, round half to even, uint8            lbasicsr/utils/img_util.py:66-90
//   to_y_channel    u8 / 255 (fp32) -> BT.601 luma in fp64 -> fp32 -> * 255 (fp32)
//                                                                          metrics/metric_util.py:32-45, utils/color_util.py:59-65
//   calculate_psnr  fp64 mean of squared Y differences                     metrics/psnr_ssim.py:42-48
//   calculate_ssim  11x11 Gaussian (sigma 1.5) 'valid' window statistics in fp64, mean of the SSIM map
//                                                                          metrics/psnr_ssim.py:172-200
// Each value goes through the same sequence of precisions as the numpy restatement in savsr_amd/metrics.py (the oracle of
// this row); only the order of the two fp64 reductions differs (per-tile partial sums, then a fixed-order sum), i.e.
// ~1e-15 relative.
//
// Workgroup = 256 threads = one 16 x 16 tile of SSIM-map positions: the 26 x 26 Y values of both images are built in
// LDS, the five windowed sums (x, y, x^2, y^2, xy) are formed separably (rows first, as _blur_valid does), and the
// tile's sum of SSIM values and of squared differences (each pixel counted by exactly one tile) go to partial[blk].
#include "common.hpp"

namespace savsr {

constexpr int MT = 16;                 // SSIM-map positions per tile side (27 KB of LDS per workgroup)
constexpr int MW = MT + 10;            // Y values per tile side (11-tap window)

__device__ __forceinline__ double y_of(const float* __restrict__ img, long long plane, long long idx) {
    // RGB planar fp32 in [0, 1] (un-clamped) -> the reference's quantised luma
    float q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = img[c * plane + idx];
        v = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;          // clamp_(0, 1); (img * 255.0)
        v = rintf(v);                                     // .round(): half to even, like numpy
        q[c] = v / 255.0f;                                // uint8 -> float32 / 255.0 (fp32 division)
    }
    // np.dot(bgr, [24.966, 128.553, 65.481]) + 16.0 in float64, / 255.0, -> float32, * 255.0 in float32
    const double y64 = ((double)q[2] * 24.966 + (double)q[1] * 128.553 + (double)q[0] * 65.481 + 16.0) / 255.0;
    const float y32 = (float)y64;
    return (double)(y32 * 255.0f);
}

// test_y_channel: false -- the metric of every colour plane of the quantised image (calculate_psnr: the mean over H x W x 3 of the
// squared uint8 differences; calculate_ssim: the mean of the three planes' SSIM, psnr_ssim.py:115-129): plane ch of tensor2img's output
__device__ __forceinline__ double level_of(const float* __restrict__ img, long long plane, long long idx, int ch) {
    float v = img[ch * plane + idx];
    v = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;
    return (double)rintf(v);
}

struct MetricsParams {
    const float* sr;
    const float* gt;
    long long sr_plane, gt_plane;
    int H, W, crop;
    int y_channel;                     // 1: BT.601 luma (one plane of blocks); 0: the three colour planes (blockIdx.z)
    double* partial;                   // [planes][blocks][2] = sum of SSIM values, sum of squared differences
};

__global__ __launch_bounds__(256) void metrics_y_kernel(const MetricsParams p) {
    __shared__ double ya[MW * MW], yb[MW * MW];
    __shared__ double rows[5][MT * MW];                   // after the vertical pass: [map][out row][in col]
    __shared__ double red[2][256];
    const int tid = threadIdx.x;
    const int Hc = p.H - 2 * p.crop, Wc = p.W - 2 * p.crop;           // cropped image
    const int oh = Hc - 10, ow = Wc - 10;                               // SSIM map ('valid')
    const int ox0 = blockIdx.x * MT, oy0 = blockIdx.y * MT;            // tile origin in the SSIM map = in the cropped image
    double sq = 0.0;
    for (int e = tid; e < MW * MW; e += 256) {
        const int r = e / MW, c = e - r * MW;
        const int y = oy0 + r, x = ox0 + c;
        double a = 0.0, b = 0.0;
        if (y < Hc && x < Wc) {
            const long long idx = (long long)(y + p.crop) * p.W + (x + p.crop);
            a = p.y_channel ? y_of(p.sr, p.sr_plane, idx) : level_of(p.sr, p.sr_plane, idx, blockIdx.z);
            b = p.y_channel ? y_of(p.gt, p.gt_plane, idx) : level_of(p.gt, p.gt_plane, idx, blockIdx.z);
            // PSNR: every cropped pixel is owned by the tile whose 16 x 16 core contains it; the last tile row / column
            // also owns the 10-pixel rim beyond the SSIM map
            const bool own_r = r < MT || blockIdx.y == gridDim.y - 1, own_c = c < MT || blockIdx.x == gridDim.x - 1;
            if (own_r && own_c) sq += (a - b) * (a - b);
        }
        ya[e] = a;
        yb[e] = b;
    }
    __syncthreads();
    // Gaussian window, normalised exp(-(i-5)^2 / (2 * 1.5^2)) (cv2.getGaussianKernel(11, 1.5))
    double k[11];
    {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 11; ++i) { const double d = (double)(i - 5); k[i] = exp(-(d * d) / (2.0 * 1.5 * 1.5)); s += k[i]; }
#pragma unroll
        for (int i = 0; i < 11; ++i) k[i] /= s;
    }
    // vertical pass (tmp += k[i] * a[i : i + h - 10, :] for i = 0..10)
    for (int e = tid; e < MT * MW; e += 256) {
        const int r = e / MW, c = e - r * MW;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const double a = ya[(r + i) * MW + c], b = yb[(r + i) * MW + c];
            s0 += k[i] * a; s1 += k[i] * b; s2 += k[i] * (a * a); s3 += k[i] * (b * b); s4 += k[i] * (a * b);
        }
        rows[0][e] = s0; rows[1][e] = s1; rows[2][e] = s2; rows[3][e] = s3; rows[4][e] = s4;
    }
    __syncthreads();
    // horizontal pass + SSIM map value, one position per thread
    const double c1 = (0.01 * 255) * (0.01 * 255), c2 = (0.03 * 255) * (0.03 * 255);
    double ss = 0.0;
    for (int e = tid; e < MT * MT; e += 256) {
        const int r = e / MT, c = e - r * MT;
        if (oy0 + r >= oh || ox0 + c >= ow) continue;
        double m[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 11; ++j)
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] += k[j] * rows[q][r * MW + c + j];
        const double mu1 = m[0], mu2 = m[1];
        const double mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const double s1 = m[2] - mu1_sq, s2 = m[3] - mu2_sq, s12 = m[4] - mu12;
        ss += ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2));
    }
    red[0][tid] = ss;
    red[1][tid] = sq;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {                   // fixed-order tree: deterministic
        if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        const long long b = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        p.partial[2 * b] = red[0][0];
        p.partial[2 * b + 1] = red[1][0];
    }
}

// out[0] = PSNR-Y (inf for identical images), out[1] = SSIM-Y
__global__ __launch_bounds__(256) void metrics_finalize_kernel(const double* partial, int nblk, double n_px, double n_map, double* out) {
    __shared__ double red[2][256];
    const int tid = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = tid; i < nblk; i += 256) { a += partial[2 * i]; b += partial[2 * i + 1]; }
    red[0][tid] = a;
    red[1][tid] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double mse = red[1][0] / n_px;
        out[0] = mse == 0.0 ? __builtin_inf() : 10.0 * log10(255.0 * 255.0 / mse);
        out[1] = red[0][0] / n_map;
    }
}

}  // namespace savsr

using namespace savsr;

extern "C" int savsr_metrics_blocks(int H, int W, int crop_border) {
    const int oh = H - 2 * crop_border - 10, ow = W - 2 * crop_border - 10;
    if (H < 1 || W < 1 || crop_border < 0 || oh < 1 || ow < 1) return -1;
    return ((ow + MT - 1) / MT) * ((oh + MT - 1) / MT);
}

extern "C" int savsr_metrics_psnr_ssim(const float* sr, int64_t sr_plane, const float* gt, int64_t gt_plane, int H, int W, int crop_border,
                                       int test_y_channel, double* partial, double* out, void* stream) {
    if (!sr || !gt || !partial || !out) return fail_arg("metrics: null pointer");
    const int planes = test_y_channel ? 1 : 3;
    const int nblk = savsr_metrics_blocks(H, W, crop_border);
    if (nblk < 1) return fail_arg("metrics: the cropped image must be at least 11 x 11");
    if (sr_plane < (int64_t)H * W || gt_plane < (int64_t)H * W) return fail_arg("metrics: plane pitch < H*W");
    const int Hc = H - 2 * crop_border, Wc = W - 2 * crop_border;
    MetricsParams p;
    p.sr = sr; p.gt = gt; p.sr_plane = sr_plane; p.gt_plane = gt_plane; p.H = H; p.W = W; p.crop = crop_border; p.partial = partial;
    p.y_channel = test_y_channel ? 1 : 0;
    dim3 grid((Wc - 10 + MT - 1) / MT, (Hc - 10 + MT - 1) / MT, planes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(metrics_y_kernel, grid, dim3(256), 0, st, p);
    int rc = check_launch("metrics_y_kernel");
    if (rc) return rc;
    // (three planes: the pooled means ARE the reference's -- one mean over H x W x 3 for the PSNR, the mean of three equally sized maps' means for the SSIM)
    hipLaunchKernelGGL(metrics_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nblk * planes, (double)Hc * Wc * planes,
                       (double)(Hc - 10) * (Wc - 10) * planes, out);
    return check_launch("metrics_finalize_kernel");
}

extern "C" int savsr_metrics_psnr_ssim_y(const float* sr, int64_t sr_plane, const float* gt, int64_t gt_plane, int H, int W, int crop_border,
                                         double* partial, double* out, void* stream) {
    return savsr_metrics_psnr_ssim(sr, sr_plane, gt, gt_plane, H, W, crop_border, 1, partial, out, stream);
}
