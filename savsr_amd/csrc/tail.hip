// HR tail: 3x3 conv 64 -> 3 (+bias) on the SATU output, plus the bilinear residual of the
// unpadded centre LR frame (savsr_arch.py:738-739).  N = 3 output channels is far too narrow for
// MFMA (a 32-wide tile would waste >90 %), so this is a VALU kernel over an LDS-staged tile;
// it is bound by the 236 MB read of the feature map (config 2).
//
// Workgroup = 256 threads, tile = 16 rows x 32 cols; each thread owns two vertically adjacent
// pixels, so one channel costs 12 LDS reads for 54 FMAs.  The 1 728 weights are wave-uniform and
// come through the scalar cache (s_load), not LDS.  Interior tile columns are staged with 16-B
// loads when the row pitch allows it.
#include "common.hpp"

namespace savsr {

constexpr int TL_TH = 16, TL_TW = 32, TL_CH = 16;
constexpr int TL_R = TL_TH + 2;            // staged rows
constexpr int TL_S = 40;                   // staged row pitch: [3] left halo, [4..35] interior, [36] right halo

__device__ __forceinline__ void bil_src(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;          // area_pixel_compute_source_index
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = fminf(fmaxf(s - (float)i0, 0.f), 1.f);
}

__global__ __launch_bounds__(256) void tail_kernel(const float* __restrict__ feat, const float* __restrict__ wgt, const float* __restrict__ bias,
                                                   const float* __restrict__ center, int h, int w, int H, int W, long long FP, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float tile[TL_CH * TL_R * TL_S];
    const int tid = threadIdx.x;
    const int tx = tid & 31, ty = tid >> 5;
    const int X0 = blockIdx.x * TL_TW, Y0 = blockIdx.y * TL_TH;
    const long long HW = (long long)H * W;
    const bool vec = (W & 3) == 0 && (FP & 3) == 0 && X0 + TL_TW <= W;     // interior columns 16-B aligned and inside the image
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};      // per output channel: {pixel row 2 ty, row 2 ty + 1}

    for (int c0 = 0; c0 < 64; c0 += TL_CH) {
        __syncthreads();
        if (vec) {
            // interior: 16 ch x 18 rows x 8 float4 = 9 per thread, all in flight before the first LDS write
            // (a load -> store loop exposes one global latency per iteration); halos: 576 scalars
            f32x4 v4[9];
            float hv[3];
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int e = tid + i * 256;
                const int cr = e >> 3, it = e & 7;
                const int ch = cr / TL_R, r = cr - ch * TL_R;
                const int gy = Y0 - 1 + r;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gy >= 0 && gy < H) v = *reinterpret_cast<const f32x4*>(feat + (long long)(c0 + ch) * FP + (long long)gy * W + X0 + 4 * it);
                v4[i] = v;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int e = tid + i * 256;
                float v = 0.f;
                if (e < TL_CH * TL_R * 2) {
                    const int cr = e >> 1, side = e & 1;
                    const int ch = cr / TL_R, r = cr - ch * TL_R;
                    const int gy = Y0 - 1 + r, gx = side ? X0 + TL_TW : X0 - 1;
                    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = feat[(long long)(c0 + ch) * FP + (long long)gy * W + gx];
                }
                hv[i] = v;
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int e = tid + i * 256;
                *reinterpret_cast<f32x4*>(tile + (e >> 3) * TL_S + 4 + 4 * (e & 7)) = v4[i];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int e = tid + i * 256;
                if (e < TL_CH * TL_R * 2) tile[(e >> 1) * TL_S + ((e & 1) ? 36 : 3)] = hv[i];
            }
        } else {
            for (int e = tid; e < TL_CH * TL_R * 34; e += 256) {
                const int cr = e / 34, c = e - cr * 34;
                const int ch = cr / TL_R, r = cr - ch * TL_R;
                const int gy = Y0 - 1 + r, gx = X0 - 1 + c;
                float v = 0.f;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = feat[(long long)(c0 + ch) * FP + (long long)gy * W + gx];
                tile[(ch * TL_R + r) * TL_S + 3 + c] = v;
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int ch = 0; ch < TL_CH; ++ch) {
            // explicit 2-vectors {row r, row r + 1} of one column: the thread's two output pixels share every weight, so one
            // v_pk_fma_f32 per (tap, output channel) with the pair read by one ds_read2_b32 (left to the SLP vectoriser the
            // loop paired COLUMNS and spent 14 v_mov + 20 s_mov per channel on shuffles beside its 27 packed FMAs)
            const float* tp = tile + (ch * TL_R + 2 * ty) * TL_S + 3 + tx;
            const float* wc = wgt + (c0 + ch) * 9;           // wave-uniform -> scalar loads
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x2 v = {tp[r * TL_S + c], tp[(r + 1) * TL_S + c]};
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        const float wk = wc[o * 64 * 9 + r * 3 + c];
                        acc2[o] = __builtin_elementwise_fma(f32x2{wk, wk}, v, acc2[o]);
                    }
                }
        }
    }
    float a[2][3];
#pragma unroll
    for (int o = 0; o < 3; ++o) { a[0][o] = acc2[o][0]; a[1][o] = acc2[o][1]; }
    const int X = X0 + tx;
    if (X >= W) return;
    // F.interpolate(x_center, size=(H, W), mode='bilinear', align_corners=False), :739
    int x0, x1;
    float lx;
    bil_src(X, (float)w / (float)W, w, x0, x1, lx);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int Y = Y0 + 2 * ty + q;
        if (Y >= H) continue;
        int y0, y1;
        float ly;
        bil_src(Y, (float)h / (float)H, h, y0, y1, ly);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float* c = center + (long long)o * h * w;
            const float top = (1.f - lx) * c[y0 * w + x0] + lx * c[y0 * w + x1];
            const float bot = (1.f - lx) * c[y1 * w + x0] + lx * c[y1 * w + x1];
            out[(long long)o * HW + (long long)Y * W + X] = (a[q][o] + bias[o]) + ((1.f - ly) * top + ly * bot);
        }
    }
}


// ------------------------------------------------------------------------------------------
// Tail of the tail-projected form (see satu.hip, HR stage): the SATU HR stage already applied the 3x3 tail conv's
// CHANNEL contraction, P[3 tap + o][Y][X] = Wt[o][:][ky][kx] . F[:][Y][X] (tap = 3 ky + kx); what is left of
// savsr_arch.py:738-739 is the spatial part and the residual:
//     out[o][Y][X] = tail_b[o] + sum_{ky,kx} P[3 (3 ky + kx) + o][Y + ky - 1][X + kx - 1]  (zero outside)  + bilinear(center)
// Pure HBM streaming: every element of the 27 planes is read once (99.5 MB at 720x1280 instead of the 236 MB feature map),
// 11 MB written.  One thread = 4 consecutive pixels of a row of ONE output channel (16-B loads; the kx = 0 / 2 planes are
// read at a +-4-B offset, which global loads allow): 9 plane loads per thread.  Measured at 720x1280 (99.5 MB + 11 MB): all
// three channels per thread (27 loads) 31.9 us, one channel per thread 22.6 us (4.9 TB/s); 8 pixels per thread 49 us,
// non-temporal loads 43.7 us, one / two / eight rows per workgroup 34.3 / 22.3 / 24.5 us -- the light thread wins.
// ------------------------------------------------------------------------------------------
#ifndef TG_OSPLIT
#define TG_OSPLIT 1               // 1: one output channel per thread (grid z = 3): 9 plane loads per thread instead of 27
#endif
#ifndef TG_ROWS
#define TG_ROWS 4                 // image rows per 256-thread workgroup
#endif
#ifndef TG_PX
#define TG_PX 4                   // pixels per thread on the vector path (4: 16-B accesses, 2: 8-B)
#endif
typedef float tg_vec __attribute__((ext_vector_type(TG_PX)));
typedef float tg_vecu __attribute__((ext_vector_type(TG_PX), aligned(4)));

template <bool VEC>   // VEC: W % TG_PX == 0 and the plane pitch a multiple of 4 floats -> vector accesses; else one pixel per thread
__global__ __launch_bounds__(256) void tail_gather_kernel(const float* __restrict__ P, long long PP, const float* __restrict__ bias,
                                                          const float* __restrict__ center, int h, int w, int H, int W, float* __restrict__ out) {
    constexpr int NPX = VEC ? TG_PX : 1;
    constexpr int TPR = 256 / TG_ROWS;                           // threads per row
    constexpr int NO = TG_OSPLIT ? 1 : 3;                        // output channels per thread
    const int X = (blockIdx.x * TPR + (threadIdx.x % TPR)) * NPX;
    const int Y = blockIdx.y * TG_ROWS + (threadIdx.x / TPR);
    const int O0 = TG_OSPLIT ? (int)blockIdx.z : 0;
    if (X >= W || Y >= H) return;
    float acc[NO][NPX];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int i = 0; i < NPX; ++i) acc[o][i] = 0.f;
    const bool interior = VEC && X >= 4 && X + NPX + 4 <= W;     // every shifted vector of this thread lies inside the row
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = Y + ky - 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const float* row = P + (long long)(3 * (3 * ky + kx) + O0 + o) * PP + (long long)yy * W;
                if (VEC && (interior || kx == 1)) {
                    const tg_vecu v = *reinterpret_cast<const tg_vecu*>(row + X + kx - 1);
#pragma unroll
                    for (int i = 0; i < NPX; ++i) acc[o][i] += v[i];
                } else {
#pragma unroll
                    for (int i = 0; i < NPX; ++i) {
                        const int xx = X + i + kx - 1;
                        if (xx >= 0 && xx < W) acc[o][i] += row[xx];
                    }
                }
            }
    }
    // F.interpolate(x_center, size=(H, W), mode='bilinear', align_corners=False), :739
    int y0, y1;
    float ly;
    bil_src(Y, (float)h / (float)H, h, y0, y1, ly);
    const long long HW = (long long)H * W;
    float res[NO][NPX];
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        int x0, x1;
        float lx;
        bil_src(X + i, (float)w / (float)W, w, x0, x1, lx);
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const float* c = center + (long long)(O0 + o) * h * w;
            const float top = (1.f - lx) * c[y0 * w + x0] + lx * c[y0 * w + x1];
            const float bot = (1.f - lx) * c[y1 * w + x0] + lx * c[y1 * w + x1];
            res[o][i] = (acc[o][i] + bias[O0 + o]) + ((1.f - ly) * top + ly * bot);
        }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        float* dst = out + (long long)(O0 + o) * HW + (long long)Y * W + X;
        if (VEC) {
            tg_vec r;
#pragma unroll
            for (int i = 0; i < NPX; ++i) r[i] = res[o][i];
            *reinterpret_cast<tg_vec*>(dst) = r;
        } else dst[0] = res[o][0];
    }
}

// Tail of the ROW-SUMMED form (savsr_satu_hr_tail_q): the HR stage has already added the three horizontal taps of every (tap row ky,
// colour o) group g = 3 ky + o inside its 32-pixel segments -- Q[g][Y][X] -- and left the two terms per segment border that cross
// it in a side buffer: seam[Y][seg][0][g] = what the FIRST pixel of segment seg still needs from its left neighbour, seam[Y][seg][1][g]
// = what its LAST pixel needs from its right neighbour.  Left here: the three vertical taps, the seams, bias, bilinear residual:
//     out[o][Y][X] = tail_b[o] + sum_ky (Q[3 ky + o][Y + ky - 1][X] + seams) + bilinear(center)
// 33 MB read instead of 99.5 (720x1280), all of it with aligned 16-B loads.
template <bool VEC>
__global__ __launch_bounds__(256) void tail_gather_q_kernel(const float* __restrict__ Q, long long QP, const float* __restrict__ seam,
                                                            int nseg, const float* __restrict__ bias, const float* __restrict__ center, int h, int w,
                                                            int H, int W, float* __restrict__ out) {
    constexpr int NPX = VEC ? 4 : 1;
    constexpr int TPR = 256 / TG_ROWS;
    const int X = (blockIdx.x * TPR + (threadIdx.x % TPR)) * NPX;
    const int Y = blockIdx.y * TG_ROWS + (threadIdx.x / TPR);
    const int o = blockIdx.z;
    if (X >= W || Y >= H) return;
    float acc[NPX];
#pragma unroll
    for (int i = 0; i < NPX; ++i) acc[i] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = Y + ky - 1;
        if (yy < 0 || yy >= H) continue;
        const int g = 3 * ky + o;
        const float* row = Q + (long long)g * QP + (long long)yy * W + X;
        if (VEC) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += v[i];
        } else acc[0] += row[0];
        const float* sa = seam + (long long)yy * nseg * 18 + g;            // [row][segment][side][group]
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int x = X + i;
            if ((x & 31) == 0 && x > 0) acc[i] += sa[(x >> 5) * 18];
            if ((x & 31) == 31 && x + 1 < W) acc[i] += sa[(x >> 5) * 18 + 9];
        }
    }
    int y0, y1;
    float ly;
    bil_src(Y, (float)h / (float)H, h, y0, y1, ly);
    const long long HW = (long long)H * W;
    float res[NPX];
    const float* c = center + (long long)o * h * w;
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        int x0, x1;
        float lx;
        bil_src(X + i, (float)w / (float)W, w, x0, x1, lx);
        const float top = (1.f - lx) * c[y0 * w + x0] + lx * c[y0 * w + x1];
        const float bot = (1.f - lx) * c[y1 * w + x0] + lx * c[y1 * w + x1];
        res[i] = (acc[i] + bias[o]) + ((1.f - ly) * top + ly * bot);
    }
    float* dst = out + (long long)o * HW + (long long)Y * W + X;
    if (VEC) *reinterpret_cast<f32x4*>(dst) = f32x4{res[0], res[VEC ? 1 : 0], res[VEC ? 2 : 0], res[VEC ? 3 : 0]};
    else dst[0] = res[0];
}

}  // namespace savsr

using namespace savsr;

extern "C" int savsr_tail_gather_q(const float* q9, int64_t q_plane, const float* seam, int64_t seam_floats, const float* b, const float* center,
                                   int h, int wd, int H, int W, float* out, void* stream) {
    if (!q9 || !seam || !b || !center || !out) return fail_arg("tail_gather_q: null pointer");
    const int nseg = (W + 31) / 32;
    if (h < 1 || wd < 1 || H < 1 || W < 1 || q_plane < (int64_t)H * W || seam_floats < (int64_t)H * nseg * 18) return fail_arg("tail_gather_q: shape");
    const bool vec = (W & 3) == 0 && (q_plane & 3) == 0 && !((reinterpret_cast<uintptr_t>(q9) | reinterpret_cast<uintptr_t>(out)) & 15);
    const int npx = vec ? 4 : 1, tpr = 256 / TG_ROWS;
    dim3 grid((W + tpr * npx - 1) / (tpr * npx), (H + TG_ROWS - 1) / TG_ROWS, 3);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (vec) hipLaunchKernelGGL(tail_gather_q_kernel<true>, grid, dim3(256), 0, st, q9, (long long)q_plane, seam, nseg, b, center, h, wd, H, W, out);
    else hipLaunchKernelGGL(tail_gather_q_kernel<false>, grid, dim3(256), 0, st, q9, (long long)q_plane, seam, nseg, b, center, h, wd, H, W, out);
    return check_launch("tail_gather_q_kernel");
}

extern "C" int savsr_tail_residual(const float* feat, int64_t feat_plane, const float* w, const float* b, const float* center, int h, int wd,
                                   int H, int W, float* out, void* stream) {
    if (!feat || !w || !b || !center || !out) return fail_arg("tail_residual: null pointer");
    if (h < 1 || wd < 1 || H < 1 || W < 1 || feat_plane < (int64_t)H * W) return fail_arg("tail_residual: shape");
    if (reinterpret_cast<uintptr_t>(feat) & 15) { set_error("tail_residual: feat must be 16-byte aligned"); return SAVSR_E_ALIGN; }
    dim3 grid((W + TL_TW - 1) / TL_TW, (H + TL_TH - 1) / TL_TH);
    hipLaunchKernelGGL(tail_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), feat, w, b, center, h, wd, H, W, (long long)feat_plane, out);
    return check_launch("tail_kernel");
}


extern "C" int savsr_tail_gather(const float* p27, int64_t p_plane, const float* b, const float* center, int h, int wd, int H, int W,
                                 float* out, void* stream) {
    if (!p27 || !b || !center || !out) return fail_arg("tail_gather: null pointer");
    if (h < 1 || wd < 1 || H < 1 || W < 1 || p_plane < (int64_t)H * W) return fail_arg("tail_gather: shape");
    const bool vec = (W & (TG_PX - 1)) == 0 && (p_plane & 3) == 0 && !((reinterpret_cast<uintptr_t>(p27) | reinterpret_cast<uintptr_t>(out)) & 15);
    const int npx = vec ? TG_PX : 1, tpr = 256 / TG_ROWS;
    dim3 grid((W + tpr * npx - 1) / (tpr * npx), (H + TG_ROWS - 1) / TG_ROWS, TG_OSPLIT ? 3 : 1);
    if (vec) hipLaunchKernelGGL(tail_gather_kernel<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p27, (long long)p_plane, b, center, h, wd, H, W, out);
    else hipLaunchKernelGGL(tail_gather_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p27, (long long)p_plane, b, center, h, wd, H, W, out);
    return check_launch("tail_gather_kernel");
}
