// Dense 3x3 / 1x1 convolution as an fp32-MFMA implicit GEMM (gfx950).
//
//   D[co][px] = sum_k A[co][k] * B[k][px],  k = (input channel, tap)
//
// A = weights (pre-packed in exactly the order the lanes consume them), B = input pixels read
// from an LDS tile with halo, D = v_mfma_f32_32x32x2_f32 accumulators whose column index is the
// lane (= pixel), so the epilogue stores 32 consecutive pixels of one output channel per
// half-wave: 128-B coalesced stores in the channel-planar layout.
//
// Block = 256 threads = 4 waves; block tile = COT output channels x (4 rows x 32 cols) pixels;
// wave w owns row w.  The reduction runs over chunks of CK input channels; each chunk's input
// tile and weight slab are staged global -> registers -> LDS one chunk ahead of the MFMAs
// (register-staged double buffer, one barrier per chunk).
//
// Replaces: every nn.Conv2d / F.conv2d of savsr_arch.py (see include/savsr_hip.h).
#include "common.hpp"

namespace savsr {

constexpr int CONV_TH = 4;
constexpr int CONV_TW = 32;


struct ConvParams {
    const float* src[SAVSR_MAX_SRC];
    long long src_plane[SAVSR_MAX_SRC];
    int src_row[SAVSR_MAX_SRC];
    int nsrc, src_ch;
    int h, w, cin, cout;
    int nchunk;
    const float* wpacked;
    const float* bias;
    int act;
    float slope;
    const float* mul_px;
    const float* res1;
    const float* res2;
    float res2_scale;
    float* out;
    long long out_plane;
    int out_row;
};

template <int KS, int CK, int NT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvParams p) {
    constexpr int TAPS = KS * KS;
    constexpr int HALO = KS / 2;
    constexpr int IR = CONV_TH + 2 * HALO;
    constexpr int IC = CONV_TW + 2 * HALO;
    constexpr int IN_N = CK * IR * IC;               // staged input floats per chunk
    constexpr int COT = 32 * NT;
    constexpr int W_N = TAPS * CK * COT;             // staged weight floats per chunk
    constexpr int IN_IT = (IN_N + 255) / 256;
    constexpr int W_IT = (W_N / 4 + 255) / 256;
    constexpr int BUF = IN_N + W_N;

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int px = lane & 31;
    const int x0 = blockIdx.x * CONV_TW;
    const int y0 = blockIdx.y * CONV_TH;
    const int cob = blockIdx.z;

    float in_reg[IN_IT];
    f32x4 w_reg[W_IT];

    auto stage_load = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < IN_IT; ++i) {
            const int e = tid + i * 256;
            float v = 0.f;
            if (e < IN_N) {
                const int ch = e / (IR * IC);
                const int rem = e - ch * (IR * IC);
                const int r = rem / IC;
                const int c = rem - r * IC;
                const int gy = y0 - HALO + r;
                const int gx = x0 - HALO + c;
                const int ci = chunk * CK + ch;
                if (ci < p.cin && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) {
                    const int s = ci / p.src_ch;
                    const int lc = ci - s * p.src_ch;
                    const float* base = p.src[0];
                    long long pl = p.src_plane[0];
                    int rw = p.src_row[0];
                    if (s == 1) { base = p.src[1]; pl = p.src_plane[1]; rw = p.src_row[1]; }
                    if (s == 2) { base = p.src[2]; pl = p.src_plane[2]; rw = p.src_row[2]; }
                    if (s == 3) { base = p.src[3]; pl = p.src_plane[3]; rw = p.src_row[3]; }
                    if (s == 4) { base = p.src[4]; pl = p.src_plane[4]; rw = p.src_row[4]; }
                    v = base[(long long)lc * pl + (long long)gy * rw + gx];
                }
            }
            in_reg[i] = v;
        }
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.wpacked + ((long long)cob * p.nchunk + chunk) * W_N);
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            const int e = tid + i * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < W_N / 4) v = wsrc[e];
            w_reg[i] = v;
        }
    };
    auto stage_store = [&](int buf) {
        float* in_l = smem + buf * BUF;
        float* w_l = in_l + IN_N;
#pragma unroll
        for (int i = 0; i < IN_IT; ++i) {
            const int e = tid + i * 256;
            if (e < IN_N) in_l[e] = in_reg[i];
        }
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            const int e = tid + i * 256;
            if (e < W_N / 4) reinterpret_cast<f32x4*>(w_l)[e] = w_reg[i];
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    stage_load(0);
    stage_store(0);
    __syncthreads();

    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
        const int buf = chunk & 1;
        const bool more = chunk + 1 < p.nchunk;
        if (more) stage_load(chunk + 1);

        const float* in_l = smem + buf * BUF;
        const float* w_l = in_l + IN_N;
        // lane's B base: channel `half` of pair 0, own row, own column
        const float* bptr = in_l + half * (IR * IC) + wave * IC + px;
        const float* aptr = w_l + half * COT + px;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
            for (int cp = 0; cp < CK / 2; ++cp) {
                const float b = bptr[(2 * cp) * (IR * IC) + ky * IC + kx];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float a = aptr[(tap * (CK / 2) + cp) * 2 * COT + t * 32];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
        }
        if (more) stage_store(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------
    const int y = y0 + wave;
    const int x = x0 + px;
    if (y >= p.h || x >= p.w) return;
    const float mul = p.mul_px ? p.mul_px[(long long)y * p.w + x] : 1.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cob * COT + t * 32 + acc_row(r, half);
            if (co < p.cout) {
                float v = acc[t][r];
                if (p.bias) v += p.bias[co];
                if (p.act == SAVSR_ACT_RELU) v = fmaxf(v, 0.f);
                else if (p.act == SAVSR_ACT_LRELU) v = v > 0.f ? v : v * p.slope;
                else if (p.act == SAVSR_ACT_SIGMOID) v = sigmoidf_(v);
                v *= mul;
                const long long o = (long long)co * p.out_plane + (long long)y * p.out_row + x;
                if (p.res1) v += p.res1[o];
                if (p.res2) v += p.res2_scale * p.res2[o];
                p.out[o] = v;
            }
        }
    }
}

template <int KS, int CK, int NT>
static int launch_conv(const ConvParams& p, hipStream_t st) {
    constexpr int TAPS = KS * KS, HALO = KS / 2;
    constexpr int IN_N = CK * (CONV_TH + 2 * HALO) * (CONV_TW + 2 * HALO);
    constexpr int W_N = TAPS * CK * 32 * NT;
    constexpr size_t lds = 2 * (IN_N + W_N) * sizeof(float);
    static_assert(IN_N % 4 == 0, "weight slab must stay 16-byte aligned in LDS");
    dim3 grid((p.w + CONV_TW - 1) / CONV_TW, (p.h + CONV_TH - 1) / CONV_TH, (p.cout + 32 * NT - 1) / (32 * NT));
    hipLaunchKernelGGL((conv_mfma_kernel<KS, CK, NT>), grid, dim3(256), lds, st, p);
    return check_launch("conv_mfma_kernel");
}

}  // namespace savsr

using namespace savsr;

extern "C" int64_t savsr_conv_packed_floats(int cout, int cin, int ksize) {
    if (cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return -1;
    const int ck = conv_ck(ksize), cot = conv_cot(cout);
    const int64_t nchunk = (cin + ck - 1) / ck, ncob = (cout + cot - 1) / cot;
    return ncob * nchunk * ksize * ksize * ck * cot;
}

extern "C" int64_t savsr_conv_pack_index(int cout, int cin, int ksize, int co, int ci, int tap) {
    const int ck = conv_ck(ksize), cot = conv_cot(cout);
    const int64_t nchunk = (cin + ck - 1) / ck;
    const int cob = co / cot, col = co % cot;
    const int chunk = ci / ck, cl = ci % ck;
    const int cp = cl / 2, hh = cl % 2;
    return ((((int64_t)(cob * nchunk + chunk) * (ksize * ksize) + tap) * (ck / 2) + cp) * 2 + hh) * cot + col;
}

extern "C" int savsr_conv2d(const savsr_conv_desc* d, void* stream) {
    if (!d) return fail_arg("conv: null descriptor");
    if (d->ksize != 1 && d->ksize != 3) return fail_arg("conv: ksize must be 1 or 3");
    if (d->nsrc < 1 || d->nsrc > SAVSR_MAX_SRC || d->src_ch < 1) return fail_arg("conv: nsrc/src_ch");
    if (d->cin != d->nsrc * d->src_ch) return fail_arg("conv: cin != nsrc*src_ch");
    if (d->h < 1 || d->w < 1 || d->cout < 1) return fail_arg("conv: shape");
    if (!d->wpacked || !d->out) return fail_arg("conv: null weights/out");
    if ((reinterpret_cast<uintptr_t>(d->wpacked) & 15) != 0) {
        set_error("conv: wpacked must be 16-byte aligned");
        return SAVSR_E_ALIGN;
    }
    ConvParams p;
    for (int i = 0; i < SAVSR_MAX_SRC; ++i) {
        const bool on = i < d->nsrc;
        if (on && !d->src[i]) return fail_arg("conv: null source");
        p.src[i] = on ? d->src[i] : nullptr;
        p.src_plane[i] = on ? d->src_plane[i] : 0;
        p.src_row[i] = on ? d->src_row[i] : 0;
    }
    p.nsrc = d->nsrc; p.src_ch = d->src_ch;
    p.h = d->h; p.w = d->w; p.cin = d->cin; p.cout = d->cout;
    const int ck = conv_ck(d->ksize);
    p.nchunk = (d->cin + ck - 1) / ck;
    p.wpacked = d->wpacked; p.bias = d->bias; p.act = d->act; p.slope = d->slope;
    p.mul_px = d->mul_px; p.res1 = d->res1; p.res2 = d->res2; p.res2_scale = d->res2_scale;
    p.out = d->out; p.out_plane = d->out_plane; p.out_row = d->out_row;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool wide = conv_cot(d->cout) == 64;
    if (d->ksize == 3) return wide ? launch_conv<3, 8, 2>(p, st) : launch_conv<3, 8, 1>(p, st);
    return wide ? launch_conv<1, 32, 2>(p, st) : launch_conv<1, 32, 1>(p, st);
}
