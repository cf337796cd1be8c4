// OSConv side kernels (gfx950): global average pool, scale routing + ScaleAttention MLP,
// gated aggregation of the 8 kernel banks into one packed conv weight, RCAN SE gate.
// HBM/L2-bound byte work -- no MFMA here on purpose.
#include "common.hpp"

namespace savsr {


struct MeanParams {
    const float* src[SAVSR_MAX_SRC];
    long long plane[SAVSR_MAX_SRC];
    int row[SAVSR_MAX_SRC];
    int src_ch, h, w;
    float* mean;
};

// one block per channel; AdaptiveAvgPool2d(1) (savsr_arch.py:129,146,515)
__global__ __launch_bounds__(256) void channel_mean_kernel(const MeanParams p) {
    const int c = blockIdx.x;
    const int s = c / p.src_ch;
    const int lc = c - s * p.src_ch;
    const float* base = p.src[0];
    long long pl = p.plane[0];
    int rw = p.row[0];
    if (s == 1) { base = p.src[1]; pl = p.plane[1]; rw = p.row[1]; }
    if (s == 2) { base = p.src[2]; pl = p.plane[2]; rw = p.row[2]; }
    if (s == 3) { base = p.src[3]; pl = p.plane[3]; rw = p.row[3]; }
    if (s == 4) { base = p.src[4]; pl = p.plane[4]; rw = p.row[4]; }
    base += (long long)lc * pl;
    float acc = 0.f;
    const int n = p.h * p.w;
    if (rw == p.w && ((reinterpret_cast<uintptr_t>(base) & 15) == 0) && (n & 3) == 0) {
        const f32x4* b4 = reinterpret_cast<const f32x4*>(base);
        for (int i = threadIdx.x; i < n / 4; i += 256) {
            const f32x4 v = b4[i];
            acc += (v[0] + v[1]) + (v[2] + v[3]);
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 256) {
            const int y = i / p.w, x = i - y * p.w;
            acc += base[(long long)y * rw + x];
        }
    }
    __shared__ float part[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p.mean[c] = ((part[0] + part[1]) + (part[2] + part[3])) / (float)n;
}

// y[r] = act(W[r][:] . v + b[r]) for r in [0, rows): one wave per row, lanes across columns.
__device__ __forceinline__ void block_matvec(const float* __restrict__ W, const float* __restrict__ b,
                                             const float* v, float* y, int rows, int cols, bool relu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int r = wave; r < rows; r += nw) {
        const float* wr = W + (long long)r * cols;
        float acc = 0.f;
        for (int c = lane; c < cols; c += 64) acc += wr[c] * v[c];
        acc = wave_sum(acc);
        if (lane == 0) {
            acc += b ? b[r] : 0.f;
            y[r] = relu ? fmaxf(acc, 0.f) : acc;
        }
    }
}

// savsr_arch.py:143-151 (scale routing) + :91-96 / :69-89 (ScaleAttention); single block.
__global__ __launch_bounds__(1024) void osconv_attention_kernel(const savsr_osconv_attn_desc d) {
    extern __shared__ float sm[];
    float* v0 = sm;                       // [cin + 2]
    float* v1 = v0 + d.cin + 2;           // [2 cin]
    float* v2 = v1 + 2 * d.cin;           // [cin]
    float* a = v2 + d.cin;                // [A]
    float* kl = a + d.hidden;             // [knum] kernel logits
    const int tid = threadIdx.x;
    if (tid == 0) { v0[0] = d.inv_sh; v0[1] = d.inv_sw; }
    for (int i = tid; i < d.cin; i += blockDim.x) v0[2 + i] = d.mean[i];
    __syncthreads();
    block_matvec(d.l1_w, d.l1_b, v0, v1, 2 * d.cin, d.cin + 2, true);
    __syncthreads();
    block_matvec(d.l2_w, d.l2_b, v1, v2, d.cin, 2 * d.cin, true);
    __syncthreads();
    block_matvec(d.fc_w, nullptr, v2, a, d.hidden, d.cin, false);
    __syncthreads();
    for (int i = tid; i < d.hidden; i += blockDim.x) a[i] = fmaxf(a[i] * d.bn_scale[i] + d.bn_shift[i], 0.f);
    __syncthreads();
    // heads: hidden is 16..20 wide -> one thread per output
    float* att = d.att;
    for (int i = tid; i < d.cin + d.cout + 9 + d.knum; i += blockDim.x) {
        const float* wr;
        float bias;
        int j = i;
        if (j < d.cin) { wr = d.ch_w + (long long)j * d.hidden; bias = d.ch_b[j]; }
        else if ((j -= d.cin) < d.cout) { wr = d.fl_w + (long long)j * d.hidden; bias = d.fl_b[j]; }
        else if ((j -= d.cout) < 9) { wr = d.sp_w + (long long)j * d.hidden; bias = d.sp_b[j]; }
        else { j -= 9; wr = d.kn_w + (long long)j * d.hidden; bias = d.kn_b[j]; }
        float acc = 0.f;
        for (int k = 0; k < d.hidden; ++k) acc += wr[k] * a[k];
        acc += bias;
        if (i < d.cin + d.cout + 9) att[i] = sigmoidf_(acc);
        else kl[j] = acc;
    }
    __syncthreads();
    if (tid == 0) {                        // softmax over the knum kernels (temperature 1, :88)
        float m = kl[0];
        for (int k = 1; k < d.knum; ++k) m = fmaxf(m, kl[k]);
        float s = 0.f;
        for (int k = 0; k < d.knum; ++k) { kl[k] = expf(kl[k] - m); s += kl[k]; }
        for (int k = 0; k < d.knum; ++k) att[d.cin + d.cout + 9 + k] = kl[k] / s;
    }
}

// Packed-layout elementwise aggregation (savsr_arch.py:158-163 with the channel/filter gates of
// :156,:171 folded into the weight, as :148-149 notes is equivalent).
__global__ __launch_bounds__(256) void osconv_aggregate_kernel(const float* __restrict__ bank, const float* __restrict__ att,
                                                               int cin, int cout, int knum, long long n4, long long bank_stride4,
                                                               int cot, int nchunk, float* __restrict__ out) {
    const long long i4 = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= n4) return;
    constexpr int CK = 8;
    long long base = i4 * 4;
    const int col = (int)(base % cot); base /= cot;
    const int hh = (int)(base & 1); base >>= 1;
    const int cp = (int)(base % (CK / 2)); base /= (CK / 2);
    const int tap = (int)(base % 9); base /= 9;
    const int chunk = (int)(base % nchunk);
    const int cob = (int)(base / nchunk);
    const int ci = chunk * CK + 2 * cp + hh;
    const int co = cob * cot + col;
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (ci < cin) {
        const float* ca = att, *fa = att + cin, *sa = att + cin + cout, *ka = att + cin + cout + 9;
        const f32x4* b4 = reinterpret_cast<const f32x4*>(bank);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < knum; ++k) {
            const f32x4 w = b4[(long long)k * bank_stride4 + i4];
            const float kk = ka[k];
            s[0] += kk * w[0]; s[1] += kk * w[1]; s[2] += kk * w[2]; s[3] += kk * w[3];
        }
        const float g = ca[ci] * sa[tap];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (co + j < cout) ? s[j] * g * fa[co + j] : 0.f;
    }
    reinterpret_cast<f32x4*>(out)[i4] = r;
}

// RCAN ChannelAttention MLP (savsr_arch.py:514-520), one block of 64 threads.
__global__ __launch_bounds__(64) void se_gate_kernel(const float* mean, const float* w1, const float* b1, const float* w2,
                                                     const float* b2, int c, int cmid, float* gate) {
    __shared__ float z[64];
    const int t = threadIdx.x;
    if (t < cmid) {
        float acc = b1[t];
        for (int i = 0; i < c; ++i) acc += w1[t * c + i] * mean[i];
        z[t] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int o = t; o < c; o += 64) {
        float acc = b2[o];
        for (int k = 0; k < cmid; ++k) acc += w2[o * cmid + k] * z[k];
        gate[o] = sigmoidf_(acc);
    }
}

// out = r * gate[c] + x  (savsr_arch.py:524,548-549)
__global__ __launch_bounds__(256) void scale_residual_kernel(const float* __restrict__ r, const float* __restrict__ gate,
                                                             const float* __restrict__ x, float* __restrict__ out,
                                                             long long n, long long n4_per_c) {
    const int c = blockIdx.y;
    const float g = gate[c];
    const f32x4* r4 = reinterpret_cast<const f32x4*>(r + (long long)c * n);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x + (long long)c * n);
    f32x4* o4 = reinterpret_cast<f32x4*>(out + (long long)c * n);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4_per_c; i += (long long)gridDim.x * 256) {
        const f32x4 a = r4[i], b = x4[i];
        f32x4 o;
        o[0] = a[0] * g + b[0]; o[1] = a[1] * g + b[1]; o[2] = a[2] * g + b[2]; o[3] = a[3] * g + b[3];
        o4[i] = o;
    }
    // scalar tail when n is not a multiple of 4
    const long long tail0 = n4_per_c * 4;
    if (blockIdx.x == 0 && threadIdx.x < (n - tail0)) {
        const long long i = (long long)c * n + tail0 + threadIdx.x;
        out[i] = r[i] * g + x[i];
    }
}

}  // namespace savsr

using namespace savsr;

extern "C" int savsr_channel_mean(const float* const* src, const int64_t* src_plane, const int32_t* src_row, int nsrc,
                                  int src_ch, int h, int w, float* mean, void* stream) {
    if (!src || !src_plane || !src_row || !mean) return fail_arg("channel_mean: null pointer");
    if (nsrc < 1 || nsrc > SAVSR_MAX_SRC || src_ch < 1 || h < 1 || w < 1) return fail_arg("channel_mean: shape");
    MeanParams p;
    for (int i = 0; i < SAVSR_MAX_SRC; ++i) {
        const bool on = i < nsrc;
        if (on && !src[i]) return fail_arg("channel_mean: null source");
        p.src[i] = on ? src[i] : nullptr;
        p.plane[i] = on ? src_plane[i] : 0;
        p.row[i] = on ? src_row[i] : 0;
    }
    p.src_ch = src_ch; p.h = h; p.w = w; p.mean = mean;
    hipLaunchKernelGGL(channel_mean_kernel, dim3(nsrc * src_ch), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return check_launch("channel_mean_kernel");
}

extern "C" int savsr_osconv_attention(const savsr_osconv_attn_desc* d, void* stream) {
    if (!d) return fail_arg("osconv_attention: null descriptor");
    if (d->cin < 1 || d->cout < 1 || d->hidden < 1 || d->knum < 1 || d->knum > 64) return fail_arg("osconv_attention: shape");
    if (!d->mean || !d->l1_w || !d->l1_b || !d->l2_w || !d->l2_b || !d->fc_w || !d->bn_scale || !d->bn_shift || !d->ch_w ||
        !d->ch_b || !d->fl_w || !d->fl_b || !d->sp_w || !d->sp_b || !d->kn_w || !d->kn_b || !d->att)
        return fail_arg("osconv_attention: null pointer");
    const size_t lds = sizeof(float) * ((size_t)d->cin + 2 + 2 * d->cin + d->cin + d->hidden + d->knum);
    if (lds > 60000) return fail_arg("osconv_attention: cin too large");
    hipLaunchKernelGGL(osconv_attention_kernel, dim3(1), dim3(1024), lds, static_cast<hipStream_t>(stream), *d);
    return check_launch("osconv_attention_kernel");
}

extern "C" int savsr_osconv_aggregate(const float* bank_packed, const float* att, int cin, int cout, int knum,
                                      float* wpacked_out, void* stream) {
    if (!bank_packed || !att || !wpacked_out) return fail_arg("osconv_aggregate: null pointer");
    if (cin < 1 || cout < 1 || knum < 1) return fail_arg("osconv_aggregate: shape");
    const int ck = conv_ck(3), cot = conv_cot(cout);
    const int nchunk = (cin + ck - 1) / ck;
    const long long total = savsr_conv_packed_floats(cout, cin, 3);
    const long long n4 = total / 4;
    hipLaunchKernelGGL(osconv_aggregate_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       bank_packed, att, cin, cout, knum, n4, n4, cot, nchunk, wpacked_out);
    return check_launch("osconv_aggregate_kernel");
}

extern "C" int savsr_se_gate(const float* mean, const float* w1, const float* b1, const float* w2, const float* b2, int c,
                             int cmid, float* gate, void* stream) {
    if (!mean || !w1 || !b1 || !w2 || !b2 || !gate) return fail_arg("se_gate: null pointer");
    if (c < 1 || cmid < 1 || cmid > 64) return fail_arg("se_gate: shape");
    hipLaunchKernelGGL(se_gate_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), mean, w1, b1, w2, b2, c, cmid, gate);
    return check_launch("se_gate_kernel");
}

extern "C" int savsr_scale_residual(const float* r, const float* gate, const float* x, float* out, int c, int64_t n, void* stream) {
    if (!r || !gate || !x || !out) return fail_arg("scale_residual: null pointer");
    if (c < 1 || n < 1) return fail_arg("scale_residual: shape");
    // vector path needs every channel plane 16-byte aligned
    const bool vec = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const long long n4 = vec ? n / 4 : 0;
    if (!vec && n >= 256) return fail_arg("scale_residual: planes must be 16-byte aligned with n % 4 == 0 (or n < 256)");
    int gx = (int)((n4 + 255) / 256);
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(scale_residual_kernel, dim3(gx, c), dim3(256), 0, static_cast<hipStream_t>(stream), r, gate, x, out, (long long)n, n4);
    return check_launch("scale_residual_kernel");
}
