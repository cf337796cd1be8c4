// OSConv side kernels on channel-last feature maps (gfx950): per-channel partial sums for the
// global average pools, the scale-routing MLP + ScaleAttention spread over many workgroups, the
// gated aggregation of the 8 kernel banks straight into the conv's split-bf16 weight image, and
// the RCAN squeeze-excite gate.  HBM/L2-bound byte work -- no MFMA here on purpose.
#include "common.hpp"

namespace savsr {

// ------------------------------------------------------------------------------------------
// partial[blk][s * src_ch + c] = sum over the block's pixels of src_s[px][c]
// (AdaptiveAvgPool2d(1), savsr_arch.py:129,146,515; the consumer sums the partials in block
//  order and divides by h*w, so the result is deterministic.)
// ------------------------------------------------------------------------------------------
struct SumParams {
    const float* src[SAVSR_MAX_SRC];
    int pix[SAVSR_MAX_SRC];
    int src_ch, npx, nblk;
    float* partial;
};

__global__ __launch_bounds__(256) void channel_sums_kernel(const SumParams p) {
    __shared__ f32x4 red[256];
    const int s = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const float* base = p.src[0];
    int pix = p.pix[0];
    if (s == 1) { base = p.src[1]; pix = p.pix[1]; }
    if (s == 2) { base = p.src[2]; pix = p.pix[2]; }
    if (s == 3) { base = p.src[3]; pix = p.pix[3]; }
    if (s == 4) { base = p.src[4]; pix = p.pix[4]; }
    const int G = p.src_ch / 4;              // float4 groups per pixel (divides 256)
    const int cg = tid % G, pl = tid / G, PL = 256 / G;
    const int per = (p.npx + p.nblk - 1) / p.nblk;
    const int p0 = blk * per, p1 = min(p.npx, p0 + per);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int q = p0 + pl; q < p1; q += PL) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)q * pix + 4 * cg);
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < G) {
        f32x4 t = red[tid];
        for (int k = 1; k < PL; ++k) {
            const f32x4 v = red[tid + k * G];
            t[0] += v[0]; t[1] += v[1]; t[2] += v[2]; t[3] += v[3];
        }
        const int ctot = gridDim.y * p.src_ch;
        *reinterpret_cast<f32x4*>(p.partial + (long long)blk * ctot + s * p.src_ch + 4 * tid) = t;
    }
}

__device__ __forceinline__ float wave_dot(const float* __restrict__ w, const float* v, int n, int lane) {
    float acc = 0.f;
    for (int c = lane; c < n; c += 64) acc += w[c] * v[c];
    return wave_sum(acc);
}

// Up to OSC_MAX_BATCH OSConvs of identical geometry per launch (blockIdx.y picks one): the two propagation
// directions' OSConvs are independent, and these kernels are launch/latency-bound (a few dozen workgroups each).
#ifndef SAVSR_OSC_MAX_BATCH
#define SAVSR_OSC_MAX_BATCH 8
#endif
constexpr int OSC_MAX_BATCH = SAVSR_OSC_MAX_BATCH;      // (2 OSConvs of a block pair x up to 4 clips of a batched launch sequence)
struct OscBatch { savsr_osconv_attn_desc d[OSC_MAX_BATCH]; };
constexpr int OSC_PARTS = 32;      // interleaved row slices of the pooled-sum reduction

// scale routing layer 1 (savsr_arch.py:123-125,143-146): v1 = ReLU(L1 [1/sh, 1/sw, mean] + c1)
// A workgroup computes OSC_L1_ROWS rows of the layer; every workgroup first rebuilds the pooled mean from the producer's
// per-tile partial sums (177 KB at cin = 192), so few, fat workgroups and 16-B loads: with 8 rows per workgroup and
// scalar loads that redundant reduction was most of this kernel's 22 us (now 16).  Tried: the whole routing + attention
// chain in ONE workgroup per OSConv -- 41-97 us: one HBM/L2 latency per dependent stage with a single CU's worth of
// loads in flight; the multi-workgroup split below is faster.
constexpr int OSC_L1_ROWS = 32;
__global__ __launch_bounds__(512) void osconv_l1_kernel(const OscBatch bt) {
    const savsr_osconv_attn_desc& d = bt.d[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) float v0s[];     // [4 pad + cin] (v0 = [1/sh, 1/sw, mean] starts at +2) then scratch [OSC_PARTS][cin]
    float* v0 = v0s + 2;
    float* scr = v0s + 4 + d.cin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { v0[0] = d.inv_sh; v0[1] = d.inv_sw; }
    // this wave's weight rows do not depend on the pooled mean: their loads go out first and land under the reduction
    // (cin <= 320, checked by the launcher: a row has <= 322 = 6 x 64 columns)
    constexpr int RPW = OSC_L1_ROWS / 8, MAXJ = 6;
    const int r0 = blockIdx.x * OSC_L1_ROWS + wave * RPW;
    float wv[RPW][MAXJ];
#pragma unroll
    for (int j = 0; j < RPW; ++j)
#pragma unroll
        for (int q = 0; q < MAXJ; ++q) {
            const int c = lane + 64 * q;
            wv[j][q] = (r0 + j < 2 * d.cin && c < d.cin + 2) ? d.l1_w[(long long)(r0 + j) * (d.cin + 2) + c] : 0.f;
        }
    // pooled mean from the block-ordered partial sums: OSC_PARTS interleaved row slices per channel quad (short, fully
    // unrolled chains of independent 16-B loads), then a fixed-order add over the slices (deterministic)
    const int c4n = d.cin / 4;
    const f32x4* part4 = reinterpret_cast<const f32x4*>(d.partial);
    for (int i = tid; i < c4n * OSC_PARTS; i += 512) {
        const int c4 = i % c4n, part = i / c4n;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int b = part; b < d.nblk; b += OSC_PARTS) {
            const f32x4 v = part4[(long long)b * c4n + c4];
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
        *reinterpret_cast<f32x4*>(scr + part * d.cin + 4 * c4) = s;
    }
    __syncthreads();
    for (int c = tid; c < d.cin; c += 512) {
        float s = 0.f;
#pragma unroll
        for (int part = 0; part < OSC_PARTS; ++part) s += scr[part * d.cin + c];
        v0[2 + c] = s * d.inv_n;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int r = r0 + j;
        float acc = 0.f;                              // per lane in column order, then across the wave: the order of wave_dot
#pragma unroll
        for (int q = 0; q < MAXJ; ++q) {
            const int c = lane + 64 * q;
            if (c < d.cin + 2) acc += wv[j][q] * v0[c];
        }
        acc = wave_sum(acc);
        if (lane == 0 && r < 2 * d.cin) d.v1[r] = fmaxf(acc + d.l1_b[r], 0.f);
    }
}

// scale routing layer 2 (savsr_arch.py:126-127): v2 = ReLU(L2 v1 + c2)
__global__ __launch_bounds__(512) void osconv_l2_kernel(const OscBatch bt) {
    const savsr_osconv_attn_desc& d = bt.d[blockIdx.y];
    extern __shared__ float v1[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this wave's weight row first (it does not depend on v1; 2 cin <= 640 = 10 x 64 columns, checked by the launcher)
    const int r = blockIdx.x * 8 + wave;
    float wv[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        const int c = lane + 64 * q;
        wv[q] = (r < d.cin && c < 2 * d.cin) ? d.l2_w[(long long)r * (2 * d.cin) + c] : 0.f;
    }
    for (int i = tid; i < 2 * d.cin; i += 512) v1[i] = d.v1[i];
    __syncthreads();
    if (r >= d.cin) return;
    float acc = 0.f;                                  // per lane in column order, then across the wave: the order of wave_dot
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        const int c = lane + 64 * q;
        if (c < 2 * d.cin) acc += wv[q] * v1[c];
    }
    acc = wave_sum(acc);
    if (lane == 0) d.v2[r] = fmaxf(acc + d.l2_b[r], 0.f);
}

// Scale routing inside ONE workgroup (savsr_osconv_attn_desc.fused): pooled mean -> layer 1 (all 2 cin rows) -> layer 2 (all cin rows) -> v2 in LDS, with
// exactly the summation orders of osconv_l1_kernel / osconv_l2_kernel (bit-identical v2).  Every aggregation workgroup repeats it (~0.8 MB of
// L2-resident reads per workgroup at cin = 192) instead of waiting for two more launches: the chain was three latency-bound launches of a few dozen
// workgroups each (12.6 + 5.1 + 15 us per OSConv set, 21 sets per frame); no inter-workgroup hand-off, so nothing to synchronise.
// scr: [4 + cin | OSC_PARTS x cin | 2 cin] floats of LDS scratch (v0 with its two scale entries at +2, the partial-sum slices, v1).
__device__ __forceinline__ void osconv_route_in_workgroup(const savsr_osconv_attn_desc& d, float* scr, float* v2) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* v0 = scr + 2;
    float* part = scr + 4 + d.cin;
    float* v1 = part + OSC_PARTS * d.cin;
    if (tid == 0) { v0[0] = d.inv_sh; v0[1] = d.inv_sw; }
    const int c4n = d.cin / 4;
    const f32x4* part4 = reinterpret_cast<const f32x4*>(d.partial);
    for (int i = tid; i < c4n * OSC_PARTS; i += 512) {
        const int c4 = i % c4n, pt = i / c4n;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int b = pt; b < d.nblk; b += OSC_PARTS) {
            const f32x4 v = part4[(long long)b * c4n + c4];
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
        *reinterpret_cast<f32x4*>(part + pt * d.cin + 4 * c4) = s;
    }
    __syncthreads();
    for (int c = tid; c < d.cin; c += 512) {
        float s = 0.f;
#pragma unroll
        for (int pt = 0; pt < OSC_PARTS; ++pt) s += part[pt * d.cin + c];
        v0[2 + c] = s * d.inv_n;
    }
    __syncthreads();
    constexpr int MAXJ = 6;                               // cin + 2 <= 322 columns (checked by the launcher)
    for (int r0 = wave; r0 < 2 * d.cin; r0 += 32) {       // four rows per wave and pass: their loads go out together
        float wv[4][MAXJ];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < MAXJ; ++q) {
                const int r = r0 + 8 * j, c = lane + 64 * q;
                wv[j][q] = (r < 2 * d.cin && c < d.cin + 2) ? d.l1_w[(long long)r * (d.cin + 2) + c] : 0.f;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + 8 * j;
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < MAXJ; ++q) {
                const int c = lane + 64 * q;
                if (c < d.cin + 2) acc += wv[j][q] * v0[c];
            }
            acc = wave_sum(acc);
            if (lane == 0 && r < 2 * d.cin) v1[r] = fmaxf(acc + d.l1_b[r], 0.f);
        }
    }
    __syncthreads();
    for (int r0 = wave; r0 < d.cin; r0 += 16) {           // two rows per wave and pass (2 cin <= 640 = 10 x 64 columns)
        float wv[2][10];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const int r = r0 + 8 * j, c = lane + 64 * q;
                wv[j][q] = (r < d.cin && c < 2 * d.cin) ? d.l2_w[(long long)r * (2 * d.cin) + c] : 0.f;
            }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = r0 + 8 * j;
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const int c = lane + 64 * q;
                if (c < 2 * d.cin) acc += wv[j][q] * v1[c];
            }
            acc = wave_sum(acc);
            if (lane == 0 && r < d.cin) v2[r] = fmaxf(acc + d.l2_b[r], 0.f);
        }
    }
    __syncthreads();
}

// ScaleAttention heads (savsr_arch.py:91-96, 69-89) recomputed per workgroup (a few k MACs),
// then  W''[co][ci][tap] = fa[co] ca[ci] sa[tap] sum_k ka[k] W[k][co][ci][tap]  (:156-163,171
// folded, :148-149) for this workgroup's slice, split to (hi, lo) bf16 and written in the conv
// weight-image order.
__global__ __launch_bounds__(512) void osconv_aggregate_kernel(const OscBatch bt) {
    const savsr_osconv_attn_desc& d = bt.d[blockIdx.y];
    extern __shared__ float sm[];
    float* v2 = sm;                        // [cin]
    float* a = v2 + d.cin;                 // [hidden]
    float* gates = a + d.hidden;           // [cin + cout + 9 + knum]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this thread's slice of the 8 banks does not depend on the gates: its loads go out first and land under the
    // attention heads below (knum <= OSC_MAX_KNUM banks are preloaded; more fall back to loading in the loop)
    constexpr int OSC_MAX_KNUM = 8;
    const long long unit = (long long)blockIdx.x * 512 + tid;
    const bool live = unit < d.nunits;
    f32x4 bw[OSC_MAX_KNUM][2];
#pragma unroll
    for (int k = 0; k < OSC_MAX_KNUM; ++k) {
        bw[k][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        bw[k][1] = bw[k][0];
        if (live && k < d.knum) {
            const f32x4* b = reinterpret_cast<const f32x4*>(d.bank + ((long long)k * d.nunits + unit) * 8);
            bw[k][0] = b[0];
            bw[k][1] = b[1];
        }
    }
    // ... and so do the head weights of this wave / thread (hidden <= 32 and cin <= 320, checked by the launcher):
    //   fc rows wave, wave + 8, .. (<= 4 rows x 5 column slices) and ONE gate row per thread (ngate <= 512 covers
    //   cin + cout + 9 + knum of every OSConv of the network; larger ones loop below)
    constexpr int FC_ROWS = 4, FC_J = 5, GH = 32;
    float fw[FC_ROWS][FC_J];
#pragma unroll
    for (int j = 0; j < FC_ROWS; ++j)
#pragma unroll
        for (int q = 0; q < FC_J; ++q) {
            const int r = wave + 8 * j, c = lane + 64 * q;
            fw[j][q] = (r < d.hidden && c < d.cin) ? d.fc_w[(long long)r * d.cin + c] : 0.f;
        }
    const int ngate = d.cin + d.cout + 9 + d.knum;
    auto gate_row = [&](int i, float& bias) -> const float* {
        int j = i;
        if (j < d.cin) { bias = d.ch_b[j]; return d.ch_w + (long long)j * d.hidden; }
        if ((j -= d.cin) < d.cout) { bias = d.fl_b[j]; return d.fl_w + (long long)j * d.hidden; }
        if ((j -= d.cout) < 9) { bias = d.sp_b[j]; return d.sp_w + (long long)j * d.hidden; }
        j -= 9;
        bias = d.kn_b[j];
        return d.kn_w + (long long)j * d.hidden;
    };
    float gw[GH], gbias = 0.f;
    {
        const float* wr = tid < ngate ? gate_row(tid, gbias) : nullptr;
#pragma unroll
        for (int k = 0; k < GH; ++k) gw[k] = (wr && k < d.hidden) ? wr[k] : 0.f;
    }
    if (d.fused) {
        osconv_route_in_workgroup(d, gates + (d.cin + d.cout + 9 + d.knum), v2);     // (LDS scratch behind the gates; ends with a barrier)
        if (blockIdx.x == 0 && d.v2) for (int i = tid; i < d.cin; i += 512) d.v2[i] = v2[i];      // (kept observable for tests / tools)
    } else {
        for (int i = tid; i < d.cin; i += 512) v2[i] = d.v2[i];
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < FC_ROWS; ++j) {
        const int r = wave + 8 * j;
        float acc = 0.f;                              // per lane in column order, then across the wave: the order of wave_dot
#pragma unroll
        for (int q = 0; q < FC_J; ++q) {
            const int c = lane + 64 * q;
            if (c < d.cin) acc += fw[j][q] * v2[c];
        }
        acc = wave_sum(acc);
        if (lane == 0 && r < d.hidden) a[r] = fmaxf(acc * d.bn_scale[r] + d.bn_shift[r], 0.f);
    }
    __syncthreads();
    if (tid < ngate) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < GH; ++k)
            if (k < d.hidden) acc += gw[k] * a[k];
        acc += gbias;
        gates[tid] = (tid < d.cin + d.cout + 9) ? sigmoidf_(acc) : acc;   // kernel logits stay raw here
    }
    for (int i = tid + 512; i < ngate; i += 512) {
        float bias;
        const float* wr = gate_row(i, bias);
        float acc = 0.f;
        for (int k = 0; k < d.hidden; ++k) acc += wr[k] * a[k];
        acc += bias;
        gates[i] = (i < d.cin + d.cout + 9) ? sigmoidf_(acc) : acc;
    }
    __syncthreads();
    float* ka = gates + d.cin + d.cout + 9;
    if (tid == 0) {                        // softmax over the kernels, temperature 1 (:88)
        float m = ka[0];
        for (int k = 1; k < d.knum; ++k) m = fmaxf(m, ka[k]);
        float s = 0.f;
        for (int k = 0; k < d.knum; ++k) { ka[k] = expf(ka[k] - m); s += ka[k]; }
        for (int k = 0; k < d.knum; ++k) ka[k] = ka[k] / s;
    }
    __syncthreads();
    if (blockIdx.x == 0 && d.att)
        for (int i = tid; i < ngate; i += 512) d.att[i] = gates[i];

    // ---- this workgroup's slice of the weight image: one 8-element lane unit per thread ----
    if (!live) return;
    const int cot = conv_cot(d.cout), nt = cot / 32, nchunk = d.cin / 16;
    const long long group = unit >> 6;            // (cob, chunk, tap, t); KSTEPS == 1 for 3x3
    const int ln = (int)(unit & 63), row = ln & 31, kh = ln >> 5;
    const int t = (int)(group % nt);
    long long g2 = group / nt;
    const int tap = (int)(g2 % 9); g2 /= 9;
    const int chunk = (int)(g2 % nchunk);
    const int cob = (int)(g2 / nchunk);
    const int co = cob * cot + 32 * t + row;
    const int ci0 = chunk * 16 + kh * 8;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
    for (int k = 0; k < OSC_MAX_KNUM; ++k)
        if (k < d.knum) {
            const f32x4 w0 = bw[k][0], w1 = bw[k][1];
            const float kk = ka[k];
            s0[0] += kk * w0[0]; s0[1] += kk * w0[1]; s0[2] += kk * w0[2]; s0[3] += kk * w0[3];
            s1[0] += kk * w1[0]; s1[1] += kk * w1[1]; s1[2] += kk * w1[2]; s1[3] += kk * w1[3];
        }
    for (int k = OSC_MAX_KNUM; k < d.knum; ++k) {
        const f32x4* b = reinterpret_cast<const f32x4*>(d.bank + ((long long)k * d.nunits + unit) * 8);
        const f32x4 w0 = b[0], w1 = b[1];
        const float kk = ka[k];
        s0[0] += kk * w0[0]; s0[1] += kk * w0[1]; s0[2] += kk * w0[2]; s0[3] += kk * w0[3];
        s1[0] += kk * w1[0]; s1[1] += kk * w1[1]; s1[2] += kk * w1[2]; s1[3] += kk * w1[3];
    }
    const float gco = (co < d.cout) ? gates[d.cin + co] * gates[d.cin + d.cout + tap] : 0.f;
    const float* ca = gates + ci0;
    s0[0] *= gco * ca[0]; s0[1] *= gco * ca[1]; s0[2] *= gco * ca[2]; s0[3] *= gco * ca[3];
    s1[0] *= gco * ca[4]; s1[1] *= gco * ca[5]; s1[2] *= gco * ca[6]; s1[3] *= gco * ca[7];
    bf16x8 hi, lo;
    const float x[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        hi[j] = h;
        lo[j] = (__bf16)(x[j] - (float)h);
    }
    bf16x8* img = reinterpret_cast<bf16x8*>(d.wimg_out);
    img[group * 128 + ln] = hi;
    img[group * 128 + 64 + ln] = lo;
}

// The same weight generation for a conv that runs in the Winograd F(2,3)-along-y form (conv_wy.hip, SAVSR_CONV_WINOGRAD_Y): per (co, ci, kx) the three
// aggregated and gated taps g_ky = fa[co] ca[ci] sa[ky, kx] sum_k ka[k] W[k][co][ci][ky][kx] -- the spatial gate sa is applied BEFORE the transform, it
// depends on ky -- then U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2, split and written in the order of
// savsr_conv_wy_pack_index.  One thread = one float4 half of an 8-element lane unit of (cob, chunk, kx, t): 3 x knum x 16 B of bank data per thread.
__global__ __launch_bounds__(512) void osconv_aggregate_wy_kernel(const OscBatch bt) {
    const savsr_osconv_attn_desc& d = bt.d[blockIdx.y];
    extern __shared__ float sm[];
    float* v2 = sm;                        // [cin]
    float* a = v2 + d.cin;                 // [hidden]
    float* gates = a + d.hidden;           // [cin + cout + 9 + knum]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int OSC_MAX_KNUM = 8;
    const int nt = 2, nchunk = d.cin / 16;                       // cout % 64 == 0 (checked by the launcher)
    const long long item = (long long)blockIdx.x * 512 + tid;   // ((((cob * nchunk + chunk) * 3 + kx) * nt + t) * 64 + ln) * 2 + h4
    const long long nitems = (long long)(d.cout / 64) * nchunk * 3 * nt * 64 * 2;
    const bool live = item < nitems;
    const int h4 = (int)(item & 1);
    const long long u = item >> 1;
    const int ln = (int)(u & 63);
    long long g2 = u >> 6;
    const int t = (int)(g2 % nt); g2 /= nt;
    const int kx = (int)(g2 % 3); g2 /= 3;
    const int chunk = (int)(g2 % nchunk);
    const int cob = (int)(g2 / nchunk);
    // bank loads first: they do not depend on the gates and land under the attention heads below
    f32x4 bw[3][OSC_MAX_KNUM];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const long long unit = ((((long long)(cob * nchunk + chunk) * 9 + (ky * 3 + kx)) * nt + t) << 6) + ln;     // the direct image's unit of this tap
#pragma unroll
        for (int k = 0; k < OSC_MAX_KNUM; ++k) {
            bw[ky][k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (live && k < d.knum) bw[ky][k] = reinterpret_cast<const f32x4*>(d.bank + ((long long)k * d.nunits + unit) * 8)[h4];
        }
    }
    // ... and so do the head weights of this wave / thread (hidden <= 32 and cin <= 320, checked by the launcher):
    //   fc rows wave, wave + 8, .. (<= 4 rows x 5 column slices) and ONE gate row per thread (ngate <= 512 covers
    //   cin + cout + 9 + knum of every OSConv of the network; larger ones loop below)
    constexpr int FC_ROWS = 4, FC_J = 5, GH = 32;
    float fw[FC_ROWS][FC_J];
#pragma unroll
    for (int j = 0; j < FC_ROWS; ++j)
#pragma unroll
        for (int q = 0; q < FC_J; ++q) {
            const int r = wave + 8 * j, c = lane + 64 * q;
            fw[j][q] = (r < d.hidden && c < d.cin) ? d.fc_w[(long long)r * d.cin + c] : 0.f;
        }
    const int ngate = d.cin + d.cout + 9 + d.knum;
    auto gate_row = [&](int i, float& bias) -> const float* {
        int j = i;
        if (j < d.cin) { bias = d.ch_b[j]; return d.ch_w + (long long)j * d.hidden; }
        if ((j -= d.cin) < d.cout) { bias = d.fl_b[j]; return d.fl_w + (long long)j * d.hidden; }
        if ((j -= d.cout) < 9) { bias = d.sp_b[j]; return d.sp_w + (long long)j * d.hidden; }
        j -= 9;
        bias = d.kn_b[j];
        return d.kn_w + (long long)j * d.hidden;
    };
    float gw[GH], gbias = 0.f;
    {
        const float* wr = tid < ngate ? gate_row(tid, gbias) : nullptr;
#pragma unroll
        for (int k = 0; k < GH; ++k) gw[k] = (wr && k < d.hidden) ? wr[k] : 0.f;
    }
    if (d.fused) {
        osconv_route_in_workgroup(d, gates + (d.cin + d.cout + 9 + d.knum), v2);     // (LDS scratch behind the gates; ends with a barrier)
        if (blockIdx.x == 0 && d.v2) for (int i = tid; i < d.cin; i += 512) d.v2[i] = v2[i];      // (kept observable for tests / tools)
    } else {
        for (int i = tid; i < d.cin; i += 512) v2[i] = d.v2[i];
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < FC_ROWS; ++j) {
        const int r = wave + 8 * j;
        float acc = 0.f;                              // per lane in column order, then across the wave: the order of wave_dot
#pragma unroll
        for (int q = 0; q < FC_J; ++q) {
            const int c = lane + 64 * q;
            if (c < d.cin) acc += fw[j][q] * v2[c];
        }
        acc = wave_sum(acc);
        if (lane == 0 && r < d.hidden) a[r] = fmaxf(acc * d.bn_scale[r] + d.bn_shift[r], 0.f);
    }
    __syncthreads();
    if (tid < ngate) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < GH; ++k)
            if (k < d.hidden) acc += gw[k] * a[k];
        acc += gbias;
        gates[tid] = (tid < d.cin + d.cout + 9) ? sigmoidf_(acc) : acc;   // kernel logits stay raw here
    }
    for (int i = tid + 512; i < ngate; i += 512) {
        float bias;
        const float* wr = gate_row(i, bias);
        float acc = 0.f;
        for (int k = 0; k < d.hidden; ++k) acc += wr[k] * a[k];
        acc += bias;
        gates[i] = (i < d.cin + d.cout + 9) ? sigmoidf_(acc) : acc;
    }
    __syncthreads();
    float* ka = gates + d.cin + d.cout + 9;
    if (tid == 0) {                        // softmax over the kernels, temperature 1 (:88)
        float m = ka[0];
        for (int k = 1; k < d.knum; ++k) m = fmaxf(m, ka[k]);
        float s = 0.f;
        for (int k = 0; k < d.knum; ++k) { ka[k] = expf(ka[k] - m); s += ka[k]; }
        for (int k = 0; k < d.knum; ++k) ka[k] = ka[k] / s;
    }
    __syncthreads();
    if (blockIdx.x == 0 && d.att)
        for (int i = tid; i < ngate; i += 512) d.att[i] = gates[i];

    if (!live) return;
    const int row = ln & 31, kh = ln >> 5;
    const int co = cob * 64 + 32 * t + row;
    const int ci0 = chunk * 16 + kh * 8 + 4 * h4;
    const float* ka_ = gates + d.cin + d.cout + 9;
    const float* ca = gates + ci0;
    f32x4 g[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < OSC_MAX_KNUM; ++k)
            if (k < d.knum) {
                const float kk = ka_[k];
                s0[0] += kk * bw[ky][k][0]; s0[1] += kk * bw[ky][k][1]; s0[2] += kk * bw[ky][k][2]; s0[3] += kk * bw[ky][k][3];
            }
        for (int k = OSC_MAX_KNUM; k < d.knum; ++k) {
            const long long unit = ((((long long)(cob * nchunk + chunk) * 9 + (ky * 3 + kx)) * nt + t) << 6) + ln;
            const f32x4 w0 = reinterpret_cast<const f32x4*>(d.bank + ((long long)k * d.nunits + unit) * 8)[h4];
            const float kk = ka_[k];
            s0[0] += kk * w0[0]; s0[1] += kk * w0[1]; s0[2] += kk * w0[2]; s0[3] += kk * w0[3];
        }
        const float gco = gates[d.cin + co] * gates[d.cin + d.cout + ky * 3 + kx];      // same product order as the direct kernel: (fa sa) ca
        s0[0] *= gco * ca[0]; s0[1] *= gco * ca[1]; s0[2] *= gco * ca[2]; s0[3] *= gco * ca[3];
        g[ky] = s0;
    }
    f32x4 uu[4];
    uu[0] = g[0];
    uu[3] = g[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float sgl = g[0][j] + g[2][j];
        uu[1][j] = 0.5f * (sgl + g[1][j]);
        uu[2][j] = 0.5f * (sgl - g[1][j]);
    }
    bf16x4* img = reinterpret_cast<bf16x4*>(d.wimg_out);
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
        const int hf = pos >> 1, vr = pos & 1;
        const long long group = ((((long long)(cob * nchunk + chunk) * 2 + hf) * 6 + (vr * 3 + kx)) * nt + t);
        bf16x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const __bf16 hh = (__bf16)uu[pos][j];
            hi[j] = hh;
            lo[j] = (__bf16)(uu[pos][j] - (float)hh);
        }
        img[(group * 128 + ln) * 2 + h4] = hi;                 // units of 16 B = two bf16x4
        img[(group * 128 + 64 + ln) * 2 + h4] = lo;
    }
}

// RCAN ChannelAttention MLP (savsr_arch.py:514-520); one workgroup of 1024 threads (latency-bound: the pooled-sum
// reduction runs as 16 short interleaved row slices per channel).
constexpr int SE_PARTS = 16;
// The gate of one RCAB, computed by a 1024-thread workgroup into LDS (g[c]); every summation order is fixed, so every workgroup
// that evaluates it gets the same bits.
__device__ __forceinline__ void se_gate_block(const float* partial, int nblk, float inv_n, const float* w1, const float* b1,
                                               const float* w2, const float* b2, int c, int cmid, float* g) {
    __shared__ float scr[SE_PARTS * 128];
    __shared__ float m[128];
    __shared__ float z[64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // the MLP weights do not depend on the pooled mean: their loads go out first and land under the reduction
    // (c <= 128, cmid <= 64 checked by the launcher; wave w < cmid owns hidden unit w, thread o < c owns output o)
    float w1v[2] = {0.f, 0.f}, b1v = 0.f;
    if (wave < cmid) {
        if (lane < c) w1v[0] = w1[wave * c + lane];
        if (lane + 64 < c) w1v[1] = w1[wave * c + lane + 64];
        b1v = b1[wave];
    }
    float w2v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b2v = 0.f;
    if (t < c) {
        b2v = b2[t];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < cmid) w2v[k] = w2[t * cmid + k];
    }
    for (int i = t; i < c * SE_PARTS; i += 1024) {
        const int ch = i % c, part = i / c;
        float s = 0.f;
#pragma unroll 8
        for (int b = part; b < nblk; b += SE_PARTS) s += partial[(long long)b * c + ch];
        scr[part * c + ch] = s;
    }
    __syncthreads();
    for (int ch = t; ch < c; ch += 1024) {
        float s = 0.f;
#pragma unroll
        for (int part = 0; part < SE_PARTS; ++part) s += scr[part * c + ch];
        m[ch] = s * inv_n;
    }
    __syncthreads();
    for (int r = wave; r < cmid; r += 16) {
        float acc;
        if (r == wave) acc = (lane < c ? w1v[0] * m[lane] : 0.f) + (lane + 64 < c ? w1v[1] * m[lane + 64] : 0.f);
        else {                                            // cmid > 16 waves: later rows are loaded here
            acc = 0.f;
            for (int i = lane; i < c; i += 64) acc += w1[r * c + i] * m[i];
        }
        acc = wave_sum(acc);
        if (lane == 0) z[r] = fmaxf(acc + (r == wave ? b1v : b1[r]), 0.f);
    }
    __syncthreads();
    if (t < c) {
        float acc = b2v;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < cmid) acc += w2v[k] * z[k];
        for (int k = 8; k < cmid; ++k) acc += w2[t * cmid + k] * z[k];
        g[t] = sigmoidf_(acc);
    }
    __syncthreads();
}


__global__ __launch_bounds__(1024) void se_gate_kernel(const float* partial, int nblk, float inv_n, const float* w1, const float* b1,
                                                     const float* w2, const float* b2, int c, int cmid, float* gate) {
    __shared__ float g[128];
    se_gate_block(partial, nblk, inv_n, w1, b1, w2, b2, c, cmid, g);
    if (threadIdx.x < c) gate[threadIdx.x] = g[threadIdx.x];
}

// ChannelAttention + RCAB residual in ONE launch (savsr_arch.py:514-524,548-549): every workgroup re-evaluates the gate from the
// pooled partial sums (a 59 KB read from L2 and a few thousand MACs per workgroup, one workgroup per CU) and then scales its
// share of the pixels: out[px][c] = r[px][c] * gate[c] + x[px][c].  Replaces the se_gate -> scale_residual pair: one launch
// boundary and one dependent kernel less per RCAB (32 per frame).
// blockIdx.y = clip of a batched launch sequence (ABI 26: savsr_se_scale_residual_batch); the `*_bs` are BYTES from one clip's operand to the next.
__global__ __launch_bounds__(1024) void se_scale_residual_kernel(const float* partial, int nblk, float inv_n, const float* w1, const float* b1,
                                                               const float* w2, const float* b2, int c, int cmid, const f32x4* __restrict__ r,
                                                               const f32x4* __restrict__ x, f32x4* __restrict__ out, long long n4,
                                                               long long part_bs, long long r_bs, long long x_bs, long long out_bs) {
    __shared__ __attribute__((aligned(16))) float g[128];
    {
        const long long cb = blockIdx.y;
        partial = reinterpret_cast<const float*>(reinterpret_cast<const char*>(partial) + cb * part_bs);
        r = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(r) + cb * r_bs);
        x = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(x) + cb * x_bs);
        out = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out) + cb * out_bs);
    }
    // The streaming operands do not depend on the gate: the loads of this thread's first SE_UNROLL elements (all of them at
    // 180x320: 3.5 per thread) go out before the gate is evaluated and land under it -- one memory round trip per thread instead
    // of one per element (the loop used to issue two loads, wait, store: 11.5 us for 44 MB).  The result is read by the next
    // kernel, not by this one: nt stores.
    constexpr int SE_UNROLL = 4;
    const long long i0 = (long long)blockIdx.x * 1024 + threadIdx.x, stride = (long long)gridDim.x * 1024;
    f32x4 av[SE_UNROLL], bv[SE_UNROLL];
#pragma unroll
    for (int u = 0; u < SE_UNROLL; ++u) {
        const long long i = i0 + u * stride;
        av[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        bv[u] = av[u];
        if (i < n4) { av[u] = r[i]; bv[u] = x[i]; }
    }
    se_gate_block(partial, nblk, inv_n, w1, b1, w2, b2, c, cmid, g);
    const int c4 = c >> 2;
    auto apply = [&](long long i, const f32x4& a, const f32x4& b2_) {
        const f32x4 gq = *reinterpret_cast<const f32x4*>(g + 4 * (int)(i % c4));
        f32x4 o;
        o[0] = a[0] * gq[0] + b2_[0]; o[1] = a[1] * gq[1] + b2_[1]; o[2] = a[2] * gq[2] + b2_[2]; o[3] = a[3] * gq[3] + b2_[3];
        __builtin_nontemporal_store(o, out + i);
    };
#pragma unroll
    for (int u = 0; u < SE_UNROLL; ++u) {
        const long long i = i0 + u * stride;
        if (i < n4) apply(i, av[u], bv[u]);
    }
    for (long long i = i0 + SE_UNROLL * stride; i < n4; i += stride) apply(i, r[i], x[i]);
}

// out[px][c] = r[px][c] * gate[c] + x[px][c]   (savsr_arch.py:524,548-549), c == 64 contiguous
__global__ __launch_bounds__(256) void scale_residual_kernel(const f32x4* __restrict__ r, const float* __restrict__ gate,
                                                             const f32x4* __restrict__ x, f32x4* __restrict__ out, long long n4, int c4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gate + 4 * (int)(i % c4));
        const f32x4 a = r[i], b = x[i];
        f32x4 o;
        o[0] = a[0] * g[0] + b[0]; o[1] = a[1] * g[1] + b[1]; o[2] = a[2] * g[2] + b[2]; o[3] = a[3] * g[3] + b[3];
        out[i] = o;
    }
}

}  // namespace savsr

using namespace savsr;

extern "C" int savsr_channel_sums(const float* const* src, const int32_t* src_pix, int nsrc, int src_ch, int64_t npx, int nblk,
                                  float* partial, void* stream) {
    if (!src || !src_pix || !partial) return fail_arg("channel_sums: null pointer");
    if (nsrc < 1 || nsrc > SAVSR_MAX_SRC || src_ch < 4 || (src_ch % 4) || (256 % (src_ch / 4)) || npx < 1 || nblk < 1 || npx > 0x7fffffff)
        return fail_arg("channel_sums: shape (src_ch/4 must divide 256)");
    SumParams p;
    for (int i = 0; i < SAVSR_MAX_SRC; ++i) {
        const bool on = i < nsrc;
        if (on && (!src[i] || (reinterpret_cast<uintptr_t>(src[i]) & 15) || (src_pix[i] & 3))) return fail_arg("channel_sums: null/unaligned source");
        p.src[i] = on ? src[i] : nullptr;
        p.pix[i] = on ? src_pix[i] : 0;
    }
    p.src_ch = src_ch; p.npx = (int)npx; p.nblk = nblk; p.partial = partial;
    hipLaunchKernelGGL(channel_sums_kernel, dim3(nblk, nsrc), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return check_launch("channel_sums_kernel");
}

static int check_osconv_desc(const savsr_osconv_attn_desc* d) {
    if (d->cin < 16 || (d->cin % 16) || d->cin > 320 || d->cout < 1 || d->hidden < 1 || d->hidden > 32 || d->knum < 1 || d->knum > 64 || d->nblk < 1)
        return fail_arg("osconv_weights: shape (cin multiple of 16, <= 320; hidden <= 32)");
    if (!d->partial || !d->l1_w || !d->l1_b || !d->l2_w || !d->l2_b || !d->fc_w || !d->bn_scale || !d->bn_shift || !d->ch_w ||
        !d->ch_b || !d->fl_w || !d->fl_b || !d->sp_w || !d->sp_b || !d->kn_w || !d->kn_b || !d->v1 || !d->v2 || !d->bank || !d->wimg_out)
        return fail_arg("osconv_weights: null pointer");
    if (d->nunits * 8 != savsr_conv_packed_elems(d->cout, d->cin, 3)) return fail_arg("osconv_weights: nunits != packed_elems/8");
    if ((reinterpret_cast<uintptr_t>(d->bank) | reinterpret_cast<uintptr_t>(d->wimg_out) | reinterpret_cast<uintptr_t>(d->partial)) & 15) {
        set_error("osconv_weights: bank / wimg_out / partial must be 16-byte aligned");
        return SAVSR_E_ALIGN;
    }
    return 0;
}

extern "C" int savsr_osconv_weights_max_batch(void) { return OSC_MAX_BATCH; }

extern "C" int savsr_osconv_weights_batch(const savsr_osconv_attn_desc* descs, int n, void* stream) {
    if (!descs) return fail_arg("osconv_weights: null descriptor");
    if (n < 1 || n > OSC_MAX_BATCH) return fail_arg("osconv_weights: batch size must be 1..savsr_osconv_weights_max_batch()");
    OscBatch bt;
    for (int i = 0; i < n; ++i) {
        const int rc = check_osconv_desc(descs + i);
        if (rc) return rc;
        if (descs[i].cin != descs[0].cin || descs[i].cout != descs[0].cout || descs[i].hidden != descs[0].hidden ||
            descs[i].knum != descs[0].knum || descs[i].nunits != descs[0].nunits || (descs[i].wy != 0) != (descs[0].wy != 0) ||
            (descs[i].fused != 0) != (descs[0].fused != 0))
            return fail_arg("osconv_weights: all OSConvs of a batch must share cin / cout / hidden / knum / wy / fused");
        bt.d[i] = descs[i];
    }
    for (int i = n; i < OSC_MAX_BATCH; ++i) bt.d[i] = descs[0];
    const savsr_osconv_attn_desc* d = descs;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = 0;
    if (!d->fused) {
        hipLaunchKernelGGL(osconv_l1_kernel, dim3((2 * d->cin + OSC_L1_ROWS - 1) / OSC_L1_ROWS, n), dim3(512), sizeof(float) * ((OSC_PARTS + 1) * d->cin + 4), st, bt);
        rc = check_launch("osconv_l1_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL(osconv_l2_kernel, dim3((d->cin + 7) / 8, n), dim3(512), sizeof(float) * 2 * d->cin, st, bt);
        rc = check_launch("osconv_l2_kernel");
        if (rc) return rc;
    }
    // (fused: + the routing scratch [4 + cin | OSC_PARTS x cin | 2 cin] behind the gates: 46 KB at cin = 320)
    const size_t lds = sizeof(float) * ((size_t)d->cin + d->hidden + d->cin + d->cout + 9 + d->knum + (d->fused ? 4 + (size_t)(OSC_PARTS + 3) * d->cin : 0));
    if (d->wy) {
        if (d->cout % 64 || d->cin % 16) return fail_arg("osconv_weights: the Winograd-y image needs cout % 64 == 0 and cin % 16 == 0");
        const long long nitems = (long long)(d->cout / 64) * (d->cin / 16) * 3 * 2 * 64 * 2;
        hipLaunchKernelGGL(osconv_aggregate_wy_kernel, dim3((unsigned)((nitems + 511) / 512), n), dim3(512), lds, st, bt);
        return check_launch("osconv_aggregate_wy_kernel");
    }
    hipLaunchKernelGGL(osconv_aggregate_kernel, dim3((unsigned)((d->nunits + 511) / 512), n), dim3(512), lds, st, bt);
    return check_launch("osconv_aggregate_kernel");
}

extern "C" int savsr_osconv_weights(const savsr_osconv_attn_desc* d, void* stream) { return savsr_osconv_weights_batch(d, 1, stream); }

extern "C" int savsr_se_gate(const float* partial, int nblk, float inv_n, const float* w1, const float* b1, const float* w2,
                             const float* b2, int c, int cmid, float* gate, void* stream) {
    if (!partial || !w1 || !b1 || !w2 || !b2 || !gate) return fail_arg("se_gate: null pointer");
    if (c < 1 || c > 128 || cmid < 1 || cmid > 64 || nblk < 1) return fail_arg("se_gate: shape");
    hipLaunchKernelGGL(se_gate_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), partial, nblk, inv_n, w1, b1, w2, b2, c, cmid, gate);
    return check_launch("se_gate_kernel");
}

extern "C" int savsr_se_scale_residual_batch(const float* partial, int nblk, float inv_n, const float* w1, const float* b1, const float* w2,
                                             const float* b2, int c, int cmid, const float* r, const float* x, float* out, int64_t npx, int nclip,
                                             int64_t partial_stride, int64_t r_stride, int64_t x_stride, int64_t out_stride, void* stream) {
    if (!partial || !w1 || !b1 || !w2 || !b2 || !r || !x || !out) return fail_arg("se_scale_residual: null pointer");
    if (nclip < 1 || nclip > 64 || ((partial_stride | r_stride | x_stride | out_stride) & 15) || (nclip > 1 && out_stride == 0))
        return fail_arg("se_scale_residual_batch: 1..64 clips, strides multiples of 16 bytes, distinct outputs");
    if (c < 4 || c > 128 || (c % 4) || cmid < 1 || cmid > 64 || nblk < 1 || npx < 1) return fail_arg("se_scale_residual: shape (c a multiple of 4, <= 128)");
    if ((reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) {
        set_error("se_scale_residual: r / x / out must be 16-byte aligned");
        return SAVSR_E_ALIGN;
    }
    const long long n4 = npx * (c / 4);
    long long g = (n4 + 1023) / 1024;
    if (g > 256) g = 256;                                              // one workgroup per CU: each re-evaluates the gate
    hipLaunchKernelGGL(se_scale_residual_kernel, dim3((unsigned)g, (unsigned)nclip), dim3(1024), 0, static_cast<hipStream_t>(stream), partial, nblk, inv_n, w1, b1, w2, b2,
                       c, cmid, reinterpret_cast<const f32x4*>(r), reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(out), n4,
                       (long long)partial_stride, (long long)r_stride, (long long)x_stride, (long long)out_stride);
    return check_launch("se_scale_residual_kernel");
}

extern "C" int savsr_se_scale_residual(const float* partial, int nblk, float inv_n, const float* w1, const float* b1, const float* w2,
                                       const float* b2, int c, int cmid, const float* r, const float* x, float* out, int64_t npx, void* stream) {
    return savsr_se_scale_residual_batch(partial, nblk, inv_n, w1, b1, w2, b2, c, cmid, r, x, out, npx, 1, 0, 0, 0, 0, stream);
}

extern "C" int savsr_scale_residual(const float* r, const float* gate, const float* x, float* out, int c, int64_t npx, void* stream) {
    if (!r || !gate || !x || !out) return fail_arg("scale_residual: null pointer");
    if (c < 4 || (c % 4) || npx < 1) return fail_arg("scale_residual: c must be a multiple of 4");
    if ((reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(gate)) & 15) {
        set_error("scale_residual: pointers must be 16-byte aligned");
        return SAVSR_E_ALIGN;
    }
    const long long n4 = npx * (c / 4);
    long long g = (n4 + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(scale_residual_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const f32x4*>(r), gate, reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(out), n4, c / 4);
    return check_launch("scale_residual_kernel");
}
