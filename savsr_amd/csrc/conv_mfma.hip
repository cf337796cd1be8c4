// Dense 3x3 / 1x1 convolution on channel-last fp32 feature maps as an implicit GEMM on the
// bf16 matrix cores with split-precision operands ("bf16x3", gfx950).
//
//   D[co][px] = sum_k A[co][k] * B[k][px],   k = (tap, input channel)
//
// Every fp32 operand v is split as v = hi + lo + O(2^-17 |v|) with hi = bf16(v), lo = bf16(v - hi),
// and each product is evaluated as hi*hi + hi*lo + lo*hi with fp32 accumulation on
// v_mfma_f32_32x32x16_bf16 (3 MFMAs at 16x the fp32-MFMA rate = 5.3x the fp32 matrix peak).
// The dropped lo*lo term is O(2^-18); measured end-to-end deviation from the fp32 oracle is
// ~1e-5 max-abs, 1e-5 dB PSNR (DESIGN.md section "Numerics").
//
// Layout.  Activations: [h][w][C] fp32 ("channel-last"): the 8 consecutive k an MFMA lane needs
// are 8 consecutive channels of one pixel = two 16-B loads, and the 4 consecutive output rows
// a lane holds per accumulator quad are 4 consecutive channels = one 16-B store.
// Weights: pre-split, pre-packed bf16 image in exactly the LDS order the lanes read it:
//   [cob][chunk][tap][kstep][t][part(hi,lo)][lane 64][8]  (DMA-able 1-KiB pieces).
//
// Block = 512 threads = 8 waves, tile = 8 rows x 32 cols of pixels x (32 NT) output channels;
// wave w owns row w.  The K loop runs over phases of KC input channels x all taps; the next
// phase's weight slab and input tile (split to hi/lo on the fly) are staged global ->
// registers -> LDS while the current phase's MFMAs run (one barrier per phase); within a phase
// operand fragments are double-buffered in registers one (tap, kstep) ahead of their MFMAs.
//
// Replaces: every nn.Conv2d / F.conv2d of savsr_arch.py (see include/savsr_hip.h).
#include "common.hpp"

namespace savsr {

struct ConvParams {
    const float* src[SAVSR_MAX_SRC];
    int src_pix[SAVSR_MAX_SRC];      // floats between pixels of source s
    int nsrc, src_ch;
    int h, w, cout, nchunk;
    const unsigned short* wimg;
    const float* bias;
    int act;
    float slope;
    const float* mul_px;
    const float* res1;
    int res1_pix;
    const float* res2;
    int res2_pix;
    float res2_scale;
    float* out;
    int out_pix;
    float* pool;          // optional [tiles][pool_stride]: per-workgroup channel sums of the stored values
    int pool_stride;
};

// Diagnostics (not used by the product path): per-workgroup s_memtime stamps of the conv kernel
// phases, enabled by savsr_debug_conv_stamps(1) and read back with savsr_debug_read_conv_stamps().
constexpr int STAMP_BLOCKS = 1024, STAMP_N = 6;
__device__ long long g_conv_stamps[STAMP_BLOCKS * STAMP_N];
__device__ int g_conv_stamps_on = 0;

__device__ __forceinline__ void stamp(int on, int slot) {
    if (on && threadIdx.x == 0) {
        const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (b < STAMP_BLOCKS) g_conv_stamps[b * STAMP_N + slot] = (slot == 5) ? (long long)__builtin_amdgcn_s_memrealtime() : (long long)__builtin_amdgcn_s_memtime();
    }
}

constexpr int CONV_MAX_BATCH = 6;
struct MultiConvParams {
    ConvParams c[CONV_MAX_BATCH];     // convs of identical geometry; blockIdx.z = conv * ncob + cob
    int ncob;
};

__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        hi[j] = h;
        lo[j] = (__bf16)(x[j] - (float)h);
    }
}

template <int KS, int NT, int TH, int DB, int PXT>
__global__ __launch_bounds__(64 * TH / PXT) void conv_bf16x3_kernel(const MultiConvParams mp) {
    // TH pixel rows per workgroup, PXT rows (32-pixel tiles) per wave: the A (weight) fragments a wave reads from
    // LDS serve PXT pixel tiles, so LDS reads per MFMA drop from 1 (PXT 1) to 2/3 (PXT 2)
    constexpr int STG = 1, NTHR = 64 * TH / PXT;
    const ConvParams& p = mp.c[blockIdx.z / mp.ncob];
    constexpr int TAPS = KS * KS, HALO = KS / 2;
    constexpr int KC = conv_kc(KS), KSTEPS = KC / 16;
    constexpr int IR = TH + 2 * HALO, IC = CONV_TW + 2 * HALO, NPX = IR * IC;
    constexpr int B_PART = KSTEPS * 2 * NPX;                 // 16-B units per part (hi or lo)
    constexpr int B_UNITS = 2 * B_PART;
    constexpr int W_UNITS = TAPS * KSTEPS * NT * 2 * 64;     // 16-B units per phase
    constexpr int B_ITEMS = STG ? B_PART * 2 : B_PART;     // STG 1: one float4 (4 channels) per item, 16 lines per wave load
    constexpr int B_IT = (B_ITEMS + NTHR - 1) / NTHR;
    constexpr int W_IT = (W_UNITS + NTHR - 1) / NTHR;
    constexpr int STEPS = TAPS * KSTEPS;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* smem = reinterpret_cast<bf16x8*>(smem_raw);      // 16-B units: [NB][B_UNITS] then [NB][W_UNITS], NB = DB ? 2 : 1
    constexpr int NB = DB ? 2 : 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, px = lane & 31;
    const int x0 = blockIdx.x * CONV_TW, y0 = blockIdx.y * TH, cob = blockIdx.z % mp.ncob;
    const int per_src = p.src_ch / KC;

    f32x4 b_reg[B_IT][2];
    f32x4 w_reg[W_IT];

    auto stage_load = [&](int chunk) {
        const int s = chunk / per_src;
        const int cb = (chunk - s * per_src) * KC;
        const float* base = p.src[0];
        int pix = p.src_pix[0];
        if (s == 1) { base = p.src[1]; pix = p.src_pix[1]; }
        if (s == 2) { base = p.src[2]; pix = p.src_pix[2]; }
        if (s == 3) { base = p.src[3]; pix = p.src_pix[3]; }
        if (s == 4) { base = p.src[4]; pix = p.src_pix[4]; }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int e = tid + i * NTHR;
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (e < B_ITEMS) {
                int q, pl, sub = 0;
                if (STG) {                                   // e = (pixel, float4-of-the-chunk): 4 lanes read one pixel's 64 B
                    constexpr int PER = KSTEPS * 4;
                    pl = e / PER;
                    const int c8 = e - pl * PER;
                    q = c8 >> 1;
                    sub = c8 & 1;
                } else {
                    q = e / NPX;                            // kstep * 2 + khalf
                    pl = e - q * NPX;
                }
                const int r = pl / IC, c = pl - r * IC;
                const int gy = y0 - HALO + r, gx = x0 - HALO + c;
                if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) {
                    const f32x4* g = reinterpret_cast<const f32x4*>(base + ((long long)gy * p.w + gx) * pix + cb + q * 8 + sub * 4);
                    v0 = g[0];
                    if (!STG) v1 = g[1];
                }
            }
            b_reg[i][0] = v0;
            b_reg[i][1] = v1;
        }
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.wimg) + ((long long)cob * p.nchunk + chunk) * W_UNITS;
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            const int e = tid + i * NTHR;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < W_UNITS) v = wsrc[e];
            w_reg[i] = v;
        }
    };
    auto stage_store = [&](int buf) {
        bf16x8* bl = smem + buf * B_UNITS;
        f32x4* wl = reinterpret_cast<f32x4*>(smem + NB * B_UNITS + buf * W_UNITS);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int e = tid + i * NTHR;
            if (e < B_ITEMS) {
                if (STG) {
                    constexpr int PER = KSTEPS * 4;
                    const int pl = e / PER, c8 = e - pl * PER, q = c8 >> 1, sub = c8 & 1;
                    bf16x4 hi, lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const __bf16 hh = (__bf16)b_reg[i][0][j];
                        hi[j] = hh;
                        lo[j] = (__bf16)(b_reg[i][0][j] - (float)hh);
                    }
                    bf16x4* dst = reinterpret_cast<bf16x4*>(bl + q * NPX + pl) + sub;
                    dst[0] = hi;
                    dst[B_PART * 2] = lo;
                } else {
                    bf16x8 hi, lo;
                    split8(b_reg[i][0], b_reg[i][1], hi, lo);
                    bl[e] = hi;
                    bl[B_PART + e] = lo;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            const int e = tid + i * NTHR;
            if (e < W_UNITS) wl[e] = w_reg[i];
        }
    };

    f32x16 acc[PXT][NT];
#pragma unroll
    for (int j = 0; j < PXT; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

    const int stamps_on = __builtin_amdgcn_readfirstlane(g_conv_stamps_on);
    stamp(stamps_on, 0);
    stamp(stamps_on, 5);
    stage_load(0);
    stage_store(0);
    __syncthreads();
    stamp(stamps_on, 1);

    struct Frag { bf16x8 ah[NT], al[NT], bh[PXT], bl[PXT]; };

    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
        const int buf = DB ? (chunk & 1) : 0;
        const bool more = chunk + 1 < p.nchunk;
        if (more) stage_load(chunk + 1);

        const bf16x8* bl = smem + buf * B_UNITS;
        const bf16x8* wl = smem + NB * B_UNITS + buf * W_UNITS;
        const bf16x8* bbase = bl + half * NPX + (wave * PXT) * IC + px;
        const bf16x8* abase = wl + lane;

        auto load_frag = [&](int s, Frag& f) {
            const int tap = s / KSTEPS, ks = s - tap * KSTEPS;
            const int ky = tap / KS, kx = tap - ky * KS;
            const int bo = ks * 2 * NPX + ky * IC + kx;
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                f.bh[j] = bbase[bo + j * IC];
                f.bl[j] = bbase[B_PART + bo + j * IC];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f.ah[t] = abase[((s * NT + t) * 2 + 0) * 64];
                f.al[t] = abase[((s * NT + t) * 2 + 1) * 64];
            }
        };
        auto mma = [&](const Frag& f) {
#pragma unroll
            for (int j = 0; j < PXT; ++j)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[t], f.bh[j], acc[j][t], 0, 0, 0);
                    acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[t], f.bl[j], acc[j][t], 0, 0, 0);
                    acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[t], f.bh[j], acc[j][t], 0, 0, 0);
                }
        };
        // Operand fragments run two (tap, kstep) steps ahead of their MFMAs.  hipcc (ROCm 7.2) otherwise
        // sinks every ds_read to just before its MFMA and re-uses one register quad for all A fragments
        // (ds_read; s_waitcnt lgkmcnt(0); v_mfma ...: MFMA pipe 33 % busy, measured), so the issue order is
        // pinned with sched_barriers: [reads of step s+2] | [6 NT/2 MFMAs of step s] | ...
        Frag f[3];
        load_frag(0, f[0]);
        if (STEPS > 1) load_frag(1, f[1]);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            if (s + 2 < STEPS) load_frag(s + 2, f[(s + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
            mma(f[s % 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (chunk == 0) stamp(stamps_on, 2);
        if (DB) {
            if (more) stage_store(buf ^ 1);
            __syncthreads();
        } else {                                  // single-buffered: everyone is done reading, then refill
            __syncthreads();
            if (more) {
                stage_store(0);
                __syncthreads();
            }
        }
    }
    stamp(stamps_on, 3);

    // ---- epilogue ---------------------------------------------------------------------------
    // Lane (pixel, half) holds channels 32 t + 8 g + 4 half + {0..3} in accumulator regs 4g..4g+3.
    // Stored straight from that layout, a wave instruction would touch 32 lines with 32 B each
    // (measured: 17.8 k cycles of store drain per workgroup).  Instead each wave transposes its
    // 32 px x COT tile through its own LDS slice and stores whole pixel records: consecutive
    // lanes write consecutive 16 B, 1 KiB contiguous per instruction when out_pix == cout; the
    // residual reads are coalesced the same way.  (All waves passed the K loop's last barrier,
    // so the staging buffers are free.)
    constexpr int COT = 32 * NT, EPS = COT + 4, U = COT / 4;
    float* ep = reinterpret_cast<float*>(smem_raw) + wave * (PXT * 32 * EPS);
#pragma unroll
    for (int j = 0; j < PXT; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[j][t][4 * g], acc[j][t][4 * g + 1], acc[j][t][4 * g + 2], acc[j][t][4 * g + 3]};
                *reinterpret_cast<f32x4*>(ep + (j * 32 + px) * EPS + 32 * t + 8 * g + 4 * half) = v;
            }
    f32x4 psum = {0.f, 0.f, 0.f, 0.f};         // pooled sums: lane l always handles channel quad l % U
#pragma unroll
    for (int i = 0; i < PXT * U / 2; ++i) {
        const int unit = lane + 64 * i;
        const int pl = unit / U, c4 = unit - pl * U;          // pl = row-in-wave * 32 + column
        const int y = y0 + wave * PXT + (pl >> 5);
        const int x = x0 + (pl & 31), co = cob * COT + 4 * c4;
        if (y >= p.h || x >= p.w || co >= p.cout) continue;
        const long long pidx = (long long)y * p.w + x;
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ep + pl * EPS + 4 * c4);
        float v[4] = {a4[0], a4[1], a4[2], a4[3]};
        const bool full = co + 3 < p.cout;
        const float mul = p.mul_px ? p.mul_px[pidx] : 1.f;
        if (p.bias) {
            if (full) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + co);
                v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
            } else {
                for (int q = 0; q < 4 && co + q < p.cout; ++q) v[q] += p.bias[co + q];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (p.act == SAVSR_ACT_RELU) v[q] = fmaxf(v[q], 0.f);
            else if (p.act == SAVSR_ACT_LRELU) v[q] = v[q] > 0.f ? v[q] : v[q] * p.slope;
            else if (p.act == SAVSR_ACT_SIGMOID) v[q] = sigmoidf_(v[q]);
            v[q] *= mul;
        }
        float* o = p.out + pidx * p.out_pix + co;
        if (full) {
            if (p.res1) {
                const f32x4 r = *reinterpret_cast<const f32x4*>(p.res1 + pidx * p.res1_pix + co);
                v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
            }
            if (p.res2) {
                const f32x4 r = *reinterpret_cast<const f32x4*>(p.res2 + pidx * p.res2_pix + co);
                v[0] += p.res2_scale * r[0]; v[1] += p.res2_scale * r[1]; v[2] += p.res2_scale * r[2]; v[3] += p.res2_scale * r[3];
            }
            const f32x4 ov = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(o) = ov;
            psum[0] += v[0]; psum[1] += v[1]; psum[2] += v[2]; psum[3] += v[3];
        } else {
            for (int q = 0; q < 4 && co + q < p.cout; ++q) {
                float vv = v[q];
                if (p.res1) vv += p.res1[pidx * p.res1_pix + co + q];
                if (p.res2) vv += p.res2_scale * p.res2[pidx * p.res2_pix + co + q];
                o[q] = vv;
            }
        }
    }
    if (p.pool) {
        // AdaptiveAvgPool2d(1) of the tensor just produced (savsr_arch.py:146,515), fused: lanes with equal
        // l % U hold the same channel quad -> butterfly over the 64 / U pixel groups, then the waves of the
        // workgroup are summed in wave order through LDS (deterministic) and ONE row per workgroup is written.
#pragma unroll
        for (int o = U; o < 64; o <<= 1) {
            psum[0] += __shfl_xor(psum[0], o, 64); psum[1] += __shfl_xor(psum[1], o, 64);
            psum[2] += __shfl_xor(psum[2], o, 64); psum[3] += __shfl_xor(psum[3], o, 64);
        }
        __syncthreads();                         // every wave is done with its transpose slice
        float* pl_ = reinterpret_cast<float*>(smem_raw);
        if (lane < U) *reinterpret_cast<f32x4*>(pl_ + wave * COT + 4 * lane) = psum;
        __syncthreads();
        constexpr int NW = TH / PXT;
        if (tid < COT) {
            float sacc = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) sacc += pl_[wv * COT + tid];
            if (cob * COT + tid < p.cout)
                p.pool[(long long)(blockIdx.y * gridDim.x + blockIdx.x) * p.pool_stride + cob * COT + tid] = sacc;
        }
    }
    if (stamps_on) {
        __builtin_amdgcn_s_waitcnt(0);          // diagnostics: include the store drain in the last stamp
        stamp(stamps_on, 4);
    }
}

template <int KS, int NT, int TH, int DB, int PXT>
static int launch_conv_v(const MultiConvParams& mp, int nconv, hipStream_t st) {
    const ConvParams& p = mp.c[0];
    constexpr int TAPS = KS * KS, HALO = KS / 2, KC = conv_kc(KS), KSTEPS = KC / 16;
    constexpr int NPX = (TH + 2 * HALO) * (CONV_TW + 2 * HALO);
    constexpr size_t stage = 16ull * (DB ? 2 : 1) * (2 * KSTEPS * 2 * NPX + TAPS * KSTEPS * NT * 2 * 64);
    constexpr size_t epi = 4ull * TH * 32 * (32 * NT + 4);
    constexpr size_t lds = stage > epi ? stage : epi;
    static bool attr_done = false;      // benign race: idempotent attribute set
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_kernel<KS, NT, TH, DB, PXT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("conv: hipFuncSetAttribute(%zu bytes LDS) failed: %s", lds, hipGetErrorString(e));
            return (int)e;
        }
        attr_done = true;
    }
    dim3 grid((p.w + CONV_TW - 1) / CONV_TW, (p.h + TH - 1) / TH, mp.ncob * nconv);
    hipLaunchKernelGGL((conv_bf16x3_kernel<KS, NT, TH, DB, PXT>), grid, dim3(64 * TH / PXT), lds, st, mp);
    return check_launch("conv_bf16x3_kernel");
}

// SAVSR_CONV_VARIANT (tuning knob, read once):
//   0 = 8-row tiles, 8 waves x 1 row,  double-buffered staging, 1 workgroup / CU
//   1 = 4-row tiles, 4 waves x 1 row,  single-buffered staging, 2-3 workgroups / CU
//   2 = 8-row tiles, 4 waves x 2 rows, double-buffered staging, 1 workgroup / CU
//   3 = 8-row tiles, 4 waves x 2 rows, single-buffered staging, 2 workgroups / CU
static int conv_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SAVSR_CONV_VARIANT");
        v = e ? atoi(e) & 3 : 0;
    }
    return v;
}

template <int KS, int NT>
static int launch_conv(const MultiConvParams& mp, int nconv, hipStream_t st) {
    switch (conv_variant()) {
        case 0: return launch_conv_v<KS, NT, 8, 1, 1>(mp, nconv, st);
        case 1: return launch_conv_v<KS, NT, 4, 0, 1>(mp, nconv, st);
        case 2: return launch_conv_v<KS, NT, 8, 1, 2>(mp, nconv, st);
        default: return launch_conv_v<KS, NT, 8, 0, 2>(mp, nconv, st);
    }
}

}  // namespace savsr

using namespace savsr;

extern "C" int savsr_debug_conv_stamps(int enable) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps_on), &enable, sizeof(int));
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int savsr_debug_read_conv_stamps(long long* host, int nblocks) {
    if (!host || nblocks < 1 || nblocks > STAMP_BLOCKS) return fail_arg("debug_read_conv_stamps");
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_stamps), sizeof(long long) * STAMP_N * nblocks);
    return e == hipSuccess ? 0 : (int)e;
}

// Rows of a conv's `pool` output = pixel tiles of its launch (depends on the tile variant in use).
extern "C" int savsr_conv_pool_blocks(int h, int w) {
    const int th = (conv_variant() == 1) ? 4 : 8;
    return ((w + CONV_TW - 1) / CONV_TW) * ((h + th - 1) / th);
}

extern "C" int64_t savsr_conv_packed_elems(int cout, int cin, int ksize) {
    if (cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return -1;
    const int kc = conv_kc(ksize), cot = conv_cot(cout);
    if (cin % kc) return -1;
    const int64_t nchunk = cin / kc, ncob = (cout + cot - 1) / cot;
    return ncob * nchunk * ksize * ksize * kc * cot;            // per part (hi or lo)
}

// Position of W[co][ci][tap] inside ONE part of the image (in elements); the hi part of a
// (cob, chunk, tap, kstep, t) group is followed by its lo part, so the bf16 image index is
//   group * 1024 + part * 512 + (index % 512)   with group = index / 512.
extern "C" int64_t savsr_conv_pack_index(int cout, int cin, int ksize, int co, int ci, int tap) {
    const int kc = conv_kc(ksize), cot = conv_cot(cout), nt = cot / 32, ksteps = kc / 16;
    const int64_t nchunk = cin / kc;
    const int cob = co / cot, col = co % cot, t = col / 32, row = col % 32;
    const int chunk = ci / kc, cl = ci % kc, ks = cl / 16, kh = (cl % 16) / 8, j = cl % 8;
    const int64_t group = (((int64_t)(cob * nchunk + chunk) * (ksize * ksize) + tap) * ksteps + ks) * nt + t;
    return group * 512 + (kh * 32 + row) * 8 + j;
}

static int fill_params(const savsr_conv_desc* d, ConvParams& p) {
    if (d->ksize != 1 && d->ksize != 3) return fail_arg("conv: ksize must be 1 or 3");
    const int kc = conv_kc(d->ksize);
    if (d->nsrc < 1 || d->nsrc > SAVSR_MAX_SRC || d->src_ch < kc || d->src_ch % kc) return fail_arg("conv: nsrc / src_ch (multiple of 16, or 32 for 1x1)");
    if (d->cin != d->nsrc * d->src_ch) return fail_arg("conv: cin != nsrc*src_ch");
    if (d->h < 1 || d->w < 1 || d->cout < 1) return fail_arg("conv: shape");
    if (!d->wpacked || !d->out) return fail_arg("conv: null weights/out");
    uintptr_t al = reinterpret_cast<uintptr_t>(d->wpacked) | reinterpret_cast<uintptr_t>(d->out) | (uintptr_t)(d->out_pix * 4);
    for (int i = 0; i < SAVSR_MAX_SRC; ++i) {
        const bool on = i < d->nsrc;
        if (on && !d->src[i]) return fail_arg("conv: null source");
        p.src[i] = on ? d->src[i] : nullptr;
        p.src_pix[i] = on ? d->src_pix[i] : 0;
        if (on) al |= reinterpret_cast<uintptr_t>(d->src[i]) | (uintptr_t)(d->src_pix[i] * 4);
    }
    if (d->res1) al |= reinterpret_cast<uintptr_t>(d->res1) | (uintptr_t)(d->res1_pix * 4);
    if (d->res2) al |= reinterpret_cast<uintptr_t>(d->res2) | (uintptr_t)(d->res2_pix * 4);
    if (d->bias) al |= reinterpret_cast<uintptr_t>(d->bias);
    if ((al & 15) && d->cout >= 4) {
        set_error("conv: sources / residuals / bias / out / weights must be 16-byte aligned with pixel strides multiple of 4 floats");
        return SAVSR_E_ALIGN;
    }
    p.nsrc = d->nsrc; p.src_ch = d->src_ch;
    p.h = d->h; p.w = d->w; p.cout = d->cout;
    p.nchunk = d->cin / kc;
    p.wimg = reinterpret_cast<const unsigned short*>(d->wpacked);
    p.bias = d->bias; p.act = d->act; p.slope = d->slope;
    p.mul_px = d->mul_px; p.res1 = d->res1; p.res1_pix = d->res1_pix; p.res2 = d->res2; p.res2_pix = d->res2_pix;
    p.res2_scale = d->res2_scale;
    p.out = d->out; p.out_pix = d->out_pix;
    p.pool = d->pool; p.pool_stride = d->pool_stride;
    if (d->pool && (d->cout % 4 || d->pool_stride < d->cout)) return fail_arg("conv: pool needs cout % 4 == 0 and pool_stride >= cout");
    return 0;
}

extern "C" int savsr_conv2d_batch(const savsr_conv_desc* descs, int n, void* stream) {
    if (!descs) return fail_arg("conv: null descriptor");
    if (n < 1 || n > CONV_MAX_BATCH) return fail_arg("conv: batch size must be 1..6");
    MultiConvParams mp;
    for (int i = 0; i < n; ++i) {
        const int rc = fill_params(descs + i, mp.c[i]);
        if (rc) return rc;
        const savsr_conv_desc& a = descs[0];
        const savsr_conv_desc& b = descs[i];
        if (b.ksize != a.ksize || b.nsrc != a.nsrc || b.src_ch != a.src_ch || b.h != a.h || b.w != a.w || b.cout != a.cout)
            return fail_arg("conv: all convs of a batch must share ksize / nsrc / src_ch / h / w / cout");
    }
    for (int i = n; i < CONV_MAX_BATCH; ++i) mp.c[i] = mp.c[0];
    const savsr_conv_desc* d = descs;
    const int cot = conv_cot(d->cout);
    mp.ncob = (d->cout + cot - 1) / cot;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool wide = cot == 64;
    if (d->ksize == 3) return wide ? launch_conv<3, 2>(mp, n, st) : launch_conv<3, 1>(mp, n, st);
    return wide ? launch_conv<1, 2>(mp, n, st) : launch_conv<1, 1>(mp, n, st);
}

extern "C" int savsr_conv2d(const savsr_conv_desc* d, void* stream) { return savsr_conv2d_batch(d, 1, stream); }
