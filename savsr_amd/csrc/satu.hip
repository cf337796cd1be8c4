// SATU -- Spatio-temporal Adaptive arbitrary-scale Upsampling (STAUpsample.forward,
// savsr_arch.py:315-376), restructured for gfx950 (derivation in DESIGN.md):
//
//   out = G(Wa sta, soff) + G(Wb x, off) + sum_n r_n (Wb E_n) ( sum_m r_m C_m G(x, off) ) + b
//
//   * the coordinate MLP only depends on (coor_h[Y], coor_w[X])  -> evaluated once per DISTINCT
//     pair (phase table), not once per HR pixel;
//   * the 1x1 `fusion` and the expert `compress` matrices commute with the bilinear gather G
//     -> applied at LR resolution (LR stage), so the HR stage only gathers 160 LR channels,
//     mixes 32 numbers and runs a K=32 MFMA per pixel;
//   * nothing of the reference's 3.5 GB of per-pixel expert weights / 369 MB kernel tensor /
//     369 MB unfolded features ever exists.
//
// Record layout of the LR-side tensor LRcat[h][w][160] (one 640-B record per LR pixel):
//   [64 hh, 64 hh + 32)       (Wa sta)[co] at q = 16 t + r  <->  co = 32 t + acc_row(r, hh)   (hh = MFMA lane half)
//   [64 hh + 32, 64 hh + 64)  (Wb x)[co]   same q
//   [128, 160)                (C_m x)[j]   at 8 m + j  (natural order, shared by both halves)
//   i.e. the 32x32 MFMA accumulator layout of the lane that produced a value (LR stage) and of
//   the lane that consumes it (HR stage): no cross-lane movement on either side.
#include "common.hpp"

// Timing experiments on the product LR kernel (results invalid; never set in a shipped build): 1 = weight fragments not
// re-read per tap, 2 = x / bias quads not re-read per group; 8 = every workgroup records its s_memtime total in stamp
// slot 7 (cycles are the comparable figure: the variants change the power draw and with it the shader clock).
#ifndef LR_EXP
#define LR_EXP 0
#endif
#ifndef LR_INTERLEAVE
#define LR_INTERLEAVE 1
#endif

namespace savsr {

// Floats per LRcat record for NB 32-row output blocks per projection: (A | B) per lane half, then the 32 compressed channels.
// NB = 2: the standalone SATU (64 fused channels, SAVSR_SATU_LRCAT = 160); NB = 1: the tail-projected form (27 of 32 rows
// used, SAVSR_SATU_LRCAT_TAIL = 96), see the header comment of the HR stage.
__host__ __device__ constexpr int rec_floats(int nb) { return 64 * nb + 32; }
static_assert(rec_floats(2) == SAVSR_SATU_LRCAT && rec_floats(1) == SAVSR_SATU_LRCAT_TAIL, "record sizes");

// Diagnostics (never used by the product path): accumulated s_memtime deltas of kernel sections, written by
// wave 0 of each workgroup when enabled with savsr_debug_satu_stamps(1).
constexpr int SSTAMP_BLOCKS = 2048, SSTAMP_N = 8;
__device__ long long g_satu_stamps[SSTAMP_BLOCKS * SSTAMP_N];
__device__ int g_satu_stamps_on = 0;
#define SATU_T() ((long long)__builtin_amdgcn_s_memtime())

// ------------------------------------------------------------------------------------------
// Phase table: one wave per distinct (coor_h, coor_w) pair; lane j owns hidden unit j.
// savsr_arch.py:335-350 (body, offset, st_offset, routing).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void satu_phase_table_kernel(const savsr_satu_weights wt, const float* __restrict__ uniq_ch, int n_uh,
                                                               const float* __restrict__ uniq_cw, int n_uw, float inv_sw, float inv_sh,
                                                               float* __restrict__ table) {
    const int lane = threadIdx.x & 63;
    const long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= (long long)n_uh * n_uw) return;
    const int uh = (int)(e / n_uw), uw = (int)(e - (long long)uh * n_uw);
    const float in0 = inv_sw, in1 = inv_sh, in2 = uniq_ch[uh], in3 = uniq_cw[uw];   // :336-339 (w before h)
    const float* w0 = wt.body0_w + lane * 4;
    float h1 = wt.body0_b[lane] + w0[0] * in0 + w0[1] * in1 + w0[2] * in2 + w0[3] * in3;
    h1 = fmaxf(h1, 0.f);
    float h2 = wt.body2_b[lane];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h1), i));
        h2 += wt.body2_w[i * 64 + lane] * hi;          // body2_w is stored [in][out]
    }
    h2 = fmaxf(h2, 0.f);
#pragma unroll
    for (int o = 0; o < SAVSR_SATU_TABLE; ++o) {
        float v = wave_sum(wt.head_w[o * 64 + lane] * h2) + wt.head_b[o];
        if (o < 4) v = sigmoidf_(v);                    // routing is sigmoid, not softmax (:253)
        if (lane == 0) table[e * SAVSR_SATU_TABLE + o] = v;
    }
}

// ------------------------------------------------------------------------------------------
// LR stage.  Block = 4 waves = 4 rows x 32 cols of LR pixels; wave w owns row w.
//   K[n][px]   = LReLU_0.1( Wk[n][:] . st[:, px] + bk[n] ),  n = 25 c + tap       (:226-228,319)
//   sta[c][px] = sum_tap K[25c+tap][px] * x_rep[c][y+ky-2][x+kx-2]                (:297-313)
// as 50 (tap, channel-group) GEMM tiles of 32 rows x 32 px x K=64 on v_mfma_f32_32x32x16_bf16 with
// split-bf16 operands (hi*hi + hi*lo + lo*hi, see conv_mfma.hip); the K tile never leaves the
// accumulator registers.  Then the three LR-side projections (Wa sta | Wb x | C x), with sta
// consumed straight from its accumulator registers (k order = accumulator order).
// x, st: channel-last [..][..][pix] fp32 crops (row pitch `row_px` pixels).
// ------------------------------------------------------------------------------------------
struct LrParams {
    savsr_satu_weights wt;
    const float* x;
    const float* st;
    int pix, row_px, h, w;
    float* lrcat;
};

constexpr int LR_TH = 8, LR_TW = 32, LR_HALO = 2;   // workgroup = 8 waves = 8 rows x 32 cols of LR pixels
constexpr int LR_XR = LR_TH + 2 * LR_HALO;     // 12
constexpr int LR_XC = LR_TW + 2 * LR_HALO;     // 36
constexpr int LR_NPX = LR_XR * LR_XC;          // 432 pixels in the x tile
constexpr int LR_XS = 36;                      // floats per pixel record in LDS (32 used; 144 B keeps b128 reads conflict-free)
constexpr int LR_SLAB = 4 * 2 * 64;            // 16-B units of one (tap, cg) weight slab: [ks][part][lane]
constexpr int LR_PHASE = 5 * LR_SLAB;          // one phase = the 5 taps of a kernel row (40 KB of weights)

__device__ __forceinline__ void split8v(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        hi[j] = h;
        lo[j] = (__bf16)(x[j] - (float)h);
    }
}

__device__ __forceinline__ f32x16 mma3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// 16 B per lane, global -> LDS without registers (LDS-DMA); `lds_dst` = wave-uniform LDS address of lane 0's 16 B.
// hipcc does not count this load: the consumer waits with an explicit s_waitcnt vmcnt(0) before its barrier.
__device__ __forceinline__ void glds16(const void* gsrc, const void* lds_dst) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

template <bool DIAG, int NB>      // DIAG: instrumented build (section stamps), launched only while savsr_debug_satu_stamps is on
__global__ __launch_bounds__(512, 2) void satu_lr_kernel(const LrParams p) {
    constexpr int REC = rec_floats(NB);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xt = smem;                                                   // [432][36]: replicate-padded x tile of one channel group
    bf16x8* wbuf = reinterpret_cast<bf16x8*>(smem + LR_NPX * LR_XS);    // [2][LR_PHASE]: weight slabs of a kernel row, double buffered
    float* kbl = smem + LR_NPX * LR_XS + 2 * LR_PHASE * 4;              // [25][64] kernel_conv bias

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, px = lane & 31;
    const int x0 = blockIdx.x * LR_TW, y0 = blockIdx.y * LR_TH;
    const int gy = y0 + wave, gx = x0 + px;
    const bool valid = gy < p.h && gx < p.w;
    const int cy = gy < p.h ? gy : p.h - 1, cx = gx < p.w ? gx : p.w - 1;
    const long long cpix = ((long long)cy * p.row_px + cx) * p.pix + 8 * half;

    const int stamps_on = DIAG ? __builtin_amdgcn_readfirstlane(g_satu_stamps_on) & 1 : 0;
    long long tacc[SSTAMP_N] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_prev = stamps_on ? SATU_T() : 0;
    const long long t_begin = t_prev;
    const long long t_lr0 = (LR_EXP & 8) ? SATU_T() : 0;
#define LR_MARK(i) do { if (stamps_on) { const long long t_now = SATU_T(); tacc[i] += t_now - t_prev; t_prev = t_now; } } while (0)

    const bf16x8* kw = reinterpret_cast<const bf16x8*>(p.wt.kconv_w);
    // weight slabs of phase ph (= channel group ph / 5, kernel row ph % 5) -> LDS buffer b: 40 pieces of 1 KiB, 5 per wave
    auto dma_phase = [&](int ph, int b) {
        const int cg = ph / 5, ky = ph - 5 * cg;
#pragma unroll
        for (int i = 0; i < 5; ++i)
            glds16(kw + (long long)((ky * 5 + i) * 2 + cg) * LR_SLAB + wave * 64 + lane, wbuf + b * LR_PHASE + i * LR_SLAB + wave * 64);
    };
    // replicate-padded x tile of channel group cg (F.pad replicate, :302): global -> registers ...
    constexpr int XT_IT = (LR_NPX * 8 + 511) / 512;                    // 7 float4 per thread
    f32x4 xv[XT_IT];
    auto xt_load = [&](int cg) {
#pragma unroll
        for (int i = 0; i < XT_IT; ++i) {
            const int e = tid + i * 512;
            const int pl = (e < LR_NPX * 8 ? e : 0) >> 3, c4 = e & 7;
            const int r = pl / LR_XC, c = pl - r * LR_XC;
            int sy = y0 - LR_HALO + r, sx = x0 - LR_HALO + c;
            sy = sy < 0 ? 0 : (sy > p.h - 1 ? p.h - 1 : sy);
            sx = sx < 0 ? 0 : (sx > p.w - 1 ? p.w - 1 : sx);
            xv[i] = *reinterpret_cast<const f32x4*>(p.x + ((long long)sy * p.row_px + sx) * p.pix + 32 * cg + 4 * c4);
        }
    };
    auto xt_store = [&]() {                                             // ... -> LDS
#pragma unroll
        for (int i = 0; i < XT_IT; ++i) {
            const int e = tid + i * 512;
            if (e < LR_NPX * 8) *reinterpret_cast<f32x4*>(xt + (e >> 3) * LR_XS + 4 * (e & 7)) = xv[i];
        }
    };

    // ---- prologue ---------------------------------------------------------------------------------------------
    dma_phase(0, 0);
    xt_load(0);
    // B operand of the kernel-prediction GEMM: st[16 ks + 8 half + j][pixel], resident for all 50 tiles
    bf16x8 sth[4], stl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const f32x4* g = reinterpret_cast<const f32x4*>(p.st + cpix + 16 * ks);
        split8v(g[0], g[1], sth[ks], stl[ks]);
    }
    for (int e = tid; e < 25 * 64 / 4; e += 512) reinterpret_cast<f32x4*>(kbl)[e] = reinterpret_cast<const f32x4*>(p.wt.kconv_b)[e];
    xt_store();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    LR_MARK(0);                                       // prologue: first slabs, x tile, st fragments

    struct AFrag { bf16x8 ah[4], al[4]; };
    f32x16 sta[2];
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) {                  // unrolled: sta[cg] must stay in registers
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll 1
        for (int ky = 0; ky < 5; ++ky) {
            const int ph = cg * 5 + ky, buf = ph & 1;
            // staged under this phase: the next kernel row's slabs; at the end of a channel group also the next x tile;
            // under the very last phase the projection weights (same 40 KB) and the centre pixel's x
            if (ph + 1 < 10) dma_phase(ph + 1, buf ^ 1);
            else {                                    // projection image: (2 NB + 1) x 8 KB
#pragma unroll
                for (int i = 0; i < 2 * NB + 1; ++i)
                    glds16(reinterpret_cast<const bf16x8*>(p.wt.proj_w) + (i * 8 + wave) * 64 + lane, wbuf + (buf ^ 1) * LR_PHASE + (i * 8 + wave) * 64);
            }

            const bf16x8* wl = wbuf + buf * LR_PHASE + lane;
            auto load_frag = [&](int kx, AFrag& f) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { f.ah[ks] = wl[kx * LR_SLAB + (ks * 2 + 0) * 64]; f.al[ks] = wl[kx * LR_SLAB + (ks * 2 + 1) * 64]; }
            };
            // LDS operands of the VALU work are read ONE group ahead of their use (a read next to its use exposes the LDS
            // latency in every group: the wave cannot issue its next MFMAs while it waits)
            auto bias_read = [&](int kx, int g) -> f32x4 {       // kernel_conv bias (the initial accumulator), quad g
                return *reinterpret_cast<const f32x4*>(kbl + (ky * 5 + kx) * 64 + cg * 32 + 4 * half + 8 * g);
            };
            auto x_read = [&](int kx, int g) -> f32x4 {          // x_pad at tap (ky, kx), channel quad g of this half
                return *reinterpret_cast<const f32x4*>(xt + ((wave + ky) * LR_XC + px + kx) * LR_XS + 4 * half + 8 * g);
            };
            auto lrelu_x = [&](int g, const f32x16& acc, const f32x4 xq) {   // sta += LeakyReLU_0.1(K) * x_pad   (:228, :297-313)
                // 2 vector instructions per element: v_pk_mul (0.1 k), v_max (one: fmaxf costs two, NaN canonicalisation), v_pk_fma.
                // The max and the accumulation are VOLATILE asm: pure arithmetic has no ordering against the sched_barriers
                // between the MFMA groups, and hipcc's DAG scheduler collected all of a phase's LeakyReLU * x work (260
                // instructions) into one clump behind the barrier, serial to the phase's 60 MFMAs (ISA listing).
                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x2 k = {acc[4 * g + 2 * h2], acc[4 * g + 2 * h2 + 1]};
                    const f32x2 t = k * f32x2{0.1f, 0.1f};
                    float m0, m1;
                    asm volatile("v_max_f32 %0, %2, %3\n\tv_max_f32 %1, %4, %5" : "=&v"(m0), "=&v"(m1) : "v"(k[0]), "v"(t[0]), "v"(k[1]), "v"(t[1]));
                    f32x2 sv = {sacc[4 * g + 2 * h2], sacc[4 * g + 2 * h2 + 1]};
                    const f32x2 m = {m0, m1}, x2 = {xq[2 * h2], xq[2 * h2 + 1]};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(sv) : "v"(m), "v"(x2));
                    sacc[4 * g + 2 * h2] = sv[0];
                    sacc[4 * g + 2 * h2 + 1] = sv[1];
                }
            };
            AFrag fr;                                 // ONE fragment set: a k-step's pair is reloaded for the next tap as soon as its
            f32x16 acc[2];                            // MFMAs are issued (they have consumed their operands); two sets spill
            load_frag(0, fr);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = bias_read(0, g);
                acc[0][4 * g] = b4[0]; acc[0][4 * g + 1] = b4[1]; acc[0][4 * g + 2] = b4[2]; acc[0][4 * g + 3] = b4[3];
            }
            f32x4 b_pf = bias_read(1, 0), x_pf = b_pf;
            // 20 groups of 3 MFMAs (tap kx = G / 4, k-step ks = G % 4).  The VALU work of the previous tap and the next tap's
            // bias go BETWEEN the groups: both waves of a SIMD run in lockstep, so VALU work placed after a tap's 12 MFMAs is
            // serial to them (7.9 k cycles per phase for 4.0 k of MFMAs, stamps).
#pragma unroll
            for (int G = 0; G < 20; ++G) {
                const int kx = G / 4, ks = G % 4;
                __builtin_amdgcn_sched_barrier(0);
                acc[kx & 1] = mma3(fr.ah[ks], fr.al[ks], sth[ks], stl[ks], acc[kx & 1]);
#if !LR_INTERLEAVE
                __builtin_amdgcn_sched_barrier(0);
#endif
                if (!(LR_EXP & 1) && kx + 1 < 5) { fr.ah[ks] = wl[(kx + 1) * LR_SLAB + (ks * 2 + 0) * 64]; fr.al[ks] = wl[(kx + 1) * LR_SLAB + (ks * 2 + 1) * 64]; }
                if (kx > 0) lrelu_x(ks, acc[(kx - 1) & 1], x_pf);
                if (kx + 1 < 5) {
                    f32x16& an = acc[(kx + 1) & 1];
                    an[4 * ks] = b_pf[0]; an[4 * ks + 1] = b_pf[1]; an[4 * ks + 2] = b_pf[2]; an[4 * ks + 3] = b_pf[3];
                }
                const int G1 = G + 1, kx1 = G1 / 4, ks1 = G1 % 4;
                if (G1 < 20 && !(LR_EXP & 2)) {
                    if (kx1 > 0) x_pf = x_read(kx1 - 1, ks1);
                    if (kx1 + 1 < 5) b_pf = bias_read(kx1 + 1, ks1);
                }
#if LR_INTERLEAVE
                // the group is one basic block: ~5 vector instructions and one LDS read behind each of its 3 MFMAs
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                }
#endif
            }
            {
                f32x4 xq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) xq[g] = x_read(4, g);
#pragma unroll
                for (int g = 0; g < 4; ++g) lrelu_x(g, acc[0], xq[g]);
            }
            LR_MARK(1);                               // 5 taps: fragment reads, 60 MFMAs, LeakyReLU * x
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the staged slabs (and x loads) have landed
            LR_MARK(2);
            __syncthreads();
            if (ky == 4 && cg == 0) {                 // every wave is done with the old x tile: swap it (once per workgroup; loading
                xt_load(1);                           // it under phase 4 would hold 28 registers across the loop and spills)
                xt_store();
                __syncthreads();
            }
            LR_MARK(3);                               // barrier(s)
        }
        sta[cg] = sacc;
    }

    // ---- LR-side projections (bf16x3): proj image = A [t < NB][kidx 4][part][lane] | B [t < NB][ks 4][part][lane] | C [ks 4][part][lane],
    // now in LDS buffer 0 (phase 9 ran from buffer 1)
    const bf16x8* pa = wbuf + lane;
    const bf16x8* pb = pa + NB * 4 * 2 * 64;
    const bf16x8* pc = pb + NB * 4 * 2 * 64;
    f32x4 xc[8];                                      // the centre pixel's x: B operand of the Wb / C projections
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const f32x4* g = reinterpret_cast<const f32x4*>(p.x + cpix + 16 * ks);
        xc[2 * ks] = g[0];
        xc[2 * ks + 1] = g[1];
    }
    f32x16 accA[NB], accB[NB], accC;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int t = 0; t < NB; ++t) { accA[t][r] = 0.f; accB[t][r] = 0.f; }
        accC[r] = 0.f;
    }
#pragma unroll
    for (int cg = 0; cg < 2; ++cg)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // accumulator regs 8s..8s+7 are rows 16 s + 8 (j >> 2) + 4 half + (j & 3): the k order of this step
            const f32x4 lo4 = {sta[cg][8 * s], sta[cg][8 * s + 1], sta[cg][8 * s + 2], sta[cg][8 * s + 3]};
            const f32x4 hi4 = {sta[cg][8 * s + 4], sta[cg][8 * s + 5], sta[cg][8 * s + 6], sta[cg][8 * s + 7]};
            bf16x8 bh, bl;
            split8v(lo4, hi4, bh, bl);
            const int kidx = cg * 2 + s;
#pragma unroll
            for (int t = 0; t < NB; ++t)
                accA[t] = mma3(pa[((t * 4 + kidx) * 2 + 0) * 64], pa[((t * 4 + kidx) * 2 + 1) * 64], bh, bl, accA[t]);
        }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bf16x8 xh, xl;
        split8v(xc[2 * ks], xc[2 * ks + 1], xh, xl);
#pragma unroll
        for (int t = 0; t < NB; ++t)
            accB[t] = mma3(pb[((t * 4 + ks) * 2 + 0) * 64], pb[((t * 4 + ks) * 2 + 1) * 64], xh, xl, accB[t]);
        accC = mma3(pc[(ks * 2 + 0) * 64], pc[(ks * 2 + 1) * 64], xh, xl, accC);
    }
    LR_MARK(4);                                        // projections
    if ((LR_EXP & 8) && !DIAG && __builtin_amdgcn_readfirstlane(tid) == 0) {      // (scalar branch: all of wave 0 stores)
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        if (b < SSTAMP_BLOCKS) g_satu_stamps[b * SSTAMP_N + 7] = SATU_T() - t_lr0;
    }
    if (stamps_on && tid == 0) {
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        if (b < SSTAMP_BLOCKS) {
            for (int i = 0; i < 7; ++i) g_satu_stamps[b * SSTAMP_N + i] = tacc[i];
            g_satu_stamps[b * SSTAMP_N + 7] = SATU_T() - t_begin;
        }
    }
    if (!valid) return;
    float* recf = p.lrcat + ((long long)gy * p.w + gx) * REC;
    f32x4* rec = reinterpret_cast<f32x4*>(recf + half * 32 * NB);
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 a = {accA[t][4 * g], accA[t][4 * g + 1], accA[t][4 * g + 2], accA[t][4 * g + 3]};
            f32x4 b = {accB[t][4 * g], accB[t][4 * g + 1], accB[t][4 * g + 2], accB[t][4 * g + 3]};
            rec[t * 4 + g] = a;
            rec[4 * NB + t * 4 + g] = b;
        }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 c = {accC[4 * g], accC[4 * g + 1], accC[4 * g + 2], accC[4 * g + 3]};
        *reinterpret_cast<f32x4*>(recf + 64 * NB + 8 * g + 4 * half) = c;     // rows 8g + 4 half + {0..3} = C-stack channels
    }
}

// ------------------------------------------------------------------------------------------
// HR stage.  One wave = 32 consecutive HR pixels of one output row x all 32 NB output channels; two lanes
// (half 0 / half 1) per pixel, each owning 16 NB of them in MFMA accumulator order.
// grid_sample (zeros padding, align_corners=True) semantics of savsr_arch.py:262-295.
//
// NB = 2 is STAUpsample.forward itself: out = [64] planes (savsr_satu_hr_upsample).
// NB = 1 is the form the network runs (savsr_satu_hr_tail): every matrix of the stage -- fusion (:374), the expert
// expand (:358) and with them the LR-side projections -- is pre-multiplied by the 3x3 tail conv's weights
// (savsr_arch.py:738) regrouped as Wt27[p = 3 tap + o][c]: the tail conv is linear and the bilinear gather commutes
// with a channel contraction, so
//     P[p] = Wt27 (fusion output) = G(Wt27 Wa sta, soff) + G(Wt27 Wb x, off) + sum_n r_n (Wt27 Wb E_n)(sum_m r_m C_m G(x, off)) + Wt27 b
// needs 27 (of 32) output rows instead of 64, gathers 96-float records instead of 160-float ones, and the
// [64][H][W] feature map (236 MB at 720x1280, written here and re-read by the tail) never exists: the stage writes
// the 27 planes P, and tail_gather_kernel (tail.hip) adds the nine shifted taps per output channel.
//
// A workgroup (HR_WAVES waves) owns an HR tile of TY rows x 32*TXW columns and first stages into LDS
//   * the LRcat records its taps can touch (tile footprint + the offset range of the phase table), with a record
//     pitch of REC + 4 floats (conflict-free b128 reads), one LDS-DMA per record;
//   * the phase-table entries of ITS rows x columns ([TY][32 TXW][8], offsets normalised once here), so the tile
//     loop never looks anything up in global memory, whatever the size of the table (4 entries at x2, 24 360 at
//     x3.9 for 180x320: the product table is indexed by (idx_h[Y], idx_w[X]) only during this staging);
//   * gyn of its rows, gxn of its columns, the expert-MFMA A operands and the bias.
// A wave whose 8 taps all fall inside the staged window gathers from LDS (256 B/clk/CU); any other wave -- and every
// wave when the caller passes lrh == 0 -- gathers the same records from global memory, so correctness never depends
// on the window the caller chose.
// ------------------------------------------------------------------------------------------
struct HrParams {
    savsr_satu_weights wt;
    const float* lrcat;
    int h, w;
    const float* table;
    int n_uw;
    const int* idx_h;
    const int* idx_w;
    const float* gyn;
    const float* gxn;
    int H, W;
    float* out;
    long long out_plane;         // floats between output channel planes (>= H*W)
    int ty, txw, lrh, lrw;       // HR tile rows, 32-px column tiles per workgroup, staged LR window
    float omin_x, omin_y;        // lower bound of the sampling offsets (window origin)
    float step_x, step_y;        // LR pixels per HR pixel (1 / scale), for the window origin only
};

constexpr int HR_MAX_ROWS = 64;
__host__ __device__ constexpr int hr_lds_rec(int nb) { return rec_floats(nb) + 4; }       // floats per staged record
__host__ __device__ constexpr int hr_wimg_floats(int nb) { return nb * 2 * 2 * 64 * 4; }  // (Wb E) image [t][ks][part][lane][8 bf16]
// floats of LDS behind the window: image | bias [half][16 NB] (padded to 64) | table [ty][32 txw][8] | gyn [64] | gxn [32 txw] | idx_h [64] | idx_w [32 txw]
__host__ __device__ inline int hr_const_floats(int nb, int ty, int txw) { return hr_wimg_floats(nb) + 64 + ty * 32 * txw * 8 + 2 * (HR_MAX_ROWS + 32 * txw); }

struct Taps {
    int ty[4], tx[4];  // LR coordinates of the 4 taps (nw, ne, sw, se), clamped into the image
    float wgt[4];      // bilinear weights, 0 for taps outside the image
};

// onx, ony: the sampling offset already normalised as the reference does, (off * 2) / (size - 1)  (:285-287)
__device__ __forceinline__ Taps make_taps(float gxn, float gyn, float onx, float ony, int h, int w) {
    const float fw1 = (float)(w - 1), fh1 = (float)(h - 1);
    const float gx = gxn + onx;
    const float gy = gyn + ony;
    float ix = ((gx + 1.f) / 2.f) * fw1;                 // grid_sampler_unnormalize, align_corners=True
    float iy = ((gy + 1.f) / 2.f) * fh1;
    ix = fminf(fmaxf(ix, -2.f), (float)w + 1.f);         // keeps every in-range tap intact
    iy = fminf(fmaxf(iy, -2.f), (float)h + 1.f);
    const float xw = floorf(ix), yn = floorf(iy);
    const float lx = ix - xw, ly = iy - yn;
    const float ex = 1.f - lx, sy = 1.f - ly;
    const int x0 = (int)xw, y0 = (int)yn, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w;
    const bool vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
    // an out-of-image tap contributes 0 (zeros padding); park it on its in-image neighbour so
    // that it never widens the window a wave needs
    const int cx0 = vx0 ? x0 : (vx1 ? x1 : (x0 < 0 ? 0 : w - 1)), cx1 = vx1 ? x1 : cx0;
    const int cy0 = vy0 ? y0 : (vy1 ? y1 : (y0 < 0 ? 0 : h - 1)), cy1 = vy1 ? y1 : cy0;
    Taps t;
    t.ty[0] = cy0; t.tx[0] = cx0; t.wgt[0] = (vy0 && vx0) ? sy * ex : 0.f;
    t.ty[1] = cy0; t.tx[1] = cx1; t.wgt[1] = (vy0 && vx1) ? sy * lx : 0.f;
    t.ty[2] = cy1; t.tx[2] = cx0; t.wgt[2] = (vy1 && vx0) ? ly * ex : 0.f;
    t.ty[3] = cy1; t.tx[3] = cx1; t.wgt[3] = (vy1 && vx1) ? ly * lx : 0.f;
    return t;
}

// acc[4g .. 4g+3] += w * v as two v_pk_fma_f32 (explicit 2-vectors: the SLP vectoriser packs only about half of these)
__device__ __forceinline__ void fma_quad(f32x16& acc, int g, float w, const f32x4& v) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 w2 = {w, w};
    const f32x2 a01 = __builtin_elementwise_fma(w2, f32x2{v[0], v[1]}, f32x2{acc[4 * g], acc[4 * g + 1]});
    const f32x2 a23 = __builtin_elementwise_fma(w2, f32x2{v[2], v[3]}, f32x2{acc[4 * g + 2], acc[4 * g + 3]});
    acc[4 * g] = a01[0]; acc[4 * g + 1] = a01[1]; acc[4 * g + 2] = a23[0]; acc[4 * g + 3] = a23[1];
}

template <bool FROM_LDS, int NB>
__device__ __forceinline__ void hr_tile(const HrParams& p, const float* lds, int ly0, int lx0, const Taps& to, const Taps& ts,
                                        const f32x4 rr, int half, int lane, bool valid, unsigned o_off, const float* cst) {
    constexpr int REC = rec_floats(NB), LREC = hr_lds_rec(NB);
    auto rec_of = [&](int ty, int tx) -> const f32x4* {
        // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): window coordinates are tiny
        if (FROM_LDS) return reinterpret_cast<const f32x4*>(lds + __mul24(__mul24(ty - ly0, p.lrw) + (tx - lx0), LREC));
        return reinterpret_cast<const f32x4*>(p.lrcat + ((long long)ty * p.w + tx) * REC);
    };
    const f32x4* ro[4] = {rec_of(to.ty[0], to.tx[0]), rec_of(to.ty[1], to.tx[1]), rec_of(to.ty[2], to.tx[2]), rec_of(to.ty[3], to.tx[3])};
    const f32x4* rs[4] = {rec_of(ts.ty[0], ts.tx[0]), rec_of(ts.ty[1], ts.tx[1]), rec_of(ts.ty[2], ts.tx[2]), rec_of(ts.ty[3], ts.tx[3])};
    // ---- the 32 compressed channels G(C x, off); t_j = sum_m r_m (C_m f0)_j -------------------------
    // Both lanes of a pixel need all 8 t_j: each computes 4 of them (j = 4 half .. 4 half + 3: half the LDS reads
    // and FMAs of this part) and the halves are exchanged with v_permlane32_swap.
    // explicit 2-vectors: one v_pk_fma_f32 per pair (left to the SLP vectoriser this loop became pk_mul + v_mov shuffles +
    // scalar adds, ~20 instructions per 8 MACs, and the kernel is VALU-issue-bound)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 t01 = {0.f, 0.f}, t23 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wk = to.wgt[k];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 v = ro[k][16 * NB + 2 * m + half];
            const float wr = wk * rr[m];
            const f32x2 w2 = {wr, wr}, v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
            t01 = __builtin_elementwise_fma(w2, v01, t01);
            t23 = __builtin_elementwise_fma(w2, v23, t23);
        }
    }
    const float tjh[4] = {t01[0], t01[1], t23[0], t23[1]};
    float tj[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned own = __float_as_uint(tjh[j]);
        const auto sw = __builtin_amdgcn_permlane32_swap(own, own, false, false);   // [0]: upper lanes get the lower half's value; [1]: lower lanes get the upper's
        const float other = __uint_as_float(half ? sw[0] : sw[1]);
        tj[j] = half ? other : tjh[j];
        tj[4 + j] = half ? tjh[j] : other;
    }
    // ---- B operand of the expert MFMA: v[(n, j)] = r_n t_j, k = 16 ks + 8 half + j <-> n = 2 ks + half ----
    bf16x8 bh[2], bl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const float rn = half ? rr[2 * ks + 1] : rr[2 * ks];
        const f32x4 v0 = {rn * tj[0], rn * tj[1], rn * tj[2], rn * tj[3]};
        const f32x4 v1 = {rn * tj[4], rn * tj[5], rn * tj[6], rn * tj[7]};
        split8v(v0, v1, bh[ks], bl[ks]);
    }
    const bf16x8* wimg = reinterpret_cast<const bf16x8*>(cst) + lane;               // LDS copy of [t][ks][part][lane]
    const f32x4* fb4 = reinterpret_cast<const f32x4*>(cst + hr_wimg_floats(NB) + half * 16 * NB);   // LDS copy of the bias, packed [half][q]
    const long long HW = p.out_plane;
    // The plane bases of a tile's stores are rebuilt from this pointer with scalar adds for every tile.  Left
    // loop-invariant, hipcc keeps all of them (64 SGPRs) across the tile loop, spills them to VGPR lanes and pays two
    // v_readlane + one 64-bit VALU add per store in a VALU-issue-bound kernel.
    float* outp = p.out;
    asm volatile("" : "+s"(outp));
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = fb4[t * 4 + g];
            acc[4 * g] = b[0]; acc[4 * g + 1] = b[1]; acc[4 * g + 2] = b[2]; acc[4 * g + 3] = b[3];
        }
        // G(Wb x, off): channels q = 16 t .. 16 t + 15 of this half
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float wk = to.wgt[k];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = ro[k][8 * NB * half + 4 * NB + 4 * t + g];
                fma_quad(acc, g, wk, v);
            }
        }
        // + (Wb E) v on the bf16 matrix cores (split operands)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            acc = mma3(wimg[((t * 2 + ks) * 2 + 0) * 64], wimg[((t * 2 + ks) * 2 + 1) * 64], bh[ks], bl[ks], acc);
        // + G(Wa sta, soff)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float wk = ts.wgt[k];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = rs[k][8 * NB * half + 4 * t + g];
                fma_quad(acc, g, wk, v);
            }
        }
        // stores: uniform plane base (scalar arithmetic) + one per-lane 32-bit byte offset, no 64-bit VALU add per store.
        // NB == 1: rows 27 .. 31 of the block are padding (27 = 9 taps x 3 colours): registers 0 .. 11 hold live rows in both
        // lane halves (rows r', r' + 4), registers 12 .. 14 only in half 0 (rows 24, 25, 26), register 15 in neither.
        constexpr int R_BOTH = NB == 1 ? 12 : 16, R_LOW = NB == 1 ? 15 : 16;
        if (valid) {
            unsigned oo = o_off;
            asm volatile("" : "+v"(oo));       // the 32 -> 64-bit extension must sit in THIS block for the (scalar base, 32-bit lane offset) store form to be selected
#pragma unroll
            for (int r = 0; r < R_BOTH; ++r) {
                float* pl = outp + (long long)(32 * t + acc_row(r, 0)) * HW;
                asm volatile("" : "+s"(pl));     // (opaque, or hipcc re-associates to (outp + lane offset) + plane: a 64-bit VALU add per store)
                *(__attribute__((address_space(1))) float*)((__attribute__((address_space(1))) char*)pl + oo) = acc[r];
            }
            if (R_LOW > R_BOTH && half == 0) {
#pragma unroll
                for (int r = R_BOTH; r < R_LOW; ++r) {
                    float* pl = outp + (long long)(32 * t + acc_row(r, 0)) * HW;
                    asm volatile("" : "+s"(pl));
                    *(__attribute__((address_space(1))) float*)((__attribute__((address_space(1))) char*)pl + oo) = acc[r];
                }
            }
        }
    }
}

#ifndef HR_OCC_TAIL
#define HR_OCC_TAIL 3             // resident workgroups per CU the tail-projected kernel is compiled for (146 VGPRs: 3 waves per SIMD fit 168)
#endif
constexpr int HR_WAVES = 4;       // waves per workgroup (the kernel needs ~180-250 VGPRs: 2 waves per SIMD; at 128 it spills and runs 3.6x slower)
// DIAG = the instrumented build (section stamps, no-store experiment); the product launch uses DIAG = false: kept as a
// run-time switch the stamp accumulators cost ~30 vector instructions per tile in a VALU-issue-bound kernel.
template <bool DIAG, int NB>
__global__ __launch_bounds__(64 * HR_WAVES, NB == 1 ? HR_OCC_TAIL : 2) void satu_hr_kernel(const HrParams p) {
    constexpr int REC = rec_floats(NB), LREC = hr_lds_rec(NB);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, px = lane & 31;
    const int X0 = blockIdx.x * 32 * p.txw, Y0 = blockIdx.y * p.ty;
    const long long t_entry = DIAG ? SATU_T() : 0;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);      // tile walk and row / column tests as scalar code
    const int ncol = 32 * p.txw;

    // ---- stage the LRcat window of this tile, the expert-MFMA A operands, the bias and the tile's table entries -------
    float* cst = lds + p.lrh * p.lrw * LREC;
    float* tab = cst + hr_wimg_floats(NB) + 64;                   // [ty][ncol][8]
    float* rowg = tab + p.ty * ncol * 8;                          // [HR_MAX_ROWS] gyn of the tile's rows
    float* colg = rowg + HR_MAX_ROWS;                             // [ncol] gxn of the tile's columns
    int* rowi = reinterpret_cast<int*>(colg + ncol);              // [HR_MAX_ROWS] table row of the tile's rows
    int* coli = rowi + HR_MAX_ROWS;                               // [ncol] table column of the tile's columns
    // (1) the per-row / per-column lookups: issued first, they land under the window's DMA issue
    const int Yr = Y0 + tid < p.H ? Y0 + tid : p.H - 1;           // (threads >= ty load a valid address and drop the value)
    const int Xr0 = X0 + tid < p.W ? X0 + tid : p.W - 1;
    const int ih_v = p.idx_h[Yr];
    const float gy_v = p.gyn[Yr];
    const int iw_v = p.idx_w[Xr0];
    const float gx_v = p.gxn[Xr0];
    // (2) the window.  Its origin is the tile's base sampling coordinate + the lower bound of the offsets, evaluated from
    // kernel arguments only (no load in front of the DMA issue); being a plan, it needs no bit-exactness (a wave whose
    // taps fall outside gathers from global memory).
    int ly0 = 0, lx0 = 0;
    if (p.lrh > 0) {
        const float by = ((float)Y0 + 0.5f) * p.step_y - 0.5f + p.omin_y - 0.01f;
        const float bx = ((float)X0 + 0.5f) * p.step_x - 0.5f + p.omin_x - 0.01f;
        ly0 = (int)floorf(fminf(fmaxf(by, 0.f), (float)(p.h - 1)));
        lx0 = (int)floorf(fminf(fmaxf(bx, 0.f), (float)(p.w - 1)));
        // one LDS-DMA per record: lanes 0 .. REC/4-1 move it straight into its padded slot (no registers, no
        // ds_write, every record of the wave in flight at once; the register-staged loop this replaces was a third
        // of the kernel).  Records outside the image are never read (taps are clamped into it) and stay unwritten.
        const int nrec = p.lrh * p.lrw;
        for (int r = wave; r < nrec; r += HR_WAVES) {
            const int ry = r / p.lrw, rx = r - ry * p.lrw;
            const int gy = ly0 + ry, gx = lx0 + rx;
            if (gy < p.h && gx < p.w && lane < REC / 4) {
                const float* src = p.lrcat + ((long long)gy * p.w + gx) * REC + 4 * lane;
                const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds + r * LREC));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
        }
    }
    if (tid < p.ty) { rowi[tid] = ih_v; rowg[tid] = gy_v; }
    if (tid < ncol) { coli[tid] = iw_v; colg[tid] = gx_v; }
    for (int e = tid + 64 * HR_WAVES; e < ncol; e += 64 * HR_WAVES) {           // (tiles wider than 256 columns)
        const int Xc = X0 + e < p.W ? X0 + e : p.W - 1;
        coli[e] = p.idx_w[Xc];
        colg[e] = p.gxn[Xc];
    }
    {
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.wt.wbe_w);
        for (int e = tid; e < hr_wimg_floats(NB) / 4; e += 64 * HR_WAVES) reinterpret_cast<f32x4*>(cst)[e] = wsrc[e];
        if (tid < 8 * NB) reinterpret_cast<f32x4*>(cst + hr_wimg_floats(NB))[tid] = reinterpret_cast<const f32x4*>(p.wt.fusion_b)[tid];
    }
    __syncthreads();
    {
        // (3) the tile's own slice of the phase table: entry (row, col) = table[idx_h[Y]][idx_w[X]]; the offset quad is normalised
        // here, once per workgroup, as the reference does per pixel ((off * 2) / (size - 1), :285-287).  Four independent
        // 16-B loads in flight per thread and round.
        const float fw1 = (float)(p.w - 1), fh1 = (float)(p.h - 1);
        const int nq = p.ty * ncol * 2;
        for (int e0 = tid; e0 < nq; e0 += 4 * 64 * HR_WAVES) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 64 * HR_WAVES;
                const int ec = e < nq ? e : tid;                  // (clamped: loaded, not stored)
                const int q = ec & 1, pe = ec >> 1;
                const int trow = pe / ncol, col = pe - trow * ncol;
                v[u] = *reinterpret_cast<const f32x4*>(p.table + ((long long)rowi[trow] * p.n_uw + coli[col]) * SAVSR_SATU_TABLE + 4 * q);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 64 * HR_WAVES;
                if (e < nq) {
                    f32x4 t = v[u];
                    if (e & 1) { t[0] = (t[0] * 2.f) / fw1; t[1] = (t[1] * 2.f) / fh1; t[2] = (t[2] * 2.f) / fw1; t[3] = (t[3] * 2.f) / fh1; }
                    *reinterpret_cast<f32x4*>(tab + (e >> 1) * 8 + 4 * (e & 1)) = t;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the window DMA (not counted by the compiler)
    __syncthreads();

    const int dbg_all = DIAG ? __builtin_amdgcn_readfirstlane(g_satu_stamps_on) : 0;
    const int stamps_on = dbg_all & 1;
    const bool dbg_nostore = dbg_all & 2;                            // timing experiment only: skip the output stores
    long long tacc[SSTAMP_N] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_prev = stamps_on ? SATU_T() : 0;
    const long long t_begin = t_entry;
    tacc[4] = t_prev - t_entry;                                      // staging of the LDS window + constants
#define HR_MARK(i) do { if (stamps_on) { const long long t_now = SATU_T(); tacc[i] += t_now - t_prev; t_prev = t_now; } } while (0)
    const int ntile = p.ty * p.txw;
    // No global load may be pending, as far as hipcc can tell, when the tile loop is entered or continued: its waits count only
    // what it can see on every path, so one conditional load in the loop (or one issued in front of it) turns into
    // s_waitcnt vmcnt(0..1) at the top of EVERY tile -- behind the previous tile's output stores, i.e. a full write
    // round trip per tile.  Everything a tile looks up is therefore in LDS.
    for (int T = wave_s; T < ntile; T += HR_WAVES) {
        const int trow = p.txw == 1 ? T : T / p.txw;                // (a run-time integer division is ~14 vector instructions)
        const int tcol = T - trow * p.txw;
        const int Y = Y0 + trow;
        const int Xb = X0 + tcol * 32;
        if (Y >= p.H || Xb >= p.W) continue;                      // wave-uniform
        const int X = Xb + px;
        const bool valid = X < p.W && !dbg_nostore;
        const float* te = tab + (trow * ncol + tcol * 32 + px) * 8;
        const f32x4 rr = *reinterpret_cast<const f32x4*>(te);
        const f32x4 oo = *reinterpret_cast<const f32x4*>(te + 4);
        const float gxn = colg[tcol * 32 + px];
        const float gyn = rowg[trow];
        if (stamps_on) { asm volatile("" :: "v"(rr[0]), "v"(oo[0])); }
        HR_MARK(0);                                                  // table lookup
        const Taps to = make_taps(gxn, gyn, oo[0], oo[1], p.h, p.w);
        const Taps ts = make_taps(gxn, gyn, oo[2], oo[3], p.h, p.w);
        if (stamps_on) { asm volatile("" :: "v"(to.wgt[3]), "v"(ts.wgt[3])); }
        HR_MARK(1);                                                  // tap arithmetic

        bool inside = p.lrh > 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            inside = inside && (unsigned)(to.ty[k] - ly0) < (unsigned)p.lrh && (unsigned)(to.tx[k] - lx0) < (unsigned)p.lrw;
            inside = inside && (unsigned)(ts.ty[k] - ly0) < (unsigned)p.lrh && (unsigned)(ts.tx[k] - lx0) < (unsigned)p.lrw;
        }
        // byte offset of this lane's pixel inside channel plane acc_row(r, 0); the half's +4 channels are folded in
        const unsigned o_off = 4u * (unsigned)(Y * p.W + X) + (half ? 16u * (unsigned)p.out_plane : 0u);
        if (__all(inside)) hr_tile<true, NB>(p, lds, ly0, lx0, to, ts, rr, half, lane, valid, o_off, cst);
        else {
            hr_tile<false, NB>(p, lds, ly0, lx0, to, ts, rr, half, lane, valid, o_off, cst);
            __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): the fallback's gathers are not left pending either (see above the loop)
        }
        HR_MARK(2);                                                  // gathers + MFMA + store issue
    }
    if (stamps_on) {
        __builtin_amdgcn_s_waitcnt(0);
        HR_MARK(3);                                                  // store drain
        if (tid == 0) {
            const int b = blockIdx.x + gridDim.x * blockIdx.y;
            if (b < SSTAMP_BLOCKS) {
                for (int i = 0; i < 7; ++i) g_satu_stamps[b * SSTAMP_N + i] = tacc[i];
                g_satu_stamps[b * SSTAMP_N + 7] = SATU_T() - t_begin;
            }
        }
    }
}

}  // namespace savsr

using namespace savsr;

static int g_satu_diag_host = 0;      // != 0: launch the instrumented kernels (tail-projected form only)

extern "C" int savsr_debug_satu_stamps(int enable) {
    g_satu_diag_host = enable;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_satu_stamps_on), &enable, sizeof(int));
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int savsr_satu_hr_occupancy_target(int tail_form) { return tail_form ? HR_OCC_TAIL : 2; }

// Diagnostics: resident workgroups per CU the runtime predicts for the HR / LR kernels (tail-projected form) with `lds_bytes` of dynamic LDS.
extern "C" int savsr_debug_satu_occupancy(int which, int lds_bytes) {
    int n = -1;
    hipError_t e = which == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, satu_hr_kernel<false, 1>, 64 * HR_WAVES, (size_t)lds_bytes)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, satu_lr_kernel<false, 1>, 512, (size_t)lds_bytes);
    return e == hipSuccess ? n : -(int)e;
}

extern "C" int savsr_debug_read_satu_stamps(long long* host, int nblocks) {
    if (!host || nblocks < 1 || nblocks > SSTAMP_BLOCKS) return fail_arg("debug_read_satu_stamps");
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_satu_stamps), sizeof(long long) * SSTAMP_N * nblocks);
    return e == hipSuccess ? 0 : (int)e;
}

static bool satu_weights_ok(const savsr_satu_weights* w) {
    return w && w->body0_w && w->body0_b && w->body2_w && w->body2_b && w->head_w && w->head_b && w->kconv_w && w->kconv_b &&
           w->proj_w && w->wbe_w && w->fusion_b;
}

extern "C" int savsr_satu_phase_table(const savsr_satu_weights* wt, const float* uniq_ch, int n_uh, const float* uniq_cw, int n_uw,
                                      float inv_sw, float inv_sh, float* table, void* stream) {
    if (!satu_weights_ok(wt) || !uniq_ch || !uniq_cw || !table) return fail_arg("satu_phase_table: null pointer");
    if (n_uh < 1 || n_uw < 1) return fail_arg("satu_phase_table: empty table");
    const long long n = (long long)n_uh * n_uw;
    hipLaunchKernelGGL(satu_phase_table_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), *wt,
                       uniq_ch, n_uh, uniq_cw, n_uw, inv_sw, inv_sh, table);
    return check_launch("satu_phase_table_kernel");
}

template <int NB>
static int lr_stage(const savsr_satu_weights* wt, const float* x, const float* st, int32_t pix, int32_t row_px, int h, int w, float* lrcat,
                    void* stream) {
    if (!satu_weights_ok(wt) || !x || !st || !lrcat) return fail_arg("satu_lr_stage: null pointer");
    if (h < 1 || w < 1 || row_px < w || pix < 64 || (pix & 3)) return fail_arg("satu_lr_stage: shape/strides");
    if ((reinterpret_cast<uintptr_t>(lrcat) | reinterpret_cast<uintptr_t>(wt->kconv_w) | reinterpret_cast<uintptr_t>(wt->kconv_b) |
         reinterpret_cast<uintptr_t>(wt->proj_w) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(st)) & 15) {
        set_error("satu_lr_stage: x / st / lrcat / kconv_w / kconv_b / proj_w must be 16-byte aligned");
        return SAVSR_E_ALIGN;
    }
    LrParams p;
    p.wt = *wt; p.x = x; p.st = st; p.pix = pix; p.row_px = row_px; p.h = h; p.w = w; p.lrcat = lrcat;
    constexpr size_t lds = LR_NPX * LR_XS * sizeof(float) + 2 * LR_PHASE * 16 + 25 * 64 * sizeof(float);     // 150.5 KB
    static_assert(lds <= 160 * 1024, "LR stage LDS budget");
    const bool diag = NB == 1 && g_satu_diag_host;
    const void* fn = diag ? reinterpret_cast<const void*>(&satu_lr_kernel<true, 1>) : reinterpret_cast<const void*>(&satu_lr_kernel<false, NB>);
    if (int rc = ensure_dynamic_lds(fn, (int)lds, "satu_lr_stage")) return rc;
    dim3 grid((w + LR_TW - 1) / LR_TW, (h + LR_TH - 1) / LR_TH);
    if (diag) hipLaunchKernelGGL((satu_lr_kernel<true, 1>), grid, dim3(512), lds, static_cast<hipStream_t>(stream), p);
    else hipLaunchKernelGGL((satu_lr_kernel<false, NB>), grid, dim3(512), lds, static_cast<hipStream_t>(stream), p);
    return check_launch("satu_lr_kernel");
}

extern "C" int savsr_satu_lr_stage(const savsr_satu_weights* wt, const float* x, const float* st, int32_t pix, int32_t row_px, int h,
                                   int w, float* lrcat, void* stream) {
    return lr_stage<2>(wt, x, st, pix, row_px, h, w, lrcat, stream);
}

extern "C" int savsr_satu_lr_stage_tail(const savsr_satu_weights* wt, const float* x, const float* st, int32_t pix, int32_t row_px, int h,
                                        int w, float* lrcat, void* stream) {
    return lr_stage<1>(wt, x, st, pix, row_px, h, w, lrcat, stream);
}

extern "C" int64_t savsr_satu_hr_lds_bytes(int tail_form, int tile_rows, int tile_cols32, int lr_rows, int lr_cols) {
    if (tile_rows < 1 || tile_rows > HR_MAX_ROWS || tile_cols32 < 1 || lr_rows < 0 || lr_cols < 0) return -1;
    const int nb = tail_form ? 1 : 2;
    return ((int64_t)lr_rows * lr_cols * hr_lds_rec(nb) + hr_const_floats(nb, tile_rows, tile_cols32)) * (int64_t)sizeof(float);
}

template <int NB>
static int hr_stage(const savsr_satu_weights* wt, const float* lrcat, int h, int w, const float* table, int n_uw, const int32_t* idx_h,
                    const int32_t* idx_w, const float* gyn, const float* gxn, int H, int W, const savsr_satu_tiling* tiling, float* out,
                    int64_t out_plane, void* stream) {
    if (!satu_weights_ok(wt) || !lrcat || !table || !idx_h || !idx_w || !gyn || !gxn || !out) return fail_arg("satu_hr: null pointer");
    if (h < 2 || w < 2 || H < 1 || W < 1 || n_uw < 1 || out_plane < (int64_t)H * W) return fail_arg("satu_hr: shape (h, w >= 2, out_plane >= H*W required)");
    if (out_plane * 32 * NB * 4 >= ((int64_t)1 << 32)) return fail_arg("satu_hr: output of 4 GiB or more is not supported (32-bit store offsets)");
    if ((reinterpret_cast<uintptr_t>(lrcat) | reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(wt->fusion_b) |
         reinterpret_cast<uintptr_t>(wt->wbe_w)) & 15) {
        set_error("satu_hr: lrcat / table / fusion_b / wbe_w must be 16-byte aligned");
        return SAVSR_E_ALIGN;
    }
    HrParams p;
    p.wt = *wt; p.lrcat = lrcat; p.h = h; p.w = w; p.table = table; p.n_uw = n_uw; p.idx_h = idx_h; p.idx_w = idx_w;
    p.gyn = gyn; p.gxn = gxn; p.H = H; p.W = W; p.out = out; p.out_plane = out_plane;
    p.ty = 8; p.txw = 1; p.lrh = 0; p.lrw = 0; p.omin_x = 0.f; p.omin_y = 0.f;      // default: no window staging, gathers from global
    p.step_x = (float)w / (float)W; p.step_y = (float)h / (float)H;
    if (tiling) {
        if (tiling->tile_rows < 1 || tiling->tile_rows > HR_MAX_ROWS || tiling->tile_cols32 < 1 || tiling->tile_cols32 > 8 || tiling->lr_rows < 0 ||
            tiling->lr_cols < 0)
            return fail_arg("satu_hr: tiling");
        p.ty = tiling->tile_rows; p.txw = tiling->tile_cols32; p.lrh = tiling->lr_rows; p.lrw = tiling->lr_cols;
        p.omin_x = tiling->off_min_x; p.omin_y = tiling->off_min_y;
        if (tiling->step_x > 0.f && tiling->step_y > 0.f) { p.step_x = tiling->step_x; p.step_y = tiling->step_y; }
        if (p.lrh == 0 || p.lrw == 0) { p.lrh = 0; p.lrw = 0; }
    }
    const size_t lds = ((size_t)p.lrh * p.lrw * hr_lds_rec(NB) + hr_const_floats(NB, p.ty, p.txw)) * sizeof(float);
    if (lds > 160 * 1024) return fail_arg("satu_hr: staged window + tile tables exceed 160 KiB of LDS");
    const bool diag = NB == 1 && g_satu_diag_host;
    const void* fn = diag ? reinterpret_cast<const void*>(&satu_hr_kernel<true, 1>) : reinterpret_cast<const void*>(&satu_hr_kernel<false, NB>);
    if (int rc = ensure_dynamic_lds(fn, 160 * 1024, "satu_hr")) return rc;
    dim3 grid((W + 32 * p.txw - 1) / (32 * p.txw), (H + p.ty - 1) / p.ty);
    if (diag) hipLaunchKernelGGL((satu_hr_kernel<true, 1>), grid, dim3(64 * HR_WAVES), lds, static_cast<hipStream_t>(stream), p);
    else hipLaunchKernelGGL((satu_hr_kernel<false, NB>), grid, dim3(64 * HR_WAVES), lds, static_cast<hipStream_t>(stream), p);
    return check_launch("satu_hr_kernel");
}

extern "C" int savsr_satu_hr_upsample(const savsr_satu_weights* wt, const float* lrcat, int h, int w, const float* table, int n_uw,
                                      const int32_t* idx_h, const int32_t* idx_w, const float* gyn, const float* gxn, int H, int W,
                                      const savsr_satu_tiling* tiling, float* out, int64_t out_plane, void* stream) {
    return hr_stage<2>(wt, lrcat, h, w, table, n_uw, idx_h, idx_w, gyn, gxn, H, W, tiling, out, out_plane, stream);
}

extern "C" int savsr_satu_hr_tail(const savsr_satu_weights* wt, const float* lrcat, int h, int w, const float* table, int n_uw,
                                  const int32_t* idx_h, const int32_t* idx_w, const float* gyn, const float* gxn, int H, int W,
                                  const savsr_satu_tiling* tiling, float* out, int64_t out_plane, void* stream) {
    return hr_stage<1>(wt, lrcat, h, w, table, n_uw, idx_h, idx_w, gyn, gxn, H, W, tiling, out, out_plane, stream);
}
