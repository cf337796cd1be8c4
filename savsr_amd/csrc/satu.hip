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

// Settled build-time choices (numbers in DESIGN.md section 4b; the switches themselves -- LR_EXP / LRS_EXP / QS_EXP timing knobs, LR_XPREFETCH, LR_ST,
// HR_ST, LRS_LRELU, HR_SCALAR_FMA, HR_LANE_PX -- are archived as tools/experiments/satu_switches.patch):
//  * LR stage: the second channel group's x tile is loaded under phase 4 of the first; plain LRcat stores; LeakyReLU_0.1(k) * x accumulated as two
//    independent FMAs (0.55 sum x k + 0.45 sum x |k|);
//  * HR stage (tail-projected forms): lane = pixel wave tiles, packed FMAs in the gathers, write-through (`sc1`) plane stores.

// LR stage, tail form: 1 = the records leave through a wave-private LDS transpose as whole 128-B lines, non-temporal (round 5: LR 42.4-43.0 ->
// 39.4-39.5 us on one lease; plain / sc1 / sc0 sc1 stores of the same lines 40.3-40.8 / 39.2-39.7 / 39.5-39.7); 0 = straight from the
// accumulators, 16-B pieces of 64 different lines per store instruction (rounds 1-4; the standalone 160-float form always)
#ifndef LR_COALESCED_STORES
#define LR_COALESCED_STORES 1
#endif

namespace savsr {

// Floats per LRcat record for NB 32-row output blocks per projection: (A | B) per lane half, then the 32 compressed channels.
// NB = 2: the standalone SATU (64 fused channels, SAVSR_SATU_LRCAT = 160); NB = 1: the tail-projected form (27 of 32 rows
// used, SAVSR_SATU_LRCAT_TAIL = 96), see the header comment of the HR stage.
__host__ __device__ constexpr int rec_floats(int nb) { return 64 * nb + 32; }
static_assert(rec_floats(2) == SAVSR_SATU_LRCAT && rec_floats(1) == SAVSR_SATU_LRCAT_TAIL, "record sizes");

// Diagnostics (never used by the product path): accumulated s_memtime deltas of kernel sections, written by
// wave 0 of each workgroup when enabled with savsr_debug_satu_stamps(1).
[[maybe_unused]] constexpr int SSTAMP_BLOCKS = 2048, SSTAMP_N = 8;
#if defined(SAVSR_DIAG)
#define SATU_HAS_STAMPS 1
__device__ long long g_satu_stamps[SSTAMP_BLOCKS * SSTAMP_N];
__device__ int g_satu_stamps_on = 0;
#else
#define SATU_HAS_STAMPS 0         // the product library: no diagnostic state at all (DIAG kernels are never instantiated)
#endif
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

// LDS-DMA with a SCALAR 64-bit base + a 32-bit per-lane byte offset (no per-piece 64-bit VGPR address: the LR kernel has no register to spare --
// 254 -> 243 VGPRs, and the standalone form no longer spills).
__device__ __forceinline__ void glds16_sbase(const void* sbase, unsigned voff, const void* lds_dst) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst);
    const unsigned long long sb = (unsigned long long)(uintptr_t)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb), hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"((((unsigned long long)hi) << 32) | lo), "s"(dst) : "memory");
}

// ------------------------------------------------------------------------------------------
// LR stage, STREAMING form (round 3).  Same arithmetic, same tile, same LDS image as the round-2 kernel (satu_lr_kernel, archived as
// tools/experiments/satu_lr_round2_kernel.patch) -- what changed is where the per-phase barrier sits.  There, a phase was [5 DMAs | first fragment
// reads | 20 MFMA groups | last tap's LeakyReLU * x | vmcnt(0) | barrier]: both waves of a SIMD leave the barrier in
// lockstep, so every phase boundary drains the matrix pipe for ~1 k cycles (MFMA latency + 48 vector instructions + barrier
// skew + DMA issue + fragment latency) of a ~5.8 k-cycle phase.  Here the MFMA stream never stops inside a channel group:
//   * fragment reads run one tap (4 groups) ahead ACROSS the phase boundary: tap 4 of phase ph is read during tap 3, tap 0 of
//     phase ph + 1 during tap 4 of phase ph;
//   * so the one barrier B(ph) of a phase sits between tap 3 and tap 4 (group 16): in front of it every wave has issued its
//     last fragment read of buffer ph & 1 (and __syncthreads drains its LDS queue), behind it every read goes to the other
//     buffer, which is complete (its DMAs were issued right behind B(ph - 1), a whole phase earlier, and every wave waited for
//     its own pieces in front of B(ph));
//   * the slabs of phase ph + 2 are DMA'd into the buffer B(ph) has just freed, one piece per group of tap 4;
//   * tap 4 accumulates into a third accumulator whose LeakyReLU * x runs under tap 0 of the next phase (its first reader is a
//     compiler-generated instruction as in the other taps); it is drained once per channel group, where the x tile changes.
// ------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(512, 2) void satu_lr_stream_kernel(const LrParams p) {
    constexpr int REC = rec_floats(NB);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xt = smem;                                                   // [432][36]: replicate-padded x tile of one channel group
    bf16x8* wbuf = reinterpret_cast<bf16x8*>(smem + LR_NPX * LR_XS);    // [2][LR_PHASE]: weight slabs of a kernel row, double buffered
    float* kbl = smem + LR_NPX * LR_XS + 2 * LR_PHASE * 4;              // [25][64] kernel_conv bias (+ 64 floats of slack: the bias of "phase 10")

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, px = lane & 31;
    const int x0 = blockIdx.x * LR_TW, y0 = blockIdx.y * LR_TH;
    const int gy = y0 + wave, gx = x0 + px;
    const bool valid = gy < p.h && gx < p.w;
    const int cy = gy < p.h ? gy : p.h - 1, cx = gx < p.w ? gx : p.w - 1;
    const long long cpix = ((long long)cy * p.row_px + cx) * p.pix + 8 * half;

#if SATU_HAS_STAMPS
    const long long ts0 = SATU_T(), rt0 = (long long)__builtin_amdgcn_s_memrealtime();
    long long tsk[5] = {0, 0, 0, 0, 0};
#define LRS_MARK(i) do { tsk[i] = SATU_T() - ts0; } while (0)
#else
#define LRS_MARK(i) do { } while (0)
#endif
    const bf16x8* kw = reinterpret_cast<const bf16x8*>(p.wt.kconv_w);
    // piece i (of 5 per wave) of the slabs of linear phase q (= channel group q / 5, kernel row q % 5) -> LDS buffer b;
    // q == 10: the projection image ((2 NB + 1) x 8 KB, consumed after the last phase).  BRANCH-FREE (a branch here cuts the
    // straight-line MFMA groups of tap 4 into basic blocks the scheduler cannot interleave): the LDS destination has the same
    // form for slabs and projection pieces (LR_SLAB = 8 x 64 units), and where there is nothing to fetch (q == 10 beyond the
    // image, q == 11) the piece re-fetches slab bytes of phase 9 into a buffer nobody reads any more.
    const unsigned dma_voff = (unsigned)((wave * 64 + lane) * 16);     // this lane's 16 B inside a 8-KiB slab / projection piece
    auto dma_piece = [&](int q, int b, int i) {
        const int qs = q < 10 ? q : 9;
        const int cg = qs >= 5 ? 1 : 0, ky = qs - 5 * cg;
        const bf16x8* src = kw + (long long)((ky * 5 + i) * 2 + cg) * LR_SLAB;            // (wave-uniform: scalar arithmetic)
        const bf16x8* prj = reinterpret_cast<const bf16x8*>(p.wt.proj_w) + i * 8 * 64;
        if (i < 2 * NB + 1) src = q == 10 ? prj : src;               // (a select; i is a compile-time constant at every call site)
        glds16_sbase(src, dma_voff, wbuf + b * LR_PHASE + i * LR_SLAB + wave * 64);
    };
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
    auto xt_store = [&]() {
#pragma unroll
        for (int i = 0; i < XT_IT; ++i) {
            const int e = tid + i * 512;
            if (e < LR_NPX * 8) *reinterpret_cast<f32x4*>(xt + (e >> 3) * LR_XS + 4 * (e & 7)) = xv[i];
        }
    };

    // ---- prologue: what the first MFMAs need -- the slabs of phase 0, the st fragments, the bias -- is waited for here; the
    // x tile (a third of the prologue's bytes) is only read by the LeakyReLU * x work, one tap behind the MFMAs: its loads fly
    // under tap 0 of phase 0 and it is published by an extra barrier in front of that phase's group 3.
#pragma unroll
    for (int i = 0; i < 5; ++i) dma_piece(0, 0, i);
    bf16x8 sth[4], stl[4];
    {
        f32x4 sv[8];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4* g = reinterpret_cast<const f32x4*>(p.st + cpix + 16 * ks);
            sv[2 * ks] = g[0];
            sv[2 * ks + 1] = g[1];
        }
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (tid < 25 * 64 / 4) bv = reinterpret_cast<const f32x4*>(p.wt.kconv_b)[tid];
        xt_load(0);                                                   // (younger than everything the first barrier waits for)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) split8v(sv[2 * ks], sv[2 * ks + 1], sth[ks], stl[ks]);
        if (tid < 25 * 64 / 4) reinterpret_cast<f32x4*>(kbl)[tid] = bv;
        if (tid < 16) reinterpret_cast<f32x4*>(kbl + 25 * 64)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(XT_IT) : "memory");     // all but the x tile's loads: this wave's pieces of phase 0 have landed
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 5; ++i) dma_piece(1, 1, i);                   // lands under phase 0 (waited for in front of B(0))
    LRS_MARK(0);

    struct AFrag { bf16x8 ah[4], al[4]; };
    auto bias_ptr = [&](int q, int kx) -> const float* {            // kernel_conv bias of tap (q % 5, kx), channel group q / 5, this lane half
        const int cg = q >= 10 ? 2 : (q >= 5 ? 1 : 0), ky = q - 5 * cg;   // q == 10 (behind the last phase): rows 0 .. 4 at column 64+ = the next rows / the zero slack; never kept
        return kbl + (ky * 5 + kx) * 64 + cg * 32 + 4 * half;
    };
    f32x16 sta[2];
    f32x4 xc[8];                                                     // the centre pixel's x (loaded at the top of the last phase)
    f32x16 acc[2], acc2;                                             // taps 0 / 2 | 1 / 3 | 4 (consumed under tap 0 of the next phase)
    AFrag fr;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    {
        const bf16x8* wl0 = wbuf + lane;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { fr.ah[ks] = wl0[(ks * 2 + 0) * 64]; fr.al[ks] = wl0[(ks * 2 + 1) * 64]; }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_ptr(0, 0) + 8 * g);
            acc[0][4 * g] = b4[0]; acc[0][4 * g + 1] = b4[1]; acc[0][4 * g + 2] = b4[2]; acc[0][4 * g + 3] = b4[3];
        }
    }
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) {                  // unrolled: sta[cg] must stay in registers
        f32x16 sacc;
        f32x16 sabs;                                                 // sum x |k| (sacc holds sum x k)
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; sabs[r] = 0.f; }
        auto x_read = [&](int ky, int kx, int g) -> f32x4 {          // x_pad at tap (ky, kx), channel quad g of this half
            return *reinterpret_cast<const f32x4*>(xt + ((wave + ky) * LR_XC + px + kx) * LR_XS + 4 * half + 8 * g);
        };
        auto lrelu_x = [&](int g, const f32x16& a, const f32x4 xq) {   // sacc += LeakyReLU_0.1(K) * x_pad   (:228, :297-313)
            // LeakyReLU_0.1(k) = 0.55 k + 0.45 |k|: two INDEPENDENT accumulations per element (sum x k, sum x |k|; combined once per
            // channel group) instead of the dependent mul -> max -> fmac chain.  The first reader of the MFMA result is the
            // compiler's fma (it inserts the wait states an accumulator read needs); the |k| source modifier needs the VOP3 form.
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float k = a[4 * g + q];
                sacc[4 * g + q] = __builtin_fmaf(k, xq[q], sacc[4 * g + q]);
                float sv = sabs[4 * g + q];
                asm volatile("v_fma_f32 %0, |%1|, %2, %0" : "+v"(sv) : "v"(k), "v"(xq[q]));
                sabs[4 * g + q] = sv;
            }
        };
        // carried across the phases: the LDS operands of the NEXT group's vector work (read one group ahead of their use)
        f32x4 x_pf = {0.f, 0.f, 0.f, 0.f};                            // (multiplies the zero accumulator in the first phase of a channel group)
        f32x4 b_pf = *reinterpret_cast<const f32x4*>(bias_ptr(cg * 5, 1));
#pragma unroll 1
        for (int ky = 0; ky < 5; ++ky) {
            const int ph = cg * 5 + ky, buf = ph & 1;
            if (ph == 4) xt_load(1);   // the second channel group's x tile: its loads fly under this phase (28 registers; written to LDS at the group boundary)
            const bf16x8* wl = wbuf + buf * LR_PHASE + lane;
            const bf16x8* wn = wbuf + (buf ^ 1) * LR_PHASE + lane;
            // x operand of the deferred tap (ky - 1, 4); in the first phase of a channel group the accumulator it multiplies is zero
            // and the operand comes from the zero slack row behind the bias (in phase 0 the x tile is not in LDS yet)
            const float* xdef = ky > 0 ? xt + ((wave + ky - 1) * LR_XC + px + 4) * LR_XS + 4 * half : kbl + 25 * 64;
            if (cg == 1 && ky == 4) {                  // the centre pixel's x (B operand of the Wb / C projections): loaded a phase ahead of its use
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const f32x4* g = reinterpret_cast<const f32x4*>(p.x + cpix + 16 * ks);
                    xc[2 * ks] = g[0];
                    xc[2 * ks + 1] = g[1];
                }
            }
#pragma unroll
            for (int G = 0; G < 20; ++G) {
                const int kx = G / 4, ks = G % 4;
                __builtin_amdgcn_sched_barrier(0);
                if (cg == 0 && G == 3 && ky == 0) {     // phase 0 only: the x tile (its loads flew under the first groups) -> LDS, published
                    xt_store();
                    __syncthreads();
                }
                if (G == 16) {
                    // B(ph): this wave's pieces of phase ph + 1 have landed; behind the barrier everybody's have, and nobody reads
                    // buffer `buf` any more (tap 4's fragments were read during tap 3)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
                if (G >= 16) {       // the slabs of phase ph + 2 into the freed buffer, one piece per group (two with the last)
                    dma_piece(ph + 2, buf, G - 16);
                    if (G == 19) dma_piece(ph + 2, buf, 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kx == 4) acc2 = mma3(fr.ah[ks], fr.al[ks], sth[ks], stl[ks], acc2);
                else acc[kx & 1] = mma3(fr.ah[ks], fr.al[ks], sth[ks], stl[ks], acc[kx & 1]);
                // this k-step's fragments of the NEXT tap (tap 0 of the next phase from the other buffer: complete since B(ph))
                if (kx < 4) { fr.ah[ks] = wl[(kx + 1) * LR_SLAB + (ks * 2 + 0) * 64]; fr.al[ks] = wl[(kx + 1) * LR_SLAB + (ks * 2 + 1) * 64]; }
                else { fr.ah[ks] = wn[(ks * 2 + 0) * 64]; fr.al[ks] = wn[(ks * 2 + 1) * 64]; }
                // LeakyReLU * x of the PREVIOUS tap (tap 4 of the previous phase under tap 0)
                if (kx == 0) lrelu_x(ks, acc2, x_pf);
                else lrelu_x(ks, acc[(kx - 1) & 1], x_pf);
                // the bias (initial accumulator) of the NEXT tap, quad ks: its previous contents were consumed a tap ago
                {
                    f32x16& an = kx == 3 ? acc2 : (kx == 4 ? acc[0] : acc[(kx + 1) & 1]);       // (tap 0 of the next phase: acc[0])
                    an[4 * ks] = b_pf[0]; an[4 * ks + 1] = b_pf[1]; an[4 * ks + 2] = b_pf[2]; an[4 * ks + 3] = b_pf[3];
                }
                // LDS operands of the next group's vector work
                {
                    const int G1 = G + 1, kx1 = (G1 % 20) / 4, ks1 = G1 % 4;
                    if (G1 < 20) {
                        x_pf = kx1 > 0 ? x_read(ky, kx1 - 1, ks1) : *reinterpret_cast<const f32x4*>(xdef + 8 * ks1);
                        b_pf = *reinterpret_cast<const f32x4*>((kx1 < 4 ? bias_ptr(ph, kx1 + 1) : bias_ptr(ph + 1, 0)) + 8 * ks1);
                    } else {                                // group 0 of the next phase: the deferred tap (ky, 4) and the bias of its tap 1
                        x_pf = x_read(ky, 4, 0);
                        b_pf = *reinterpret_cast<const f32x4*>(bias_ptr(ph + 1, 1));
                    }
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                }
            }
        }
        LRS_MARK(cg == 0 ? 1 : 3);
        // ---- end of the channel group: drain the deferred tap (4, 4); the x tile changes here ----
        {
            f32x4 xq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) xq[g] = x_read(4, 4, g);
#pragma unroll
            for (int g = 0; g < 4; ++g) lrelu_x(g, acc2, xq[g]);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.55f * sacc[r] + 0.45f * sabs[r];
        sta[cg] = sacc;
        if (cg == 0) {
            // (the x loads are older than the five pieces of phase 6 issued behind B(4): VMEM returns in order)
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            __syncthreads();                           // every wave is done with the old x tile
            xt_store();
            __syncthreads();
            LRS_MARK(2);
        }
    }

    // ---- LR-side projections (bf16x3): image in LDS buffer 0 (DMA'd behind B(8), published by B(9)) ----
    // No LDS-DMA of this wave may be in flight when the workgroup's LDS is released: the last phase's (dummy) pieces were issued
    // ~4 groups ago; this also covers the centre-pixel loads the projections consume next.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bf16x8* pa = wbuf + lane;
    const bf16x8* pb = pa + NB * 4 * 2 * 64;
    const bf16x8* pc = pb + NB * 4 * 2 * 64;
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
#if SATU_HAS_STAMPS
    asm volatile("" :: "v"(accC[0]), "v"(accA[0][0]), "v"(accB[0][0]));
    LRS_MARK(4);
    if ((tid & 63) == 0 && (wave == 0 || wave == 4) && __builtin_amdgcn_readfirstlane(g_satu_stamps_on)) {    // waves 0 and 4 of every workgroup: [blk][2][8]
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        if (2 * b + 1 < SSTAMP_BLOCKS) {
            long long* o = g_satu_stamps + (2 * b + (wave >> 2)) * SSTAMP_N;
            for (int i = 0; i < 5; ++i) o[i] = tsk[i];
            o[5] = rt0;
            o[6] = (long long)__builtin_amdgcn_s_memrealtime();
            o[7] = SATU_T() - ts0;
        }
    }
#endif
#if LR_COALESCED_STORES
    if constexpr (NB == 1) {
        // ---- LRcat write-back, COALESCED (round 5).  A lane holds 12 of the 24 quads of ITS pixel's 384-B record; stored straight
        // from the accumulators every store instruction touched 64 different 128-B lines with 16 B each (768 partial-line writes per
        // wave and tile, nothing a write-through policy can stream).  The 32 records of a wave's row are one contiguous 12-KB run of
        // LRcat, so they go through a wave-private LDS region (record pitch 100 floats: conflict-free b128 in both directions) and
        // leave as twelve 1-KiB stores of eight whole lines each.
        // LDS that is free here: the x tile (every wave read it for the last time in the drain above) and weight buffer 1 (last read in
        // phase 9; its last -- dummy -- DMAs were waited for above); buffer 0 still holds the projection image ((2 NB + 1) x 8 KB).
        __syncthreads();
        // (every index below is rebuilt from an opaque copy of the thread id: values kept live from the top of the kernel through the phase
        // loops cost registers the loops do not have -- 256 allocated -- and the first version of this block spilled one of the centre-pixel
        // loads of the last phase behind an s_waitcnt vmcnt(1))
        unsigned t2 = threadIdx.x;
        asm volatile("" : "+v"(t2));
        const int ln = (int)(t2 & 63u), pxe = ln & 31, hfe = ln >> 5;
        const int wv = __builtin_amdgcn_readfirstlane((int)(t2 >> 6));
        constexpr int SP = 100;                                       // staged record pitch (floats)
        float* stg = wv < 4 ? smem + wv * (32 * SP)
                   : (wv < 7 ? smem + LR_NPX * LR_XS + LR_PHASE * 4 + (wv - 4) * (32 * SP)
                             : smem + LR_NPX * LR_XS + (2 * NB + 1) * LR_SLAB * 4);
        static_assert(4 * 32 * SP <= LR_NPX * LR_XS && 3 * 32 * SP <= LR_PHASE * 4 && (2 * NB + 1) * LR_SLAB * 4 + 32 * SP <= LR_PHASE * 4, "staging regions");
        f32x4* mine = reinterpret_cast<f32x4*>(stg + pxe * SP);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mine[hfe * 8 + g] = f32x4{accA[0][4 * g], accA[0][4 * g + 1], accA[0][4 * g + 2], accA[0][4 * g + 3]};
            mine[hfe * 8 + 4 + g] = f32x4{accB[0][4 * g], accB[0][4 * g + 1], accB[0][4 * g + 2], accB[0][4 * g + 3]};
            mine[16 + 2 * g + hfe] = f32x4{accC[4 * g], accC[4 * g + 1], accC[4 * g + 2], accC[4 * g + 3]};
        }
        const int ex0 = blockIdx.x * LR_TW, ey0 = blockIdx.y * LR_TH;    // (scalar)
        const int egy = ey0 + wv;
        if (egy >= p.h) return;                                       // (wave-uniform; no barrier follows)
        const int nq = (p.w - ex0 < 32 ? p.w - ex0 : 32) * (REC / 4);  // valid quads of this wave's run
        f32x4* row = reinterpret_cast<f32x4*>(p.lrcat + ((long long)egy * p.w + ex0) * REC);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const unsigned Q = 64u * j + (unsigned)ln;
            const unsigned rp = Q / 24u, rq = Q - rp * 24u;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stg + rp * SP + 4 * rq);
            if ((int)Q < nq) __builtin_nontemporal_store(v, row + Q);
        }
        return;
    }
#endif
    if (!valid) return;
    float* recf = p.lrcat + ((long long)gy * p.w + gx) * REC;
    f32x4* rec = reinterpret_cast<f32x4*>(recf + half * 32 * NB);
    // LRcat stores: plain.  (Write-through `sc1` / `sc0 sc1` and `nt` forms were measured in round 3 -- the 22 MB of records would leave the XCD's L2
    // while the kernel still computes instead of in the end-of-kernel write-back the dependent HR launch waits behind -- and were not faster.)
    auto st16 = [&](f32x4* dst, const f32x4& v) { *dst = v; };
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 a = {accA[t][4 * g], accA[t][4 * g + 1], accA[t][4 * g + 2], accA[t][4 * g + 3]};
            f32x4 b = {accB[t][4 * g], accB[t][4 * g + 1], accB[t][4 * g + 2], accB[t][4 * g + 3]};
            st16(rec + t * 4 + g, a);
            st16(rec + 4 * NB + t * 4 + g, b);
        }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 c = {accC[4 * g], accC[4 * g + 1], accC[4 * g + 2], accC[4 * g + 3]};
        st16(reinterpret_cast<f32x4*>(recf + 64 * NB + 8 * g + 4 * half), c);
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
    const float* table;          // [n_table][8] raw phase table (savsr_satu_phase_table); kept whole in LDS when n_table <= HR_TABLE_LDS
    int n_table, n_uw;
    const int* idx_h;            // [H] / [W] (+ padding to a multiple of 4): table row / column of each HR row / column
    const int* idx_w;
    const float* ptab;           // [H][W][8] per-pixel entries, offsets normalised (savsr_satu_expand_table); used when n_table > HR_TABLE_LDS
    const float* gyn;            // [H] / [W] (+ padding to a multiple of 4)
    const float* gxn;
    int H, W;
    float* out;
    long long out_plane;         // floats between output channel planes (>= H*W)
    int ty, txw, lrh, lrw;       // HR tile rows, 32-px column tiles per tile, staged LR window
    float omin_x, omin_y;        // lower bound of the sampling offsets (window origin)
    float step_x, step_y;        // LR pixels per HR pixel (1 / scale), for the window origin only
    int ntx, nty;                // tiles per row / per column of the HR image
    int* sched;                  // [16] tile-queue heads (one per XCD chunk) + exit count, all zero between launches; NULL: static walk
    float* seam;                 // row-summed form (QS kernels): [H][nseg][2 sides][9 groups], see hr_tile_px
    int nseg;                    // 32-pixel column segments per HR row
};

constexpr int HR_MAX_ROWS = 64;
constexpr int HR_TABLE_LDS = 256;   // phase tables of up to this many entries live whole in LDS (x4: 16, x2: 4, x1.5 x 4: 76)
__host__ __device__ constexpr int hr_lds_rec(int nb) { return rec_floats(nb) + 4; }       // floats per staged record
__host__ __device__ constexpr int hr_wimg_floats(int nb) { return nb * 2 * 2 * 64 * 4; }  // (Wb E) image [t][ks][part][lane][8 bf16]
// LDS of a workgroup, in floats: once  = image | bias [half][16 NB] (padded to 64) | whole table (small tables only)
//                                per staging buffer (two of them) = window | tile slice of the per-pixel table (large tables only) |
//                                gyn [64] | gxn [32 txw] | idx_h [64] | idx_w [32 txw] (small tables only)
__host__ __device__ inline int hr_once_floats(int nb, bool small) { return hr_wimg_floats(nb) + 64 + 16 + (small ? HR_TABLE_LDS * SAVSR_SATU_TABLE : 0); }
__host__ __device__ inline int hr_buf_floats(int nb, int ty, int txw, int lrh, int lrw, bool small) {
    return lrh * lrw * hr_lds_rec(nb) + (small ? 0 : ty * 32 * txw * SAVSR_SATU_TABLE) + HR_MAX_ROWS + 32 * txw + (small ? HR_MAX_ROWS + 32 * txw : 0);
}

struct Taps {
    int y0, x0;        // LR coordinates of the north-west tap, clamped into the image
    int dy, dx;        // step to the southern / eastern taps: 1, or 0 where that neighbour is the same (clamped) pixel
    float wgt[4];      // bilinear weights (nw, ne, sw, se), 0 for taps outside the image (zeros padding)
};

// onx, ony: the sampling offset already normalised as the reference does, (off * 2) / (size - 1)  (:285-287).
// An out-of-image tap contributes 0; its coordinate is clamped into the image (never widening the window a wave needs):
// whatever record it then reads is finite data multiplied by a zero weight.
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
    const int x0 = (int)xw, y0 = (int)yn;
    // per-axis weights with the validity folded in: the four products are the reference's (y weight) * (x weight) or exactly 0
    const float wx0 = (unsigned)x0 < (unsigned)w ? ex : 0.f, wx1 = (unsigned)(x0 + 1) < (unsigned)w ? lx : 0.f;
    const float wy0 = (unsigned)y0 < (unsigned)h ? sy : 0.f, wy1 = (unsigned)(y0 + 1) < (unsigned)h ? ly : 0.f;
    Taps t;
    t.x0 = min(max(x0, 0), w - 1);
    t.y0 = min(max(y0, 0), h - 1);
    t.dx = min(max(x0 + 1, 0), w - 1) - t.x0;
    t.dy = min(max(y0 + 1, 0), h - 1) - t.y0;
    t.wgt[0] = wy0 * wx0; t.wgt[1] = wy0 * wx1; t.wgt[2] = wy1 * wx0; t.wgt[3] = wy1 * wx1;
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
    // the four records of a gather: north-west one + the (0 or 1) steps; 24-bit multiplies (full rate; v_mul_lo_u32 is quarter
    // rate): window coordinates are tiny
    auto recs = [&](const Taps& t, const f32x4* (&r)[4]) {
        if (FROM_LDS) {
            const float* b = lds + __mul24(__mul24(t.y0 - ly0, p.lrw) + (t.x0 - lx0), LREC);
            const int sx = t.dx ? LREC : 0, sy = t.dy ? __mul24(p.lrw, LREC) : 0;
            r[0] = reinterpret_cast<const f32x4*>(b); r[1] = reinterpret_cast<const f32x4*>(b + sx);
            r[2] = reinterpret_cast<const f32x4*>(b + sy); r[3] = reinterpret_cast<const f32x4*>(b + sy + sx);
        } else {
            const float* b = p.lrcat + ((long long)t.y0 * p.w + t.x0) * REC;
            const int sx = t.dx ? REC : 0, sy = t.dy ? p.w * REC : 0;
            r[0] = reinterpret_cast<const f32x4*>(b); r[1] = reinterpret_cast<const f32x4*>(b + sx);
            r[2] = reinterpret_cast<const f32x4*>(b + sy); r[3] = reinterpret_cast<const f32x4*>(b + sy + sx);
        }
    };
    const f32x4* ro[4];
    const f32x4* rs[4];
    recs(to, ro);
    recs(ts, rs);
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

// Tail-projected form, LANE = PIXEL: a wave tile is 32 pixels x 2 rows (lanes 0-31 row Y, lanes 32-63 row Y + 1); every lane
// gathers all 32 output rows of ITS pixel, so the per-pixel work that both lanes of a pixel repeated in hr_tile (table lookup,
// tap arithmetic, window test, record addresses, the expert operand) is done once: ~300 vector instructions per 32 pixels
// instead of ~424 in a kernel whose vector issue is 64 % busy.  The 32x32x16 MFMA wants (pixel, row half) lanes, so the
// accumulators and the expert B operands are transposed in registers in front of it: with a = rows of half 0 and b = rows of
// half 1 of the lane's own pixel, v_permlane32_swap(a, b) leaves [a of lanes 0-31 | b of lanes 0-31] = the accumulator layout
// of the first row's 32 pixels in a, and that of the second row's in b (one instruction per register pair).
// QS (row-summed form, savsr_satu_hr_tail_q): the 27 planes are ordered so that the three horizontal taps of a (tap row ky, colour o)
// group sit in ONE lane half at accumulator registers 3 gi, 3 gi + 1, 3 gi + 2 (groups 0 .. 4 in half 0, 5 .. 8 in half 1: the
// caller's Wt27 row order), and the stage adds them itself -- Q[g][Y][X] = P(kx = 1)[X] + P(kx = 0)[X - 1] + P(kx = 2)[X + 1], the
// neighbours' values by wave-wide DPP shifts -- and stores 9 planes instead of 27 (33 MB instead of 99.5 at 720x1280); the terms
// that cross a 32-pixel segment border go to two small side planes (`seam`: what the right neighbour's first pixel / the left
// neighbour's last pixel still needs), which savsr_tail_gather_q adds with the three vertical taps.
template <bool FROM_LDS, bool QS = false>
__device__ __forceinline__ void hr_tile_px(const HrParams& p, const float* lds, int ly0, int lx0, const Taps& to, const Taps& ts,
                                           const f32x4 rr, int lane, bool valid0, bool valid1, unsigned o_off0, unsigned o_off1,
                                           const float* cst, int Y = 0, int seg = 0, bool row1 = false) {
    constexpr int REC = rec_floats(1), LREC = hr_lds_rec(1);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int half = lane >> 5;
    auto recs = [&](const Taps& t, const f32x4* (&r)[4]) {
        if (FROM_LDS) {
            const float* b = lds + __mul24(__mul24(t.y0 - ly0, p.lrw) + (t.x0 - lx0), LREC);
            const int sx = t.dx ? LREC : 0, sy = t.dy ? __mul24(p.lrw, LREC) : 0;
            r[0] = reinterpret_cast<const f32x4*>(b); r[1] = reinterpret_cast<const f32x4*>(b + sx);
            r[2] = reinterpret_cast<const f32x4*>(b + sy); r[3] = reinterpret_cast<const f32x4*>(b + sy + sx);
        } else {
            const float* b = p.lrcat + ((long long)t.y0 * p.w + t.x0) * REC;
            const int sx = t.dx ? REC : 0, sy = t.dy ? p.w * REC : 0;
            r[0] = reinterpret_cast<const f32x4*>(b); r[1] = reinterpret_cast<const f32x4*>(b + sx);
            r[2] = reinterpret_cast<const f32x4*>(b + sy); r[3] = reinterpret_cast<const f32x4*>(b + sy + sx);
        }
    };
    const f32x4* ro[4];
    const f32x4* rs[4];
    recs(to, ro);
    recs(ts, rs);
    // ---- gathers into a (rows acc_row(r, 0)) and b (rows acc_row(r, 1)) of the lane's own pixel, bias first ----
    const f32x4* fb4 = reinterpret_cast<const f32x4*>(cst + hr_wimg_floats(1));     // LDS copy of the bias, packed [half][16]
    f32x16 a, b;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 ba = fb4[g], bb = fb4[4 + g];
        a[4 * g] = ba[0]; a[4 * g + 1] = ba[1]; a[4 * g + 2] = ba[2]; a[4 * g + 3] = ba[3];
        b[4 * g] = bb[0]; b[4 * g + 1] = bb[1]; b[4 * g + 2] = bb[2]; b[4 * g + 3] = bb[3];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                    // G(Wt27 Wb x, off): record quads 4 .. 7 (half 0), 12 .. 15 (half 1)
        const float wk = to.wgt[k];
#pragma unroll
        for (int g = 0; g < 4; ++g) { fma_quad(a, g, wk, ro[k][4 + g]); if (g < 3) fma_quad(b, g, wk, ro[k][12 + g]); }   // (b's quad 3 = rows 28 .. 31: padding, never stored)
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                    // G(Wt27 Wa sta, soff): record quads 0 .. 3, 8 .. 11
        const float wk = ts.wgt[k];
#pragma unroll
        for (int g = 0; g < 4; ++g) { fma_quad(a, g, wk, rs[k][g]); if (g < 3) fma_quad(b, g, wk, rs[k][8 + g]); }
    }
    // ---- t_j = sum_m r_m (C_m G(x, off))_j : the record's 32 compressed channels (quads 16 .. 23 = (m, j) at 8 m + j) ----
    f32x2 tq[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wk = to.wgt[k];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 v0 = ro[k][16 + 2 * m], v1 = ro[k][16 + 2 * m + 1];
            const float wr = wk * rr[m];
            const f32x2 w2 = {wr, wr};
            tq[0] = __builtin_elementwise_fma(w2, f32x2{v0[0], v0[1]}, tq[0]);
            tq[1] = __builtin_elementwise_fma(w2, f32x2{v0[2], v0[3]}, tq[1]);
            tq[2] = __builtin_elementwise_fma(w2, f32x2{v1[0], v1[1]}, tq[2]);
            tq[3] = __builtin_elementwise_fma(w2, f32x2{v1[2], v1[3]}, tq[3]);
        }
    }
    // ---- B operands of the expert MFMA: v[(n, j)] = r_n t_j; k = 16 ks + 8 kh + j <-> n = 2 ks + kh.  Own-pixel values for
    // n = 2 ks (X) and 2 ks + 1 (Y); swap -> X = operand of the first row's pixels, Y = of the second row's ----
    bf16x8 bh[2][2], bl[2][2];                       // [pixel row][ks]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 xh, xl, yh, yl;
        {
            const float r0 = rr[2 * ks], r1 = rr[2 * ks + 1];
            const f32x4 a0 = {r0 * tq[0][0], r0 * tq[0][1], r0 * tq[1][0], r0 * tq[1][1]}, a1 = {r0 * tq[2][0], r0 * tq[2][1], r0 * tq[3][0], r0 * tq[3][1]};
            const f32x4 c0 = {r1 * tq[0][0], r1 * tq[0][1], r1 * tq[1][0], r1 * tq[1][1]}, c1 = {r1 * tq[2][0], r1 * tq[2][1], r1 * tq[3][0], r1 * tq[3][1]};
            split8v(a0, a1, xh, xl);
            split8v(c0, c1, yh, yl);
        }
        u32x4 uxh = __builtin_bit_cast(u32x4, xh), uxl = __builtin_bit_cast(u32x4, xl), uyh = __builtin_bit_cast(u32x4, yh), uyl = __builtin_bit_cast(u32x4, yl);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const auto sh = __builtin_amdgcn_permlane32_swap(uxh[d], uyh[d], false, false);
            uxh[d] = sh[0]; uyh[d] = sh[1];
            const auto sl = __builtin_amdgcn_permlane32_swap(uxl[d], uyl[d], false, false);
            uxl[d] = sl[0]; uyl[d] = sl[1];
        }
        bh[0][ks] = __builtin_bit_cast(bf16x8, uxh); bl[0][ks] = __builtin_bit_cast(bf16x8, uxl);
        bh[1][ks] = __builtin_bit_cast(bf16x8, uyh); bl[1][ks] = __builtin_bit_cast(bf16x8, uyl);
    }
    // ---- to the MFMA's (pixel, row half) layout, then + (Wt27 Wb E) v on the bf16 matrix cores ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[r]), __float_as_uint(b[r]), false, false);
        a[r] = __uint_as_float(sw[0]);
        b[r] = __uint_as_float(sw[1]);
    }
    const bf16x8* wimg = reinterpret_cast<const bf16x8*>(cst) + lane;               // LDS copy of [ks][part][lane]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 wh = wimg[(ks * 2 + 0) * 64], wl = wimg[(ks * 2 + 1) * 64];
        a = mma3(wh, wl, bh[0][ks], bl[0][ks], a);
        b = mma3(wh, wl, bh[1][ks], bl[1][ks], b);
    }
    // ---- stores (see hr_tile): rows 27 .. 31 are padding ----
    const long long HW = p.out_plane;
    float* outp = p.out;
    asm volatile("" : "+s"(outp));
    if constexpr (QS) {
        const int px = lane & 31;
        const bool vx = valid0;                                   // this lane's column lies inside the image (X < W)
        // q = c1 + shr(c0) [unless px == 0] + shl(c2) [unless px == 31 or the right neighbour lies beyond the image] (wave-wide DPP shifts).
        // keep_l keeps lane 0 of a half from seeing the OTHER half's lane 31, keep_r lane 31 from the other half's lane 0 -- those two terms are the seams' business -- and zeroes the right neighbour's term
        // where that neighbour lies beyond the image's last column (the tail conv's zero padding; a left neighbour beyond it only
        // feeds pixels that are not stored).
        // The neighbours' terms are SELECTED away, not multiplied by 0 (rounds 4-5 used 0 / 1 factors): the lane to the right of the image's
        // last column -- and lane 31 of a partial last segment, which lane 32 sees as its left neighbour -- are lanes beyond the image, whose
        // accumulators are computed from whatever lies behind the per-pixel table / coordinate arrays; when that happens to be a NaN or an
        // infinity, 0 x NaN put a NaN into column W - 1 or into the segment's first column (seen on fresh boxes: ~1 process in 3, a handful of
        // pixels; tools/soak.py demands finite outputs since round 6).  Same arithmetic for every other lane: x * 1 + y == x + y exactly.
        const bool keep_l = px != 0, keep_r = !(px == 31 || 32 * seg + px + 1 >= p.W);
#pragma unroll
        for (int G = 0; G < 2; ++G) {
            const f32x16& acc = G ? b : a;
            const bool row_ok = G == 0 || row1;                   // (wave-uniform) the pixel row exists
            const bool valid = G ? valid1 : valid0;
            unsigned oo = G ? o_off1 : o_off0;
            asm volatile("" : "+v"(oo));
            auto st1 = [&](float* pl, unsigned off, float v) {
                __attribute__((address_space(1))) float* a_ = (__attribute__((address_space(1))) float*)((__attribute__((address_space(1))) char*)pl + off);
                __hip_atomic_store(a_, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // emitted as a `sc1` (write-through) store
            };
            float q[5];
#pragma unroll
            for (int gi = 0; gi < 5; ++gi) {
                const float l = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[3 * gi]), 0x138, 0xf, 0xf, true));        // wave_shr:1: lane i <- lane i - 1
                const float r = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[3 * gi + 2]), 0x130, 0xf, 0xf, true));    // wave_shl:1: lane i <- lane i + 1
                q[gi] = (keep_r ? r : 0.f) + ((keep_l ? l : 0.f) + acc[3 * gi + 1]);
            }
            if (valid) {
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    float* pl = outp + (long long)gi * HW;
                    asm volatile("" : "+s"(pl));
                    st1(pl, oo, q[gi]);
                }
                if (half == 0) {
                    float* pl = outp + 4ll * HW;
                    asm volatile("" : "+s"(pl));
                    st1(pl, oo, q[4]);
                }
            }
            // seams: [row][segment][side A | side B][9 groups], 72 B per (row, segment): what the right neighbour's first pixel (side A of
            // segment seg + 1: this segment's last pixel's kx = 0 terms) and the left neighbour's last pixel (side B of seg - 1) still need
            // (two separate blocks with static register indices: a select between acc[3 gi] and acc[3 gi + 2] on the lane becomes a
            // dynamic register index -- 16 compares + selects per value and a scratch reload behind an s_waitcnt vmcnt(0), i.e. behind
            // every output store in flight: 35 -> 47 us.  The row's seam record has a wave-uniform base: scalar address arithmetic,
            // the lane supplies only its half's 20-byte offset.)
            typedef float f32x4u_ __attribute__((ext_vector_type(4), aligned(4)));
            if (row_ok) {
                float* rowb = p.seam + (long long)((Y + G) * p.nseg + seg) * 18;      // this segment's record: [side A 9 | side B 9]
                asm volatile("" : "+s"(rowb));
                const unsigned ho = half ? 20u : 0u;
                if (vx && px == 31 && seg + 1 < p.nseg) {                               // side A of segment seg + 1
                    __attribute__((address_space(1))) char* d_ = (__attribute__((address_space(1))) char*)(rowb + 18) + ho;
                    *(__attribute__((address_space(1))) f32x4u_*)d_ = f32x4u_{acc[0], acc[3], acc[6], acc[9]};
                    if (half == 0) *(__attribute__((address_space(1))) float*)(d_ + 16) = acc[12];
                }
                if (vx && px == 0 && seg > 0) {                                         // side B of segment seg - 1
                    __attribute__((address_space(1))) char* d_ = (__attribute__((address_space(1))) char*)(rowb - 9) + ho;
                    *(__attribute__((address_space(1))) f32x4u_*)d_ = f32x4u_{acc[2], acc[5], acc[8], acc[11]};
                    if (half == 0) *(__attribute__((address_space(1))) float*)(d_ + 16) = acc[14];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int G = 0; G < 2; ++G) {
        const f32x16& acc = G ? b : a;
        if (G ? valid1 : valid0) {
            unsigned oo = G ? o_off1 : o_off0;
            asm volatile("" : "+v"(oo));
            // `sc1` (write-through) stores -- the planes leave L2 as they are written, so the kernel's end has (almost) no
            // dirty lines left to write back in front of the dependent tail launch
            auto st1 = [&](float* pl, unsigned off, float v) {
                __attribute__((address_space(1))) float* a_ = (__attribute__((address_space(1))) float*)((__attribute__((address_space(1))) char*)pl + off);
                __hip_atomic_store(a_, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // emitted as a `sc1` (write-through) store
            };
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                float* pl = outp + (long long)acc_row(r, 0) * HW;
                asm volatile("" : "+s"(pl));
                st1(pl, oo, acc[r]);
            }
            if (half == 0) {
#pragma unroll
                for (int r = 12; r < 15; ++r) {
                    float* pl = outp + (long long)acc_row(r, 0) * HW;
                    asm volatile("" : "+s"(pl));
                    st1(pl, oo, acc[r]);
                }
            }
        }
    }
}

// ONE workgroup per CU: HR_WAVES compute waves (wave tiles are dealt round-robin over them) + HR_PRODUCERS producer waves that
// only issue the next tile's LDS-DMAs.  A wave's gather chain is latency-bound (LDS round trips): alone on its SIMD a wave
// needs ~4.1 k cycles per 32-pixel tile, three waves sharing a SIMD finish one every ~1.3 k, so the kernel wants as many
// compute waves per SIMD as the register file holds (the launch bounds give the VGPR cap the kernel is compiled for).
// A producer wave gets ~4-5 LDS-DMAs in flight (one 1-KiB DMA per ~450 cycles, measured).
// The tail-projected forms (NB == 1) run lane = pixel wave tiles (hr_tile_px); the standalone form (NB == 2) two lanes per pixel (hr_tile).
// Two wave splits are compiled (savsr_satu_tiling.variant; the caller picks per size / scale, results do not depend on it):
//   variant 0 =  8 compute + 4 producer waves (3 waves per SIMD)
//   variant 1 = 10 compute + 6 producer waves (4 per SIMD: the lane = pixel tile needs 126 VGPRs)
// measured at 180x320, tail form (us; 8+4 -> 10+6): x4 40.4 -> 37.1, x3.7 37.3 -> 36.3, x3.9 37.2 -> 38.4, x3 28.0 -> 29.3, x2 22.0 -> 23.7,
// (2.95, 3.75) 38.6 -> 43.9, 480x318 x(1.5, 4) 51.4 -> 60.0; other splits at x4: 12+4 38.6-39.1, 10+2 38.5-38.9, 8+8 37.9-39.3, 6+10 42-43, 14+2 50.
//   variant 2 = 12 compute + 4 producer waves, variant 3 = 8 + 8 (round 3: with write-through plane stores the balance between
//               gather waves and staging waves is measured again per size / scale; the engine times every plan)
//               variant 4 = 6 + 10: staging-bound shapes (low vertical scale: a tile's LRcat window is large against its HR pixels)
constexpr int HR_VARIANTS = 5;
__host__ __device__ constexpr int hr_compute_waves(int variant) { return variant == 1 ? 10 : (variant == 2 ? 12 : (variant == 4 ? 6 : 8)); }
__host__ __device__ constexpr int hr_producer_waves(int variant) { return variant == 1 ? 6 : (variant == 3 ? 8 : (variant == 4 ? 10 : 4)); }
// One LDS-DMA: the active lanes move 16 B each, global (uniform 64-bit base in SGPRs + a 32-bit byte offset per lane) ->
// lds_base + 16 * lane (no registers, no ds_write).  The scalar-base form keeps the whole address computation of a staging
// loop on the scalar unit: with a per-lane 64-bit address every DMA cost ~40 vector instructions (quarter-rate 64-bit
// multiply-adds, an integer division) and issuing a tile's ~18 DMAs took a wave 5 k cycles.
// hipcc does not count these loads: the consumer waits with an explicit s_waitcnt vmcnt in front of its barrier.
#define HR_DMA16(sbase, voff, lds_base) do { \
        const unsigned dst_ = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_base)); \
        const unsigned long long sb_ = (unsigned long long)(uintptr_t)(sbase); \
        const unsigned sb_lo_ = __builtin_amdgcn_readfirstlane((unsigned)sb_), sb_hi_ = __builtin_amdgcn_readfirstlane((unsigned)(sb_ >> 32)); \
        unsigned keep_; \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(voff), "s"((((unsigned long long)sb_hi_) << 32) | sb_lo_), "s"(dst_) : "memory"); \
    } while (0)

// PERSISTENT kernel: gridDim.x workgroups (a multiple of 8, `occupancy x CUs`) walk the HR tiles.
//   * Tile order is XCD-aware: workgroups b, b + 8, b + 16, ... share an XCD (round-robin dispatch), so XCD x = b % 8 is given the
//     x-th eighth of the row-major tile sequence and its workgroups take consecutive tiles of it: the LRcat band an XCD
//     gathers from (1/8 of 22 MB + halos) stays in its own 4 MiB L2 and neighbouring tiles are staged at the same time.
//     (Placement is a speed assumption only; results never depend on it.)
//   * Staging is double-buffered and has its own wave: while the four compute waves gather the current tile from one LDS
//     buffer, the PRODUCER waves (waves 4, 5) issue the LDS-DMAs of the next tile into the other (window records, its slice of the
//     table, gyn / gxn / table indices -- every address is arithmetic on kernel arguments, on the scalar unit), waits for them
//     (its vmcnt holds nothing else) and meets the compute waves at the one barrier per tile.  Measured before this split
//     (DESIGN.md): staging alone 16-21 us, gathers + stores alone 37 us, together 50 us in a one-tile-per-workgroup kernel
//     (workgroups launched together stage together), and 52-55 us with the compute waves issuing the DMAs themselves -- the
//     CU's memory pipe queues them behind the output stores, ~200 cycles of issue stall per DMA, 3.7 k cycles per tile.
// DIAG = the instrumented build (section stamps, timing experiments); the product launch uses DIAG = false.
template <bool DIAG, int NB, int VAR, bool QS = false>
__global__ __launch_bounds__(64 * (hr_compute_waves(VAR) + hr_producer_waves(VAR)), (hr_compute_waves(VAR) + hr_producer_waves(VAR) + 3) / 4)
void satu_hr_kernel(const HrParams p) {
    static_assert(!QS || NB == 1, "the row-summed form is the lane = pixel tail form");
    constexpr int HR_WAVES = hr_compute_waves(VAR), HR_PRODUCERS = hr_producer_waves(VAR);
    constexpr int REC = rec_floats(NB);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, px = lane & 31;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);      // tile walk and row / column tests as scalar code
    const int ncol = 32 * p.txw;
    const bool small = p.n_table <= HR_TABLE_LDS;                 // (uniform) whole table in LDS vs per-pixel slices
    const long long t_entry = DIAG ? SATU_T() : 0;
    [[maybe_unused]] const long long rt_entry = DIAG ? (long long)__builtin_amdgcn_s_memrealtime() : 0;     // 100 MHz wall clock

    // ---- this workgroup's tiles --------------------------------------------------------------------------------
    const int ntile_img = p.ntx * p.nty;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int chunk0 = (int)(((long long)xcd * ntile_img) >> 3), chunk_n = (int)(((long long)(xcd + 1) * ntile_img) >> 3) - chunk0;
    // Tile walk.  Static: tiles slot, slot + nslot, ... of the chunk.  Dynamic (p.sched): the first two tiles are the static ones
    // (no atomic in front of the first DMAs), every later one comes from the chunk's queue head -- a workgroup that got cheap
    // tiles (image borders) or a fast CU simply takes more, instead of every CU waiting for the one with ceil(tiles / CUs).
    // The last workgroup to leave zeroes the heads again, so the caller zero-fills `sched` once, not per launch.
    auto leave = [&]() {
        if (p.sched && tid == 64 * HR_WAVES) {
            const int old = __hip_atomic_fetch_add(p.sched + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (int)gridDim.x - 1) {
                for (int i = 0; i < 9; ++i) __hip_atomic_store(p.sched + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    if (slot >= chunk_n) { leave(); return; }                     // (uniform) more workgroups than tiles

    // ---- LDS carve-up -------------------------------------------------------------------------------------------
    float* cst = lds;                                             // image | bias
    int* ring = reinterpret_cast<int*>(cst + hr_wimg_floats(NB) + 64);   // [4] tile ids handed out by the queue (dynamic walk)
    float* tabl = cst + hr_wimg_floats(NB) + 64 + 16;             // whole table, offsets normalised (small tables)
    const int once = hr_once_floats(NB, small), bufsz = hr_buf_floats(NB, p.ty, p.txw, p.lrh, p.lrw, small);
    const int win_floats = p.lrh * p.lrw * hr_lds_rec(NB), slice_floats = small ? 0 : p.ty * ncol * SAVSR_SATU_TABLE;

#if SATU_HAS_STAMPS
    const int dbg_all = DIAG ? __builtin_amdgcn_readfirstlane(g_satu_stamps_on) : 0;
#else
    static_assert(!DIAG, "DIAG kernels exist in the instrumented library only");
    const int dbg_all = 0;
#endif
    const int stamps_on = dbg_all & 1;
    const bool dbg_nostore = dbg_all & 2, dbg_stage_only = dbg_all & 4, dbg_nostage = dbg_all & 8;     // timing experiments (results invalid)

    // window origin of a tile: its base sampling coordinate + the lower bound of the offsets, evaluated from kernel arguments
    // only; being a plan, it needs no bit-exactness (a wave whose taps fall outside gathers from global memory)
    // (the window is then shifted back inside the image: every staged record exists, so the DMAs need no per-record test)
    auto origin = [&](int Y0, int X0, int& ly0, int& lx0) {
        const float by = ((float)Y0 + 0.5f) * p.step_y - 0.5f + p.omin_y - 0.01f;
        const float bx = ((float)X0 + 0.5f) * p.step_x - 0.5f + p.omin_x - 0.01f;
        const int iy = (int)floorf(fminf(fmaxf(by, 0.f), (float)(p.h - 1)));
        const int ix = (int)floorf(fminf(fmaxf(bx, 0.f), (float)(p.w - 1)));
        ly0 = __builtin_amdgcn_readfirstlane(iy < p.h - p.lrh ? iy : p.h - p.lrh);
        lx0 = __builtin_amdgcn_readfirstlane(ix < p.w - p.lrw ? ix : p.w - p.lrw);
    };
    // every DMA of one tile into staging buffer `buf`, dealt over the producer waves
    const unsigned lane16 = 16u * (unsigned)lane;
    const bool producer = wave_s >= HR_WAVES;                      // (uniform per wave)
    const int pk = wave_s - HR_WAVES;                              // producer index
    auto stage = [&](int tile, float* buf) {
        if (dbg_nostage) return;
        const int tyi = tile / p.ntx, txi = tile - tyi * p.ntx;
        const int Y0 = tyi * p.ty, X0 = txi * ncol;
        if (p.lrh > 0) {
            int ly0, lx0;
            origin(Y0, X0, ly0, lx0);
            // The window is a linear array of padded records (REC / 4 data chunks of 16 B + one pad chunk: the pitch that keeps
            // the b128 gathers conflict-free), filled by FULL 1-KiB DMAs: lane i of DMA j owns LDS chunk q = 64 j + i, i.e.
            // chunk q % CHP of record q / CHP, and fetches it from its record's place in LRcat (the pad chunk's lane sits out).
            // One DMA per record (24 of 64 lanes) needed 2.7x as many DMAs, and a DMA costs its wave ~250 cycles whatever its size.
            constexpr int CHP = REC / 4 + 1;
            const int nq = p.lrh * p.lrw * CHP;
            const float* sbase = p.lrcat + ((long long)ly0 * p.w + lx0) * REC;
            for (int j = pk; j * 64 < nq; j += HR_PRODUCERS) {
                const int q = j * 64 + lane;
                const int rec = q / CHP, c = q - rec * CHP;
                const int ry = rec / p.lrw, rx = rec - ry * p.lrw;
                if (q < nq && c < REC / 4) HR_DMA16(sbase, (unsigned)(((ry * p.w + rx) * REC + 4 * c) * 4), buf + j * 256);
            }
        }
        float* slice = buf + win_floats;
        float* rowg = slice + slice_floats;
        float* colg = rowg + HR_MAX_ROWS;
        if (!small) {
            // the tile's slice of the per-pixel table: one DMA (1 KiB = 32 entries of 32 B) per (tile row, 32-pixel column tile)
            for (int trow = pk; trow < p.ty; trow += HR_PRODUCERS) {
                const int Yc = Y0 + trow < p.H ? Y0 + trow : p.H - 1;
                for (int tcol = 0; tcol < p.txw; ++tcol) {
                    const int Xb = X0 + tcol * 32;
                    if (Xb + (lane >> 1) < p.W) HR_DMA16(p.ptab + ((long long)Yc * p.W + Xb) * SAVSR_SATU_TABLE, lane16, slice + (trow * ncol + tcol * 32) * 8);
                }
            }
        }
        // per-row / per-column scalars, 4 per lane (the arrays are padded to a multiple of 4 elements by the caller)
        const int H4 = (p.H + 3) & ~3, W4 = (p.W + 3) & ~3;
        if (pk == 0) {
            if (4 * lane < p.ty && Y0 + 4 * lane < H4) HR_DMA16(p.gyn + Y0, lane16, rowg);
            if (4 * lane < ncol && X0 + 4 * lane < W4) HR_DMA16(p.gxn + X0, lane16, colg);
        }
        if (small && pk == HR_PRODUCERS - 1) {
            int* rowi = reinterpret_cast<int*>(colg + ncol);
            int* coli = rowi + HR_MAX_ROWS;
            if (4 * lane < p.ty && Y0 + 4 * lane < H4) HR_DMA16(p.idx_h + Y0, lane16, rowi);
            if (4 * lane < ncol && X0 + 4 * lane < W4) HR_DMA16(p.idx_w + X0, lane16, coli);
        }
    };

    // ---- prologue: first tile's DMAs, then the once-per-workgroup constants under them --------------------------
    float* buf_cur = lds + once;
    float* buf_nxt = buf_cur + bufsz;
    int idx = slot;                                                // index of the current tile inside the chunk
    int idx_n = slot + nslot;                                      // ... of the next one (>= chunk_n: none)
    if (producer) stage(chunk0 + idx, buf_cur);
    else {
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.wt.wbe_w);
        for (int e = tid; e < hr_wimg_floats(NB) / 4; e += 64 * HR_WAVES) reinterpret_cast<f32x4*>(cst)[e] = wsrc[e];
        if (tid < 8 * NB) reinterpret_cast<f32x4*>(cst + hr_wimg_floats(NB))[tid] = reinterpret_cast<const f32x4*>(p.wt.fusion_b)[tid];
        if (small) {
            const float fw1 = (float)(p.w - 1), fh1 = (float)(p.h - 1);
            for (int e = tid; e < p.n_table * 2; e += 64 * HR_WAVES) {
                f32x4 v = reinterpret_cast<const f32x4*>(p.table)[e];
                if (e & 1) {               // the offset quad: normalised once here, as the reference does per pixel ((off * 2) / (size - 1), :285-287)
                    v[0] = (v[0] * 2.f) / fw1; v[1] = (v[1] * 2.f) / fh1; v[2] = (v[2] * 2.f) / fw1; v[3] = (v[3] * 2.f) / fh1;
                }
                reinterpret_cast<f32x4*>(tabl)[e] = v;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    long long tacc[SSTAMP_N] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_prev = stamps_on ? SATU_T() : 0;
    tacc[4] = t_prev - t_entry;                                      // prologue (stamps are written by wave 0, a compute wave)
#define HR_MARK(i) do { if (stamps_on) { const long long t_now = SATU_T(); tacc[i] += t_now - t_prev; t_prev = t_now; } } while (0)
    const int ntile = p.ty * p.txw;
    for (int it = 0;; ++it) {
        const int tile = chunk0 + idx;
        if (producer) {
            if (idx_n < chunk_n) stage(chunk0 + idx_n, buf_nxt);      // next tile's DMAs fly under this tile's gathers
            if (p.sched && tid == 64 * HR_WAVES)                      // the tile after that: one queue pop per iteration
                ring[(it + 2) & 3] = 2 * nslot + __hip_atomic_fetch_add(p.sched + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (these waves issue nothing else on vmcnt)
        } else {
        const int tyi = tile / p.ntx, txi = tile - tyi * p.ntx;
        const int Y0 = tyi * p.ty, X0 = txi * ncol;
        int ly0 = 0, lx0 = 0;
        if (p.lrh > 0) origin(Y0, X0, ly0, lx0);
        const float* slice = buf_cur + win_floats;
        const float* rowg = slice + slice_floats;
        const float* colg = rowg + HR_MAX_ROWS;
        const int* rowi = reinterpret_cast<const int*>(colg + ncol);
        const int* coli = rowi + HR_MAX_ROWS;
        if (!dbg_stage_only && NB == 1) {
            // lane = pixel: wave tiles of 32 pixels x 2 rows (tile_rows is a multiple of 4)
            const int npair = (p.ty >> 1) * p.txw, hl = lane >> 5;
            for (int T = wave_s; T < npair; T += HR_WAVES) {
                const int trow2 = p.txw == 1 ? T : T / p.txw;
                const int tcol = T - trow2 * p.txw;
                const int Y = Y0 + 2 * trow2;
                const int Xb = X0 + tcol * 32;
                if (Y >= p.H || Xb >= p.W) continue;                  // wave-uniform
                const bool row1 = Y + 1 < p.H;                        // (uniform) odd image height: the second row of the last pair is absent
                const int trow = 2 * trow2 + (row1 ? hl : 0);         // its lanes recompute the first row (nothing stored)
                const int X = Xb + px;
                const float* te = small ? tabl + (rowi[trow] * p.n_uw + coli[tcol * 32 + px]) * SAVSR_SATU_TABLE
                                        : slice + (trow * ncol + tcol * 32 + px) * SAVSR_SATU_TABLE;
                const f32x4 rr = *reinterpret_cast<const f32x4*>(te);
                const f32x4 oo = *reinterpret_cast<const f32x4*>(te + 4);
                const float gxn = colg[tcol * 32 + px];
                const float gyn = rowg[trow];
                const Taps to = make_taps(gxn, gyn, oo[0], oo[1], p.h, p.w);
                const Taps ts = make_taps(gxn, gyn, oo[2], oo[3], p.h, p.w);
                const bool inside = p.lrh > 0 &&
                    (unsigned)(to.y0 - ly0) < (unsigned)(p.lrh - to.dy) && (unsigned)(to.x0 - lx0) < (unsigned)(p.lrw - to.dx) &&
                    (unsigned)(ts.y0 - ly0) < (unsigned)(p.lrh - ts.dy) && (unsigned)(ts.x0 - lx0) < (unsigned)(p.lrw - ts.dx);
                // after the transposes lane (px, hl) holds rows acc_row(r, hl) of pixel (row, Xb + px) for each of the two rows
                const bool vx = X < p.W && !dbg_nostore;
                // (QS: lane half 1 holds groups 5 .. 8 = planes 5 .. 8; else rows acc_row(r, 1) = acc_row(r, 0) + 4)
                const unsigned o_off0 = 4u * (unsigned)(Y * p.W + X) + (hl ? (QS ? 20u : 16u) * (unsigned)p.out_plane : 0u);
                const unsigned o_off1 = o_off0 + 4u * (unsigned)p.W;
                if (__all(inside)) hr_tile_px<true, QS>(p, buf_cur, ly0, lx0, to, ts, rr, lane, vx, vx && row1, o_off0, o_off1, cst, Y, Xb >> 5, row1);
                else {
                    hr_tile_px<false, QS>(p, buf_cur, ly0, lx0, to, ts, rr, lane, vx, vx && row1, o_off0, o_off1, cst, Y, Xb >> 5, row1);
                    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the fallback's gathers are not left pending
                }
            }
        } else if (!dbg_stage_only)
        for (int T = wave_s; T < ntile; T += HR_WAVES) {
            const int trow = p.txw == 1 ? T : T / p.txw;                // (a run-time integer division is ~14 vector instructions)
            const int tcol = T - trow * p.txw;
            const int Y = Y0 + trow;
            const int Xb = X0 + tcol * 32;
            if (Y >= p.H || Xb >= p.W) continue;                      // wave-uniform
            const int X = Xb + px;
            const bool valid = X < p.W && !dbg_nostore;
            // everything a tile looks up is in LDS: no global load is pending, as far as hipcc can tell, anywhere in this loop
            // (one would turn into s_waitcnt vmcnt(0..1) behind the previous tile's output stores: a write round trip per tile)
            const float* te = small ? tabl + (rowi[trow] * p.n_uw + coli[tcol * 32 + px]) * SAVSR_SATU_TABLE
                                    : slice + (trow * ncol + tcol * 32 + px) * SAVSR_SATU_TABLE;
            const f32x4 rr = *reinterpret_cast<const f32x4*>(te);
            const f32x4 oo = *reinterpret_cast<const f32x4*>(te + 4);
            const float gxn = colg[tcol * 32 + px];
            const float gyn = rowg[trow];
            const Taps to = make_taps(gxn, gyn, oo[0], oo[1], p.h, p.w);
            const Taps ts = make_taps(gxn, gyn, oo[2], oo[3], p.h, p.w);
            // all 8 taps inside the staged window?  (unsigned compare: below the origin wraps to a huge value)
            const bool inside = p.lrh > 0 &&
                (unsigned)(to.y0 - ly0) < (unsigned)(p.lrh - to.dy) && (unsigned)(to.x0 - lx0) < (unsigned)(p.lrw - to.dx) &&
                (unsigned)(ts.y0 - ly0) < (unsigned)(p.lrh - ts.dy) && (unsigned)(ts.x0 - lx0) < (unsigned)(p.lrw - ts.dx);
            // byte offset of this lane's pixel inside channel plane acc_row(r, 0); the half's +4 channels are folded in
            const unsigned o_off = 4u * (unsigned)(Y * p.W + X) + (half ? 16u * (unsigned)p.out_plane : 0u);
            if (__all(inside)) hr_tile<true, NB>(p, buf_cur, ly0, lx0, to, ts, rr, half, lane, valid, o_off, cst);
            else {
                hr_tile<false, NB>(p, buf_cur, ly0, lx0, to, ts, rr, half, lane, valid, o_off, cst);
                __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the fallback's gathers are not left pending (see above)
            }
        }
        }
        HR_MARK(2);                                                  // compute waves: gathers + MFMA + store issue; producer: DMA issue + landing
        if (idx_n >= chunk_n) break;                                  // (uniform) last tile
        __syncthreads();                                              // buffer hand-over: the staged tile is complete, the computed one is free
        HR_MARK(3);                                                  // barrier
        float* t_ = buf_cur; buf_cur = buf_nxt; buf_nxt = t_;
        idx = idx_n;
        idx_n = p.sched ? ring[(it + 2) & 3] : idx + nslot;
    }
    leave();
#if SATU_HAS_STAMPS
    if (stamps_on) {
        __builtin_amdgcn_s_waitcnt(0);
        if (tid == 0 && blockIdx.x < SSTAMP_BLOCKS) {
            for (int i = 0; i < 5; ++i) g_satu_stamps[blockIdx.x * SSTAMP_N + i] = tacc[i];
            g_satu_stamps[blockIdx.x * SSTAMP_N + 5] = rt_entry;                                     // wall-clock start / end (100 MHz ticks)
            g_satu_stamps[blockIdx.x * SSTAMP_N + 6] = (long long)__builtin_amdgcn_s_memrealtime();
            g_satu_stamps[blockIdx.x * SSTAMP_N + 7] = SATU_T() - t_entry;
        }
    }
#endif
}

// Per-pixel expansion of the phase table, once per (size, scale, weights): ptab[Y][X] = table[idx_h[Y]][idx_w[X]] with the two
// offset pairs normalised exactly as the reference normalises them per pixel ((off * 2) / (size - 1), savsr_arch.py:285-287).
// The HR stage then stages a tile's entries with address arithmetic only (no index -> entry dependency in its prologue).
__global__ __launch_bounds__(256) void satu_expand_table_kernel(const float* __restrict__ table, int n_uw, const int* __restrict__ idx_h,
                                                                const int* __restrict__ idx_w, int h, int w, int H, int W, float* __restrict__ ptab) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;           // one 16-B half of an entry per thread
    if (e >= (long long)H * W * 2) return;
    const int q = (int)(e & 1);
    const long long px = e >> 1;
    const int Y = (int)(px / W), X = (int)(px - (long long)Y * W);
    f32x4 v = *reinterpret_cast<const f32x4*>(table + ((long long)idx_h[Y] * n_uw + idx_w[X]) * SAVSR_SATU_TABLE + 4 * q);
    if (q) {
        const float fw1 = (float)(w - 1), fh1 = (float)(h - 1);
        v[0] = (v[0] * 2.f) / fw1; v[1] = (v[1] * 2.f) / fh1; v[2] = (v[2] * 2.f) / fw1; v[3] = (v[3] * 2.f) / fh1;
    }
    *reinterpret_cast<f32x4*>(ptab + e * 4) = v;
}

}  // namespace savsr

using namespace savsr;

#ifdef SAVSR_DIAG
static int g_satu_diag_host = 0;      // != 0: launch the instrumented kernels (tail-projected form only; instrumented library only)

extern "C" int savsr_debug_satu_stamps(int enable) {
    g_satu_diag_host = enable;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_satu_stamps_on), &enable, sizeof(int));
    return e == hipSuccess ? 0 : (int)e;
}
#endif

extern "C" int savsr_satu_hr_occupancy_target(int tail_form) { (void)tail_form; return 1; }
extern "C" int savsr_satu_hr_variants(void) { return HR_VARIANTS; }
extern "C" int savsr_satu_hr_compute_waves(int variant) { return variant < 0 || variant >= HR_VARIANTS ? -1 : hr_compute_waves(variant); }
extern "C" int savsr_satu_hr_rows_per_wave_tile(int tail_form) { return tail_form ? 2 : 1; }

#ifdef SAVSR_DIAG
// Diagnostics: resident workgroups per CU the runtime predicts for the HR / LR kernels (tail-projected form) with `lds_bytes` of dynamic LDS.
extern "C" int savsr_debug_satu_occupancy(int which, int lds_bytes) {
    int n = -1;
    hipError_t e = which == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, satu_hr_kernel<false, 1, 0>, 64 * (hr_compute_waves(0) + hr_producer_waves(0)), (size_t)lds_bytes)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, satu_lr_stream_kernel<1>, 512, (size_t)lds_bytes);
    return e == hipSuccess ? n : -(int)e;
}

#endif
#if SATU_HAS_STAMPS
extern "C" int savsr_debug_read_satu_stamps(long long* host, int nblocks) {
    if (!host || nblocks < 1 || nblocks > SSTAMP_BLOCKS) return fail_arg("debug_read_satu_stamps");
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_satu_stamps), sizeof(long long) * SSTAMP_N * nblocks);
    return e == hipSuccess ? 0 : (int)e;
}
#endif

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

constexpr size_t LR_LDS_BYTES = LR_NPX * LR_XS * sizeof(float) + 2 * LR_PHASE * 16 + 26 * 64 * sizeof(float);     // 150.7 KB
static_assert(LR_LDS_BYTES <= 160 * 1024, "LR stage LDS budget");

namespace savsr {
// every product instantiation's dynamic-LDS attribute on the current device (savsr_prepare_device)
int satu_prepare_device() {
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&satu_lr_stream_kernel<1>), (int)LR_LDS_BYTES, "satu_lr_stage")) return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&satu_lr_stream_kernel<2>), (int)LR_LDS_BYTES, "satu_lr_stage")) return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&satu_hr_kernel<false, 1, 0>), 160 * 1024, "satu_hr")) return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&satu_hr_kernel<false, 1, 1>), 160 * 1024, "satu_hr")) return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&satu_hr_kernel<false, 1, 2>), 160 * 1024, "satu_hr")) return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&satu_hr_kernel<false, 1, 3>), 160 * 1024, "satu_hr")) return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&satu_hr_kernel<false, 1, 4>), 160 * 1024, "satu_hr")) return rc;
    return ensure_dynamic_lds(reinterpret_cast<const void*>(&satu_hr_kernel<false, 2, 0>), 160 * 1024, "satu_hr");
}
}  // namespace savsr

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
    constexpr size_t lds = LR_LDS_BYTES;
    dim3 grid((w + LR_TW - 1) / LR_TW, (h + LR_TH - 1) / LR_TH);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&satu_lr_stream_kernel<NB>), (int)lds, "satu_lr_stage")) return rc;
    hipLaunchKernelGGL((satu_lr_stream_kernel<NB>), grid, dim3(512), lds, static_cast<hipStream_t>(stream), p);
    return check_launch("satu_lr_stream_kernel");
}

extern "C" int savsr_satu_lr_stage(const savsr_satu_weights* wt, const float* x, const float* st, int32_t pix, int32_t row_px, int h,
                                   int w, float* lrcat, void* stream) {
    return lr_stage<2>(wt, x, st, pix, row_px, h, w, lrcat, stream);
}

extern "C" int savsr_satu_lr_stage_tail(const savsr_satu_weights* wt, const float* x, const float* st, int32_t pix, int32_t row_px, int h,
                                        int w, float* lrcat, void* stream) {
    return lr_stage<1>(wt, x, st, pix, row_px, h, w, lrcat, stream);
}

extern "C" int64_t savsr_satu_hr_lds_bytes(int tail_form, int n_table, int tile_rows, int tile_cols32, int lr_rows, int lr_cols) {
    if (tile_rows < 1 || tile_rows > HR_MAX_ROWS || tile_cols32 < 1 || lr_rows < 0 || lr_cols < 0 || n_table < 1) return -1;
    const int nb = tail_form ? 1 : 2;
    const bool small = n_table <= HR_TABLE_LDS;
    return ((int64_t)hr_once_floats(nb, small) + 2 * (int64_t)hr_buf_floats(nb, tile_rows, tile_cols32, lr_rows, lr_cols, small)) * (int64_t)sizeof(float);
}

extern "C" int savsr_satu_expand_table(const float* table, int n_uw, const int32_t* idx_h, const int32_t* idx_w, int h, int w, int H, int W,
                                       float* ptab, void* stream) {
    if (!table || !idx_h || !idx_w || !ptab) return fail_arg("satu_expand_table: null pointer");
    if (h < 2 || w < 2 || H < 1 || W < 1 || n_uw < 1) return fail_arg("satu_expand_table: shape");
    if ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(ptab)) & 15) {
        set_error("satu_expand_table: table / ptab must be 16-byte aligned");
        return SAVSR_E_ALIGN;
    }
    const long long n = (long long)H * W * 2;
    hipLaunchKernelGGL(satu_expand_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), table, n_uw,
                       idx_h, idx_w, h, w, H, W, ptab);
    return check_launch("satu_expand_table_kernel");
}

template <int NB, int VAR>
static int hr_launch(const HrParams& p, size_t lds, int grid, bool diag, hipStream_t st) {
    constexpr int threads = 64 * (hr_compute_waves(VAR) + hr_producer_waves(VAR));
    if constexpr (NB == 1) {
        if (p.seam) {                                               // the row-summed form (savsr_satu_hr_tail_q)
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&satu_hr_kernel<false, 1, VAR, true>), 160 * 1024, "satu_hr")) return rc;
            hipLaunchKernelGGL((satu_hr_kernel<false, 1, VAR, true>), dim3(grid), dim3(threads), lds, st, p);
            return check_launch("satu_hr_kernel");
        }
    }
#ifdef SAVSR_DIAG
    if (diag) {
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&satu_hr_kernel<true, 1, VAR>), 160 * 1024, "satu_hr")) return rc;
        hipLaunchKernelGGL((satu_hr_kernel<true, 1, VAR>), dim3(grid), dim3(threads), lds, st, p);
        return check_launch("satu_hr_kernel");
    }
#endif
    (void)diag;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&satu_hr_kernel<false, NB, VAR>), 160 * 1024, "satu_hr")) return rc;
    hipLaunchKernelGGL((satu_hr_kernel<false, NB, VAR>), dim3(grid), dim3(threads), lds, st, p);
    return check_launch("satu_hr_kernel");
}

template <int NB>
static int hr_stage(const savsr_satu_weights* wt, const float* lrcat, int h, int w, const float* table, int n_uh, int n_uw, const int32_t* idx_h,
                    const int32_t* idx_w, const float* ptab, const float* gyn, const float* gxn, int H, int W, const savsr_satu_tiling* tiling,
                    int32_t* sched, float* out, int64_t out_plane, void* stream, float* seam = nullptr, int64_t seam_floats = 0) {
    if (!satu_weights_ok(wt) || !lrcat || !table || !idx_h || !idx_w || !gyn || !gxn || !out) return fail_arg("satu_hr: null pointer");
    if (h < 2 || w < 2 || H < 1 || W < 1 || n_uh < 1 || n_uw < 1 || out_plane < (int64_t)H * W) return fail_arg("satu_hr: shape (h, w >= 2, out_plane >= H*W required)");
    if (out_plane * 32 * NB * 4 >= ((int64_t)1 << 32)) return fail_arg("satu_hr: output of 4 GiB or more is not supported (32-bit store offsets)");
    const bool small = (int64_t)n_uh * n_uw <= HR_TABLE_LDS;
    if (!small && !ptab) return fail_arg("satu_hr: tables of more than 256 entries need the per-pixel expansion (savsr_satu_expand_table)");
    if ((reinterpret_cast<uintptr_t>(lrcat) | reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(ptab) | reinterpret_cast<uintptr_t>(wt->fusion_b) |
         reinterpret_cast<uintptr_t>(wt->wbe_w) | reinterpret_cast<uintptr_t>(gyn) | reinterpret_cast<uintptr_t>(gxn) | reinterpret_cast<uintptr_t>(idx_h) |
         reinterpret_cast<uintptr_t>(idx_w)) & 15) {
        set_error("satu_hr: lrcat / table / ptab / fusion_b / wbe_w / gyn / gxn / idx_h / idx_w must be 16-byte aligned");
        return SAVSR_E_ALIGN;
    }
    HrParams p;
    p.wt = *wt; p.lrcat = lrcat; p.h = h; p.w = w; p.table = table; p.n_table = small ? n_uh * n_uw : (1 << 30); p.n_uw = n_uw;
    p.idx_h = idx_h; p.idx_w = idx_w; p.ptab = ptab;
    p.gyn = gyn; p.gxn = gxn; p.H = H; p.W = W; p.out = out; p.out_plane = out_plane; p.sched = sched;
    p.seam = seam; p.nseg = (W + 31) / 32;
    if (seam && (seam_floats < (int64_t)H * p.nseg * 18 || (reinterpret_cast<uintptr_t>(seam) & 3))) return fail_arg("satu_hr: seam buffer (>= H * ceil(W / 32) * 18 floats)");
    p.ty = 8; p.txw = 1; p.lrh = 0; p.lrw = 0; p.omin_x = 0.f; p.omin_y = 0.f;      // default: no window staging, gathers from global
    p.step_x = (float)w / (float)W; p.step_y = (float)h / (float)H;
    int variant = 0;
    if (tiling) {
        if (tiling->variant < 0 || tiling->variant >= HR_VARIANTS || (NB != 1 && tiling->variant != 0))
            return fail_arg("satu_hr: tiling.variant (0 .. savsr_satu_hr_variants() - 1; the standalone form has variant 0 only)");
        variant = tiling->variant;
        if (tiling->tile_rows < 4 || tiling->tile_rows > HR_MAX_ROWS || (tiling->tile_rows & 3) || tiling->tile_cols32 < 1 || tiling->tile_cols32 > 8 ||
            tiling->lr_rows < 0 || tiling->lr_cols < 0)
            return fail_arg("satu_hr: tiling (tile_rows a multiple of 4 in 4..64, tile_cols32 in 1..8)");
        p.ty = tiling->tile_rows; p.txw = tiling->tile_cols32; p.lrh = tiling->lr_rows; p.lrw = tiling->lr_cols;
        p.omin_x = tiling->off_min_x; p.omin_y = tiling->off_min_y;
        if (tiling->step_x > 0.f && tiling->step_y > 0.f) { p.step_x = tiling->step_x; p.step_y = tiling->step_y; }
        if (p.lrh == 0 || p.lrw < 2) { p.lrh = 0; p.lrw = 0; }
        if (p.lrh > h || p.lrw > w) return fail_arg("satu_hr: tiling (the LR window must fit the LR image: lr_rows <= h, lr_cols <= w)");
    }
    p.ntx = (W + 32 * p.txw - 1) / (32 * p.txw);
    p.nty = (H + p.ty - 1) / p.ty;
    const size_t lds = ((size_t)hr_once_floats(NB, small) + 2 * (size_t)hr_buf_floats(NB, p.ty, p.txw, p.lrh, p.lrw, small)) * sizeof(float);
    if (lds > 160 * 1024) return fail_arg("satu_hr: staged windows + tile tables exceed 160 KiB of LDS");
#ifdef SAVSR_DIAG
    const bool diag = NB == 1 && g_satu_diag_host;
#else
    const bool diag = false;
#endif
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    long long ntile = (long long)p.ntx * p.nty;
    int grid = ncu;                                                    // one fat workgroup per CU
    if (grid > ntile) grid = (int)ntile;
    grid = (grid + 7) & ~7;                                            // the XCD split needs a multiple of 8
    hipStream_t st = static_cast<hipStream_t>(stream);
    if constexpr (NB == 1) {
        if (variant == 1) return hr_launch<1, 1>(p, lds, grid, diag, st);
        if (variant == 2) return hr_launch<1, 2>(p, lds, grid, diag, st);
        if (variant == 3) return hr_launch<1, 3>(p, lds, grid, diag, st);
        if (variant == 4) return hr_launch<1, 4>(p, lds, grid, diag, st);
    }
    return hr_launch<NB, 0>(p, lds, grid, diag, st);
}

extern "C" int savsr_satu_hr_upsample(const savsr_satu_weights* wt, const float* lrcat, int h, int w, const float* table, int n_uh, int n_uw,
                                      const int32_t* idx_h, const int32_t* idx_w, const float* ptab, const float* gyn, const float* gxn, int H, int W,
                                      const savsr_satu_tiling* tiling, int32_t* sched, float* out, int64_t out_plane, void* stream) {
    return hr_stage<2>(wt, lrcat, h, w, table, n_uh, n_uw, idx_h, idx_w, ptab, gyn, gxn, H, W, tiling, sched, out, out_plane, stream);
}

extern "C" int savsr_satu_hr_tail(const savsr_satu_weights* wt, const float* lrcat, int h, int w, const float* table, int n_uh, int n_uw,
                                  const int32_t* idx_h, const int32_t* idx_w, const float* ptab, const float* gyn, const float* gxn, int H, int W,
                                  const savsr_satu_tiling* tiling, int32_t* sched, float* out, int64_t out_plane, void* stream) {
    return hr_stage<1>(wt, lrcat, h, w, table, n_uh, n_uw, idx_h, idx_w, ptab, gyn, gxn, H, W, tiling, sched, out, out_plane, stream);
}

extern "C" int savsr_satu_hr_tail_q(const savsr_satu_weights* wt, const float* lrcat, int h, int w, const float* table, int n_uh, int n_uw,
                                    const int32_t* idx_h, const int32_t* idx_w, const float* ptab, const float* gyn, const float* gxn, int H, int W,
                                    const savsr_satu_tiling* tiling, int32_t* sched, float* q9, int64_t q_plane, float* seam, int64_t seam_floats,
                                    void* stream) {
    if (!seam) return fail_arg("satu_hr_tail_q: null seam planes");
    if (!tiling) return fail_arg("satu_hr_tail_q: needs a tiling (the lane = pixel kernels)");
    return hr_stage<1>(wt, lrcat, h, w, table, n_uh, n_uw, idx_h, idx_w, ptab, gyn, gxn, H, W, tiling, sched, q9, q_plane, stream, seam, seam_floats);
}
