// 3x3 convolution (pad 1, stride 1) on channel-last fp32 feature maps by Winograd F(2x2, 3x3) on the bf16 matrix cores
// with split-precision operands ("bf16x3", as conv_mfma.hip), for convs with cout % 64 == 0 and src_ch % 16 == 0.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 kernel g
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// 16 transform positions xi = 4 i + j, each an independent [64 co x 16 ci] x [16 ci x tiles] product: 16 MFMA products per
// 4 outputs instead of 36 (2.25x fewer), paid for with the input transform (VALU), 4x larger Winograd-domain accumulators
// and an output transform.  The direct kernel is power-managed down to ~1.5 GHz by its MFMA + LDS density;
// this one trades matrix work for vector / LDS work the chip has headroom for.
//
// Workgroup = 8 waves, tile = 8 rows x 32 cols of output pixels (64 Winograd tiles) x 64 output channels, K phases of 16
// input channels.  Wave w owns transform ROW i = w & 3 (positions 4i .. 4i+3) of tile group tt = w >> 2 (32 Winograd tiles =
// 4 output rows): 4 positions x 2 channel blocks = 8 accumulators of 32 x 32 (128 registers), 24 MFMAs per phase.
//   * weights: every position's A fragments are used by ONE transform row, so they never touch LDS -- each wave loads its 16 KiB
//     per phase straight from the packed image (already in lane order) into registers, one phase ahead of use;
//   * activations: the 10 x 34-pixel fp32 patch of a phase goes global -> LDS by LDS-DMA (ring of 3 slots, 80-B pixel pitch:
//     conflict-free reads), two phases ahead; each wave transforms exactly the operands IT consumes (row i of B^T d, then the
//     four columns) from the patch into a wave-private 8-KiB B-operand image: no block barrier between transform and MFMAs;
//   * ONE barrier per phase (it publishes a ring slot and frees another).  Waves 0-3 run "transform, MFMAs" after the barrier,
//     waves 4-7 "MFMAs, next transform": the two waves of a SIMD (w, w + 4) are half a phase apart, so one's vector / LDS
//     work runs under the other's MFMAs;
//   * output transform: columns in registers (4 positions -> 2), rows across the four waves of a tile group through LDS, 16
//     output channels at a time, then the fused epilogue of the direct kernel (bias / activation / per-pixel mask / two
//     residuals / global-average-pool partials), whole 64-B pieces of pixel records per 4 lanes.
// Replaces: the same nn.Conv2d / F.conv2d calls as conv_mfma.hip (include/savsr_hip.h), selected per descriptor (`algo`).
//
// STATUS (round 2): an opt-in experiment, parity-tested (tests/test_gpu_conv_wino.py) and NOT selected by the engine.  Measured
// on MI355X, 6 x 128->64 at 180x320: 196-228 us against the direct kernel's 150 us; 64->64: 22.7-25.4 against 19.1 us; rounding
// error 2-2.5x the direct kernel's.  Why it loses (ablations with -DWINO_EXP): a position's weights are
// used by one transform row only, so there is no cross-wave reuse of a weight byte -- every workgroup streams 64 KB (128 KB
// with both tile groups loading their copy) of weights per 16-channel phase from L2, 3.6x the direct kernel's weight bytes per
// output pixel; the Winograd-domain accumulators (256 KB per workgroup) cap the tile at 8 x 32 pixels, so nothing amortises
// them.  With transform, MFMAs and epilogue all compiled out, the loads and barriers alone take 100 us.
#include "conv_common.hpp"

// Timing experiments (results invalid; never set in a shipped build): 1 = no input transform, 2 = no MFMAs, 4 = no output
// transform / epilogue, 8 = no patch DMAs in the loop, 16 = no weight loads.
#ifndef WINO_EXP
#define WINO_EXP 0
#endif

namespace savsr {

constexpr int WN_ROWS = 8, WN_COLS = 32;                   // output pixels per tile
constexpr int WN_PR = WN_ROWS + 2, WN_PC = WN_COLS + 2;    // input patch
constexpr int WN_NPIX = WN_PR * WN_PC;                     // 340
constexpr int WN_PITCH = 5;                                // 16-B units per patch pixel: 16 channels + 16 B pad
constexpr int WN_RAW_UNITS = WN_NPIX * WN_PITCH;           // 1700
constexpr int WN_RAW_DMAS = (WN_RAW_UNITS + 63) / 64;      // 27 DMAs of 1 KiB
constexpr int WN_RAW_SLOT = WN_RAW_DMAS * 64;              // units per ring slot
constexpr int WN_RING = 3;
constexpr int WN_DMA_PER_WAVE = (WN_RAW_DMAS + 3) / 4;     // 7: waves 4-7 issue them (see the phase loop)
constexpr int WN_V_WAVE = 4 * 2 * 64;                      // units of a wave's B-operand image: [j][part][lane]
constexpr int WN_V_UNITS = 8 * WN_V_WAVE;                  // 64 KiB
constexpr int WN_E_PITCH = 20;                             // floats per (row i, column c, tile) in the output-transform buffer: 16 co + pad
constexpr int WN_POOL_FLOATS = 8 * 16;
constexpr size_t WN_LDS = 16ull * (WN_V_UNITS + WN_RING * WN_RAW_SLOT) + 4ull * WN_POOL_FLOATS;
static_assert(4 * 2 * 64 * WN_E_PITCH * 4 <= WN_V_UNITS * 16, "output-transform buffer aliases the B-operand images");
static_assert(WN_LDS <= 160 * 1024, "LDS budget");

// 16 B per lane global -> LDS (M0 = the instruction's LDS base; lane l lands at base + 16 l) with a SCALAR global base and a
// 32-bit per-lane byte offset.  hipcc does not count these loads: the issuing wave waits with s_waitcnt vmcnt(0) in front of
// the publishing barrier.
#define WN_DMA16(sbase, voff, lds_base) do { \
        const unsigned dst_ = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_base)); \
        const unsigned long long sb_ = (unsigned long long)(uintptr_t)(sbase); \
        const unsigned sb_lo_ = __builtin_amdgcn_readfirstlane((unsigned)sb_), sb_hi_ = __builtin_amdgcn_readfirstlane((unsigned)(sb_ >> 32)); \
        unsigned keep_; \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(voff), "s"((((unsigned long long)sb_hi_) << 32) | sb_lo_), "s"(dst_) : "memory"); \
    } while (0)

// ---- weight image ---------------------------------------------------------------------------------------------------------
// [cob][chunk][i 4][j 4][ct 2][part 2][lane 64][8 bf16]: lane = kh * 32 + row holds U[4i+j][co = 64 cob + 32 ct + row][ci = 16 chunk + 8 kh ..+7],
// U = G g G^T of the fp32 kernel g given as [cout][9][cin] (tap-major, channel fastest).  One thread per (co, 8 ci, i).
__global__ __launch_bounds__(256) void conv_wino_pack_kernel(const float* __restrict__ w, int cout, int cin, bf16x8* __restrict__ img) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n8 = cin / 8;
    if (idx >= cout * n8 * 4) return;
    const int i = idx & 3, o8 = (idx >> 2) % n8, co = (idx >> 2) / n8;
    float t[3][8];                                        // row i of G g: over ky
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        float g0[8], g1[8], g2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            g0[e] = w[((long long)co * 9 + 0 * 3 + kx) * cin + o8 * 8 + e];
            g1[e] = w[((long long)co * 9 + 1 * 3 + kx) * cin + o8 * 8 + e];
            g2[e] = w[((long long)co * 9 + 2 * 3 + kx) * cin + o8 * 8 + e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
            t[kx][e] = i == 0 ? g0[e] : (i == 1 ? 0.5f * (g0[e] + g1[e] + g2[e]) : (i == 2 ? 0.5f * (g0[e] - g1[e] + g2[e]) : g2[e]));
    }
    const int cob = co >> 6, ct = (co >> 5) & 1, row = co & 31, chunk = o8 >> 1, kh = o8 & 1, nchunk = cin / 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float u = j == 0 ? t[0][e] : (j == 1 ? 0.5f * (t[0][e] + t[1][e] + t[2][e]) : (j == 2 ? 0.5f * (t[0][e] - t[1][e] + t[2][e]) : t[2][e]));
            const __bf16 h = (__bf16)u;
            hi[e] = h;
            lo[e] = (__bf16)(u - (float)h);
        }
        const long long unit = ((((((long long)cob * nchunk + chunk) * 4 + i) * 4 + j) * 2 + ct) * 2) * 64 + kh * 32 + row;
        img[unit] = hi;
        img[unit + 64] = lo;
    }
}

// ---- the conv ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void conv_wino_kernel(const MultiConvParams mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* vimg = reinterpret_cast<bf16x8*>(smem_raw);                                  // [8 waves][j 4][part 2][lane 64]
    f32x4* raw = reinterpret_cast<f32x4*>(smem_raw) + WN_V_UNITS;                        // [3 slots][340 px][5 units]
    float* ebuf = reinterpret_cast<float*>(smem_raw);                                    // output transform: aliases vimg
    float* poolbuf = reinterpret_cast<float*>(smem_raw + 16ull * (WN_V_UNITS + WN_RING * WN_RAW_SLOT));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 3, tt = wave >> 2;              // transform row, tile group (= skew group: SIMD s runs waves s and s + 4)
    const int H = mp.h, W = mp.w;
    const int tiles_per_cob = mp.ntx * mp.nty;
    const int total = mp.nconv * mp.ncob * tiles_per_cob;
    const int nchunk = mp.nchunk;

    struct TileInfo { int conv, cob, x0, y0, tx, ty; };
    auto decode = [&](int tile) {
        const int cc = tile / tiles_per_cob, rem = tile - cc * tiles_per_cob;
        TileInfo ti;
        ti.ty = rem / mp.ntx;
        ti.tx = rem - ti.ty * mp.ntx;
        ti.conv = cc / mp.ncob;
        ti.cob = cc - ti.conv * mp.ncob;
        ti.x0 = ti.tx * WN_COLS;
        ti.y0 = ti.ty * WN_ROWS;
        return ti;
    };

    // ---- patch DMA cursor (two phases ahead of the compute cursor), driven by waves 4-7 ----
    // DMA k = (wave - 4) + 4 q of a slot covers units u = 64 k + lane = (patch pixel u / 5, 16-B piece u % 5); d_code packs
    // (patch row | patch column << 4 | piece << 10), piece 4 = nothing to load (a pixel's pad piece / past the last pixel).
    // Pixels outside the image get zeros by a plain LDS store.
    int d_code[WN_DMA_PER_WAVE];
#pragma unroll
    for (int q = 0; q < WN_DMA_PER_WAVE; ++q) {
        const int k = (wave & 3) + 4 * q, u = k * 64 + lane;
        const int px = u / WN_PITCH, pr = px / WN_PC;
        d_code[q] = pr | ((px - pr * WN_PC) << 4) | (((k < WN_RAW_DMAS && px < WN_NPIX) ? u - px * WN_PITCH : 4) << 10);
    }
    int d_tile = blockIdx.x, d_chunk = 0, d_conv = 0, d_y0 = 0, d_x0 = 0, d_count = 0;      // d_count: phases issued (ring slot = d_count % 3)
    auto dma_begin_tile = [&]() {
        const TileInfo ti = decode(d_tile);
        d_conv = ti.conv;
        d_y0 = ti.y0 - 1;
        d_x0 = ti.x0 - 1;
    };
    auto dma_issue = [&]() {                              // the cursor's phase -> ring slot d_count % 3; then advance
        if (d_tile < total) {
            const int cb = d_chunk * 16, sidx = cb / mp.src_ch, cl = cb - sidx * mp.src_ch;
            const float* base = mp.c[d_conv].src[sidx];
            const int pix = mp.c[d_conv].src_pix[sidx];
            f32x4* slot = raw + (d_count % WN_RING) * WN_RAW_SLOT;
#pragma unroll
            for (int q = 0; q < WN_DMA_PER_WAVE; ++q) {
                const int k = (wave & 3) + 4 * q;
                if (k < WN_RAW_DMAS) {                    // scalar
                    const int code = d_code[q], sub = code >> 10;
                    const int gy = d_y0 + (code & 15), gx = d_x0 + ((code >> 4) & 63);
                    const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                    if (sub != 4) {
                        if (inside) WN_DMA16(base, (unsigned)(((gy * W + gx) * pix + cl + 4 * sub) * 4), slot + k * 64);
                        else slot[k * 64 + lane] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            ++d_count;
            if (++d_chunk == nchunk) {
                d_chunk = 0;
                d_tile += gridDim.x;
                if (d_tile < total) dma_begin_tile();
            }
        }
    };

    // ---- transform of one phase: ring slot -> this wave's B-operand image ----
    // item e = lane + 64 m (m = 0, 1): channel quad q = e & 3, tile n = e >> 2 of the wave's 32.  Row i of B^T d is dA + sg dB.
    const int rA = wi == 0 ? 0 : (wi == 2 ? 2 : 1), rB = wi == 0 ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float sg = wi == 1 ? 1.f : -1.f;
    bf16x8* vme = vimg + wave * WN_V_WAVE;
    auto transform = [&](int slot_idx) {
        const f32x4* rs = raw + slot_idx * WN_RAW_SLOT;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int e = lane + 64 * m, q = e & 3, n = e >> 2;
            const int trow = 2 * tt + (n >> 4), tcol = n & 15;
            const f32x4* pa = rs + ((2 * trow + rA) * WN_PC + 2 * tcol) * WN_PITCH + q;
            const f32x4* pb = rs + ((2 * trow + rB) * WN_PC + 2 * tcol) * WN_PITCH + q;
            f32x4 t[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) t[b] = pa[b * WN_PITCH] + sg * pb[b * WN_PITCH];
            const f32x4 v[4] = {t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]};
            bf16x4* dst = reinterpret_cast<bf16x4*>(vme + (q >> 1) * 32 + n) + (q & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 hi, lo;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const __bf16 hh = (__bf16)v[j][c];
                    hi[c] = hh;
                    lo[c] = (__bf16)(v[j][c] - (float)hh);
                }
                dst[(j * 2 + 0) * 64 * 2] = hi;
                dst[(j * 2 + 1) * 64 * 2] = lo;
            }
        }
    };

    // ---- prologue: patches of the first two phases ----
    int tile = blockIdx.x;
    if (tt == 1) {
        if (tile < total) dma_begin_tile();
        dma_issue();
        dma_issue();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    int count = 0;                                        // compute cursor: phases done (ring slot = count % 3)
    for (; tile < total; tile += gridDim.x) {
        const TileInfo cur = decode(tile);
        const ConvParams& p = mp.c[cur.conv];
        f32x16 acc[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][ct][r] = 0.f;
        const float* wbase = reinterpret_cast<const float*>(p.wimg) + ((long long)cur.cob * nchunk * 4 + wi) * 4096;     // (scalar)

        for (int chunk = 0; chunk < nchunk; ++chunk, ++count) {
            if (tt == 0) __syncthreads();                 // scalar: waves 0-3 pass the phase's barrier before their transform ...
            f32x4 a[16];                                  // this phase's weights: [j][ct][part], 1 KiB per load
            const float* wp = wbase + (long long)chunk * 16384;
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = (WINO_EXP & 16) ? f32x4{1.f, 1.f, 1.f, (float)k} : ldg4(wp, (unsigned)(lane * 16 + k * 1024));
            if (!(WINO_EXP & 1)) transform(count % WN_RING);
            if (tt == 1) {                                // ... waves 4-7 between their transform and their MFMAs; they also
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drive the patch ring: the DMAs of phase + 1 (issued a whole
                __syncthreads();                          // "MFMAs, transform" ago) have landed, phase + 2 goes out into the
                if (!(WINO_EXP & 8)) dma_issue();         // slot every wave has just finished reading
            }
            if (WINO_EXP & 2) {
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("" :: "v"(a[k]));
            } else
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 bh = vme[(j * 2 + 0) * 64 + lane], bl = vme[(j * 2 + 1) * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, a[(j * 2 + ct) * 2 + 0]), al = __builtin_bit_cast(bf16x8, a[(j * 2 + ct) * 2 + 1]);
                    acc[j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j][ct], 0, 0, 0);
                    acc[j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j][ct], 0, 0, 0);
                    acc[j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j][ct], 0, 0, 0);
                }
            }
        }

        // ---- output transform + epilogue ----
        // columns, in registers: yc[c] = sum_j A^T[c][j] M[i][j]
        f32x16 yc[2][2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            yc[0][ct] = acc[0][ct] + acc[1][ct] + acc[2][ct];
            yc[1][ct] = acc[1][ct] - acc[2][ct] - acc[3][ct];
        }
        const float* e_bias = p.bias;
        const float* e_mul = p.mul_px;
        const float* e_r1 = p.res1;
        const float* e_r2 = p.res2;
        float* e_out = p.out;
        float* e_pool = p.pool;
        const int e_act = p.act, e_opix = p.out_pix, e_r1pix = p.res1_pix, e_r2pix = p.res2_pix, e_pstride = p.pool_stride;
        const float e_slope = p.slope, e_r2s = p.res2_scale;
        const int half = lane >> 5, n = lane & 31;
        // reader item: channel quad cq = tid & 3, Winograd tile T = (tid >> 2) & 63, output column c = tid >> 8
        const int cq = tid & 3, T = (tid >> 2) & 63, oc = tid >> 8;
        const int px_x = cur.x0 + 2 * (T & 15) + oc, px_y = cur.y0 + 2 * (T >> 4);
        const bool x_ok = px_x < W, y0_ok = px_y < H, y1_ok = px_y + 1 < H;
        if (WINO_EXP & 4) {
            asm volatile("" :: "v"(yc[0][0][0]), "v"(yc[1][0][0]), "v"(yc[0][1][0]), "v"(yc[1][1][0]));
        } else
#pragma unroll
        for (int k = 0; k < 4; ++k) {                     // 16 output channels at a time: block ct = k >> 1, register groups 2 (k & 1), + 1
            __syncthreads();                              // the B-operand images (first pass) / the previous pass's reads are done
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int g = 2 * (k & 1) + gg;
                    const f32x16& y = yc[c][k >> 1];
                    const f32x4 v = {y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3]};
                    *reinterpret_cast<f32x4*>(ebuf + ((wi * 2 + c) * 64 + 32 * tt + n) * WN_E_PITCH + 8 * gg + 4 * half) = v;
                }
            __syncthreads();
            f32x4 r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const f32x4*>(ebuf + ((i * 2 + oc) * 64 + T) * WN_E_PITCH + 4 * cq);
            f32x4 o[2] = {r[0] + r[1] + r[2], r[1] - r[2] - r[3]};           // output rows 2 trow, 2 trow + 1
            const int co = cur.cob * 64 + 16 * k + 4 * cq;
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (e_bias) b4 = ldg4(e_bias, 4u * (unsigned)co);
            f32x4 ps = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const bool ok = x_ok && (rr == 0 ? y0_ok : y1_ok);
                if (!ok) continue;
                const int pidx = (px_y + rr) * W + px_x;
                f32x4 v = o[rr] + b4;
                if (e_act == SAVSR_ACT_RELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                } else if (e_act == SAVSR_ACT_LRELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * e_slope;
                } else if (e_act == SAVSR_ACT_SIGMOID) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = sigmoidf_(v[q]);
                }
                if (e_mul) v = v * ldg1(e_mul, (unsigned)pidx);
                if (e_r1) v = v + ldg4(e_r1, 4u * (unsigned)(pidx * e_r1pix + co));
                if (e_r2) v = v + e_r2s * ldg4(e_r2, 4u * (unsigned)(pidx * e_r2pix + co));
                stg4(e_out, 4u * (unsigned)(pidx * e_opix + co), v);
                ps = ps + v;
            }
            if (e_pool) {
                // AdaptiveAvgPool2d(1) partials (savsr_arch.py:146,515): lanes with equal l % 4 hold the same channel quad; the
                // waves are summed in wave order through LDS (deterministic); one row per tile = 8-row band x 32-column block,
                // the numbering of savsr_conv_pool_blocks
#pragma unroll
                for (int s = 4; s < 64; s <<= 1) {
                    ps[0] += __shfl_xor(ps[0], s, 64); ps[1] += __shfl_xor(ps[1], s, 64);
                    ps[2] += __shfl_xor(ps[2], s, 64); ps[3] += __shfl_xor(ps[3], s, 64);
                }
                if (lane < 4) *reinterpret_cast<f32x4*>(poolbuf + wave * 16 + 4 * lane) = ps;
                __syncthreads();
                if (tid < 16) {
                    float sacc = 0.f;
#pragma unroll
                    for (int wv = 0; wv < 8; ++wv) sacc += poolbuf[wv * 16 + tid];
                    stg1(e_pool, (unsigned)((cur.ty * mp.ntx + cur.tx) * e_pstride + cur.cob * 64 + 16 * k + tid), sacc);
                }
            }
        }
        __syncthreads();                                  // the output-transform buffer becomes the B-operand images again
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int launch_conv_wino(const MultiConvParams& mp, hipStream_t st) {
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_wino_kernel), (int)WN_LDS, "conv_wino")) return rc;
    const int total = mp.nconv * mp.ncob * mp.ntx * mp.nty;
    const int grid = total < CONV_PERSISTENT_BLOCKS ? total : CONV_PERSISTENT_BLOCKS;
    hipLaunchKernelGGL(conv_wino_kernel, dim3(grid), dim3(512), WN_LDS, st, mp);
    return check_launch("conv_wino_kernel");
}

}  // namespace savsr

using namespace savsr;

extern "C" int64_t savsr_conv_wino_packed_elems(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cout % 64 || cin % 16) return -1;
    return (int64_t)cout * cin * 16 * 2;                 // bf16 elements: 16 positions x (hi, lo)
}

extern "C" int savsr_conv_wino_pack(const float* w_ohwi, int cout, int cin, void* image, void* stream) {
    if (!w_ohwi || !image) return fail_arg("conv_wino_pack: null pointer");
    if (savsr_conv_wino_packed_elems(cout, cin) < 0) return fail_arg("conv_wino_pack: cout must be a multiple of 64, cin of 16");
    if (reinterpret_cast<uintptr_t>(image) & 15) {
        set_error("conv_wino_pack: image must be 16-byte aligned");
        return SAVSR_E_ALIGN;
    }
    const int n = cout * (cin / 8) * 4;
    hipLaunchKernelGGL(conv_wino_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w_ohwi, cout, cin,
                       reinterpret_cast<bf16x8*>(image));
    return check_launch("conv_wino_pack_kernel");
}
