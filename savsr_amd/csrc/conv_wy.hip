// 3x3 convolution (stride 1, zero pad 1, cout a multiple of 64) on channel-last fp32 feature maps as a 1-D WINOGRAD F(2,3) ALONG Y
// implicit GEMM on the bf16 matrix cores with split-precision ("bf16x3") operands -- the algebraic form of conv_mfma.hip's direct
// kernel with 2/3 of its matrix work (savsr_conv_desc.algo = SAVSR_CONV_WINOGRAD_Y, round 4).
//
//   out rows (Y, Y+1) from input rows d0..d3 = Y-1 .. Y+2, per horizontal tap kx and input channel:
//     V0 = d0 - d2,  V1 = d1 + d2,  V2 = d2 - d1,  V3 = d1 - d3                      (input transform, fp32, then (hi, lo) bf16 split)
//     U0 = g0,  U1 = (g0 + g1 + g2) / 2,  U2 = (g0 - g1 + g2) / 2,  U3 = g2            (weights g_ky; float64 on the host, then split)
//     M_i = sum_{kx, ci} U_i[co][ci][kx] * V_i[ci](x + kx - 1)                           (4 "positions" x 3 kx = 12 taps instead of 2 x 9)
//     out(Y) = M0 + M1 + M2,  out(Y+1) = M1 - M2 - M3                                    (output transform, in registers)
//
// Why along y, and how it maps to a CU.  Workgroup = 8 waves = a 16-row x 32-pixel x 64-channel output tile (round 6: the image's
// last h % 16 <= 8 rows can go as STRIP tiles -- 1 / 2 / 4 row pairs x 8 / 4 / 2 segments of 32 pixels; see `decode`); wave w owns the row
// PAIR (2w, 2w+1), so its four transformed rows are built from its own four input rows and live in a WAVE-PRIVATE LDS region: no
// activation is shared between waves (a Winograd form along x, or the 2-D form of round 2, shares transformed rows / needs its
// weights per transform position in registers).  Only the weights are shared: per 16-channel phase the 12 taps are a 48 KB slab
// in two halves (positions {0,1} | {2,3}), each half re-filled by LDS-DMA while the other is being read and published by its own
// barrier -- 2 barriers per phase, none for the activations: positions {0,1} of the next phase are written while {2,3} of this
// one are read and vice versa, so the 70 KB V image needs no double buffer (LDS: 70 + 48 + 37 KB epilogue slices = 152 KB).
// Per (position, kx) step a wave reads 4 weight + 2 activation fragments and issues 6 MFMAs; 72 MFMAs per wave and phase against
// 108 in the direct kernel for the same outputs; 8 accumulators (4 positions x 2 channel blocks = 128 registers).
// Measured before it was built (tools/micro/conv_skel.hip, K-loop skeletons with the real staging traffic, one MI355X, steady
// state): 897 us direct vs 753 us for this form (-16 %): the launch is power-limited, the matrix instructions are most of the
// energy, and fewer of them buy clock (1.68 -> 1.80 GHz) on top of cycles (7.85 k -> 7.0 k per phase); fewer LDS fragment bytes at
// equal MFMAs bought nothing (same skeleton file, "B reuse").
//
// Numerics: the same split products and fp32 accumulation as the direct kernel; the transforms add one fp32 rounding on the
// activations (d0 - d2 ...) and the cancellation of the output transform: measured max-abs error 1.3-1.6 x the direct kernel's
// (tests/test_gpu_kernels.py::test_conv2d_winograd_y), inside the same per-kernel bound.
//
// Replaces (when selected by the engine): the static-weight 3x3 convs of savsr_arch.py:388-397,429-442,480-483,541-543,567,723,733.
#include "conv_common.hpp"

#include <type_traits>

// Diagnostic build switch (tools/ab_conv.sh "-DWY_STAMPS=1"; never set in the shipped library): per-wave s_memtime section stamps, read back with
// savsr_debug_read_wy_stamps -- 1 = the K loop's halves, 2 = the epilogue's pieces.  The ablation knobs the round-4 tables of DESIGN.md section 4c
// were measured with (no fragment reads / staging stores / row loads / residual loads / output stores / epilogue, nt row loads, static wave
// priority) are archived as tools/experiments/conv_wy_timing_knobs.patch.
#ifndef WY_STAMPS
#define WY_STAMPS 0
#endif
#ifndef WY_VALU
#define WY_VALU 6                 // vector instructions the scheduler may put behind every MFMA of a step
#endif
#ifndef WY_FAST_EPILOGUE
#define WY_FAST_EPILOGUE 1        // 1: launches whose convs all have a max-form activation, no mask and no second residual run the straight-line epilogue (conv_wy_kernel<true>)
#endif
#ifndef WY_BUF
#define WY_BUF 1                  // 1: activation rows by buffer loads (uniform 64-bit row base in the resource, one constant per-lane column offset, out-of-range
#endif                            //    lanes / rows answered with zeros by the bounds check): no per-load address arithmetic on the vector unit.  0: flat global loads (round 4)

namespace savsr {

namespace wy {
constexpr int TH = 16, TW = 32, NTHR = 512, COT = 64;
[[maybe_unused]] constexpr int IC = TW + 2;         // pixels of a staged row: the 32 of the tile + one halo column on each side
#ifndef WY_ICP
#define WY_ICP 36                                  // LDS pitch (16-B units) of one channel-octet row of a plane: 34 pixels + 2 pad.  With the natural 34 the
#endif                                             // staging stores of the two octets of a pixel quad-set fall 136 dwords apart = 8 banks: a 2-way conflict in every pass
constexpr int ICP = WY_ICP;
constexpr int V_PLANE = 2 * ICP;                   // 16-B units of one (half, row, part) plane: [q 2][px ICP]
constexpr int V_WAVE = 2 * 2 * 2 * V_PLANE;        // [hf][vr][part] planes per wave = 576 units
constexpr int W_HALF = 6 * 2 * 2 * 64;             // [s = vr * 3 + kx][t][part][lane] = 1536 units = 24 KiB
constexpr int W_PHASE = 2 * W_HALF;                // 48 KiB per (cob, chunk)
constexpr int EPS = 36;                            // floats per pixel in an epilogue slice
constexpr int LDS_UNITS = 8 * V_WAVE + 2 * W_HALF; // 7424 units = 118 784 B
constexpr size_t LDS_BYTES = 16ull * LDS_UNITS + 4ull * 8 * 32 * EPS;      // + 36 864 B of epilogue slices = 155 648 B
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
}  // namespace wy

__device__ __attribute__((aligned(16))) const float g_wy_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // what padding lanes load

#if WY_STAMPS
// instrumented build: per (workgroup, wave) accumulated s_memtime of 8 sections (savsr_debug_read_wy_stamps):
// 0 half A steps 0-4 | 1 half A wait + barrier | 2 half A step 5 rest | 3 half B steps 0-4 | 4 half B wait + barrier | 5 half B step 5 rest | 6 transform + epilogue | 7 total
__device__ long long g_wy_stamps[256 * 8 * 8];
#define WY_KSEC(i) (WY_STAMPS == 2 ? 0 : (i))
#define WY_MARK(i) do { const long long t_now = (long long)__builtin_amdgcn_s_memtime(); wy_sec[i] += t_now - wy_prev; wy_prev = t_now; } while (0)
#if WY_STAMPS == 2             // the epilogue in pieces instead of the K-loop halves: sections 0 = K loop, 1 = epilogue loads issued + output transform, 2..5 = the four groups
#define WY_EMARK(i) WY_MARK(i)
#endif
#else
#define WY_KSEC(i) (i)
#define WY_MARK(i) do { } while (0)
#endif

__device__ __forceinline__ void wy_split_store(const f32x4& v, bf16x4* hi_dst, bf16x4* lo_dst) {
    bf16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 hh = (__bf16)v[j];
        hi[j] = hh;
        lo[j] = (__bf16)(v[j] - (float)hh);
    }
    *hi_dst = hi;
    *lo_dst = lo;
}

// FAST: every conv of the launch has an activation of the max form (none / ReLU / LeakyReLU with 0 <= slope <= 1), no per-pixel mask and no second
// residual -- all conv launches of the network but OSAdapt's final one.  The epilogue is then straight-line code: no activation dispatch, no
// joins whose phi copies cost 8 v_mov each (the generic epilogue carries ~480 of them and is half of the kernel's vector instructions).
template <bool FAST>
__global__ __launch_bounds__(512) void conv_wy_kernel(const MultiConvParams mp) {
    using namespace wy;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* smem = reinterpret_cast<bf16x8*>(smem_raw);       // [8 waves][V_WAVE] | [2][W_HALF] | epilogue slices
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, px = lane & 31;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int tiles_per_cob = mp.wy_tiles;
    const int total = mp.nconv * mp.ncob * tiles_per_cob;
    const int H = mp.h, W = mp.w;

    // A tile is 8 wave tasks, each a (row pair, 32-pixel segment) of the image with everything but the weight slab private to its wave.  Full tiles
    // stack the 8 row pairs of 16 image rows over ONE segment.  When H % 16 leaves at most 8 rows, the image's last rows are walked by STRIP tiles
    // (round 6): the 2^l row pairs the strip needs, side by side over 8 >> l segments -- at 180 rows 3 strip tiles instead of 10 tiles whose waves
    // 2 .. 7 stage zeros and skip their matrix work (113 instead of 120 tiles per conv and channel block).  Every wave task computes what it computed
    // in a full tile: results bit for bit.  All of this is scalar arithmetic on wave_s and kernel arguments.
    struct TileInfo { int conv, cob, x0, y0, tx, ty, strip; };      // x0: this WAVE's segment; y0 + 2 wave_s: this wave's first row (>= H: no task); tx: the tile's first segment
    auto decode = [&](int tile) {
        // (unsigned divisions: a signed one by a loop-invariant divisor keeps the divisor's magnitude and sign beside its reciprocal in scalar registers)
        const int cc = (int)((unsigned)tile / (unsigned)tiles_per_cob), rem = tile - cc * tiles_per_cob;
        const bool strip = rem >= mp.wy_full;
        const int tyf = (int)((unsigned)rem / (unsigned)mp.ntx);
        const int l2 = strip ? mp.wy_strip_l2 : 3;
        const int ty = strip ? mp.nty - 1 : tyf;
        const int tx = strip ? (rem - mp.wy_full) << (3 - l2) : rem - tyf * mp.ntx;
        const int wx = tx + (wave_s >> l2);
        TileInfo ti;
        ti.conv = (int)((unsigned)cc / (unsigned)mp.ncob);
        ti.cob = cc - ti.conv * mp.ncob;
        ti.x0 = (wx < mp.ntx ? wx : tx) * TW;
        // (the wave's first row as y0 + 2 wave_s, y0 = the tile's row less the row pairs of the segments to the wave's left: hipcc folds the 2 wave_s
        // term into per-lane constants as it did with full tiles only; a masked wave index in its place cost 11 spilled vector registers)
        ti.y0 = wx < mp.ntx ? ty * TH - 2 * ((wave_s >> l2) << l2) : mp.nty * TH;
        ti.tx = tx;
        ti.ty = ty;
        ti.strip = strip ? 1 : 0;
        return ti;
    };

    // ---- staging cursor: walks the phases (tile, source, 16-channel chunk) of this workgroup's tiles ---------------------------
    // Per lane and tile: the pixel index of (input row d0, its column) for the two full rounds (column = (px c, channel quad q4):
    // c = lane / 4 + 16 r, q4 = lane % 4) and for the tail (the 8 columns of c = 32, 33: lanes < 32 hold ONE (row, column) each).
    f32x4 d[2][4], dx;
#if !WY_BUF
    int st_pxb[2], st_pxt;                                    // pixel index of (row d0 [+ own row for the tail], x), or INVALID
    constexpr int INVALID = -(1 << 30);
#endif
    const float* st_base = nullptr;
    const bf16x8* st_w = nullptr;
    int st_pix = 0, st_cb = 0, st_src = 0, st_conv = 0, st_row0 = 0;
    const int q4 = lane & 3;
#if WY_BUF
    // Buffer-load form: the ROW and the channel chunk go into the resource's 64-bit base (scalar arithmetic), the lane keeps ONE byte offset per
    // round -- its column inside a row, (x * pix + 4 q4) * 4, constant over the tile's phases of a source -- and everything that must read as zero
    // is out of range: columns outside the image carry the offset OOB (>= num_records), rows outside the image get num_records = 0.
    constexpr unsigned OOB = 0x80000000u;
    unsigned st_vcol[2], st_vtail;
    int st_x0 = 0;
    auto stage_cols = [&]() {                                  // per tile and per source (the pixel pitch may differ between the sources of a conv)
#pragma unroll
        // (branch-free on purpose: a per-lane BRANCH here joins in the block where the cursor's scalar fields merge, and hipcc's uniformity
        // analysis then takes those for divergent too -- single unsigned compares and selects instead of short-circuit conditions)
        for (int r = 0; r < 2; ++r) {
            const int x = st_x0 - 1 + (lane >> 2) + 16 * r;
            const unsigned v = (unsigned)(x * st_pix + 4 * q4) * 4u;
            st_vcol[r] = (unsigned)x < (unsigned)W ? v : OOB;
        }
        const int x = st_x0 - 1 + 32 + ((lane & 7) >> 2), rr = (lane >> 3) & 3, row = st_row0 + rr;
        const unsigned vt = (unsigned)((rr * W + x) * st_pix + 4 * q4) * 4u;
        const bool okt = (lane < 32) & ((unsigned)x < (unsigned)W) & ((unsigned)row < (unsigned)H);
        st_vtail = okt ? vt : OOB;
    };
#endif
#if WY_BUF
    // (the cursor's fields are wave-uniform by construction; saying so keeps them -- and the resources built from them -- in scalar registers:
    // without it hipcc carries them in vector registers and wraps every buffer load in a readfirstlane "waterfall" loop)
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto uni_ptr = [](const float* q) {
        const unsigned long long a = (unsigned long long)(uintptr_t)q;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
        return (const float*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    };
#else
    auto uni = [](int v) { return v; };
    auto uni_ptr = [](const float* q) { return q; };
#endif
    auto stage_begin_tile = [&](const TileInfo& ti) {
        st_conv = ti.conv;
        st_src = 0;
        st_cb = 0;
        st_base = uni_ptr(mp.c[ti.conv].src[0]);
        st_pix = uni(mp.c[ti.conv].src_pix[0]);
        st_w = reinterpret_cast<const bf16x8*>(mp.c[ti.conv].wimg) + (long long)ti.cob * mp.nchunk * W_PHASE;
        st_row0 = uni(ti.y0 + 2 * wave_s - 1);                // image row of d0 (scalar)
#if WY_BUF
        st_x0 = ti.x0;
        stage_cols();
#else
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int x = ti.x0 - 1 + (lane >> 2) + 16 * r;
            st_pxb[r] = (x >= 0 && x < W) ? st_row0 * W + x : INVALID;
        }
        {
            const int x = ti.x0 - 1 + 32 + ((lane & 7) >> 2), row = st_row0 + ((lane >> 3) & 3);
            st_pxt = (lane < 32 && x >= 0 && x < W && row >= 0 && row < H) ? row * W + x : INVALID;
        }
#endif
    };
    auto stage_advance = [&]() {
        st_cb += 16;
        st_w += W_PHASE;
        if (st_cb >= mp.src_ch) {
            st_cb = 0;
            ++st_src;
            st_base = uni_ptr(mp.c[st_conv].src[st_src]);
#if WY_BUF
            const int pix_new = uni(mp.c[st_conv].src_pix[st_src]);
            if (pix_new != st_pix) { st_pix = pix_new; stage_cols(); }
#else
            st_pix = mp.c[st_conv].src_pix[st_src];
#endif
        }
    };
    int st_tile = blockIdx.x, st_chunk = 0;
    auto stage_next = [&]() -> bool {                         // cursor -> following phase; false (cursor unchanged) when the block has no more
        if (st_chunk + 1 < mp.nchunk) { ++st_chunk; stage_advance(); return true; }
        if (st_tile + (int)gridDim.x >= total) return false;
        st_tile += gridDim.x;
        st_chunk = 0;
        stage_begin_tile(decode(st_tile));
        return true;
    };
    // one load instruction per call and lane, always (padding reads 16 B of zeros): hipcc's wait insertion then counts exactly.  (Asm loads with
    // hand-counted waits -- hipcc does not see the LDS-DMAs queued behind the rows, so its counts are short of them -- were tried: 304 k -> 322 k
    // cycles per 6 x 128->64 launch, the 64-bit address pairs cost registers the loop does not have.)
#if WY_BUF
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    auto row_rsrc = [&](int i, bool whole) {                  // resource of image row st_row0 + i, channels from st_cb on (all scalar); whole: no row check (the tail's lanes carry their own)
        const int off = ((st_row0 + i) * W * st_pix + st_cb) * 4;      // (32-bit: every tensor of a launch spans < 2 GiB; row -1 gives a negative offset that is never dereferenced)
        const bool row_ok = whole || (st_row0 + i >= 0 && st_row0 + i < H);
        return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)st_base + off), (short)0, row_ok ? 0x7fffffff : 0, 0x00020000);
    };
    auto issue_row = [&](int r, int i) {                      // round r of the cursor's phase, row d_i of this lane's column
        d[r][i] = __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(row_rsrc(i, false), (int)st_vcol[r], 0, 0));
    };
    auto issue_d = [&](int r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_row(r, i);
    };
    auto issue_dx = [&]() { dx = __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(row_rsrc(0, true), (int)st_vtail, 0, 0)); };
#else
    auto load_at = [&](int pixel, bool ok) -> f32x4 {
        const SAVSR_GLOBAL float* src = ok ? (const SAVSR_GLOBAL float*)st_base + (pixel * st_pix + st_cb + 4 * q4)
                                           : (const SAVSR_GLOBAL float*)g_wy_zero16;
        return *(const SAVSR_GLOBAL f32x4*)src;      // (not nt: two waves load each input row and the second one's loads are L1 hits -- nt cost +17 %)
    };
    auto issue_row = [&](int r, int i) {                      // round r of the cursor's phase, row d_i of this lane's column
        const bool row_ok = st_row0 + i >= 0 && st_row0 + i < H;
        d[r][i] = load_at(st_pxb[r] + i * W, row_ok && st_pxb[r] != INVALID);
    };
    auto issue_d = [&](int r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_row(r, i);
    };
    auto issue_dx = [&]() { dx = load_at(st_pxt, st_pxt != INVALID); };
#endif
    // LDS byte offsets of this lane's staging stores inside a (hf, vr, part) plane of its wave's V region
    bf16x8* vwave = smem + wave * V_WAVE;
    constexpr int PLANE_B = V_PLANE * 16;                     // bytes per plane; plane index = (hf * 2 + vr) * 2 + part
    // (byte offsets into the dynamic LDS block, not pointers: every access below is formed as smem_raw + offset, so hipcc keeps them LDS accesses)
    const unsigned vst0 = (unsigned)(wave * V_WAVE + (q4 >> 1) * ICP + (lane >> 2)) * 16u + (unsigned)(q4 & 1) * 8u;      // round r: + r * 16 px * 16 B
    const unsigned vst_t = (unsigned)(wave * V_WAVE + (q4 >> 1) * ICP + 32 + ((lane & 7) >> 2)) * 16u + (unsigned)(q4 & 1) * 8u;
    auto vptr = [&](unsigned off) { return reinterpret_cast<bf16x4*>(smem_raw + off); };
    // Transform + split + store.  Half B of a phase builds positions {0, 1} of the NEXT phase from the rows in d / dx (V0 = d0 - d2, V1 = d1 + d2) and, in
    // place, the fp32 values of positions {2, 3} (V2 = d2 - d1 -> d[r][0], V3 = d1 - d3 -> d[r][1]: the other two row registers are dead from
    // there on -- 18 instead of 36 staging registers live across the tile epilogue); half A of the next phase only splits and stores those.
    auto store_pair = [&](int r, int hf, const f32x4& va, const f32x4& vb) {
        const unsigned p0 = vst0 + (unsigned)(r * 256 + (hf * 2 + 0) * 2 * PLANE_B);
        wy_split_store(va, vptr(p0), vptr(p0 + PLANE_B));
        const unsigned p1 = vst0 + (unsigned)(r * 256 + (hf * 2 + 1) * 2 * PLANE_B);
        wy_split_store(vb, vptr(p1), vptr(p1 + PLANE_B));
    };
    auto store_one = [&](int r, int hf, int vr, const f32x4& v) {
        const unsigned p0 = vst0 + (unsigned)(r * 256 + (hf * 2 + vr) * 2 * PLANE_B);
        wy_split_store(v, vptr(p0), vptr(p0 + PLANE_B));
    };
    auto store_v0 = [&](int r) { store_one(r, 0, 0, d[r][0] - d[r][2]); };          // (half B, piece 1) position 0 out; d0 is dead from here
    auto store_v1 = [&](int r) {                                                     // (half B, piece 2) position 1 out, positions 2, 3 kept in place
        store_one(r, 0, 1, d[r][1] + d[r][2]);
        const f32x4 v2 = d[r][2] - d[r][1], v3 = d[r][1] - d[r][3];
        d[r][0] = v2;
        d[r][1] = v3;
    };
    auto store_v01 = [&](int r) { store_v0(r); store_v1(r); };
    auto store_v23 = [&](int r) { store_pair(r, 1, d[r][0], d[r][1]); };      // (half A)
    // the tail columns: lane l < 32 holds row l / 8 of column 128 + l % 8; the partner row comes by a cross-lane move.
    //   positions 0, 1: lanes 0-7 (d0) take d2 from lane ^ 16 -> V0 = own - other; lanes 8-15 (d1) take d2 from lane ^ 24 -> V1 = own + other
    //   positions 2, 3: lanes 16-23 (d2) take d1 from lane ^ 24 -> V2 = own - other; lanes 8-15 (d1) take d3 from lane ^ 16 -> V3 = own - other
    const int grp = (lane >> 3) & 3;
    auto tail_value = [&](int hf) -> f32x4 {
        const int partner = hf == 0 ? (grp == 0 ? lane ^ 16 : lane ^ 24) : (grp == 2 ? lane ^ 24 : lane ^ 16);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __shfl(dx[j], partner, 64);
        return (hf == 0 && grp == 1) ? dx + o : dx - o;
    };
    auto tail_store = [&](int hf, const f32x4& v) {
        const bool mine = lane < 32 && (hf == 0 ? grp < 2 : (grp == 2 || grp == 1));
        const int vr = hf == 0 ? grp : (grp == 2 ? 0 : 1);
        if (mine) {
            const unsigned pp = vst_t + (unsigned)((hf * 2 + vr) * 2 * PLANE_B);
            wy_split_store(v, vptr(pp), vptr(pp + PLANE_B));
        }
    };
    auto store_vt01 = [&]() {                                 // (half B) positions 0, 1 out; dx <- this lane's value of positions 2, 3
        const f32x4 v01 = tail_value(0), v23 = tail_value(1);
        tail_store(0, v01);
        dx = v23;
    };
    auto store_vt23 = [&]() { tail_store(1, dx); };           // (half A)
    // weight slab half hf of the cursor's phase: 24 pieces of 1 KiB, 3 per wave, straight into LDS (the packed image IS the LDS image)
    bf16x8* wlds = smem + 8 * V_WAVE;
    [[maybe_unused]] const unsigned lane16 = (unsigned)lane * 16u;
    auto issue_w = [&](int j, int hf) {
        const int piece = wave_s * 3 + j;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(wlds + hf * W_HALF + piece * 64));
        unsigned keep;
#if WY_BUF
        // scalar 64-bit piece address + one constant per-lane byte offset (the cursor is scalar: no 64-bit vector address per piece)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane16), "s"(st_w + hf * W_HALF + piece * 64), "s"(dst) : "memory");
#else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(st_w + hf * W_HALF + piece * 64 + lane), "s"(dst) : "memory");
#endif
    };

    // Fragment registers: per step s = (position-in-half, kx) one activation pair (hi, lo) and per sub-step (s, t) one weight pair; both
    // double-buffered one SUB-step (3 MFMAs) ahead -- a whole step's six fragments twice over (48 registers) do not fit beside the 128
    // accumulator registers and the 36 staging registers.
    struct FragA { bf16x8 ah, al; };
    struct FragB { bf16x8 bh, bl; };
    const bf16x8* vrd = vwave + half * ICP + px;               // + plane * V_PLANE + kx
    const bf16x8* wrd = wlds + lane;
    auto load_b = [&](int hf, int s, FragB& fr) {             // s = vr * 3 + kx
        const int vr = s / 3, kx = s - 3 * vr;
        fr.bh = vrd[((hf * 2 + vr) * 2 + 0) * V_PLANE + kx];
        fr.bl = vrd[((hf * 2 + vr) * 2 + 1) * V_PLANE + kx];
    };
    auto load_a = [&](int hf, int s, int t, FragA& fr) {
        fr.ah = wrd[hf * W_HALF + ((s * 2 + t) * 2 + 0) * 64];
        fr.al = wrd[hf * W_HALF + ((s * 2 + t) * 2 + 1) * 64];
    };

    int tile = blockIdx.x;
#if WY_STAMPS
    long long wy_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long wy_prev = (long long)__builtin_amdgcn_s_memtime();
    const long long wy_t0 = wy_prev;
#endif
    FragA fa[2];
    FragB fb[2];
    if (tile < total) {
        // prologue: WA(0) by DMA, rows of phase 0 -> VA(0) (positions 0, 1); VB(0) and WB(0) follow in half A of phase 0 like in every phase
        stage_begin_tile(decode(tile));
#pragma unroll
        for (int j = 0; j < 3; ++j) issue_w(j, 0);
        issue_d(0);
        issue_d(1);
        issue_dx();
        store_v01(0);
        store_v01(1);
        store_vt01();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tile < total) { load_b(0, 0, fb[0]); load_a(0, 0, 0, fa[0]); }

    for (; tile < total; tile += gridDim.x) {
        const TileInfo cur = decode(tile);
        const ConvParams& p = mp.c[cur.conv];
        const int cob = cur.cob, x0 = cur.x0, y0 = cur.y0;
        f32x16 acc[4][2];                                     // [position][output-channel block]
#pragma unroll
        for (int pz = 0; pz < 4; ++pz)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[pz][t][i] = 0.f;
        // the pair's first row inside the image.  From wave_s (scalar), not through a readfirstlane of the comparison: hipcc folds that into
        // readfirstlane.i1, which its uniformity analysis does not know to be uniform -- the branch below then counts as divergent and every
        // field of the staging cursor that the two phase bodies advance is carried in VECTOR registers from there on
        const bool rows_in = y0 + 2 * wave_s < H;
        for (int chunk = 0; chunk < mp.nchunk; ++chunk) {
            auto phase = [&](auto mm) {
                constexpr bool MM = decltype(mm)::value;      // false: both rows below the image -> no matrix work (everything else as usual)
                // Staging work of (half, step).  Two rules, both from the section stamps (WY_STAMPS):
                //  * at most TWO vector-memory instructions per wave and step: with the three DMAs and the nine row loads of a phase in steps 0-2 the
                //    eight waves queued 96 KB on the CU's address path at once and every wave stalled at issue in front of its MFMAs (half A 5.1 k
                //    cycles per phase against 3.2 k for half B, which carries more arithmetic);
                //  * the transform / split arithmetic of a step is spread over all SIX of its MFMAs (one scheduling region per step, sched_group_barrier
                //    pipeline below): placed in one 3-MFMA sub-step it ran 28 vector instructions beside 3 MFMAs in BOTH waves of a SIMD at once.
                auto pieces = [&](int hf, int s) {
                    if (hf == 0) {
                        // half A: WB of THIS phase by DMA first (the three pieces must be older than the row loads of the next phase, so that the barrier's
                        // counted wait -- vmcnt(8) -- covers the DMAs and nothing else); positions {2, 3} of this phase out of d (split + store only);
                        // then, into the registers just released, the rows of the NEXT phase
                        if (s == 0) { issue_w(0, 1); issue_w(1, 1); store_v23(0); }
                        if (s == 1) { issue_w(2, 1); stage_next(); issue_row(0, 0); }
                        if (s == 2) { store_v23(1); issue_row(0, 1); issue_row(0, 2); }
                        if (s == 3) { store_vt23(); issue_row(0, 3); issue_row(1, 0); }
                        if (s == 4) { issue_row(1, 1); issue_row(1, 2); }
                        if (s == 5) issue_row(1, 3);              // (sub-step (5, 0); the tail load follows the barrier)
                    } else {
                        // half B (cursor already advanced): WA of the next phase by DMA, one piece per step; positions {0, 1} of the next phase
                        if (s < 3) issue_w(s, 0);
                        if (s == 1) store_v0(0);
                        if (s == 2) store_v1(0);
                        if (s == 3) store_v0(1);
                        if (s == 4) store_v1(1);
                        if (s == 5) store_vt01();                 // (sub-step (5, 0))
                    }
                };
                auto mma3 = [&](int pz, int t, int sb) {
                    const FragA& a = fa[t];
                    const FragB& b = fb[sb];
                    if (MM) {
                        acc[pz][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.al, b.bh, acc[pz][t], 0, 0, 0);
                        acc[pz][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.ah, b.bl, acc[pz][t], 0, 0, 0);
                        acc[pz][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.ah, b.bh, acc[pz][t], 0, 0, 0);
                    }
                };
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                    for (int s = 0; s < 5; ++s) {
                        // step s < 5, one scheduling region: fragments of sub-step (s, 1) and the next step's activation pair first, 3 MFMAs, the
                        // next step's first weight pair (into the registers the first three MFMAs have read), 3 MFMAs; the staging work between them
                        const int pz = hf * 2 + s / 3;
                        load_a(hf, s, 1, fa[1]);
                        load_b(hf, s + 1, fb[(s + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        mma3(pz, 0, s & 1);
                        pieces(hf, s);
                        load_a(hf, s + 1, 0, fa[0]);
                        mma3(pz, 1, s & 1);
                        if (MM) {
#pragma unroll
                            for (int i = 0; i < 6; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                                if (i == 3) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // the two weight-fragment reads for the next step, behind the third MFMA
                                __builtin_amdgcn_sched_group_barrier(0x002, WY_VALU, 0);      // up to WY_VALU vector instructions
                                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // up to one DS write
                                if (i == 1 || i == 4) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);      // one vector-memory instruction
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    {   // step 5: sub-step (5, 0), then the barrier that publishes the other weight half, then (5, 1) on the next half's first fragments
                        const int pz = hf * 2 + 1;
                        load_a(hf, 5, 1, fa[1]);
                        __builtin_amdgcn_sched_barrier(0);
                        mma3(pz, 0, 1);
                        pieces(hf, 5);
                        if (MM) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        WY_MARK(WY_KSEC(hf * 3 + 0));
                        // the weight half DMA'd during this half (for the NEXT half to run) has landed; publish it.  Half A: the 8 row loads of the
                        // next phase (steps 1-5) are younger and may fly on; half B: the DMAs are the youngest operations.
                        if (hf == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __syncthreads();
                        WY_MARK(WY_KSEC(hf * 3 + 1));
                        load_b(hf ^ 1, 0, fb[0]);
                        load_a(hf ^ 1, 0, 0, fa[0]);
                        __builtin_amdgcn_sched_barrier(0);
                        mma3(pz, 1, 1);
                        if (hf == 0) issue_dx();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    WY_MARK(WY_KSEC(hf * 3 + 2));
                }
            };
            if (rows_in) phase(std::true_type{}); else phase(std::false_type{});
        }

        // ---- epilogue: as the direct kernel's (conv_mfma.hip): transpose through the wave's LDS slice 32 channels at a time, whole
        // pixel records out, fused bias / activation / per-pixel mask / two residuals / global-average-pool partials.  Rows of wave w:
        // y0 + 2 w + r.  Every load below is unconditional and used on every path (absent operands read 16 B of zeros).
        const float* e_bias = p.bias;
        const float* e_mul = p.mul_px;
        const float* e_r1 = p.res1;
        const float* e_r2 = p.res2;
        float* e_out = p.out;
        float* e_pool = p.pool;
        int e_act = p.act, e_opix = p.out_pix, e_r1pix = p.res1_pix, e_r2pix = p.res2_pix;
        float e_slope = p.slope, e_r2s = p.res2_scale;
        asm volatile("" : "+s"(e_bias), "+s"(e_mul), "+s"(e_r1), "+s"(e_r2), "+s"(e_out), "+s"(e_pool));
        asm volatile("" : "+s"(e_act), "+s"(e_opix), "+s"(e_r1pix), "+s"(e_r2pix), "+s"(e_slope), "+s"(e_r2s));
        const bool act_as_max = e_act == SAVSR_ACT_NONE || e_act == SAVSR_ACT_RELU || (e_act == SAVSR_ACT_LRELU && e_slope >= 0.f && e_slope <= 1.f);
        const float slope_eff = e_act == SAVSR_ACT_NONE ? 1.f : (e_act == SAVSR_ACT_RELU ? 0.f : e_slope);
        float* ep_base = reinterpret_cast<float*>(smem + LDS_UNITS);
        float* ep = ep_base + wave * (32 * EPS);
        const int c4 = lane & 7;
        const bool x_inside = x0 + TW <= W;
        const float* zero16 = (const float*)g_wy_zero16;
        const float* b_base = e_bias ? e_bias : zero16;
        const unsigned b_off = e_bias ? 4u * (unsigned)(cob * COT + 4 * c4) : 0u, b_step = e_bias ? 128u : 0u;
        f32x4 bias4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bias4[t] = ldg4(b_base, b_off + (unsigned)t * b_step);
        const float* r1_base = e_r1 ? e_r1 : zero16;
        constexpr int RR = 2;                      // residual quads of two (row, channel-group) steps in flight (a ring): the first two go out together.
        f32x4 rr[RR][4];                           // (all four at once, into the registers of the accumulator sets the transform has retired: 22 spilled registers)
        auto load_r1 = [&](int r, int t, f32x4 (&dst)[4]) {
            const int y = y0 + 2 * wave_s + r;
            const int co = cob * COT + 32 * t + 4 * c4;
            const unsigned off0 = 4u * (unsigned)((y * W + x0 + (lane >> 3)) * e_r1pix + co), ustride = 32u * (unsigned)e_r1pix;
            const bool row_ok = e_r1 && y < H;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = row_ok && (x_inside || x0 + (lane >> 3) + 8 * i < W);
                dst[i] = ldg4(r1_base, ok ? off0 + (unsigned)i * ustride : 0u);
            }
        };
#pragma unroll
        for (int gi = 0; gi < RR; ++gi) load_r1(gi / 2, gi % 2, rr[gi]);
        // ---- output transform (in registers), under the round trip of the loads above: row 0 = M0 + M1 + M2 -> acc[0], row 1 = M1 - M2 - M3 -> acc[3]
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float m1 = acc[1][t][i], m2 = acc[2][t][i];
                acc[0][t][i] = (acc[0][t][i] + m1) + m2;
                acc[3][t][i] = (m1 - m2) - acc[3][t][i];
            }
        f32x4 psum[2][2];
#ifdef WY_EMARK
        asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[3][1][15]));
        WY_EMARK(1);
#endif
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = y0 + 2 * wave_s + r;
            const int pbase = y * W + x0 + (lane >> 3);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                psum[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int gi = r * 2 + t;
                if (y >= H) {
                    asm volatile("" :: "v"(bias4[t]), "v"(rr[gi % RR][0]), "v"(rr[gi % RR][1]), "v"(rr[gi % RR][2]), "v"(rr[gi % RR][3]));
                    if (gi + RR < 4) load_r1((gi + RR) / 2, (gi + RR) % 2, rr[gi % RR]);
                    continue;
                }
                const f32x16& a = r == 0 ? acc[0][t] : acc[3][t];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                    *reinterpret_cast<f32x4*>(ep + px * EPS + 8 * g + 4 * half) = v;
                }
                const int co = cob * COT + 32 * t + 4 * c4;
                const f32x4 b4 = bias4[t];
                const unsigned ooff0 = 4u * (unsigned)(pbase * e_opix + co), ostride = 32u * (unsigned)e_opix;
                f32x4 ps = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ih = 0; ih < 2; ++ih) {                  // two units (pixels pbase + 16 ih, + 8) at a time: register budget
                    const int p0 = pbase + 16 * ih;
                    const bool ok0 = x_inside || x0 + (lane >> 3) + 16 * ih < W, ok1 = x_inside || x0 + (lane >> 3) + 16 * ih + 8 < W;
                    f32x4 v[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ep + ((lane >> 3) + 16 * ih + 8 * i) * EPS + 4 * c4);
                        v[i] = a4 + b4;
                    }
                    if constexpr (FAST) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const f32x4 sv = v[i] * slope_eff;
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[i][q] = vmax_raw(v[i][q], sv[q]);
                        }
                    } else if (e_act == SAVSR_ACT_NONE) {
                    } else if (act_as_max) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const f32x4 sv = v[i] * slope_eff;
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[i][q] = vmax_raw(v[i][q], sv[q]);
                        }
                    } else if (e_act == SAVSR_ACT_LRELU) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[i][q] = v[i][q] > 0.f ? v[i][q] : v[i][q] * e_slope;
                    } else if (e_act == SAVSR_ACT_SIGMOID) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[i][q] = sigmoidf_(v[i][q]);
                    }
                    if (!FAST && e_mul) {
                        const float m0 = ok0 ? ldg1(e_mul, (unsigned)p0) : 0.f, m1 = ok1 ? ldg1(e_mul, (unsigned)(p0 + 8)) : 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { v[0][q] *= m0; v[1][q] *= m1; }
                    }
                    v[0] += rr[gi % RR][2 * ih];
                    v[1] += rr[gi % RR][2 * ih + 1];
                    if (!FAST && e_r2) {
                        f32x4 ra = {0.f, 0.f, 0.f, 0.f}, rb = {0.f, 0.f, 0.f, 0.f};
                        if (ok0) ra = ldg4(e_r2, 4u * (unsigned)(p0 * e_r2pix + co));
                        if (ok1) rb = ldg4(e_r2, 4u * (unsigned)((p0 + 8) * e_r2pix + co));
#pragma unroll
                        for (int q = 0; q < 4; ++q) { v[0][q] += e_r2s * ra[q]; v[1][q] += e_r2s * rb[q]; }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (i == 0 ? ok0 : ok1) {
                            stg4(e_out, ooff0 + (unsigned)(2 * ih + i) * ostride, v[i]);
                            if (e_pool) ps += v[i];
                        }
                    }
                }
                psum[r][t] = ps;
#ifdef WY_EMARK
                WY_EMARK(2 + gi);
#endif
                if (gi + RR < 4) load_r1((gi + RR) / 2, (gi + RR) % 2, rr[gi % RR]);       // the ring slot is free: next residual group out
            }
        }
        if (e_pool) {
            // AdaptiveAvgPool2d(1) partials (savsr_arch.py:146,515): one row per 8-row band and 32-pixel column of the image, as the direct
            // kernel writes them (same row numbering: savsr_conv_pool_blocks); band of wave w's rows = w / 4; waves summed in wave order
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int o = 8; o < 64; o <<= 1) {
                        psum[r][t][0] += __shfl_xor(psum[r][t][0], o, 64); psum[r][t][1] += __shfl_xor(psum[r][t][1], o, 64);
                        psum[r][t][2] += __shfl_xor(psum[r][t][2], o, 64); psum[r][t][3] += __shfl_xor(psum[r][t][3], o, 64);
                    }
            __syncthreads();                         // every wave is done with its transpose slice
            float* pl_ = ep_base;
            if (lane < 8)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4*>(pl_ + (wave * 2 + r) * COT + 32 * t + 4 * lane) = psum[r][t];
            __syncthreads();
            // full tile: two bands (waves 4 bl .. 4 bl + 3) of one segment; strip tile: one band, 8 >> l segments of 2^l waves each.  The rows of a
            // (band, segment) are summed in the same order either way (rows below the image contribute +0 in a full tile, nothing in a strip)
            const TileInfo pt = decode(tile);           // (tx / ty / strip again, from the tile index: not kept in scalar registers across the tile)
            const int l2p = pt.strip ? mp.wy_strip_l2 : 2;
            if (tid < (8 >> l2p) * COT) {
                const int sg = tid / COT, ch = tid - sg * COT;          // (wave-uniform: COT = 64)
                float sacc = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < (2 << l2p)) sacc += pl_[(((sg << l2p) * 2) + k) * COT + ch];       // waves sg 2^l .. (sg + 1) 2^l - 1, rows r = 0, 1 each
                const int band = pt.ty * 2 + (pt.strip ? 0 : sg), ptx = pt.tx + (pt.strip ? sg : 0);
                if (band * 8 < H && ptx < mp.ntx) stg1(e_pool, (unsigned)((band * mp.ntx + ptx) * p.pool_stride + cob * COT + ch), sacc);
            }
            __syncthreads();                         // the slices are reused by the next tile's epilogue
        }
        WY_MARK(6);
    }
#if WY_STAMPS
    wy_sec[7] = (long long)__builtin_amdgcn_s_memtime() - wy_t0;
    if (lane == 0 && blockIdx.x < 256)
        for (int i = 0; i < 8; ++i) g_wy_stamps[(blockIdx.x * 8 + wave) * 8 + i] = wy_sec[i];
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the last phases re-stage unconditionally: no LDS-DMA may be in flight when the LDS is released
}

int launch_conv_wy(const MultiConvParams& mp, hipStream_t st) {
    if (int rc = conv_wy_prepare_device()) return rc;
    const int total = mp.nconv * mp.ncob * mp.wy_tiles;
    const int grid = total < CONV_PERSISTENT_BLOCKS ? total : CONV_PERSISTENT_BLOCKS;
    bool fast = WY_FAST_EPILOGUE != 0;
    for (int i = 0; i < mp.nconv && fast; ++i) {
        const ConvParams& c = mp.c[i];
        const bool as_max = c.act == SAVSR_ACT_NONE || c.act == SAVSR_ACT_RELU || (c.act == SAVSR_ACT_LRELU && c.slope >= 0.f && c.slope <= 1.f);
        fast = as_max && !c.mul_px && !c.res2;
    }
    if (fast) hipLaunchKernelGGL(conv_wy_kernel<true>, dim3(grid), dim3(wy::NTHR), wy::LDS_BYTES, st, mp);
    else hipLaunchKernelGGL(conv_wy_kernel<false>, dim3(grid), dim3(wy::NTHR), wy::LDS_BYTES, st, mp);
    return check_launch("conv_wy_kernel");
}
int conv_wy_prepare_device() {
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_wy_kernel<true>), (int)wy::LDS_BYTES, "conv_wy")) return rc;
    return ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_wy_kernel<false>), (int)wy::LDS_BYTES, "conv_wy");
}

}  // namespace savsr

using namespace savsr;

#if WY_STAMPS
extern "C" int savsr_debug_read_wy_stamps(long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wy_stamps), sizeof(long long) * 256 * 8 * 8);
}
#endif

// Elements PER PART of the Winograd-y weight image: [cob][chunk][hf 2][s = vr * 3 + kx 6][t 2][part][512]: 12 taps instead of 9.
extern "C" int64_t savsr_conv_wy_packed_elems(int cout, int cin) {
    if (cout <= 0 || cout % 64 || cin <= 0 || cin % 16) return -1;
    return (int64_t)(cout / 64) * (cin / 16) * 12 * 16 * 64;
}
// Position of U[pos][co][ci][kx] (pos = Winograd position 0..3) inside ONE part; the hi part of a (.., t) group of 512 elements is
// followed by its lo part, as in savsr_conv_pack_index.
extern "C" int64_t savsr_conv_wy_pack_index(int cout, int cin, int co, int ci, int pos, int kx) {
    const int64_t nchunk = cin / 16;
    const int cob = co / 64, col = co % 64, t = col / 32, row = col % 32;
    const int chunk = ci / 16, cl = ci % 16, kh = cl / 8, j = cl % 8;
    const int hf = pos / 2, vr = pos % 2, s = vr * 3 + kx;
    const int64_t group = ((((int64_t)(cob * nchunk + chunk) * 2 + hf) * 6 + s) * 2 + t);
    return group * 512 + (kh * 32 + row) * 8 + j;
}
