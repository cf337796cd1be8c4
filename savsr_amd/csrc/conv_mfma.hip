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
// are 8 consecutive channels of one pixel, and the 4 consecutive output rows a lane holds per
// accumulator quad are 4 consecutive channels = one 16-B store.
// Weights: pre-split, pre-packed bf16 image in exactly the LDS order the lanes read it:
//   [cob][chunk][tap][kstep][t][part(hi,lo)][lane 64][8]  (1-KiB pieces).
//
// Workgroup = 8 waves; tile = 8 rows x 32 cols of pixels x (32 NT) output channels, wave w owns row w (PXT = 1), or
// 16 rows with wave w owning rows w and w + 8 (PXT = 2: one weight-fragment read feeds two pixel rows; used when a
// launch has >= 200 such tiles).  The kernel is PERSISTENT: <= 256 workgroups walk tiles t = blockIdx.x, + gridDim.x,
// ... of up to 6 independent convs of identical geometry (savsr_conv2d_batch).
//
// Pipeline (DESIGN.md section 4a has the measurements behind each choice).  The phases (KC input channels x all taps)
// of a workgroup's tiles form one linear sequence walked by a staging cursor.  While phase c computes from LDS buffer
// c % 2: the weight slab of phase c+1 arrives in the other buffer by LDS-DMA (the packed image IS the LDS image); the
// activations of phase c+1, loaded into registers during phase c-1, are split to (hi, lo) bf16 and stored to LDS; the
// activations of phase c+2 are loaded into the freed registers.  All of that is issued in small pieces BETWEEN the
// three MFMA groups of every (tap, kstep) step, because the two waves of a SIMD run in lockstep and share the issue
// port.  ONE barrier per phase, placed where every fragment read of the phase has been issued, so that the last steps
// already read the next phase's -- or the next TILE's -- first fragments: the MFMA stream does not stop at phase or
// tile boundaries, and a tile's epilogue stores drain under its successor's MFMAs.  Fragment reads run one or two
// steps ahead of their MFMAs with the issue order pinned by sched_barriers (hipcc otherwise sinks every ds_read to
// just before its MFMA).  Epilogue: each wave transposes its tile through an LDS slice, 32 channels at a time, and
// stores whole pixel records, with fused bias / activation / per-pixel mask / two residuals / global-average-pool
// partial sums; descriptor fields are pinned in scalars once per tile and every access names the global address space.
//
// Replaces: every nn.Conv2d / F.conv2d of savsr_arch.py (see include/savsr_hip.h).
#include "conv_common.hpp"

#include <type_traits>

// Settled build-time choices (their A/B numbers are in DESIGN.md sections 4 / 10; the switches themselves -- CONV_INTERLEAVE, CONV_IL_VALU,
// EPI_PREFETCH, CONV_PRIO, ROWS_SKIP and the CONV_EXP timing knobs -- are archived as tools/experiments/conv_mfma_switches.patch):
//  * straight-line steps with a sched_group_barrier pipeline: up to 3 vector instructions and one DS write behind every MFMA;
//  * the epilogue's first global loads (bias quads, first residual group) are issued in front of the tile's LAST phase (3x3 kernels);
//  * static issue priority: waves 4-7 at priority 1 for the whole kernel (rounds 1-3 alternated per step: a scalar branch around s_setprio in every
//    step, which cut each step into its own basic block; A/B in one process, round 4: 6 x 128->64 150.6 -> 147.6 us, 6 x 64->64 88.6 -> 87.8);
//  * a wave with NO row inside the image runs a phase body without MFMAs (a third, one-row body spilled 69 VGPRs in the 16-row kernel).

#ifndef CONV_BUF
#define CONV_BUF 1                // 1: activation fetch by buffer loads off a scalar staging cursor (round 6); 0: flat global loads with per-load 64-bit vector addresses
#endif

namespace savsr {

// Diagnostics (not used by the product path): per-workgroup s_memtime stamps, enabled by
// savsr_debug_conv_stamps(1) and read back with savsr_debug_read_conv_stamps():
// [blk][6] = entry, after the first staging, after the first K phase, after the first tile's K loop,
// kernel end (stores drained), s_memrealtime at entry.
[[maybe_unused]] constexpr int STAMP_BLOCKS = 1024, STAMP_N = 6;
#if defined(SAVSR_DIAG)
#define CONV_HAS_STAMPS 1
__device__ long long g_conv_stamps[STAMP_BLOCKS * STAMP_N];
__device__ int g_conv_stamps_on = 0;
#else
#define CONV_HAS_STAMPS 0      // the product library: no diagnostic state at all
#endif
__device__ __attribute__((aligned(16))) const float g_conv_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // what padding lanes load

__device__ __forceinline__ void stamp(int on, int slot) {
#if CONV_HAS_STAMPS
    if (on == 1 && threadIdx.x == 0 && blockIdx.x < STAMP_BLOCKS)
        g_conv_stamps[blockIdx.x * STAMP_N + slot] = (slot == 5) ? (long long)__builtin_amdgcn_s_memrealtime() : (long long)__builtin_amdgcn_s_memtime();
#endif
}

// DIAG: the instrumented build (section stamps, timing experiments), launched only while savsr_debug_conv_stamps is on;
// as run-time switches the diagnostics cost scalar registers (and spills) in every step of the product kernel.
// (A straight-line epilogue variant as conv_wy_kernel<true> was built and measured in round 6: 0.4-2.9 % per launch, nothing on the frame or on
// config 5 -- this kernel is 4 % of the GPU time -- and not kept: profiles/r06_ab_conv_direct_fast_epilogue.log.)
template <int KS, int NT, int PXT, bool DIAG>
__global__ __launch_bounds__(512) void conv_bf16x3_kernel(const MultiConvParams mp) {
    constexpr int TH = CONV_TH * PXT, NTHR = 64 * CONV_TH;   // wave w owns tile rows w, w + 8, .. (PXT of them)
    constexpr int TAPS = KS * KS, HALO = KS / 2;
    constexpr int KC = conv_kc(KS), KSTEPS = KC / 16;
    constexpr int IR = TH + 2 * HALO, IC = CONV_TW + 2 * HALO, NPX = IR * IC;
    constexpr int B_PART = KSTEPS * 2 * NPX;                 // 16-B units per part (hi or lo)
    constexpr int B_UNITS = 2 * B_PART;
    constexpr int W_UNITS = TAPS * KSTEPS * NT * 2 * 64;     // 16-B units per phase
    constexpr int PER = KSTEPS * 4;                          // float4 items per pixel and phase
    constexpr int B_ITEMS = NPX * PER;                       // 4 lanes read one pixel's 64 B: 16 lines per wave load
    constexpr int B_IT = (B_ITEMS + NTHR - 1) / NTHR;
    constexpr int W_IT = (W_UNITS + NTHR - 1) / NTHR;
    constexpr int STEPS = TAPS * KSTEPS;
    constexpr int COT = 32 * NT;
    constexpr int EPS = 36;                                  // floats per pixel in the epilogue slice (32 channels + pad)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* smem = reinterpret_cast<bf16x8*>(smem_raw);      // 16-B units: [2][B_UNITS] | [2][W_UNITS] | epilogue slices
    // PXT == 2 has no LDS left for dedicated epilogue slices: they alias the staging buffer the tile's last phase
    // has just released (free until the store of the phase after next; see the barrier after the epilogue)
    constexpr bool EP_ALIAS = PXT > 1;
    static_assert(!EP_ALIAS || B_UNITS * 16 >= CONV_TH * 32 * EPS * 4, "aliased epilogue slices must fit one input buffer");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, px = lane & 31;
    [[maybe_unused]] const int prio_group = __builtin_amdgcn_readfirstlane(wave >> 2);      // wave-uniform by construction: a scalar for the s_setprio branch
    const int tiles_per_cob = mp.ntx * mp.nty;
    const int total = mp.nconv * mp.ncob * tiles_per_cob;
#if CONV_HAS_STAMPS
    const int dbg_all = DIAG ? __builtin_amdgcn_readfirstlane(g_conv_stamps_on) : 0;
#else
    static_assert(!DIAG, "DIAG kernels exist in the instrumented library only");
    const int dbg_all = 0;
#endif
    const int stamps_on = dbg_all & 15;
    const bool dbg_nostage = dbg_all & 16, dbg_nofrag = dbg_all & 32;   // timing experiments only (results are wrong)
    const bool dbg_nost = dbg_all & 64, dbg_nolds = dbg_all & 128;      // epilogue without its global stores / without the LDS transpose

    f32x4 b_reg[B_IT];
    [[maybe_unused]] const unsigned tid16 = (unsigned)tid * 16u;

    struct TileInfo { int conv, cob, x0, y0, tx, ty; };
    auto decode = [&](int tile) {
        const int cc = tile / tiles_per_cob, rem = tile - cc * tiles_per_cob;
        const int ty = rem / mp.ntx, tx = rem - ty * mp.ntx;
        TileInfo ti;
        ti.conv = cc / mp.ncob;
        ti.cob = cc - ti.conv * mp.ncob;
        ti.x0 = tx * CONV_TW;
        ti.y0 = ty * TH;
        ti.tx = tx;
        ti.ty = ty;
        return ti;
    };
    // Staging, global -> registers.  A cursor walks the phases: (tile, source, channel base); per tile it keeps each
    // thread's pixel offsets (they do not change over the K loop), per source the base pointer and pixel pitch, and
    // the weight slab pointer just advances.  Descriptor fields with a run-time conv / source index are therefore
    // read once per tile / source, not per phase (15 dependent s_loads per phase cost ~3 k cycles, stamps mode 3).
    const int H = mp.h, W = mp.w;
    const float* st_base = nullptr;
    const f32x4* st_w = nullptr;
    int st_pix = 0, st_cb = 0, st_src = 0, st_conv = 0;
#if CONV_BUF
    // Buffer-load form of the activation fetch (round 6, as in conv_wy.hip): the source base and the channel chunk go into the resource's 64-bit
    // base (scalar ALU), a thread keeps ONE byte offset per staged item -- its pixel inside the tensor, (pixel * pix + 4 c8) * 4, constant over the
    // K loop of a source -- and halo pixels outside the image carry the offset OOB (>= num_records), which the bounds check answers with zeros.
    // Everything per-lane here is BRANCH-FREE: a per-lane branch joins in the block where the cursor's scalar fields merge, and hipcc's
    // uniformity analysis then takes st_base / st_pix / st_cb / st_tile for divergent and carries them in vector registers.
    constexpr unsigned OOB = 0x80000000u;
    unsigned st_voff[B_IT];
    int st_x0 = 0, st_y0 = 0;
    auto stage_offsets = [&]() {
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int e = tid + i * NTHR;
            const int pl = e / PER, c8 = e - pl * PER;              // pixel of the tile, float4 of the chunk
            const int r = pl / IC, c = pl - r * IC;
            const int gy = st_y0 - HALO + r, gx = st_x0 - HALO + c;
            const bool ok = (e < B_ITEMS) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
            const unsigned v = (unsigned)((gy * W + gx) * st_pix + c8 * 4) * 4u;
            st_voff[i] = ok ? v : OOB;
        }
    };
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto uni_ptr = [](const float* q) {
        const unsigned long long a = (unsigned long long)(uintptr_t)q;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
        return (const float*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    };
#else
    int st_pixoff[B_IT];
    auto uni = [](int v) { return v; };
    auto uni_ptr = [](const float* q) { return q; };
#endif
    auto stage_begin_tile = [&](const TileInfo& ti) {
        st_conv = ti.conv;
        st_src = 0;
        st_cb = 0;
        st_base = uni_ptr(mp.c[ti.conv].src[0]);
        st_pix = uni(mp.c[ti.conv].src_pix[0]);
        st_w = reinterpret_cast<const f32x4*>(mp.c[ti.conv].wimg) + (long long)ti.cob * mp.nchunk * W_UNITS;
#if CONV_BUF
        st_x0 = ti.x0;
        st_y0 = ti.y0;
        stage_offsets();
#else
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int e = tid + i * NTHR;
            int off = -1;
            if (e < B_ITEMS) {
                const int pl = e / PER;                             // pixel of the tile
                const int r = pl / IC, c = pl - r * IC;
                const int gy = ti.y0 - HALO + r, gx = ti.x0 - HALO + c;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) off = gy * W + gx;
            }
            st_pixoff[i] = off;
        }
#endif
    };
    auto stage_advance = [&]() {
        st_cb += KC;
        st_w += W_UNITS;
        if (st_cb >= mp.src_ch) {
            st_cb = 0;
            ++st_src;
            st_base = uni_ptr(mp.c[st_conv].src[st_src]);
#if CONV_BUF
            const int pix_new = uni(mp.c[st_conv].src_pix[st_src]);
            if (pix_new != st_pix) { st_pix = pix_new; stage_offsets(); }
#else
            st_pix = mp.c[st_conv].src_pix[st_src];
#endif
        }
    };
    // Loads of the cursor's phase.  issue_w(g, wbuf): 1 KiB per wave of the weight slab straight into LDS buffer `wbuf`
    // (LDS-DMA: the packed image IS the LDS image, so no registers and no ds_write).  hipcc does not count an asm load:
    // the wait is the explicit counted vmcnt in front of the publishing barrier.  issue_b(i): one activation float4 per
    // lane into registers -- always exactly one load instruction (out-of-image pixels read a 16-B block of zeros), so
    // that count is exact, and nothing touches the value before its store.
    auto issue_w = [&](int g, int wbuf) {
        const int e = tid + g * NTHR;
        if (e < W_UNITS) {                                              // wave-uniform (W_UNITS is a multiple of 64)
            const unsigned dst = __builtin_amdgcn_readfirstlane(
                (unsigned)(uintptr_t)(smem + 2 * B_UNITS + wbuf * W_UNITS + g * NTHR + wave * 64));
            unsigned keep;
#if CONV_BUF
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(tid16), "s"(st_w + g * NTHR), "s"(dst) : "memory");
#else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(st_w + e), "s"(dst) : "memory");
#endif
        }
    };
#if CONV_BUF
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    auto issue_b_to = [&](int i, f32x4 (&dstreg)[B_IT]) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(st_base + st_cb), (short)0, 0x7fffffff, 0x00020000);
        dstreg[i] = __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rs, (int)st_voff[i], 0, 0));
    };
#else
    auto issue_b_to = [&](int i, f32x4 (&dstreg)[B_IT]) {
        const int e = tid + i * NTHR;
        const int c8 = e % PER;                                         // float4 of the chunk
        const int po = st_pixoff[i];
        // padding reads 16 B of zeros; explicit global address space (a select between two pointers makes hipcc emit a FLAT
        // load, which counts on lgkmcnt and ties the fragment-read waits to these loads)
        const SAVSR_GLOBAL float* src = po < 0 ? (const SAVSR_GLOBAL float*)g_conv_zero16
                                               : (const SAVSR_GLOBAL float*)st_base + (po * st_pix + st_cb + c8 * 4);
        dstreg[i] = *(const SAVSR_GLOBAL f32x4*)src;
    };
#endif
    auto issue_b = [&](int i) { issue_b_to(i, b_reg); };
    auto stage_issue = [&](int j, int wbuf) {
        if (j < W_IT) issue_w(j, wbuf);
        else if (j < W_IT + B_IT) issue_b(j - W_IT);
    };
    // registers -> LDS buffer `buf`, splitting the activations to (hi, lo) bf16
    auto stage_store_item_from = [&](const f32x4 (&srcreg)[B_IT], int i, int buf) {
        bf16x8* bl = smem + buf * B_UNITS;
        {
            const int e = tid + i * NTHR;
            if (e < B_ITEMS) {
                const int pl = e / PER, c8 = e - pl * PER, q = c8 >> 1, sub = c8 & 1;   // q = kstep * 2 + khalf
                bf16x4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const __bf16 hh = (__bf16)srcreg[i][j];
                    hi[j] = hh;
                    lo[j] = (__bf16)(srcreg[i][j] - (float)hh);
                }
                bf16x4* dst = reinterpret_cast<bf16x4*>(bl + q * NPX + pl) + sub;
                dst[0] = hi;
                dst[B_PART * 2] = lo;
            }
        }
    };
    auto stage_store_item = [&](int i, int buf) { stage_store_item_from(b_reg, i, buf); };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < B_IT; ++i) stage_store_item(i, buf);
    };

    struct Frag { bf16x8 ah[NT], al[NT], bh[PXT], bl[PXT]; };

    // diagnostics: accumulated section times of this wave (stamps mode 3): steps after the barrier (+ load issue) | steps before it | wait+split+store | barrier | epilogue
    long long sec[5] = {0, 0, 0, 0, 0};
    long long t_prev = (stamps_on >= 3) ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define CV_MARK(i) do { if (stamps_on >= 3) { const long long t_now = (long long)__builtin_amdgcn_s_memtime(); sec[i] += t_now - t_prev; t_prev = t_now; } } while (0)

    if (prio_group) __builtin_amdgcn_s_setprio(1);
    stamp(stamps_on, 0);
    stamp(stamps_on, 5);

    // ---- software pipeline over the block's phases (tiles x chunks, walked linearly by the staging cursor) ----
    //   LDS buffer c%2 holds phase c.  While phase c computes: phase c+1 sits in (or is arriving into) the staging
    //   registers; at step SB it is split + stored to the other buffer, ONE barrier publishes it, the cursor moves to
    //   phase c+2 and its global loads are issued (a full phase of latency budget).  All fragment reads of phase c are
    //   issued before that barrier (they run two steps ahead), so the last two steps already read phase c+1's first
    //   fragments: the MFMA stream does not stop at phase or tile boundaries.
    constexpr int RING = PXT > 1 ? 2 : 3;                    // fragment sets in registers; reads run RING - 1 steps ahead
    constexpr int LEAD = RING - 1;
    constexpr int SB = STEPS >= LEAD ? STEPS - LEAD : 0;     // the step whose fragment prefetch is the first of the next phase
    // FINE schedule (3x3): the staging traffic is spread over the steps, at most one store and two loads per wave and
    // step -- issued in one burst after the barrier, the 8 waves queue 80 KiB on the CU's 64 B/clk address path and
    // every wave stalls ~1.3 k cycles in front of its next MFMAs.  In phase c: the weight DMA of phase c+1 starts right
    // after barrier(c-1) (its LDS buffer is free from there) and continues in steps 0 .. W_IT-LEAD-1; steps SD .. SB-1
    // split + store the activations of phase c+1 (loaded during phase c-1), the LAST steps in front of the barrier;
    // step LB0 + i = SD + 1 + i re-loads register set i with the activations of phase c+2.  The stores sit late because
    // hipcc's wait in front of store i is "all but the loads issued after load i", and the DMAs it does not see are in
    // that queue too: stored in steps 0 .. B_IT-1 the last store waited for a DMA issued one step (~0.7 us) earlier.
    // Now everything a store or the barrier waits for was issued >= 4 steps before.
    constexpr bool FINE = KS == 3;
    constexpr int SD = SB - B_IT;                            // first activation-store step
    constexpr int LB0 = SD + 1;                              // first activation-load step (the cursor advances there)
    constexpr int NB = B_IT - 1;                             // activation loads younger than the last weight DMA at the barrier
    static_assert(!FINE || (SD >= 0 && LB0 < SB && W_IT - LEAD <= LB0 && LB0 + B_IT <= STEPS), "FINE staging schedule");
    constexpr int N_LD = B_IT + W_IT;
    constexpr int LD_PER = (N_LD + (STEPS - SB) - 1) / (STEPS - SB);
    int st_tile = blockIdx.x, st_chunk = 0;
    auto stage_next = [&]() -> bool {                        // cursor -> following phase; false when the block has no more
        if (st_chunk + 1 < mp.nchunk) { ++st_chunk; stage_advance(); return true; }
        st_tile += gridDim.x;
        if (st_tile >= total) return false;
        st_chunk = 0;
        stage_begin_tile(decode(st_tile));
        return true;
    };
    auto load_frag = [&](int buf, int s, Frag& f) {
        const bf16x8* bbase = smem + buf * B_UNITS + half * NPX + wave * IC + px;
        const bf16x8* abase = smem + 2 * B_UNITS + buf * W_UNITS + lane;
        const int tap = s / KSTEPS, ks = s - tap * KSTEPS;
        const int ky = tap / KS, kx = tap - ky * KS;
        const int bo = ks * 2 * NPX + ky * IC + kx;
#pragma unroll
        for (int r = 0; r < PXT; ++r) {
            f.bh[r] = bbase[bo + r * CONV_TH * IC];
            f.bl[r] = bbase[B_PART + bo + r * CONV_TH * IC];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f.ah[t] = abase[((s * NT + t) * 2 + 0) * 64];
            f.al[t] = abase[((s * NT + t) * 2 + 1) * 64];
        }
    };

    int tile = blockIdx.x;
    int buf = 0;
    bool pend = false;                                       // phase c+1 exists (staging registers hold / are receiving it)
    Frag f[RING];
    if (tile < total) {
        // Prologue: the loads of phase 1 are issued BEFORE phase 0's are waited for (a second register set, live only
        // here), unconditionally (without a phase 1 they re-load phase 0 into registers / a buffer nobody uses), so the
        // block start pays one global latency, not two.  VMEM returns in order: once the activation loads of phase 0
        // have landed (the compiler's wait in front of the split), the older weight DMA of phase 0 has landed too.
        stage_begin_tile(decode(tile));
        f32x4 b0[B_IT];
#pragma unroll
        for (int j = 0; j < W_IT; ++j) issue_w(j, 0);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) issue_b_to(i, b0);
        pend = stage_next();
#pragma unroll
        for (int j = 0; j < W_IT; ++j)
            if (!FINE || j < LEAD) issue_w(j, 1);            // FINE: the rest of phase 1's weight DMA follows in phase 0's steps
#pragma unroll
        for (int i = 0; i < B_IT; ++i) issue_b(i);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) stage_store_item_from(b0, i, 0);
    }
    __syncthreads();
    if (tile < total) {
        load_frag(0, 0, f[0]);
        if (LEAD > 1 && STEPS > 1) load_frag(0, 1, f[1]);
    }
    stamp(stamps_on, 1);

    for (; tile < total; tile += gridDim.x) {
        const TileInfo cur = decode(tile);
        const ConvParams& p = mp.c[cur.conv];
        const int cob = cur.cob, x0 = cur.x0, y0 = cur.y0;

        f32x16 acc[PXT][NT];
#pragma unroll
        for (int r = 0; r < PXT; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;

        // Pixel rows of this wave that lie inside the image (wave-uniform, fixed per tile): in the image's last band (180 rows =
        // 11 x 16 + 4) a wave owns one row or none, and MFMAs on all-zero operands cost the same time and energy as useful ones
        // -- 6 % of the MFMA work of a 16-row-tile launch at 180 rows.  The phase body is instantiated with and without matrix work:
        // every variant stages, reads its fragments (the ring feeds the NEXT tile too) and meets the barriers alike; only the MFMA
        // groups of absent rows are left out.
        const int rows_valid = __builtin_amdgcn_readfirstlane((y0 + wave < H ? 1 : 0) + (PXT > 1 && y0 + wave + CONV_TH < H ? 1 : 0));
        // Epilogue prefetch: the loads the epilogue would start with -- the bias quads of both channel groups and the residual quads of
        // its first (row, channel-group) step -- go out in front of the tile's LAST phase, so their global latency (one exposed
        // HBM round trip per tile and wave, ~1.5 k of a single-tile launch's ~30 k cycles) runs under that phase's MFMAs.  They are
        // older than everything the phase's barrier waits for, so its counted vmcnt is unchanged.  Always issued, always used
        // (absent operands read zeros): hipcc's wait insertion keeps count only of unconditional loads.
        constexpr bool EPI_PF = !DIAG && KS == 3;       // (the 1x1 kernels have no registers to spare: 25 spills)
        [[maybe_unused]] f32x4 pf_bias[NT], pf_r[4];
        auto epi_prefetch = [&]() {
            const float* pb = p.bias;
            const float* pr = p.res1;
            int prpix = p.res1_pix;
            asm volatile("" : "+s"(pb), "+s"(pr), "+s"(prpix));
            const float* zero16 = (const float*)g_conv_zero16;
            const bool full = (PXT > 1) || (cob + 1) * COT <= mp.cout;      // the epilogue path that uses them
            const int c4 = lane & 7;
            const float* b_base = (pb && full) ? pb : zero16;
            const unsigned b_off = (pb && full) ? 4u * (unsigned)(cob * COT + 4 * c4) : 0u, b_step = (pb && full) ? 128u : 0u;
#pragma unroll
            for (int t = 0; t < NT; ++t) pf_bias[t] = ldg4(b_base, b_off + (unsigned)t * b_step);
            const int y = y0 + __builtin_amdgcn_readfirstlane(wave);
            const bool row_ok = pr && full && y < H, x_in = x0 + CONV_TW <= W;
            const float* r1_base = row_ok ? pr : zero16;
            const unsigned off0 = 4u * (unsigned)((y * W + x0 + (lane >> 3)) * prpix + cob * COT + 4 * c4), ustride = 32u * (unsigned)prpix;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = row_ok && (x_in || x0 + (lane >> 3) + 8 * i < W);
                pf_r[i] = ldg4(r1_base, ok ? off0 + (unsigned)i * ustride : 0u);
            }
        };
        auto chunk_body = [&](const int chunk) __attribute__((always_inline)) {
          bool pend2 = false;                               // phase c+2 exists (decided at the barrier)
          auto phase = [&](auto rows_tag) {
            constexpr int R = decltype(rows_tag)::value;    // rows of this wave that get MFMAs
            // per accumulator: lo*hi, hi*lo, hi*hi (the order is part of the numerics), as three MFMA groups so that the
            // staging work of a step can be issued BETWEEN them and run under matrix-pipe time (both waves of a SIMD leave
            // the barrier in lockstep: work placed after the whole MFMA burst is serial to it)
            auto mma_part = [&](const Frag& fr, int part) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if (part == 0) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.al[t], fr.bh[r], acc[r][t], 0, 0, 0);
                        if (part == 1) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.ah[t], fr.bl[r], acc[r][t], 0, 0, 0);
                        if (part == 2) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.ah[t], fr.bh[r], acc[r][t], 0, 0, 0);
                    }
            };
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                if (FINE && s == LB0) pend2 = pend && stage_next();
                if (s == SB) {
                    if (stamps_on >= 3) { asm volatile("" :: "v"(acc[0][0][0])); }
                    CV_MARK(1);
                    if (pend) {
                        if (!FINE) stage_store(buf ^ 1);
                        // the weight DMA of phase c+1 has landed (FINE: the NB activation loads of phase c+2 behind it may fly on)
                        if (FINE && pend2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NB) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    CV_MARK(2);
                    __syncthreads();
                    CV_MARK(3);
                    if (!FINE) pend2 = pend && stage_next();
                }
                if (dbg_nofrag) {
                } else if (s + LEAD < STEPS) load_frag(buf, s + LEAD, f[(s + LEAD) % RING]);
                else if (pend) load_frag(buf ^ 1, s + LEAD - STEPS, f[(s + LEAD) % RING]);
                __builtin_amdgcn_sched_barrier(0);
                // Straight-line step (FINE, product build): the staging pieces are issued unconditionally -- without a next
                // phase they re-stage the cursor's last phase into a buffer nobody reads -- so that the step is ONE basic
                // block and the scheduler can be told to put a few vector instructions behind every MFMA instead of
                // clumps between the MFMA groups (both waves of a SIMD clump at the same time and the matrix pipe drains).
                if (FINE && !DIAG) {
                    mma_part(f[s % RING], 0);
                    if (s >= SD && s < SD + B_IT) stage_store_item(s - SD, buf ^ 1);
                    mma_part(f[s % RING], 1);
                    if (s < SB) { if (LEAD + s < W_IT) issue_w(LEAD + s, buf ^ 1); }
                    else if (s - SB < W_IT) issue_w(s - SB, buf);
                    if (s >= LB0 && s - LB0 < B_IT) issue_b(s - LB0);
                    mma_part(f[s % RING], 2);
#pragma unroll
                    for (int i = 0; i < 3 * R * NT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // up to 3 vector instructions
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // up to one DS write
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
                mma_part(f[s % RING], 0);
                __builtin_amdgcn_sched_barrier(0);
                if (FINE && s >= SD && s < SD + B_IT && pend && !dbg_nostage) stage_store_item(s - SD, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                mma_part(f[s % RING], 1);
                __builtin_amdgcn_sched_barrier(0);
                if (FINE && !dbg_nostage) {
                    if (s < SB) {
                        if (LEAD + s < W_IT && pend) issue_w(LEAD + s, buf ^ 1);
                    } else {
                        if (s - SB < W_IT && pend2) issue_w(s - SB, buf);
                    }
                    if (s >= LB0 && s - LB0 < B_IT && pend2) issue_b(s - LB0);
                } else if (s >= SB && pend2) {
#pragma unroll
                    for (int j = (s - SB) * LD_PER; j < (s - SB + 1) * LD_PER && j < N_LD; ++j) stage_issue(j, buf);
                }
                __builtin_amdgcn_sched_barrier(0);
                mma_part(f[s % RING], 2);
                __builtin_amdgcn_sched_barrier(0);
            }
          };
          if (rows_valid > 0) phase(std::integral_constant<int, PXT>{});
          else phase(std::integral_constant<int, 0>{});
            if (STEPS % RING != 0) {                          // keep the ring aligned: the next phase starts at slots 0 ..
                const Frag n0 = f[STEPS % RING], n1 = f[(STEPS + 1) % RING];
                f[0] = n0;
                if (LEAD > 1) f[1] = n1;
            }
            if (stamps_on >= 3) { asm volatile("" :: "v"(acc[0][0][0])); }
            CV_MARK(0);
            if (chunk == 0 && tile == (int)blockIdx.x) stamp(stamps_on, 2);
            pend = pend2;
            buf ^= 1;
        };
        if (EPI_PF) {
            for (int chunk = 0; chunk + 1 < mp.nchunk; ++chunk) chunk_body(chunk);
            epi_prefetch();
            chunk_body(mp.nchunk - 1);
        } else {
            for (int chunk = 0; chunk < mp.nchunk; ++chunk) chunk_body(chunk);
        }
        if (tile == (int)blockIdx.x) stamp(stamps_on, 3);

        // ---- epilogue: transpose through the wave's private LDS slice, 32 channels at a time ----------------
        // The descriptor fields are read ONCE per tile into pinned scalars: left to the compiler, every use below
        // re-issued its s_load from the run-time-indexed descriptor (144 dependent scalar loads, ~14 k cycles per tile).
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
        const int COUT = mp.cout;
        const bool act_as_max = e_act == SAVSR_ACT_NONE || e_act == SAVSR_ACT_RELU || (e_act == SAVSR_ACT_LRELU && e_slope >= 0.f && e_slope <= 1.f);
        const float slope_eff = e_act == SAVSR_ACT_NONE ? 1.f : (e_act == SAVSR_ACT_RELU ? 0.f : e_slope);
        float* ep_base = reinterpret_cast<float*>(EP_ALIAS ? smem + (buf ^ 1) * B_UNITS : smem + 2 * B_UNITS + 2 * W_UNITS);
        float* ep = ep_base + wave * (32 * EPS);
        const int c4 = lane & 7;                                // lane l always handles channel quad l % 8
        f32x4 psum[PXT][NT];
        // Interior tiles (all but the image's last band / column and a partial channel block) take a path without
        // per-unit bounds tests and with one activation / residual branch per 32-channel group instead of per unit
        // (the epilogue's ~10 scalar branches per unit were a third of its time).
        const bool chan_full = (cob + 1) * COT <= COUT;       // the 16-row variant is only launched with cout % 64 == 0
        const bool x_inside = x0 + CONV_TW <= W;
        if (DIAG && (dbg_all & 256)) {
            // timing experiment: no epilogue body at all (results invalid)
#pragma unroll
            for (int r = 0; r < PXT; ++r)
#pragma unroll
                for (int t = 0; t < NT; ++t) { psum[r][t] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" :: "v"(acc[r][t][0])); }
        } else if (PXT > 1 || chan_full) {
            // (a second instance of this path without per-lane bounds tests for tiles inside the image, chosen once per tile,
            // measured the same twice: A/B/A/B on one box, before and after the wait-count repairs below)
            // loads that do not depend on the accumulators go out first: the bias quads of both channel groups, and the
            // residual quads one (row, channel-group) step ahead of their use (issued next to their use they exposed one
            // global-load latency per step)
            // Every one of these loads is issued UNCONDITIONALLY and used on every path (absent operands read a 16-B block
            // of zeros, out-of-image lanes the tensor's first quad, whose sum is never stored): a load under a branch, or
            // one whose use a path skips, makes hipcc's wait insertion lose count -- it then drained all of the wave's
            // loads (s_waitcnt vmcnt(0)) at the top of EVERY K phase, at the start of every epilogue and right behind
            // each residual prefetch, and copied the prefetched quads around between the branches.
            const int wave_s = __builtin_amdgcn_readfirstlane(wave);          // row tests as scalar branches
            const float* zero16 = (const float*)g_conv_zero16;
            const float* b_base = e_bias ? e_bias : zero16;
            const unsigned b_off = e_bias ? 4u * (unsigned)(cob * COT + 4 * c4) : 0u, b_step = e_bias ? 128u : 0u;
            f32x4 bias4[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (EPI_PF) bias4[t] = pf_bias[t];
                else bias4[t] = (DIAG && (dbg_all & 512)) ? f32x4{0.f, 0.f, 0.f, 0.f} : ldg4(b_base, b_off + (unsigned)t * b_step);   // (512: timing experiment without the bias load)
            }
            const float* r1_base = e_r1 ? e_r1 : zero16;
            f32x4 rr[2][4];
            auto load_r1 = [&](int r, int t, f32x4 (&dst)[4]) {
                const int y = y0 + wave_s + CONV_TH * r;
                const int co = cob * COT + 32 * t + 4 * c4;
                // one integer multiply per group, then uniform strides (v_mul_lo_u32 is a quarter-rate instruction)
                const unsigned off0 = 4u * (unsigned)((y * W + x0 + (lane >> 3)) * e_r1pix + co), ustride = 32u * (unsigned)e_r1pix;
                const bool row_ok = e_r1 && y < H;          // scalar
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = row_ok && (x_inside || x0 + (lane >> 3) + 8 * i < W);
                    dst[i] = ldg4(r1_base, ok ? off0 + (unsigned)i * ustride : 0u);
                }
            };
            if (EPI_PF) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rr[0][i] = pf_r[i];
            } else load_r1(0, 0, rr[0]);
#pragma unroll
            for (int r = 0; r < PXT; ++r) {
                const int y = y0 + wave_s + CONV_TH * r;
#pragma unroll
                for (int t = 0; t < NT; ++t) psum[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int pbase = y * W + x0 + (lane >> 3);                              // unit i is pixel pbase + 8 i
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int gi = r * NT + t;
                    if (gi + 1 < PXT * NT) load_r1((gi + 1) / NT, (gi + 1) % NT, rr[(gi + 1) & 1]);
                    if (y >= H) {                                                        // scalar: rows below the image
                        asm volatile("" :: "v"(bias4[t]), "v"(rr[gi & 1][0]), "v"(rr[gi & 1][1]), "v"(rr[gi & 1][2]), "v"(rr[gi & 1][3]));   // (used on every path, see above)
                        continue;
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {acc[r][t][4 * g], acc[r][t][4 * g + 1], acc[r][t][4 * g + 2], acc[r][t][4 * g + 3]};
                        if (!(DIAG && dbg_nolds)) *reinterpret_cast<f32x4*>(ep + px * EPS + 8 * g + 4 * half) = v;
                    }
                    const int co = cob * COT + 32 * t + 4 * c4;
                    const f32x4 b4 = bias4[t];
                    const unsigned ooff0 = 4u * (unsigned)(pbase * e_opix + co), ostride = 32u * (unsigned)e_opix;   // unit u = pixel pbase + 8 u
                    f32x4 ps = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ih = 0; ih < 2; ++ih) {                  // two units (pixels pbase + 16 ih, + 8) at a time: register budget
                        const int p0 = pbase + 16 * ih;
                        const bool ok0 = x_inside || x0 + (lane >> 3) + 16 * ih < W, ok1 = x_inside || x0 + (lane >> 3) + 16 * ih + 8 < W;
                        f32x4 v[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            f32x4 a4;
                            if (DIAG && dbg_nolds) a4 = f32x4{acc[r][t][8 * ih + 4 * i], acc[r][t][8 * ih + 4 * i + 1], acc[r][t][8 * ih + 4 * i + 2], acc[r][t][8 * ih + 4 * i + 3]};
                            else a4 = *reinterpret_cast<const f32x4*>(ep + ((lane >> 3) + 16 * ih + 8 * i) * EPS + 4 * c4);
                            v[i] = a4 + b4;           // (whole-vector forms: two v_pk_add_f32 / v_pk_mul_f32 per quad; written per element
                        }                             //  hipcc issued 4 scalar instructions each, and the epilogue is vector-issue-bound)
                        if (e_act == SAVSR_ACT_NONE) {
                            // nothing to compute (half of the launches: second convs of the residual blocks, merges)
                        } else if (act_as_max) {      // ReLU / LeakyReLU(0..1) as ONE v_max_f32(v, v * s), s = 0 / slope: no branch chain (its
#pragma unroll                                        // phi copies were 300 v_mov per tile) and no NaN-canonicalising second v_max (fmaxf
                            for (int i = 0; i < 2; ++i) {   // costs two); ReLU of a negative value gives -0 instead of +0
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
                        if (e_mul) {
                            const float m0 = ok0 ? ldg1(e_mul, (unsigned)p0) : 0.f, m1 = ok1 ? ldg1(e_mul, (unsigned)(p0 + 8)) : 0.f;
#pragma unroll
                            for (int q = 0; q < 4; ++q) { v[0][q] *= m0; v[1][q] *= m1; }
                        }
                        v[0] += rr[gi & 1][2 * ih];                                                   // zeros without a residual
                        v[1] += rr[gi & 1][2 * ih + 1];
                        if (e_r2) {
                            f32x4 ra = {0.f, 0.f, 0.f, 0.f}, rb = {0.f, 0.f, 0.f, 0.f};
                            if (ok0) ra = ldg4(e_r2, 4u * (unsigned)(p0 * e_r2pix + co));
                            if (ok1) rb = ldg4(e_r2, 4u * (unsigned)((p0 + 8) * e_r2pix + co));
#pragma unroll
                            for (int q = 0; q < 4; ++q) { v[0][q] += e_r2s * ra[q]; v[1][q] += e_r2s * rb[q]; }
                        }
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            if (i == 0 ? ok0 : ok1) {
                                if (!(DIAG && dbg_nost)) stg4(e_out, ooff0 + (unsigned)(2 * ih + i) * ostride, v[i]);
                                else asm volatile("" :: "v"(v[i][0]), "v"(v[i][1]), "v"(v[i][2]), "v"(v[i][3]));
                                if (e_pool) ps += v[i];
                            }
                        }
                    }
                    psum[r][t] = ps;
                }
            }
        } else if constexpr (PXT == 1) {
        if (EPI_PF) asm volatile("" :: "v"(pf_bias[0]), "v"(pf_bias[NT - 1]), "v"(pf_r[0]), "v"(pf_r[1]), "v"(pf_r[2]), "v"(pf_r[3]));   // (zeros on this path; used on every path)
#pragma unroll
        for (int r = 0; r < PXT; ++r) {
        const int y = y0 + wave + CONV_TH * r;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[r][t][4 * g], acc[r][t][4 * g + 1], acc[r][t][4 * g + 2], acc[r][t][4 * g + 3]};
                *reinterpret_cast<f32x4*>(ep + px * EPS + 8 * g + 4 * half) = v;
            }
            psum[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int co = cob * COT + 32 * t + 4 * c4;
            const bool full = co + 3 < COUT;
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (e_bias) {
                if (full) {
                    const f32x4 bv = ldg4(e_bias, 4u * (unsigned)co);
                    b4[0] = bv[0]; b4[1] = bv[1]; b4[2] = bv[2]; b4[3] = bv[3];
                } else {
                    for (int q = 0; q < 4 && co + q < COUT; ++q) b4[q] = ldg1(e_bias, (unsigned)(co + q));
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {                       // 32 px x 8 channel quads = 256 units, 4 per lane
                const int pl = (lane >> 3) + 8 * i;
                const int x = x0 + pl;
                if (y >= H || x >= W || co >= COUT) continue;
                const int pidx = y * W + x;
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ep + pl * EPS + 4 * c4);
                float v[4] = {a4[0] + b4[0], a4[1] + b4[1], a4[2] + b4[2], a4[3] + b4[3]};
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
                if (e_mul) {
                    const float mul = ldg1(e_mul, (unsigned)pidx);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] *= mul;
                }
                if (full) {
                    if (e_r1) {
                        const f32x4 r = ldg4(e_r1, 4u * (unsigned)(pidx * e_r1pix + co));
                        v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
                    }
                    if (e_r2) {
                        const f32x4 r = ldg4(e_r2, 4u * (unsigned)(pidx * e_r2pix + co));
                        v[0] += e_r2s * r[0]; v[1] += e_r2s * r[1]; v[2] += e_r2s * r[2]; v[3] += e_r2s * r[3];
                    }
                    const f32x4 ov = {v[0], v[1], v[2], v[3]};
                    stg4(e_out, 4u * (unsigned)(pidx * e_opix + co), ov);
                    psum[r][t][0] += v[0]; psum[r][t][1] += v[1]; psum[r][t][2] += v[2]; psum[r][t][3] += v[3];
                } else {
                    for (int q = 0; q < 4 && co + q < COUT; ++q) {
                        float vv = v[q];
                        if (e_r1) vv += ldg1(e_r1, (unsigned)(pidx * e_r1pix + co + q));
                        if (e_r2) vv += e_r2s * ldg1(e_r2, (unsigned)(pidx * e_r2pix + co + q));
                        stg1(e_out, (unsigned)(pidx * e_opix + co + q), vv);
                    }
                }
            }
        }
        }
        }
        CV_MARK(4);
        if (e_pool) {
            // AdaptiveAvgPool2d(1) of the tensor just produced (savsr_arch.py:146,515), fused: lanes with equal
            // l % 8 hold the same channel quad -> butterfly over the 8 pixel groups, then the waves are summed in
            // wave order through LDS (deterministic) and one row per 8-row band of the tile is written (the row
            // numbering of savsr_conv_pool_blocks does not depend on the kernel variant).
#pragma unroll
            for (int r = 0; r < PXT; ++r)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int o = 8; o < 64; o <<= 1) {
                        psum[r][t][0] += __shfl_xor(psum[r][t][0], o, 64); psum[r][t][1] += __shfl_xor(psum[r][t][1], o, 64);
                        psum[r][t][2] += __shfl_xor(psum[r][t][2], o, 64); psum[r][t][3] += __shfl_xor(psum[r][t][3], o, 64);
                    }
            __syncthreads();                         // every wave is done with its transpose slice
            float* pl_ = ep_base;
            if (lane < 8)
#pragma unroll
                for (int r = 0; r < PXT; ++r)
#pragma unroll
                    for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(pl_ + (r * CONV_TH + wave) * COT + 32 * t + 4 * lane) = psum[r][t];
            __syncthreads();
            if (tid < COT * PXT) {
                const int r = tid / COT, ch = tid - r * COT;
                float sacc = 0.f;
#pragma unroll
                for (int wv = 0; wv < CONV_TH; ++wv) sacc += pl_[(r * CONV_TH + wv) * COT + ch];
                const int band = cur.ty * PXT + r;                       // 8-row band of the image
                if (cob * COT + ch < COUT && band * CONV_TH < H)
                    stg1(e_pool, (unsigned)((band * mp.ntx + cur.tx) * p.pool_stride + cob * COT + ch), sacc);   // (a FLAT store here makes hipcc force the next VMEM wait of the following tile to vmcnt(0))
            }
            __syncthreads();                         // the slices are reused by the next tile's epilogue
        } else if (EP_ALIAS) {
            __syncthreads();                         // the aliased slices become a staging buffer again
        }
    }
    // The last phases re-stage unconditionally (straight-line steps): no LDS-DMA of this wave may still be in flight when
    // its LDS is released.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (stamps_on == 1) {
        __builtin_amdgcn_s_waitcnt(0);              // diagnostics: include the store drain in the last stamp
        stamp(stamps_on, 4);
    }
#if CONV_HAS_STAMPS
    if (stamps_on >= 3 && tid == (stamps_on - 3) * 64 && blockIdx.x < STAMP_BLOCKS)   // mode 3 + w: sections of wave w
        for (int i = 0; i < 5; ++i) g_conv_stamps[blockIdx.x * STAMP_N + i] = sec[i];
#endif
}

#ifdef SAVSR_DIAG
static int g_conv_diag_host = 0;      // != 0: launch the instrumented kernels (instrumented library only)
#endif

template <int KS, int NT, int PXT>
constexpr size_t conv_lds_bytes() {
    constexpr int TAPS = KS * KS, HALO = KS / 2, KC = conv_kc(KS), KSTEPS = KC / 16;
    constexpr int NPX = (CONV_TH * PXT + 2 * HALO) * (CONV_TW + 2 * HALO);
    return 16ull * 2 * (2 * KSTEPS * 2 * NPX + TAPS * KSTEPS * NT * 2 * 64) + (PXT > 1 ? 0ull : 4ull * CONV_TH * 32 * 36);
}

template <int KS, int NT, int PXT, bool DIAG>
static int launch_conv_impl(const MultiConvParams& mp, hipStream_t st) {
    constexpr size_t lds = conv_lds_bytes<KS, NT, PXT>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_bf16x3_kernel<KS, NT, PXT, DIAG>), (int)lds, "conv")) return rc;
    const int total = mp.nconv * mp.ncob * mp.ntx * mp.nty;
    const int grid = total < CONV_PERSISTENT_BLOCKS ? total : CONV_PERSISTENT_BLOCKS;   // one resident workgroup per CU
    hipLaunchKernelGGL((conv_bf16x3_kernel<KS, NT, PXT, DIAG>), dim3(grid), dim3(64 * CONV_TH), lds, st, mp);
    return check_launch("conv_bf16x3_kernel");
}

template <int KS, int NT, int PXT>
static int launch_conv(const MultiConvParams& mp, hipStream_t st) {
#ifdef SAVSR_DIAG
    if (g_conv_diag_host) return launch_conv_impl<KS, NT, PXT, true>(mp, st);
#endif
    return launch_conv_impl<KS, NT, PXT, false>(mp, st);
}

template <int KS, int NT, int PXT>
static int conv_attr() {
    return ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_bf16x3_kernel<KS, NT, PXT, false>), (int)conv_lds_bytes<KS, NT, PXT>(), "conv");
}
// every product instantiation's dynamic-LDS attribute on the current device (savsr_prepare_device)
int conv_prepare_device() {
    if (int rc = conv_attr<3, 2, 2>()) return rc;
    if (int rc = conv_attr<3, 2, 1>()) return rc;
    if (int rc = conv_attr<3, 1, 1>()) return rc;
    if (int rc = conv_attr<1, 2, 1>()) return rc;
    if (int rc = conv_attr<1, 2, 2>()) return rc;
    if (int rc = conv_attr<1, 1, 1>()) return rc;
    return conv_wy_prepare_device();
}

}  // namespace savsr

using namespace savsr;

#if CONV_HAS_STAMPS
extern "C" int savsr_debug_read_conv_stamps(long long* host, int nblocks) {
    if (!host || nblocks < 1 || nblocks > STAMP_BLOCKS) return fail_arg("debug_read_conv_stamps");
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_stamps), sizeof(long long) * STAMP_N * nblocks);
    return e == hipSuccess ? 0 : (int)e;
}
#endif
#ifdef SAVSR_DIAG
extern "C" int savsr_debug_conv_stamps(int enable) {
    g_conv_diag_host = enable;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps_on), &enable, sizeof(int));
    return e == hipSuccess ? 0 : (int)e;
}

#endif

// Rows of a conv's `pool` output = pixel tiles of its launch.
extern "C" int savsr_conv_pool_blocks(int h, int w) {
    return ((w + CONV_TW - 1) / CONV_TW) * ((h + CONV_TH - 1) / CONV_TH);
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
    // sources, weights and bias are read with 16-B LDS-DMA / vector loads whatever the output width; only the OUTPUT-side
    // tensors (out, residuals: addressed per output channel) fall back to scalar accesses when cout < 4 (the 1-channel mask conv)
    uintptr_t al = reinterpret_cast<uintptr_t>(d->wpacked);
    uintptr_t al_out = reinterpret_cast<uintptr_t>(d->out) | (uintptr_t)(d->out_pix * 4);
    for (int i = 0; i < SAVSR_MAX_SRC; ++i) {
        const bool on = i < d->nsrc;
        if (on && !d->src[i]) return fail_arg("conv: null source");
        p.src[i] = on ? d->src[i] : nullptr;
        p.src_pix[i] = on ? d->src_pix[i] : 0;
        if (on) al |= reinterpret_cast<uintptr_t>(d->src[i]) | (uintptr_t)(d->src_pix[i] * 4);
    }
    if (d->res1) al_out |= reinterpret_cast<uintptr_t>(d->res1) | (uintptr_t)(d->res1_pix * 4);
    if (d->res2) al_out |= reinterpret_cast<uintptr_t>(d->res2) | (uintptr_t)(d->res2_pix * 4);
    if (d->bias) al |= reinterpret_cast<uintptr_t>(d->bias);
    if ((al & 15) || ((al_out & 15) && d->cout >= 4)) {
        set_error("conv: sources / residuals / bias / out / weights must be 16-byte aligned with pixel strides multiple of 4 floats");
        return SAVSR_E_ALIGN;
    }
    p.wimg = reinterpret_cast<const unsigned short*>(d->wpacked);
    p.bias = d->bias; p.act = d->act; p.slope = d->slope;
    p.mul_px = d->mul_px; p.res1 = d->res1; p.res1_pix = d->res1_pix; p.res2 = d->res2; p.res2_pix = d->res2_pix;
    p.res2_scale = d->res2_scale;
    p.out = d->out; p.out_pix = d->out_pix;
    p.pool = d->pool; p.pool_stride = d->pool_stride;
    if (d->pool && (d->cout % 4 || d->pool_stride < d->cout)) return fail_arg("conv: pool needs cout % 4 == 0 and pool_stride >= cout");
    {   // the kernel addresses every tensor with 32-bit offsets
        int64_t max_pix = d->out_pix;
        for (int i = 0; i < d->nsrc; ++i) max_pix = d->src_pix[i] > max_pix ? d->src_pix[i] : max_pix;
        if (d->res1 && d->res1_pix > max_pix) max_pix = d->res1_pix;
        if (d->res2 && d->res2_pix > max_pix) max_pix = d->res2_pix;
        if ((int64_t)d->h * d->w * max_pix * 4 >= (int64_t)1 << 31) return fail_arg("conv: tensors of 2 GiB or more are not supported");
    }
    return 0;
}

// Tiles of the Winograd-y form per (conv, channel block): mp.ntx / mp.nty set by the caller (16-row tiles); `per` = convs x channel blocks of the launch.
// The last h % 16 rows: up to 8 of them can go as strip tiles (their 1 / 2 / 4 row pairs side by side over 8 / 4 / 2 segments per workgroup).
static void wy_tile_plan(int h, int per, int algo, MultiConvParams& mp) {
    const int left = h % 16, pairs = (left + 1) / 2;
    mp.wy_strip_l2 = pairs <= 1 ? 0 : (pairs <= 2 ? 1 : 2);
    mp.wy_full = mp.wy_tiles = mp.ntx * mp.nty;
    if (WY_STRIP && left > 0 && left <= 8) {
        // Strips when they save the persistent grid a ROUND of tiles -- what a launch running ALONE pays for: a last-row tile of the full form is
        // cheap (its idle waves leave the matrix pipe to the others), a strip tile costs a full tile, and with the rounds equal the full form is
        // the faster one (6 x 128->64 at 180x320: 720 / 678 tiles, 3 rounds both, 126.1 against 129.1 us; 24 x 64->64: 12 -> 11 rounds, 366.8 ->
        // 352.9 us; one-clip-at-a-time config 3 with strips in every launch: -0.9 %).  WINOGRAD_Y_THROUGHPUT -- another stream's launch fills
        // the tail, the tile count decides -- also whenever at most 2 row pairs are left (>= 6 of a last-row tile's 8 waves idle: 10 such tiles
        // become 3 or 2): bench line +1.5 ... +1.7 % against +1.0 % by rounds only.  Same bits either way.
        const int segs = 8 >> mp.wy_strip_l2;
        const int full = mp.ntx * (mp.nty - 1), tiles = full + (mp.ntx + segs - 1) / segs;
        const int wgs = CONV_PERSISTENT_BLOCKS;
        if (WY_STRIP == 2 || (pairs <= 2 && algo == SAVSR_CONV_WINOGRAD_Y_THROUGHPUT) || (per * tiles + wgs - 1) / wgs < (per * mp.wy_tiles + wgs - 1) / wgs) {
            mp.wy_full = full;
            mp.wy_tiles = tiles;
        }
    }
}

// Workgroup tiles of a savsr_conv2d_batch launch in the Winograd-y form (host arithmetic only: the plan the launcher applies).
extern "C" int64_t savsr_conv_wy_tile_count(int h, int w, int cout, int nconv, int algo) {
    if (h < 1 || w < 1 || cout < 64 || cout % 64 || nconv < 1 || nconv > CONV_MAX_BATCH ||
        (algo != SAVSR_CONV_WINOGRAD_Y && algo != SAVSR_CONV_WINOGRAD_Y_THROUGHPUT)) return -1;
    MultiConvParams mp;
    mp.ncob = cout / 64;
    mp.ntx = (w + CONV_TW - 1) / CONV_TW;
    mp.nty = (h + 15) / 16;
    wy_tile_plan(h, nconv * mp.ncob, algo, mp);
    return (int64_t)nconv * mp.ncob * mp.wy_tiles;
}

extern "C" int savsr_conv2d_max_batch(void) { return CONV_MAX_BATCH; }

extern "C" int savsr_conv2d_batch(const savsr_conv_desc* descs, int n, void* stream) {
    if (!descs) return fail_arg("conv: null descriptor");
    if (n < 1 || n > CONV_MAX_BATCH) return fail_arg("conv: batch size must be 1..24 (savsr_conv2d_max_batch())");
    MultiConvParams mp;
    for (int i = 0; i < n; ++i) {
        const int rc = fill_params(descs + i, mp.c[i]);
        if (rc) return rc;
        const savsr_conv_desc& a = descs[0];
        const savsr_conv_desc& b = descs[i];
        if (b.ksize != a.ksize || b.nsrc != a.nsrc || b.src_ch != a.src_ch || b.h != a.h || b.w != a.w || b.cout != a.cout || b.algo != a.algo)
            return fail_arg("conv: all convs of a batch must share ksize / nsrc / src_ch / h / w / cout / algo");
    }
    for (int i = n; i < CONV_MAX_BATCH; ++i) mp.c[i] = mp.c[0];
    const savsr_conv_desc* d = descs;
    const int cot = conv_cot(d->cout);
    mp.h = d->h; mp.w = d->w; mp.cout = d->cout;
    mp.nchunk = d->cin / conv_kc(d->ksize);
    mp.src_ch = d->src_ch;
    mp.nconv = n;
    mp.ncob = (d->cout + cot - 1) / cot;
    mp.ntx = (d->w + CONV_TW - 1) / CONV_TW;
    mp.nty = (d->h + CONV_TH - 1) / CONV_TH;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d->algo == SAVSR_CONV_WINOGRAD_Y || d->algo == SAVSR_CONV_WINOGRAD_Y_THROUGHPUT) {
        // wpacked is the Winograd-y image (savsr_conv_wy_pack_index); one kernel, 16-row tiles, whatever the launch size
        if (d->ksize != 3 || d->cout % 64) return fail_arg("conv: algo WINOGRAD_Y needs ksize 3 and cout a multiple of 64");
        mp.nty = (d->h + 15) / 16;
        wy_tile_plan(d->h, n * mp.ncob, d->algo, mp);
        return launch_conv_wy(mp, st);
    }
    if (d->algo != SAVSR_CONV_DIRECT && d->algo != SAVSR_CONV_DIRECT_THROUGHPUT) return fail_arg("conv: unknown algo");
    const bool wide = cot == 64;
    if (d->ksize == 3 && wide) {
        // 16-row tiles (each wave 64 channels x 2 rows: one weight-fragment read feeds two pixel rows, 2/3 of the LDS
        // traffic per MFMA and half the barriers) once they still fill the chip; 8-row tiles for small launches
        const int nty2 = (d->h + 2 * CONV_TH - 1) / (2 * CONV_TH);
        // A launch of 100 .. 199 such tiles (a single 64 -> 64 conv at 180x320: 120) fills half the chip: alone it is slower than
        // 230 8-row tiles (26 vs 19-22 us), with other streams' launches beside it the aggregate is faster (bench +2.4 %)
        if (d->cout % 64 == 0 && n * mp.ncob * mp.ntx * nty2 >= (d->algo == SAVSR_CONV_DIRECT_THROUGHPUT ? CONV_WIDE_MIN_TILES_TP : CONV_WIDE_MIN_TILES)) {
            mp.nty = nty2;
            return launch_conv<3, 2, 2>(mp, st);
        }
        return launch_conv<3, 2, 1>(mp, st);
    }
    if (d->ksize == 3) return launch_conv<3, 1, 1>(mp, st);
    if (wide && d->cout % 64 == 0) {               // 1x1: 16-row tiles from the same tile counts up (half the barriers and fragment reads per pixel)
        const int nty2 = (d->h + 2 * CONV_TH - 1) / (2 * CONV_TH);
        if (n * mp.ncob * mp.ntx * nty2 >= (d->algo == SAVSR_CONV_DIRECT_THROUGHPUT ? CONV_WIDE_MIN_TILES_TP : CONV_WIDE_MIN_TILES)) {
            mp.nty = nty2;
            return launch_conv<1, 2, 2>(mp, st);
        }
    }
    return wide ? launch_conv<1, 2, 1>(mp, st) : launch_conv<1, 1, 1>(mp, st);
}

extern "C" int savsr_conv2d(const savsr_conv_desc* d, void* stream) { return savsr_conv2d_batch(d, 1, stream); }
