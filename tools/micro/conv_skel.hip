// Developer micro-benchmark: what would a 1-D Winograd F(2,3) form of the 3x3 conv cost on gfx950
// against the direct form, WITH the weight and activation traffic in the loop?  Two K-loop skeletons over the same conv work
// per phase (a 16-row x 32-pixel x 64-output-channel tile, 16 input channels, split-bf16 products), 8 waves per workgroup,
// one persistent workgroup per CU, random data, real staging (global fp32 loads -> (transform) -> (hi, lo) bf16 split -> LDS,
// weight slabs by LDS-DMA), no epilogue:
//
//   MODE 0  direct (the product kernel's shape): 9 taps x 12 MFMAs per wave and phase, 8 fragment reads per 12 MFMAs, B image
//           18 x 34 px shared by the waves (double-buffered, 2 x 39 KB), weight slab 36 KB x 2, ONE barrier per phase,
//           5 float4 loads + 60 VALU + 10 ds_write_b64 per lane and phase.
//   MODE 1  Winograd F(2,3) ALONG Y: wave w owns the output row pair (2w, 2w+1); its four transformed rows
//           V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3 (d = input rows 2w-1 .. 2w+2, 34 px) live in a
//           WAVE-PRIVATE LDS region (no cross-wave activation sharing, so no barrier and no double buffer for them: positions
//           {0,1} are rewritten while {2,3} are being read and vice versa); weights = 4 positions x 3 kx "taps" (48 KB per
//           phase) in two halves of 24 KB, each half republished behind its own barrier (2 per phase);
//           per (position, kx): 4 A + 2 B fragment reads -> 6 MFMAs; 72 MFMAs per wave and phase instead of 108;
//           8 accumulators (128 registers); staging per lane and phase: 12 float4 loads, 16 transform + 48 split VALU per
//           round x 3 rounds, 24 ds_write_b64.
// Prints us per launch, cycles per phase, clock.  Equal conv work per phase in both modes, so us (not TFLOP/s) compares.
// Build: hipcc --offload-arch=gfx950 -O3 conv_skel.hip -o conv_skel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL __attribute__((address_space(1)))

constexpr int IC = 34;
// MODE 0 geometry
constexpr int D_NPX = 18 * IC, D_BPART = 2 * D_NPX, D_BUNITS = 2 * D_BPART, D_WUNITS = 9 * 2 * 2 * 64;   // 16-B units
// MODE 1 geometry
constexpr int W_VWAVE = 2 * 2 * 2 * 2 * IC;      // [half][row in half][part][q][px] units per wave = 544
constexpr int W_WHALF = 2 * 3 * 2 * 2 * 64;      // [pos in half][kx][t][part][lane] = 1536 units = 24 KB
// MODE 3 geometry (F(4,3) along y)
constexpr int F_VQUAD = 2 * 3 * 2 * 2 * IC;      // [half][pos in half][part][q][px] units per row QUAD = 816 (13 KB); 4 quads per workgroup
constexpr int F_WHALF = 3 * 3 * 2 * 2 * 64;      // [pos in half][kx][t][part][lane] = 2304 units = 36.9 KB

__device__ __forceinline__ void dma1k(const void* src_lane, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src_lane), "s"(lds_wave_base) : "memory");
}
template <int X>
__device__ __forceinline__ void split_store(const f32x4& v, bf16x4* hi_dst, bf16x4* lo_dst) {
    if (X & 1) {                                      // timing knob: no split arithmetic, the raw bits go to LDS
        union { f32x4 f; bf16x4 h[2]; } u;
        u.f = v;
        *hi_dst = u.h[0];
        *lo_dst = u.h[1];
        return;
    }
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

template <int MODE, int X>
__global__ __launch_bounds__(512) void k(const float* __restrict__ act, long long act_floats, const bf16x8* __restrict__ wts, float* out, long long* cyc, int phases) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    bf16x8* lds = reinterpret_cast<bf16x8*>(raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, px = lane & 31;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    float sum = 0.f;
    long long t0 = 0, t1 = 0;
    // every workgroup walks its own stretch of the activation buffer (distinct bytes per phase, like a real tile sequence)
    const long long wg_base = ((long long)blockIdx.x * 1315423911ll) % (act_floats - (long long)phases * 16384 - 65536);
    if (MODE == 0) {
        constexpr int B_IT = 5;                                   // 18 x 34 px x 4 float4 = 2448 items / 512 threads
        f32x16 acc[2][2];
        for (int r = 0; r < 2; ++r) for (int t = 0; t < 2; ++t) for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;
        f32x4 breg[B_IT];
        float dummy[4] = {1.f + lane, 2.f, 3.f, 4.f};
        for (int i = tid; i < 2 * D_BUNITS + 2 * D_WUNITS; i += 512) lds[i] = wts[i & 8191];
        __syncthreads();
        struct Frag { bf16x8 ah[2], al[2], bh[2], bl[2]; } f[2];
        auto load_frag = [&](int buf, int s, Frag& fr) {
            const bf16x8* bbase = lds + buf * D_BUNITS + half * D_NPX + wave * IC + px;
            const bf16x8* abase = lds + 2 * D_BUNITS + buf * D_WUNITS + lane;
            const int ky = s / 3, kx = s - 3 * ky, bo = ky * IC + kx;
#pragma unroll
            for (int r = 0; r < 2; ++r) { fr.bh[r] = bbase[bo + r * 8 * IC]; fr.bl[r] = bbase[D_BPART + bo + r * 8 * IC]; }
#pragma unroll
            for (int t = 0; t < 2; ++t) { fr.ah[t] = abase[((s * 2 + t) * 2 + 0) * 64]; fr.al[t] = abase[((s * 2 + t) * 2 + 1) * 64]; }
        };
        auto issue_b = [&](int i, int ph) {
            if (X & 8) { asm volatile("" : "+v"(breg[i])); return; }
            const int e = tid + i * 512;
            const int pl = e >> 2, c4 = e & 3;
            const long long off = wg_base + (long long)ph * 16384 + (e < 2448 ? pl * 16 + c4 * 4 : 0);
            breg[i] = *(const GLOBAL f32x4*)((const GLOBAL float*)act + off);
        };
        auto store_b = [&](int i, int buf) {
            const int e = tid + i * 512;
            if (e < 2448) {
                const int pl = e >> 2, c8 = e & 3, q = c8 >> 1, sub = c8 & 1;
                bf16x4* dst = reinterpret_cast<bf16x4*>(lds + buf * D_BUNITS + q * D_NPX + pl) + sub;
                split_store<X>(breg[i], dst, dst + D_BPART * 2);
            }
        };
        auto issue_w = [&](int g, int buf, int ph) {
            if (!(X & 2) && g * 512 + wave_s * 64 < D_WUNITS)
                dma1k(wts + ((ph & 7) * D_WUNITS + g * 512 + tid) % 8192 * 1 + ((ph & 7) * 2304 % 4096), (unsigned)(uintptr_t)(lds + 2 * D_BUNITS + buf * D_WUNITS + g * 512 + wave_s * 64));
        };
#pragma unroll
        for (int i = 0; i < B_IT; ++i) issue_b(i, 0);
        load_frag(0, 0, f[0]);
        int buf = 0;
        t0 = __builtin_amdgcn_s_memtime();
        for (int ph = 0; ph < phases; ++ph) {
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                if (s == 8) {
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    __syncthreads();
                }
                if (X & 4) { asm volatile("" : "+v"(f[(s + 1) & 1].ah[0]), "+v"(f[(s + 1) & 1].bh[0])); }
                else if (s + 1 < 9) load_frag(buf, s + 1, f[(s + 1) & 1]); else load_frag(buf ^ 1, 0, f[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const Frag& fr = f[s & 1];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.al[t], fr.bh[r], acc[r][t], 0, 0, 0);
                if (s >= 3 && s < 8) store_b(s - 3, buf ^ 1);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.ah[t], fr.bl[r], acc[r][t], 0, 0, 0);
                if (s < 5) issue_w(s, buf ^ 1, ph + 1);
                if (s >= 4) issue_b(s - 4, ph + 2);
#pragma unroll
                for (int v = 0; v < (X >> 4) * 4; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(dummy[v & 3]) : "v"(1.0001f));
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.ah[t], fr.bh[r], acc[r][t], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            buf ^= 1;
        }
        t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int r = 0; r < 2; ++r) for (int t = 0; t < 2; ++t) for (int i = 0; i < 16; ++i) sum += acc[r][t][i];
        for (int i = 0; i < B_IT; ++i) sum += breg[i][0];
        sum += dummy[0] + dummy[1] + dummy[2] + dummy[3];
    } else if (MODE == 2) {
        // direct, B-operand reuse: wave w owns the ADJACENT output rows (2w, 2w+1), so the four input rows 2w-1 .. 2w+2 serve both
        // (row ky + r feeds output row r at tap row ky), and the three horizontal taps are ONE read: the kx = 1, 2 operands are the kx = 0
        // registers shifted by a lane (DPP wave_shl:1) with lanes 31 / 63 patched by a two-lane LDS read.  Per phase: 8 full B reads + 16
        // two-lane patch reads + 36 A reads (44 KB instead of 72 KB of fragment traffic per wave), 64 DPP moves; same MFMAs as MODE 0.
        constexpr int B_IT = 5;
        f32x16 acc[2][2];
        for (int r = 0; r < 2; ++r) for (int t = 0; t < 2; ++t) for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;
        f32x4 breg[B_IT];
        for (int i = tid; i < 2 * D_BUNITS + 2 * D_WUNITS; i += 512) lds[i] = wts[i & 8191];
        __syncthreads();
        struct AF { bf16x8 ah[2], al[2]; } fa[2];
        bf16x8 bb[4][2], nb[4][2];                                // [input row][hi, lo]: current phase (shifted in place), next phase
        auto load_a = [&](int buf, int s, AF& fr) {
            const bf16x8* abase = lds + 2 * D_BUNITS + buf * D_WUNITS + lane;
#pragma unroll
            for (int t = 0; t < 2; ++t) { fr.ah[t] = abase[((s * 2 + t) * 2 + 0) * 64]; fr.al[t] = abase[((s * 2 + t) * 2 + 1) * 64]; }
        };
        auto load_b = [&](int buf, bf16x8 (&dst)[4][2]) {
            const bf16x8* bbase = lds + buf * D_BUNITS + half * D_NPX + (2 * wave) * IC + px;
#pragma unroll
            for (int r = 0; r < 4; ++r) { dst[r][0] = bbase[r * IC]; dst[r][1] = bbase[D_BPART + r * IC]; }
        };
        auto shift_b = [&](int buf, int kx) {                     // lane px <- lane px + 1; px = 31 re-reads its pixel (32 + kx - 1 + ...) from LDS
            const bf16x8* bbase = lds + buf * D_BUNITS + half * D_NPX + (2 * wave) * IC + 31 + kx;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    union { bf16x8 v; int i[4]; } u, o;
                    u.v = bb[r][p];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o.i[j] = __builtin_amdgcn_update_dpp(u.i[j], u.i[j], 0x130, 0xf, 0xf, false);   // wave_shl:1
                    bb[r][p] = o.v;
                    if (px == 31) bb[r][p] = bbase[p * D_BPART + r * IC];
                }
        };
        auto issue_b = [&](int i, int ph) {
            const int e = tid + i * 512;
            const int pl = e >> 2, c4 = e & 3;
            const long long off = wg_base + (long long)ph * 16384 + (e < 2448 ? pl * 16 + c4 * 4 : 0);
            breg[i] = *(const GLOBAL f32x4*)((const GLOBAL float*)act + off);
        };
        auto store_b = [&](int i, int buf) {
            const int e = tid + i * 512;
            if (e < 2448) {
                const int pl = e >> 2, c8 = e & 3, q = c8 >> 1, sub = c8 & 1;
                bf16x4* dst = reinterpret_cast<bf16x4*>(lds + buf * D_BUNITS + q * D_NPX + pl) + sub;
                split_store<X>(breg[i], dst, dst + D_BPART * 2);
            }
        };
        auto issue_w = [&](int g, int buf, int ph) {
            if (g * 512 + wave_s * 64 < D_WUNITS)
                dma1k(wts + ((ph & 7) * D_WUNITS + g * 512 + tid) % 8192 * 1 + ((ph & 7) * 2304 % 4096), (unsigned)(uintptr_t)(lds + 2 * D_BUNITS + buf * D_WUNITS + g * 512 + wave_s * 64));
        };
#pragma unroll
        for (int i = 0; i < B_IT; ++i) issue_b(i, 0);
        load_a(0, 0, fa[0]);
        load_b(0, nb);
        int buf = 0;
        t0 = __builtin_amdgcn_s_memtime();
        for (int ph = 0; ph < phases; ++ph) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { bb[r][0] = nb[r][0]; bb[r][1] = nb[r][1]; }
#pragma unroll
            for (int s = 0; s < 9; ++s) {                          // s = kx * 3 + ky
                const int kx = s / 3, ky = s - 3 * kx;
                if (s == 8) {
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    __syncthreads();
                    load_b(buf ^ 1, nb);
                }
                if (s + 1 < 9) load_a(buf, (s + 1) % 3 * 3 + (s + 1) / 3, fa[(s + 1) & 1]); else load_a(buf ^ 1, 0, fa[(s + 1) & 1]);
                if (ky == 0 && kx > 0) shift_b(buf, kx);
                __builtin_amdgcn_sched_barrier(0);
                const AF& fr = fa[s & 1];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.al[t], bb[ky + r][0], acc[r][t], 0, 0, 0);
                if (s >= 3 && s < 8) store_b(s - 3, buf ^ 1);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.ah[t], bb[ky + r][1], acc[r][t], 0, 0, 0);
                if (s < 5) issue_w(s, buf ^ 1, ph + 1);
                if (s >= 4) issue_b(s - 4, ph + 2);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.ah[t], bb[ky + r][0], acc[r][t], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            buf ^= 1;
        }
        t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int r = 0; r < 2; ++r) for (int t = 0; t < 2; ++t) for (int i = 0; i < 16; ++i) sum += acc[r][t][i];
        for (int i = 0; i < B_IT; ++i) sum += breg[i][0];
    } else if (MODE == 3) {
        // Winograd F(4,3) ALONG Y (round 5 costing).  6 positions x 3 kx = 18 taps for FOUR output rows (direct 36, F(2,3) 24).
        // A wave cannot own a row quad with all 64 output channels (6 positions x 2 channel blocks x 16 = 192 accumulator registers), so
        // wave = (row quad = wave / 2, 32-channel block = wave % 2): 6 accumulators (96 registers), 54 MFMAs per wave and phase for the same
        // 16-row x 32-px x 64-channel tile.  The two waves of a quad SHARE the quad's transformed rows V0..V5 (each builds half of the 136
        // (px, channel quad) columns); the two barriers per phase that publish the weight halves publish the V halves too (positions {0,1,2}
        // are rewritten while {3,4,5} are read and vice versa, as in MODE 1).  Weights 18 taps = 73.7 KB per phase in two halves of 36.9 KB.
        // Per step (position, kx): 2 A + 2 B fragment reads -> 3 MFMAs.  Staging per lane and phase: 7 float4 loads (6 rows of one column +
        // the tail), ~56 transform + 72 split VALU, 12 ds_write_b64; 9 (waves 0-3) or 8 weight DMAs of 1 KiB per wave (F(2,3): 6).
        f32x16 acc[6];
        for (int p = 0; p < 6; ++p) for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;
        f32x4 d[6];                                               // rows d0..d5 of this lane's (px, quad) column
        f32x4 dx;                                                 // tail: the last 4 columns x 6 rows of this wave's share, one (row, column) per lane < 24
        const int quad = wave >> 1, cob = wave & 1;
        bf16x8* vbase = lds + quad * F_VQUAD;                     // [hf][pos][part][q][px]
        bf16x8* wbase = lds + 4 * F_VQUAD;                        // [hf][2304]
        for (int i = tid; i < 4 * F_VQUAD + 2 * F_WHALF; i += 512) lds[i] = wts[i & 8191];
        __syncthreads();
        struct Frag { bf16x8 ah, al, bh, bl; } f[2];
        auto load_frag = [&](int hf, int s, Frag& fr) {           // s = pos-in-half * 3 + kx
            const int pr = s / 3, kx = s - 3 * pr;
            const bf16x8* bb = vbase + ((hf * 3 + pr) * 2 * 2 + half) * IC + px + kx;
            fr.bh = bb[0];
            fr.bl = bb[2 * IC];
            const bf16x8* ab = wbase + hf * F_WHALF + s * 256 + lane;
            fr.ah = ab[(cob * 2 + 0) * 64];
            fr.al = ab[(cob * 2 + 1) * 64];
        };
        const int col0 = cob * 68 + lane;                         // this lane's column of the quad's 136 (px, channel quad) columns
        auto issue_d = [&](int row, int ph) {                     // row 0..5: one float4 per lane; row 6: the tail item
            if (X & 8) { if (row == 6) { asm volatile("" : "+v"(dx)); } else { asm volatile("" : "+v"(d[row])); } return; }
            if (row == 6) {
                const int r = lane >> 2, col = cob * 68 + 64 + (lane & 3);
                const long long off = wg_base + (long long)ph * 16384 + ((4 * quad + (r < 6 ? r : 0)) * IC + (col >> 2)) * 16 + (col & 3) * 4;
                dx = *(const GLOBAL f32x4*)((const GLOBAL float*)act + off);
                return;
            }
            const long long off = wg_base + (long long)ph * 16384 + ((4 * quad + row) * IC + (col0 >> 2)) * 16 + (col0 & 3) * 4;
            d[row] = *(const GLOBAL f32x4*)((const GLOBAL float*)act + off);
        };
        // transform + split + store of position pr (0..2) of half hf from the six rows in registers
        auto store_v = [&](int pr, int hf) {
            f32x4 v;
            if (X & 1) v = d[hf * 3 + pr];
            else if (hf == 0) v = pr == 0 ? 4.f * d[0] - 5.f * d[2] + d[4] : (pr == 1 ? (d[4] - 4.f * d[2]) + (d[3] - 4.f * d[1]) : (d[4] - 4.f * d[2]) - (d[3] - 4.f * d[1]));
            else v = pr == 0 ? (d[4] - d[2]) + 2.f * (d[3] - d[1]) : (pr == 1 ? (d[4] - d[2]) - 2.f * (d[3] - d[1]) : 4.f * d[1] - 5.f * d[3] + d[5]);
            const int pxl = col0 >> 2, c8 = col0 & 3, q = c8 >> 1, sub = c8 & 1;
            bf16x4* dst = reinterpret_cast<bf16x4*>(vbase + ((hf * 3 + pr) * 2 * 2 + q) * IC + pxl) + sub;
            split_store<X>(v, dst, dst + 2 * IC * 2);
        };
        auto store_tail = [&](int hf) {                           // the 4 tail columns: rows sit 4 lanes apart; three shuffles + adds stand in for the row combination
            f32x4 o1, o2, o3;
#pragma unroll
            for (int j = 0; j < 4; ++j) { o1[j] = __shfl_xor(dx[j], 4, 64); o2[j] = __shfl_xor(dx[j], 8, 64); o3[j] = __shfl_xor(dx[j], 16, 64); }
            const f32x4 v = (lane & 4) ? (dx - 4.f * o1) + (o2 - 4.f * o3) : 4.f * dx - 5.f * o2 + o3;
            if (lane < 12) {
                const int col = cob * 68 + 64 + (lane & 3), pxl = col >> 2, c8 = col & 3, q = c8 >> 1, sub = c8 & 1;
                bf16x4* dst = reinterpret_cast<bf16x4*>(vbase + ((hf * 3 + (lane >> 2)) * 2 * 2 + q) * IC + pxl) + sub;
                split_store<X>(v, dst, dst + 2 * IC * 2);
            }
        };
        auto issue_w = [&](int g, int hf, int ph) {              // 36 pieces of 1 KiB per half: waves 0-3 five, waves 4-7 four
            if (!(X & 2) && g * 8 + wave_s < 36)
                dma1k(wts + ((ph & 7) * 512 + hf * 2304 + (g * 8 + wave_s) * 64 + lane) % 8192, (unsigned)(uintptr_t)(wbase + hf * F_WHALF + (g * 8 + wave_s) * 64));
        };
#pragma unroll
        for (int r = 0; r < 7; ++r) issue_d(r, 0);
        load_frag(0, 0, f[0]);
        t0 = __builtin_amdgcn_s_memtime();
        for (int ph = 0; ph < phases; ++ph) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    if (s == 8) {                                 // the other half's weights and V rows are published
                        if (hf == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");               // six row loads of the next-but-one phase are younger than the last DMA
                        __syncthreads();
                    }
                    if (X & 4) { asm volatile("" : "+v"(f[(s + 1) & 1].ah), "+v"(f[(s + 1) & 1].bh)); }
                    else if (s + 1 < 9) load_frag(hf, s + 1, f[(s + 1) & 1]); else load_frag(hf ^ 1, 0, f[(s + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    const Frag& fr = f[s & 1];
                    const int p = hf * 3 + s / 3;
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.al, fr.bh, acc[p], 0, 0, 0);
                    if (s >= 1 && s < 4) store_v(s - 1, hf ^ 1);
                    if (s == 4) store_tail(hf ^ 1);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.ah, fr.bl, acc[p], 0, 0, 0);
                    if (s < 5) issue_w(s, hf ^ 1, ph + 1);
                    if (hf == 1 && s >= 2) issue_d(s - 2, ph + 2);  // rows 0..5 at s = 2..7, the tail item at s = 8
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.ah, fr.bh, acc[p], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int p = 0; p < 6; ++p) for (int i = 0; i < 16; ++i) sum += acc[p][i];
        for (int r = 0; r < 6; ++r) sum += d[r][0];
        sum += dx[0];
    } else {
        constexpr int RND = 3;                                   // 34 px x 4 channel quads = 136 (px, quad) columns per wave / 64 lanes
        f32x16 acc[4][2];                                         // [position][co block]
        for (int p = 0; p < 4; ++p) for (int t = 0; t < 2; ++t) for (int i = 0; i < 16; ++i) acc[p][t][i] = 0.f;
        f32x4 d[RND - 1][4];                                      // rows d0..d3 of the wave's (px, quad) columns, rounds 0 and 1 (128 columns)
        f32x4 dx;                                                 // round 2: the last 8 columns x 4 rows, one (row, column) per lane < 32 (rows combined across lanes)
        bf16x8* vbase = lds + wave * W_VWAVE;                     // [hf][row][part][q][px]
        bf16x8* wbase = lds + 8 * W_VWAVE;                        // [hf][1536]
        for (int i = tid; i < 8 * W_VWAVE + 2 * W_WHALF; i += 512) lds[i] = wts[i & 8191];
        __syncthreads();
        struct Frag { bf16x8 ah[2], al[2], bh, bl; } f[2];
        auto load_frag = [&](int hf, int s, Frag& fr) {           // s = pos-in-half * 3 + kx
            const int pr = s / 3, kx = s - 3 * pr;
            const bf16x8* bb = vbase + ((hf * 2 + pr) * 2 * 2 + half) * IC + px + kx;       // part 0 (hi), q = half
            fr.bh = bb[0];
            fr.bl = bb[2 * IC];
            const bf16x8* ab = wbase + hf * W_WHALF + s * 256 + lane;
#pragma unroll
            for (int t = 0; t < 2; ++t) { fr.ah[t] = ab[(t * 2 + 0) * 64]; fr.al[t] = ab[(t * 2 + 1) * 64]; }
        };
        auto issue_d = [&](int r, int ph) {
            if (X & 8) { if (r == 2) { asm volatile("" : "+v"(dx)); } else { asm volatile("" : "+v"(d[r][0]), "+v"(d[r][1]), "+v"(d[r][2]), "+v"(d[r][3])); } return; }
            if (r == 2) {
                const int row = (lane >> 3) & 3, col = 128 + (lane & 7);
                const long long off = wg_base + (long long)ph * 16384 + ((2 * wave + row) * IC + (col >> 2)) * 16 + (col & 3) * 4;
                dx = *(const GLOBAL f32x4*)((const GLOBAL float*)act + off);
                return;
            }
            const int col = lane + 64 * r;                        // (px, quad) column
            const int pxl = col >> 2, c4 = col & 3;
#pragma unroll
            for (int row = 0; row < 4; ++row) {
                const long long off = wg_base + (long long)ph * 16384 + ((2 * wave + row) * IC + pxl) * 16 + c4 * 4;
                d[r][row] = *(const GLOBAL f32x4*)((const GLOBAL float*)act + off);
            }
        };
        // transform + split + store of round r's two rows of half hf (hf 0: V0 = d0 - d2, V1 = d1 + d2; hf 1: V2 = d2 - d1, V3 = d1 - d3)
        auto store_v = [&](int r, int hf) {
            if (r == 2) {                                         // the 8 tail columns: the partner row arrives by a cross-lane move (rows are 8 lanes apart)
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = __shfl_xor(dx[j], hf == 0 ? 16 : 8, 64);
                const f32x4 va = (lane & 8) ? dx + o : dx - o;
                if (lane < 16) {
                    const int col = 128 + (lane & 7), pxl = col >> 2, c8 = col & 3, q = c8 >> 1, sub = c8 & 1;
                    bf16x4* dst = reinterpret_cast<bf16x4*>(vbase + ((hf * 2 + ((lane >> 3) & 1)) * 2 * 2 + q) * IC + pxl) + sub;
                    split_store<X>(va, dst, dst + 2 * IC * 2);
                }
                return;
            }
            const int col = lane + 64 * r;
            {
                const int pxl = col >> 2, c8 = col & 3, q = c8 >> 1, sub = c8 & 1;
                const f32x4 va = (X & 1) ? d[r][hf] : (hf == 0 ? d[r][0] - d[r][2] : d[r][2] - d[r][1]);
                const f32x4 vb = (X & 1) ? d[r][2 + hf] : (hf == 0 ? d[r][1] + d[r][2] : d[r][1] - d[r][3]);
                bf16x4* dst = reinterpret_cast<bf16x4*>(vbase + ((hf * 2 + 0) * 2 * 2 + q) * IC + pxl) + sub;
                split_store<X>(va, dst, dst + 2 * IC * 2);
                bf16x4* dst2 = reinterpret_cast<bf16x4*>(vbase + ((hf * 2 + 1) * 2 * 2 + q) * IC + pxl) + sub;
                split_store<X>(vb, dst2, dst2 + 2 * IC * 2);
            }
        };
        auto issue_w = [&](int g, int hf, int ph) {              // 24 pieces of 1 KiB per half / 8 waves = 3 per wave
            if (!(X & 2)) dma1k(wts + ((ph & 7) * 512 + hf * 1536 + g * 512 + tid) % 8192, (unsigned)(uintptr_t)(wbase + hf * W_WHALF + g * 512 + wave_s * 64));
        };
#pragma unroll
        for (int r = 0; r < RND; ++r) issue_d(r, 0);
        load_frag(0, 0, f[0]);
        t0 = __builtin_amdgcn_s_memtime();
        for (int ph = 0; ph < phases; ++ph) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    if (s == 5) {                                 // the other half's weights (DMA'd during the previous half) are published
                        if (hf == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMAs are the youngest operations
                        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");               // 2 x 4 row loads of the next-but-one phase are younger
                        __syncthreads();
                    }
                    if (X & 4) { asm volatile("" : "+v"(f[(s + 1) & 1].ah[0]), "+v"(f[(s + 1) & 1].bh)); }
                    else if (s + 1 < 6) load_frag(hf, s + 1, f[(s + 1) & 1]); else load_frag(hf ^ 1, 0, f[(s + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    const Frag& fr = f[s & 1];
                    const int p = hf * 2 + s / 3;
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.al[t], fr.bh, acc[p][t], 0, 0, 0);
                    // while half hf computes: the OTHER half's V rows of the next use are rebuilt (hf 0: half 1 of this phase's data
                    // was read in the previous phase... skeleton: half 1 - hf gets the rows of the phase whose loads have landed)
                    if (s >= 1 && s < 4) store_v(s - 1, hf ^ 1);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.ah[t], fr.bl, acc[p][t], 0, 0, 0);
                    if (s < 3) issue_w(s, hf ^ 1, ph + 1);
                    if (hf == 1 && s >= 3) issue_d(s - 3, ph + 2);  // the d registers are free once both halves of their phase are stored
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.ah[t], fr.bh, acc[p][t], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int p = 0; p < 4; ++p) for (int t = 0; t < 2; ++t) for (int i = 0; i < 16; ++i) sum += acc[p][t][i];
        for (int r = 0; r < RND - 1; ++r) sum += d[r][0][0];
        sum += dx[0];
    }
    out[blockIdx.x * 512 + tid] = sum;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int X = 0>
void run(const char* name, const float* act, long long act_floats, const bf16x8* wts, int reps, int phases) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int lds = MODE == 3 ? (4 * F_VQUAD + 2 * F_WHALF) * 16 : (MODE != 1 ? (2 * D_BUNITS + 2 * D_WUNITS) * 16 : (8 * W_VWAVE + 2 * W_WHALF) * 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, X>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<MODE, X>), dim3(256), dim3(512), lds, 0, act, act_floats, wts, out, cyc, phases);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, X>), dim3(256), dim3(512), lds, 0, act, act_floats, wts, out, cyc, phases);
    hipEventRecord(e1); hipEventSynchronize(e1);
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    long long med = h[128];
    const double mfma = (MODE == 3 ? 54.0 : (MODE != 1 ? 108.0 : 72.0)) * 8 * phases;         // per CU
    printf("%-64s %8.1f us  %7.0f cycles/phase  %.2f GHz  MFMA pipe busy %.0f %%  (LDS %d KB)\n", name, ms * 1e3, (double)med / phases, med / (ms * 1e6),
           100.0 * mfma / 4 * 32 / med, lds / 1024);
    hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv) {
    const long long act_floats = 64ll << 20;                      // 256 MB of activations (beyond L2; the real frames' tensors come from MALL / HBM)
    float* act; bf16x8* wts;
    hipMalloc(&act, act_floats * 4); hipMalloc(&wts, 8192 * 16 * 2);
    {
        float* h = (float*)malloc(act_floats * 4);
        srand(1);
        for (long long i = 0; i < act_floats; ++i) h[i] = (float)(rand() % 20001 - 10000) * 2e-4f;
        hipMemcpy(act, h, act_floats * 4, hipMemcpyHostToDevice);
        free(h);
        unsigned short* w = (unsigned short*)malloc(8192 * 16 * 2);
        for (int i = 0; i < 8192 * 8 * 2; ++i) {
            const unsigned short sign = (rand() & 1) << 15, ex = (unsigned short)(122 + rand() % 4) << 7, man = rand() & 127;
            w[i] = sign | ex | man;
        }
        hipMemcpy(wts, w, 8192 * 16 * 2, hipMemcpyHostToDevice);
        free(w);
    }
    const int phases = 24 * 8;                                    // = 8 launches' worth of the 6 x 128->64 launch's 3 tiles x 8 phases per workgroup
    if (argc > 4) {          // round 5: F(4,3) along y against F(2,3) and the direct form, with ablations
        for (int round = 0; round < 3; ++round) {
            run<0, 0>("direct 3x3 (108 MFMAs / wave / phase)", act, act_floats, wts, 10, phases);
            run<1, 0>("Winograd F(2,3) along y (72 MFMAs / wave / phase)", act, act_floats, wts, 10, phases);
            run<3, 0>("Winograd F(4,3) along y (54 MFMAs / wave / phase)", act, act_floats, wts, 10, phases);
            run<3, 1>("F(4,3), no transform / split VALU", act, act_floats, wts, 10, phases);
            run<3, 2>("F(4,3), no weight DMA", act, act_floats, wts, 10, phases);
            run<3, 4>("F(4,3), no fragment reads", act, act_floats, wts, 10, phases);
            run<3, 8>("F(4,3), no global loads", act, act_floats, wts, 10, phases);
            run<1, 2>("F(2,3), no weight DMA", act, act_floats, wts, 10, phases);
        }
        return 0;
    }
    if (argc > 3) {          // B-operand reuse (adjacent rows + DPP-shifted horizontal taps)
        for (int round = 0; round < 3; ++round) {
            run<0, 0>("direct (72 KB of fragment reads per wave and phase)", act, act_floats, wts, 10, phases);
            run<2, 0>("direct, B reuse: adjacent rows + DPP taps (44 KB)", act, act_floats, wts, 10, phases);
            run<1, 0>("Winograd F(2,3) along y", act, act_floats, wts, 10, phases);
        }
        return 0;
    }
    if (argc > 2) {          // sensitivity to vector instructions: N extra v_fma per lane and phase beside the direct loop's own
        for (int round = 0; round < 2; ++round) {
            run<0, 0>("direct", act, act_floats, wts, 10, phases);
            run<0, 16>("direct + 36 v_fma per phase (4 per step)", act, act_floats, wts, 10, phases);
            run<0, 32>("direct + 72 v_fma per phase", act, act_floats, wts, 10, phases);
            run<0, 64>("direct + 144 v_fma per phase", act, act_floats, wts, 10, phases);
            run<0, 1>("direct, no split arithmetic (-60)", act, act_floats, wts, 10, phases);
        }
        return 0;
    }
    if (argc > 1) {          // footprint sweep: where the activations are served from (HBM / Infinity Cache / L2)
        for (int round = 0; round < 2; ++round)
            for (long long mb : {256ll, 96ll, 32ll, 8ll}) {
                char nm[96];
                snprintf(nm, sizeof nm, "direct, activations from a %lld MB buffer", mb);
                run<0>(nm, act, mb << 18, wts, 10, mb >= 32 ? phases : 48);
                snprintf(nm, sizeof nm, "Winograd-y, activations from a %lld MB buffer", mb);
                run<1>(nm, act, mb << 18, wts, 10, mb >= 32 ? phases : 48);
            }
        return 0;
    }
    for (int round = 0; round < 2; ++round) {
        run<0>("direct 3x3 (108 MFMAs / wave / phase, 1 barrier)", act, act_floats, wts, 10, phases);
        run<1>("Winograd F(2,3) along y (72 MFMAs / wave / phase, 2 barriers)", act, act_floats, wts, 10, phases);
        run<0, 1>("direct, no split VALU", act, act_floats, wts, 10, phases);
        run<1, 1>("Winograd-y, no transform / split VALU", act, act_floats, wts, 10, phases);
        run<0, 2>("direct, no weight DMA", act, act_floats, wts, 10, phases);
        run<1, 2>("Winograd-y, no weight DMA", act, act_floats, wts, 10, phases);
        run<0, 4>("direct, no fragment reads", act, act_floats, wts, 10, phases);
        run<1, 4>("Winograd-y, no fragment reads", act, act_floats, wts, 10, phases);
        run<0, 8>("direct, no global loads", act, act_floats, wts, 10, phases);
        run<1, 8>("Winograd-y, no global loads", act, act_floats, wts, 10, phases);
        run<1, 11>("Winograd-y, MFMAs + fragment reads + ds_writes only", act, act_floats, wts, 10, phases);
        run<1, 15>("Winograd-y, MFMAs + ds_writes only", act, act_floats, wts, 10, phases);
    }
    return 0;
}
