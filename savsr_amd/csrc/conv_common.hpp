// Launch parameters savsr_conv2d_batch fills from the descriptors, and the global-memory access helpers of the conv kernel
// (conv_mfma.hip; the archived Winograd experiment tools/experiments/conv_wino.hip included this header too).
#pragma once
#include "common.hpp"

namespace savsr {

struct ConvParams {
    const float* src[SAVSR_MAX_SRC];
    int src_pix[SAVSR_MAX_SRC];      // floats between pixels of source s
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
    float* pool;          // optional [tiles][pool_stride]: per-tile channel sums of the stored values
    int pool_stride;
};

#ifndef SAVSR_CONV_MAX_BATCH
#define SAVSR_CONV_MAX_BATCH 24
#endif
constexpr int CONV_MAX_BATCH = SAVSR_CONV_MAX_BATCH;     // convs of identical geometry per launch: 6 (both propagation directions x 3 streams of a block) x up to 4 CLIPS of one
                                       // (shape, scale) batched into the launches (round 5: 3, round 6: 4); 24 x 160 B of descriptors + 36 B = 3 876 B < the 4 KB a kernel argument may have
constexpr int CONV_WIDE_MIN_TILES = 200;       // 16-row tiles are used when a launch has at least this many of them ...
constexpr int CONV_WIDE_MIN_TILES_TP = 100;    // ... or this many in throughput mode (SAVSR_CONV_DIRECT_THROUGHPUT)
#ifndef SAVSR_CONV_BLOCKS
#define SAVSR_CONV_BLOCKS 256
#endif
constexpr int CONV_PERSISTENT_BLOCKS = SAVSR_CONV_BLOCKS;   // one resident workgroup per CU (117-154 KB of LDS each)
#ifndef WY_STRIP
#define WY_STRIP 1                // Winograd-y form: the image's last h % 16 <= 8 rows as strip tiles (conv_wy.hip) where that saves the grid a round of tiles; 0: full tiles only; 2: strips always
#endif
struct MultiConvParams {
    ConvParams c[CONV_MAX_BATCH];     // convs of identical geometry:
    int h, w, cout, nchunk, src_ch;   //   shared shape (fixed kernarg offsets: read once, not per tile)
    int nconv, ncob, ntx, nty;        // tile id = ((conv * ncob + cob) * nty + ty) * ntx + tx
    int wy_tiles, wy_full, wy_strip_l2;   // Winograd-y form: tiles per (conv, cob) = wy_full full tiles (ty * ntx + tx) + strip tiles over the image's last rows, each 2^wy_strip_l2 row pairs x (8 >> wy_strip_l2) segments (conv_wy.hip)
};

int launch_conv_wy(const MultiConvParams& mp, hipStream_t st);      // conv_wy.hip: the Winograd F(2,3)-along-y form (3x3, cout % 64 == 0)
int conv_wy_prepare_device();

// global accesses as (uniform base, 32-bit byte offset): one VGPR per address instead of a 64-bit pair
// (savsr_conv2d validates that every tensor of a launch spans < 2 GiB).  The explicit global address space matters: the
// epilogue's pointers pass through an asm pin, after which hipcc no longer knows their address space and emits FLAT
// loads / stores, which also count on lgkmcnt and so tie every LDS wait of the transpose to the outstanding stores.
#define SAVSR_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ f32x4 ldg4(const float* base, unsigned byte_off) {
    return *(const SAVSR_GLOBAL f32x4*)((const SAVSR_GLOBAL char*)base + byte_off);
}
__device__ __forceinline__ float ldg1(const float* base, unsigned idx) {
    return *((const SAVSR_GLOBAL float*)base + idx);
}
#ifndef CONV_ST
#define CONV_ST 1                 // output stores of the conv epilogue: 0 plain, 1 nt (whole 128-B lines per wave store; solo 6 x 128->64 launch 182 -> 167 us, frame +0.4 %)
#endif
__device__ __forceinline__ void stg4(float* base, unsigned byte_off, const f32x4& v) {
#if CONV_ST == 1
    __builtin_nontemporal_store(v, (SAVSR_GLOBAL f32x4*)((SAVSR_GLOBAL char*)base + byte_off));
#else
    *(SAVSR_GLOBAL f32x4*)((SAVSR_GLOBAL char*)base + byte_off) = v;
#endif
}
// max without the NaN canonicalisation hipcc puts in front of fmaxf (one extra v_max_f32 per call); operands here are
// results of fp32 arithmetic, never signalling NaNs
__device__ __forceinline__ float vmax_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void stg1(float* base, unsigned idx, float v) {
    *((SAVSR_GLOBAL float*)base + idx) = v;
}

}  // namespace savsr
