/*
 * savsr_hip.h -- C ABI of libsavsr_hip.so, the MI355X (gfx950) kernel library behind
 * savsr_amd.archs.savsr_arch.SAVSR.forward().
 *
 * The reference (Weepingchestnut/SAVSR) has no native code on this path: every step of
 * lbasicsr/archs/savsr_arch.py is an un-fused ATen call.  Each entry point below therefore cites
 * the reference Python lines whose arithmetic it replaces, not a reference FFI symbol.
 *
 * Conventions:
 *   - plain C, raw DEVICE pointers + explicit int shapes/strides (in floats) + a hipStream_t
 *     passed as void*; no torch types anywhere in this file;
 *   - every call only ENQUEUES work on the caller's stream: no allocation, no host sync, no
 *     global mutable state => re-entrant across streams/threads and hipGraph-capturable;
 *   - return 0 = ok, <0 = invalid argument (SAVSR_E_*), >0 = hipError_t of the failed launch;
 *     the message is available from savsr_last_error() (thread-local);
 *   - LR feature maps are fp32 channel-last ([h][w][C]); the clip, the SATU output and the result
 *     are channel-planar ([C][h][w]);
 *   - output sizes H, W are computed by the CALLER with Python round() so that get_HW
 *     (savsr_arch.py:745-751) stays bit-exact;
 *   - specialisation: the SATU entry points (savsr_satu_*) and savsr_pack_windows are built for the shipped configuration of the
 *     reference constructor (savsr_arch.py:576-589): num_feat = 64, slid_win = 3, num_in_ch = 3.  The conv / OSConv entry points
 *     take any channel counts that are multiples of 16 (32 for 1x1).  A checkpoint trained with another num_feat does not run.
 */
#ifndef SAVSR_HIP_H
#define SAVSR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAVSR_ABI_VERSION 28

#define SAVSR_E_ARG   (-1)   /* bad shape / null pointer / unsupported combination */
#define SAVSR_E_ALIGN (-2)   /* pointer or stride alignment requirement violated  */

/* activation codes for the conv epilogue */
#define SAVSR_ACT_NONE    0
#define SAVSR_ACT_RELU    1
#define SAVSR_ACT_LRELU   2   /* slope in savsr_conv_desc.slope (0.2 in propagation, savsr_arch.py:426) */
#define SAVSR_ACT_SIGMOID 3

#define SAVSR_MAX_SRC 5

const char* savsr_version(void);
const char* savsr_last_error(void);
int savsr_abi_version(void);
/* One-time, per DEVICE (the current one), not a stream operation: sets the > 64 KiB dynamic-LDS attribute of every kernel of
 * the library, which the launch entry points otherwise do lazily on a kernel's first use.  Optional -- a caller that records
 * the launches into a hipGraph calls it before the capture so that no attribute call falls inside it.  Idempotent, thread-safe. */
int savsr_prepare_device(void);
/* First 16 hex digits of the sha256 over the kernel sources, headers and compile flags this library was built from (build.sh):
 * lets a measurement file name the build it was taken on (profiles/satu_traffic.json; bench.py drops `traffic` when it differs). */
const char* savsr_source_hash(void);
const char* savsr_source_hash_satu(void);     /* the same over the SATU + tail kernel sources only (satu.hip, tail.hip, common.hpp) */
/* Measurement aid (bench.py `clock_mhz`): ONE wave spins through `windows` (1 .. 64) consecutive windows of `window_ticks` ticks of the
 * 100 MHz s_memrealtime counter each and writes out[2 i] = s_memtime delta (shader cycles), out[2 i + 1] = s_memrealtime delta of window i:
 * shader clock = out[2 i] / out[2 i + 1] x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).  Launched on a side stream beside other
 * work it reads the clock of the CU it sits on while that work runs (one wave with s_sleep in its loop; no LDS, co-resides with any
 * workgroup).  window_ticks x windows <= 10^8 (1 s). */
int savsr_clock_probe(int64_t* out, int window_ticks, int windows, void* stream);

/* ------------------------------------------------------------------------------------------
 * Feature-map layout: every LR-resolution feature map is CHANNEL-LAST fp32, [h][w][C], addressed
 * as base + (y*w + x)*pix + c where `pix` (floats between pixels, multiple of 4) may exceed the
 * number of channels used, so a tensor can be read or written as a channel slice of a wider one.
 *
 * Dense conv (3x3 pad 1, or 1x1), stride 1: implicit GEMM on v_mfma_f32_32x32x16_bf16 with
 * split-bf16 ("bf16x3": hi*hi + hi*lo + lo*hi, fp32 accumulate) operands and a fused epilogue.
 * Replaces every nn.Conv2d / F.conv2d on the path:
 *   WindowUnit_l1/l2 convs   savsr_arch.py:429-442,456-462,480-483,488,498
 *   ResidualBlock convs      savsr_arch.py:388-397,402-415  (cat-free: the `torch.cat` inputs
 *                            of :404,:412,:462,:498,:721 are passed as separate sources)
 *   OSConv2d dynamic conv    savsr_arch.py:156-171 (weights produced on device by
 *                            savsr_osconv_weights; gates folded into the weights, :148-149)
 *   OSAdapt mask convs       savsr_arch.py:189-206 (eval BatchNorm folded by the caller)
 *   RCAB / ResidualGroup     savsr_arch.py:541-543,567, conv_last :733, h_win_conv_h :723
 *
 *   y   = act( sum_{src,ci,ky,kx} W[co][ci][ky][kx] * in[ci](y+ky-p, x+kx-p) + bias[co] )
 *   out = y * (mul_px ? mul_px[y*w+x] : 1) + (res1 ? res1[px][co] : 0) + (res2 ? res2_scale*res2[px][co] : 0)
 * Zero padding outside [0,h) x [0,w).  The input channel axis is the concatenation of `nsrc`
 * sources of `src_ch` channels each (src_ch a multiple of 16; of 32 for 1x1).
 * ------------------------------------------------------------------------------------------ */
typedef struct savsr_conv_desc {
    const float* src[SAVSR_MAX_SRC];   /* device pointers (channel offset already applied) */
    int32_t      src_pix[SAVSR_MAX_SRC];   /* floats between pixels of each source */
    int32_t      nsrc;
    int32_t      src_ch;
    int32_t      h, w;
    int32_t      cin;                  /* = nsrc * src_ch */
    int32_t      cout;
    int32_t      ksize;                /* 1 or 3 */
    const void*  wpacked;              /* device; split-bf16 weight image, see savsr_conv_pack_index() */
    const float* bias;                 /* [cout] or NULL */
    int32_t      act;
    float        slope;
    const float* mul_px;               /* [h*w] or NULL                   -- OSAdapt mask, :214 */
    const float* res1; int32_t res1_pix;   /* or NULL */
    const float* res2; int32_t res2_pix;   /* or NULL */
    float        res2_scale;           /* gamma (savsr_arch.py:732)                              */
    float*       out;                  /* channel offset already applied */
    int32_t      out_pix;
    float*       pool;                 /* optional: fused AdaptiveAvgPool2d(1) partials of the stored tensor,
                                          row t (t < savsr_conv_pool_blocks(h, w)) = channel sums of pixel tile t,
                                          written at pool[t * pool_stride + co]; consumers add rows in order */
    int32_t      pool_stride;
    int32_t      algo;                 /* SAVSR_CONV_DIRECT, SAVSR_CONV_DIRECT_THROUGHPUT (tiling only; same results), SAVSR_CONV_WINOGRAD_Y or
                                          SAVSR_CONV_WINOGRAD_Y_THROUGHPUT (tiling only; same results as WINOGRAD_Y) */
} savsr_conv_desc;

#define SAVSR_CONV_DIRECT   0
/* (1 was the Winograd F(2x2, 3x3) experiment of round 2: measured slower, archived under tools/experiments/) */
#define SAVSR_CONV_DIRECT_THROUGHPUT 2   /* the direct kernel, tiled for several launches in flight on different streams: 16-row
                                            tiles from 100 of them up (120 workgroups for a 64 -> 64 conv at 180x320: slower
                                            alone, faster in aggregate -- DESIGN.md 4a); results are bit-identical to DIRECT */

#define SAVSR_CONV_WINOGRAD_Y 3          /* 3x3, cout % 64 == 0: 1-D Winograd F(2,3) along y on the same split-bf16 matrix products (2/3 of the
                                            matrix work; conv_wy.hip).  `wpacked` must then be the Winograd-y image (savsr_conv_wy_pack_index):
                                            U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2 over the tap ROWS g_ky, per kx.
                                            Results differ from DIRECT by rounding only (max-abs error ~1.5 x DIRECT's) */
#define SAVSR_CONV_WINOGRAD_Y_THROUGHPUT 4   /* (ABI 28) WINOGRAD_Y for launches with other streams' launches in flight beside them.  The form's
                                            workgroups walk 16-row tiles; when h % 16 leaves at most 8 rows, those rows can go as STRIP tiles
                                            (their 1 / 2 / 4 row pairs side by side over 8 / 4 / 2 column segments per workgroup instead of
                                            one tile per segment with idle waves: 113 instead of 120 tiles per conv and 64 channels at 180
                                            rows).  WINOGRAD_Y takes strips when they save the persistent grid a round of tiles (what a
                                            launch running alone pays for); _THROUGHPUT also whenever at most 2 row pairs are left (the tile
                                            count decides when another launch fills the tail).  Results are bit-identical to WINOGRAD_Y. */

/* Elements PER PART (hi or lo) of the weight image of a (cout, cin, ksize) conv; the bf16 image
 * holds 2x that many 2-byte elements.  -1 for unsupported shapes. */
int64_t savsr_conv_packed_elems(int cout, int cin, int ksize);
/* Position p of W[co][ci][ky*ksize+kx] inside one part.  The image interleaves parts in groups of
 * 512 elements (one (tap, kstep, 32-row tile) MFMA A operand): bf16 index of the hi value is
 * (p/512)*1024 + p%512, of the lo value (p/512)*1024 + 512 + p%512.  An fp32 kernel bank for
 * savsr_osconv_weights stores W at index p directly.  Unaddressed entries must be zero. */
int64_t savsr_conv_pack_index(int cout, int cin, int ksize, int co, int ci, int tap);
/* The Winograd-y image of a 3x3 conv: elements per part (cout % 64 == 0, cin % 16 == 0, else -1) and the position of
 * U[pos][co][ci][kx] (pos = 0..3) inside one part; hi / lo interleaved in groups of 512 exactly as above. */
int64_t savsr_conv_wy_packed_elems(int cout, int cin);
int64_t savsr_conv_wy_pack_index(int cout, int cin, int co, int ci, int pos, int kx);
int savsr_conv_pool_blocks(int h, int w);
int savsr_conv2d(const savsr_conv_desc* d, void* stream);
/* n (1..savsr_conv2d_max_batch() = 24 since ABI 27; 18 in ABI 26, 6 before) independent convs of identical geometry (ksize, nsrc, src_ch, h, w, cout) in ONE launch:
 * e.g. the per-stream convs of a ResidualBlock (savsr_arch.py:402,413) of both propagation directions -- and, since ABI 26 / 27, of up to three / four
 * clips of one (shape, scale) whose launch sequences the caller runs as one (small clips are launch-latency-bound).  More workgroups than CUs, so workgroups run out of phase and the
 * load/store bursts of one overlap the MFMA phases of another. */
int savsr_conv2d_max_batch(void);
/* (ABI 28) Workgroup tiles a savsr_conv2d_batch launch of `nconv` convs walks in the Winograd-y form with `algo` (SAVSR_CONV_WINOGRAD_Y or
 * _THROUGHPUT): nconv x cout / 64 x (16-row x 32-px tiles, the image's last h % 16 <= 8 rows as strip tiles where the algo's rule takes
 * them).  Host arithmetic only -- the plan the launcher applies; -1 for shapes the form does not take. */
int64_t savsr_conv_wy_tile_count(int h, int w, int cout, int nconv, int algo);
int savsr_conv2d_batch(const savsr_conv_desc* descs, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-channel partial sums for the global average pools (AdaptiveAvgPool2d(1),
 * savsr_arch.py:129,146 for OSConv; :515 for RCAN): partial[blk][s*src_ch + c] over `nblk`
 * pixel ranges; consumers add the partials in block order and scale by 1/(h*w).
 * ------------------------------------------------------------------------------------------ */
int savsr_channel_sums(const float* const* src, const int32_t* src_pix, int nsrc, int src_ch, int64_t npx,
                       int nblk, float* partial, void* stream);

/* ------------------------------------------------------------------------------------------
 * OSConv scale routing + ScaleAttention + kernel aggregation (savsr_arch.py:143-163, 91-96, 69-89):
 *   v  = ReLU(L2 ReLU(L1 [1/sh, 1/sw, mean] + c1) + c2)
 *   a  = ReLU(bn_scale * (Wfc v) + bn_shift)            (eval BatchNorm folded by the caller)
 *   ca = sigmoid(Wc a + bc) (cin), fa = sigmoid(Wf a + bf) (cout), sa = sigmoid(Ws a + bs) (9),
 *   ka = softmax(Wk a + bk) (knum)
 *   W''[co][ci][tap] = fa[co] * ca[ci] * sa[tap] * sum_k ka[k] * W[k][co][ci][tap]
 * written as the split-bf16 weight image savsr_conv2d consumes (three launches, no host sync).
 * ------------------------------------------------------------------------------------------ */
typedef struct savsr_osconv_attn_desc {
    int32_t cin, cout, hidden /* A */, knum /* 8 */;
    float   inv_sh, inv_sw;
    const float* partial; int32_t nblk; float inv_n;   /* pooled input: savsr_channel_sums output, 1/(h*w) */
    const float* l1_w; const float* l1_b;    /* [2cin][cin+2], [2cin] */
    const float* l2_w; const float* l2_b;    /* [cin][2cin],   [cin]  */
    const float* fc_w;                       /* [A][cin] */
    const float* bn_scale; const float* bn_shift; /* [A] */
    const float* ch_w; const float* ch_b;    /* [cin][A],  [cin]  */
    const float* fl_w; const float* fl_b;    /* [cout][A], [cout] */
    const float* sp_w; const float* sp_b;    /* [9][A],    [9]    */
    const float* kn_w; const float* kn_b;    /* [knum][A], [knum] */
    float* v1; float* v2;                    /* scratch [2cin], [cin] */
    const float* bank;                       /* [knum][packed_elems] fp32, index = savsr_conv_pack_index */
    int64_t nunits;                          /* packed_elems / 8 */
    void*  wimg_out;                         /* split-bf16 weight image, 2 * packed_elems * 2 bytes */
    float* att;                              /* optional [cin + cout + 9 + knum] = ca | fa | sa | ka */
    int32_t wy;                              /* != 0: wimg_out receives the Winograd-y image (savsr_conv_wy_pack_index order, 4/3 of the direct size;
                                                cout % 64 == 0) for a conv launched with algo SAVSR_CONV_WINOGRAD_Y: the spatial gate sa[ky, kx] is
                                                applied per tap, then the F(2,3) weight transform over ky */
    int32_t fused;                           /* != 0: ONE launch -- every aggregation workgroup runs the scale routing (pooled mean, layers 1 and 2) itself
                                                instead of two launches in front of it; v2 comes out bit-identical (v1 is not written).  Measured SLOWER
                                                than the three launches on MI355X; the engine leaves it off */
} savsr_osconv_attn_desc;
int savsr_osconv_weights(const savsr_osconv_attn_desc* d, void* stream);
/* n (1..savsr_osconv_weights_max_batch() = 8 since ABI 27; 6 before) independent OSConvs of identical cin / cout / hidden / knum in one set of
 * launches (the two propagation directions of a ResidualBlock pair, savsr_arch.py:399-415, x up to four clips of a batched launch sequence). */
int savsr_osconv_weights_max_batch(void);     /* 8 */
int savsr_osconv_weights_batch(const savsr_osconv_attn_desc* descs, int n, void* stream);

/* RCAN ChannelAttention gate (savsr_arch.py:514-520): gate = sigmoid(W2 ReLU(W1 mean + b1) + b2) */
int savsr_se_gate(const float* partial, int nblk, float inv_n, const float* w1, const float* b1,
                  const float* w2, const float* b2, int c, int cmid, float* gate, void* stream);
/* out[px][c] = r[px][c] * gate[c] + x[px][c]   (savsr_arch.py:524,548-549); contiguous [npx][c] */
int savsr_scale_residual(const float* r, const float* gate, const float* x, float* out, int c, int64_t npx, void* stream);
/* the two above in ONE launch (what SAVSR.forward runs, 32 x per frame): every workgroup re-evaluates the gate from the pooled
 * partial sums, then out = r * gate + x.  Bit-identical to savsr_se_gate followed by savsr_scale_residual. */
int savsr_se_scale_residual(const float* partial, int nblk, float inv_n, const float* w1, const float* b1,
                            const float* w2, const float* b2, int c, int cmid,
                            const float* r, const float* x, float* out, int64_t npx, void* stream);
/* The same for `nclip` clips of a batched launch sequence in ONE launch (ABI 26): clip b's partial / r / x / out lie b * the given byte strides
 * (multiples of 16) behind clip 0's; per clip bit-identical to savsr_se_scale_residual. */
int savsr_se_scale_residual_batch(const float* partial, int nblk, float inv_n, const float* w1, const float* b1,
                                  const float* w2, const float* b2, int c, int cmid, const float* r, const float* x, float* out,
                                  int64_t npx, int nclip, int64_t partial_stride, int64_t r_stride, int64_t x_stride, int64_t out_stride, void* stream);

/* nn.AvgPool2d(2) (savsr_arch.py:193): [h][w][c] -> [h/2][w/2][c], h and w even, contiguous. */
int savsr_avgpool2(const float* in, float* out, int c, int h, int w, void* stream);
/* nn.Upsample(scale_factor=2, bilinear, align_corners=False) (savsr_arch.py:202), channel-last. */
int savsr_upsample2x(const float* in, float* out, int c, int h, int w, void* stream);
/* WindowUnit_l1 input windows (savsr_arch.py:448-454, :661-668) with SAVSR.pad_spatial's reflect
 * padding (:670-690) folded in.  lq: [T][3][h][w] planar -> out: [T-2][hp][wp][16] channel-last,
 * channels = frame t | frame t-1 | frame t+1 | 7 zeros for window centre t = position + 1. */
int savsr_pack_windows(const float* lq, float* out, int T, int h, int w, int hp, int wp, void* stream);

/* ------------------------------------------------------------------------------------------
 * SATU = STAUpsample.forward (savsr_arch.py:315-376), restructured:
 *   out = G(Wa sta, soff) + G(Wb x, off) + sum_n r_n (Wb E_n) (sum_m r_m C_m G(x, off)) + b
 * in three launches:
 *   savsr_satu_phase_table : coordinate MLP (:344-350) on the DISTINCT (coor_h, coor_w) values
 *   savsr_satu_expand_table: that table per HR pixel (both: once per size / scale / weights, not per frame)
 *   savsr_satu_lr_stage    : kernel_conv + LeakyReLU(0.1) + sta_conv (:226-228,297-313,319-320)
 *                            and the three LR-side projections -> LRcat [h][w][160]
 *   savsr_satu_hr_upsample : bilinear gathers (:262-295), expert mixing (:353-370), fusion (:374)
 * ------------------------------------------------------------------------------------------ */
#define SAVSR_SATU_C      64
#define SAVSR_SATU_LRCAT  160
#define SAVSR_SATU_TABLE  8    /* r0 r1 r2 r3 off_x off_y soff_x soff_y */
#define SAVSR_SATU_LRCAT_TAIL 96 /* LRcat record of the tail-projected form (savsr_satu_*_tail) */
#define SAVSR_TAIL_PLANES 27     /* 9 taps x 3 colours */

typedef struct savsr_satu_weights {      /* all device pointers; packed by the caller */
    const float* body0_w; const float* body0_b;   /* [64][4], [64]   savsr_arch.py:245 */
    const float* body2_w; const float* body2_b;   /* [64 in][64 out] (transposed), [64]  :247 */
    const float* head_w;  const float* head_b;    /* [8][64], [8]: routing(4) | offset(2) | st_offset(2)  :252,256,257 */
    const void*  kconv_w; const float* kconv_b;   /* split-bf16 image [25][2][4 ks][part][64 lanes][8], fp32 [25][64]  :227 */
    const void*  proj_w;                          /* split-bf16 image of the LR projections (Wa | Wb | C-stack) */
    const void*  wbe_w;                           /* split-bf16 image of (Wb E_n): [2 t][2 ks][part][64 lanes][8] */
    const float* fusion_b;                        /* [64] :260 */
} savsr_satu_weights;

/* table[uh][uw][8] for uh < n_uh, uw < n_uw.  uniq_ch/uniq_cw are the distinct fp32 values of
 * coor_h/coor_w (savsr_arch.py:331-333) computed by the caller; inv_sw, inv_sh are 1/scale. */
int savsr_satu_phase_table(const savsr_satu_weights* wt, const float* uniq_ch, int n_uh,
                           const float* uniq_cw, int n_uw, float inv_sw, float inv_sh,
                           float* table, void* stream);

/* x, st: channel-last crops [h][w][64] of padded tensors (savsr_arch.py:737): element (y, x, c) at
 * base + (y*row_px + x)*pix + c. */
int savsr_satu_lr_stage(const savsr_satu_weights* wt, const float* x, const float* st,
                        int32_t pix, int32_t row_px, int h, int w, float* lrcat, void* stream);

/* LDS staging plan of the HR stage (a pure performance hint; results never depend on it).  The HR kernel is persistent:
 * its workgroups walk tiles of tile_rows x (32 * tile_cols32) HR pixels and, for each, stage an lr_rows x lr_cols window of
 * LRcat records whose origin is the tile's base sampling coordinate + (off_min_x, off_min_y), double-buffered (the next tile's
 * window arrives by LDS-DMA while the current tile is computed).  Waves whose taps leave the window gather from global
 * memory instead.  NULL = 8 x 32 tiles without a window.  tile_rows must be a multiple of 4.  The LDS need is
 * savsr_satu_hr_lds_bytes(form, n_uh * n_uw, tile_rows, tile_cols32, lr_rows, lr_cols) <= 160 KiB (one workgroup per CU). */
typedef struct savsr_satu_tiling {
    int32_t tile_rows, tile_cols32, lr_rows, lr_cols;
    float   off_min_x, off_min_y;
    int32_t table_entries;   /* n_uh * n_uw; informational */
    float   step_x, step_y;  /* LR pixels per HR pixel (1 / scale_w, 1 / scale_h) for the window origin; <= 0: w / W, h / H */
    int32_t variant;         /* wave split of the workgroup, 0 .. savsr_satu_hr_variants() - 1 (0: 8 compute + 4 producer waves; 1, tail
                                form only: 10 + 6).  Which one is faster depends on size and scale; the caller may time both. */
} savsr_satu_tiling;
int savsr_satu_hr_variants(void);
int64_t savsr_satu_hr_lds_bytes(int tail_form, int n_table, int tile_rows, int tile_cols32, int lr_rows, int lr_cols);
/* resident workgroups per CU the HR kernel is written for (plan tiles so that this many fit 160 KiB of LDS) */
int savsr_satu_hr_occupancy_target(int tail_form);
/* compute waves of an HR workgroup: a tile of tile_rows x tile_cols32 "wave tiles" (one row x 32 pixels) is dealt over them */
int savsr_satu_hr_compute_waves(int variant);
int savsr_satu_hr_rows_per_wave_tile(int tail_form);   /* HR rows one wave tile covers (32 pixels wide) */

/* Per-pixel expansion of the phase table, once per (size, scale, weights): ptab[Y][X][8] = table[idx_h[Y]][idx_w[X]] with the
 * two offset pairs normalised as the reference normalises them per pixel ((off * 2) / (size - 1), savsr_arch.py:285-287).
 * Needed by the HR stage only for tables of more than 256 entries (smaller ones are kept whole in LDS). */
int savsr_satu_expand_table(const float* table, int n_uw, const int32_t* idx_h, const int32_t* idx_w,
                            int h, int w, int H, int W, float* ptab, void* stream);

/* table[n_uh][n_uw][8]: savsr_satu_phase_table's output; idx_h[H], idx_w[W]: index of each row's / column's (coor_h, coor_w)
 * value in it; gxn[W], gyn[H]: normalised base grid coordinates (savsr_arch.py:270-280) computed by the caller in fp32.
 * These four arrays are read in 16-byte groups: 16-byte aligned and READABLE up to the next multiple of 4 elements.
 * ptab: savsr_satu_expand_table's output (may be NULL when n_uh * n_uw <= 256).
 * sched: 16 int32 of device scratch for the kernel's tile queue, ZERO-FILLED ONCE by the caller (the kernel leaves them zero) and
 * not shared by launches that may run concurrently (one per stream); NULL = static tile walk (no scratch, ~10 % slower).
 * out: [64] planes of [H][W], `out_plane` floats apart (>= H*W; a pitch that is not a multiple of a few KiB keeps the
 * 64 planes of one pixel on different HBM channels). */
int savsr_satu_hr_upsample(const savsr_satu_weights* wt, const float* lrcat, int h, int w,
                           const float* table, int n_uh, int n_uw, const int32_t* idx_h, const int32_t* idx_w, const float* ptab,
                           const float* gyn, const float* gxn, int H, int W,
                           const savsr_satu_tiling* tiling, int32_t* sched, float* out, int64_t out_plane, void* stream);

/* ------------------------------------------------------------------------------------------
 * Tail-projected form of SATU + tail = savsr_arch.py:315-376 followed by :738-739, the form SAVSR.forward runs.
 * The tail conv is linear and a bilinear gather commutes with a channel contraction, so the tail's weights, regrouped as
 * Wt27[p = 3 (3 ky + kx) + o][c] (27 rows, padded to 32), are multiplied into every matrix of the stage by the caller:
 *   wt->proj_w   = LR projections (Wt27 Wa | Wt27 Wb | C-stack), wt->wbe_w = (Wt27 Wb E_n), wt->fusion_b = Wt27 b
 *   P[p] = G(Wt27 Wa sta, soff) + G(Wt27 Wb x, off) + sum_n r_n (Wt27 Wb E_n)(sum_m r_m C_m G(x, off)) + Wt27 b
 * savsr_satu_lr_stage_tail -> LRcat [h][w][96]; savsr_satu_hr_tail -> P: [27] planes of [H][W]; savsr_tail_gather adds
 * the nine shifted taps per colour, the tail bias and the bilinear residual of the unpadded centre frame.
 * The [64][H][W] SATU output never exists (236 MB written + re-read per 720x1280 frame in the two-kernel form).
 * Same argument meaning as the standalone entry points above. */
int savsr_satu_lr_stage_tail(const savsr_satu_weights* wt, const float* x, const float* st,
                             int32_t pix, int32_t row_px, int h, int w, float* lrcat, void* stream);
int savsr_satu_hr_tail(const savsr_satu_weights* wt, const float* lrcat, int h, int w,
                       const float* table, int n_uh, int n_uw, const int32_t* idx_h, const int32_t* idx_w, const float* ptab,
                       const float* gyn, const float* gxn, int H, int W,
                       const savsr_satu_tiling* tiling, int32_t* sched, float* out, int64_t out_plane, void* stream);
/* p27: [27] planes of [H][W], p_plane floats apart; center: [3][h][w]; out: [3][H][W] contiguous. */
int savsr_tail_gather(const float* p27, int64_t p_plane, const float* tail_b, const float* center,
                      int h, int w, int H, int W, float* out, void* stream);
/* The ROW-SUMMED tail form.  Same stage, with the rows of Wt27 in the order savsr_satu_hr_tail_q wants them -- the three horizontal
 * taps kx of group g = 3 ky + o at MFMA rows acc_row(3 gi + kx, half), gi = g (half 0) for g < 5, g - 5 (half 1) otherwise, where
 * acc_row(r, half) = 8 (r / 4) + 4 half + r % 4 -- in every tail-form matrix (LR stage included: savsr_satu_lr_stage_tail with those
 * weights).  The HR stage adds the three horizontal taps itself and writes q9: [9] planes Q[g] of [H][W] (q_plane floats apart) plus
 * seam: [H][ceil(W / 32)][2][9] floats (the terms that cross a 32-pixel segment border); savsr_tail_gather_q adds the vertical
 * taps, the seams, tail_b and the bilinear residual.  Results equal the 27-plane form's up to the summation order of the nine taps. */
int savsr_satu_hr_tail_q(const savsr_satu_weights* wt, const float* lrcat, int h, int w,
                         const float* table, int n_uh, int n_uw, const int32_t* idx_h, const int32_t* idx_w, const float* ptab,
                         const float* gyn, const float* gxn, int H, int W, const savsr_satu_tiling* tiling, int32_t* sched,
                         float* q9, int64_t q_plane, float* seam, int64_t seam_floats, void* stream);
int savsr_tail_gather_q(const float* q9, int64_t q_plane, const float* seam, int64_t seam_floats, const float* tail_b, const float* center,
                        int h, int w, int H, int W, float* out, void* stream);

/* tail conv 3x3 64->3 + bias at HR plus the bilinear residual of the (unpadded) centre frame
 * (savsr_arch.py:738-739).  feat: [64] planes of [H][W], feat_plane floats apart; center: [3][h][w];
 * out: [3][H][W] contiguous. */
int savsr_tail_residual(const float* feat, int64_t feat_plane, const float* tail_w /* [3][64][3][3] */, const float* tail_b,
                        const float* center, int h, int w, int H, int W, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * PSNR-Y / SSIM-Y of one output frame with the reference's numerics (SURVEY section 8, row f3 -- the step after the
 * hot path): tensor2img quantisation (lbasicsr/utils/img_util.py:66-90), BT.601 luma (metrics/metric_util.py:32-45,
 * utils/color_util.py:59-65), calculate_psnr / calculate_ssim (metrics/psnr_ssim.py:42-48, 172-200).
 * sr, gt: RGB planar fp32 [3] planes of [H][W] (*_plane floats apart), values in [0,1] before clamping.
 * partial: scratch of savsr_metrics_blocks(H, W, crop_border) * 2 doubles; out: 2 doubles = PSNR-Y (inf for
 * identical images), SSIM-Y.  Both stay on the device; the calls only enqueue. */
int savsr_metrics_blocks(int H, int W, int crop_border);       /* < 0: the cropped image is smaller than 11 x 11 */
int savsr_metrics_psnr_ssim_y(const float* sr, int64_t sr_plane, const float* gt, int64_t gt_plane, int H, int W,
                              int crop_border, double* partial, double* out, void* stream);
/* The same with `test_y_channel` as the YAML's metric option (psnr_ssim.py:12,85): != 0 is savsr_metrics_psnr_ssim_y; 0 takes the
 * three colour planes of the quantised image -- PSNR from the mean squared difference over H x W x 3, SSIM as the mean of the
 * planes' SSIM (psnr_ssim.py:115-129) -- and needs partial[3 * savsr_metrics_blocks(..) * 2]. */
int savsr_metrics_psnr_ssim(const float* sr, int64_t sr_plane, const float* gt, int64_t gt_plane, int H, int W, int crop_border,
                            int test_y_channel, double* partial, double* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * One axis of the anti-aliased bicubic resize behind the reference's LR synthesis (SURVEY section 8, row f2 -- the step
 * before the hot path): lbasicsr/data/data_util.py:371-420 -> torchvision T.Resize(BICUBIC, antialias=True) -> ATen
 * _upsample_bicubic2d_aa.  in: [planes][h][w] fp32; axis 0 resizes the width (out [planes][h][out_size]), axis 1 the
 * height (out [planes][out_size][w]).  xmin / xsize / weights[out_size][max_taps]: the per-output-index windows and
 * normalised weights as ATen computes them (savsr_amd/resize_gpu.py::aa_tables).  Width first, then height. */
int savsr_resize_aa_axis(const float* in, int planes, int h, int w, int axis, int out_size, const int32_t* xmin,
                         const int32_t* xsize, const float* weights, int max_taps, float* out, void* stream);

/* ---- diagnostics: compiled ONLY into the instrumented library (SAVSR_DIAG=1 savsr_amd/csrc/build.sh ->
 * libsavsr_hip_diag.so, loaded by the tools through SAVSR_LIB_PATH).  The product library libsavsr_hip.so carries neither
 * these entry points nor the DIAG kernel instantiations nor any process-global switch: every product call is a pure
 * function of its arguments.  Synchronous; while a stamps mode is on, the conv / SATU launches run INSTRUMENTED builds of their kernels (template parameter
 * DIAG); with the mode off (the default) the product kernels carry no diagnostic code at all.
 * Conv kernel, savsr_debug_conv_stamps(mode): 0 off; 1 per-workgroup s_memtime stamps [blk][6] = entry, after
 * the prologue, after the first K phase, after the first tile's K loop, after the stores drained,
 * s_memrealtime at entry; 3 + w: accumulated section times of wave w ([blk][0..4] = steps after the barrier,
 * steps before it, wait, barrier, epilogue); + 16 / + 32 / + 64 / + 128 / + 256 / + 512: timing experiments that skip
 * the staging / the fragment reads / the epilogue's stores / its LDS transpose / its whole body / its bias load
 * (results invalid). */
#ifdef SAVSR_DIAG
int savsr_debug_conv_stamps(int enable);
int savsr_debug_read_conv_stamps(long long* host, int nblocks);
/* SATU LR / HR kernels: [blk][8] accumulated section times of wave 0 (see satu.hip), last = total. */
int savsr_debug_satu_stamps(int enable);
int savsr_debug_read_satu_stamps(long long* host, int nblocks);
/* Resident workgroups per CU predicted by the runtime for the SATU HR (which = 0) / LR (1) kernel with
 * lds_bytes of dynamic LDS; < 0 = -hipError_t. */
int savsr_debug_satu_occupancy(int which, int lds_bytes);
#endif /* SAVSR_DIAG */

#ifdef __cplusplus
}
#endif
#endif /* SAVSR_HIP_H */
