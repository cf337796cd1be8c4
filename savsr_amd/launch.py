"""Launch helpers of the engine: descriptors and C-ABI calls for the convs, the OSConv weight generation and the SATU stages.

`Launcher` is the part of `HipEngine` that turns (weight key, source slices, output slice) into `savsr_conv_desc` /
`savsr_osconv_attn_desc` structures and issues `savsr_conv2d_batch` / `savsr_osconv_weights_batch` / `savsr_satu_*` on the engine's
stream -- including the expansion of a descriptor over the clips of a batched launch sequence, the per-launch choice of the conv
form (direct / Winograd-y) and the SATU axis tables + HR launch plan of a (size, scale).  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ConvDesc, OSConvAttnDesc, SatuTiling
from ._xfer import h2d
from .packing import get_hw, satu_axis_tables

MAX_SUM_BLOCKS = 256     # workgroups of one savsr_channel_sums launch


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class Src:
    """A channel-last feature map slice: element (px, c) at ptr + 4*(px*pix + c), c < ch.  bs = bytes from one clip's copy of the tensor to the
    next when several clips of one (shape, scale) share the launches (HipEngine.nb > 1), 0 otherwise."""
    __slots__ = ("t", "ptr", "ch", "pix", "bs")

    def __init__(self, t: torch.Tensor, ch: int, pix: int, ch_off: int = 0, float_off: int = 0, bs: int = 0):
        self.t = t
        self.ptr = t.data_ptr() + 4 * (ch_off + float_off)
        self.ch, self.pix, self.bs = ch, pix, bs


class Launcher:
    """Mixin of HipEngine: descriptors, batched launches, SATU tables and launch plans."""

    def _stream(self) -> int:
        """Raw handle of the current HIP stream.  Inside a frame's stage functions it is looked up once (`_stage_stream`): ~330 launches per
        frame asked torch for it ~190 times each 9 us -- a fifth of an eager or capturing frame's host time."""
        st = self._st
        return st if st is not None else torch.cuda.current_stream().cuda_stream

    class _StageStream:
        def __init__(self, eng):
            self.eng = eng

        def __enter__(self):
            self.prev = self.eng._st
            self.eng._st = torch.cuda.current_stream().cuda_stream

        def __exit__(self, *a):
            self.eng._st = self.prev

    def full(self, t: torch.Tensor, ch: Optional[int] = None, ch_off: int = 0) -> Src:
        """Channel slice [ch_off, ch_off+ch) of a contiguous channel-last tensor [h][w][C]."""
        c_total = t.shape[-1]
        return Src(t, c_total - ch_off if ch is None else ch, c_total, ch_off, bs=self._bs(t))

    def _bs(self, t: Optional[torch.Tensor]) -> int:
        """Bytes between the clips' copies of a named buffer (0: one clip, or a tensor every clip shares)."""
        return 0 if (t is None or self.nb == 1) else self._bstride.get(t.data_ptr(), 0)

    def conv_desc(self, key, srcs: List[Src], out: Src, h: int, w: int, act=ACT_NONE, slope=0.0,
                  mul_px=None, res1: Optional[Src] = None, res2: Optional[Src] = None, res2_scale=0.0, weights=None,
                  pool: Optional[Tuple[torch.Tensor, int, int]] = None) -> ConvDesc:
        """pool = (partial tensor, column offset, row stride): fused global-average-pool partials of the output."""
        wpk, bias, cout, cin, ks, *rest = weights if weights is not None else self.pw[key]
        d = ConvDesc()
        d.algo = rest[0] if rest else self.conv_algo
        d._wy = self.pw_wy[key].data_ptr() if (weights is None and key in self.pw_wy) else None      # (a Python attribute, not a field of the C struct)
        assert len(srcs) <= _lib.MAX_SRC and all(s.ch == srcs[0].ch for s in srcs)
        assert cin == len(srcs) * srcs[0].ch, (key, cin, len(srcs), srcs[0].ch)
        assert out.ch == cout, (key, out.ch, cout)
        for i, s in enumerate(srcs):
            d.src[i] = s.ptr
            d.src_pix[i] = s.pix
        d.nsrc, d.src_ch, d.h, d.w, d.cin, d.cout, d.ksize = len(srcs), srcs[0].ch, h, w, cin, cout, ks
        d.wpacked, d.bias, d.act, d.slope = wpk.data_ptr(), _ptr(bias), act, slope
        d.mul_px = _ptr(mul_px)
        if res1 is not None:
            d.res1, d.res1_pix = res1.ptr, res1.pix
        if res2 is not None:
            d.res2, d.res2_pix = res2.ptr, res2.pix
        d.res2_scale = res2_scale
        d.out, d.out_pix = out.ptr, out.pix
        if pool is not None:
            d.pool, d.pool_stride = pool[0].data_ptr() + 4 * pool[1], pool[2]
        if self.nb > 1:      # bytes from clip b's operand to clip b + 1's (Python attribute): sources, out, res1, res2, mul_px, pool, weights
            d._bs = ([s.bs for s in srcs], out.bs, res1.bs if res1 is not None else 0, res2.bs if res2 is not None else 0, self._bs(mul_px),
                     self._bs(pool[0]) if pool is not None else 0, self._bs(wpk) if weights is not None else 0)
            assert out.bs > 0, (key, "a batched launch writes one output per clip")
        return d

    @staticmethod
    def _clip_desc(d: ConvDesc, b: int) -> ConvDesc:
        """Descriptor of the same conv for clip b of a batched launch sequence: every per-clip pointer moved on by b clip strides."""
        if b == 0:
            return d
        n = ConvDesc.from_buffer_copy(d)
        src_bs, out_bs, r1_bs, r2_bs, mp_bs, pool_bs, w_bs = d._bs
        for i in range(d.nsrc):
            n.src[i] = d.src[i] + b * src_bs[i]
        n.out = d.out + b * out_bs
        if d.res1:
            n.res1 = d.res1 + b * r1_bs
        if d.res2:
            n.res2 = d.res2 + b * r2_bs
        if d.mul_px:
            n.mul_px = d.mul_px + b * mp_bs
        if d.pool:
            n.pool = d.pool + b * pool_bs
        n.wpacked = d.wpacked + b * w_bs
        n._wy = getattr(d, "_wy", None)
        return n

    def _wy_algo(self) -> int:
        """The Winograd-y form's algo value of the current flow (tiling of the image's last rows only: same bits)."""
        return _lib.CONV_WINOGRAD_Y_THROUGHPUT if self.conv_algo == _lib.CONV_DIRECT_THROUGHPUT else _lib.CONV_WINOGRAD_Y

    def conv_launch(self, descs: List[ConvDesc], label: str = "conv"):
        """Independent convs of identical geometry, up to 6 per launch and clip (savsr_conv2d_batch; x nb clips of a batched launch sequence)."""
        st = self._stream()
        for i in range(0, len(descs), 6):
            chunk = descs[i:i + 6]
            per_clip = len(chunk)
            if self.nb > 1:
                chunk = [self._clip_desc(d, b) for b in range(self.nb) for d in chunk]
            # Winograd-y form when every conv of the launch has the image and its 16-row x 32-px x 64-channel tiles fill the chip: measured on
            # 180x320 (tools/ab_conv.py --wy): 6 x 128->64 -10 %, 6 x 64->64 -7..-9 %; a lone 64->64 conv (120 tiles) +35 % against the 8-row
            # direct tiling, -4 % against the 16-row direct tiling of the throughput mode
            d0 = chunk[0]
            if all(getattr(c, "_wy", None) for c in chunk) and d0.algo in (_lib.CONV_DIRECT, _lib.CONV_DIRECT_THROUGHPUT):
                # (the count is the launch's when `form_nb` clips share it -- a property of the frame's shape and flow, NOT of how many clips
                # happen to be batched: a clip's result never depends on the clips it was grouped with)
                tiles = per_clip * self.form_nb * (d0.cout // 64) * ((d0.h + 15) // 16) * ((d0.w + 31) // 32)
                if tiles >= (self.wy_min_tiles_tp if d0.algo == _lib.CONV_DIRECT_THROUGHPUT else self.wy_min_tiles):
                    for c in chunk:
                        c.wpacked, c.algo = c._wy, self._wy_algo()
            if self.census is not None:      # diagnostics (bench.py): matrix work of this launch, by the form it takes
                self._count_conv(chunk)
            arr = (ConvDesc * len(chunk))(*chunk)
            _lib.check(self.lib.savsr_conv2d_batch(arr, len(chunk), st), f"savsr_conv2d_batch[{label}]")

    def _count_conv(self, chunk) -> None:
        """Census of one conv launch for bench.py's matrix-utilisation figures.  `alg` = 2 x MACs of the convs as the reference states them;
        `issued` = flops of the bf16 MFMAs the launch really executes: 3 split products per MAC in the direct form, 2 in the Winograd F(2,3)-y
        form (12 taps for two output rows instead of 18), on the padded tile grid -- 32-pixel column blocks, row PAIRS (waves whose rows lie
        below the image run an MFMA-free body), output channels in blocks of 32 / 64.  For 6 x 128->64 at 180x320 this gives 3.110 M
        instructions, the count the PMC pass reads (profiles/r04_conv_wy_pmc_summary.csv)."""
        mode = "tp" if self.conv_algo == _lib.CONV_DIRECT_THROUGHPUT else "b1"
        c = self.census
        for d in chunk:
            taps = d.ksize * d.ksize
            alg = 2.0 * d.h * d.w * d.cin * d.cout * taps
            cot = 64 if d.cout > 32 else 32
            px = (2 * ((d.h + 1) // 2) if d.ksize == 3 else d.h) * (32 * ((d.w + 31) // 32))
            wy = int(d.algo) in _lib.CONV_WY_FORMS
            issued = 2.0 * px * d.cin * (cot * ((d.cout + cot - 1) // cot)) * taps * (2.0 if wy else 3.0)
            for k, v in (("alg_" + mode, alg), ("issued_" + mode, issued), ("direct_eq_" + mode, 3.0 * alg), ("wy_alg_" + mode, alg if wy else 0.0)):
                c[k] = c.get(k, 0.0) + v

    def conv(self, key, srcs: List[Src], out: Src, h: int, w: int, act=ACT_NONE, slope=0.0,
             mul_px=None, res1: Optional[Src] = None, res2: Optional[Src] = None, res2_scale=0.0, weights=None, pool=None):
        self.conv_launch([self.conv_desc(key, srcs, out, h, w, act, slope, mul_px, res1, res2, res2_scale, weights, pool)], key)
        return out

    def channel_sums(self, srcs: List[Src], npx: int, partial: torch.Tensor) -> int:
        n = len(srcs)
        nblk = max(1, min(MAX_SUM_BLOCKS, npx // 128))
        pix = (C.c_int32 * n)(*[s.pix for s in srcs])
        for b in range(self.nb):
            ptrs = (_lib.fptr * n)(*[s.ptr + b * s.bs for s in srcs])
            _lib.check(self.lib.savsr_channel_sums(ptrs, pix, n, srcs[0].ch, npx, nblk, partial.data_ptr() + b * self._bs(partial), self._stream()),
                       "savsr_channel_sums")
        return nblk

    def pool_rows(self, h: int, w: int) -> int:
        return int(self.lib.savsr_conv_pool_blocks(h, w))

    def pool_buf(self, key: str, h: int, w: int, cin: int) -> torch.Tensor:
        """Partial-sum rows for a pooled tensor: one row per conv pixel tile (fused pooling) or per
        savsr_channel_sums workgroup."""
        return self.buf("pool." + key, max(self.pool_rows(h, w), MAX_SUM_BLOCKS) * cin)

    def osconv_wy(self, n_convs: int, cout: int, h: int, w: int) -> bool:
        """Whether the dynamic convs of a launch of `n_convs` OSConvs run in the Winograd-y form (the rule of conv_launch)."""
        if not self.conv_wy or cout % 64:
            return False
        tiles = n_convs * self.form_nb * (cout // 64) * ((h + 15) // 16) * ((w + 31) // 32)
        return tiles >= (self.wy_min_tiles_tp if self.conv_algo == _lib.CONV_DIRECT_THROUGHPUT else self.wy_min_tiles)

    def osconv_desc(self, key: str, srcs: List[Src], h: int, w: int, scale, pooled: bool = False, wy: bool = False) -> OSConvAttnDesc:
        """Descriptor of one OSConv's weight generation (pool -> routing/attention -> aggregated split-bf16 image,
        savsr_arch.py:143-163).  pooled=True: the producing convs already wrote the pool partials (fused epilogue);
        otherwise the pooling kernel is launched here."""
        e = self.osc[key]
        partial = self.pool_buf(key, h, w, e["cin"])
        nblk = self.pool_rows(h, w) if pooled else self.channel_sums(srcs, h * w, partial)
        d = OSConvAttnDesc()
        d.cin, d.cout, d.hidden, d.knum = e["cin"], e["cout"], e["hidden"], e["knum"]
        d.inv_sh, d.inv_sw = 1.0 / scale[0], 1.0 / scale[1]
        d.nblk, d.inv_n, d.nunits = nblk, 1.0 / (h * w), e["nunits"]
        d.partial = partial.data_ptr()
        for k in ("l1_w", "l1_b", "l2_w", "l2_b", "fc_w", "bn_scale", "bn_shift", "ch_w", "ch_b", "fl_w", "fl_b",
                  "sp_w", "sp_b", "kn_w", "kn_b", "v1", "v2", "bank", "att"):
            setattr(d, k, e[k].data_ptr())
        d.wy = 1 if wy else 0
        d.fused = 1 if self.osconv_fused else 0
        wimg = e["wdyn_wy"] if wy else e["wdyn"]
        d.wimg_out = wimg.data_ptr()
        if self.nb > 1:      # per-clip operands of a batched launch sequence: pool partials, routing vectors, gates, the generated image
            d._bs = {"partial": self._bs(partial), "v1": self._bs(e["v1"]), "v2": self._bs(e["v2"]), "att": self._bs(e["att"]), "wimg_out": self._bs(wimg)}
            assert all(v > 0 for v in d._bs.values()), (key, d._bs)
        return d

    def osconv_launch(self, keys: List[str], descs: List[OSConvAttnDesc]):
        """Weight generation of independent OSConvs of identical geometry, up to 6 per set of launches
        (savsr_osconv_weights_batch); returns the conv `weights` tuples."""
        st = self._stream()
        per = max(1, int(self.lib.savsr_osconv_weights_max_batch()) // self.nb)      # OSConvs per set of launches when every one of them goes out once per clip
        for i in range(0, len(descs), per):
            chunk = descs[i:i + per]
            if self.nb > 1:
                clips = []
                for b in range(self.nb):
                    for d in chunk:
                        n = d if b == 0 else OSConvAttnDesc.from_buffer_copy(d)
                        if b:
                            for f, bs in d._bs.items():
                                setattr(n, f, getattr(d, f) + b * bs)
                        clips.append(n)
                chunk = clips
            arr = (OSConvAttnDesc * len(chunk))(*chunk)
            _lib.check(self.lib.savsr_osconv_weights_batch(arr, len(chunk), st), f"savsr_osconv_weights_batch[{keys[i]}]")
        return [(self.osc[k]["wdyn_wy"], None, self.osc[k]["cout"], self.osc[k]["cin"], 3, self._wy_algo()) if dsc.wy else
                (self.osc[k]["wdyn"], None, self.osc[k]["cout"], self.osc[k]["cin"], 3) for k, dsc in zip(keys, descs)]

    def osconv_weights(self, key: str, srcs: List[Src], h: int, w: int, scale, pooled: bool = False, wy: bool = False):
        return self.osconv_launch([key], [self.osconv_desc(key, srcs, h, w, scale, pooled, wy)])[0]

    # ------------------------------------------------------------------ SATU
    def satu_axes(self, h: int, w: int, scale):
        key = (h, w, float(scale[0]), float(scale[1]))
        ent = self._axes.get(key)
        if ent is None:
            H, W = get_hw(h, w, scale)
            ch, _, gyn = satu_axis_tables(H, h, scale[0])
            cw, _, gxn = satu_axis_tables(W, w, scale[1])
            uh, ih = np.unique(ch, return_inverse=True)
            uw, iw = np.unique(cw, return_inverse=True)
            def up(a, dt):          # device copy padded to a multiple of 4 elements (the HR stage reads these arrays in 16-byte groups)
                a = np.ascontiguousarray(a.astype(dt)).reshape(-1)
                pad = (-len(a)) % 4
                return h2d(torch.from_numpy(np.concatenate([a, np.repeat(a[-1:], pad)]) if pad else a), self.dev)
            ent = dict(H=H, W=W, n_uh=len(uh), n_uw=len(uw), uh=up(uh, np.float32), uw=up(uw, np.float32),
                       ih=up(ih.reshape(-1), np.int32), iw=up(iw.reshape(-1), np.int32), gyn=up(gyn, np.float32), gxn=up(gxn, np.float32))
            self._plan_hr_tiling(ent, h, w, scale)
            # The tables are a function of (size, scale, weights): ONE set for the engines of all streams (the dict is shared with the
            # siblings: three streams used to build every set three times, 11 ms of host work each).  Another stream's first use waits for
            # the event below (everything that filled the tables is ordered before it on this engine's stream).
            ent["ready"] = torch.cuda.Event()
            ent["ready"].record(torch.cuda.current_stream())
            ent["seen"] = {id(self)}
            self._axes[key] = ent
            while len(self._axes) > min(64, max(self.max_shapes, self.max_scales)):      # (live graphs hold their own reference: _forward_graphed)
                self._axes.popitem(last=False)
        else:
            self._axes.move_to_end(key)
            if id(self) not in ent["seen"] and not torch.cuda.is_current_stream_capturing():
                cs = torch.cuda.current_stream()
                cs.wait_event(ent["ready"])
                for v in ent.values():               # (allocator bookkeeping: this stream reads the tables too)
                    if isinstance(v, torch.Tensor) and v.is_cuda:
                        v.record_stream(cs)
                ent["seen"].add(id(self))
        return ent

    HR_TABLE_LDS = 256          # phase tables up to this size live whole in LDS (mirrors satu.hip)

    def _plan_hr_tiling(self, ent: dict, h: int, w: int, scale):
        """One-time (per size / scale) preparation of the HR stage: evaluate the phase table (and, for tables too large for
        LDS, its per-pixel expansion), read the range of the sampling offsets back and list every FEASIBLE launch plan -- wave
        split x HR tile whose double-buffered LRcat window (tile footprint + offset range + bilinear tap) fits the LDS.  Which
        plan runs is decided by measurement only: satu_hr() times the candidates once on the first real frame of this size /
        scale (there is no cost model).  Purely a performance plan: waves whose taps leave the window gather from global
        memory, so results never depend on it."""
        sw = C.byref(self.satu_w)
        n_table = ent["n_uh"] * ent["n_uw"]
        ent["ptab"] = None
        # The tables depend on (size, scale, weights) only, not on anything the compute stream holds: they are evaluated and read
        # back on a side stream, so the read-back's host wait does not stand behind the frames still in flight.
        cur = torch.cuda.current_stream()
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=self.dev)
        with torch.cuda.stream(self._side_stream):
            # (allocated under the side stream: a block the allocator recycles from the compute stream could still have work
            # pending there, and this stream does not wait for it)
            ent["table"] = torch.empty(n_table * _lib.SATU_TABLE, device=self.dev)
            ent["table"].record_stream(cur)
            _lib.check(self.lib.savsr_satu_phase_table(sw, ent["uh"].data_ptr(), ent["n_uh"], ent["uw"].data_ptr(), ent["n_uw"],
                                                       1.0 / scale[1], 1.0 / scale[0], ent["table"].data_ptr(), self._side_stream.cuda_stream),
                       "savsr_satu_phase_table")
            if n_table > self.HR_TABLE_LDS:
                ent["ptab"] = torch.empty(ent["H"] * ent["W"] * _lib.SATU_TABLE, device=self.dev)     # the table per HR pixel, offsets normalised
                ent["ptab"].record_stream(cur)
                _lib.check(self.lib.savsr_satu_expand_table(ent["table"].data_ptr(), ent["n_uw"], ent["ih"].data_ptr(), ent["iw"].data_ptr(), h, w,
                                                            ent["H"], ent["W"], ent["ptab"].data_ptr(), self._side_stream.cuda_stream), "savsr_satu_expand_table")
            tab = ent["table"].view(-1, _lib.SATU_TABLE).cpu().numpy()   # waits for the side stream only
        cur.wait_stream(self._side_stream)
        ox = np.concatenate([tab[:, 4], tab[:, 6]])
        oy = np.concatenate([tab[:, 5], tab[:, 7]])
        finite = bool(np.isfinite(ox).all() and np.isfinite(oy).all())
        forced = self.knobs.hr_tile                                         # "rows,cols32": experiments only
        forced_v = self.knobs.hr_variant
        nvar = int(self.lib.savsr_satu_hr_variants())

        def plans(tail_form: bool, variant: int) -> List[SatuTiling]:
            cw = int(self.lib.savsr_satu_hr_compute_waves(variant))                      # compute waves of a workgroup
            rpw = int(self.lib.savsr_satu_hr_rows_per_wave_tile(int(tail_form)))         # rows of a wave tile
            out = []
            if finite:
                rx, ry = float(ox.max() - ox.min()), float(oy.max() - oy.min())
                # tile rows: whole rounds of the compute waves first (rpw * cw, 2 rpw * cw), then the generic 8 / 16 / 32
                cands = [(int(forced.split(",")[0]), int(forced.split(",")[1]))] if forced else \
                    [(r, c) for c in (1, 2) for r in sorted({4, 8, 16, 32, rpw * cw, 2 * rpw * cw}) if r % 4 == 0 and r <= 64]   # (4 rows: the only window that fits below ~x1.6)
                for trows, tcols in cands:
                    lr_c = min(max(int(np.ceil(32 * tcols / scale[1] + rx)) + 2, 2), w)
                    lr_r = min(int(np.ceil(trows / scale[0] + ry)) + 2, h)
                    if self.lib.savsr_satu_hr_lds_bytes(int(tail_form), n_table, trows, tcols, lr_r, lr_c) > 160 * 1024 - 1024:
                        continue
                    t = SatuTiling()
                    t.variant, t.table_entries = variant, n_table
                    t.step_x, t.step_y = 1.0 / float(scale[1]), 1.0 / float(scale[0])
                    t.tile_rows, t.tile_cols32, t.lr_rows, t.lr_cols = trows, tcols, lr_r, lr_c
                    t.off_min_x, t.off_min_y = float(ox.min()), float(oy.min())
                    out.append(t)
            if not out:                                                    # no window fits (or non-finite offsets): gathers go to global memory
                t = SatuTiling()
                t.variant, t.table_entries = variant, n_table
                t.step_x, t.step_y = 1.0 / float(scale[1]), 1.0 / float(scale[0])
                t.tile_rows, t.tile_cols32, t.lr_rows, t.lr_cols = 8, 1, 0, 0
                t.off_min_x, t.off_min_y = 0.0, 0.0
                out.append(t)
            return out
        # the standalone 64-channel form (tests / taps only, never timed): the feasible plan with the fewest staged bytes per HR pixel
        ent["tiling"] = min(plans(False, 0), key=lambda t: (t.lr_rows * t.lr_cols) / float(t.tile_rows * t.tile_cols32 * 32))
        ent["tail_plans"] = [t for v in (range(nvar) if forced_v is None else [forced_v]) for t in plans(True, v)]
        ent["tiling_tail"] = ent["tail_plans"][0] if len(ent["tail_plans"]) == 1 else None

    @staticmethod
    def seam_floats(H: int, W: int) -> int:
        """Floats of the row-summed form's side buffer: [H][ceil(W / 32)][2 sides][9 groups]."""
        return ((H * ((W + 31) // 32) * 18 + 63) // 64) * 64

    @staticmethod
    def hr_plane(H: int, W: int) -> int:
        """Plane pitch (floats) of the planar HR feature map: H*W rounded up to 1 KiB plus 4352 B, so the
        64 channel planes of one pixel do not alias onto the same HBM channel (H*W*4 is a multiple of
        16 KiB at 720x1280)."""
        return ((H * W + 255) // 256) * 256 + 1088

    def satu_lr(self, x: Src, st: Src, row_px: int, h: int, w: int, tail_form: bool = False, q: bool = False, b: int = 0) -> torch.Tensor:
        """LR stage of SATU (kernel_conv + LeakyReLU + sta_conv + LR-side projections, savsr_arch.py:226-228,297-320).
        tail_form: the projections carry the tail conv's channel contraction (include/savsr_hip.h); q: in the row order of
        the row-summed form (savsr_satu_hr_tail_q)."""
        assert x.pix == st.pix
        if tail_form:
            lrcat = self.buf("satu.lrcat_tailq" if q else "satu.lrcat_tail", h, w, _lib.SATU_LRCAT_TAIL)
            fn, wts = self.lib.savsr_satu_lr_stage_tail, (self.satu_w_tailq if q else self.satu_w_tail)
        else:
            lrcat = self.buf("satu.lrcat", h, w, _lib.SATU_LRCAT)
            fn, wts = self.lib.savsr_satu_lr_stage, self.satu_w
        # (b: the clip of a batched launch sequence this call works on; the returned tensor is clip 0's copy either way)
        _lib.check(fn(C.byref(wts), x.ptr + b * x.bs, st.ptr + b * st.bs, x.pix, row_px, h, w, lrcat.data_ptr() + b * self._bs(lrcat), self._stream()), "savsr_satu_lr_stage")
        return lrcat

    def satu_hr(self, lrcat: torch.Tensor, h: int, w: int, scale, out: torch.Tensor, out_plane: Optional[int] = None, tail_form: bool = False,
                seam: Optional[torch.Tensor] = None, b: int = 0):
        """HR stage of SATU (grid_sample x2, expert mixing, fusion, savsr_arch.py:262-295,353-374) -> out [64] planes of [H][W];
        tail_form: -> the 27 tail-projected planes P, or with `seam` (seam_floats(H, W) floats) the row-summed form: out = the 9 planes Q
        (lrcat from satu_lr(..., q=True))."""
        ax = self.satu_axes(h, w, scale)       # incl. the phase table: a function of (size, scale, weights) only, evaluated once
        fn, wts = (self.lib.savsr_satu_hr_tail, self.satu_w_tail) if tail_form else (self.lib.savsr_satu_hr_upsample, self.satu_w)
        if seam is not None:
            assert tail_form
            fn, wts = self.lib.savsr_satu_hr_tail_q, self.satu_w_tailq
        sched = self.hr_sched.data_ptr() if not self.knobs.hr_static else None
        plane = out_plane if out_plane is not None else ax["H"] * ax["W"]

        p_lr, p_out = lrcat.data_ptr() + b * self._bs(lrcat), out.data_ptr() + b * self._bs(out)          # (clip b of a batched launch sequence)
        p_seam = None if seam is None else seam.data_ptr() + b * self._bs(seam)

        def launch(til):
            _lib.check(fn(C.byref(wts), p_lr, h, w, ax["table"].data_ptr(), ax["n_uh"], ax["n_uw"], ax["ih"].data_ptr(), ax["iw"].data_ptr(),
                          _ptr(ax["ptab"]), ax["gyn"].data_ptr(), ax["gxn"].data_ptr(), ax["H"], ax["W"],
                          C.byref(til), sched, p_out, plane, *(() if seam is None else (p_seam, seam.numel())), self._stream()), "savsr_satu_hr")
        if not tail_form:
            launch(ax["tiling"])
            return out
        if ax["tiling_tail"] is None:
            cands = ax["tail_plans"]
            ckey = (h, w, float(scale[0]), float(scale[1]))
            pick = lambda k: next((t for t in cands if (t.variant, t.tile_rows, t.tile_cols32) == k), cands[0])
            skey = ("scale", float(scale[0]), float(scale[1]))
            near = self._hr_choice.get(skey)    # (plan, h, w) measured at this scale on another LR size
            tab = self._hr_table.get((float(scale[0]), float(scale[1])))
            if ckey in self._hr_choice:         # (a sibling engine has timed this size / scale already)
                ax["tiling_tail"] = pick(self._hr_choice[ckey])
            elif tab is not None and 0.5 <= (h * w) / float(tab[1] * tab[2]) <= 2.0 and any((t.variant, t.tile_rows, t.tile_cols32) == tab[0] for t in cands):
                # measured for this build of the kernels at a comparable LR size (savsr_amd/hr_plans.json): nothing to time.  (A plan is a
                # function of the scale AND of how many tiles the image gives the 256 CUs: the x(3.5, 2) plan of a 180x320 frame ran a 204x636
                # frame's HR stage in 60.5 instead of 48.7 us, and a 64x112 frame has 91 tiles of 20 rows x 64 px -- outside 0.5 ... 2 x the
                # measured pixel count the engine measures, as before.)
                ax["tiling_tail"] = pick(tab[0])
                self._hr_choice[ckey] = tab[0]
            elif near is not None and 0.5 <= (h * w) / float(near[1] * near[2]) <= 2.0 and any((t.variant, t.tile_rows, t.tile_cols32) == near[0] for t in cands):
                # the folders of a YAML dataset differ by a few rows / columns at one scale (Vid4 x4: 144x180, 144x176, 120x180): the
                # plan is a function of the scale and the offset range far more than of the size -- one measurement per scale
                ax["tiling_tail"] = pick(near[0])
                self._hr_choice[ckey] = near[0]
            elif torch.cuda.is_current_stream_capturing():
                ax["tiling_tail"] = cands[0]
            else:                               # one-time choice by measurement: every plan writes the same `out`, bit for bit
                evs = []                        # (the plans' timings queue up on the stream; ONE host sync at the end)
                for til in cands:
                    launch(til)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        launch(til)
                    e1.record()
                    evs.append((e0, e1, til))
                evs[-1][1].synchronize()
                best = min(evs, key=lambda e: e[0].elapsed_time(e[1]))[2]
                if self.knobs.hr_print_plans:     # diagnostics: what every feasible plan measured (us per launch)
                    for e0, e1, t in evs:
                        print(f"[hr plan] {h}x{w} x{scale}: variant {t.variant} tile {t.tile_rows} x {32 * t.tile_cols32} window {t.lr_rows} x {t.lr_cols}: "
                              f"{1e3 * e0.elapsed_time(e1) / 3:.1f} us", flush=True)
                ax["tiling_tail"] = best
                self._hr_choice[ckey] = (best.variant, best.tile_rows, best.tile_cols32)
                self._hr_choice[skey] = (self._hr_choice[ckey], h, w)
        launch(ax["tiling_tail"])
        return out

    def satu(self, x: Src, st: Src, row_px: int, h: int, w: int, scale, out: torch.Tensor, out_plane: Optional[int] = None):
        """STAUpsample.forward (savsr_arch.py:315-376).  x, st: channel-last crops (row pitch row_px
        pixels) of [..][..][64] maps; out: [64][H][W] planar."""
        return self.satu_hr(self.satu_lr(x, st, row_px, h, w), h, w, scale, out, out_plane)
