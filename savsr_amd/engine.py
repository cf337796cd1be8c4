"""Host-side driver of the HIP kernels for the SAVSR inference path.

`HipEngine` = the launch sequence that replaces `SAVSR.forward` (/root/reference/lbasicsr/archs/savsr_arch.py:692-742): the stage
functions (bidirectional propagation, pyramid fusion, RCAN trunk + OSAdapt, SATU, tail), their capture into hipGraphs per
(shape, scale), the fan-out of independent clips over HIP streams and the batching of equal clips into the launches.  What it stands
on lives next door:

    packing.py   WeightPacking   state_dict -> split-bf16 weight images, OSConv banks, SATU matrices (once per engine)
    cache.py     ContextCache    (shape, scale) buffer contexts, arena, liveness plan, byte budget, eviction limbo
    launch.py    Launcher        descriptors + C-ABI calls: convs, OSConv weight generation, SATU tables / plans / stages
    config.py    EngineConfig    every SAVSR_* switch, read once when the engine is built

PyTorch is used for device memory and streams only: every arithmetic step is a call into libsavsr_hip.so through the C ABI of
include/savsr_hip.h.  There is no CPU / eager-PyTorch fallback.
"""
from __future__ import annotations

import gc
import os
import weakref
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ConvDesc, OSConvAttnDesc, SatuTiling, SatuWeights      # noqa: F401
from .cache import ContextCache
from .config import EngineConfig
from .launch import MAX_SUM_BLOCKS, Launcher, Src, _ptr      # noqa: F401
from .packing import (BN_EPS, CONV_TH, CONV_TW, WeightPacking, acc_row, conv_pack_geometry, conv_pack_index, conv_wy_pack_index, get_hw,      # noqa: F401  (re-exported:
                      pack_conv_part, pack_conv_weight, pack_conv_weight_wy, satu_axis_tables, split_bf16_image)                            # tests and tools import them from here)


class HipEngine(WeightPacking, ContextCache, Launcher):
    def __init__(self, state: Dict[str, torch.Tensor], cfg: dict, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("savsr_amd runs on an AMD GPU only (device 'cuda' under PyTorch-ROCm); "
                               "there is no CPU fallback")
        self.lib = _lib.load()
        self.dev = device
        self.knobs = kn = EngineConfig.from_env()        # every SAVSR_* switch, read here and nowhere else (config.py)
        # clips per batched launch sequence the library's batch limits allow: 6 convs (a block's two directions x 3 streams) and 2 OSConvs per clip
        self.NB_MAX = max(1, min(int(self.lib.savsr_conv2d_max_batch()) // 6, int(self.lib.savsr_osconv_weights_max_batch()) // 2))
        with torch.cuda.device(device):                      # per device: every kernel's > 64 KiB dynamic-LDS attribute, before any capture
            _lib.check(self.lib.savsr_prepare_device(), "savsr_prepare_device")
        self.cfg = dict(cfg)
        self.nf = cfg["num_feat"]
        if self.nf != 64:
            raise RuntimeError("the HIP SATU kernels are specialised for num_feat == 64")
        if cfg["slid_win"] != 3 or cfg["num_in_ch"] != 3:
            raise RuntimeError("the input-window packing is specialised for slid_win == 3, num_in_ch == 3")
        self.pw: Dict[str, tuple] = {}      # conv key -> (wimage, bias, cout, cin, ks)
        self.pw_wy: Dict[str, torch.Tensor] = {}      # conv key -> Winograd-y weight image (static 3x3 convs with cout % 64 == 0)
        self.osc: Dict[str, dict] = {}      # osconv key -> tensors
        self.se: Dict[str, tuple] = {}
        self._keep: List[torch.Tensor] = []
        self._init_caches()
        self.satu_events: Optional[list] = None     # bench.py: (start, end, clips) HIP events around the SATU stage(s) of a launch sequence
        self._st: Optional[int] = None              # cached stream handle while a frame's launches are being issued (_stream)
        # Clips of ONE (shape, scale) batched into the launches (round 5).  A small clip is launch-latency-bound -- 330 dependent launches at ~11 us
        # each whatever its size (tools/probe_small_clips.py) -- and more than three streams do not help (a conv workgroup holds its CU's LDS).  With
        # nb clips in one launch sequence every named buffer holds nb copies (`_bstride`: bytes between them), every conv / OSConv descriptor is
        # issued once per clip INSIDE the same batched launch (savsr_conv2d_batch takes 18 convs since ABI 26) and the per-clip kernels (SE gate,
        # SATU, tail, ...) are looped: 151 + 63 + nb x ~116 launches for nb clips instead of nb x 330.  Results are those of a one-clip launch
        # sequence of the SAME flow bit for bit (the convs of a batched launch are independent and their form is chosen by `form_nb`, not `nb`).  forward_many groups equal (shape, scale) clips up to
        # SAVSR_CLIP_BATCH (default 3) when the LR frame has at most SAVSR_CLIP_BATCH_MAX_PX pixels.
        self.nb = 1
        # Clips the conv-form rule of a launch counts (conv_launch): `clip_batch` for every frame of the throughput flow whose shape is eligible for
        # batching, 1 otherwise -- whatever `nb` the launch sequence at hand really carries.  A unit of 1, 2 or 3 clips of a folder therefore takes
        # the same form in every launch, and a clip's output does not depend on its group (remainders of a folder, the world-size partition).
        self.form_nb = 1
        self._bstride: Dict[int, int] = {}
        self.clip_batch = max(1, min(self.NB_MAX, kn.clip_batch))
        self.clip_batch_max_px = kn.clip_batch_max_px
        self.census: Optional[dict] = None          # bench.py: per-launch matrix-work census (_count_conv), shared with the sibling engines
        # SAVSR_CAPTURE_AFTER = n: a (shape, scale) context's first n frames run EAGERLY and the hipGraphs are captured on visit n + 1 (eager,
        # captured and replayed frames are the same launch sequence: bit-identical results).  Default 0 = capture on the first visit, by
        # measurement (bench.py --config run_test --emulate-world 8, cProfile of a rank's cold pass): the Python launch sequence of one frame costs
        # ~8 ms of host time -- more than the 4-5 ms the GPU needs for a Vid4-sized frame -- and a capture is that same sequence issued once, so
        # eager frames are host-bound and a block of >= 3 frames is already faster captured (8 + 4.5 n against 8 n ms).
        self.capture_after = kn.capture_after
        self.host_stats = {"captures": 0, "capture_s": 0.0, "plan_s": 0.0, "eager_frames": 0}     # shared with the sibling engines (bench.py)
        self.conv_algo = _lib.CONV_DIRECT           # CONV_DIRECT_THROUGHPUT while several clips are in flight (forward_many / batches)
        self._hr_choice: Dict[tuple, int] = {}      # (h, w, sh, sw) -> timed choice of the HR kernel's wave split; shared with the sibling engines
        self._hr_table = self._load_hr_plans()      # scale -> plan measured once per build of the SATU kernels (savsr_amd/hr_plans.json)
        self.use_graphs = kn.graphs
        # SATU in the row-summed tail form (savsr_satu_hr_tail_q + savsr_tail_gather_q: 9 planes + seams between the HR stage and the
        # tail instead of 27 planes); SAVSR_SATU_Q=0: the 27-plane form
        self.satu_q = kn.satu_q
        # static-weight 3x3 convs in the Winograd F(2,3)-along-y form (SAVSR_CONV_WINOGRAD_Y); SAVSR_CONV_WY=0: the direct kernel everywhere
        self.conv_wy = kn.conv_wy
        # OSConv weight generation as ONE launch (savsr_osconv_attn_desc.fused: the routing recomputed in every aggregation workgroup; bit-identical).
        # OFF: measured slower -- one clip 8.96 -> 9.77 ms, three in flight 120.3 -> 118.0 HR Mpixel/s (A/B/A on one lease): a workgroup pulling
        # the 0.8 MB of routing weights + pool partials through ONE CU takes ~40 us longer than the two extra launches it saves
        self.osconv_fused = kn.osconv_fused
        self.reuse_buffers = kn.reuse_buffers      # liveness-planned LR buffers (release()); 0: every name its own memory
        self.wy_min_tiles = kn.wy_min_tiles          # launches with at least this many 16-row tiles take the Winograd form ...
        self.wy_min_tiles_tp = kn.wy_min_tiles_tp    # ... or this many with several clips in flight (throughput tiling)
        self.n_streams = kn.streams   # launch units of a batch in flight concurrently ...
        # ... fewer when the frames are large (streams_for): a launch of a >= 40 kpx frame fills the chip on its own, a second stream fills the
        # launch boundaries and the conv tails, a third only adds contention (config 2, 16-24 clips per step on one lease: 4 clips x 2 streams
        # 130.4-130.9, x 3 streams 130.1 HR Mpixel/s; the YAML workflow 151 vs 145 frames/s); small clips are launch-latency-bound and want the
        # third (config 5: 329 clips/s with 3 streams, 311 with 2)
        self.n_streams_large, self.streams_large_px = kn.streams_large, kn.streams_large_px
        self._siblings: List["HipEngine"] = []
        self._streams: List[torch.cuda.Stream] = []
        self._pack_all({k: v.detach() for k, v in state.items()})

    NB_MAX = 4                  # (class default; the instance reads the library's batch limits: 24 convs / 8 OSConvs per launch => 4 clips)

    HR_PLANS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hr_plans.json")

    def _load_hr_plans(self) -> Dict[tuple, tuple]:
        """The HR stage's launch plan per scale as measured for THIS build of the SATU kernels (tools/tune_hr_plans.py writes the file with the
        library's savsr_source_hash_satu()): every rank of a multi-GPU run, and every run, then launches the same plan without timing the
        candidates on its first frame of a (folder, scale) -- eight ranks used to make eight measurements and could pick eight plans.  A table
        from another build is ignored (the engine measures, as before); so is an entry whose plan is not feasible for the offsets of the loaded
        weights.  Results never depend on the plan (bit-identical in every plan)."""
        if not self.knobs.hr_plans:
            return {}
        try:
            import json
            with open(self.HR_PLANS_FILE) as f:
                t = json.load(f)
            if t.get("satu_source_hash") != self.lib.savsr_source_hash_satu().decode():
                return {}
            # entry: [variant, tile rows, tile columns / 32, LR h, LR w of the measurement]
            return {tuple(float(v) for v in k.split(",")): (tuple(int(x) for x in p[:3]), int(p[3]), int(p[4])) for k, p in t.get("plans", {}).items() if len(p) >= 5}
        except (OSError, ValueError, AttributeError):
            return {}

    def clone_for_stream(self) -> "HipEngine":
        """A sibling engine that shares every read-only packed weight with this one but owns its
        dynamic state (OSConv scratch / weight images, buffers, SATU tables, graphs), so two clips can be
        in flight on two HIP streams."""
        e = HipEngine.__new__(HipEngine)
        e.lib, e.dev, e.cfg, e.nf = self.lib, self.dev, self.cfg, self.nf
        e.knobs = self.knobs
        e.NB_MAX = self.NB_MAX
        e.pw, e.se, e._keep = self.pw, self.se, self._keep
        e.pw_wy, e.conv_wy, e.wy_min_tiles, e.wy_min_tiles_tp = self.pw_wy, self.conv_wy, self.wy_min_tiles, self.wy_min_tiles_tp
        e.reuse_buffers, e.osconv_fused = self.reuse_buffers, self.osconv_fused
        e.satu_t, e.satu_w, e.tail_w, e.tail_b, e.gamma, e.n_l2 = self.satu_t, self.satu_w, self.tail_w, self.tail_b, self.gamma, self.n_l2
        e.iter_win, e.fwd_idx, e.bwd_idx = self.iter_win, self.fwd_idx, self.bwd_idx
        e.satu_tail_t, e.satu_w_tail = self.satu_tail_t, self.satu_w_tail
        e.satu_tailq_t, e.satu_w_tailq, e.satu_q = self.satu_tailq_t, self.satu_w_tailq, self.satu_q
        e.nb, e._bstride, e.clip_batch, e.clip_batch_max_px = 1, {}, self.clip_batch, self.clip_batch_max_px
        e.n_streams, e.n_streams_large, e.streams_large_px = self.n_streams, self.n_streams_large, self.streams_large_px
        e.form_nb = 1
        e.osc = {}
        for k, ent in self.osc.items():
            c = dict(ent)
            c.update(e._osc_scratch(ent["cin"], ent["cout"], ent["knum"], ent["nunits"] * 8))
            e.osc[k] = c
        e.se_gate = torch.empty_like(self.se_gate)
        e._init_caches()
        e.max_shapes, e.max_scales, e._budget, e._axes = self.max_shapes, self.max_scales, self._budget, self._axes
        self._budget["engines"].append(weakref.ref(e))
        e.satu_events, e.use_graphs, e.census, e._st = None, self.use_graphs, self.census, None
        e.capture_after, e.host_stats = self.capture_after, self.host_stats
        e.conv_algo = _lib.CONV_DIRECT
        e._hr_choice, e._hr_table = self._hr_choice, self._hr_table
        e._siblings, e._streams = [], []
        return e

    # ------------------------------------------------------------------ network pieces
    def residual_blocks(self, groups: List[Tuple[str, List[Src], str]], hp: int, wp: int, scale, use_osconv: bool) -> List[List[Src]]:
        """ResidualBlock (savsr_arch.py:399-415), cat-free, for several independent blocks of the same
        shape at once (the two propagation directions): the per-stream convs of all of them go out as
        single batched launches.  groups: (weight prefix, input streams, buffer tag)."""
        nf = self.nf
        L = ACT_LRELU
        x1s, d0 = [], []
        for pfx, xs, tag in groups:
            n = len(xs)
            x1 = [self.full(self.buf(f"{tag}.x1.{i}", hp, wp, nf)) for i in range(n)]
            pb = self.pool_buf(pfx + ".osconv", hp, wp, n * nf) if use_osconv else None     # OSConv pools cat(x1) (:146)
            d0 += [self.conv_desc(f"{pfx}.conv0.{i}", [xs[i]], x1[i], hp, wp, L, 0.2,
                                  pool=(pb, i * nf, n * nf) if use_osconv else None) for i in range(n)]
            x1s.append(x1)
        self.conv_launch(d0, "conv0")
        bases, d1 = [], []
        if use_osconv:                                   # the groups' OSConvs are independent: one batched weight generation
            keys = [pfx + ".osconv" for pfx, _, _ in groups]
            wy = self.osconv_wy(len(groups), nf, hp, wp)      # the dynamic convs below go out as ONE launch: its form decides the image's
            wds = self.osconv_launch(keys, [self.osconv_desc(k, x1, hp, wp, scale, pooled=True, wy=wy) for k, x1 in zip(keys, x1s)])
        for gi, ((pfx, xs, tag), x1) in enumerate(zip(groups, x1s)):
            base = self.full(self.buf(f"{tag}.base", hp, wp, nf))
            if use_osconv:
                d1.append(self.conv_desc(pfx + ".osconv", x1, base, hp, wp, L, 0.2, weights=wds[gi]))
            else:
                d1.append(self.conv_desc(pfx + ".conv1", x1, base, hp, wp, L, 0.2))
            bases.append(base)
        self.conv_launch(d1, "osconv" if use_osconv else "conv1")
        outs, d2 = [], []
        for (pfx, xs, tag), x1, base in zip(groups, x1s, bases):
            o = [self.full(self.buf(f"{tag}.out.{i}", hp, wp, nf)) for i in range(len(xs))]
            d2 += [self.conv_desc(f"{pfx}.conv2.{i}", [base, x1[i]], o[i], hp, wp, L, 0.2, res1=xs[i]) for i in range(len(xs))]
            outs.append(o)
        self.conv_launch(d2, "conv2")
        for x1, base in zip(x1s, bases):                 # the block's temporaries are dead (their last readers are enqueued)
            self.release(*x1, base)
        return outs

    def residual_block(self, pfx: str, xs: List[Src], hp: int, wp: int, scale, use_osconv: bool, tag: str) -> List[Src]:
        return self.residual_blocks([(pfx, xs, tag)], hp, wp, scale, use_osconv)[0]

    def windows_l1(self, units: List[Tuple[str, Src, Src, Src, str]], hp: int, wp: int, scale):
        """WindowUnit_l1 (savsr_arch.py:444-464) for independent units at once (f2p and p2f of one
        recurrence step).  units: (prefix, packed window [hp][wp][16], h_past, merge output, tag)."""
        nf = self.nf
        hcs = [self.buf(f"{tag}.hcs", hp, wp, 2 * nf) for _, _, _, _, tag in units]       # h_c | h_sup from one fused conv
        self.conv_launch([self.conv_desc(pfx + ".win", [win], self.full(h), hp, wp, ACT_LRELU, 0.2)
                          for (pfx, win, _, _, _), h in zip(units, hcs)], "win")
        feats = [[self.full(h, nf, 0), self.full(h, nf, nf), past] for (_, _, past, _, _), h in zip(units, hcs)]
        for k in range(self.cfg["w1_num_block"]):
            prev = feats
            feats = self.residual_blocks([(f"{u[0]}.blocks.{k}", f, f"{u[4]}.b{k}") for u, f in zip(units, feats)], hp, wp, scale, k >= 1)
            if k >= 1:                                   # the previous block's outputs (this block's inputs / residuals) are dead
                for f in prev:
                    self.release(*f)
        self.conv_launch([self.conv_desc(u[0] + ".merge", f, u[3], hp, wp) for u, f in zip(units, feats)], "merge")
        for f in feats:
            self.release(*f)
        self.release(*hcs)
        return [u[3] for u in units]

    def rcab(self, pfx: str, x: Src, out: Src, hp: int, wp: int, tag: str) -> Src:
        """savsr_arch.py:527-549."""
        nf = self.nf
        r1 = self.conv(pfx + ".0", [x], self.full(self.buf(f"{tag}.t1", hp, wp, nf)), hp, wp, ACT_RELU)
        part = self.pool_buf("se", hp, wp, nf)
        r2 = self.conv(pfx + ".2", [r1], self.full(self.buf(f"{tag}.t2", hp, wp, nf)), hp, wp, ACT_NONE, pool=(part, 0, nf))
        nblk = self.pool_rows(hp, wp)
        w1, b1, w2, b2, cm = self.se[pfx]
        st = self._stream()
        assert x.pix == nf and out.pix == nf
        # (one launch for all clips of a batched launch sequence: grid.y = clip, byte strides between the clips' operands)
        _lib.check(self.lib.savsr_se_scale_residual_batch(part.data_ptr(), nblk, 1.0 / (hp * wp), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                                          nf, cm, r2.ptr, x.ptr, out.ptr, hp * wp, self.nb, self._bs(part), r2.bs, x.bs, out.bs, st),
                   "savsr_se_scale_residual")
        return out

    def osadapt(self, g: int, x: Src, share: Optional[Src], out: Src, hp: int, wp: int, scale, pooled: bool = False) -> Src:
        """savsr_arch.py:186-214 fused with `+ gamma * share` of :732 (share=None: OSAdapt alone).
        pooled=True: the conv that produced x already wrote the pool partials of adapt.{g}.adapt."""
        m = f"adapt.{g}.mask"
        st = self._stream()
        c4 = self.pw[m + ".0"][2]
        h2, w2 = hp // 2, wp // 2
        m1 = self.conv(m + ".0", [x], self.full(self.buf("ad.m1", hp, wp, c4)), hp, wp, ACT_RELU)
        m2 = self.buf("ad.m2", h2, w2, c4)
        for b in range(self.nb):
            _lib.check(self.lib.savsr_avgpool2(m1.ptr + b * m1.bs, m2.data_ptr() + b * self._bs(m2), c4, hp, wp, st), "savsr_avgpool2")
        m3 = self.conv(m + ".4", [self.full(m2)], self.full(self.buf("ad.m3", h2, w2, c4)), h2, w2, ACT_RELU)
        m4 = self.conv(m + ".7", [m3], self.full(self.buf("ad.m4", h2, w2, c4)), h2, w2, ACT_RELU)
        m5 = self.buf("ad.m5", hp, wp, c4)
        for b in range(self.nb):
            _lib.check(self.lib.savsr_upsample2x(m4.ptr + b * m4.bs, m5.data_ptr() + b * self._bs(m5), c4, h2, w2, st), "savsr_upsample2x")
        mask = self.buf("ad.mask", hp, wp, 1)
        self.conv(m + ".11", [self.full(m5)], self.full(mask), hp, wp, ACT_SIGMOID)
        wd = self.osconv_weights(f"adapt.{g}.adapt", [x], hp, wp, scale, pooled=pooled, wy=self.osconv_wy(1, self.nf, hp, wp))
        return self.conv(f"adapt.{g}.adapt", [x], out, hp, wp, ACT_NONE, mul_px=mask, res1=x, res2=share,
                         res2_scale=self.gamma, weights=wd)

    # ------------------------------------------------------------------ diagnostics
    def time_satu_parts(self, lq: torch.Tensor, scale, timer) -> dict:
        """Diagnostics (tools/scale_sweep.py, bench.py): the SATU LR / HR launches and the tail of the product path, each timed
        alone by `timer(fn) -> us` on the tensors of a real frame.  lq: [T, 3, h, w] on the device."""
        lq = lq.contiguous()
        self._select(lq.shape, scale)
        c = self._stage_body(lq, scale)
        out = torch.empty(3, c["H"], c["W"], device=self.dev)
        q = self.satu_q
        lrcat = self.satu_lr(c["hfeat"], c["align"], c["wp"], c["h"], c["w"], tail_form=True, q=q)
        self._stage_satu(c, scale)
        hr = (lambda: self.satu_hr(lrcat, c["h"], c["w"], scale, c["q9"], c["plane"], tail_form=True, seam=c["seam"])) if q else \
            (lambda: self.satu_hr(lrcat, c["h"], c["w"], scale, c["p27"], c["plane"], tail_form=True))
        return {"satu_lr_us": timer(lambda: self.satu_lr(c["hfeat"], c["align"], c["wp"], c["h"], c["w"], tail_form=True, q=q)),
                "satu_hr_us": timer(hr),
                "tail_us": timer(lambda: self._stage_tail(c, lq, out))}

    # ------------------------------------------------------------------ whole frame
    def _stage_body(self, lq: torch.Tensor, scale) -> dict:
        with HipEngine._StageStream(self):
            return self._stage_body_impl(lq, scale)

    def _stage_body_impl(self, lq: torch.Tensor, scale) -> dict:
        """Everything up to the SATU inputs (savsr_arch.py:692-734).  lq: [T, 3, h, w] on device."""
        cfg, nf = self.cfg, self.nf
        if lq.dim() == 5:          # [nb, T, 3, h, w]: nb clips of one (shape, scale) in one launch sequence (see `nb`)
            assert lq.shape[0] == self.nb and lq.is_contiguous() and cfg["interval"] == 0
        else:
            assert self.nb == 1
        T, cin, h_in, w_in = lq.shape[-4:]
        clip_bytes = 4 * T * cin * h_in * w_in
        assert T == cfg["num_frame"] and cin == cfg["num_in_ch"] == 3
        if self.census is not None:
            k = "frames_tp" if self.conv_algo == _lib.CONV_DIRECT_THROUGHPUT else "frames_b1"
            self.census[k] = self.census.get(k, 0) + self.nb
        if h_in < 2 or w_in < 2:
            raise ValueError("SAVSR needs h, w >= 2")
        hp, wp = h_in + (h_in & 1), w_in + (w_in & 1)              # pad_spatial to even (savsr_arch.py:670-690)
        st = self._stream()
        sw, fw = cfg["slid_win"], cfg["fusion_win"]
        if cfg["interval"] == 0:
            wins = self.buf("windows", T - 2, hp, wp, 16)
            for b in range(self.nb):
                _lib.check(self.lib.savsr_pack_windows(lq.data_ptr() + b * clip_bytes, wins.data_ptr() + b * self._bs(wins), T, h_in, w_in, hp, wp, st), "savsr_pack_windows")
            win_b = win_f = lambda t: Src(wins, 16, 16, 0, float_off=(t - 1) * hp * wp * 16, bs=self._bs(wins))
            T = self.iter_win
        else:
            # frame_sample (:638-659, :699): each direction walks its own sub-sequence of the clip -- gathered (a device copy of
            # iter_win frames) and packed into its own window buffer
            T = self.iter_win
            packs = []
            for tag, idx in (("f", self.fwd_idx), ("b", self.bwd_idx)):
                sel = self.buf("frames_" + tag, T, 3, h_in, w_in)
                for k, fi in enumerate(idx[:T]):
                    sel[k].copy_(lq[fi])
                wb = self.buf("windows_" + tag, T - 2, hp, wp, 16)
                _lib.check(self.lib.savsr_pack_windows(sel.data_ptr(), wb.data_ptr(), T, h_in, w_in, hp, wp, st), "savsr_pack_windows")
                packs.append(wb)
            win_f = lambda t, wb=packs[0]: Src(wb, 16, 16, 0, float_off=(t - 1) * hp * wp * 16)
            win_b = lambda t, wb=packs[1]: Src(wb, 16, 16, 0, float_off=(t - 1) * hp * wp * 16)
        steps = T - sw + 1
        zero = self.buf("zero", hp, wp, nf)
        if self.nb == 1:
            zero.zero_()      # hidden state restarts from zero every window (savsr_arch.py:705-706)
        else:                 # (every clip's copy: the whole allocation behind the name)
            self._cur["raw"][zero.data_ptr()].zero_()
        hb = hf = self.full(zero)
        hpair = [self.buf(f"hpair{i}", hp, wp, 2 * nf) for i in range(steps)]   # cat(f2p[i], p2f[i]) of :721, written in place
        for idx in range(steps):                                                    # :708-719, both directions per launch
            cur_b, cur_f = T - 1 - sw // 2 - idx, idx + sw // 2
            hb, hf = self.windows_l1([("f2p_win", win_b(cur_b), hb, self.full(hpair[steps - 1 - idx], nf, 0), "f2p"),
                                      ("p2f_win", win_f(cur_f), hf, self.full(hpair[idx], nf, nf), "p2f")], hp, wp, scale)
        self.release(zero, *([wins] if cfg["interval"] == 0 else packs))             # (liveness: dead once the recurrence is through)
        # pyramid fusion (:616-618, :485-501, :721-722)
        level: List[Src] = [self.full(t) for t in hpair]
        for i in range(self.n_l2):
            u = f"h_win.{i}"
            ws = steps - 2 * i
            hfs = [self.full(self.buf(f"l2.{i}.hf{j}", hp, wp, nf)) for j in range(ws)]       # :488: ws independent convs, one launch
            self.conv_launch([self.conv_desc(f"{u}.conv_h.{j}", [level[j]], hfs[j], hp, wp, ACT_LRELU, 0.2) for j in range(ws)], "conv_h")
            self.release(*level)                         # this level's inputs (hpair at level 0) are dead
            nxt: List[Src] = []
            for j in range(ws - fw + 1):
                swf = hfs[j:j + fw]
                for k in range(cfg["w2_num_block"]):
                    prev = swf
                    swf = self.residual_block(f"{u}.blocks.{k}", swf, hp, wp, scale, True, f"l2.{i}.{j}.b{k}")
                    if k >= 1:
                        self.release(*prev)
                nxt.append(self.conv(u + ".merge", swf, self.full(self.buf(f"l2.{i}.o{j}", hp, wp, 2 * nf)), hp, wp))
                if cfg["w2_num_block"] >= 1:
                    self.release(*swf)
            self.release(*hfs)                           # (the windows of a level overlap: released once all of them are through)
            level = nxt
        align = self.conv("h_win_conv_h", [level[0]], self.full(self.buf("align", hp, wp, nf)), hp, wp, ACT_LRELU, 0.2)   # :723
        share = align
        hcur = align
        for g in range(cfg["n_resgroups"]):                                         # :728-732
            xin = hcur
            r = xin
            for k in range(cfg["n_resblocks"]):
                r = self.rcab(f"RG.{g}.residual_group.{k}.rcab", r, self.full(self.buf(f"rg.r{k & 1}", hp, wp, nf)), hp, wp, "rg")
            rg = self.conv(f"RG.{g}.conv", [r], self.full(self.buf("rg.out", hp, wp, nf)), hp, wp, res1=xin,
                           pool=(self.pool_buf(f"adapt.{g}.adapt", hp, wp, nf), 0, nf))      # OSAdapt's OSConv pools this tensor
            hcur = self.osadapt(g, rg, share, self.full(self.buf(f"rg.h{g & 1}", hp, wp, nf)), hp, wp, scale, pooled=True)
        hfeat = self.conv("conv_last", [hcur], self.full(self.buf("hfeat", hp, wp, nf)), hp, wp, res1=share)   # :733-734
        self.seal_buffers()                              # the LR buffer plan of this shape is final (align / hfeat / SATU buffers are never shared)
        H, W = get_hw(h_in, w_in, scale)
        plane = self.hr_plane(H, W)
        d = dict(align=align, hfeat=hfeat, wp=wp, h=h_in, w=w_in, H=H, W=W, plane=plane)
        if self.satu_q:
            d["q9"] = self.sbuf("satu.q9", 9, plane)
            d["seam"] = self.sbuf("satu.seam", self.seam_floats(H, W))
        else:                # (the 27 planes -- 99.5 MB at 720x1280 -- exist only in the 27-plane form; taps allocate them on demand)
            d["p27"] = self.sbuf("satu.p27", _lib.TAIL_PLANES, plane)
        return d

    def _stage_satu(self, c: dict, scale):
        with HipEngine._StageStream(self):
            return self._stage_satu_impl(c, scale)

    def _stage_satu_impl(self, c: dict, scale):
        """SATU in the tail-projected form (savsr_arch.py:315-376 with the channel contraction of :738 folded in): -> P [27][H][W]."""
        for b in range(self.nb):       # (per-clip kernels: looped over the clips of a batched launch sequence)
            if self.satu_q:        # row-summed form: the HR stage adds the horizontal taps itself -> 9 planes + seams
                lrcat = self.satu_lr(c["hfeat"], c["align"], c["wp"], c["h"], c["w"], tail_form=True, q=True, b=b)
                self.satu_hr(lrcat, c["h"], c["w"], scale, c["q9"], c["plane"], tail_form=True, seam=c["seam"], b=b)
                continue
            lrcat = self.satu_lr(c["hfeat"], c["align"], c["wp"], c["h"], c["w"], tail_form=True, b=b)       # crops of :737 via (row pitch, h, w)
            self.satu_hr(lrcat, c["h"], c["w"], scale, c["p27"], c["plane"], tail_form=True, b=b)

    def _stage_tail(self, c: dict, lq: torch.Tensor, out: torch.Tensor):
        with HipEngine._StageStream(self):
            return self._stage_tail_impl(c, lq, out)

    def _stage_tail_impl(self, c: dict, lq: torch.Tensor, out: torch.Tensor):
        """What is left of :738-739: the nine shifted taps per colour, the tail bias, the bilinear residual."""
        cfg = self.cfg
        T = lq.shape[-4]
        center = T // 2 if cfg["center_frame_idx"] is None else cfg["center_frame_idx"]
        clip_bytes, out_bytes = 4 * T * 3 * c["h"] * c["w"], 4 * 3 * c["H"] * c["W"]      # (lq [nb, T, 3, h, w] and out [nb, 3, H, W] are contiguous)
        for b in range(self.nb):
            cptr = lq.data_ptr() + b * clip_bytes + 4 * center * 3 * c["h"] * c["w"]    # unpadded centre frame (:696)
            if self.satu_q:
                _lib.check(self.lib.savsr_tail_gather_q(c["q9"].data_ptr() + b * self._bs(c["q9"]), c["plane"], c["seam"].data_ptr() + b * self._bs(c["seam"]),
                                                        c["seam"].numel(), self.tail_b.data_ptr(), cptr,
                                                        c["h"], c["w"], c["H"], c["W"], out.data_ptr() + b * out_bytes, self._stream()), "savsr_tail_gather_q")
                continue
            _lib.check(self.lib.savsr_tail_gather(c["p27"].data_ptr() + b * self._bs(c["p27"]), c["plane"], self.tail_b.data_ptr(), cptr,
                                                  c["h"], c["w"], c["H"], c["W"], out.data_ptr() + b * out_bytes, self._stream()), "savsr_tail_gather")

    def _satu_standalone(self, c: dict, scale) -> torch.Tensor:
        """STAUpsample.forward as such ([64][H][W]; tests / taps only -- the product path never materialises it)."""
        o = self.sbuf("satu.out", self.nf, c["plane"])
        self.satu(c["hfeat"], c["align"], c["wp"], c["h"], c["w"], scale, o, c["plane"])
        return o[:, : c["H"] * c["W"]].view(self.nf, c["H"], c["W"])

    def forward_one(self, lq: torch.Tensor, scale, out: torch.Tensor, taps: Optional[dict] = None):
        """Eager launch sequence.  lq: [T, 3, h, w] fp32 contiguous on device; out: [3, H, W] (or [nb, T, 3, h, w] -> [nb, 3, H, W]: nb clips
        of one (shape, scale) in one launch sequence)."""
        self.nb = int(lq.shape[0]) if lq.dim() == 5 else 1
        assert self.nb <= self.NB_MAX and (self.nb == 1 or taps is None)
        try:
            return self._forward_one(lq, scale, out, taps)
        finally:
            self.nb = 1

    def _forward_one(self, lq: torch.Tensor, scale, out: torch.Tensor, taps: Optional[dict] = None):
        self._select(lq.shape, scale)
        try:
            c = self._stage_body(lq, scale)
        except BaseException:
            self._abort_frame()
            raise
        if self.satu_events is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        self._stage_satu(c, scale)
        if self.satu_events is not None:
            ev1.record()
            self.satu_events.append((ev0, ev1, self.nb))
        if taps is not None:                    # channel-last [hp][wp][64] tensors; SATU output planar
            taps["align_feat"] = c["align"].t
            taps["h_feat"] = c["hfeat"].t
            taps["satu"] = self._satu_standalone(c, scale)
            if self.satu_q:     # the 27-plane form beside the row-summed one the frame runs (taps only)
                c["p27"] = self.sbuf("satu.p27", _lib.TAIL_PLANES, c["plane"])
                self.satu_hr(self.satu_lr(c["hfeat"], c["align"], c["wp"], c["h"], c["w"], tail_form=True), c["h"], c["w"], scale, c["p27"], c["plane"], tail_form=True)
            taps["p27"] = c["p27"][:, : c["H"] * c["W"]].view(_lib.TAIL_PLANES, c["H"], c["W"])
        self._stage_tail(c, lq, out)
        return out

    def _forward_graphed(self, lq: torch.Tensor, scale, out: torch.Tensor, throughput: bool = False):
        """hipGraph replay of the same launch sequence (three graphs: body | SATU | tail, so the SATU
        stage can be bracketed by HIP events).  The ~1400 launches of a frame cost ~11 us of host time
        each when issued from Python; captured once per (shape, scale) they replay in tens of us.
        throughput=True (several clips in flight on different streams): the convs are launched as
        SAVSR_CONV_DIRECT_THROUGHPUT -- the direct kernel's results bit for bit, its own captured graphs.  (Round 4: a launch's conv FORM --
        direct or Winograd-y -- depends on its tile count and on this mode (conv_launch), so a frame in throughput mode can differ from the
        one-clip flow by the two forms' rounding, ~1e-5; each mode is bitwise reproducible.)
        lq [nb, T, 3, h, w] / out [nb, 3, H, W]: nb clips of one (shape, scale) in ONE launch sequence (see `nb`), its own context and graphs."""
        self.nb = int(lq.shape[0]) if lq.dim() == 5 else 1
        assert self.nb <= self.NB_MAX
        try:
            return self._forward_graphed_impl(lq, scale, out, throughput)
        finally:
            self.nb = 1

    def _set_flow(self, lq: torch.Tensor, throughput: bool) -> None:
        """The two flows of a frame.  Latency (one clip in flight, `net(lq)` with b = 1): 8-row direct conv tiles, Winograd-y from 200 tiles.
        Throughput (several clips in flight: b >= 2, forward_many): 16-row tiles, Winograd-y from 100 tiles, counted as if `clip_batch` clips
        shared every launch when the shape is eligible for batching (`form_nb`).  Each flow is bitwise reproducible and independent of the
        grouping; the two differ from each other by the conv forms' rounding (~1e-5)."""
        self.conv_algo = _lib.CONV_DIRECT_THROUGHPUT if throughput else _lib.CONV_DIRECT
        h, w = int(lq.shape[-2]), int(lq.shape[-1])
        self.form_nb = self.clip_batch if (throughput and self.cfg["interval"] == 0 and h * w <= self.clip_batch_max_px) else 1

    def _forward_graphed_impl(self, lq: torch.Tensor, scale, out: torch.Tensor, throughput: bool = False):
        sc = self._select(lq.shape, scale)
        if sc["graphs"] is None:
            sc["graphs"] = {}
        self._set_flow(lq, throughput)
        g = sc["graphs"].get(throughput)
        if g is None:
            used = sc.setdefault("uses", {}).get(throughput, 0)
            if used < self.capture_after:            # the context's first frames: eager (see capture_after)
                sc["uses"][throughput] = used + 1
                self.host_stats["eager_frames"] += 1
                return self._forward_one(lq, scale, out)
            s_in = torch.empty_like(lq)
            s_out = torch.empty_like(out)
            s_in.copy_(lq)
            # Everything a capture cannot hold happens here, once per (size, scale): the SATU tables (host arithmetic, H2D copies,
            # one read-back of the offset range) and the measured choice of the HR launch plan, on this context's own LRcat / P
            # buffers (their contents do not matter: no control flow of the HR kernel depends on the feature values).  The kernels'
            # LDS attributes were set by savsr_prepare_device.  There is no eager run of the frame: the launch sequence is issued
            # exactly once, into the capture, and the buffers it names are allocated there (arena chunks from the graphs' pool).
            import time as _time
            _t0 = _time.perf_counter()
            h_, w_ = int(lq.shape[-2]), int(lq.shape[-1])
            H_, W_ = get_hw(h_, w_, scale)
            plane_ = self.hr_plane(H_, W_)
            if self.satu_q:
                self.satu_hr(self.buf("satu.lrcat_tailq", h_, w_, _lib.SATU_LRCAT_TAIL), h_, w_, scale, self.sbuf("satu.q9", 9, plane_), plane_,
                             tail_form=True, seam=self.sbuf("satu.seam", self.seam_floats(H_, W_)))
            else:
                self.satu_hr(self.buf("satu.lrcat_tail", h_, w_, _lib.SATU_LRCAT_TAIL), h_, w_, scale,
                             self.sbuf("satu.p27", _lib.TAIL_PLANES, plane_), plane_, tail_form=True)
            # (no host synchronisation here: the capture stream waits for this one -- `_capture` --, and a plan that had to be MEASURED has
            # synchronised on its own events.  A sync per new context stalled the host behind the units already queued on this stream, 38 ms a
            # time with three streams in flight: 2.6 s of a 6.5 s cold pass of the YAML workflow, during which the other streams got nothing new.)
            _t1 = _time.perf_counter()
            graphs = [torch.cuda.CUDAGraph() for _ in range(3)]
            ev = self.satu_events
            self.satu_events = None
            box = {}
            try:
                self._capture(graphs[0], None, lambda: box.update(c=self._stage_body(s_in, scale)))
                _t2 = _time.perf_counter()
                self._capture(graphs[1], graphs[0].pool(), lambda: self._stage_satu(box["c"], scale))
                self._capture(graphs[2], graphs[0].pool(), lambda: self._stage_tail(box["c"], s_in, s_out))
            except BaseException:
                self._abort_frame()
                raise
            finally:
                self.satu_events = ev
            self._charge(sc, s_in.numel() * 4 + s_out.numel() * 4)
            _t3 = _time.perf_counter()
            self.host_stats["captures"] += 1
            self.host_stats["plan_s"] += _t1 - _t0
            self.host_stats["capture_s"] += _t3 - _t1
            if self.knobs.profile_capture:
                print(f"[capture] {tuple(lq.shape)} x{scale}: plan {1e3 * (_t1 - _t0):.1f} ms, body {1e3 * (_t2 - _t1):.1f} ms "
                      f"(python launches {1e3 * box.get('t_launch', 0):.1f}), satu+tail {1e3 * (_t3 - _t2):.1f} ms", file=__import__("sys").stderr, flush=True)
            # The captured launches bake in the raw device pointers of this (size, scale)'s SATU tables (phase table, per-pixel
            # expansion, row / column index and coordinate arrays).  Replays never go through satu_axes(), so its LRU neither sees
            # them nor may it free them: the graph tuple owns a reference and the tables live exactly as long as the graph does.
            g = (s_in, s_out, graphs, self.satu_axes(lq.shape[-2], lq.shape[-1], scale))
            if g[3].get("ptab") is not None and not sc.get("ptab_charged"):
                self._charge(sc, g[3]["ptab"].numel() * 4)               # (the per-pixel table lives as long as a graph that names it)
                sc["ptab_charged"] = True
            sc["graphs"][throughput] = g
        s_in, s_out, graphs = g[:3]
        s_in.copy_(lq)
        graphs[0].replay()
        if self.satu_events is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        graphs[1].replay()
        if self.satu_events is not None:
            ev1.record()
            self.satu_events.append((ev0, ev1, self.nb))      # (start, end, clips whose SATU stages lie between them)
        graphs[2].replay()
        out.copy_(s_out)
        return out

    def _capture(self, graph: "torch.cuda.CUDAGraph", pool, fn) -> None:
        """Record fn()'s launches into `graph` on a side stream.  This is torch.cuda.graph() without its entry ritual
        (device-wide synchronize + gc.collect() + empty_cache() per graph: tens of ms, and the emptied cache turns the next
        shape's allocations into fresh hipMallocs) -- a YAML sweep captures one graph set per (folder, scale, stream)."""
        cur = torch.cuda.current_stream()
        if getattr(self, "_cap_stream", None) is None:
            self._cap_stream = torch.cuda.Stream(device=self.dev)
        cap = self._cap_stream
        cap.wait_stream(cur)
        # No cyclic garbage collection while a capture is open: a collector run triggered by the ~330 descriptor allocations of a frame could
        # finalise some OTHER object that owns device memory or a graph (an engine of an earlier test module, a dropped model), and freeing
        # device memory inside a capture aborts the process.  Objects that die by reference count are ours and die outside captures.
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.stream(cap):
                if pool is None:
                    graph.capture_begin()
                else:
                    graph.capture_begin(pool=pool)
                try:
                    fn()
                finally:
                    graph.capture_end()
        finally:
            if gc_was_on:
                gc.enable()
        cur.wait_stream(cap)

    def streams_for(self, px: float) -> int:
        """HIP streams (launch units in flight) for frames of `px` LR pixels on average."""
        return self.n_streams_large if px >= self.streams_large_px else self.n_streams

    def _ensure_streams(self, ns: int):
        while len(self._siblings) < ns - 1:
            self._siblings.append(self.clone_for_stream())
        while len(self._streams) < ns:
            self._streams.append(torch.cuda.Stream(device=self.dev))
        return [self] + self._siblings

    def forward_many(self, items) -> List[torch.Tensor]:
        """A stream of independent clips of MIXED shapes / scales (BASELINE config 5): items = [(lq [T, 3, h, w], (sh, sw))] ->
        [out [3, H, W]].  Clip i runs on HIP stream i % n_streams with that stream's sibling engine, so small clips (whose ~360
        launches are latency-bound) overlap.  Every clip's result is that of the throughput flow (`_set_flow`) whatever
        the grouping: forward_many(items)[i] == forward_many([items[i]])[0] bit for bit; against the one-clip latency flow of `forward` it
        agrees to the conv forms' rounding (~1e-5) where a launch takes another form."""
        if not self.use_graphs:            # SAVSR_GRAPHS=0 (diagnostics): the same flow issued eagerly, one clip after the other
            outs = []
            for lq, sc in items:
                lq = lq.to(torch.float32).contiguous()
                o = torch.empty((3,) + get_hw(lq.shape[-2], lq.shape[-1], sc), device=self.dev, dtype=torch.float32)
                self._set_flow(lq, True)
                outs.append(self.forward_one(lq, sc, o))
            return outs
        # (a lone clip, or SAVSR_STREAMS=1, takes the throughput flow too: what forward_many returns for a clip does not depend on how many
        # came with it or on how many streams carry them)
        # Launch units: clips of equal (shape, scale) whose LR frame is small enough to be launch-latency-bound go out up to `clip_batch` at a
        # time in ONE launch sequence (see `nb`); everything else one clip per unit, as before.  Units are dealt round-robin over the streams.
        units: List[List[int]] = []
        if self.clip_batch > 1 and self.cfg["interval"] == 0:
            from collections import OrderedDict
            groups: "OrderedDict[tuple, List[int]]" = OrderedDict()
            for i, (lq, sc) in enumerate(items):
                if lq.shape[-2] * lq.shape[-1] <= self.clip_batch_max_px:
                    groups.setdefault((tuple(lq.shape), float(sc[0]), float(sc[1])), []).append(i)
                else:
                    units.append([i])
            for idxs in groups.values():       # balanced units (10 clips -> 3 + 3 + 2 + 2, not 3 + 3 + 3 + 1: a lone clip would need a capture of its own)
                k = -(-len(idxs) // self.clip_batch)
                base, rem, a = len(idxs) // k, len(idxs) % k, 0
                for u in range(k):
                    n = base + (1 if u < rem else 0)
                    units.append(idxs[a:a + n])
                    a += n
            units.sort(key=lambda u: u[0])
        else:
            units = [[i] for i in range(len(items))]
        ns = min(self.streams_for(sum(lq.shape[-2] * lq.shape[-1] for lq, _ in items) / len(items)), len(units))
        engines = self._ensure_streams(max(ns, 1))
        cur = torch.cuda.current_stream()
        outs: List[Optional[torch.Tensor]] = [None] * len(items)
        for lq, sc in items:
            if lq.device != self.dev:
                raise RuntimeError(f"input on {lq.device}, engine on {self.dev}")
        for k in range(ns):
            self._streams[k].wait_stream(cur)
        for u, unit in enumerate(units):
            k = u % ns
            sc = items[unit[0]][1]
            H, W = get_hw(items[unit[0]][0].shape[-2], items[unit[0]][0].shape[-1], sc)
            with torch.cuda.stream(self._streams[k]):
                if len(unit) == 1:
                    i = unit[0]
                    outs[i] = torch.empty(3, H, W, device=self.dev, dtype=torch.float32)
                    engines[k]._forward_graphed(items[i][0].to(torch.float32).contiguous(), sc, outs[i], throughput=True)
                else:
                    lqb = torch.stack([items[i][0].to(torch.float32) for i in unit], 0)
                    outb = torch.empty(len(unit), 3, H, W, device=self.dev, dtype=torch.float32)
                    engines[k]._forward_graphed(lqb, sc, outb, throughput=True)
                    for j, i in enumerate(unit):
                        outs[i] = outb[j]
        for k in range(ns):
            cur.wait_stream(self._streams[k])
        for o in {id(t._base if t._base is not None else t): (t._base if t._base is not None else t) for t in outs if t is not None}.values():
            o.record_stream(cur)       # allocated under a side stream, handed to the caller's: its block is not recycled on the side stream while `cur` still reads it
        return outs

    def forward(self, lq: torch.Tensor, scale, taps: Optional[dict] = None) -> torch.Tensor:
        """lq: [b, T, 3, h, w] -> [b, 3, H, W] (savsr_arch.py:692-742)."""
        if lq.device != self.dev:
            raise RuntimeError(f"input on {lq.device}, engine on {self.dev}")
        lq = lq.to(torch.float32).contiguous()
        b, _, _, h, w = lq.shape
        H, W = get_hw(h, w, scale)
        out = torch.empty(b, 3, H, W, device=self.dev, dtype=torch.float32)
        if b >= 2 and self.n_streams >= 2 and self.use_graphs and taps is None:
            # clips are independent (no cross-clip state, savsr_arch.py:705-706): keep n_streams of them in flight
            # on separate HIP streams so one clip's load/store-bound kernel phases overlap another's MFMA phases
            # ... and up to `clip_batch` consecutive clips per launch sequence (see `nb`): the batch shares one (shape, scale)
            cb = self.clip_batch if (self.cfg["interval"] == 0 and h * w <= self.clip_batch_max_px) else 1
            units = [(i0, min(i0 + cb, b)) for i0 in range(0, b, cb)]
            ns = min(self.streams_for(h * w), len(units))
            engines = self._ensure_streams(ns)
            cur = torch.cuda.current_stream()
            for k in range(ns):
                self._streams[k].wait_stream(cur)
                engines[k].satu_events = self.satu_events
            for u, (i0, i1) in enumerate(units):
                with torch.cuda.stream(self._streams[u % ns]):
                    if i1 - i0 == 1:
                        engines[u % ns]._forward_graphed(lq[i0], scale, out[i0], throughput=True)
                    else:                # (contiguous slices of the batch: no copy)
                        engines[u % ns]._forward_graphed(lq[i0:i1], scale, out[i0:i1], throughput=True)
            for k in range(ns):
                cur.wait_stream(self._streams[k])
            return out
        self.conv_algo, self.form_nb = _lib.CONV_DIRECT, 1
        for i in range(b):      # samples are independent (OSConv groups=b, savsr_arch.py:166-167)
            if self.use_graphs and taps is None:
                self._forward_graphed(lq[i], scale, out[i])
            else:
                self.forward_one(lq[i], scale, out[i], taps if i == 0 else None)
        return out
