"""Host-side driver of the HIP kernels for the SAVSR inference path.

`HipEngine` owns (i) the weights re-laid-out once for the kernels (BatchNorm folded, conv
weights as split-bf16 images in MFMA lane order, SATU matrices pre-multiplied), (ii) a pool of
named channel-last device buffers
per input shape and (iii) the launch sequence that replaces `SAVSR.forward`
(/root/reference/lbasicsr/archs/savsr_arch.py:692-742).  PyTorch is used for device memory and
streams only: every arithmetic step below is a call into libsavsr_hip.so through the C ABI of
include/savsr_hip.h.  There is no CPU / eager fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._xfer import h2d
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ConvDesc, OSConvAttnDesc, SatuTiling, SatuWeights

BN_EPS = 1e-5
MAX_SUM_BLOCKS = 256     # workgroups of one savsr_channel_sums launch


# ----------------------------------------------------------------------------- host helpers (integer / grid logic)
def get_hw(h: int, w: int, scale: Sequence[float]) -> Tuple[int, int]:
    """Output size, savsr_arch.py:745-751 (Python round = half-to-even on the double product)."""
    return round(h * scale[0]), round(w * scale[1])


def satu_axis_tables(n_out: int, n_in: int, s: float):
    """Per-axis SATU tables, evaluated in fp32 exactly like the reference's torch CPU ops.

    Returns (coor, floor_idx, grid_norm):
      coor      = (i+.5)/s - floor((i+.5)/s + 1e-3) - .5          savsr_arch.py:331-333
      floor_idx = floor((i+.5)/s + 1e-3)  (the integer LR index grid, bit-exact contract)
      grid_norm = ((i+.5)/s - .5) * 2 / (n_in-1) - 1              savsr_arch.py:270-280
    """
    f32 = np.float32
    i = np.arange(n_out, dtype=np.float32)
    q = (i + f32(0.5)) / f32(s)
    fl = np.floor(q + f32(1e-3))
    coor = (q - fl) - f32(0.5)
    g = (i + f32(0.5)) / f32(s) - f32(0.5)
    g = (g * f32(2)) / f32(n_in - 1) - f32(1)
    return coor.astype(np.float32), fl.astype(np.int32), g.astype(np.float32)


_PACK_IDX_CACHE: Dict[Tuple[int, int, int], Tuple[np.ndarray, int]] = {}

CONV_TH, CONV_TW = 8, 32      # pixel tile of one conv workgroup (mirrors common.hpp)


def conv_pack_geometry(cout: int, cin: int, ks: int):
    kc = 16 if ks == 3 else 32
    cot = 64 if cout > 32 else 32
    if cin % kc:
        raise ValueError(f"conv cin={cin} must be a multiple of {kc} (pad the weight with zero channels)")
    return kc, cot, cin // kc, (cout + cot - 1) // cot


def conv_pack_index(cout: int, cin: int, ks: int):
    """Index map [cout, cin, ks*ks] -> position inside one part of the weight image
    (mirror of savsr_conv_pack_index)."""
    key = (cout, cin, ks)
    if key not in _PACK_IDX_CACHE:
        kc, cot, nchunk, ncob = conv_pack_geometry(cout, cin, ks)
        taps, nt, ksteps = ks * ks, cot // 32, kc // 16
        co = np.arange(cout, dtype=np.int64)[:, None, None]
        ci = np.arange(cin, dtype=np.int64)[None, :, None]
        tap = np.arange(taps, dtype=np.int64)[None, None, :]
        cob, col = co // cot, co % cot
        t, row = col // 32, col % 32
        chunk, cl = ci // kc, ci % kc
        kstep, kh, j = cl // 16, (cl % 16) // 8, cl % 8
        group = (((cob * nchunk + chunk) * taps + tap) * ksteps + kstep) * nt + t
        idx = group * 512 + (kh * 32 + row) * 8 + j
        total = ncob * nchunk * taps * kc * cot
        _PACK_IDX_CACHE[key] = (np.array(np.broadcast_to(idx, (cout, cin, taps))).reshape(-1), total)      # (a writable copy: torch.from_numpy warns on read-only views)
    return _PACK_IDX_CACHE[key]


_IDX_DEV_CACHE: Dict[tuple, torch.Tensor] = {}


def _index_on(kind: str, key: tuple, idx: np.ndarray, device: torch.device) -> torch.Tensor:
    """The (cached) index map of a weight-image layout as a tensor on `device`."""
    k = (kind, key, str(device))
    t = _IDX_DEV_CACHE.get(k)
    if t is None:
        t = torch.from_numpy(idx).to(device)
        _IDX_DEV_CACHE[k] = t
    return t


def _scatter_image(idx: np.ndarray, total: int, values: torch.Tensor, kind: str, key: tuple, device: Optional[torch.device]) -> torch.Tensor:
    """zeros[total] with values scattered to idx: numpy on the host, one index_put on a GPU (round 5: the engine packs its ~190 conv
    images and 12 OSConv banks ON THE DEVICE -- 1.7 s of host scatter / split work per process became a few ms; a rank of an 8-GPU run of
    a YAML spends 2-7 s on the GPU in all).  Pure data movement + RNE conversions: bit-identical either way
    (tests/test_gpu_kernels.py::test_weight_images_packed_on_device_equal_host_packing)."""
    if device is None or device.type == "cpu":
        out = np.zeros(total, dtype=np.float32)
        out[idx] = values.detach().to("cpu", torch.float32).contiguous().numpy().reshape(-1)
        return torch.from_numpy(out)
    out = torch.zeros(total, dtype=torch.float32, device=device)
    out[_index_on(kind, key, idx, device)] = values.detach().to(device, torch.float32).reshape(-1)
    return out


def pack_conv_part(w: torch.Tensor, device: Optional[torch.device] = None) -> torch.Tensor:
    """[cout, cin, k, k] -> fp32 tensor of one image part (zero padded), lane order; on `device` (default: host)."""
    cout, cin, ks, _ = w.shape
    idx, total = conv_pack_index(cout, cin, ks)
    return _scatter_image(idx, total, w, "direct", (cout, cin, ks), device)


def split_bf16_image(part: torch.Tensor) -> torch.Tensor:
    """fp32 part [n*512] -> int16 image [n][2][512]: hi = bf16(v), lo = bf16(v - hi) (RNE both)."""
    hi = part.to(torch.bfloat16)
    lo = (part - hi.to(torch.float32)).to(torch.bfloat16)
    img = torch.stack([hi.view(-1, 512), lo.view(-1, 512)], dim=1).contiguous()
    return img.view(torch.int16).reshape(-1)


def pack_conv_weight(w: torch.Tensor, device: Optional[torch.device] = None) -> torch.Tensor:
    """[cout, cin, k, k] -> split-bf16 weight image (int16 tensor) for savsr_conv2d."""
    return split_bf16_image(pack_conv_part(w, device))


_WY_IDX_CACHE: Dict[tuple, tuple] = {}


def conv_wy_pack_index(cout: int, cin: int):
    """Index map [4 pos, cout, cin, 3 kx] -> position inside one part of the Winograd-y weight image (mirror of
    savsr_conv_wy_pack_index): [cob][chunk][hf][s = vr * 3 + kx][t] groups of 512 = (kh * 32 + row) * 8 + j."""
    key = (cout, cin)
    if key not in _WY_IDX_CACHE:
        if cout % 64 or cin % 16:
            raise ValueError("Winograd-y conv image: cout must be a multiple of 64 and cin of 16")
        nchunk = cin // 16
        pos = np.arange(4, dtype=np.int64)[:, None, None, None]
        co = np.arange(cout, dtype=np.int64)[None, :, None, None]
        ci = np.arange(cin, dtype=np.int64)[None, None, :, None]
        kx = np.arange(3, dtype=np.int64)[None, None, None, :]
        cob, col = co // 64, co % 64
        t, row = col // 32, col % 32
        chunk, cl = ci // 16, ci % 16
        kh, j = cl // 8, cl % 8
        hf, vr = pos // 2, pos % 2
        group = (((cob * nchunk + chunk) * 2 + hf) * 6 + (vr * 3 + kx)) * 2 + t
        idx = group * 512 + (kh * 32 + row) * 8 + j
        total = (cout // 64) * nchunk * 12 * 16 * 64
        _WY_IDX_CACHE[key] = (np.array(np.broadcast_to(idx, (4, cout, cin, 3))).reshape(-1), total)
    return _WY_IDX_CACHE[key]


def pack_conv_weight_wy(w: torch.Tensor, device: Optional[torch.device] = None) -> torch.Tensor:
    """[cout, cin, 3, 3] -> split-bf16 Winograd-y weight image (SAVSR_CONV_WINOGRAD_Y): the F(2,3) weight transform over the tap ROWS
    g_ky in float64 -- U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2, per kx -- rounded to fp32, then (hi, lo)."""
    cout, cin, ks, _ = w.shape
    assert ks == 3
    dev = device if device is not None and device.type != "cpu" else torch.device("cpu")
    g = w.detach().to(dev, torch.float64)                                  # [co][ci][ky][kx]
    g0, g1, g2 = g[:, :, 0], g[:, :, 1], g[:, :, 2]
    u = torch.stack([g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2], 0).to(torch.float32)      # [pos][co][ci][kx]
    idx, total = conv_wy_pack_index(cout, cin)
    return split_bf16_image(_scatter_image(idx, total, u, "wy", (cout, cin), device))


def acc_row(r: int, half: int) -> int:
    """Row of register r of a 32x32 MFMA accumulator for lane half `half`."""
    return (r & 3) + 8 * (r >> 2) + 4 * half


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


class HipEngine:
    def __init__(self, state: Dict[str, torch.Tensor], cfg: dict, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("savsr_amd runs on an AMD GPU only (device 'cuda' under PyTorch-ROCm); "
                               "there is no CPU fallback")
        self.lib = _lib.load()
        self.dev = device
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
        self.clip_batch = max(1, min(self.NB_MAX, int(os.environ.get("SAVSR_CLIP_BATCH", "3"))))
        self.clip_batch_max_px = int(os.environ.get("SAVSR_CLIP_BATCH_MAX_PX", str(200 * 352)))
        self.census: Optional[dict] = None          # bench.py: per-launch matrix-work census (_count_conv), shared with the sibling engines
        # SAVSR_CAPTURE_AFTER = n: a (shape, scale) context's first n frames run EAGERLY and the hipGraphs are captured on visit n + 1 (eager,
        # captured and replayed frames are the same launch sequence: bit-identical results).  Default 0 = capture on the first visit, by
        # measurement (bench.py --config run_test --emulate-world 8, cProfile of a rank's cold pass): the Python launch sequence of one frame costs
        # ~8 ms of host time -- more than the 4-5 ms the GPU needs for a Vid4-sized frame -- and a capture is that same sequence issued once, so
        # eager frames are host-bound and a block of >= 3 frames is already faster captured (8 + 4.5 n against 8 n ms).
        self.capture_after = max(0, int(os.environ.get("SAVSR_CAPTURE_AFTER", "0")))
        self.host_stats = {"captures": 0, "capture_s": 0.0, "plan_s": 0.0, "eager_frames": 0}     # shared with the sibling engines (bench.py)
        self.conv_algo = _lib.CONV_DIRECT           # CONV_DIRECT_THROUGHPUT while several clips are in flight (forward_many / batches)
        self._hr_choice: Dict[tuple, int] = {}      # (h, w, sh, sw) -> timed choice of the HR kernel's wave split; shared with the sibling engines
        self._hr_table = self._load_hr_plans()      # scale -> plan measured once per build of the SATU kernels (savsr_amd/hr_plans.json)
        self.use_graphs = os.environ.get("SAVSR_GRAPHS", "1") != "0"
        # SATU in the row-summed tail form (savsr_satu_hr_tail_q + savsr_tail_gather_q: 9 planes + seams between the HR stage and the
        # tail instead of 27 planes); SAVSR_SATU_Q=0: the 27-plane form
        self.satu_q = os.environ.get("SAVSR_SATU_Q", "1") != "0"
        # static-weight 3x3 convs in the Winograd F(2,3)-along-y form (SAVSR_CONV_WINOGRAD_Y); SAVSR_CONV_WY=0: the direct kernel everywhere
        self.conv_wy = os.environ.get("SAVSR_CONV_WY", "1") != "0"
        # OSConv weight generation as ONE launch (savsr_osconv_attn_desc.fused: the routing recomputed in every aggregation workgroup; bit-identical).
        # OFF: measured slower -- one clip 8.96 -> 9.77 ms, three in flight 120.3 -> 118.0 HR Mpixel/s (A/B/A on one lease): a workgroup pulling
        # the 0.8 MB of routing weights + pool partials through ONE CU takes ~40 us longer than the two extra launches it saves
        self.osconv_fused = os.environ.get("SAVSR_OSCONV_FUSED", "0") != "0"
        self.reuse_buffers = os.environ.get("SAVSR_REUSE_BUFFERS", "1") != "0"      # liveness-planned LR buffers (release()); 0: every name its own memory
        self.wy_min_tiles = int(os.environ.get("SAVSR_WY_MIN_TILES", "200"))          # launches with at least this many 16-row tiles take the Winograd form ...
        self.wy_min_tiles_tp = int(os.environ.get("SAVSR_WY_MIN_TILES_TP", "100"))    # ... or this many with several clips in flight (throughput tiling)
        self.n_streams = max(1, int(os.environ.get("SAVSR_STREAMS", "3")))   # clips of a batch in flight concurrently
        self._siblings: List["HipEngine"] = []
        self._streams: List[torch.cuda.Stream] = []
        self._pack_all({k: v.detach() for k, v in state.items()})

    NB_MAX = 3                  # (class default; the instance reads the library's batch limits: 18 convs / 6 OSConvs per launch => 3 clips)

    HR_PLANS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hr_plans.json")

    def _load_hr_plans(self) -> Dict[tuple, tuple]:
        """The HR stage's launch plan per scale as measured for THIS build of the SATU kernels (tools/tune_hr_plans.py writes the file with the
        library's savsr_source_hash_satu()): every rank of a multi-GPU run, and every run, then launches the same plan without timing the
        candidates on its first frame of a (folder, scale) -- eight ranks used to make eight measurements and could pick eight plans.  A table
        from another build is ignored (the engine measures, as before); so is an entry whose plan is not feasible for the offsets of the loaded
        weights.  Results never depend on the plan (bit-identical in every plan)."""
        if os.environ.get("SAVSR_HR_PLANS", "1") == "0":
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

    # ------------------------------------------------------------------ weight preparation
    def _dev(self, t: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        d = t.to(self.dev, dtype).contiguous()
        self._keep.append(d)
        return d

    def _fold(self, sd, key: str, bn: Optional[str]):
        w = sd[key + ".weight"].to("cpu", torch.float32)
        b = sd.get(key + ".bias")
        b = None if b is None else b.to("cpu", torch.float32)
        if bn is not None:      # eval BatchNorm folded into the conv (savsr_arch.py:191,196,199,204)
            s = sd[bn + ".weight"].cpu() / torch.sqrt(sd[bn + ".running_var"].cpu() + BN_EPS)
            w = w * s.view(-1, 1, 1, 1)
            b0 = b if b is not None else torch.zeros_like(s)
            b = (b0 - sd[bn + ".running_mean"].cpu()) * s + sd[bn + ".bias"].cpu()
        return w, b

    def _register(self, key: str, w: torch.Tensor, b: Optional[torch.Tensor]):
        cout, cin, ks, _ = w.shape
        bias = None if b is None else self._dev(b)
        wd = w.to(self.dev)                                  # (the images are built on the device: _scatter_image)
        self.pw[key] = (self._dev(pack_conv_weight(wd, self.dev), torch.int16), bias, cout, cin, ks)
        if self.conv_wy and ks == 3 and cout % 64 == 0 and cin % 16 == 0:
            # static 3x3 weights also as the Winograd F(2,3)-along-y image (conv_wy.hip: 2/3 of the matrix work); which form a launch takes is
            # decided per launch in conv_launch (the 16-row Winograd tiles need a launch that fills the chip)
            self.pw_wy[key] = self._dev(pack_conv_weight_wy(wd, self.dev), torch.int16)

    def _add_conv(self, sd, key: str, bn: Optional[str] = None):
        w, b = self._fold(sd, key, bn)
        self._register(key, w, b)

    def _add_window_conv(self, sd, d: str):
        """conv_c (3->64) and conv_sup (6->64) of one direction fused into a 16 -> 128 conv over
        the packed window tensor (channels: frame t | t-1 | t+1 | zeros), savsr_arch.py:429-431,456-457."""
        nf = self.nf
        wc, bc = sd[d + ".conv_c.weight"].cpu().float(), sd[d + ".conv_c.bias"].cpu().float()
        ws, bs = sd[d + ".conv_sup.weight"].cpu().float(), sd[d + ".conv_sup.bias"].cpu().float()
        w = torch.zeros(2 * nf, 16, 3, 3)
        w[:nf, 0:3] = wc
        w[nf:, 3:9] = ws
        self._register(d + ".win", w, torch.cat([bc, bs]))

    def _add_osconv(self, sd, key: str):
        bank = sd[key + ".weight"].to(self.dev, torch.float32)    # [K, cout, cin, 3, 3]
        knum, cout, cin = bank.shape[:3]
        packed = torch.stack([pack_conv_part(bank[k], self.dev) for k in range(knum)], 0)
        a = key + ".attention"
        bn_s = sd[a + ".bn.weight"].cpu() / torch.sqrt(sd[a + ".bn.running_var"].cpu() + BN_EPS)
        bn_b = sd[a + ".bn.bias"].cpu() - sd[a + ".bn.running_mean"].cpu() * bn_s
        hidden = sd[a + ".fc.weight"].shape[0]
        g = lambda k: self._dev(sd[k].reshape(sd[k].shape[0], -1) if sd[k].dim() > 1 else sd[k])
        elems = packed.shape[1]
        ent = dict(cin=cin, cout=cout, knum=knum, hidden=hidden, bank=self._dev(packed), nunits=elems // 8,
                   l1_w=g(key + ".scale_routing.0.weight"), l1_b=g(key + ".scale_routing.0.bias"),
                   l2_w=g(key + ".scale_routing.2.weight"), l2_b=g(key + ".scale_routing.2.bias"),
                   fc_w=g(a + ".fc.weight"), bn_scale=self._dev(bn_s), bn_shift=self._dev(bn_b),
                   ch_w=g(a + ".channel_fc.weight"), ch_b=g(a + ".channel_fc.bias"),
                   fl_w=g(a + ".filter_fc.weight"), fl_b=g(a + ".filter_fc.bias"),
                   sp_w=g(a + ".spatial_fc.weight"), sp_b=g(a + ".spatial_fc.bias"),
                   kn_w=g(a + ".kernel_fc.weight"), kn_b=g(a + ".kernel_fc.bias"),
                   **self._osc_scratch(cin, cout, knum, elems))
        self.osc[key] = ent

    def _osc_scratch(self, cin: int, cout: int, knum: int, elems: int) -> dict:
        """Per-engine scratch of one OSConv (routing vectors, gates, the generated weight images), NB_MAX copies: one per clip of a batched
        launch sequence (the tensors handed around are clip 0's; `_bstride` knows the distance to the next)."""
        nb = self.NB_MAX
        al = lambda n, unit: ((n * unit + 255) // 256) * 256 // unit          # copies stay 256-byte aligned
        out = {}
        for name, n, dt in (("v1", 2 * cin, torch.float32), ("v2", cin, torch.float32), ("att", cin + cout + 9 + knum, torch.float32),
                            ("wdyn", 2 * elems, torch.int16), ("wdyn_wy", 2 * (elems * 4 // 3) if cout % 64 == 0 else 0, torch.int16)):      # (12 taps instead of 9)
            unit = 4 if dt == torch.float32 else 2
            pitch = al(n, unit)
            full = torch.empty(nb * pitch, device=self.dev, dtype=dt)
            self._keep.append(full)
            t = full[:n]
            self._bstride[t.data_ptr()] = pitch * unit
            out[name] = t
        return out

    def _pack_satu(self, sd):
        p = "upsample."
        c = self.nf
        f32 = torch.float32
        wk = sd[p + "kernel_conv.0.weight"].to("cpu", f32).reshape(25 * c, c).numpy()     # [n = 25 ch + tap][k]
        bk = sd[p + "kernel_conv.0.bias"].to("cpu", f32).numpy()
        lane = np.arange(64)
        li, lh = lane & 31, lane >> 5
        jj = np.arange(8)
        # kconv part [tap][cg][ks][lane][j] = Wk[25 (32 cg + (lane & 31)) + tap][16 ks + 8 (lane >> 5) + j]
        tap = np.arange(25)[:, None, None, None, None]
        cg = np.arange(2)[None, :, None, None, None]
        ks = np.arange(4)[None, None, :, None, None]
        n_idx = 25 * (32 * cg + li[None, None, None, :, None]) + tap
        k_idx = 16 * ks + 8 * lh[None, None, None, :, None] + jj[None, None, None, None, :]
        n_idx, k_idx = np.broadcast_arrays(n_idx, k_idx)
        kconv = wk[n_idx, k_idx].astype(np.float32)                                  # [25,2,4,64,8]
        kconv_b = bk.reshape(c, 25).T.copy()                                          # [tap][ch]
        fus = sd[p + "fusion.weight"].to("cpu", f32).reshape(c, 2 * c).numpy()
        wa, wb = fus[:, :c], fus[:, c:]                                              # cat((sta, fea)), :374
        comp = sd[p + "weight_compress"].to("cpu", f32).reshape(4, 8, c).numpy()     # C_m[j][c]
        expd = sd[p + "weight_expand"].to("cpu", f32).reshape(4, c, 8).numpy()       # E_n[c][j]
        # projections, one 512-element group per (matrix tile, k step): [lane][j]
        pa = np.zeros((2, 4, 64, 8), dtype=np.float32)
        pb = np.zeros((2, 4, 64, 8), dtype=np.float32)
        pc = np.zeros((4, 64, 8), dtype=np.float32)
        for t in range(2):
            for kidx in range(4):
                cgi, s = kidx // 2, kidx % 2
                # k order of an accumulator used as B operand: row 16 s + 8 (j >> 2) + 4 half + (j & 3)
                ch = 32 * cgi + 16 * s + 8 * (jj[None, :] >> 2) + 4 * lh[:, None] + (jj[None, :] & 3)
                pa[t, kidx] = wa[(32 * t + li)[:, None], ch]
            for ksi in range(4):
                pb[t, ksi] = wb[(32 * t + li)[:, None], 16 * ksi + 8 * lh[:, None] + jj[None, :]]
        cstack = comp.reshape(32, c)                                                  # row 8 m + j (natural order in the record)
        for ksi in range(4):
            pc[ksi] = cstack[li[:, None], 16 * ksi + 8 * lh[:, None] + jj[None, :]]
        proj = np.concatenate([pa.reshape(-1), pb.reshape(-1), pc.reshape(-1)])
        wbe = np.einsum("oc,ncj->noj", wb.astype(np.float64), expd.astype(np.float64)).astype(np.float32)   # (Wb E_n)[co][j]
        wbe_p = np.zeros((2, 2, 64, 8), dtype=np.float32)                            # [t][ks][lane][j], k = 16 ks + 8 kh + j = 8 n + j
        for t in range(2):
            for ksi in range(2):
                wbe_p[t, ksi] = wbe[(2 * ksi + lh)[:, None], (32 * t + li)[:, None], jj[None, :]]
        fb = sd[p + "fusion.bias"].to("cpu", f32).numpy()
        fb_p = np.zeros((2, 32), dtype=np.float32)
        for hh in range(2):
            for t in range(2):
                for r in range(16):
                    fb_p[hh, 16 * t + r] = fb[32 * t + acc_row(r, hh)]
        head_w = torch.cat([sd[p + "routing.0.weight"], sd[p + "offset.weight"], sd[p + "st_offset.weight"]], 0)
        head_b = torch.cat([sd[p + "routing.0.bias"], sd[p + "offset.bias"], sd[p + "st_offset.bias"]], 0)
        t_ = lambda a: self._dev(torch.from_numpy(np.ascontiguousarray(a)))
        img = lambda a: self._dev(split_bf16_image(torch.from_numpy(np.ascontiguousarray(a.reshape(-1)))), torch.int16)
        self.satu_t = dict(
            body0_w=self._dev(sd[p + "body.0.weight"].reshape(64, 4)), body0_b=self._dev(sd[p + "body.0.bias"]),
            body2_w=self._dev(sd[p + "body.2.weight"].reshape(64, 64).t()), body2_b=self._dev(sd[p + "body.2.bias"]),
            head_w=self._dev(head_w.reshape(8, 64)), head_b=self._dev(head_b),
            kconv_w=img(kconv), kconv_b=t_(kconv_b), proj_w=img(proj), wbe_w=img(wbe_p), fusion_b=t_(fb_p))
        sw = SatuWeights()
        for k, v in self.satu_t.items():
            setattr(sw, k, v.data_ptr())
        self.satu_w = sw
        self.tail_w = self._dev(sd["tail.weight"].reshape(3, 64 * 9))
        self.tail_b = self._dev(sd["tail.bias"])
        # ---- tail-projected form (include/savsr_hip.h, savsr_satu_*_tail): the 3x3 tail conv's channel contraction
        # Wt27[p][c] (rows 27..31 zero) multiplied into fusion / expand / the LR projections in float64.  Two row orders:
        # p = 3 (3 ky + kx) + o (savsr_satu_hr_tail + savsr_tail_gather), and the row-summed form's (savsr_satu_hr_tail_q: the three kx
        # of group g = 3 ky + o at MFMA rows acc_row(3 gi + kx, half), groups 0 .. 4 in lane half 0, 5 .. 8 in half 1)
        tw = sd["tail.weight"].to("cpu", torch.float64).numpy()                          # [3 o][64 c][3 ky][3 kx]

        def fold(row_of):
            wt27 = np.zeros((32, c), dtype=np.float64)
            for ky in range(3):
                for kx in range(3):
                    for o in range(3):
                        wt27[row_of(ky, kx, o)] = tw[o, :, ky, kx]
            ta = (wt27 @ wa.astype(np.float64)).astype(np.float32)                           # [32][64] applies to sta
            tb = (wt27 @ wb.astype(np.float64)).astype(np.float32)                           # [32][64] applies to x
            pa1 = np.zeros((1, 4, 64, 8), dtype=np.float32)
            pb1 = np.zeros((1, 4, 64, 8), dtype=np.float32)
            for kidx in range(4):
                cgi, s_ = kidx // 2, kidx % 2
                ch = 32 * cgi + 16 * s_ + 8 * (jj[None, :] >> 2) + 4 * lh[:, None] + (jj[None, :] & 3)
                pa1[0, kidx] = ta[li[:, None], ch]
                pb1[0, kidx] = tb[li[:, None], 16 * kidx + 8 * lh[:, None] + jj[None, :]]
            proj1 = np.concatenate([pa1.reshape(-1), pb1.reshape(-1), pc.reshape(-1)])
            twbe = np.einsum("pc,ncj->npj", wt27 @ wb.astype(np.float64), expd.astype(np.float64)).astype(np.float32)   # (Wt27 Wb E_n)[p][j]
            twbe_p = np.zeros((1, 2, 64, 8), dtype=np.float32)
            for ksi in range(2):
                twbe_p[0, ksi] = twbe[(2 * ksi + lh)[:, None], li[:, None], jj[None, :]]
            tfb = (wt27 @ fb.astype(np.float64)).astype(np.float32)
            tfb_p = np.zeros((2, 16), dtype=np.float32)
            for hh in range(2):
                for r in range(16):
                    tfb_p[hh, r] = tfb[acc_row(r, hh)]
            tens = dict(proj_w=img(proj1), wbe_w=img(twbe_p), fusion_b=t_(tfb_p))
            swt = SatuWeights()
            for k, v in self.satu_t.items():
                setattr(swt, k, v.data_ptr())
            for k, v in tens.items():
                setattr(swt, k, v.data_ptr())
            return tens, swt

        def row_q(ky, kx, o):
            g = 3 * ky + o
            return acc_row(3 * g + kx, 0) if g < 5 else acc_row(3 * (g - 5) + kx, 1)
        self.satu_tail_t, self.satu_w_tail = fold(lambda ky, kx, o: 3 * (3 * ky + kx) + o)
        self.satu_tailq_t, self.satu_w_tailq = fold(row_q)

    def _pack_all(self, sd):
        cfg = self.cfg
        for d in ("f2p_win", "p2f_win"):
            self._add_window_conv(sd, d)
            for k in range(cfg["w1_num_block"]):
                b = f"{d}.blocks.{k}"
                for i in range(3):
                    self._add_conv(sd, f"{b}.conv0.{i}")
                    self._add_conv(sd, f"{b}.conv2.{i}")
                if k >= 1:
                    self._add_osconv(sd, b + ".osconv")
                else:
                    self._add_conv(sd, b + ".conv1")
            self._add_conv(sd, d + ".merge")
        from .archs.savsr_arch import frame_sample_indices, iteration_window
        center = cfg["num_frame"] // 2 if cfg["center_frame_idx"] is None else cfg["center_frame_idx"]
        self.iter_win = iteration_window(cfg["num_frame"], cfg["interval"], center)      # frames per propagation direction (:597-604)
        self.fwd_idx, self.bwd_idx = frame_sample_indices(cfg["num_frame"], cfg["interval"])   # frame_sample (:638-659)
        if cfg["interval"] != 0 and (len(self.fwd_idx) < self.iter_win or len(self.bwd_idx) < self.iter_win):
            raise ValueError("num_frame / interval: the sampled frame lists are shorter than the iteration window")
        steps = self.iter_win - cfg["slid_win"] + 1
        self.n_l2 = (self.iter_win - cfg["fusion_win"] + 1) // 2
        for i in range(self.n_l2):
            u = f"h_win.{i}"
            for j in range(steps - 2 * i):
                self._add_conv(sd, f"{u}.conv_h.{j}")
            for k in range(cfg["w2_num_block"]):
                b = f"{u}.blocks.{k}"
                for j in range(cfg["fusion_win"]):
                    self._add_conv(sd, f"{b}.conv0.{j}")
                    self._add_conv(sd, f"{b}.conv2.{j}")
                self._add_osconv(sd, b + ".osconv")
            self._add_conv(sd, u + ".merge")
        self._add_conv(sd, "h_win_conv_h")
        for g in range(cfg["n_resgroups"]):
            for k in range(cfg["n_resblocks"]):
                r = f"RG.{g}.residual_group.{k}.rcab"
                self._add_conv(sd, r + ".0")
                self._add_conv(sd, r + ".2")
                a = r + ".3.attention"
                cm = sd[a + ".1.weight"].shape[0]
                self.se[r] = (self._dev(sd[a + ".1.weight"].reshape(cm, -1)), self._dev(sd[a + ".1.bias"]),
                              self._dev(sd[a + ".3.weight"].reshape(-1, cm)), self._dev(sd[a + ".3.bias"]), cm)
            self._add_conv(sd, f"RG.{g}.conv")
            m = f"adapt.{g}.mask"
            self._add_conv(sd, m + ".0", bn=m + ".1")
            self._add_conv(sd, m + ".4", bn=m + ".5")
            self._add_conv(sd, m + ".7", bn=m + ".8")
            self._add_conv(sd, m + ".11", bn=m + ".12")
            self._add_osconv(sd, f"adapt.{g}.adapt")
        self._add_conv(sd, "conv_last")
        self.gamma = float(sd["gamma"].reshape(-1)[0])
        self._pack_satu(sd)
        self.se_gate = torch.empty(self.nf, device=self.dev)

    def clone_for_stream(self) -> "HipEngine":
        """A sibling engine that shares every read-only packed weight with this one but owns its
        dynamic state (OSConv scratch / weight images, buffers, SATU tables, graphs), so two clips can be
        in flight on two HIP streams."""
        e = HipEngine.__new__(HipEngine)
        e.lib, e.dev, e.cfg, e.nf = self.lib, self.dev, self.cfg, self.nf
        e.NB_MAX = self.NB_MAX
        e.pw, e.se, e._keep = self.pw, self.se, self._keep
        e.pw_wy, e.conv_wy, e.wy_min_tiles, e.wy_min_tiles_tp = self.pw_wy, self.conv_wy, self.wy_min_tiles, self.wy_min_tiles_tp
        e.reuse_buffers, e.osconv_fused = self.reuse_buffers, self.osconv_fused
        e.satu_t, e.satu_w, e.tail_w, e.tail_b, e.gamma, e.n_l2 = self.satu_t, self.satu_w, self.tail_w, self.tail_b, self.gamma, self.n_l2
        e.iter_win, e.fwd_idx, e.bwd_idx = self.iter_win, self.fwd_idx, self.bwd_idx
        e.satu_tail_t, e.satu_w_tail = self.satu_tail_t, self.satu_w_tail
        e.satu_tailq_t, e.satu_w_tailq, e.satu_q = self.satu_tailq_t, self.satu_w_tailq, self.satu_q
        e.nb, e._bstride, e.clip_batch, e.clip_batch_max_px = 1, {}, self.clip_batch, self.clip_batch_max_px
        e.form_nb = 1
        e.osc = {}
        for k, ent in self.osc.items():
            c = dict(ent)
            c.update(e._osc_scratch(ent["cin"], ent["cout"], ent["knum"], ent["nunits"] * 8))
            e.osc[k] = c
        e.se_gate = torch.empty_like(self.se_gate)
        e._init_caches()
        e.max_shapes, e.max_scales, e._budget, e._axes = self.max_shapes, self.max_scales, self._budget, self._axes
        e.satu_events, e.use_graphs, e.census, e._st = None, self.use_graphs, self.census, None
        e.capture_after, e.host_stats = self.capture_after, self.host_stats
        e.conv_algo = _lib.CONV_DIRECT
        e._hr_choice, e._hr_table = self._hr_choice, self._hr_table
        e._siblings, e._streams = [], []
        return e

    # ------------------------------------------------------------------ buffers / launch helpers
    # Device memory is cached per LR clip shape (the ~200 named channel-last feature maps) and, inside a shape, per scale
    # (HR-sized buffers, the captured hipGraphs with their static input / output).  Both levels are small LRU caches: an
    # arbitrary-scale sweep over many (shape, scale) pairs (BASELINE configs 3 / 5) keeps a bounded working set instead of
    # pinning every size it has ever seen (the reference frees everything per frame, video_base_model.py:72-74).
    def _init_caches(self):
        from collections import OrderedDict
        # Count caps (secondary: the byte budget below is what normally decides) ...
        self.max_shapes = max(1, int(os.environ.get("SAVSR_CACHE_SHAPES", "256")))
        self.max_scales = max(1, int(os.environ.get("SAVSR_CACHE_SCALES", "48")))
        # ... and the BYTE budget of everything this engine and its sibling engines (one per HIP stream) keep resident per (shape, scale):
        # arena chunks of the LR / HR buffers, the graphs' static input / output, per-pixel SATU tables.  SAVSR_CACHE_GB, default half of the
        # HBM that is free when the engine is built: a Vimeo-shaped stream (51 LR shapes x 3 streams x ~0.3 GB) stays resident, a 540x960
        # stream (5 GB per shape and stream) keeps what fits -- one knob for both instead of a shape count that suits one of them.
        gb = os.environ.get("SAVSR_CACHE_GB")
        if gb is not None:
            limit = int(float(gb) * (1 << 30))
        else:
            try:
                limit = int(0.5 * torch.cuda.mem_get_info(self.dev)[0])
            except RuntimeError:
                limit = 64 << 30
        self._budget = {"limit": max(limit, 1), "used": 0, "evictions": 0, "trim": False}    # shared with the sibling engines (clone_for_stream)
        self._ctx: "OrderedDict[tuple, dict]" = OrderedDict()
        self._axes: "OrderedDict[tuple, dict]" = OrderedDict()
        self._default_ctx = dict(bufs={}, scales=OrderedDict(), bytes=0, untracked=True)      # direct kernel-level calls (tests, tools) outside a forward
        self._default_sc = dict(bufs={}, graphs=None, chunk=0, bytes=0, untracked=True)
        self.hr_sched = torch.zeros(16, dtype=torch.int32, device=self.dev)     # tile-queue scratch of the SATU HR kernel (one per engine = per stream)
        self._cur, self._cur_sc = self._default_ctx, self._default_sc
        self._cur_key = None

    def _charge(self, owner: dict, nbytes: int) -> None:
        """Account `nbytes` of device memory to a (shape) or (shape, scale) context and to the shared budget."""
        owner["bytes"] = owner.get("bytes", 0) + int(nbytes)
        if not owner.get("untracked"):
            self._budget["used"] += int(nbytes)

    @staticmethod
    def _ctx_bytes(ctx: dict) -> int:
        return ctx.get("bytes", 0) + sum(sc.get("bytes", 0) for sc in ctx["scales"].values())

    def _drop(self, skey: tuple) -> None:
        """Evict one shape context of THIS engine.  Its graphs may still be replaying on this engine's stream: their memory came from the
        graphs' private pool, which the allocator hands to nobody else and returns to the device only through hipFree (device-synchronous,
        empty_cache() below or the allocator's own out-of-memory path) -- dropping the references is safe at any time."""
        ctx = self._ctx.pop(skey)
        self._budget["used"] -= self._ctx_bytes(ctx)
        self._budget["evictions"] += 1
        self._budget["trim"] = True

    def _drop_scale(self, ctx: dict, ckey: tuple) -> None:
        sc = ctx["scales"].pop(ckey)
        self._budget["used"] -= sc.get("bytes", 0)
        self._budget["evictions"] += 1
        self._budget["trim"] = True

    def _evict_to_budget(self, keep: tuple) -> None:
        """Least recently used shape contexts of this engine go until the shared budget holds (never the current one; a sibling's
        contexts are its own stream's business).  Then, inside the current shape, its least recently used scales."""
        b = self._budget
        while b["used"] > b["limit"] and len(self._ctx) > 1:
            victim = next(k for k in self._ctx if k != keep)
            self._drop(victim)
        cur = self._ctx.get(keep)
        while cur is not None and b["used"] > b["limit"] and len(cur["scales"]) > 1:
            self._drop_scale(cur, next(iter(cur["scales"])))

    def _select(self, shape: tuple, scale) -> dict:
        """Make (clip shape, scale) the current buffer context; evicts least recently used ones beyond the byte budget / the count caps."""
        skey = tuple(int(v) for v in shape)
        ctx = self._ctx.get(skey)
        fresh = ctx is None
        if fresh:
            from collections import OrderedDict
            ctx = dict(bufs={}, scales=OrderedDict(), bytes=0)
            self._ctx[skey] = ctx
            while len(self._ctx) > self.max_shapes:
                self._drop(next(iter(self._ctx)))
        else:
            self._ctx.move_to_end(skey)
        ckey = (float(scale[0]), float(scale[1]))
        sc = ctx["scales"].get(ckey)
        if sc is None:
            fresh = True
            sc = dict(bufs={}, graphs=None, chunk=0, bytes=0)        # (a scale context holds the HR-sized buffers: exact-size allocations)
            ctx["scales"][ckey] = sc
            while len(ctx["scales"]) > self.max_scales:
                self._drop_scale(ctx, next(iter(ctx["scales"])))
        else:
            ctx["scales"].move_to_end(ckey)
        if fresh:
            self._evict_to_budget(skey)
            if self._budget["trim"] and torch.cuda.memory_reserved(self.dev) > self._budget["limit"]:
                # evicted graphs' pools are only returned by an explicit trim (rare: the budget was exceeded AND the allocator holds more
                # than the budget); device-synchronous, so never with a capture under way
                if not torch.cuda.is_current_stream_capturing():
                    torch.cuda.empty_cache()
                    self._budget["trim"] = False
        self._cur, self._cur_sc, self._cur_key = ctx, sc, (skey, ckey)
        return sc

    def cache_stats(self) -> dict:
        """Resident contexts of THIS engine; `bytes` = device memory they hold (arena chunks + graph I/O + per-pixel tables), `budget_*` = the
        account shared with the sibling engines."""
        return {"shapes": len(self._ctx), "scales": sum(len(c["scales"]) for c in self._ctx.values()), "axes": len(self._axes),
                "bytes": sum(self._ctx_bytes(c) for c in self._ctx.values()),
                "budget_used": self._budget["used"], "budget_limit": self._budget["limit"], "evictions": self._budget["evictions"]}

    ARENA_CHUNK = 64 << 20      # bytes per arena chunk (larger requests get a chunk of their own)

    def _get_buf(self, owner: dict, name: str, shape: tuple) -> torch.Tensor:
        """Named fp32 buffer of a context, carved out of the context's ARENA: the ~110 feature maps of a clip shape are never
        freed one by one (the context is dropped as a whole), so they are bump-allocated from a few large device allocations
        instead of one allocator round trip each -- a new LR shape (every folder x scale of the YAML sweep is one) costs a
        handful of hipMallocs, not a hundred.  256-byte aligned (the kernels ask for 16)."""
        store = owner["bufs"]
        key = (name,) + tuple(shape)
        t = store.get(key)
        if t is None:
            n = 1
            for d in shape:
                n *= int(d)
            nbytes1 = (4 * n + 255) & ~255
            nbytes = nbytes1 * self.nb                    # (nb copies: clip b's lives nbytes1 * b further on)
            free = owner.get("free", {}).get(nbytes) if not owner.get("sealed") else None
            if free:
                raw = free.pop()                          # a slot whose previous owner's last reader is already enqueued (release())
            else:
                arena = owner.setdefault("arena", [])
                if not arena or arena[-1][1] + nbytes > arena[-1][0].numel():
                    # (nb clips per launch sequence: nb x the chunk, so that a batched context costs the same handful of allocations -- 26 64-MiB
                    # hipMallocs inside a capture were 40 ms of a 45 ms capture)
                    arena.append([torch.empty(max(nbytes, owner.get("chunk", self.ARENA_CHUNK * self.nb)), device=self.dev, dtype=torch.uint8), 0])
                    self._charge(owner, arena[-1][0].numel())
                chunk, off = arena[-1]
                raw = chunk[off:off + nbytes]
                arena[-1][1] = off + nbytes
            t = raw[:4 * n].view(torch.float32).view(shape)
            store[key] = t
            owner.setdefault("raw", {})[t.data_ptr()] = raw
            if self.nb > 1:
                self._bstride[t.data_ptr()] = nbytes1
            else:
                self._bstride.pop(t.data_ptr(), None)     # (an address an evicted batched context used to own)
        return t

    # Buffer liveness.  The launch sequence of a clip shape is static, so the assignment of named buffers to memory is decided ONCE, on the
    # context's first frame: release(x) there returns x's slot to a per-size free list (every reader of x has been enqueued on the one
    # stream of this engine, and the stream is in-order, so a later writer cannot overtake them), and the next new name of that size takes
    # it.  After the first frame the context is sealed: names keep their slots (captured hipGraphs hold the pointers), release() does
    # nothing, and a name first seen later gets fresh memory.  What is released, and where: the network pieces below.
    def release(self, *xs) -> None:
        owner = self._cur
        if owner.get("sealed") or not self.reuse_buffers:
            return
        raws = owner.get("raw", {})
        for x in xs:
            t = x.t if isinstance(x, Src) else x
            raw = raws.get(t.data_ptr()) if t is not None else None
            if raw is not None and not any(raw.data_ptr() == r.data_ptr() for r in owner.setdefault("free", {}).setdefault(raw.numel(), [])):
                owner["free"][raw.numel()].append(raw)

    def seal_buffers(self) -> None:
        """End of a context's first frame: the name -> memory assignment is final."""
        self._cur["sealed"] = True
        self._cur.pop("free", None)

    def _abort_frame(self) -> None:
        """A frame's launch sequence raised (allocation failure, a capture error, ...).  While a shape's buffer plan is still being made (first
        frame, not sealed) a partly consumed free list would hand live memory to the next new name on a retry -- the plan is all or nothing:
        the whole shape context goes.  A sealed shape keeps its plan; only the half-built (shape, scale) context is dropped."""
        if self._cur_key is None:
            return
        skey, ckey = self._cur_key
        ctx = self._ctx.get(skey)
        if ctx is not None:
            if not ctx.get("sealed"):
                self._drop(skey)
            elif ckey in ctx["scales"] and not ctx["scales"][ckey].get("graphs"):
                self._drop_scale(ctx, ckey)
        self._cur, self._cur_sc, self._cur_key = self._default_ctx, self._default_sc, None

    def buf(self, name: str, *shape: int) -> torch.Tensor:
        """Named LR-sized buffer of the current clip shape."""
        return self._get_buf(self._cur, name, shape)

    def sbuf(self, name: str, *shape: int) -> torch.Tensor:
        """Named buffer whose size depends on the scale (HR-sized), owned by the current (shape, scale) context."""
        return self._get_buf(self._cur_sc, name, shape)

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
                        c.wpacked, c.algo = c._wy, _lib.CONV_WINOGRAD_Y
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
            wy = int(d.algo) == _lib.CONV_WINOGRAD_Y
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
        return [(self.osc[k]["wdyn_wy"], None, self.osc[k]["cout"], self.osc[k]["cin"], 3, _lib.CONV_WINOGRAD_Y) if dsc.wy else
                (self.osc[k]["wdyn"], None, self.osc[k]["cout"], self.osc[k]["cin"], 3) for k, dsc in zip(keys, descs)]

    def osconv_weights(self, key: str, srcs: List[Src], h: int, w: int, scale, pooled: bool = False, wy: bool = False):
        return self.osconv_launch([key], [self.osconv_desc(key, srcs, h, w, scale, pooled, wy)])[0]

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
        forced = os.environ.get("SAVSR_HR_TILE")                                         # "rows,cols32": experiments only
        forced_v = os.environ.get("SAVSR_HR_VARIANT")
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
        ent["tail_plans"] = [t for v in (range(nvar) if not forced_v else [int(forced_v)]) for t in plans(True, v)]
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
        sched = self.hr_sched.data_ptr() if os.environ.get("SAVSR_HR_STATIC") != "1" else None
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
                if os.environ.get("SAVSR_HR_PRINT_PLANS"):     # diagnostics: what every feasible plan measured (us per launch)
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
            if os.environ.get("SAVSR_PROFILE_CAPTURE"):
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
        with torch.cuda.stream(cap):
            if pool is None:
                graph.capture_begin()
            else:
                graph.capture_begin(pool=pool)
            try:
                fn()
            finally:
                graph.capture_end()
        cur.wait_stream(cap)

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
        if not self.use_graphs or self.n_streams < 2:
            return [self.forward(lq.unsqueeze(0), sc)[0] for lq, sc in items]
        # (a lone clip takes the throughput flow too: what forward_many returns for a clip does not depend on how many came with it)
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
        ns = min(self.n_streams, len(units))
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
            ns = min(self.n_streams, len(units))
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
