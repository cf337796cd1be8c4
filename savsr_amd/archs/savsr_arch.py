"""MI355X-native SAVSR behind the reference's ARCH_REGISTRY / forward() surface.

Drop-in for the class registered as `SAVSR` at /root/reference/lbasicsr/archs/savsr_arch.py:574:
same constructor kwargs (:576-589, `network_g` of options/test/SAVSR/*.yml), `.set_scale()`
(:635-636), `__call__(lq[b,t,3,h,w]) -> [b,3,H,W]`, and a parameter tree whose 791
state_dict keys/shapes equal the reference's, so `savsr_best.pth` loads with strict=True.

What differs is everything below that surface.  The modules here only HOLD parameters (they
are the checkpoint schema); no torch operator of theirs ever runs.  `forward()` hands the
clip to `savsr_amd.engine.HipEngine`, which re-lays the weights out once and then drives the
hand-written gfx950 kernels of libsavsr_hip.so.  On a machine without that library, or with
the module on a CPU device, `forward()` raises -- there is no eager fallback.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from ..registry import ARCH_REGISTRY


class _Holder(nn.Module):
    """A parameter container; calling it is a bug (the HIP engine does the arithmetic)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} holds parameters only; run SAVSR.forward()")


def _seq(mods):
    return nn.Sequential(*mods)


class ScaleAttention(_Holder):
    """Parameters of the reference's ScaleAttention (savsr_arch.py:16-60)."""

    def __init__(self, cin, cout, ksize=3, knum=8, reduction=0.0625, min_channel=16):
        super().__init__()
        hidden = max(int(cin * reduction), min_channel)
        self.fc = nn.Conv2d(cin, hidden, 1, bias=False)
        self.bn = nn.BatchNorm2d(hidden)
        self.channel_fc = nn.Conv2d(hidden, cin, 1)
        self.filter_fc = nn.Conv2d(hidden, cout, 1)
        self.spatial_fc = nn.Conv2d(hidden, ksize * ksize, 1)
        self.kernel_fc = nn.Conv2d(hidden, knum, 1)
        for m in (self.fc, self.channel_fc, self.filter_fc, self.spatial_fc, self.kernel_fc):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.bias is not None:
                nn.init.zeros_(m.bias)


class OSConv2d(_Holder):
    """Kernel bank + attention + scale routing of OSConv (savsr_arch.py:99-134)."""

    def __init__(self, cin, cout, ksize=3, knum=8):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(knum, cout, cin, ksize, ksize))
        for k in range(knum):
            nn.init.kaiming_normal_(self.weight.data[k], mode="fan_out", nonlinearity="relu")
        self.attention = ScaleAttention(cin, cout, ksize, knum)
        self.scale_routing = _seq([nn.Linear(cin + 2, cin * 2), nn.ReLU(True), nn.Linear(cin * 2, cin), nn.ReLU(True)])


class OSAdapt(_Holder):
    """savsr_arch.py:186-208: mask branch (indices match the reference Sequential) + OSConv."""

    def __init__(self, ch, ratio=4):
        super().__init__()
        q = ch // ratio
        self.mask = _seq([
            nn.Conv2d(ch, q, 3, 1, 1), nn.BatchNorm2d(q), nn.ReLU(True), nn.AvgPool2d(2),
            nn.Conv2d(q, q, 3, 1, 1), nn.BatchNorm2d(q), nn.ReLU(True),
            nn.Conv2d(q, q, 3, 1, 1), nn.BatchNorm2d(q), nn.ReLU(True),
            nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False),
            nn.Conv2d(q, 1, 3, 1, 1), nn.BatchNorm2d(1), nn.Sigmoid()])
        self.adapt = OSConv2d(ch, ch)


class STAUpsample(_Holder):
    """savsr_arch.py:217-260."""

    def __init__(self, ch, num_experts=4, st_ksize=5):
        super().__init__()
        def expert(o, i):
            w = torch.empty(num_experts, o, i, 1, 1)
            for n in range(num_experts):
                nn.init.kaiming_uniform_(w[n], a=math.sqrt(5))
            return nn.Parameter(w)
        self.weight_compress = expert(ch // 8, ch)
        self.weight_expand = expert(ch, ch // 8)
        self.kernel_conv = _seq([nn.Conv2d(ch, ch * st_ksize ** 2, 1), nn.LeakyReLU(0.1, True)])
        self.body = _seq([nn.Conv2d(4, 64, 1), nn.ReLU(True), nn.Conv2d(64, 64, 1), nn.ReLU(True)])
        self.routing = _seq([nn.Conv2d(64, num_experts, 1), nn.Sigmoid()])
        self.offset = nn.Conv2d(64, 2, 1)
        self.st_offset = nn.Conv2d(64, 2, 1)
        self.fusion = nn.Conv2d(2 * ch, ch, 1)


class ResidualBlock(_Holder):
    """savsr_arch.py:379-397."""

    def __init__(self, nf, nfr, use_osconv):
        super().__init__()
        self.conv0 = _seq([nn.Conv2d(nf, nf, 3, 1, 1) for _ in range(nfr)])
        if use_osconv:
            self.osconv = OSConv2d(nf * nfr, nf)
        else:
            self.conv1 = nn.Conv2d(nf * nfr, nf, 1)
        self.conv2 = _seq([nn.Conv2d(2 * nf, nf, 3, 1, 1) for _ in range(nfr)])


class WindowUnit_l1(_Holder):
    """savsr_arch.py:418-442."""

    def __init__(self, cin, nf, win, nblock):
        super().__init__()
        self.conv_c = nn.Conv2d(cin, nf, 3, 1, 1)
        self.conv_sup = nn.Conv2d(cin * (win - 1), nf, 3, 1, 1)
        self.blocks = _seq([ResidualBlock(nf, 3, use_osconv=(i >= 1)) for i in range(nblock)])
        self.merge = nn.Conv2d(3 * nf, nf, 3, 1, 1)


class WindowUnit_l2(_Holder):
    """savsr_arch.py:467-483."""

    def __init__(self, nf, win, slid, nblock):
        super().__init__()
        self.conv_h = _seq([nn.Conv2d(2 * nf, nf, 3, 1, 1) for _ in range(win)])
        self.blocks = _seq([ResidualBlock(nf, slid, True) for _ in range(nblock)])
        self.merge = nn.Conv2d(slid * nf, 2 * nf, 3, 1, 1)


class ChannelAttention(_Holder):
    def __init__(self, nf, squeeze=16):
        super().__init__()
        self.attention = _seq([nn.AdaptiveAvgPool2d(1), nn.Conv2d(nf, nf // squeeze, 1), nn.ReLU(True),
                               nn.Conv2d(nf // squeeze, nf, 1), nn.Sigmoid()])


class RCAB(_Holder):
    def __init__(self, nf, squeeze=16):
        super().__init__()
        self.rcab = _seq([nn.Conv2d(nf, nf, 3, 1, 1), nn.ReLU(True), nn.Conv2d(nf, nf, 3, 1, 1), ChannelAttention(nf, squeeze)])


class ResidualGroup(_Holder):
    def __init__(self, nf, nblock, squeeze=16):
        super().__init__()
        self.residual_group = _seq([RCAB(nf, squeeze) for _ in range(nblock)])
        self.conv = nn.Conv2d(nf, nf, 3, 1, 1)


def iteration_window(num_frame: int, interval: int, center_frame_idx: int) -> int:
    """Frames each propagation direction iterates over (savsr_arch.py:597-604): all of them without frame sampling,
    center + 1 / + 2 (even / odd centre index) with it."""
    if interval == 0:
        return num_frame
    return center_frame_idx + 1 if center_frame_idx % 2 == 0 else center_frame_idx + 2


def frame_sample_indices(num_frame: int, interval: int):
    """SAVSR.frame_sample (savsr_arch.py:638-659) as index lists: (past -> future frames, future -> past frames).  Both hold
    the centre frame (num_frame // 2: the method computes its own, whatever center_frame_idx the module was given)."""
    index = list(range(num_frame))
    if interval == 0:
        return index, index
    c = num_frame // 2
    if c % 2 == 0:
        fwd = index[1::interval + 1]
        fwd.insert(c // 2, c)
        bwd = index[::interval + 1]
    else:
        fwd = index[::interval + 1]
        fwd.insert(c // 2 + 1, c)
        bwd = index[1::interval + 1]
        if len(fwd) != len(bwd):
            bwd.append(fwd[-1])
            bwd.insert(0, fwd[0])
    return fwd, bwd


@ARCH_REGISTRY.register()
class SAVSR(nn.Module):
    def __init__(self, num_in_ch=3, num_feat=64, num_frame=7, slid_win=3, fusion_win=5, interval=0, w1_num_block=4,
                 w2_num_block=2, n_resgroups=4, n_resblocks=8, downsample_scale=2, center_frame_idx=None):
        super().__init__()
        self.cfg = dict(num_in_ch=num_in_ch, num_feat=num_feat, num_frame=num_frame, slid_win=slid_win,
                        fusion_win=fusion_win, interval=interval, w1_num_block=w1_num_block, w2_num_block=w2_num_block,
                        n_resgroups=n_resgroups, n_resblocks=n_resblocks, downsample_scale=downsample_scale,
                        center_frame_idx=center_frame_idx)
        self.scale: Tuple[float, float] = (4, 4)
        self.center_frame_idx = num_frame // 2 if center_frame_idx is None else center_frame_idx
        self.num_frame, self.num_feat = num_frame, num_feat
        iter_win = iteration_window(num_frame, interval, self.center_frame_idx)
        self.iter_win, self.interval = iter_win, interval
        if iter_win < slid_win:
            raise ValueError("num_frame / interval leave fewer frames than the sliding window")
        if (iter_win - fusion_win + 1) // 2 > 1:
            # two pyramid levels: the reference constructs them (:616-618) but its forward fails (WindowUnit_l2 :488 reads
            # win_size inputs, the level above returns win_size - fusion_win + 1) -- nothing to be a drop-in for
            raise ValueError("more than one pyramid level (num_frame - fusion_win + 1 >= 4 without frame sampling): "
                             "the reference's own forward raises an IndexError for this configuration")
        self.f2p_win = WindowUnit_l1(num_in_ch, num_feat, slid_win, w1_num_block)
        self.p2f_win = WindowUnit_l1(num_in_ch, num_feat, slid_win, w1_num_block)
        self.h_win = _seq([WindowUnit_l2(num_feat, (iter_win - slid_win + 1) - 2 * i, fusion_win, w2_num_block)
                           for i in range((iter_win - fusion_win + 1) // 2)])
        self.h_win_act = nn.LeakyReLU(0.2, True)
        self.h_win_conv_h = nn.Conv2d(2 * num_feat, num_feat, 3, 1, 1)
        self.RG = nn.ModuleList([ResidualGroup(num_feat, n_resblocks) for _ in range(n_resgroups)])
        self.adapt = nn.ModuleList([OSAdapt(num_feat) for _ in range(n_resgroups)])
        self.gamma = nn.Parameter(torch.ones(1))
        self.conv_last = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.upsample = STAUpsample(num_feat)
        self.tail = nn.Conv2d(num_feat, num_in_ch, 3, 1, 1)
        self._engine = None
        self._engine_sig = None
        self._sig_tensors = None

    def set_scale(self, scale: Union[tuple, float, int]):
        """savsr_arch.py:635-636; a bare number means a symmetric scale."""
        if isinstance(scale, (int, float)):
            scale = (scale, scale)
        self.scale = tuple(scale)

    # ---- engine management ---------------------------------------------------------------
    def _apply(self, fn, *a, **k):          # .to() / .cuda() / .float() move the parameters
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def _signature(self):
        if self._engine is None or self._sig_tensors is None:
            self._sig_tensors = list(self.state_dict(keep_vars=True).values())
        return (str(self.gamma.device), sum(t._version for t in self._sig_tensors), self._sig_tensors[0].data_ptr())

    def engine(self):
        """The HipEngine for the current parameters (rebuilt when they move or change in place)."""
        sig = self._signature()
        if self._engine is None or sig != self._engine_sig:
            from ..engine import HipEngine
            self._engine = HipEngine(self.state_dict(), self.cfg, self.gamma.device)
            self._sig_tensors = list(self.state_dict(keep_vars=True).values())
            self._engine_sig = self._signature()
        return self._engine

    def forward_many(self, clips, scales):
        """Extension for mixed-scale streams (BASELINE config 5; the reference's flow would call set_scale + forward per clip):
        clips[i]: [t, c, h, w] on the GPU, scales[i]: (sh, sw) -> list of [c, H, W]; independent clips overlap on HIP streams."""
        if self.training:
            raise RuntimeError("savsr_amd.SAVSR implements the inference path only; call .eval() first")
        with torch.no_grad():
            return self.engine().forward_many(list(zip(clips, [tuple(s) if not isinstance(s, (int, float)) else (s, s) for s in scales])))

    def forward(self, x: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        if self.training:
            raise RuntimeError("savsr_amd.SAVSR implements the inference path only; call .eval() first")
        if x.dim() != 5:
            raise ValueError("expected lq of shape [b, t, c, h, w]")
        with torch.no_grad():
            return self.engine().forward(x, self.scale, taps)
