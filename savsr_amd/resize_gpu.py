"""LR synthesis on the GPU: the step before the hot path.

Mirrors lbasicsr/data/data_util.py:371-420 `arbitrary_scale_downsample` (degradation 'BI', mode 'torch'):
torchvision T.Resize(size=(round(h / sh), round(w / sw)), BICUBIC, antialias=True) on float tensors, which is
ATen's separable `_upsample_bicubic2d_aa` (align_corners=False), and lbasicsr/data/transforms.py:31-69
`cal_step` / `as_mod_crop` (the integer crop applied to the ground truth before it).

The per-axis window / weight tables are built on the host in fp32 with ATen's formulas (HelperInterpBase::
_compute_indices_weights_aa: support = 2 * scale when down-scaling, centre = scale * (i + 0.5), cubic a = -0.5,
weights normalised by their fp32 sum) and cached per (input size, output size); the weighted gathers run in
csrc/resize.hip, width first, then height, like ATen.  Oracle of this row: torch's CPU implementation of the
same op (tests/test_gpu_resize.py), measured difference <= 4e-7 (fp32 re-association).
"""
from __future__ import annotations

from math import floor
from typing import Dict, Tuple, Union

import numpy as np
import torch

from . import _lib

_F = np.float32


def cal_step(scale: float) -> int:
    """transforms.py:31-45 -- the smallest k in (1, 2, 5, 10, 20, 50) with k * scale integral (|.| < 1e-3)."""
    for k in (1, 2, 5, 10, 20, 50):
        if abs(scale * k - round(scale * k)) < 0.001:
            return k
    raise ValueError(f"scale {scale} is not a multiple of 0.02")      # the reference falls off the chain with an UnboundLocalError


def as_mod_crop_hw(h: int, w: int, scale: Union[float, Tuple[float, float]]) -> Tuple[int, int]:
    """transforms.py:48-69 -- GT size after the arbitrary-scale mod crop (Python round, half to even)."""
    if not isinstance(scale, tuple):
        scale = (scale, scale)
    sh, sw = cal_step(scale[0]), cal_step(scale[1])
    return round(floor(h / sh / scale[0]) * sh * scale[0]), round(floor(w / sw / scale[1]) * sw * scale[1])


def _cubic_aa(x: np.float32) -> np.float32:
    a = _F(-0.5)
    x = _F(abs(x))
    if x < 1:
        return _F(((a + 2) * x - (a + 3)) * x * x + 1)
    if x < 2:
        return _F((((x - 5) * x + 8) * x - 4) * a)
    return _F(0)


def aa_tables(in_size: int, out_size: int):
    """(xmin[out], xsize[out], weights[out][max_taps]) of one axis, as ATen computes them (fp32)."""
    scale = _F(_F(in_size) / _F(out_size))
    support = _F(2.0) * scale if scale >= 1 else _F(2.0)
    invscale = _F(1.0) / scale if scale >= 1 else _F(1.0)
    max_taps = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    xsize = np.zeros(out_size, np.int32)
    wt = np.zeros((out_size, max_taps), np.float32)
    for i in range(out_size):
        center = _F(scale * _F(i + 0.5))
        lo = max(int(_F(center - support + _F(0.5))), 0)
        n = min(int(_F(center + support + _F(0.5))), in_size) - lo
        w = np.array([_cubic_aa(_F((_F(j + lo) - center + _F(0.5)) * invscale)) for j in range(n)], np.float32)
        tot = _F(0)
        for v in w:
            tot = _F(tot + v)
        if tot != 0:
            w = (w / tot).astype(np.float32)
        xmin[i], xsize[i] = lo, n
        wt[i, :n] = w
    return xmin, xsize, wt


_TABLES: Dict[tuple, tuple] = {}


def _device_tables(in_size: int, out_size: int, dev: torch.device):
    key = (in_size, out_size, str(dev))
    t = _TABLES.get(key)
    if t is None:
        xmin, xsize, wt = aa_tables(in_size, out_size)
        t = (torch.from_numpy(xmin).to(dev), torch.from_numpy(xsize).to(dev), torch.from_numpy(wt).to(dev), wt.shape[1])
        _TABLES[key] = t
    return t


def resize_bicubic_aa(x: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """[..., h, w] fp32 device tensor -> [..., size[0], size[1]] (F.interpolate(mode='bicubic', antialias=True,
    align_corners=False) semantics).  Asynchronous on the current stream."""
    if not x.is_cuda:
        raise RuntimeError("resize_bicubic_aa needs a device tensor (the CPU path is torch's own interpolate)")
    x = x.to(torch.float32).contiguous()
    h, w = x.shape[-2:]
    oh, ow = int(size[0]), int(size[1])
    planes = x.numel() // (h * w)
    lib = _lib.load()
    st = torch.cuda.current_stream(x.device).cuda_stream
    xm, xs, wt, mt = _device_tables(w, ow, x.device)
    tmp = torch.empty(planes, h, ow, device=x.device, dtype=torch.float32)
    _lib.check(lib.savsr_resize_aa_axis(x.data_ptr(), planes, h, w, 0, ow, xm.data_ptr(), xs.data_ptr(), wt.data_ptr(), mt,
                                        tmp.data_ptr(), st), "savsr_resize_aa_axis[w]")
    ym, ys, wy, mty = _device_tables(h, oh, x.device)
    out = torch.empty(*x.shape[:-2], oh, ow, device=x.device, dtype=torch.float32)
    _lib.check(lib.savsr_resize_aa_axis(tmp.data_ptr(), planes, h, ow, 1, oh, ym.data_ptr(), ys.data_ptr(), wy.data_ptr(), mty,
                                        out.data_ptr(), st), "savsr_resize_aa_axis[h]")
    return out


def arbitrary_scale_downsample(x: torch.Tensor, scale: Union[float, Tuple[float, float]]) -> torch.Tensor:
    """data_util.py:371-420 (degradation 'BI'): x [b, t, c, h, w] or [t, c, h, w] -> frames of size
    (round(h / sh), round(w / sw)); no crop, no re-quantisation, as in the reference."""
    sh, sw = scale if isinstance(scale, tuple) else (scale, scale)
    h, w = x.shape[-2:]
    return resize_bicubic_aa(x, (round(h / sh), round(w / sw)))
