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


def core_tables(in_size: int, out_size: int):
    """(xmin[out], xsize[out], weights[out][max_taps]) of one axis of the reference's second bicubic implementation,
    lbasicsr/data/core.py::resize_1d (:276-345; MATLAB-style, `downsampling_mode: core`, data_util.py:411-412), or None for an
    identity axis (:295-296).  Positions / weights in fp32 with torch CPU ops as core.py evaluates them: kernel_size =
    ceil(4 / scale) + 2 when shrinking, pos = (i + 0.5) / scale - 0.5, base = floor(pos) - kernel_size // 2 + 1, weight_k =
    cubic((pos - base - k) * scale) normalised over k (get_weight :180-200).  Its padding repeats the border pixel (index -1 -> 0,
    -2 -> 1, n -> n - 1: reflect_padding :105-137), so the taps that fall outside fold back onto pixels inside: their weights are
    added to those pixels' (one contiguous window per output; only the summation order differs from core.py's, ~1e-7)."""
    scale = out_size / in_size                                # imresize :425-426 (Python floats)
    if scale == 1:
        return None
    ksize = 4
    aa = 1.0
    if scale < 1:
        aa = scale
        ksize = int(np.ceil(ksize / aa))
    ksize += 2
    pos = torch.linspace(0, out_size - 1, steps=out_size, dtype=torch.float32)
    pos = (pos + 0.5) / scale - 0.5
    base = pos.floor() - (ksize // 2) + 1
    dist = pos - base
    buf = torch.stack([dist - k for k in range(ksize)], 0) * aa                # [ksize][out]
    ax = buf.abs()
    ax2 = ax * ax
    ax3 = ax * ax2
    a = -0.5
    w01 = ((a + 2) * ax3 - (a + 3) * ax2 + 1) * ax.le(1).to(torch.float32)
    w12 = ((a * ax3) - (5 * a * ax2) + (8 * a * ax) - (4 * a)) * torch.logical_and(ax.gt(1), ax.le(2)).to(torch.float32)
    wt = w01 + w12
    wt = (wt / wt.sum(dim=0, keepdim=True)).numpy()                            # [ksize][out]
    base = base.long().numpy()
    xmin = np.zeros(out_size, np.int32)
    xsize = np.zeros(out_size, np.int32)
    out = np.zeros((out_size, ksize), np.float32)
    for i in range(out_size):
        idx = base[i] + np.arange(ksize)
        idx = np.where(idx < 0, -idx - 1, idx)
        idx = np.where(idx >= in_size, 2 * in_size - 1 - idx, idx)
        if idx.min() < 0 or idx.max() >= in_size:
            raise ValueError("core resize: the kernel reaches more than one image size beyond the border")
        lo, hi = int(idx.min()), int(idx.max())
        xmin[i], xsize[i] = lo, hi - lo + 1
        for k in range(ksize):
            out[i, idx[k] - lo] = _F(out[i, idx[k] - lo] + wt[k, i])
    return xmin, xsize, out


def _device_core_tables(in_size: int, out_size: int, dev: torch.device):
    key = ("core", in_size, out_size, str(dev))
    t = _TABLES.get(key)
    if t is None:
        tb = core_tables(in_size, out_size)
        t = None if tb is None else (torch.from_numpy(tb[0]).to(dev), torch.from_numpy(tb[1]).to(dev), torch.from_numpy(tb[2]).to(dev), tb[2].shape[1])
        _TABLES[key] = t if t is not None else "identity"
    return None if isinstance(t, str) else t


def imresize_core(x: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """lbasicsr/data/core.py::imresize(x, sizes=size) (cubic, antialiasing, reflect padding) on the GPU: height first, then width
    (:438-439), each axis through the same weighted-gather kernel as the ATen-style resize, with core.py's tables."""
    if not x.is_cuda:
        raise RuntimeError("imresize_core needs a device tensor")
    x = x.to(torch.float32).contiguous()
    h, w = x.shape[-2:]
    oh, ow = int(size[0]), int(size[1])
    planes = x.numel() // (h * w)
    lib = _lib.load()
    st = torch.cuda.current_stream(x.device).cuda_stream
    cur, ch = x, h
    ty = _device_core_tables(h, oh, x.device)
    if ty is not None:
        tmp = torch.empty(planes, oh, w, device=x.device, dtype=torch.float32)
        _lib.check(lib.savsr_resize_aa_axis(cur.data_ptr(), planes, h, w, 1, oh, ty[0].data_ptr(), ty[1].data_ptr(), ty[2].data_ptr(), ty[3],
                                            tmp.data_ptr(), st), "savsr_resize_aa_axis[core h]")
        cur, ch = tmp, oh
    tx = _device_core_tables(w, ow, x.device)
    if tx is not None:
        out = torch.empty(planes, ch, ow, device=x.device, dtype=torch.float32)
        _lib.check(lib.savsr_resize_aa_axis(cur.data_ptr(), planes, ch, w, 0, ow, tx[0].data_ptr(), tx[1].data_ptr(), tx[2].data_ptr(), tx[3],
                                            out.data_ptr(), st), "savsr_resize_aa_axis[core w]")
        cur = out
    return cur.reshape(*x.shape[:-2], oh if ty is not None else h, ow if tx is not None else w)


def arbitrary_scale_downsample(x: torch.Tensor, scale: Union[float, Tuple[float, float]], mode: str = "torch") -> torch.Tensor:
    """data_util.py:371-420 (degradation 'BI'): x [b, t, c, h, w] or [t, c, h, w] -> frames of size
    (round(h / sh), round(w / sw)); no crop, no re-quantisation, as in the reference.  mode 'torch': T.Resize(BICUBIC,
    antialias=True) (:408-410, every shipped YAML); 'core': lbasicsr/data/core.py::imresize (:411-412)."""
    sh, sw = scale if isinstance(scale, tuple) else (scale, scale)
    h, w = x.shape[-2:]
    size = (round(h / sh), round(w / sw))
    if mode == "torch":
        return resize_bicubic_aa(x, size)
    if mode == "core":
        return imresize_core(x, size)
    raise ValueError(f"downsampling_mode '{mode}' (data_util.py:408-412 knows 'torch' and 'core')")
