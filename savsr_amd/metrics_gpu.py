"""PSNR-Y / SSIM-Y on the GPU with the reference's numerics.

Device-side counterpart of savsr_amd/metrics.py (which stays the oracle of this row): the validation
loop no longer copies every 720x1280 frame to the host and spends ~100 ms of numpy on it; the two
numbers are produced by two kernel launches and stay on the device until the per-dataset gather.
Reference lines: lbasicsr/utils/img_util.py:66-90, metrics/metric_util.py:32-45,
utils/color_util.py:59-65, metrics/psnr_ssim.py:42-48,172-200 (details in csrc/metrics.hip).
"""
from __future__ import annotations

import torch

from . import _lib


def psnr_ssim_y(sr: torch.Tensor, gt: torch.Tensor, crop_border: int = 0, out: torch.Tensor = None, test_y_channel: bool = True) -> torch.Tensor:
    """sr, gt: [3, H, W] (or [1, 3, H, W]) fp32 RGB on the same GPU, values nominally in [0, 1]
    (clamped and quantised to uint8 levels exactly as tensor2img does).  Returns a float64 device
    tensor [2] = (PSNR-Y, SSIM-Y); PSNR is inf for identical images.  Asynchronous on the current stream.
    test_y_channel=False: the metrics over the three colour planes instead of the luma (the YAML option of psnr_ssim.py:12,85)."""
    if sr.dim() == 4:
        sr = sr[0]
    if gt.dim() == 4:
        gt = gt[0]
    if sr.shape != gt.shape or sr.dim() != 3 or sr.shape[0] != 3:
        raise ValueError(f"Image shapes are different or not [3, H, W]: {tuple(sr.shape)}, {tuple(gt.shape)}")
    if not (sr.is_cuda and gt.is_cuda):
        raise RuntimeError("psnr_ssim_y needs device tensors (the CPU path is savsr_amd.metrics)")
    sr = sr.to(torch.float32).contiguous()
    gt = gt.to(torch.float32).contiguous()
    _, H, W = sr.shape
    lib = _lib.load()
    nblk = lib.savsr_metrics_blocks(H, W, crop_border)
    if nblk < 1:
        raise ValueError("the cropped image must be at least 11 x 11")
    partial = torch.empty(nblk * 2 * (1 if test_y_channel else 3), dtype=torch.float64, device=sr.device)
    if out is None:
        out = torch.empty(2, dtype=torch.float64, device=sr.device)
    _lib.check(lib.savsr_metrics_psnr_ssim(sr.data_ptr(), H * W, gt.data_ptr(), H * W, H, W, crop_border, int(bool(test_y_channel)), partial.data_ptr(),
                                           out.data_ptr(), torch.cuda.current_stream(sr.device).cuda_stream), "savsr_metrics_psnr_ssim")
    return out
