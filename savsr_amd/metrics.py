"""PSNR-Y / SSIM-Y and tensor->image quantisation with the reference's numerics.

Host-side (numpy, float64) restatement of what the reference's validation loop applies to every
output frame -- the step AFTER the hot path:
  tensor2img        lbasicsr/utils/img_util.py:38-94   (clamp, x255, round, uint8, RGB->BGR)
  to_y_channel      lbasicsr/metrics/metric_util.py:32-45 + utils/color_util.py:59-65 (BT.601 Y)
  calculate_psnr    lbasicsr/metrics/psnr_ssim.py:11-48
  calculate_ssim    lbasicsr/metrics/psnr_ssim.py:84-129,172-200 (11x11 Gaussian, sigma 1.5, valid)
cv2 is not available offline: the Gaussian window is restated as the normalised
exp(-(i-5)^2 / (2 * 1.5^2)) outer product and the 'valid' correlation is done separably in
float64 (the window is separable and symmetric; border mode is irrelevant because of the
[5:-5] crop) -- SSIM parity is therefore unpinned at the cv2 boundary (known-answer tests only).
"""
from __future__ import annotations

import numpy as np
import torch

from .registry import METRIC_REGISTRY


def tensor2img(tensor: torch.Tensor, rgb2bgr: bool = True, min_max=(0, 1)) -> np.ndarray:
    """[3|1, H, W] or [1, 3, H, W] float tensor in RGB -> uint8 HWC image in BGR."""
    t = tensor.squeeze(0).float().detach().cpu().clamp(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    if t.dim() == 3:
        img = t.numpy().transpose(1, 2, 0)
        if img.shape[2] == 1:
            img = np.squeeze(img, axis=2)
        elif rgb2bgr:
            img = img[:, :, ::-1]
    elif t.dim() == 2:
        img = t.numpy()
    else:
        raise TypeError(f"Only support 3D or 2D tensor here, got {t.dim()}D")
    return np.ascontiguousarray((img * 255.0).round().astype(np.uint8))


def bgr2ycbcr_y(img: np.ndarray) -> np.ndarray:
    """BT.601 luma of a BGR float32 image in [0, 1]; returns float32 in [0, 1]."""
    out = np.dot(img, [24.966, 128.553, 65.481]) + 16.0
    out /= 255.0
    return out.astype(np.float32)


def to_y_channel(img: np.ndarray) -> np.ndarray:
    img = img.astype(np.float32) / 255.0
    if img.ndim == 3 and img.shape[2] == 3:
        img = bgr2ycbcr_y(img)[..., None]
    return img * 255.0


def _prep(img, img2, crop_border, input_order, test_y_channel):
    assert img.shape == img2.shape, f"Image shapes are different: {img.shape}, {img2.shape}."
    if input_order not in ("HWC", "CHW"):
        raise ValueError(f'Wrong input_order {input_order}. Supported input_orders are "HWC" and "CHW"')

    def reorder(a):
        if a.ndim == 2:
            a = a[..., None]
        if input_order == "CHW":
            a = a.transpose(1, 2, 0)
        return a

    img, img2 = reorder(img), reorder(img2)
    if crop_border != 0:
        img = img[crop_border:-crop_border, crop_border:-crop_border, ...]
        img2 = img2[crop_border:-crop_border, crop_border:-crop_border, ...]
    if test_y_channel:
        img, img2 = to_y_channel(img), to_y_channel(img2)
    return img.astype(np.float64), img2.astype(np.float64)


@METRIC_REGISTRY.register()
def calculate_psnr(img, img2, crop_border, input_order="HWC", test_y_channel=False, **kwargs):
    img, img2 = _prep(img, img2, crop_border, input_order, test_y_channel)
    mse = np.mean((img - img2) ** 2)
    if mse == 0:
        return float("inf")
    return 10.0 * np.log10(255.0 * 255.0 / mse)


def _gauss11():
    x = np.arange(11, dtype=np.float64) - 5.0
    k = np.exp(-(x * x) / (2.0 * 1.5 * 1.5))
    return k / k.sum()


def _blur_valid(a: np.ndarray, k: np.ndarray) -> np.ndarray:
    n = len(k)
    h, w = a.shape
    tmp = np.zeros((h - n + 1, w), dtype=np.float64)
    for i in range(n):
        tmp += k[i] * a[i:i + h - n + 1, :]
    out = np.zeros((h - n + 1, w - n + 1), dtype=np.float64)
    for j in range(n):
        out += k[j] * tmp[:, j:j + w - n + 1]
    return out


def _ssim(img, img2):
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    k = _gauss11()
    mu1, mu2 = _blur_valid(img, k), _blur_valid(img2, k)
    mu1_sq, mu2_sq, mu12 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = _blur_valid(img ** 2, k) - mu1_sq
    s2 = _blur_valid(img2 ** 2, k) - mu2_sq
    s12 = _blur_valid(img * img2, k) - mu12
    m = ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    return m.mean()


@METRIC_REGISTRY.register()
def calculate_ssim(img, img2, crop_border, input_order="HWC", test_y_channel=False, **kwargs):
    img, img2 = _prep(img, img2, crop_border, input_order, test_y_channel)
    return float(np.array([_ssim(img[..., i], img2[..., i]) for i in range(img.shape[2])]).mean())
