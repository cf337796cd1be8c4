"""PSNR-Y / SSIM-Y on the GPU against the numpy restatement of the reference's
metric code (savsr_amd/metrics.py = the oracle of this row).  Both run every value through the same sequence
of precisions (uint8 levels -> fp32 -> fp64 luma -> fp32 -> fp64 statistics); only the order of the two
fp64 reductions differs, so the tolerance is 1e-9 (dB / SSIM units), far inside the 1e-3 / 1e-4 bar."""
import math

import numpy as np
import pytest
import torch

from savsr_amd.metrics import calculate_psnr, calculate_ssim, tensor2img
from savsr_amd.utils import synth

pytestmark = pytest.mark.gpu


def _cpu(sr, gt, crop):
    a, b = tensor2img(sr), tensor2img(gt)
    return calculate_psnr(a, b, crop, test_y_channel=True), calculate_ssim(a, b, crop, test_y_channel=True)


@pytest.mark.parametrize("H,W,crop", [(64, 80, 0), (37, 53, 0), (48, 48, 4), (11, 11, 0), (27, 90, 2), (720, 1280, 0)])
def test_psnr_ssim_y_vs_numpy(H, W, crop):
    from savsr_amd.metrics_gpu import psnr_ssim_y
    g = np.random.RandomState(H * 7 + W)
    gt = synth.synth_gt(3, H, W, seed=H + W)
    # un-clamped "network output": GT + noise, with values below 0 and above 1 so that the clamp matters
    sr = gt + torch.from_numpy(g.normal(0, 0.05, (3, H, W)).astype(np.float32))
    p_ref, s_ref = _cpu(sr, gt, crop)
    got = psnr_ssim_y(sr.cuda(), gt.cuda(), crop).cpu()
    assert abs(float(got[0]) - p_ref) <= 1e-9, (float(got[0]), p_ref)
    assert abs(float(got[1]) - s_ref) <= 1e-9, (float(got[1]), s_ref)


@pytest.mark.parametrize("H,W,crop", [(64, 80, 0), (37, 53, 3), (11, 11, 0), (180, 320, 0)])
def test_psnr_ssim_rgb_vs_numpy(H, W, crop):
    """test_y_channel: false (psnr_ssim.py:12,85): the mean squared difference over H x W x 3 and the mean of the three colour
    planes' SSIM, against the numpy restatement on the quantised BGR images."""
    from savsr_amd.metrics_gpu import psnr_ssim_y
    g = np.random.RandomState(H * 5 + W)
    gt = synth.synth_gt(3, H, W, seed=H + 2 * W)
    sr = gt + torch.from_numpy(g.normal(0, 0.05, (3, H, W)).astype(np.float32))
    a, b = tensor2img(sr), tensor2img(gt)
    p_ref, s_ref = calculate_psnr(a, b, crop, test_y_channel=False), calculate_ssim(a, b, crop, test_y_channel=False)
    got = psnr_ssim_y(sr.cuda(), gt.cuda(), crop, test_y_channel=False).cpu()
    assert abs(float(got[0]) - p_ref) <= 1e-9, (float(got[0]), p_ref)
    assert abs(float(got[1]) - s_ref) <= 1e-9, (float(got[1]), s_ref)
    y = psnr_ssim_y(sr.cuda(), gt.cuda(), crop).cpu()
    assert abs(float(y[0]) - float(got[0])) > 1e-6                   # (not the luma metric)


def test_identical_images_and_batch_dim():
    from savsr_amd.metrics_gpu import psnr_ssim_y
    gt = synth.synth_gt(3, 40, 56, seed=3).cuda()
    got = psnr_ssim_y(gt[None], gt).cpu()
    assert math.isinf(float(got[0])) and float(got[0]) > 0           # calculate_psnr: mse == 0 -> inf
    assert abs(float(got[1]) - 1.0) <= 1e-12                         # known answer: identical images -> SSIM 1


def test_half_level_rounding_matches_numpy():
    """Values that land exactly on x.5 of the 255-level grid: numpy rounds half to even, so must the kernel."""
    from savsr_amd.metrics_gpu import psnr_ssim_y
    H, W = 16, 255
    lv = (torch.arange(W, dtype=torch.float32) + 0.5) / 255.0          # 0.5, 1.5, ... -> .round() half to even
    sr = lv.repeat(3, H, 1).contiguous()
    gt = synth.synth_gt(3, H, W, seed=9)
    p_ref, s_ref = _cpu(sr, gt, 0)
    got = psnr_ssim_y(sr.cuda(), gt.cuda(), 0).cpu()
    assert abs(float(got[0]) - p_ref) <= 1e-9 and abs(float(got[1]) - s_ref) <= 1e-9


def test_rejects_small_or_mismatched():
    from savsr_amd.metrics_gpu import psnr_ssim_y
    a = torch.zeros(3, 10, 40, device="cuda")
    with pytest.raises(ValueError):
        psnr_ssim_y(a, a)
    with pytest.raises(ValueError):
        psnr_ssim_y(torch.zeros(3, 20, 40, device="cuda"), torch.zeros(3, 20, 41, device="cuda"))
    with pytest.raises(RuntimeError):
        psnr_ssim_y(torch.zeros(3, 20, 40), torch.zeros(3, 20, 40))


def test_validate_folder_gpu_metrics_match_cpu_metrics(synth_sd):
    """harness.validate_folder with device metrics == the same outputs through the numpy metrics."""
    import savsr_amd
    from savsr_amd import harness
    net = savsr_amd.build_network(dict(type="SAVSR")).eval()
    net.load_state_dict(synth_sd, strict=True)
    net.to("cuda")
    n, h, w, sc = 5, 16, 20, (2, 2)
    lq = torch.cat([synth.synth_clip(1, 3, h, w, seed=50 + i)[0] for i in range(n)], 0)       # [n, 3, h, w]
    gts = [synth.synth_gt(3, 2 * h, 2 * w, seed=60 + i) for i in range(n)]
    rows = harness.validate_folder(net, lq, gts, sc, device=torch.device("cuda"))
    assert tuple(rows.shape) == (n, 2) and rows.dtype == torch.float64
    net.set_scale(sc)
    for i in range(n):
        win = lq[harness.window_indices(i, n, 7, "reflection")].unsqueeze(0).cuda()
        out = net(win)[0].cpu()
        p_ref, s_ref = _cpu(out, gts[i], 0)
        assert abs(float(rows[i, 0]) - p_ref) <= 1e-9 and abs(float(rows[i, 1]) - s_ref) <= 1e-9


def test_validate_folder_from_gt_all_on_gpu(synth_sd):
    """GT -> mod crop -> GPU LR synthesis -> SAVSR -> GPU metrics == the same chain with torch-CPU resize and numpy
    metrics (LR differs by fp32 re-association only: |dPSNR| <= 1e-3 dB, |dSSIM| <= 1e-4, the north-star tolerance)."""
    import torch.nn.functional as F
    import savsr_amd
    from savsr_amd import harness
    from savsr_amd.resize_gpu import as_mod_crop_hw
    net = savsr_amd.build_network(dict(type="SAVSR")).eval()
    net.load_state_dict(synth_sd, strict=True)
    net.to("cuda")
    n, sc = 5, (3.5, 2)
    gt = torch.stack([synth.synth_gt(3, 58, 44, seed=70 + i) for i in range(n)], 0)
    rows = harness.validate_folder_from_gt(net, gt, sc, device=torch.device("cuda"))
    H, W = as_mod_crop_hw(58, 44, sc)
    assert (H, W) == (56, 44)
    g = gt[..., :H, :W].contiguous()
    lq = F.interpolate(g, size=(round(H / sc[0]), round(W / sc[1])), mode="bicubic", align_corners=False, antialias=True)
    net.set_scale(sc)
    for i in range(n):
        out = net(lq[harness.window_indices(i, n, 7, "reflection")].unsqueeze(0).cuda())[0].cpu()
        p_ref, s_ref = _cpu(out, g[i], 0)
        assert abs(float(rows[i, 0]) - p_ref) <= 1e-3 and abs(float(rows[i, 1]) - s_ref) <= 1e-4

