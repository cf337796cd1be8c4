"""Row f3, PSNR half: the host restatement savsr_amd/metrics.py (the oracle of the GPU metric kernel) against vectors
produced by the REFERENCE's own calculate_psnr / to_y_channel / bgr2ycbcr (tools/gen_golden_metrics.py,
lbasicsr/metrics/psnr_ssim.py:11-48, metric_util.py:32-45, utils/color_util.py:59-65).
SSIM (psnr_ssim.py:172-200) and tensor2img's cvtColor stay UNPINNED at the cv2 boundary: known-answer tests only
(tests/test_host_logic.py)."""
import os

import numpy as np
import pytest

from savsr_amd import metrics as M

PATH = os.path.join(os.path.dirname(__file__), "golden", "psnr_y.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(PATH)


def test_to_y_channel_bit_exact(gold):
    for i in gold["cases"]:
        a = gold[f"img/{i}/a"]
        ref = gold[f"img/{i}/y_a"]
        got = M.to_y_channel(a)
        assert got.dtype == ref.dtype and got.shape == ref.shape
        assert np.array_equal(got, ref)
        assert np.array_equal(M.bgr2ycbcr_y(a.astype(np.float32) / 255.0), gold[f"img/{i}/ycbcr_f32"])


def test_psnr_matches_reference(gold):
    for i in gold["cases"]:
        a, b = gold[f"img/{i}/a"], gold[f"img/{i}/b"]
        for crop in (0, 2):
            for ych in (True, False):
                ref = float(gold[f"psnr/{i}/{crop}/{int(ych)}"])
                got = M.calculate_psnr(a, b, crop, test_y_channel=ych)
                assert (np.isinf(ref) and np.isinf(got)) or got == ref, (i, crop, ych, got, ref)
        ref = float(gold[f"psnr_chw/{i}"])
        got = M.calculate_psnr(a.transpose(2, 0, 1), b.transpose(2, 0, 1), 1, input_order="CHW", test_y_channel=True)
        assert (np.isinf(ref) and np.isinf(got)) or got == ref


def test_fixture_live_vs_reference(gold):
    """In the build container the fixture is re-derived from the reference and must be unchanged."""
    if not os.path.isfile("/root/reference/lbasicsr/metrics/psnr_ssim.py"):
        pytest.skip("reference tree not present")
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools"))
    import gen_golden_metrics as G
    _, mu, ps = G.load_reference_metrics()
    for i in gold["cases"]:
        a, b = gold[f"img/{i}/a"], gold[f"img/{i}/b"]
        assert np.array_equal(mu.to_y_channel(a), gold[f"img/{i}/y_a"])
        assert ps.calculate_psnr(a, b, 0, test_y_channel=True) == float(gold[f"psnr/{i}/0/1"])


def test_ssim_and_tensor2img_pins_if_the_fixture_has_them(gold):
    """tools/gen_golden_metrics.py adds calculate_ssim / tensor2img vectors from the reference when a real OpenCV is importable
    where it runs (never in the offline build container so far: then this test only records the fact)."""
    if "pinned_ssim" not in gold.files or not bool(gold["pinned_ssim"]):
        pytest.skip("fixture was generated without cv2: SSIM / tensor2img stay unpinned at the cv2 boundary")
    import torch
    for i in gold["cases"]:
        a, b = gold[f"img/{i}/a"], gold[f"img/{i}/b"]
        for crop in (0, 2):
            assert abs(M.calculate_ssim(a, b, crop, test_y_channel=True) - float(gold[f"ssim/{i}/{crop}"])) < 1e-12
    assert np.array_equal(M.tensor2img(torch.from_numpy(gold["tensor2img/in"])), gold["tensor2img/out"])


# ---- independent checks of the two unpinned pieces.  NOT pins to the reference: cv2 (getGaussianKernel, filter2D, cvtColor)
# ---- is absent offline, so these compare savsr_amd/metrics.py with a separately written textbook formulation (scipy) and
# ---- with hand-computed known answers of psnr_ssim.py:172-200 / img_util.py:66-90's documented behaviour. -----------------
def _textbook_ssim(x, y):
    """Wang et al. 2004 as psnr_ssim.py:172-200 states it: 11x11 Gaussian (sigma 1.5) window, 'valid' region, C1 = (0.01*255)^2,
    C2 = (0.03*255)^2 -- with the 2-D kernel and scipy's 2-D correlation instead of the product's separable passes."""
    from scipy.signal import correlate2d
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    g /= g.sum()
    win = np.outer(g, g)
    f = lambda a: correlate2d(a, win, mode="valid")
    mx, my = f(x), f(y)
    sx, sy, sxy = f(x * x) - mx * mx, f(y * y) - my * my, f(x * y) - mx * my
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    return (((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sx + sy + c2))).mean()


def test_ssim_against_independent_scipy_formulation(gold):
    for i in gold["cases"]:
        a, b = gold[f"img/{i}/a"], gold[f"img/{i}/b"]
        for crop in (0, 1):
            if min(a.shape[:2]) - 2 * crop < 11:                                 # smaller than the window: no 'valid' region
                continue
            ac = a[crop:a.shape[0] - crop, crop:a.shape[1] - crop]
            bc = b[crop:b.shape[0] - crop, crop:b.shape[1] - crop]
            ya, yb = M.to_y_channel(ac)[..., 0].astype(np.float64), M.to_y_channel(bc)[..., 0].astype(np.float64)
            assert abs(M.calculate_ssim(a, b, crop, test_y_channel=True) - _textbook_ssim(ya, yb)) < 1e-13
        want = np.mean([_textbook_ssim(a[..., c].astype(np.float64), b[..., c].astype(np.float64)) for c in range(3)])
        assert abs(M.calculate_ssim(a, b, 0, test_y_channel=False) - want) < 1e-13          # per-channel mean (psnr_ssim.py:126-128)
    a = gold["img/0/a"]
    assert M.calculate_ssim(a, a, 0, test_y_channel=True) == 1.0


def test_tensor2img_known_answers():
    """img_util.py:66-90: clamp to [0, 1], x255, round (half to even, numpy), uint8, RGB -> BGR, CHW -> HWC."""
    import torch
    t = torch.zeros(3, 2, 3)
    t[0] = torch.tensor([[-0.3, 0.0, 0.5 / 255], [1.5 / 255, 2.5 / 255, 1.7]])            # R: below range | 0 | ties 0.5, 1.5, 2.5 | above
    t[1] = torch.tensor([[1.0, 0.999, 0.5], [254.5 / 255, 0.25, 0.75]])                    # G
    t[2] = 0.2                                                                             # B: 51
    img = M.tensor2img(t)
    assert img.dtype == np.uint8 and img.shape == (2, 3, 3)
    assert np.array_equal(img[:, :, 2], np.round(np.clip(t[0].numpy(), 0, 1) * 255.0).astype(np.uint8))     # R lands in BGR slot 2
    assert img[0, 0, 2] == 0 and img[0, 1, 2] == 0 and img[1, 2, 2] == 255
    ties = (np.float32(0.5 / 255) * 255.0, np.float32(1.5 / 255) * 255.0, np.float32(2.5 / 255) * 255.0)
    assert [int(img[0, 2, 2]), int(img[1, 0, 2]), int(img[1, 1, 2])] == [int(np.round(v)) for v in ties]    # numpy round-half-even on the fp32 product
    assert img[0, 0, 1] == 255 and img[0, 1, 1] == 255 and img[0, 2, 1] == 128 and img[1, 1, 1] == 64 and img[1, 2, 1] == 191
    assert (img[:, :, 0] == 51).all()
    assert np.array_equal(M.tensor2img(t.unsqueeze(0)), img)                                # [1, 3, H, W] accepted
    assert M.tensor2img(t[:1]).shape == (2, 3)                                              # single channel -> HW
    assert np.array_equal(M.tensor2img(t, rgb2bgr=False)[:, :, 0], img[:, :, 2])
