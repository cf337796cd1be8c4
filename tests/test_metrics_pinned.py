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
