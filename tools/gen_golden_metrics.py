"""Golden vectors for the PSNR half of row f3, produced by the REFERENCE's own functions (build container only).

    PYTHONDONTWRITEBYTECODE=1 python3 tools/gen_golden_metrics.py        # writes tests/golden/psnr_y.npz

Imports, by file path, lbasicsr/utils/color_util.py (bgr2ycbcr), lbasicsr/metrics/metric_util.py (to_y_channel,
reorder_image) and lbasicsr/metrics/psnr_ssim.py (calculate_psnr).  psnr_ssim.py has `import cv2` at module level and cv2
is not installed here: an EMPTY module named cv2 satisfies that import statement -- nothing on the PSNR path calls into
it.  calculate_ssim (cv2.getGaussianKernel / cv2.filter2D, psnr_ssim.py:187-197) and tensor2img (cv2.cvtColor,
img_util.py:77) DO call cv2 and therefore stay unpinned at the cv2 boundary; this file records that fact in the fixture.
Only data (inputs, expected outputs) is stored; no reference source text enters the repository.
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SAVSR_REFERENCE_ROOT", "/root/reference")


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def have_real_cv2() -> bool:
    """True when a REAL OpenCV is importable (not the empty import placeholder below)."""
    try:
        import cv2
        return hasattr(cv2, "filter2D") and hasattr(cv2, "getGaussianKernel") and hasattr(cv2, "cvtColor")
    except ImportError:
        return False


def load_reference_metrics():
    sys.dont_write_bytecode = True
    for name in ("lbasicsr", "lbasicsr.utils", "lbasicsr.metrics"):
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = []
            sys.modules[name] = pkg
    if "cv2" not in sys.modules and not have_real_cv2():
        sys.modules["cv2"] = types.ModuleType("cv2")          # import-only placeholder, see the module docstring
    _load("lbasicsr.utils.registry", "lbasicsr/utils/registry.py")
    cu = _load("lbasicsr.utils.color_util", "lbasicsr/utils/color_util.py")
    sys.modules["lbasicsr.utils"].bgr2ycbcr = cu.bgr2ycbcr
    mu = _load("lbasicsr.metrics.metric_util", "lbasicsr/metrics/metric_util.py")
    ps = _load("lbasicsr.metrics.psnr_ssim", "lbasicsr/metrics/psnr_ssim.py")
    return cu, mu, ps


def main():
    cu, mu, ps = load_reference_metrics()
    rs = np.random.RandomState(2024)
    out = {"note": np.array("PSNR-Y / to_y_channel / bgr2ycbcr from the reference's functions; SSIM and tensor2img unpinned (cv2)")}
    cases = []
    for i, (h, w) in enumerate([(24, 31), (17, 40), (33, 33), (12, 12)]):
        a = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        noise = rs.randint(-9, 10, (h, w, 3))
        b = np.clip(a.astype(np.int64) + noise, 0, 255).astype(np.uint8)
        if i == 3:
            b = a.copy()                                       # identical images -> inf
        out[f"img/{i}/a"], out[f"img/{i}/b"] = a, b
        out[f"img/{i}/y_a"] = mu.to_y_channel(a)               # float, [h, w, 1], range [0, 255] without rounding
        out[f"img/{i}/ycbcr_f32"] = cu.bgr2ycbcr(a.astype(np.float32) / 255.0, y_only=True)
        for crop in (0, 2):
            for ych in (True, False):
                out[f"psnr/{i}/{crop}/{int(ych)}"] = np.float64(ps.calculate_psnr(a, b, crop, input_order="HWC", test_y_channel=ych))
        out[f"psnr_chw/{i}"] = np.float64(ps.calculate_psnr(a.transpose(2, 0, 1), b.transpose(2, 0, 1), 1, input_order="CHW", test_y_channel=True))
        if have_real_cv2():
            # a box with OpenCV: the SSIM half and tensor2img get pinned too (tests/test_metrics_pinned.py picks the keys up)
            for crop in (0, 2):
                out[f"ssim/{i}/{crop}"] = np.float64(ps.calculate_ssim(a, b, crop, input_order="HWC", test_y_channel=True))
        cases.append(i)
    if have_real_cv2():
        import torch
        iu = _load("lbasicsr.utils.img_util", "lbasicsr/utils/img_util.py")
        t = torch.from_numpy(rs.uniform(-0.2, 1.2, (3, 19, 23)).astype(np.float32))
        out["tensor2img/in"] = t.numpy()
        out["tensor2img/out"] = iu.tensor2img(t)
    out["pinned_ssim"] = np.array(bool(have_real_cv2()))
    out["cases"] = np.array(cases)
    path = os.path.join(ROOT, "tests", "golden", "psnr_y.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
