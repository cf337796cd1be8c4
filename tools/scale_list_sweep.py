"""Every scale pair the reference's own lists name, at its working LR size, HIP forward against the CPU oracle (test infrastructure: the
oracle is the checker, as in tools/fuzz_network.py).

  * the 42 datasets of options/test/SAVSR/test_SAVSR_Vid4_asBI.yml (30 symmetric + 12 asymmetric pairs) at LR 180 x 320;
  * the 60 training pairs of lbasicsr/data/vimeo90k_dataset.py:178-203 at the LR size the Vimeo90K GT (256 x 448) gives them.

Per pair: output shape (reference get_HW), max-abs, |dPSNR-Y| and |dSSIM-Y| against one synthetic GT (north_star: 1e-3 dB / 1e-4), bitwise
rerun.  `gpurun -- python3 tools/scale_list_sweep.py > profiles/r05_scale_lists_vs_oracle.log` (about 5 CPU-minutes on the box's 16 cores).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import savsr_oracle as O  # noqa: E402
from savsr_amd.archs.savsr_arch import SAVSR  # noqa: E402
from savsr_amd.engine import get_hw  # noqa: E402
from savsr_amd.metrics import calculate_psnr, calculate_ssim, tensor2img  # noqa: E402
from savsr_amd.utils import synth, workloads  # noqa: E402
from savsr_amd.utils.host import cpu_model, effective_cpus  # noqa: E402

TOL, TOL_PSNR, TOL_SSIM = 5e-5, 1e-3, 1e-4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lists", default="yaml,train")
    ap.add_argument("--limit", type=int, default=0, help="first N pairs of each list (smoke runs)")
    a = ap.parse_args()
    torch.set_num_threads(effective_cpus())
    sd = synth.synth_state_dict(seed=0)
    net = SAVSR()
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda:0").eval()
    cases = []
    if "yaml" in a.lists:
        ys = workloads.YAML_SCALES[: a.limit or None]
        cases += [("yaml", 180, 320, sc) for sc in ys]
    if "train" in a.lists:
        ts = workloads.TRAIN_SCALES[: a.limit or None]
        cases += [("train",) + workloads.lr_shape(workloads.VIMEO_GT, sc) + (sc,) for sc in ts]
    print(f"# {len(cases)} (list, LR size, scale) cases; oracle on {effective_cpus()} threads of {cpu_model()}; tolerances max-abs {TOL}, dPSNR-Y {TOL_PSNR} dB, dSSIM-Y {TOL_SSIM}")
    worst = dict(err=0.0, dp=0.0, ds=0.0)
    bad = 0
    t_all = time.time()
    for name, h, w, sc in cases:
        lq = synth.synth_clip(7, 3, h, w, seed=h * 1000 + w)
        H, W = get_hw(h, w, sc)
        t0 = time.time()
        with torch.no_grad():
            ref = O.forward(sd, lq, sc)
        t_or = time.time() - t0
        net.set_scale(sc)
        out = net(lq.to("cuda:0")).cpu()
        again = net(lq.to("cuda:0")).cpu()
        gt = tensor2img(synth.synth_gt(3, H, W, seed=7))
        a_img, r_img = tensor2img(out[0]), tensor2img(ref[0])
        err = float((out - ref).abs().max())
        dp = abs(calculate_psnr(a_img, gt, 0, test_y_channel=True) - calculate_psnr(r_img, gt, 0, test_y_channel=True))
        ds = abs(calculate_ssim(a_img, gt, 0, test_y_channel=True) - calculate_ssim(r_img, gt, 0, test_y_channel=True))
        ok = tuple(out.shape) == (1, 3, H, W) == tuple(ref.shape) and err < TOL and dp <= TOL_PSNR and ds <= TOL_SSIM and torch.equal(out, again) and bool(torch.isfinite(out).all())
        worst = dict(err=max(worst["err"], err), dp=max(worst["dp"], dp), ds=max(worst["ds"], ds))
        bad += 0 if ok else 1
        print(f"{name:5s} {h:3d}x{w:3d} x({sc[0]:g}, {sc[1]:g}) -> {H}x{W}: max-abs {err:.2e}  dPSNR-Y {dp:.1e} dB  dSSIM-Y {ds:.1e}  {'ok' if ok else 'FAIL'}  (oracle {t_or:.1f} s)", flush=True)
    print(f"# {len(cases)} cases, {bad} failed; worst max-abs {worst['err']:.2e}, worst dPSNR-Y {worst['dp']:.1e} dB, worst dSSIM-Y {worst['ds']:.1e}; {time.time() - t_all:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
