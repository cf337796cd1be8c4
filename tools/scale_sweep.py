"""BASELINE config 3 / 4 shapes: per-scale frame time, HR Mpixel/s and the SATU / tail kernels timed alone
(HIP events on the launch stream), one JSON line per (shape, scale) -- developer tool; `bench.py --config 3/4/5`
prints the judged lines.

    python3 tools/scale_sweep.py                 # 180x320 LR clips, a spread of the YAML scale list + config-4 shapes
    python3 tools/scale_sweep.py --all           # all 30 symmetric scales x1.1 ... x4.0 of config 3
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import savsr_amd  # noqa: E402
from savsr_amd.engine import get_hw  # noqa: E402
from savsr_amd.utils import synth  # noqa: E402


def time_events(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters       # us


def measure(net, h, w, sc, frames=10):
    eng = net.engine()
    lq = synth.synth_clip(7, 3, h, w, seed=0).cuda()
    net.set_scale(sc)
    H, W = get_hw(h, w, sc)
    for _ in range(3):
        out = net(lq)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        out = net(lq)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    assert tuple(out.shape) == (1, 3, H, W) and bool(torch.isfinite(out).all())
    parts = eng.time_satu_parts(lq[0], sc, time_events)
    ax = eng.satu_axes(h, w, sc)
    rec = {"lr": [h, w], "scale": list(sc), "hr": [H, W], "ms_per_frame": round(1e3 * dt, 3), "hr_mpix_per_s": round(H * W / dt / 1e6, 2),
           "table_entries": ax["n_uh"] * ax["n_uw"], "n_uh": ax["n_uh"], "n_uw": ax["n_uw"]}
    rec.update({k: round(v, 1) for k, v in parts.items()})
    alg = 4 * 64 * (2 * h * w + H * W)
    rec["satu_frac_of_8TBs"] = round(alg / ((parts["satu_lr_us"] + parts["satu_hr_us"]) * 1e-6) / 8e12, 4)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true")
    a = ap.parse_args()
    net = savsr_amd.build_network(dict(type="SAVSR")).eval()
    net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
    net.to("cuda")
    if a.all:
        cases = [(180, 320, (k / 10, k / 10)) for k in range(11, 41)]
    else:
        cases = [(180, 320, s) for s in [(4, 4), (1.1, 1.1), (1.5, 1.5), (2, 2), (2.5, 2.5), (3, 3), (3.7, 3.7), (3.9, 3.9), (2.95, 3.75)]]
        cases += [(480, 318, (1.5, 4)), (204, 636, (3.5, 2)), (180, 320, (4, 4))]
    for h, w, sc in cases:
        print(json.dumps(measure(net, h, w, sc)), flush=True)


if __name__ == "__main__":
    main()
