"""BASELINE config 3 / 4 shapes: per-scale frame time and HR Mpixel/s on one GPU (developer tool).

    python3 tools/scale_sweep.py            # 180x320 LR clips, a sample of the YAML scale list
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import savsr_amd  # noqa: E402
from savsr_amd.engine import get_hw  # noqa: E402
from savsr_amd.utils import synth  # noqa: E402


def main():
    net = savsr_amd.build_network(dict(type="SAVSR")).eval()
    net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
    net.to("cuda")
    lq = synth.synth_clip(7, 3, 180, 320, seed=0).cuda()
    scales = [(1.1, 1.1), (1.5, 1.5), (2, 2), (2.5, 2.5), (3, 3), (3.7, 3.7), (4, 4), (1.5, 4), (3.5, 2), (2.95, 3.75)]
    print("scale -> HxW | ms/frame | HR Mpixel/s")
    for sc in scales:
        net.set_scale(sc)
        H, W = get_hw(180, 320, sc)
        for _ in range(3):
            out = net(lq)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            out = net(lq)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        assert tuple(out.shape) == (1, 3, H, W) and bool(torch.isfinite(out).all())
        print(f"{sc} -> {H}x{W} | {1e3 * dt:7.3f} | {H * W / dt / 1e6:7.2f}")


if __name__ == "__main__":
    main()
