"""Randomised parity sweep of the whole HIP forward against the CPU oracle (diagnostics; `gpurun -- python3 tools/fuzz_network.py --cases 24`).

Random LR sizes (odd and even, down to 4 x 5) and random scale pairs in [1.05, 4.3] -- symmetric, asymmetric, integer and awkward
fractions -- on key-seeded weights: output shape, max-abs error against the oracle, eager == captured == replayed bits."""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import savsr_oracle as O  # noqa: E402
from savsr_amd.archs.savsr_arch import SAVSR  # noqa: E402
from savsr_amd.utils import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=24)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-side", type=int, default=40)
    ap.add_argument("--throughput", action="store_true",
                    help="the throughput flow on medium frames (33..200 x 64..352): two clips of one (shape, scale) through forward_many -- Winograd-y launches with "
                         "strip tiles for the image's last rows wherever the launcher's rule takes them -- clip 0 against the oracle, repeat bitwise")
    a = ap.parse_args()
    rnd = random.Random(a.seed)
    sd = synth.synth_state_dict(seed=0)
    net = SAVSR()
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda:0").eval()
    torch.set_num_threads(min(16, os.cpu_count() or 8))     # (the oracle's small convs crawl with one thread per core of a 128-core host: minutes per case)
    worst = 0.0
    for k in range(a.cases):
        h, w = (rnd.randint(33, 200), rnd.randint(64, 352)) if a.throughput else (rnd.randint(4, a.max_side), rnd.randint(5, a.max_side))
        kind = rnd.choice(["sym", "asym", "int", "frac"])
        if kind == "sym":
            s = round(rnd.uniform(1.05, 4.3), 2)
            sc = (s, s)
        elif kind == "int":
            sc = (float(rnd.randint(2, 4)), float(rnd.randint(2, 4)))
        elif kind == "frac":
            sc = (rnd.choice([1.1, 1.25, 1.5, 1.75, 2.95, 3.05, 3.3, 3.9]), rnd.choice([1.4, 1.6, 2.4, 2.5, 3.75, 4.0]))
        else:
            sc = (round(rnd.uniform(1.05, 4.3), 3), round(rnd.uniform(1.05, 4.3), 3))
        lq = synth.synth_clip(7, 3, h, w, seed=100 + k)
        t0 = time.time()
        with torch.no_grad():
            ref = O.forward(sd, lq, sc)
        net.set_scale(sc)
        if a.throughput:
            lq2 = synth.synth_clip(7, 3, h, w, seed=5000 + k)
            items = [lq[0].to("cuda:0"), lq2[0].to("cuda:0")]
            with torch.no_grad():
                cap = net.forward_many(items, [sc, sc])[0][None].cpu()
                rep = net.forward_many(items, [sc, sc])[0][None].cpu()
                eager = net.forward_many(items[:1], [sc])[0][None].cpu()       # (a lone clip: same flow, same conv forms -> same bits)
        else:
            taps = {}
            eager = net(lq.to("cuda:0"), taps=taps).cpu()
            cap = net(lq.to("cuda:0")).cpu()
            rep = net(lq.to("cuda:0")).cpu()
        err = float((cap - ref).abs().max())
        worst = max(worst, err)
        ok = tuple(cap.shape) == tuple(ref.shape) and err < 5e-5 and torch.equal(cap, rep) and torch.equal(cap, eager) and bool(torch.isfinite(cap).all())
        print(f"case {k:3d}: {h}x{w} x{sc} -> {tuple(cap.shape[-2:])}  max-abs {err:.2e}  {'ok' if ok else 'FAIL'}  ({time.time() - t0:.1f} s)", flush=True)
        if not ok:
            sys.exit(1)
    print(f"{a.cases} cases, worst max-abs {worst:.2e}")


if __name__ == "__main__":
    main()
