"""Per-launch census of the conv kernel over one frame (developer tool, GPU only).

Wraps savsr_conv2d / savsr_conv2d_batch of one eager (graph-free) forward in HIP events and prints, per
launch geometry (batch, cin, cout, ksize, h, w), the launch count, mean duration and achieved
fp32-equivalent TFLOP/s, so the slow geometries can be told from the fast ones.
"""
import collections
import os
import sys

os.environ["SAVSR_GRAPHS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import savsr_amd  # noqa: E402
from savsr_amd import _lib  # noqa: E402
from savsr_amd.utils.synth import synth_clip, synth_state_dict  # noqa: E402


def main():
    net = savsr_amd.build_network(dict(type="SAVSR")).eval()
    net.load_state_dict(synth_state_dict(seed=0), strict=True)
    net.to("cuda")
    net.set_scale((4, 4))
    lq = synth_clip(7, 3, 180, 320, seed=0).cuda()
    net(lq)
    torch.cuda.synchronize()
    eng = net._engine
    log = []
    lib = eng.lib
    orig_batch, orig_one = lib.savsr_conv2d_batch, lib.savsr_conv2d

    class Wrap:
        def __init__(self, lib):
            self._lib = lib

        def __getattr__(self, k):
            return getattr(self._lib, k)

        def savsr_conv2d_batch(self, arr, n, st):
            d = arr[0]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = orig_batch(arr, n, st)
            e1.record()
            log.append(((n, d.cin, d.cout, d.ksize, d.h, d.w, bool(d.pool), bool(d.res1), bool(d.res2)), e0, e1))
            return rc

        def savsr_conv2d(self, dref, st):
            d = dref._obj
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = orig_one(dref, st)
            e1.record()
            log.append(((1, d.cin, d.cout, d.ksize, d.h, d.w, bool(d.pool), bool(d.res1), bool(d.res2)), e0, e1))
            return rc

    eng.lib = Wrap(lib)
    for _ in range(3):
        log.clear()
        net(lq)
        torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for k, e0, e1 in log:
        agg.setdefault(k, []).append(e0.elapsed_time(e1) * 1e3)
    tot = 0.0
    print("batch cin cout ks h w pool res1 res2 | launches  mean us  total us  TFLOP/s(fp32-equiv)")
    for k, v in agg.items():
        n, cin, cout, ks, h, w = k[:6]
        flop = 2.0 * n * cin * cout * ks * ks * h * w
        mean = sum(v) / len(v)
        tot += sum(v)
        print(k, "|", len(v), f"{mean:9.1f} {sum(v):10.1f} {flop / mean / 1e6:8.1f}")
    print(f"conv total {tot / 1e3:.2f} ms over {len(log)} launches")


if __name__ == "__main__":
    main()
