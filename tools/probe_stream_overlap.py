"""What running several clips' launch sequences on several HIP streams adds beyond batching clips INTO the launches:
config-2 clips (7x3x180x320, x4) through ONE stream with nb = 1 / 2 / 3 clips per launch sequence, against the product's three streams
(`gpurun -- python3 tools/probe_stream_overlap.py`).  Timing only; results are those of the product path."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import savsr_amd
from savsr_amd.engine import get_hw
from savsr_amd.utils import synth

dev = torch.device("cuda:0")
net = savsr_amd.build_network(dict(type="SAVSR")).eval()
net.load_state_dict(synth.synth_state_dict(seed=0))
net = net.to(dev)
net.set_scale((4.0, 4.0))
eng = net.engine()
lq = torch.stack([synth.synth_clip(7, 3, 180, 320, seed=i)[0] for i in range(18)], 0).to(dev)
H, W = get_hw(180, 320, (4.0, 4.0))
out = torch.empty(18, 3, H, W, device=dev)


def timed(fn, secs=3.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        fn(); n += 18
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


for nb in (1, 2, 3):
    def one_stream(nb=nb):
        for i0 in range(0, 18, nb):
            if nb == 1:
                eng._forward_graphed(lq[i0], (4.0, 4.0), out[i0], throughput=True)
            else:
                eng._forward_graphed(lq[i0:i0 + nb], (4.0, 4.0), out[i0:i0 + nb], throughput=True)
    f = timed(one_stream)
    print(f"one stream, {nb} clip(s) per launch sequence: {f:.1f} clips/s = {f * H * W / 1e6:.1f} HR Mpixel/s", flush=True)
f = timed(lambda: net(lq))
print(f"product (3 streams x 3 clips per launch sequence): {f:.1f} clips/s = {f * H * W / 1e6:.1f} HR Mpixel/s", flush=True)
