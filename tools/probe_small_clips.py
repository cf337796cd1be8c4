"""Where a stream of SMALL clips (BASELINE config 5) spends its time: clips/s of one (shape, scale) repeated through forward_many with 1 and 3 HIP
streams, and the host's enqueue cost per clip (`gpurun -- python3 tools/probe_small_clips.py`).  Round 5 reading (one MI355X): 64x112 x4 runs
3.82 ms per clip on one stream -- 330 dependent launches at ~11.6 us each: the scale-independent latency chains (OSConv weight generation 21 x 33 us,
SE gates, SATU) and ~1.7 us per kernel boundary, not arithmetic -- and 1.89 ms with three streams; the host needs 0.16-0.28 ms per clip (graph
replays), so the stream is GPU-dispatch-bound, not host-bound.  More than three streams do not help (config 5: 284 / 245 / 283 / 268 clips/s with
3 / 4 / 6 / 8 streams): a conv workgroup holds its CU's whole LDS (152 KB), so three concurrent launches of <= 96 workgroups each already occupy
the chip's LDS."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import savsr_amd
from savsr_amd.utils import synth
dev = torch.device("cuda:0")
sd = synth.synth_state_dict(seed=0)
import itertools
SIZES = [tuple(float(v) if '.' in v else int(v) for v in c.split(',')) for c in os.environ.get('PROBE_SIZES', '64,112,4.0;128,224,2.0;144,180,4.0;180,320,4.0').split(';')]
for ns, cb in ((3, 1), (3, 3), (3, 2)):
    os.environ["SAVSR_STREAMS"], os.environ["SAVSR_CLIP_BATCH"], os.environ["SAVSR_CLIP_BATCH_MAX_PX"] = str(ns), str(cb), str(1 << 30)
    net = savsr_amd.build_network(dict(type="SAVSR")).eval(); net.load_state_dict(sd); net = net.to(dev)
    for (h, w, s_) in SIZES:
        sc = (float(s_), float(s_))
        n_clips = 18
        clips = [synth.synth_clip(7, 3, h, w, seed=i)[0].to(dev) for i in range(n_clips)]
        for _ in range(3):
            net.forward_many(clips, [sc] * n_clips)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 1.0:
            net.forward_many(clips, [sc] * n_clips); n += n_clips
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"streams {ns} clip_batch {cb} {h}x{w} x{sc[0]}: {n / dt:.1f} clips/s = {1e3 * dt / n:.2f} ms per clip", flush=True)
