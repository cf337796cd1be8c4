#!/usr/bin/env python3
"""SAVSR hot-path benchmark on MI355X: HR Mpixels/s on synthetic 7x3x180x320 -> 720x1280 clips.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (SAVSR.forward, BASELINE config 2) over one synthetic clip
already resident in HBM = one 720x1280 output frame.  Clips are independent, so for N > 1 every
rank runs its own K clips (weak scaling, no data-path collective); the only collective is one
RCCL all_gather of the per-frame [PSNR-Y, checksum] rows, as the reference reduces its metric
tensor once per dataset (video_base_model.py:108-113).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

LR_H, LR_W, SCALE = 180, 320, (4, 4)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM bytes of one SATU stage from rocprofv3 PMC passes (FETCH_SIZE x 2 for 16-B/lane streaming reads on gfx950,
# WRITE_SIZE as read; separate passes): profiles/r01b_satu_pmc_traffic.csv (LR + HR kernels).  Config 2 only.
SATU_PMC_TRAFFIC_BYTES = 383112499


def effective_cpus():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sd, threads):
    """Oracle (CPU restatement of the reference path, kind='port') on a bounded sample of the same
    workload: whole 7x3x180x320 -> 720x1280 frames until ~12 s of CPU time are spent (>= 1 frame),
    after one untimed quarter-area warm-up that pages the weights in."""
    from oracle import savsr_oracle as O
    from savsr_amd.utils import synth
    torch.set_num_threads(threads)
    lq = synth.synth_clip(7, 3, LR_H, LR_W, seed=0)
    with torch.no_grad():
        O.forward(sd, lq[..., :90, :160].contiguous(), SCALE)
        n, t0 = 0, time.perf_counter()
        while True:
            out = O.forward(sd, lq, SCALE)
            n += 1
            dt = time.perf_counter() - t0
            if dt > 12.0 or n >= 8:
                break
    hr_px = n * out.shape[-1] * out.shape[-2]
    return {"value": round(hr_px / dt / 1e6, 5), "unit": "HR Mpixel/s", "cores": threads, "kind": "port",
            "sample": f"{n} frame(s) of the workload clip (7x3x180x320, x4 -> 720x1280) through oracle/savsr_oracle.py, {dt:.2f} s"}, out, lq


MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 matrix peak (sustained under load: ~1870, tools/micro/mfma_rate.hip)


def conv_roofline(eng, dev, iters=20):
    """The kernel that dominates GPU time (conv_bf16x3_kernel, ~80 % of a frame) on its most frequent launch
    geometry: one savsr_conv2d_batch of six 128->64 3x3 convs at 180x320 with bias, LeakyReLU and a residual
    (the ResidualBlock conv2 launches, savsr_arch.py:412-414; 20 of them per frame).  HIP events on the launch
    stream.  The frame's launches replay inside a hipGraph, which cannot hold timing events, so this times the
    same launch right after the timed region; the rocprofv3 summary of the bench command holds the in-graph
    average of this kernel (profiles/)."""
    from savsr_amd import engine as E
    from savsr_amd._lib import ACT_LRELU
    g = torch.Generator().manual_seed(1)
    n, cin, cout = 6, 128, 64
    keep, descs = [], []
    for _ in range(n):
        wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        weights = (E.pack_conv_weight(wt).to(dev), torch.randn(cout, generator=g).to(dev), cout, cin, 3)
        xs = [torch.randn(LR_H, LR_W, 64, generator=g).to(dev) for _ in range(2)]
        res, out = torch.randn(LR_H, LR_W, cout, generator=g).to(dev), torch.empty(LR_H, LR_W, cout, device=dev)
        keep.append((weights, xs, res, out))
        descs.append(eng.conv_desc("bench", [eng.full(x) for x in xs], eng.full(out), LR_H, LR_W, ACT_LRELU, 0.2,
                                   res1=eng.full(res), weights=weights))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        eng.conv_launch(descs)
    ev0.record()
    for _ in range(iters):
        eng.conv_launch(descs)
    ev1.record()
    torch.cuda.synchronize()
    sec = ev0.elapsed_time(ev1) / 1e3 / iters
    alg = 2.0 * n * cin * cout * 9 * LR_H * LR_W            # fp32-equivalent flops (2 x MACs)
    issued = 3.0 * alg                                      # split-bf16: three bf16 MFMA products per fp32 product
    return {"kernel": "conv_bf16x3_kernel<3,2,2> (6 x conv3x3 128->64 + bias + LeakyReLU + residual, 180x320)", "bound": "mfma",
            "achieved": round(issued / sec / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(issued / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            "traffic": None, "algorithmic_flops": alg, "fp32_equivalent_tflops": round(alg / sec / 1e12, 1), "avg_ms": round(1e3 * sec, 4),
            "note": "achieved counts the bf16 MFMA flops issued (3 per fp32-equivalent product); launch timed solo after the timed region"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--clips-per-step", type=int, default=3, help="independent clips per step (kept in flight on separate HIP streams)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("SAVSR_BENCH_FORCE_DIST"):     # (the env switch exercises the RCCL path with one rank)
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)    # RCCL over xGMI

    import savsr_amd
    from savsr_amd.metrics import calculate_psnr, tensor2img
    from savsr_amd.utils import synth

    sd = synth.synth_state_dict(seed=0)                   # random-init weights of the full architecture
    net = savsr_amd.build_network(dict(type="SAVSR", num_in_ch=3, num_feat=64, num_frame=7, slid_win=3, fusion_win=5,
                                       interval=0, w1_num_block=4, w2_num_block=2, n_resgroups=4, n_resblocks=8,
                                       center_frame_idx=None)).eval()
    net.load_state_dict(sd, strict=True)
    net.to(dev)
    net.set_scale(SCALE)
    eng = net.engine()
    H, W = LR_H * SCALE[0], LR_W * SCALE[1]

    # distinct clips per rank and per step, resident in HBM before the timed region
    cps = max(1, args.clips_per_step)
    n_clips = min(args.steps, 4)
    clips = [torch.cat([synth.synth_clip(7, 3, LR_H, LR_W, seed=100 * rank + cps * i + j) for j in range(cps)], 0).to(dev)
             for i in range(n_clips)]
    gt = synth.synth_gt(3, H, W, seed=0)

    for i in range(args.warmup):
        net(clips[i % n_clips])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()

    eng.satu_events = []          # HIP events on the launch stream around the SATU launches
    rows = torch.zeros(args.steps, 2, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for i in range(args.steps):
        out = net(clips[i % n_clips])
        rows[i, 0] = out.sum()    # device-side checksum; PSNR of the last frame is computed after timing
    if dist is not None:
        gathered = torch.empty(world * args.steps, 2, device=dev)
        dist.all_gather_into_tensor(gathered, rows)
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    satu_ms = [a.elapsed_time(b) for a, b in eng.satu_events]
    eng.satu_events = None
    # the same SATU launches once more with nothing else on the GPU (outside the timed region): how much of the
    # in-flight figure is contention from the other clips' kernels
    solo_ms = []
    if rank == 0:
        eng.satu_events = []
        for i in range(8):
            net(clips[0][:1])
        torch.cuda.synchronize()
        solo_ms = sorted(a.elapsed_time(b) for a, b in eng.satu_events)[:5]      # the 5 fastest of 8 (clock ramp after the multi-stream phase)
        eng.satu_events = None

    if rank == 0:
        hr_mpx = H * W / 1e6
        value = world * args.steps * cps * hr_mpx / elapsed
        satu_avg_s = (sum(satu_ms) / len(satu_ms)) / 1e3
        alg_bytes = 4 * 64 * (2 * LR_H * LR_W + H * W)        # read x, st once + write out once
        achieved = alg_bytes / satu_avg_s / 1e9
        psnr = calculate_psnr(tensor2img(out[0].cpu()), tensor2img(gt), 0, test_y_channel=True)
        line = {
            "metric": "HR Mpixels/sec (Vid4-shape x4, 7-frame window)", "value": round(value, 3), "unit": "HR Mpixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: synthetic 7x3x180x320 clips, scale x4 -> 720x1280, key-seeded random-init weights",
                       "frames_per_step": cps, "streams_per_gpu": min(cps, eng.n_streams), "parallelism": f"clip-sharded dp{world}"},
            "psnr_y_vs_synthetic_gt": round(float(psnr), 4),
            "roofline": {"kernel": "SATU = satu_lr_kernel + satu_hr_kernel (the phase table is evaluated once per size / scale / weights)", "bound": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": SATU_PMC_TRAFFIC_BYTES,
                         "algorithmic_bytes": alg_bytes, "avg_ms": round(1e3 * satu_avg_s, 4),
                         "note": "timed-region figure: the other in-flight clips' kernels share the GPU with these launches",
                         "solo_avg_ms": round(sum(solo_ms) / len(solo_ms), 4), "solo_frac": round(alg_bytes / (sum(solo_ms) / len(solo_ms) / 1e3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        line["roofline_conv"] = conv_roofline(eng, dev)
        if world == 1 and not args.no_cpu_baseline:
            threads = effective_cpus()
            cb, ref, lq_c = cpu_baseline(sd, threads)
            got = net(lq_c.to(dev))
            cb["gpu_vs_oracle_max_abs_on_sample"] = float((got.cpu() - ref).abs().max())
            line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
