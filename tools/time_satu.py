"""SATU LR / HR launches of the product path timed separately (HIP events, each kernel looped on real-frame-shaped tensors).

    python3 tools/time_satu.py [--h 180 --w 320 --scale 4 4 --iters 50 --reps 3]
    SAVSR_LIB_PATH=savsr_amd/csrc/libsavsr_hip_exp_x.so python3 tools/time_satu.py      # an experiment build
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from savsr_amd import _lib  # noqa: E402
from savsr_amd import engine as E  # noqa: E402
from savsr_amd.archs.savsr_arch import SAVSR  # noqa: E402
from savsr_amd.utils import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=180)
    ap.add_argument("--w", type=int, default=320)
    ap.add_argument("--scale", type=float, nargs=2, default=[4, 4])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--warm-frames", type=int, default=0,
                    help="whole network frames run before every timed repetition: the SATU launches are then timed on a board in the state the frame leaves it in "
                         "(loaded clocks; an idle board's first ~100 launches run up to 10 %% slower) -- what rocprofv3's per-kernel AVERAGES need to mean the steady state")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    eng = E.HipEngine(synth.synth_state_dict(), SAVSR().cfg, dev)
    h, w, sc = a.h, a.w, tuple(a.scale)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(h, w, 64, generator=g).to(dev)
    st = (0.6 * torch.randn(h, w, 64, generator=g)).to(dev)
    H, W = E.get_hw(h, w, sc)
    plane = eng.hr_plane(H, W)
    p27 = torch.empty(_lib.TAIL_PLANES, plane, device=dev)
    q = eng.satu_q                                     # SAVSR_SATU_Q=0: the 27-plane form
    seam = torch.empty(eng.seam_floats(H, W), device=dev) if q else None
    lr = lambda: eng.satu_lr(eng.full(x), eng.full(st), w, h, w, tail_form=True, q=q)
    lrcat = lr()
    hr = (lambda: eng.satu_hr(lrcat, h, w, sc, p27, plane, tail_form=True, seam=seam)) if q else (lambda: eng.satu_hr(lrcat, h, w, sc, p27, plane, tail_form=True))
    hr()
    torch.cuda.synchronize()

    def t(fn):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / a.iters
    if hasattr(eng.lib, "savsr_debug_satu_stamps") and os.environ.get("SAVSR_LIB_PATH", "").endswith("diag.so"):
        import numpy as np
        # instrumented library: cycle stamps of waves 0 and 4 of every LR workgroup (satu_lr_stream_kernel)
        eng.lib.savsr_debug_satu_stamps(1)
        for _ in range(20):
            lr()
        torch.cuda.synchronize()
        nb = min(2 * ((w + 31) // 32) * ((h + 7) // 8), 2048)
        buf = (C.c_longlong * (8 * nb))()
        eng.lib.savsr_debug_read_satu_stamps(buf, nb)
        eng.lib.savsr_debug_satu_stamps(0)
        stt = np.array(buf[:], dtype=np.int64).reshape(nb // 2, 2, 8)
        for wv in (0, 1):
            m = np.median(stt[:, wv, :], axis=0)
            clk = np.median(stt[:, wv, 7] / np.maximum(stt[:, wv, 6] - stt[:, wv, 5], 1)) * 100.0
            print(f"LR stream kernel, wave {4 * wv}: cycles at [prologue end, cg0 loop end, tile swap, cg1 loop end, projections] = {m[:5].astype(int).tolist()}, "
                  f"total {int(m[7])} cycles, clock {clk:.0f} MHz; workgroup start spread {(stt[:, 0, 5].max() - stt[:, 0, 5].min()) / 100.0:.2f} us, "
                  f"first start -> last end {(stt[:, :, 6].max() - stt[:, 0, 5].min()) / 100.0:.2f} us")
    center = torch.rand(3, h, w, generator=g).to(dev)
    outb = torch.empty(3, H, W, device=dev)
    tb = torch.zeros(3, device=dev)
    st_ = torch.cuda.current_stream().cuda_stream
    if q:
        tail = lambda: _lib.check(eng.lib.savsr_tail_gather_q(p27.data_ptr(), plane, seam.data_ptr(), seam.numel(), tb.data_ptr(), center.data_ptr(), h, w, H, W, outb.data_ptr(), st_), "tail_q")
    else:
        tail = lambda: _lib.check(eng.lib.savsr_tail_gather(p27.data_ptr(), plane, tb.data_ptr(), center.data_ptr(), h, w, H, W, outb.data_ptr(), st_), "tail")
    warm_clip = synth.synth_clip(7, 3, h, w, seed=0).to(dev) if a.warm_frames else None
    for _ in range(a.reps):
        for _ in range(a.warm_frames):
            eng.forward(warm_clip, sc)
        tt, tht = t(tail), t(lambda: (hr(), tail()))
        print(f"tail_gather alone {tt:.1f} us   HR->tail pair {tht:.1f} us", flush=True)
        tl, th = t(lr), t(hr)
        tp = t(lambda: (lr(), hr()))          # the stage as the frame runs it: LR, then the dependent HR launch
        til = eng.satu_axes(h, w, sc)["tiling_tail"]
        print(f"{os.path.basename(_lib.LIB_PATH)} {h}x{w} x{sc}: LR {tl:.1f} us  HR {th:.1f} us  sum {tl + th:.1f} us  LR->HR pair {tp:.1f} us   "
              f"(HR plan: variant {til.variant}, {til.tile_rows} rows x {32 * til.tile_cols32} px, window {til.lr_rows} x {til.lr_cols})", flush=True)


if __name__ == "__main__":
    main()
