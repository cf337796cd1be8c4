"""Micro-benchmarks of single kernels on the GPU box (used under rocprofv3 for PMC passes).

    python3 tools/bench_kernels.py conv 64 64 3 --iters 20
    python3 tools/bench_kernels.py satu --iters 20
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from savsr_amd import engine as E  # noqa: E402
from savsr_amd._lib import ACT_LRELU  # noqa: E402
from savsr_amd.archs.savsr_arch import SAVSR  # noqa: E402
from savsr_amd.utils import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["conv", "satu", "tail"])
    ap.add_argument("cin", type=int, nargs="?", default=64)
    ap.add_argument("cout", type=int, nargs="?", default=64)
    ap.add_argument("ks", type=int, nargs="?", default=3)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--h", type=int, default=180)
    ap.add_argument("--w", type=int, default=320)
    ap.add_argument("--batch", type=int, default=1, help="conv: convs per launch (savsr_conv2d_batch)")
    ap.add_argument("--cycles", action="store_true", help="conv / satu, instrumented library (libsavsr_hip_diag.so; the CONV_EXP / LR_EXP stamp knobs are archived under tools/experiments/): per-workgroup s_memtime totals")
    ap.add_argument("--stamps", action="store_true", help="conv / satu: print per-workgroup section timings (s_memtime); needs the instrumented library: SAVSR_DIAG=1 bash savsr_amd/csrc/build.sh, SAVSR_LIB_PATH=savsr_amd/csrc/libsavsr_hip_diag.so")
    ap.add_argument("--distinct", action="store_true", help="conv: every conv of the batch gets its own inputs and weights")
    ap.add_argument("--wy", action="store_true", help="conv (3x3, cout % 64 == 0): the Winograd-y form (SAVSR_CONV_WINOGRAD_Y) instead of the direct kernel")
    ap.add_argument("--wy-tp", action="store_true", help="with --wy: algo SAVSR_CONV_WINOGRAD_Y_THROUGHPUT (strip tiles for the image's last rows whenever <= 2 row pairs are left)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    eng = E.HipEngine(synth.synth_state_dict(), SAVSR().cfg, dev)
    h, w = a.h, a.w
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if a.what == "conv":
        g = torch.Generator().manual_seed(0)
        wt = torch.randn(a.cout, a.cin, a.ks, a.ks, generator=g) / (a.cin * a.ks * a.ks) ** 0.5
        bias = torch.randn(a.cout, generator=g).to(dev)
        from savsr_amd import _lib as L
        mk = (lambda wt_: (E.pack_conv_weight_wy(wt_).to(dev), bias, a.cout, a.cin, a.ks, L.CONV_WINOGRAD_Y_THROUGHPUT if a.wy_tp else L.CONV_WINOGRAD_Y)) if a.wy else \
            (lambda wt_: (E.pack_conv_weight(wt_).to(dev), bias, a.cout, a.cin, a.ks))
        nsrc = max(1, a.cin // 64)
        nset = a.batch if a.distinct else 1
        wsets = [mk(wt if k == 0 else torch.randn(a.cout, a.cin, a.ks, a.ks, generator=g) / (a.cin * a.ks * a.ks) ** 0.5) for k in range(nset)]
        xsets = [[torch.randn(h, w, a.cin // nsrc, generator=g).to(dev) for _ in range(nsrc)] for _ in range(nset)]
        outs = [torch.empty(h, w, a.cout, device=dev) for _ in range(a.batch)]
        descs = [eng.conv_desc("bench", [eng.full(x) for x in xsets[k % nset]], eng.full(o), h, w, ACT_LRELU, 0.2, weights=wsets[k % nset])
                 for k, o in enumerate(outs)]
        run = lambda: eng.conv_launch(descs)
        flop = 2.0 * a.batch * a.cin * a.cout * a.ks * a.ks * h * w
    elif a.what == "satu":
        g = torch.Generator().manual_seed(0)
        x, st = torch.randn(h, w, 64, generator=g).to(dev), torch.randn(h, w, 64, generator=g).to(dev)
        H, W = E.get_hw(h, w, (4, 4))
        plane = eng.hr_plane(H, W)
        out = torch.empty(27, plane, device=dev)          # the product (tail-projected) form: 27 planes
        run = lambda: eng.satu_hr(eng.satu_lr(eng.full(x), eng.full(st), w, h, w, tail_form=True), h, w, (4, 4), out, plane, tail_form=True)
        flop = 0.0
    else:
        from savsr_amd import _lib
        H, W = 4 * h, 4 * w
        plane = eng.hr_plane(H, W)
        feat = torch.randn(27, plane, device=dev)
        center = torch.rand(3, h, w, device=dev)
        out = torch.empty(3, H, W, device=dev)
        run = lambda: _lib.check(eng.lib.savsr_tail_gather(feat.data_ptr(), plane, eng.tail_b.data_ptr(),
                                                           center.data_ptr(), h, w, H, W, out.data_ptr(), eng._stream()), "tail")
        flop = 0.0
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(a.iters):
        run()
    ev1.record()
    torch.cuda.synchronize()
    us = 1e3 * ev0.elapsed_time(ev1) / a.iters
    if a.cycles and a.what == "conv":
        import ctypes as C
        import numpy as np
        nb = min(256, a.batch * ((w + 31) // 32) * ((h + 7) // 8))
        buf = (C.c_longlong * (6 * nb))()
        eng.lib.savsr_debug_read_conv_stamps(buf, nb)
        st = np.array(buf[:], dtype=np.int64).reshape(nb, 6)
        tot = st[:, 4] - st[:, 0]
        print("workgroup s_memtime totals: median %d  max %d  -> %.2f GHz against the event time" % (np.median(tot), tot.max(), tot.max() / us / 1e3))
    if a.stamps and a.what == "conv":
        import ctypes as C
        import numpy as np
        eng.lib.savsr_debug_conv_stamps(1)
        run()
        torch.cuda.synchronize()
        nb = min(256, a.batch * ((w + 31) // 32) * ((h + 7) // 8))
        buf = (C.c_longlong * (6 * nb))()
        eng.lib.savsr_debug_read_conv_stamps(buf, nb)
        eng.lib.savsr_debug_conv_stamps(0)
        st = np.array(buf[:], dtype=np.int64).reshape(nb, 6)
        d = np.diff(st[:, :5], axis=1)
        print("stamps (shader cycles, median over workgroups): first staging %d  phase0 %d  rest of the first tile K loop %d  later tiles + epilogues + drain %d  total %d" %
              tuple(np.median(d, axis=0).tolist() + [np.median(st[:, 4] - st[:, 0])]))
        for wv in (0, 3, 4, 7):
            eng.lib.savsr_debug_conv_stamps(3 + wv)
            run()
            torch.cuda.synchronize()
            eng.lib.savsr_debug_read_conv_stamps(buf, nb)
            eng.lib.savsr_debug_conv_stamps(0)
            s3 = np.array(buf[:], dtype=np.int64).reshape(nb, 6)[:, :5]
            print("wave-%d section cycles summed over the phases (median): steps after the barrier (+ staging issue) %d | steps before the barrier %d | wait %d | barrier %d | epilogue %d"
                  % tuple([wv] + np.median(s3, axis=0).tolist()))
        for flag, name in ((16, "no staging"), (32, "no fragment reads"), (48, "MFMAs only"), (64, "epilogue without stores"), (128, "epilogue without LDS transpose"), (192, "epilogue without either"), (256, "no epilogue body"), (512, "epilogue without the bias load")):
            eng.lib.savsr_debug_conv_stamps(7 + flag)
            run()
            torch.cuda.synchronize()
            eng.lib.savsr_debug_read_conv_stamps(buf, nb)
            s3 = np.array(buf[:], dtype=np.int64).reshape(nb, 6)[:, :5]
            ev0.record()
            for _ in range(5):
                run()
            ev1.record()
            torch.cuda.synchronize()
            eng.lib.savsr_debug_conv_stamps(0)
            print("experiment (%s; results invalid): %.2f us/iter; wave-4 sections %s" % (name, 1e3 * ev0.elapsed_time(ev1) / 5, np.median(s3, axis=0).astype(int).tolist()))
        rt = st[:, 5]
        print("workgroup start spread (100 MHz ticks): min %d max %d -> %.2f us" % (rt.min(), rt.max(), (rt.max() - rt.min()) / 100.0))
    if a.cycles and a.what == "satu":
        import ctypes as C
        import numpy as np
        nb = ((w + 31) // 32) * ((h + 7) // 8)
        buf = (C.c_longlong * (8 * nb))()
        eng.lib.savsr_debug_read_satu_stamps(buf, nb)
        tot = np.array(buf[:], dtype=np.int64).reshape(nb, 8)[:, 7]
        print("LR workgroup s_memtime totals (instrumented library): median %d  max %d" % (np.median(tot), tot.max()))
    if a.stamps and a.what == "satu":
        import ctypes as C
        import numpy as np
        from savsr_amd import _lib
        sw = C.byref(eng.satu_w)
        ax = eng.satu_axes(h, w, (4, 4))
        sw = C.byref(eng.satu_w_tail)
        lrcat = eng.buf("satu.lrcat_tail", h, w, _lib.SATU_LRCAT_TAIL)
        for name, call, nb in (
            ("LR", lambda: eng.lib.savsr_satu_lr_stage_tail(sw, x.data_ptr(), st.data_ptr(), 64, w, h, w, lrcat.data_ptr(), eng._stream()), ((w + 31) // 32) * ((h + 7) // 8)),
            ("HR", lambda: eng.lib.savsr_satu_hr_tail(sw, lrcat.data_ptr(), h, w, ax["table"].data_ptr(), ax["n_uh"], ax["n_uw"], ax["ih"].data_ptr(), ax["iw"].data_ptr(),
                                                      E._ptr(ax["ptab"]), ax["gyn"].data_ptr(), ax["gxn"].data_ptr(), H, W, C.byref(ax["tiling_tail"]), eng.hr_sched.data_ptr() if os.environ.get("SAVSR_HR_STATIC") != "1" else None, out.data_ptr(), plane, eng._stream()),
             512)):
            eng.lib.savsr_debug_satu_stamps(1)
            call()
            torch.cuda.synchronize()
            nb = min(nb, 2048)
            buf = (C.c_longlong * (8 * nb))()
            eng.lib.savsr_debug_read_satu_stamps(buf, nb)
            eng.lib.savsr_debug_satu_stamps(0)
            stt = np.array(buf[:], dtype=np.int64).reshape(nb, 8)
            print(name, "section cycles (median over workgroups, wave 0):", np.median(stt, axis=0).astype(int).tolist())
            tot = stt[:, 7]
            print(name, "workgroup totals: mean %d  p10 %d  p50 %d  p90 %d  p99 %d  max %d  (n = %d)" %
                  (tot.mean(), np.percentile(tot, 10), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), nb))
            if name == "HR":
                t0, t1 = stt[:, 5], stt[:, 6]
                ok = t1 > 0
                t0, t1 = t0[ok], t1[ok]
                til = ax["tiling_tail"]
                lds_b = int(eng.lib.savsr_satu_hr_lds_bytes(1, ax["n_uh"] * ax["n_uw"], til.tile_rows, til.tile_cols32, til.lr_rows, til.lr_cols))
                print("HR wall clock (100 MHz ticks): last start %.1f us after the first, first end %.1f us, last end %.1f us; workgroup duration p50 %.1f us -> %.2f GHz; "
                      "LDS %d B, occupancy query: %d workgroups / CU" %
                      ((t0.max() - t0.min()) / 100.0, (t1.min() - t0.min()) / 100.0, (t1.max() - t0.min()) / 100.0, np.median(t1 - t0) / 100.0,
                       np.median(tot[ok]) / (np.median(t1 - t0) * 10.0), lds_b, eng.lib.savsr_debug_satu_occupancy(0, lds_b)))
        for flag, nm in ((0, "normal"), (2, "no output stores (invalid results)"), (4, "staging only"), (8, "no window / table staging (invalid results)"),
                         (10, "no staging, no stores")):
            eng.lib.savsr_debug_satu_stamps(flag)
            hr = lambda: eng.lib.savsr_satu_hr_tail(sw, lrcat.data_ptr(), h, w, ax["table"].data_ptr(), ax["n_uh"], ax["n_uw"], ax["ih"].data_ptr(), ax["iw"].data_ptr(),
                                                    E._ptr(ax["ptab"]), ax["gyn"].data_ptr(), ax["gxn"].data_ptr(), H, W, C.byref(ax["tiling_tail"]), eng.hr_sched.data_ptr() if os.environ.get("SAVSR_HR_STATIC") != "1" else None, out.data_ptr(), plane, eng._stream())
            for _ in range(3):
                hr()
            ev0.record()
            for _ in range(10):
                hr()
            ev1.record()
            torch.cuda.synchronize()
            print("HR kernel alone, %s: %.1f us" % (nm, 1e2 * ev0.elapsed_time(ev1)))
        eng.lib.savsr_debug_satu_stamps(0)
        t = ax["tiling_tail"]
        print("HR tiling: rows", t.tile_rows, "cols32", t.tile_cols32, "window", t.lr_rows, "x", t.lr_cols)
    print(f"{a.what} cin={a.cin} cout={a.cout} ks={a.ks} {h}x{w}: {us:.2f} us/iter" + (f"  {flop / us / 1e6:.1f} TFLOP/s fp32-equivalent" if flop else ""))


if __name__ == "__main__":
    main()
