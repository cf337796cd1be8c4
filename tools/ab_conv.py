#!/usr/bin/env python3
"""A/B of conv-kernel builds in ONE process, interleaved rounds (cdna_hip_programming.md rule 24): every library named on the
command line gets the SAME descriptors (device tensors, packed weights) through savsr_conv2d_batch; per round each variant is
looped `--iters` times between HIP events; median / min over rounds, plus the shader clock the chip held under each variant's
loop (savsr_clock_probe of the product library on a side stream) and max |difference| of the outputs to the first variant.

    python3 tools/ab_conv.py --libs savsr_amd/csrc/libsavsr_hip.so savsr_amd/csrc/libsavsr_hip_exp_p1.so [--shapes 6x128 1x64 6x64]
    AB_ONLY=conv_wy.hip bash tools/ab_conv.sh "-DWY_VALU=4" "-DWY_STAMPS=1"        # builds the variants, then runs this
"""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from savsr_amd import _lib  # noqa: E402
from savsr_amd import engine as E  # noqa: E402
from savsr_amd._lib import ACT_LRELU  # noqa: E402
from savsr_amd.archs.savsr_arch import SAVSR  # noqa: E402
from savsr_amd.engine import ConvDesc  # noqa: E402
from savsr_amd.utils import synth  # noqa: E402


def bind(path):
    lib = C.CDLL(path)
    res, args = _lib.SIGNATURES["savsr_conv2d_batch"]
    lib.savsr_conv2d_batch.restype, lib.savsr_conv2d_batch.argtypes = res, args
    lib.savsr_prepare_device.restype = C.c_int
    lib.savsr_last_error.restype = C.c_char_p
    assert lib.savsr_prepare_device() == 0, lib.savsr_last_error()
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--shapes", nargs="+", default=["6x128", "6x64", "1x64"], help="BATCHxCIN (3x3, cout 64, 180x320)")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--h", type=int, default=180)
    ap.add_argument("--w", type=int, default=320)
    ap.add_argument("--throughput", action="store_true", help="algo = CONV_DIRECT_THROUGHPUT (16-row tiles from 100 tiles up)")
    ap.add_argument("--only-wy", action="store_true", help="with --wy: the first library's direct form, then only the Winograd-y forms")
    ap.add_argument("--wy", action="store_true", help="also time every library's Winograd-y form (SAVSR_CONV_WINOGRAD_Y) of the same convs")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    eng = E.HipEngine(synth.synth_state_dict(), SAVSR().cfg, dev)
    if a.throughput:
        eng.conv_algo = _lib.CONV_DIRECT_THROUGHPUT
    libs_in = [(os.path.basename(p), bind(os.path.join(ROOT, p) if not os.path.isabs(p) else p)) for p in a.libs]
    probe_stream = torch.cuda.Stream(device=dev)
    probe_buf = torch.zeros(16, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    h, w, cout = a.h, a.w, 64
    for shape in a.shapes:
        nb, cin = (int(v) for v in shape.split("x"))
        g = torch.Generator().manual_seed(0)
        nsrc = max(1, cin // 64)
        keep, descs, descs_wy, outs = [], [], [], []
        for k in range(nb):                                   # distinct inputs / weights per conv, bias + LeakyReLU + residual
            wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
            weights = (E.pack_conv_weight(wt).to(dev), torch.randn(cout, generator=g).to(dev), cout, cin, 3)
            xs = [torch.randn(h, w, cin // nsrc, generator=g).to(dev) for _ in range(nsrc)]
            res, out = torch.randn(h, w, cout, generator=g).to(dev), torch.empty(h, w, cout, device=dev)
            keep.append((weights, xs, res))
            outs.append(out)
            descs.append(eng.conv_desc("bench", [eng.full(x) for x in xs], eng.full(out), h, w, ACT_LRELU, 0.2, res1=eng.full(res), weights=weights))
            if a.wy:
                wy = (E.pack_conv_weight_wy(wt).to(dev), weights[1], cout, cin, 3, _lib.CONV_WINOGRAD_Y_THROUGHPUT if a.throughput else _lib.CONV_WINOGRAD_Y)
                keep.append(wy)
                descs_wy.append(eng.conv_desc("bench", [eng.full(x) for x in xs], eng.full(out), h, w, ACT_LRELU, 0.2, res1=eng.full(res), weights=wy))
        arr0 = (ConvDesc * nb)(*descs)
        variants = [(n, l, arr0) for n, l in libs_in]
        if a.wy:
            arr1 = (ConvDesc * nb)(*descs_wy)
            variants += [(n + " [winograd-y]", l, arr1) for n, l in libs_in]
            if a.only_wy:
                variants = variants[:1] + variants[len(libs_in):]
        libs = [(n, (l, ar)) for n, l, ar in variants]
        times = {n: [] for n, _ in libs}
        clocks = {n: [] for n, _ in libs}
        ref, diffs = None, {}
        for name, (lib, arr) in libs:
            rc = lib.savsr_conv2d_batch(arr, nb, st)
            assert rc == 0, (name, rc, lib.savsr_last_error())
            torch.cuda.synchronize()
            cur = torch.stack([o.clone() for o in outs])
            if ref is None:
                ref = cur
            diffs[name] = float((cur - ref).abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for r in range(a.rounds):
            order = libs if r % 2 == 0 else libs[::-1]
            for name, (lib, arr) in order:
                for _ in range(3):
                    lib.savsr_conv2d_batch(arr, nb, st)
                torch.cuda.synchronize()
                est = (times[name][-1] if times[name] else 100.0) * a.iters          # us of the loop
                probe_buf.zero_()
                torch.cuda.synchronize()
                eng.lib.savsr_clock_probe(probe_buf.data_ptr(), max(100, int(0.6 * est * 100 / 8)), 8, probe_stream.cuda_stream)
                e0.record()
                for _ in range(a.iters):
                    lib.savsr_conv2d_batch(arr, nb, st)
                e1.record()
                torch.cuda.synchronize()
                times[name].append(1e3 * e0.elapsed_time(e1) / a.iters)
                win = sorted(100.0 * c / rt for c, rt in probe_buf.cpu().view(8, 2).tolist() if rt > 0)
                if win and r > 0:
                    clocks[name].append(win[len(win) // 2])
                    if r == 1:
                        print(f"    [{name}] probe windows MHz: " + " ".join(f"{v:.0f}" for v in (100.0 * c / rt for c, rt in probe_buf.cpu().view(8, 2).tolist() if rt > 0)))
        for name, (lib, arr) in libs:                          # instrumented Winograd builds (-DWY_STAMPS=1 / 2): section cycles per wave
            if hasattr(lib, "savsr_debug_read_wy_stamps") and "winograd" in name:
                import numpy as np
                for _ in range(5):
                    lib.savsr_conv2d_batch(arr, nb, st)
                torch.cuda.synchronize()
                buf = (C.c_longlong * (256 * 8 * 8))()
                lib.savsr_debug_read_wy_stamps(buf)
                stt = np.array(buf[:], dtype=np.int64).reshape(256, 8, 8)
                med = np.median(stt, axis=0)                   # [wave][section]
                names = ["A s0-4", "A wait+bar", "A s5", "B s0-4", "B wait+bar", "B s5", "epilogue", "total"]
                if os.environ.get("AB_STAMP_MODE", "") in ("2", "8192"):       # the library was built with -DWY_STAMPS=2
                    names = ["K loop", "epi loads + transform", "group 0", "group 1", "group 2", "group 3", "pool / rest", "total"]
                print(f"    [{name}] median cycles per wave over the launch (sections: " + ", ".join(names) + ")")
                for wv in (0, 3, 4, 7):
                    print(f"      wave {wv}: " + "  ".join(f"{int(v):7d}" for v in med[wv]))
        flop = 3 * 2.0 * nb * cin * cout * 9 * h * w
        print(f"== {nb} x conv3x3 {cin}->{cout} at {h}x{w} (distinct inputs, bias + LeakyReLU + residual){' [throughput tiling]' if a.throughput else ''}")
        base = statistics.median(times[libs[0][0]][1:])
        for name, _ in libs:
            t = times[name][1:]
            med = statistics.median(t)
            ck = statistics.median(clocks[name]) if clocks[name] else float("nan")
            print(f"  {name:<34} median {med:8.2f} us  min {min(t):8.2f}  ({100 * (med / base - 1):+5.1f} %)  {flop / med / 1e6:7.1f} TF bf16 = {flop / med / 1e6 / 2500:.3f} of 2.5 PF"
                  f"  clock {ck:6.0f} MHz  cycles/launch {med * ck / 1e3:7.1f} k  max|d| vs first {diffs[name]:.2e}")


if __name__ == "__main__":
    main()
