#!/usr/bin/env python3
"""GPU-busy fraction and kernel concurrency of a run from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ...;  python3 tools/trace_busy.py DIR [--last-s 3.0]

Reads *kernel_trace.csv (Start_Timestamp / End_Timestamp in ns), takes the last `--last-s` seconds of kernel activity that precede the
final idle gap (the bench's timed regions run back to back at the end of the GPU work) and prints: wall, union-busy time (some kernel
running), sum of kernel durations (average concurrency = sum / union), the largest idle gaps, and the time by kernel name inside the window.
"""
import argparse
import csv
import glob
import os
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--last-s", type=float, default=3.0)
    ap.add_argument("--skip-tail-s", type=float, default=0.0, help="ignore this much at the very end (roofline / cpu-baseline legs)")
    a = ap.parse_args()
    rows = []
    for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
    rows.sort()
    t_end = max(e for _, e, _ in rows) - int(a.skip_tail_s * 1e9)
    t0 = t_end - int(a.last_s * 1e9)
    win = [(max(s, t0), min(e, t_end), n) for s, e, n in rows if e > t0 and s < t_end]
    union, cur_s, cur_e, gaps = 0, None, None, []
    for s, e, _ in win:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
                gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    total = sum(e - s for s, e, _ in win)
    wall = t_end - t0
    print(f"window {wall / 1e9:.3f} s: some kernel running {100 * union / wall:.1f} %, sum of kernel durations {total / 1e9:.3f} s = {total / union:.2f} kernels in flight on average")
    gaps.sort(reverse=True)
    print(f"idle gaps: {len(gaps)}, total {sum(gaps) / 1e6:.1f} ms, largest {[round(g / 1e3, 1) for g in gaps[:5]]} us, median {gaps[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us")
    by = defaultdict(int)
    for s, e, n in win:
        by[n] += e - s
    for n, t in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {100 * t / total:5.1f} % of kernel time  {t / wall:5.2f} x wall  {n[:100]}")


if __name__ == "__main__":
    main()
