"""Per-frame kernel breakdown from a rocprofv3 kernel trace of bench.py (developer tool).

    rocprofv3 --kernel-trace -d out -o p --output-format csv -- python3 bench.py --clips-per-step 1 --steps 5 --warmup 2 --no-cpu-baseline
    python3 tools/frame_breakdown.py out/p_kernel_trace.csv
"""
import collections
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "pack_windows" in r["Kernel_Name"]]
    seg = rows[idx[-3]:idx[-2]]                     # one steady-state frame (pack_windows opens every frame)
    c = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        n = r["Kernel_Name"][:52]
        c[n][0] += 1
        c[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(v[1] for v in c.values())
    for k, v in sorted(c.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:54s} {v[0]:4d} {v[1] / 1e3:9.1f} us {100 * v[1] / tot:5.1f}%")
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    print("busy", tot / 1e3, "span", (t1 - t0) / 1e3, "kernels", len(seg))


if __name__ == "__main__":
    main(sys.argv[1])
