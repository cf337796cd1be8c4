"""Mean counter value per kernel and launch from rocprofv3 --pmc output directories.

    python3 tools/pmc_summary.py DIR [DIR ...] > summary.csv

Each DIR is the -d directory of one `rocprofv3 --kernel-trace --pmc <counters> -- python3 ...` pass (counter passes are
collected in their own runs, without any other trace flag, as the pool requires).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    print("pass,kernel,counter,launches,mean_per_launch")
    for d in sys.argv[1:]:
        acc = defaultdict(lambda: defaultdict(float))       # (kernel, counter) -> dispatch -> value (summed over instances)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    acc[(k, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (k, c), per in sorted(acc.items()):
            if "savsr" not in k:
                continue
            v = list(per.values())
            print('%s,"%s",%s,%d,%.1f' % (os.path.basename(d.rstrip("/")), k, c, len(v), sum(v) / len(v)))     # (kernel names hold commas: quoted)


if __name__ == "__main__":
    main()
