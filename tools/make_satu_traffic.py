#!/usr/bin/env python3
"""profiles/satu_traffic.json from a PMC summary of the SATU launches (tools/pmc_satu.sh -> tools/pmc_summary.py).

    python3 tools/make_satu_traffic.py gpurun_out/r04_prof/satu_pmc.csv [--source TEXT] > profiles/satu_traffic.json

HBM bytes per launch = FETCH_SIZE x 2 (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced
streaming read) + WRITE_SIZE; both counters are in KiB (the LR stage's WRITE_SIZE reads 21 600.0 for its 57 600 x 384 B =
21 600 KiB of records), so x 1024 -- the round-1..3 files multiplied by 1000 and understated the bytes by 2.4 %.  The file is stamped with the hash of the
library sources it was measured on (savsr_source_hash(), compiled in by build.sh); bench.py drops `roofline.traffic` when the
library it runs differs.
"""
import argparse
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summary")
    ap.add_argument("--source", default=None)
    a = ap.parse_args()
    val = {}
    for r in csv.DictReader(open(a.summary)):
        k = r["kernel"]
        which = "lr" if "satu_lr" in k else ("hr" if "satu_hr" in k else ("tail" if "tail_gather" in k else None))
        if which and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
            # several template instances of one stage may appear (plan timing): keep the one with most launches
            cur = val.get((which, r["counter"]))
            if cur is None or int(r["launches"]) > cur[0]:
                val[(which, r["counter"])] = (int(r["launches"]), float(r["mean_per_launch"]), k)
    out = {}
    total = 0.0
    for which in ("lr", "hr", "tail"):
        if (which, "FETCH_SIZE") not in val or (which, "WRITE_SIZE") not in val:
            continue
        f, w = val[(which, "FETCH_SIZE")][1], val[(which, "WRITE_SIZE")][1]
        out[which] = {"fetch_kb_x2": round(2 * f, 1), "write_kb": round(w, 1), "kernel": val[(which, "FETCH_SIZE")][2]}
        if which != "tail":
            total += (2 * f + w) * 1024.0
    from savsr_amd import _lib
    lib = _lib.load()
    doc = {"bytes_per_stage": int(round(total)), **out,
           "bytes_satu_plus_tail": int(round(total + sum((out[k]["fetch_kb_x2"] + out[k]["write_kb"]) * 1024.0 for k in ("tail",) if k in out))),
           "lib_source_hash": lib.savsr_source_hash_satu().decode(),
           "source": a.source or f"{a.summary} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/time_satu.py with the HR plan forced, "
                                 "tools/pmc_satu.sh; FETCH_SIZE x 2 per MI355X_MICROARCH.md)",
           "workload": "config 2 (180x320 x4)"}
    json.dump(doc, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
