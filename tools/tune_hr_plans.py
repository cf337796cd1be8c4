"""Measure the SATU HR stage's launch plan for every scale the reference's lists name and write savsr_amd/hr_plans.json, stamped with the
library's SATU source hash (`gpurun -- python3 tools/tune_hr_plans.py`; re-run after any change of satu.hip / tail.hip / common.hpp -- a table
from another build is ignored by the engine, which then times the candidates on a (size, scale)'s first frame as before).

Scales: the 42 datasets of the shipped YAMLs at LR 180 x 320, the 60 Vimeo90K training pairs at their LR sizes (a scale in both lists keeps the
YAML measurement).  Every feasible plan (wave split x HR tile) is timed on a loaded board (two passes of 3 + 20 launches, network frames in front of each pass); the fastest is kept."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from savsr_amd import _lib  # noqa: E402
from savsr_amd import engine as E  # noqa: E402
from savsr_amd.archs.savsr_arch import SAVSR  # noqa: E402
from savsr_amd.utils import synth, workloads  # noqa: E402


def main():
    os.environ["SAVSR_HR_PLANS"] = "0"                      # measure, whatever table is lying around
    dev = torch.device("cuda:0")
    eng = E.HipEngine(synth.synth_state_dict(), SAVSR().cfg, dev)
    cases = [(180, 320, sc) for sc in workloads.YAML_SCALES]
    seen = {tuple(sc) for _, _, sc in cases}
    cases += [workloads.lr_shape(workloads.VIMEO_GT, sc) + (sc,) for sc in workloads.TRAIN_SCALES if tuple(sc) not in seen]
    plans, log = {}, []
    warm = synth.synth_clip(7, 3, 180, 320, seed=0).to(dev)
    g = torch.Generator().manual_seed(0)
    for h, w, sc in cases:
        sc = (float(sc[0]), float(sc[1]))
        if sc in plans:
            continue
        x = torch.randn(h, w, 64, generator=g).to(dev)
        st = (0.6 * torch.randn(h, w, 64, generator=g)).to(dev)
        H, W = E.get_hw(h, w, sc)
        plane = eng.hr_plane(H, W)
        q9 = torch.empty(9, plane, device=dev)
        seam = torch.empty(eng.seam_floats(H, W), device=dev)
        lrcat = eng.satu_lr(eng.full(x), eng.full(st), w, h, w, tail_form=True, q=True)
        ax = eng.satu_axes(h, w, sc)
        # Two passes over all plans on a LOADED board (whole network frames in front of each pass: an idle board's first launches run up to
        # 10 % slower, which would favour whatever is measured last), 3 + 20 launches per plan and pass, the minimum of the passes per plan.
        times = {}
        for _ in range(2):
            for _ in range(12):
                eng.forward(warm, (4.0, 4.0))
            for k, til in enumerate(ax["tail_plans"]):
                ax["tiling_tail"] = til
                run = lambda: eng.satu_hr(lrcat, h, w, sc, q9, plane, tail_form=True, seam=seam)
                for _ in range(3):
                    run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run()
                e1.record()
                e1.synchronize()
                us = 1e3 * e0.elapsed_time(e1) / 20
                times[k] = min(us, times.get(k, 1e9))
        kbest = min(times, key=times.get)
        best = (times[kbest], ax["tail_plans"][kbest])
        us, til = best
        plans[sc] = [int(til.variant), int(til.tile_rows), int(til.tile_cols32), int(h), int(w)]
        log.append(f"{h}x{w} x{sc}: variant {til.variant}, {til.tile_rows} rows x {32 * til.tile_cols32} px, window {til.lr_rows} x {til.lr_cols}: {us:.1f} us  ({len(ax['tail_plans'])} plans)")
        print(log[-1], flush=True)
    out = {"satu_source_hash": eng.lib.savsr_source_hash_satu().decode(),
           "note": "SATU HR stage launch plan [variant, tile rows, tile columns / 32, LR h, LR w of the measurement] per scale 'sh,sw' (applied to LR sizes within 0.5 ... 2 x those pixels), measured by tools/tune_hr_plans.py on one MI355X with key-seeded "
                   "weights; used only when the hash equals the loaded library's savsr_source_hash_satu() and the plan is feasible for the loaded weights' offsets",
           "plans": {f"{k[0]:g},{k[1]:g}": v for k, v in plans.items()}}
    path = sys.argv[1] if len(sys.argv) > 1 else E.HipEngine.HR_PLANS_FILE
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(plans), "scales")


if __name__ == "__main__":
    main()
