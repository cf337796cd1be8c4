"""`run_test(opt)`: a parsed options/test/SAVSR/*.yml -> metric table, the function form of the reference's
lbasicsr/test.py:11-48 (test_pipeline) without its CLI, loggers and experiment-directory bookkeeping.

    from savsr_amd.options import parse_test_options
    from savsr_amd.test import run_test
    results = run_test(parse_test_options("options/test/SAVSR/test_SAVSR_Vid4_asBI.yml"))
    # [{'dataset': 'Vid4_x4', 'scale': (4, 4), 'metrics': {'psnr_y': ..., 'ssim_y': ...}, 'folders': {...}}, ...]

Multi-GPU: launch one process per GPU (torch.distributed.run); `run_test` reads RANK / WORLD_SIZE / LOCAL_RANK,
initialises the "nccl" (= RCCL) process group when WORLD_SIZE > 1 and every rank returns the same table.

Command line (lbasicsr/test.py's `-opt`), with the real-data checker:

    python -m savsr_amd.test -opt options/test/SAVSR/test_SAVSR_Vid4_asBI.yml --check-readme

runs the unchanged YAML (weights at path.pretrain_network_g, frames at datasets.*.dataroot_gt), prints the per-dataset metric table
and, with --check-readme, each dataset's difference to the numbers the reference publishes for `savsr_best.pth`
(/root/reference/README.md:86-124, committed as data in savsr_amd/data/readme_psnr.json); exits 1 when any dataset is further than
0.01 dB / 1e-4 (+ half a unit of the published rounding) from its README entry, 2 when a dataset has no README entry.
"""
from __future__ import annotations

import json
import os
import re
import sys
from typing import Dict, List, Optional, Tuple, Union

import torch

from .datasets import build_dataset
from .models import build_model
from .options import parse_test_options


def run_test(opt: Union[str, dict], root_path: str = ".", model=None) -> List[dict]:
    """model: an already built model (build_model(opt)) to run the datasets through -- several YAMLs / passes over one set of
    weights then share its engine (captured graphs, packed weights); None builds one from `opt` as test.py:35 does."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if isinstance(opt, str):
        opt = parse_test_options(opt, root_path, rank, world)
    opt.setdefault("rank", rank)
    opt.setdefault("world_size", world)
    opt.setdefault("dist", opt["world_size"] > 1)
    if not torch.cuda.is_available():
        raise RuntimeError("run_test needs an AMD GPU (no CPU fallback)")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    own_group = False
    if opt["dist"]:
        import torch.distributed as dist
        if not dist.is_initialized():
            backend = os.environ.get("SAVSR_DIST_BACKEND", "nccl")          # "gloo": tests with several ranks on one GPU
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", torch.cuda.current_device()))   # RCCL over xGMI
            else:
                dist.init_process_group(backend=backend)
            own_group = True
    try:
        test_sets = [build_dataset(d) for _, d in sorted(opt["datasets"].items())]       # test.py:26-32
        if model is None:
            model = build_model(opt)                                                      # :35
        else:
            # a prebuilt model runs the NEW opt's validation settings (metrics, suffix, output paths, name); the network it was built
            # with must be the one the new opt describes
            if dict(model.opt.get("network_g", {})) != dict(opt.get("network_g", model.opt.get("network_g", {}))):
                raise ValueError("run_test(opt, model=...): opt['network_g'] differs from the network the model was built with")
            new_ckpt, old_ckpt = opt.get("path", {}).get("pretrain_network_g"), model.opt.get("path", {}).get("pretrain_network_g")
            if new_ckpt is not None and new_ckpt != old_ckpt:
                raise ValueError("run_test(opt, model=...): opt['path']['pretrain_network_g'] differs from the checkpoint the model loaded")
            for key in ("val", "name", "emulate_world"):
                if key in opt:
                    model.opt[key] = opt[key]
            if "path" in opt:
                model.opt.setdefault("path", {}).update({k: v for k, v in opt["path"].items() if k != "pretrain_network_g"})
            model.opt["rank"], model.opt["world_size"], model.opt["dist"] = opt["rank"], opt["world_size"], opt["dist"]
        save_img = opt["val"].get("save_img", False)
        if hasattr(model, "validate_job") and all(hasattr(ds, "units") for ds in test_sets) and os.environ.get("SAVSR_JOB_PLAN", "1") != "0":
            # all datasets of the YAML as one job: (dataset, folder) units cut over the ranks, ONE gather of the metric rows (models.validate_job)
            return model.validate_job(test_sets, current_iter=opt["name"], tb_logger=None, save_img=save_img)
        results = []
        for ds in test_sets:                                                              # :37-48, dataset by dataset (SAVSR_JOB_PLAN=0)
            results.append(model.validation(ds, current_iter=opt["name"], tb_logger=None, save_img=save_img))
        return results
    finally:
        if own_group:
            import torch.distributed as dist
            dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------- real-data checker
README_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "readme_psnr.json")
TOL_PSNR, TOL_SSIM = 0.01, 1e-4                 # north_star: Vid4 x4 PSNR within 0.01 dB of the reference; SSIM 1e-4
ROUND_PSNR, ROUND_SSIM = 0.005, 0.00005         # the README prints 2 / 4 decimals


def _scale_key(scale) -> str:
    def f(v):
        v = float(v)
        return str(int(v)) if v == int(v) else repr(v)
    return f"{f(scale[0])},{f(scale[1])}"


def readme_entry(table: dict, dataset: str, scale) -> Optional[Tuple[float, float]]:
    """README entry of a result row: table key = the dataset name up to its first '_x' ('Vid4_x1.5_x4' -> 'Vid4'), then 'sh,sw'."""
    fam = re.split(r"_x", dataset, maxsplit=1)[0]
    ent = table.get(fam, {}).get(_scale_key(scale))
    return None if ent is None else (float(ent[0]), float(ent[1]))


def check_readme(results: List[dict], table: Optional[dict] = None, psnr_key: str = "psnr_y", ssim_key: str = "ssim_y") -> Tuple[List[dict], int]:
    """Compare run_test's rows with the published tables.  Returns (rows, status): rows = [{dataset, scale, psnr, ssim, readme_psnr,
    readme_ssim, d_psnr, d_ssim, ok}], status 0 = every dataset within tolerance, 1 = at least one outside, 2 = a dataset without a
    README entry (nothing to compare with is not a pass)."""
    if table is None:
        table = json.load(open(README_TABLE))
    rows, status = [], 0
    for r in results:
        ent = readme_entry(table, r["dataset"], r["scale"])
        row: Dict[str, object] = {"dataset": r["dataset"], "scale": tuple(r["scale"]), "psnr": r["metrics"].get(psnr_key), "ssim": r["metrics"].get(ssim_key)}
        if ent is None or row["psnr"] is None or row["ssim"] is None:
            row.update(readme_psnr=None, readme_ssim=None, d_psnr=None, d_ssim=None, ok=False)
            status = max(status, 2)
        else:
            dp, ds_ = float(row["psnr"]) - ent[0], float(row["ssim"]) - ent[1]
            ok = abs(dp) <= TOL_PSNR + ROUND_PSNR and abs(ds_) <= TOL_SSIM + ROUND_SSIM
            row.update(readme_psnr=ent[0], readme_ssim=ent[1], d_psnr=dp, d_ssim=ds_, ok=ok)
            if not ok:
                status = max(status, 1)
        rows.append(row)
    return rows, status


def format_check(rows: List[dict]) -> str:
    out = [f"{'dataset':<22}{'PSNR-Y':>9}{'README':>9}{'d dB':>9}   {'SSIM-Y':>8}{'README':>8}{'d':>10}  "]
    for r in rows:
        if r["readme_psnr"] is None:
            out.append(f"{r['dataset']:<22}{(r['psnr'] if r['psnr'] is not None else float('nan')):9.4f}{'-':>9}{'-':>9}   "
                       f"{(r['ssim'] if r['ssim'] is not None else float('nan')):8.4f}{'-':>8}{'-':>10}  NO README ENTRY")
        else:
            out.append(f"{r['dataset']:<22}{r['psnr']:9.4f}{r['readme_psnr']:9.2f}{r['d_psnr']:+9.4f}   {r['ssim']:8.4f}{r['readme_ssim']:8.4f}{r['d_ssim']:+10.5f}  "
                       + ("ok" if r["ok"] else "OUTSIDE TOLERANCE"))
    return "\n".join(out)


def main(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(prog="python -m savsr_amd.test", description=__doc__.split("\n")[0])
    ap.add_argument("-opt", required=True, help="options/test/SAVSR/*.yml (unchanged)")
    ap.add_argument("--root", default=".", help="root for results/ (lbasicsr/test.py uses the repo root)")
    ap.add_argument("--check-readme", action="store_true", help="compare every dataset with the reference's published PSNR-Y / SSIM-Y")
    a = ap.parse_args(argv)
    results = run_test(a.opt, a.root)
    if int(os.environ.get("RANK", "0")) != 0:
        return 0
    for r in results:
        print(f"{r['dataset']}: " + ", ".join(f"{k} {v:.4f}" for k, v in r["metrics"].items()))
    if not a.check_readme:
        return 0
    rows, status = check_readme(results)
    print(format_check(rows))
    print({0: "check-readme: every dataset within 0.01 dB / 1e-4 of the README", 1: "check-readme: FAILED (outside tolerance)",
           2: "check-readme: FAILED (dataset without a README entry)"}[status])
    return status


if __name__ == "__main__":
    sys.exit(main())
