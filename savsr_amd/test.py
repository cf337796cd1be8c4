"""`run_test(opt)`: a parsed options/test/SAVSR/*.yml -> metric table, the function form of the reference's
lbasicsr/test.py:11-48 (test_pipeline) without its CLI, loggers and experiment-directory bookkeeping.

    from savsr_amd.options import parse_test_options
    from savsr_amd.test import run_test
    results = run_test(parse_test_options("options/test/SAVSR/test_SAVSR_Vid4_asBI.yml"))
    # [{'dataset': 'Vid4_x4', 'scale': (4, 4), 'metrics': {'psnr_y': ..., 'ssim_y': ...}, 'folders': {...}}, ...]

Multi-GPU: launch one process per GPU (torch.distributed.run); `run_test` reads RANK / WORLD_SIZE / LOCAL_RANK,
initialises the "nccl" (= RCCL) process group when WORLD_SIZE > 1 and every rank returns the same table.
"""
from __future__ import annotations

import os
from typing import List, Union

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
            model.opt["rank"], model.opt["world_size"], model.opt["dist"] = opt["rank"], opt["world_size"], opt["dist"]
        results = []
        for ds in test_sets:                                                              # :37-48
            results.append(model.validation(ds, current_iter=opt["name"], tb_logger=None, save_img=opt["val"].get("save_img", False)))
        return results
    finally:
        if own_group:
            import torch.distributed as dist
            dist.destroy_process_group()
