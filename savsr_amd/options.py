"""YAML option loading with the reference's semantics (lbasicsr/utils/options.py:14-52,149-204):
mappings become OrderedDict, `!!python/tuple`, `!!float` and `~` resolve, dataset `phase` is derived
from the key prefix and test result paths are filled in.  `options/test/SAVSR/*.yml` parse unchanged."""
from __future__ import annotations

import os
from collections import OrderedDict
from os import path as osp

import yaml


def _loader():
    class Loader(yaml.SafeLoader):
        pass

    def construct_mapping(loader, node):
        return OrderedDict(loader.construct_pairs(node))

    def construct_tuple(loader, node):
        return tuple(loader.construct_sequence(node))

    Loader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG, construct_mapping)
    Loader.add_constructor("tag:yaml.org,2002:python/tuple", construct_tuple)
    return Loader


def yaml_load(f: str):
    """Load a YAML file path or a YAML string."""
    if os.path.isfile(f):
        with open(f, "r") as fh:
            return yaml.load(fh, Loader=_loader())
    return yaml.load(f, Loader=_loader())


def parse_test_options(opt_path_or_str: str, root_path: str = ".", rank: int = 0, world_size: int = 1) -> OrderedDict:
    """The test-time part of parse_options (options.py:100-204) without argparse / dist init."""
    opt = yaml_load(opt_path_or_str)
    opt["dist"] = world_size > 1
    opt["rank"], opt["world_size"] = rank, world_size
    opt["is_train"] = False
    for phase, dataset in opt["datasets"].items():
        dataset["phase"] = phase.split("_")[0]
        if "scale" in opt:
            dataset["scale"] = opt["scale"]
        for k in ("dataroot_gt", "dataroot_lq"):
            if dataset.get(k) is not None:
                dataset[k] = osp.expanduser(dataset[k])
    for key, val in opt["path"].items():
        if val is not None and ("resume_state" in key or "pretrain_network" in key):
            opt["path"][key] = osp.expanduser(val)
    results_root = opt["path"].get("results_root") or osp.join(root_path, "results")
    results_root = osp.join(results_root, opt["name"])
    opt["path"]["results_root"] = results_root
    opt["path"]["log"] = results_root
    opt["path"]["visualization"] = osp.join(results_root, "visualization")
    return opt
