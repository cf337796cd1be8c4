"""Deterministic, key-seeded synthetic weights and clips.

There is no network access to `savsr_best.pth`, so parity tests, golden fixtures and bench.py
all use weights generated from the state_dict key itself: every tensor is drawn from a numpy
RandomState seeded by crc32(key) ^ seed, so any process (build container, GPU box, any rank)
regenerates bit-identical weights from the committed key/shape manifest.  BatchNorm running
statistics, affine terms and every bias are randomised so those code paths are exercised.
"""
from __future__ import annotations

import json
import os
import zlib
from collections import OrderedDict

import numpy as np
import torch

_MANIFEST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "state_manifest.json")


def load_manifest(path: str | None = None):
    with open(path or _MANIFEST, "r") as f:
        return json.load(f)


def manifest_of(state_dict) -> list:
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in state_dict.items()]


def _draw(key: str, shape, seed: int) -> np.ndarray:
    rng = np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    n = int(np.prod(shape)) if shape else 1
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return (0.1 * rng.standard_normal(n)).reshape(shape)
    if leaf == "running_var":
        return rng.uniform(0.5, 1.5, n).reshape(shape)
    if key == "gamma":
        return np.full(shape, 0.9)
    if leaf == "bias":
        if ".bn." in key or ".mask." in key and len(shape) == 1 and _is_bn_key(key):
            return (0.1 * rng.standard_normal(n)).reshape(shape)
        return (0.05 * rng.standard_normal(n)).reshape(shape)
    if leaf in ("weight_compress", "weight_expand"):
        fan_in = shape[2]
        b = 1.0 / np.sqrt(fan_in)
        return rng.uniform(-b, b, n).reshape(shape)
    # leaf == 'weight'
    if len(shape) == 1:                      # BatchNorm affine scale
        return rng.uniform(0.5, 1.5, n).reshape(shape)
    if len(shape) == 5:                      # OSConv kernel bank [K, Cout, Cin, k, k]
        fan_in = shape[2] * shape[3] * shape[4]
    elif len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
    else:
        fan_in = shape[1]
    gain = _GAINS.get(key.split(".")[0] if not key.startswith("upsample") else ".".join(key.split(".")[:2]), 1.0)
    return (gain * rng.standard_normal(n) / np.sqrt(fan_in)).reshape(shape)


# Per-subtree gains keep activations O(1) through the 32-RCAB trunk and give sampling offsets of
# up to about a pixel, so random-weight outputs land in a range where tolerances are meaningful.
_GAINS = {"RG": 0.35, "adapt": 0.5, "conv_last": 0.5, "tail": 0.25,
          "upsample.kernel_conv": 0.25, "upsample.fusion": 0.5,
          "upsample.body": 2.5, "upsample.offset": 0.5, "upsample.st_offset": 0.4, "upsample.routing": 3.0}


def _is_bn_key(key: str) -> bool:
    # OSAdapt.mask Sequential: indices 1, 5, 8, 12 are BatchNorm2d (savsr_arch.py:189-206)
    parts = key.split(".")
    return "mask" in parts and parts[parts.index("mask") + 1] in ("1", "5", "8", "12")


def synth_state_dict(manifest=None, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Build a full state_dict (reference key names) from the committed manifest."""
    manifest = manifest if manifest is not None else load_manifest()
    out = OrderedDict()
    for key, shape, dtype in manifest:
        arr = _draw(key, shape, seed)
        if dtype == "int64":
            out[key] = torch.from_numpy(arr.astype(np.int64)).reshape(shape)
        else:
            out[key] = torch.from_numpy(arr.astype(np.float32)).reshape(shape)
    return out


def synth_clip(t: int, c: int, h: int, w: int, seed: int = 0, batch: int = 1) -> torch.Tensor:
    """Seeded U[0,1) clip [b, t, c, h, w] (numpy RandomState so it is reproducible everywhere)."""
    rng = np.random.RandomState(1000 + seed)
    return torch.from_numpy(rng.uniform(0.0, 1.0, (batch, t, c, h, w)).astype(np.float32))


def synth_gt(c: int, H: int, W: int, seed: int = 0) -> torch.Tensor:
    """Seeded smooth-ish ground truth in [0,1]: low-pass filtered uniform noise."""
    rng = np.random.RandomState(2000 + seed)
    a = rng.uniform(0.0, 1.0, (c, H + 8, W + 8)).astype(np.float32)
    k = np.ones(9, dtype=np.float32) / 9.0
    a = np.apply_along_axis(lambda r: np.convolve(r, k, mode="valid"), 1, a)
    a = np.apply_along_axis(lambda r: np.convolve(r, k, mode="valid"), 2, a)
    a = (a - a.min()) / max(float(a.max() - a.min()), 1e-6)
    return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))
