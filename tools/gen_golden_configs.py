"""Golden fixtures for the constructor configurations beside the shipped one (frame sampling, other clip lengths):
outputs of the REFERENCE's `SAVSR(**cfg)` on key-seeded weights, the index lists its `frame_sample` picks, and a hash of
its state_dict manifest (names + shapes) so that the tests can check this repo's parameter tree without storing 4 x 791 keys.

Build-container only (needs /root/reference).  Writes tests/golden/config_outputs.npz.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_configs.py
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_import  # noqa: E402
from golden_cases import CONFIG_CASES, manifest_hash  # noqa: E402
from savsr_amd.utils import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    torch.set_num_threads(8)
    ref = ref_import.load_reference_arch()
    out = {}
    with torch.no_grad():
        for name, cfg, h, w, sc in CONFIG_CASES:
            net = ref.SAVSR(**cfg).eval()
            manifest = synth.manifest_of(net.state_dict())
            out[f"{name}/manifest_sha"] = np.frombuffer(bytes.fromhex(manifest_hash(manifest)), dtype=np.uint8)
            out[f"{name}/n_keys"] = np.array([len(manifest)], dtype=np.int32)
            net.load_state_dict(synth.synth_state_dict(manifest, seed=3), strict=True)
            t = cfg.get("num_frame", 7)
            ar = torch.arange(t, dtype=torch.float32).view(1, t, 1, 1, 1)
            f, b = net.frame_sample(ar, t, interval=net.interval)
            out[f"{name}/fwd_idx"] = f.flatten().numpy().astype(np.int32)
            out[f"{name}/bwd_idx"] = b.flatten().numpy().astype(np.int32)
            out[f"{name}/iter_win"] = np.array([net.iter_win], dtype=np.int32)
            lq = synth.synth_clip(t, 3, h, w, seed=5)
            net.set_scale(sc)
            sr = net(lq)
            out[f"{name}/sr"] = sr.numpy()
            print(name, cfg, tuple(sr.shape), float(sr.abs().max()), f.flatten().tolist(), b.flatten().tolist(), net.iter_win)
    np.savez_compressed(os.path.join(GOLD, "config_outputs.npz"), **out)
    print("wrote", os.path.getsize(os.path.join(GOLD, "config_outputs.npz")) / 1e3, "KB")


if __name__ == "__main__":
    main()
