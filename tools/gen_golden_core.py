"""Golden fixtures for `downsampling_mode: core`: outputs of the REFERENCE's lbasicsr/data/core.py::imresize(x, sizes=...) (the call
of data_util.py:411-412) on seeded inputs that any machine can regenerate.  Build-container only (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_core.py        -> tests/golden/core_resize.npz
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_cases import CORE_RESIZE_CASES, core_input  # noqa: E402


def main():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ref_core", "/root/reference/lbasicsr/data/core.py")
    core = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(core)
    out = {}
    with torch.no_grad():
        for name, c, h, w, sc in CORE_RESIZE_CASES:
            x = core_input(c, h, w)
            size = (round(h / sc[0]), round(w / sc[1]))
            y = core.imresize(x, sizes=size)
            assert tuple(y.shape) == (c, size[0], size[1]) and y.dtype == torch.float32
            out[name] = y.numpy()
            print(name, tuple(x.shape), "->", tuple(y.shape), float(y.abs().max()))
    path = os.path.join(ROOT, "tests", "golden", "core_resize.npz")
    np.savez_compressed(path, **out)
    print("wrote", os.path.getsize(path) / 1e3, "KB")


if __name__ == "__main__":
    main()
