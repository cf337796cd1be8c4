"""Isolated import of the reference SAVSR arch (build-container only).

The reference package cannot be imported as a package here (cv2 / torchvision / generated
version.py are absent), so the arch file and the registry are loaded by file path after
registering empty parent packages and two stub modules.

This module is used ONLY by golden-vector generation (tools/gen_golden.py) and by
container-only validation tests that skip when /root/reference is absent. Nothing from the
reference is copied into this repository and this file never runs on the GPU box.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("SAVSR_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "lbasicsr", "archs", "savsr_arch.py"))


def load_reference_arch():
    """Returns the reference `savsr_arch` module object (classes SAVSR, OSConv2d, ...)."""
    sys.dont_write_bytecode = True
    if "lbasicsr.archs.savsr_arch" in sys.modules:
        return sys.modules["lbasicsr.archs.savsr_arch"]
    import torch.nn as nn

    for name in ("lbasicsr", "lbasicsr.archs", "lbasicsr.utils", "lbasicsr.metrics"):
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = []
            sys.modules[name] = pkg

    def _load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_ROOT, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    _load("lbasicsr.utils.registry", "lbasicsr/utils/registry.py")

    au = types.ModuleType("lbasicsr.archs.arch_util")

    def make_layer(basic_block, num_basic_block, **kwarg):
        return nn.Sequential(*[basic_block(**kwarg) for _ in range(num_basic_block)])

    au.make_layer = make_layer
    sys.modules["lbasicsr.archs.arch_util"] = au

    rt = types.ModuleType("lbasicsr.metrics.runtime")
    rt.VSR_runtime_test = lambda *a, **k: None
    sys.modules["lbasicsr.metrics.runtime"] = rt

    return _load("lbasicsr.archs.savsr_arch", "lbasicsr/archs/savsr_arch.py")
