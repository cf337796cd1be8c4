"""`build_network(opt)` with the reference's contract (lbasicsr/archs/__init__.py:19-28):
pop `type`, look it up in ARCH_REGISTRY, construct with the remaining kwargs."""
from copy import deepcopy

from ..registry import ARCH_REGISTRY
from . import savsr_arch  # noqa: F401  (registers SAVSR)

__all__ = ["build_network", "ARCH_REGISTRY"]


def build_network(opt):
    opt = deepcopy(opt)
    network_type = opt.pop("type")
    return ARCH_REGISTRY.get(network_type)(**opt)
