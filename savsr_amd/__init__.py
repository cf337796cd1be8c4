"""savsr_amd -- MI355X-native SAVSR inference path."""
from .registry import ARCH_REGISTRY, DATASET_REGISTRY, METRIC_REGISTRY, MODEL_REGISTRY  # noqa: F401
from .archs import build_network  # noqa: F401
from .archs.savsr_arch import SAVSR  # noqa: F401

__version__ = "0.1.0"
