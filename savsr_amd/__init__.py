"""savsr_amd -- MI355X-native SAVSR inference path."""
from .registry import ARCH_REGISTRY, DATASET_REGISTRY, METRIC_REGISTRY, MODEL_REGISTRY  # noqa: F401
from .archs import build_network  # noqa: F401
from .archs.savsr_arch import SAVSR  # noqa: F401
from . import datasets as _datasets, models as _models  # noqa: F401,E402  (register ASVideoTestDataset / ASVSRModel)
from .datasets import build_dataset  # noqa: F401,E402
from .models import build_model  # noqa: F401,E402

__version__ = "0.1.0"
