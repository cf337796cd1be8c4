"""Synthetic workload definitions of BASELINE.json's configs 3-5 (shapes and scale lists only).

  config 3  Vid4 sweep: LR 180x320, the 30 symmetric scales x1.1 ... x4.0 (options/test/SAVSR/test_SAVSR_Vid4_asBI.yml:24-517)
  config 4  UDM10 asymmetric: GT 720x1272 -> as-mod-crop -> LR 480x318 at x(1.5, 4), LR 204x636 at x(3.5, 2)
            (lbasicsr/data/transforms.py:31-69, data_util.py:409-410)
  config 5  Vimeo90K training shapes: GT 256x448, (sh, sw) drawn from the 60-entry training list
            (lbasicsr/data/vimeo90k_dataset.py:178-203: the h- and w-lists are paired by index), LR = as-mod-crop / scale
"""
from __future__ import annotations

import random
from typing import List, Tuple

from ..resize_gpu import as_mod_crop_hw

CONFIG3_SCALES: List[Tuple[float, float]] = [(k / 10, k / 10) for k in range(11, 41)]

# the 12 asymmetric pairs the shipped YAMLs test beside the 30 symmetric scales (options/test/SAVSR/test_SAVSR_Vid4_asBI.yml:518-826,
# the same list in test_SAVSR_UDM10_asBI.yml); with CONFIG3_SCALES = the 42 datasets of a YAML
VID4_ASYM_SCALES: List[Tuple[float, float]] = [(1.5, 4.0), (2.0, 4.0), (2.0, 3.75), (1.5, 3.5), (1.6, 3.05), (1.7, 3.75),
                                               (2.95, 3.75), (3.9, 2.0), (3.5, 1.5), (3.5, 2.0), (3.5, 1.75), (4.0, 1.4)]
YAML_SCALES: List[Tuple[float, float]] = CONFIG3_SCALES + VID4_ASYM_SCALES

CONFIG4_CASES = [((720, 1272), (1.5, 4.0)), ((720, 1272), (3.5, 2.0))]

_SYM = [k / 10 for k in range(11, 41)]
_H_EXTRA = [s for s in (1.5, 2.0, 2.5, 3.0, 3.5, 4.0) for _ in range(5)]
_W_EXTRA = [2.0, 2.5, 3.0, 3.5, 4.0,  1.5, 2.5, 3.0, 3.5, 4.0,  1.5, 2.0, 3.0, 3.5, 4.0,
            1.5, 2.0, 2.5, 3.5, 4.0,  1.5, 2.0, 2.5, 3.0, 4.0,  1.5, 2.0, 2.5, 3.0, 3.5]
TRAIN_SCALES: List[Tuple[float, float]] = list(zip(_SYM + _H_EXTRA, _SYM + _W_EXTRA))       # 60 (sh, sw) pairs
VIMEO_GT = (256, 448)
VIMEO_CLIPS = 64612            # lines of meta_info_Vimeo90K_train_GT.txt


def lr_shape(gt_hw: Tuple[int, int], scale: Tuple[float, float]) -> Tuple[int, int]:
    """LR size the test / train flow feeds the network: as-mod-crop of the GT, then round(size / scale)."""
    H, W = as_mod_crop_hw(gt_hw[0], gt_hw[1], tuple(scale))
    return round(H / scale[0]), round(W / scale[1])


def config5_cases(n: int, seed: int = 0):
    """n seeded draws (with replacement, random.Random(seed)) from the training list: [(h_lr, w_lr, (sh, sw))]."""
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        sc = TRAIN_SCALES[rng.randrange(len(TRAIN_SCALES))]
        h, w = lr_shape(VIMEO_GT, sc)
        out.append((h, w, sc))
    return out
