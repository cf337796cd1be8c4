"""Case tables shared by the golden generator (tools/gen_golden.py) and the tests."""
import numpy as np
import torch

NET_CASES = [
    ("cfg1_64x64_x2", 64, 64, (2, 2)),
    ("odd_17x21_x1p5_4", 17, 21, (1.5, 4)),
    ("s12x14_x3p7", 12, 14, (3.7, 3.7)),
    ("s13x16_x4", 13, 16, (4, 4)),
    ("s16x20_x3p5_2", 16, 20, (3.5, 2)),
]
SATU_CASES = [("x4", 6, 7, (4, 4)), ("x1p5_4", 7, 6, (1.5, 4)), ("x3p7", 5, 6, (3.7, 3.7)),
              ("x2p95_3p75", 6, 5, (2.95, 3.75))]
OSCONV_CASES = [("c192", "f2p_win.blocks.1.osconv", 192), ("c320", "h_win.0.blocks.0.osconv", 320),
                ("c64", "adapt.2.adapt", 64)]
OSCONV_SCALES = [(4, 4), (1.5, 4), (3.7, 3.7)]
YAML_SCALES = [(round(4.0 - 0.1 * i, 1),) * 2 for i in range(30)] + [
    (1.5, 4), (2, 4), (2, 3.75), (1.5, 3.5), (1.6, 3.05), (1.7, 3.75),
    (2.95, 3.75), (3.9, 2), (3.5, 1.5), (3.5, 2), (3.5, 1.75), (4, 1.4)]
GRID_SIZES = [(180, 320), (135, 239), (144, 176)]


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))


# Constructor configurations beside the shipped one (savsr_arch.py:576-604,638-659): frame sampling and other clip lengths.
# (name, ctor kwargs, h, w, scale); goldens: tools/gen_golden_configs.py -> tests/golden/config_outputs.npz
CONFIG_CASES = [
    ("t7_i1", dict(num_frame=7, interval=1), 12, 14, (2.5, 3.0)),     # odd centre index: iter_win 5, no pyramid level
    ("t9_i1", dict(num_frame=9, interval=1), 10, 12, (4, 4)),         # even centre index: the other branch of frame_sample
    ("t5_i0", dict(num_frame=5), 11, 13, (3.5, 2)),                   # 5 frames, no pyramid level, odd LR size
    # (num_frame = 9 without frame sampling has two pyramid levels: the REFERENCE's forward fails there -- WindowUnit_l2 :488 indexes
    #  five inputs, level one returns three -- so there is nothing to match; this repo's constructor rejects it)
]


def manifest_hash(manifest) -> str:
    """sha256 over the (name, shape, dtype) entries of a state_dict manifest (synth.manifest_of), in order."""
    import hashlib
    m = hashlib.sha256()
    for k, shape, dtype in manifest:
        m.update(k.encode())
        m.update(repr(tuple(int(v) for v in shape)).encode())
        m.update(str(dtype).encode())
    return m.hexdigest()


# `downsampling_mode: core` (lbasicsr/data/core.py::imresize, data_util.py:411-412): (name, channels, h, w, (sh, sw)); the output
# size is data_util.py's (round(h / sh), round(w / sw)).  Goldens: tools/gen_golden_core.py -> tests/golden/core_resize.npz
CORE_RESIZE_CASES = [
    ("x4", 3, 40, 52, (4, 4)), ("x1p5_4", 3, 42, 48, (1.5, 4)), ("x3p5_2", 2, 35, 44, (3.5, 2)), ("x3p7", 3, 37, 53, (3.7, 3.7)),
    ("x1p1", 1, 33, 44, (1.1, 1.1)), ("x2p95_3p75", 3, 59, 75, (2.95, 3.75)), ("id_h", 2, 20, 36, (1, 3)), ("small", 1, 9, 11, (4, 4)),
]


def core_input(c, h, w):
    return torch.from_numpy(np.random.RandomState(c * 1000 + h * 10 + w).uniform(0, 1, (c, h, w)).astype(np.float32))
