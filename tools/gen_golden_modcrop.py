"""Golden vectors of the reference's as_mod_crop (lbasicsr/data/transforms.py:48-69) for the YAML scale list.

Run in the build container only (imports the reference by file path with a cv2 stub; the reference never travels):
    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_modcrop.py   ->   tests/golden/as_mod_crop.json
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden_cases import YAML_SCALES  # noqa: E402


def main():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("ref_transforms", "/root/reference/lbasicsr/data/transforms.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = []
    for sc in YAML_SCALES:
        sc = tuple(float(s) for s in sc)
        for h, w in [(576, 720), (720, 1272), (480, 704), (256, 448), (1080, 1920)]:
            oh, ow = mod.as_mod_crop(np.zeros((h, w, 3), np.uint8), sc).shape[:2]
            rows.append([list(sc), h, w, int(oh), int(ow), int(mod.cal_step(sc[0])), int(mod.cal_step(sc[1]))])
    with open(os.path.join(ROOT, "tests", "golden", "as_mod_crop.json"), "w") as f:
        json.dump(rows, f)
    print(len(rows), "cases")


if __name__ == "__main__":
    main()
