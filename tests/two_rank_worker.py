"""Helper of tests/test_gpu_run_test.py::test_two_ranks_on_one_gpu: one rank of a world-size-2 run_test (both ranks on cuda:0,
gloo process group); writes this rank's result table to <out>.<rank>.pt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SAVSR_DIST_BACKEND"] = "gloo"
os.environ["LOCAL_RANK"] = "0"                    # every rank on the one GPU of the box

import torch  # noqa: E402

from savsr_amd import io as sio  # noqa: E402
from savsr_amd.options import parse_test_options  # noqa: E402
from savsr_amd.test import run_test  # noqa: E402

yaml_path, root, out = sys.argv[1:4]
opt = parse_test_options(open(yaml_path).read(), root_path=root)
opt["val"]["save_img"] = False
res = run_test(opt)
st = dict(sio.frame_store().stats)
torch.save({"results": [{k: r[k] for k in ("dataset", "scale", "metrics", "folders", "frames")} for r in res], "store": st},
           f"{out}.{os.environ.get('RANK', '0')}.pt")
