"""Live comparison oracle <-> reference arch; runs only where /root/reference exists."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import  # noqa: E402

from oracle import savsr_oracle as O  # noqa: E402
from savsr_amd.utils import synth  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref_net():
    ref = ref_import.load_reference_arch()
    net = ref.SAVSR().eval()
    sd = synth.synth_state_dict(synth.manifest_of(net.state_dict()), seed=3)
    net.load_state_dict(sd, strict=True)
    return net, sd


def test_manifest_matches_reference(ref_net):
    net, _ = ref_net
    assert synth.manifest_of(net.state_dict()) == synth.load_manifest()


@pytest.mark.parametrize("h,w,sc", [(10, 12, (4, 4)), (11, 9, (2.5, 2.5)), (9, 11, (1.7, 3.75))])
def test_forward_identical(ref_net, h, w, sc):
    net, sd = ref_net
    lq = synth.synth_clip(7, 3, h, w, seed=4)
    net.set_scale(sc)
    with torch.no_grad():
        a = net(lq)
        b = O.forward(sd, lq, sc)
    assert float((a - b).abs().max()) <= 1e-6
