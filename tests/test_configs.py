"""Constructor configurations beside the shipped one (savsr_arch.py:576-604, 638-659): frame sampling (`interval`) and other
clip lengths.  Fixtures (tests/golden/config_outputs.npz, tools/gen_golden_configs.py) hold the REFERENCE's outputs on key-seeded
weights, the index lists its `frame_sample` picks, its iteration window and a hash of its state_dict manifest."""
import os

import numpy as np
import pytest
import torch

from oracle import savsr_oracle as O
from savsr_amd.utils import synth
from tests.golden_cases import CONFIG_CASES, manifest_hash

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cgold():
    return np.load(os.path.join(ROOT, "tests", "golden", "config_outputs.npz"))


def _net(cfg):
    from savsr_amd.archs.savsr_arch import SAVSR
    return SAVSR(**cfg)


@pytest.mark.parametrize("name,cfg,h,w,sc", CONFIG_CASES)
def test_parameter_tree_and_frame_sampling_match_the_reference(cgold, name, cfg, h, w, sc):
    from savsr_amd.archs.savsr_arch import frame_sample_indices, iteration_window
    net = _net(cfg)
    manifest = synth.manifest_of(net.state_dict())
    assert len(manifest) == int(cgold[f"{name}/n_keys"][0])
    assert manifest_hash(manifest) == bytes(cgold[f"{name}/manifest_sha"]).hex()       # names, shapes, dtypes, order
    t = cfg.get("num_frame", 7)
    fwd, bwd = frame_sample_indices(t, cfg.get("interval", 0))
    assert fwd == cgold[f"{name}/fwd_idx"].tolist() and bwd == cgold[f"{name}/bwd_idx"].tolist()
    assert iteration_window(t, cfg.get("interval", 0), t // 2) == int(cgold[f"{name}/iter_win"][0]) == net.iter_win
    # the oracle's restatement picks the same frames
    ar = torch.arange(t, dtype=torch.float32).view(1, t, 1, 1, 1)
    of, ob = O.frame_sample(ar, t, cfg.get("interval", 0))
    assert of.flatten().int().tolist() == fwd and ob.flatten().int().tolist() == bwd


@pytest.mark.parametrize("name,cfg,h,w,sc", CONFIG_CASES)
def test_oracle_vs_reference_golden(cgold, name, cfg, h, w, sc):
    sd = synth.synth_state_dict(synth.manifest_of(_net(cfg).state_dict()), seed=3)
    lq = synth.synth_clip(cfg.get("num_frame", 7), 3, h, w, seed=5)
    with torch.no_grad():
        sr = O.forward(sd, lq, sc, cfg=cfg)
    gold = torch.from_numpy(cgold[f"{name}/sr"])
    assert sr.shape == gold.shape
    assert float((sr - gold).abs().max()) <= 1e-6


def test_two_pyramid_levels_are_rejected_like_the_reference_fails():
    with pytest.raises(ValueError, match="pyramid"):
        _net(dict(num_frame=9))


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg,h,w,sc", CONFIG_CASES)
def test_gpu_vs_reference_golden(cgold, name, cfg, h, w, sc):
    """The HIP path with frame sampling / 5- and 9-frame clips: within 5e-5 max-abs of the reference's output, eager and replayed."""
    net = _net(cfg)
    net.load_state_dict(synth.synth_state_dict(synth.manifest_of(net.state_dict()), seed=3), strict=True)
    net = net.to("cuda:0").eval()
    net.set_scale(sc)
    lq = synth.synth_clip(cfg.get("num_frame", 7), 3, h, w, seed=5).to("cuda:0")
    gold = torch.from_numpy(cgold[f"{name}/sr"])
    taps = {}
    eager = net(lq, taps=taps).cpu()                     # taps force the eager launch sequence
    a = net(lq).cpu()                                    # captured
    b = net(lq).cpu()                                    # replayed
    assert eager.shape == gold.shape
    err = float((a - gold).abs().max())
    print(name, "max-abs vs reference", err)
    assert err < 5e-5
    assert torch.equal(a, b) and torch.equal(a, eager)
