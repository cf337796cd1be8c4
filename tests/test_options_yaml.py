"""The reference's test YAMLs must parse unchanged."""
import os

import pytest

import savsr_amd
from savsr_amd.options import parse_test_options, yaml_load

SAMPLE = """
name: test_SAVSR_sample
model_type: ASVSRModel
num_gpu: 1
manual_seed: 0
datasets:
  test_01:
    name: Vid4_x4
    type: ASVideoTestDataset
    dataroot_gt: datasets/Vid4/GT
    io_backend:
      type: disk
    cache_data: false
    num_frame: 7
    padding: reflection
    use_arbitrary_scale_downsampling: true
    downsampling_scale: !!python/tuple [1.5, 4]
    downsampling_mode: torch
network_g:
  type: SAVSR
  num_in_ch: 3
  num_feat: 64
  num_frame: 7
  slid_win: 3
  fusion_win: 5
  interval: 0
  w1_num_block: 4
  w2_num_block: 2
  n_resgroups: 4
  n_resblocks: 8
  center_frame_idx: ~
path:
  pretrain_network_g: experiments/pretrained_models/SAVSR/savsr_best.pth
  strict_load_g: true
  resume_state: ~
val:
  save_img: true
  suffix: ~
  metrics:
    psnr_y:
      type: calculate_psnr
      crop_border: 0
      test_y_channel: true
    ssim_y:
      type: calculate_ssim
      crop_border: 0
      test_y_channel: true
"""


def _check(opt):
    ds = list(opt["datasets"].values())
    assert all(isinstance(d["downsampling_scale"], tuple) and len(d["downsampling_scale"]) == 2 for d in ds)
    assert all(d["phase"] == "test" for d in ds)
    assert opt["network_g"]["center_frame_idx"] is None and opt["path"]["resume_state"] is None
    net = savsr_amd.build_network(opt["network_g"])
    assert type(net).__name__ == "SAVSR"
    for m in opt["val"]["metrics"].values():
        assert m["type"] in savsr_amd.METRIC_REGISTRY
    assert opt["path"]["visualization"].endswith(os.path.join(opt["name"], "visualization"))


def test_sample_yaml():
    opt = parse_test_options(SAMPLE)
    assert opt["datasets"]["test_01"]["downsampling_scale"] == (1.5, 4)
    _check(opt)


@pytest.mark.parametrize("name", ["test_SAVSR_Vid4_asBI.yml", "test_SAVSR_UDM10_asBI.yml"])
def test_reference_yaml_parses_unchanged(name):
    path = os.path.join("/root/reference/options/test/SAVSR", name)
    if not os.path.isfile(path):
        pytest.skip("reference tree not present")
    opt = parse_test_options(path)
    assert len(opt["datasets"]) == 42
    assert yaml_load(path)["model_type"] == "ASVSRModel"
    _check(opt)


def _resolves(opt):
    """`model_type` and every dataset `type` resolve through the registries (YAML surface)."""
    assert savsr_amd.MODEL_REGISTRY.get(opt["model_type"]).__name__ == opt["model_type"]
    for d in opt["datasets"].values():
        assert savsr_amd.DATASET_REGISTRY.get(d["type"]).__name__ == d["type"]


def test_sample_yaml_registry_resolution():
    _resolves(parse_test_options(SAMPLE))
    with pytest.raises(KeyError):
        savsr_amd.MODEL_REGISTRY.get("NoSuchModel")


@pytest.mark.parametrize("name", ["test_SAVSR_Vid4_asBI.yml", "test_SAVSR_UDM10_asBI.yml"])
def test_reference_yaml_registry_resolution(name):
    path = os.path.join("/root/reference/options/test/SAVSR", name)
    if not os.path.isfile(path):
        pytest.skip("reference tree not present")
    _resolves(parse_test_options(path))


def test_model_and_dataset_refuse_cpu():
    """No CPU fallback behind the YAML surface either: without a GPU both constructors raise."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    opt = parse_test_options(SAMPLE)
    with pytest.raises(RuntimeError):
        savsr_amd.build_model(opt)
    with pytest.raises(RuntimeError):
        savsr_amd.build_dataset(opt["datasets"]["test_01"])
