"""`downsampling_mode: core` (data_util.py:411-412 -> lbasicsr/data/core.py::imresize, a MATLAB-style antialiased bicubic with
border-repeating reflect padding): this repo's per-axis tables (savsr_amd/resize_gpu.py::core_tables), applied on the host here and by
the HIP gather kernel on the GPU, against outputs of the reference's own function (tests/golden/core_resize.npz)."""
import os

import numpy as np
import pytest
import torch

from tests.golden_cases import CORE_RESIZE_CASES, core_input

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 2e-6          # fp32; the folded border taps are summed in another order than core.py's (measured ~2e-7)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "core_resize.npz"))


def _apply(x, tb, axis):
    """dense application of one axis table (float64 accumulation: checks the TABLE, not the summation order)"""
    if tb is None:
        return x
    xmin, xsize, wt = tb
    n_in, n_out = x.shape[axis], len(xmin)
    m = np.zeros((n_out, n_in))
    for i in range(n_out):
        m[i, xmin[i]:xmin[i] + xsize[i]] = wt[i, :xsize[i]]
    return np.moveaxis(np.tensordot(m, np.moveaxis(x, axis, 0), axes=(1, 0)), 0, axis)


@pytest.mark.parametrize("name,c,h,w,sc", CORE_RESIZE_CASES)
def test_core_tables_vs_reference(gold, name, c, h, w, sc):
    from savsr_amd.resize_gpu import core_tables
    x = core_input(c, h, w).numpy().astype(np.float64)
    oh, ow = round(h / sc[0]), round(w / sc[1])
    y = _apply(_apply(x, core_tables(h, oh), 1), core_tables(w, ow), 2)             # height first, then width (core.py:438-439)
    assert y.shape == gold[name].shape
    assert float(np.abs(y - gold[name]).max()) < TOL
    ty = core_tables(h, oh)
    assert (ty is None) == (oh == h)                                                # identity axis (core.py:295-296)
    if ty is not None:
        assert np.allclose(ty[2].sum(1), 1.0, atol=1e-6) and int(ty[0].min()) >= 0 and int((ty[0] + ty[1]).max()) <= h


@pytest.mark.gpu
@pytest.mark.parametrize("name,c,h,w,sc", CORE_RESIZE_CASES)
def test_gpu_core_resize_vs_reference(gold, name, c, h, w, sc):
    from savsr_amd.resize_gpu import arbitrary_scale_downsample
    x = core_input(c, h, w)
    got = arbitrary_scale_downsample(x[None].cuda(), tuple(sc), mode="core")[0].cpu()      # [t=1, c, h, w] like a frame stack
    assert tuple(got.shape) == gold[name].shape
    assert float((got - torch.from_numpy(gold[name])).abs().max()) < TOL


@pytest.mark.gpu
def test_unknown_mode_raises():
    from savsr_amd.resize_gpu import arbitrary_scale_downsample
    with pytest.raises(ValueError, match="downsampling_mode"):
        arbitrary_scale_downsample(torch.rand(1, 3, 8, 8).cuda(), 2.0, mode="bilinear")
