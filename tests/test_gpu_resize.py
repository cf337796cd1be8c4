"""LR synthesis: anti-aliased bicubic resize on the GPU against torch's CPU implementation of
the op the reference calls (T.Resize(BICUBIC, antialias=True) -> F.interpolate(..., antialias=True)), tolerance 2e-6
(fp32 re-association; measured 4e-7), plus the size bookkeeping of arbitrary_scale_downsample."""
import pytest
import torch
import torch.nn.functional as F

from savsr_amd.utils import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w,scale", [(64, 80, (4, 4)), (72, 96, (1.5, 4)), (70, 64, (3.5, 2)), (66, 74, (3.7, 3.7)),
                                        (44, 52, (1.1, 1.1)), (59, 75, (2.95, 3.75)), (256, 448, (4, 4))])
def test_arbitrary_scale_downsample_vs_torch_cpu(h, w, scale):
    from savsr_amd.resize_gpu import arbitrary_scale_downsample
    x = synth.synth_clip(3, 3, h, w, seed=h + w)[0]                       # [t, c, h, w] in [0, 1)
    oh, ow = round(h / scale[0]), round(w / scale[1])
    ref = F.interpolate(x, size=(oh, ow), mode="bicubic", align_corners=False, antialias=True)
    got = arbitrary_scale_downsample(x.cuda(), scale).cpu()
    assert tuple(got.shape) == (3, 3, oh, ow)
    assert float((got - ref).abs().max()) < 2e-6


def test_batch_dims_and_upscale():
    from savsr_amd.resize_gpu import resize_bicubic_aa
    x = torch.rand(2, 2, 3, 20, 24)
    ref = F.interpolate(x.view(-1, 3, 20, 24), size=(31, 30), mode="bicubic", align_corners=False, antialias=True).view(2, 2, 3, 31, 30)
    got = resize_bicubic_aa(x.cuda(), (31, 30)).cpu()
    assert float((got - ref).abs().max()) < 2e-6


def test_full_size_vid4_frame():
    """7 x 3 x 720 x 1280 -> 180 x 320 (BASELINE config 2's LR clip from its ground truth)."""
    from savsr_amd.resize_gpu import arbitrary_scale_downsample
    x = torch.stack([synth.synth_gt(3, 720, 1280, seed=i) for i in range(2)], 0)
    ref = F.interpolate(x, size=(180, 320), mode="bicubic", align_corners=False, antialias=True)
    got = arbitrary_scale_downsample(x.cuda(), (4, 4)).cpu()
    assert float((got - ref).abs().max()) < 2e-6
