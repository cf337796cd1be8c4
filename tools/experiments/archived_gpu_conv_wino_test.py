"""Parity of the Winograd F(2x2, 3x3) form of the 3x3 conv (csrc/conv_wino.hip, `algo = SAVSR_CONV_WINOGRAD`; an opt-in
EXPERIMENT of round 2, never selected by the engine -- DESIGN.md section 10) through the C ABI: against F.conv2d fp32 incl. the
fused epilogue, batched vs single launches, and the weight image against a float64 G g G^T.  Tolerance: 6e-5 max-abs on outputs
of magnitude ~4: the transformed operands are up to 4x larger than the raw ones and the output transform cancels, so the
bf16x3 products cost 2-2.5x the direct kernel's error (measured 2.3 - 4.2e-5 against 0.8 - 1.8e-5; printed with -rA)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_kernels import _dev, _maxerr, cl, eng, pl      # noqa: F401  (eng: module-scoped engine fixture)

pytestmark = pytest.mark.gpu

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)


def test_wino_weight_image(eng):
    """savsr_conv_wino_pack: hi + lo of every image element == G g G^T (float64) to 2^-16 relative, in the documented order."""
    g = np.random.RandomState(3)
    cout, cin = 128, 48
    w = torch.from_numpy(g.standard_normal((cout, cin, 3, 3)).astype(np.float32))
    img = eng.wino_weights(w, None)[0]
    torch.cuda.synchronize()
    v = img.view(torch.bfloat16).to(torch.float32).cpu().numpy().reshape(cout // 64, cin // 16, 4, 4, 2, 2, 2, 32, 8)   # [cob][chunk][i][j][ct][part][kh][row][8]
    u = np.einsum("ia,ocab,jb->ocij", G, w.numpy().astype(np.float64), G)                                             # [co][ci][i][j]
    got = (v[:, :, :, :, :, 0] + v[:, :, :, :, :, 1])                                                                 # [cob][chunk][i][j][ct][kh][row][8]
    got = got.transpose(0, 4, 6, 1, 5, 7, 2, 3).reshape(cout, cin, 4, 4)                                              # co = (cob, ct, row), ci = (chunk, kh, e)
    assert np.abs(got - u).max() < 2.0 ** -16 * np.abs(u).max()


@pytest.mark.parametrize("cin,cout,nsrc,h,w", [
    (64, 64, 1, 10, 12), (192, 64, 3, 9, 40), (128, 64, 2, 12, 33), (320, 128, 5, 8, 35), (16, 128, 1, 13, 31),
    (128, 64, 1, 19, 11), (64, 64, 1, 37, 70)])
def test_conv2d_winograd(eng, cin, cout, nsrc, h, w):
    from savsr_amd import engine as E
    from savsr_amd._lib import ACT_LRELU
    g = np.random.RandomState(cin * 7 + cout)
    wt = torch.from_numpy((g.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32))
    bias = torch.from_numpy(g.standard_normal(cout).astype(np.float32))
    x = torch.from_numpy(g.standard_normal((cin, h, w)).astype(np.float32))
    res = torch.from_numpy(g.standard_normal((cout, h, w)).astype(np.float32))
    res2 = torch.from_numpy(g.standard_normal((cout, h, w)).astype(np.float32))
    mul = torch.from_numpy(g.uniform(0, 1, (h, w)).astype(np.float32))
    ref = F.leaky_relu(F.conv2d(x[None], wt, bias, padding=1), 0.2)[0] * mul + res + 0.9 * res2
    sch = cin // nsrc
    xall = cl(x)
    srcs = [eng.full(xall, sch, i * sch) for i in range(nsrc)]
    wide = torch.full((h, w, cout + 4), float("nan"), device="cuda:0")
    out = E.Src(wide, cout, cout + 4, 4)
    rows = eng.pool_rows(h, w)
    part = torch.full((rows, cout), float("nan"), device="cuda:0")
    eng.conv("test", srcs, out, h, w, ACT_LRELU, 0.2, mul_px=_dev(mul), res1=eng.full(cl(res)), res2=eng.full(cl(res2)),
             res2_scale=0.9, weights=eng.wino_weights(wt, bias), pool=(part, 0, cout))
    torch.cuda.synchronize()
    assert bool(torch.isnan(wide[..., :4]).all())
    got = pl(wide[..., 4:])
    e = _maxerr(got, ref)
    print('winograd conv', cin, cout, 'max-abs', e)
    assert e < 6e-5
    assert _maxerr(part.sum(0).cpu() / (h * w), ref.mean(dim=(1, 2))) < 2e-5


def test_conv2d_winograd_batch(eng):
    """Six convs of one geometry in one launch (ragged 90x330), every workgroup walking several tiles: each vs F.conv2d and
    bit-identical to the same conv launched alone; no activation, no bias."""
    g = np.random.RandomState(78)
    h, w, cin, cout, n = 90, 330, 128, 64, 6
    keep, descs, singles, outs, outs1, refs = [], [], [], [], [], []
    for k in range(n):
        wt = torch.from_numpy((g.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32))
        x = torch.from_numpy(g.standard_normal((cin, h, w)).astype(np.float32))
        refs.append(F.conv2d(x[None], wt, None, padding=1)[0])
        xall = cl(x)
        srcs = [eng.full(xall, 64, 0), eng.full(xall, 64, 64)]
        weights = eng.wino_weights(wt, None)
        keep.append((xall, weights))
        for outl, dl in ((outs, descs), (outs1, singles)):
            o = torch.full((h, w, cout), float("nan"), device="cuda:0")
            outl.append(o)
            dl.append(eng.conv_desc("t", srcs, eng.full(o), h, w, weights=weights))
    eng.conv_launch(descs)
    for d in singles:
        eng.conv_launch([d])
    torch.cuda.synchronize()
    for k in range(n):
        e = _maxerr(pl(outs[k]), refs[k])
        print('winograd batch', k, e)
        assert e < 6e-5
        assert torch.equal(outs[k], outs1[k])
