"""Split-plane intermediates (include/savsr_hip.h, savsr_conv_desc.src_split / out_split): a conv that WRITES the format stores
exactly what a consumer's fp32 staging would compute (hi = bf16(v), lo = bf16(v - hi)); a conv that READS it gets its
activations by LDS-DMA.  Every combination must be BIT-identical to the fp32-tensor path -- per conv, and for the whole
network (savsr_arch.py:692-742 with ResidualBlock x1 / base and RCAB t1 in the format)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(synth_sd):
    from savsr_amd.engine import HipEngine
    from savsr_amd.archs.savsr_arch import SAVSR
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return HipEngine(synth_sd, SAVSR().cfg, torch.device("cuda:0"))


def to_split(x: torch.Tensor) -> torch.Tensor:
    """channel-last fp32 [h][w][C] (GPU) -> the same bytes' worth as a split-plane tensor, returned as an fp32-typed [h][w][C] buffer"""
    h, w, c = x.shape
    hi = x.to(torch.bfloat16)
    lo = (x - hi.to(torch.float32)).to(torch.bfloat16)
    planes = torch.stack([hi, lo], 0).reshape(2, h * w, c // 8, 8).permute(2, 0, 1, 3).contiguous()     # [C/8][part][px][8]
    return planes.view(torch.float32).reshape(h, w, c)


def from_split(buf: torch.Tensor) -> torch.Tensor:
    """inverse view: the (hi, lo) pairs of a split-plane buffer as two channel-last bf16 tensors"""
    h, w, c = buf.shape
    planes = buf.reshape(-1).view(torch.bfloat16).reshape(c // 8, 2, h * w, 8).permute(1, 2, 0, 3).reshape(2, h, w, c)
    return planes[0], planes[1]


def cl(x):
    return x.permute(1, 2, 0).contiguous().to("cuda:0")


CASES = [   # cin, cout, ks, nsrc, h, w, batch
    (64, 64, 3, 1, 10, 12, 1), (128, 64, 3, 2, 12, 33, 1), (192, 64, 3, 3, 9, 40, 2), (192, 64, 1, 3, 7, 50, 1),
    (128, 64, 3, 2, 90, 330, 4),            # 16-row tiles, ragged last band / column, three tiles per workgroup
    (192, 64, 1, 3, 100, 320, 2),           # 1x1, 16-row tiles
    (320, 128, 3, 5, 19, 35, 1),            # two output-channel blocks
]


@pytest.mark.parametrize("cin,cout,ks,nsrc,h,w,batch", CASES)
def test_conv_split_modes_are_bit_identical(eng, cin, cout, ks, nsrc, h, w, batch):
    from savsr_amd import engine as E
    from savsr_amd._lib import ACT_LRELU
    g = np.random.RandomState(cin + 3 * cout + h)
    sch = cin // nsrc
    rows = eng.pool_rows(h, w)
    keep, descs = [], {m: [] for m in ("ff", "sf", "fs", "ss")}
    outs = {m: [] for m in descs}
    pools = {m: [] for m in descs}
    for k in range(batch):
        wt = torch.from_numpy((g.standard_normal((cout, cin, ks, ks)) / np.sqrt(cin * ks * ks)).astype(np.float32))
        bias = torch.from_numpy(g.standard_normal(cout).astype(np.float32)).to("cuda:0")
        xs = [cl(torch.from_numpy(g.standard_normal((sch, h, w)).astype(np.float32))) for _ in range(nsrc)]
        xsp = [to_split(x) for x in xs]
        res = cl(torch.from_numpy(g.standard_normal((cout, h, w)).astype(np.float32)))
        weights = (E.pack_conv_weight(wt).to("cuda:0"), bias, cout, cin, ks)
        keep.append((xs, xsp, res, weights))
        for m in descs:
            o = torch.full((h, w, cout), float("nan"), device="cuda:0")
            pt = torch.full((rows, cout), float("nan"), device="cuda:0")
            srcs = [E.Src(t, sch, sch, split=(m[0] == "s")) for t in (xsp if m[0] == "s" else xs)]
            out = E.Src(o, cout, cout, split=(m[1] == "s"))
            descs[m].append(eng.conv_desc("t", srcs, out, h, w, ACT_LRELU, 0.2, res1=eng.full(res), weights=weights, pool=(pt, 0, cout)))
            outs[m].append(o)
            pools[m].append(pt)
    for m in descs:
        eng.conv_launch(descs[m])
    torch.cuda.synchronize()
    for k in range(batch):
        ref = outs["ff"][k]
        assert bool(torch.isfinite(ref).all())
        assert torch.equal(outs["sf"][k], ref), "pre-split sources must give the bits of the fp32 staging path"
        want = to_split(ref)
        assert torch.equal(outs["fs"][k].view(torch.int32), want.view(torch.int32)), "split-plane output != split of the fp32 output"
        assert torch.equal(outs["ss"][k].view(torch.int32), want.view(torch.int32))
        for m in ("sf", "fs", "ss"):
            assert torch.equal(pools[m][k], pools["ff"][k]), "fused pool rows are taken from the fp32 values"
        hi, lo = from_split(outs["ss"][k])
        assert float((hi.float() + lo.float() - ref).abs().max()) <= float(ref.abs().max()) * 2.0 ** -15


def test_split_flags_are_validated(eng):
    from savsr_amd import engine as E
    g = np.random.RandomState(1)
    wt = torch.from_numpy((g.standard_normal((16, 64, 3, 3)) / 24.0).astype(np.float32))
    x = cl(torch.from_numpy(g.standard_normal((64, 10, 12)).astype(np.float32)))
    o = torch.empty(10, 12, 16, device="cuda:0")
    w16 = (E.pack_conv_weight(wt).to("cuda:0"), None, 16, 64, 3)
    d = eng.conv_desc("t", [E.Src(x, 64, 64, split=True)], eng.full(o), 10, 12, weights=w16)       # 16 output channels: the narrow kernel
    assert eng.lib.savsr_conv2d(C.byref(d), None) < 0 and b"split" in eng.lib.savsr_last_error()
    d = eng.conv_desc("t", [eng.full(x)], E.Src(o, 16, 16, split=True), 10, 12, weights=w16)
    assert eng.lib.savsr_conv2d(C.byref(d), None) < 0 and b"split" in eng.lib.savsr_last_error()
    # a batch mixes fp32 and pre-split sources
    wt64 = torch.from_numpy((g.standard_normal((64, 64, 3, 3)) / 24.0).astype(np.float32))
    w64 = (E.pack_conv_weight(wt64).to("cuda:0"), None, 64, 64, 3)
    o64 = torch.empty(10, 12, 64, device="cuda:0")
    da = eng.conv_desc("t", [eng.full(x)], eng.full(o64), 10, 12, weights=w64)
    db = eng.conv_desc("t", [E.Src(x, 64, 64, split=True)], eng.full(o64), 10, 12, weights=w64)
    from savsr_amd._lib import ConvDesc
    arr = (ConvDesc * 2)(da, db)
    assert eng.lib.savsr_conv2d_batch(arr, 2, None) < 0 and b"src_split" in eng.lib.savsr_last_error()


@pytest.mark.parametrize("shape,scale", [((36, 44), (4.0, 4.0)), ((17, 21), (2.5, 3.5)), ((64, 96), (1.5, 4.0))])
def test_network_is_bit_identical_with_and_without_presplit(synth_sd, shape, scale):
    """The whole forward with ResidualBlock x1 / base and RCAB t1 in the split-plane format == with fp32 tensors, bit for bit
    (eager and through the captured graphs)."""
    from savsr_amd.engine import HipEngine
    from savsr_amd.archs.savsr_arch import SAVSR
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(shape[0])
    lq = torch.rand(1, 7, 3, *shape, generator=g).to(dev)
    outs = {}
    old = os.environ.get("SAVSR_PRESPLIT")
    try:
        for flag in ("0", "1", "3"):                    # none | RCAB t1 | + ResidualBlock x1 / base
            os.environ["SAVSR_PRESPLIT"] = flag
            e = HipEngine(synth_sd, SAVSR().cfg, dev)
            assert e.presplit == int(flag)
            a = e.forward(lq, scale).clone()
            b = e.forward(lq, scale).clone()            # second call: graph replay
            torch.cuda.synchronize()
            assert torch.equal(a, b)
            outs[flag] = a
    finally:
        if old is None:
            os.environ.pop("SAVSR_PRESPLIT", None)
        else:
            os.environ["SAVSR_PRESPLIT"] = old
    assert bool(torch.isfinite(outs["1"]).all())
    assert torch.equal(outs["0"], outs["1"]) and torch.equal(outs["0"], outs["3"])
