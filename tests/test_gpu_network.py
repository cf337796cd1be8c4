"""End-to-end parity of the HIP path (through the ARCH_REGISTRY surface) with the reference
goldens and the CPU oracle.  `pytest -m gpu` on the MI355X box."""
import numpy as np
import pytest
import torch

from oracle import savsr_oracle as O
from savsr_amd.utils import synth
from tests.golden_cases import NET_CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net(synth_sd):
    import savsr_amd
    assert torch.cuda.is_available()
    n = savsr_amd.build_network(dict(type="SAVSR", num_in_ch=3, num_feat=64, num_frame=7, slid_win=3, fusion_win=5,
                                     interval=0, w1_num_block=4, w2_num_block=2, n_resgroups=4, n_resblocks=8,
                                     center_frame_idx=None)).eval()
    n.load_state_dict(synth_sd, strict=True)
    return n.to("cuda:0")


@pytest.mark.parametrize("name,h,w,sc", NET_CASES)
def test_network_vs_reference_golden(net, golden, name, h, w, sc):
    """fp32 output within 5e-5 max-abs of the REFERENCE's own output (tests/golden); SATU tap within 2e-5 x its magnitude (~4)."""
    lq = synth.synth_clip(7, 3, h, w, seed=0)
    net.set_scale(sc)
    taps = {}
    out = net(lq.to("cuda:0"), taps=taps)
    torch.cuda.synchronize()
    gold = torch.from_numpy(golden[f"net/{name}/sr"])
    assert tuple(out.shape) == tuple(gold.shape)
    err = float((out.cpu() - gold).abs().max())
    gs = torch.from_numpy(golden[f"net/{name}/satu_s"])[0]
    satu_err = float((taps["satu"].cpu()[::4, ::3, ::3] - gs).abs().max()) / float(gs.abs().max())
    print(name, "max-abs", err, "satu (relative to its magnitude %.1f)" % float(gs.abs().max()), satu_err)
    assert satu_err < 2e-5 and err < 5e-5          # the tap carries the trunk's accumulated error; the kernel alone: test_gpu_kernels


def test_stage_taps_vs_oracle(net, synth_sd):
    """Intermediate tensors (propagation output, trunk output, SATU) against the oracle."""
    h, w, sc = 15, 18, (2.5, 3.5)
    lq = synth.synth_clip(7, 3, h, w, seed=9)
    net.set_scale(sc)
    taps, otaps = {}, {}
    out = net(lq.to("cuda:0"), taps=taps)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.forward(synth_sd, lq, sc, taps=otaps)
    for k in ("align_feat", "h_feat", "satu"):
        got = taps[k].cpu()
        if k != "satu":                       # channel-last [hp][wp][64] -> planar
            got = got.permute(2, 0, 1)
        mag = float(otaps[k][0].abs().max())
        e = float((got - otaps[k][0]).abs().max()) / mag
        print(k, "relative max-abs", e, "magnitude", mag)
        assert e < 2e-5, (k, e)            # relative to the tensor's maximum (measured 1.2e-5 after 50 convs)
    assert float((out.cpu() - ref).abs().max()) < 5e-5


def test_batch_and_determinism(net, synth_sd):
    lq = torch.cat([synth.synth_clip(7, 3, 12, 14, seed=1), synth.synth_clip(7, 3, 12, 14, seed=2)], 0)
    net.set_scale((2, 2))
    a = net(lq.to("cuda:0")).cpu()
    b = net(lq.to("cuda:0")).cpu()
    assert torch.equal(a, b), "kernels are deterministic: bitwise identical reruns"
    with torch.no_grad():
        ref = O.forward(synth_sd, lq[1:], (2, 2))
    assert float((a[1:] - ref).abs().max()) < 5e-5


def test_psnr_ssim_tolerance(net, synth_sd):
    """north_star tolerance: |dPSNR-Y| <= 1e-3 dB and |dSSIM-Y| <= 1e-4 vs the oracle, same GT."""
    from savsr_amd.metrics import calculate_psnr, calculate_ssim, tensor2img
    h, w, sc = 32, 40, (4, 4)
    lq = synth.synth_clip(7, 3, h, w, seed=3)
    gt = synth.synth_gt(3, 128, 160, seed=3)
    net.set_scale(sc)
    out = net(lq.to("cuda:0")).cpu()
    with torch.no_grad():
        ref = O.forward(synth_sd, lq, sc)
    gi = tensor2img(gt)
    p1, p2 = calculate_psnr(tensor2img(out[0]), gi, 0, test_y_channel=True), calculate_psnr(tensor2img(ref[0]), gi, 0, test_y_channel=True)
    s1, s2 = calculate_ssim(tensor2img(out[0]), gi, 0, test_y_channel=True), calculate_ssim(tensor2img(ref[0]), gi, 0, test_y_channel=True)
    print("psnr", p1, p2, "ssim", s1, s2)
    assert abs(p1 - p2) <= 1e-3 and abs(s1 - s2) <= 1e-4


def test_full_size_config2(net, synth_sd, oracle_config2):
    """BASELINE config 2 (7x3x180x320, x4 -> 720x1280): shape, finiteness, bitwise rerun
    determinism, and parity with the CPU oracle on the same clip (max-abs, PSNR-Y, SSIM-Y)."""
    from savsr_amd.metrics import calculate_psnr, calculate_ssim, tensor2img
    lq = synth.synth_clip(7, 3, 180, 320, seed=0)
    net.set_scale((4, 4))
    a = net(lq.to("cuda:0"))
    assert tuple(a.shape) == (1, 3, 720, 1280) and bool(torch.isfinite(a).all())
    b = net(lq.to("cuda:0"))
    assert torch.equal(a, b)
    ref = oracle_config2(0)
    err = float((a.cpu() - ref).abs().max())
    gt = tensor2img(synth.synth_gt(3, 720, 1280, seed=0))
    dp = abs(calculate_psnr(tensor2img(a[0].cpu()), gt, 0, test_y_channel=True) - calculate_psnr(tensor2img(ref[0]), gt, 0, test_y_channel=True))
    ds = abs(calculate_ssim(tensor2img(a[0].cpu()), gt, 0, test_y_channel=True) - calculate_ssim(tensor2img(ref[0]), gt, 0, test_y_channel=True))
    print("config2 max-abs", err, "dPSNR", dp, "dSSIM", ds)
    assert err < 5e-5 and dp <= 1e-3 and ds <= 1e-4


def test_headline_configuration_vs_oracle(net, oracle_config2):
    """The configuration bench.py's headline times (savsr_arch.py:692-742 on a batch, video_base_model.py:51-53 with several frames in
    flight): `net(lq)` with b = 16 at 180x320 x4 under product defaults -- two HIP streams x four clips per launch sequence (two units per
    stream), throughput conv tiling, ~99 % of the conv MACs in the Winograd-y form.  Three distinct clips spread over the streams and over
    the positions inside a batched launch: every output < 5e-5 from the oracle, and all copies of a clip bit-identical (a clip's result
    does not depend on the stream, on its position in the launch sequence or on the clips it shares the launches with)."""
    eng = net.engine()
    assert eng.streams_for(180 * 320) == 2 and eng.clip_batch == 4 and eng.use_graphs and eng.conv_wy and eng.knobs.knobs() == {}, "product defaults"
    seeds = [0, 1, 2]
    clips = {s: synth.synth_clip(7, 3, 180, 320, seed=s) for s in seeds}
    order = [0, 1, 2, 0, 1, 2, 0, 1, 2, 1, 0, 2, 2, 0, 1, 0]
    lq = torch.cat([clips[s] for s in order], 0).to("cuda:0")
    net.set_scale((4, 4))
    eng.census = {}
    try:
        out = net(lq)
        out2 = net(lq)          # replay
        torch.cuda.synchronize()
        cen = dict(eng.census)
    finally:
        eng.census = None
    assert tuple(out.shape) == (16, 3, 720, 1280) and torch.equal(out, out2)
    if cen.get("alg_tp"):       # (filled while the launch sequences are captured: empty when an earlier test already captured this context)
        assert cen["frames_tp"] % 4 == 0 and cen["wy_alg_tp"] / cen["alg_tp"] > 0.98, cen
    first = {}
    for i, s in enumerate(order):
        err = float((out[i].cpu() - oracle_config2(s)[0]).abs().max())
        print("headline configuration: clip", i, "seed", s, "max-abs vs oracle", err)
        assert err < 5e-5, (i, s, err)
        if s in first:
            assert torch.equal(out[i], out[first[s]]), (i, first[s])
        first.setdefault(s, i)


def test_forward_many_group_of_seven_vs_oracle(net, synth_sd):
    """A small-clip group as the YAML workflow / config 5 hands it over: forward_many with 7 clips of 64x112 at x2 -> launch units of
    4 + 3 clips (up to four clips per launch sequence).  Every clip < 5e-5 from the oracle; equal clips bit-identical whatever unit they ran in; and each
    clip's result equals that of forward_many called with that clip alone (the conv form of a launch is chosen by the flow, not the group)."""
    sc = (2, 2)
    clips = [synth.synth_clip(7, 3, 64, 112, seed=20 + k) for k in range(3)]
    order = [0, 1, 2, 0, 1, 2, 0]
    items = [clips[k][0].to("cuda:0") for k in order]
    outs = net.forward_many(items, [sc] * 7)
    torch.cuda.synchronize()
    refs = []
    with torch.no_grad():
        for c in clips:
            refs.append(O.forward(synth_sd, c, sc)[0])
    for i, k in enumerate(order):
        err = float((outs[i].cpu() - refs[k]).abs().max())
        print("group of seven: clip", i, "max-abs vs oracle", err)
        assert err < 5e-5, (i, err)
        assert torch.equal(outs[i], outs[order.index(k)]), i
    for k in range(3):
        alone = net.forward_many([items[k]], [sc])[0]
        pair = net.forward_many([items[k], items[(k + 1) % 3]], [sc] * 2)[0]
        torch.cuda.synchronize()
        assert torch.equal(alone, outs[k]) and torch.equal(pair, outs[k]), k


def test_full_size_satu_linearity(net):
    """Size-independent property at 180x320 -> 720x1280: for fixed st_feat, SATU is affine in x
    (dynamic filters depend on st only, offsets/routing on coordinates only)."""
    eng = net.engine()
    g = torch.Generator(device="cpu").manual_seed(0)
    x1, x2, st = (torch.randn(180, 320, 64, generator=g).to("cuda:0") for _ in range(3))
    z = torch.zeros_like(x1)
    outs = []
    for x in (x1, x2, 0.5 * x1 - 2.0 * x2, z):
        o = torch.empty(64, 720, 1280, device="cuda:0")
        eng.satu(eng.full(x), eng.full(st), 320, 180, 320, (4, 4), o)
        outs.append(o)
    torch.cuda.synchronize()
    y1, y2, y3, y0 = outs
    lin = 0.5 * (y1 - y0) - 2.0 * (y2 - y0) + y0
    rel = float((y3 - lin).abs().max() / y3.abs().max())
    print("satu linearity rel err", rel)
    assert rel < 5e-5


def test_buffer_liveness_plan_bitwise_and_footprint(synth_sd):
    """The liveness-planned LR buffers (HipEngine.release / seal_buffers): outputs bit for bit those of an engine in which every named
    buffer has its own memory (SAVSR_REUSE_BUFFERS=0), over several frames (the plan is made on a shape's first frame and replayed),
    on two shapes and with the graphs off as well; and the footprint of a 180x320 shape is a third of the unshared one."""
    import os
    import savsr_amd

    def build(reuse, graphs="1"):
        os.environ["SAVSR_REUSE_BUFFERS"], os.environ["SAVSR_GRAPHS"] = reuse, graphs
        try:
            n = savsr_amd.build_network(dict(type="SAVSR")).eval()
            n.load_state_dict(synth_sd, strict=True)
            n = n.to("cuda:0")
            n.engine()                                        # (the engine reads the switches when it is built)
            return n
        finally:
            os.environ.pop("SAVSR_REUSE_BUFFERS", None)
            os.environ.pop("SAVSR_GRAPHS", None)
    nets = {"shared": build("1"), "own": build("0"), "shared_eager": build("1", "0")}
    for (h, w, sc) in ((36, 40, (2.5, 3.5)), (21, 50, (4, 4)), (36, 40, (2.5, 3.5))):
        outs = {}
        for name, n in nets.items():
            n.set_scale(sc)
            frames = [n(synth.synth_clip(7, 3, h, w, seed=s).to("cuda:0")).clone() for s in (0, 1, 0)]
            torch.cuda.synchronize()
            assert torch.equal(frames[0], frames[2])
            outs[name] = frames
        for k in range(3):
            assert torch.equal(outs["shared"][k], outs["own"][k]) and torch.equal(outs["shared_eager"][k], outs["own"][k])
    big = {}
    for name in ("shared", "own"):
        n = nets[name]
        n.set_scale((4, 4))
        n(synth.synth_clip(7, 3, 180, 320, seed=0).to("cuda:0"))
        torch.cuda.synchronize()
        eng = n.engine()
        ctx = eng._ctx[(7, 3, 180, 320)] if (7, 3, 180, 320) in eng._ctx else list(eng._ctx.values())[-1]
        big[name] = sum(a[1] for a in ctx.get("arena", []))     # bytes bump-allocated for this shape's LR buffers
    print("LR buffer bytes per 180x320 shape: shared %.3f GB, own %.3f GB" % (big["shared"] / 1e9, big["own"] / 1e9))
    assert big["shared"] < 0.45 * big["own"]


def test_eager_first_frames_then_capture_bitwise(synth_sd, monkeypatch):
    """SAVSR_CAPTURE_AFTER = 4 (default 0 = capture on the first visit): a (shape, scale) context's first four frames are launched eagerly, the
    fifth is captured into hipGraphs and replayed from then on -- the same launch sequence either way, so every frame equals the
    capture-at-once engine's bit for bit, one clip in flight and three (forward_many)."""
    import savsr_amd

    def build(after):
        monkeypatch.setenv("SAVSR_CAPTURE_AFTER", after)
        monkeypatch.setenv("SAVSR_CLIP_BATCH", "1")              # (one clip per launch sequence: the counts below are per clip)
        n = savsr_amd.build_network(dict(type="SAVSR")).eval()
        n.load_state_dict(synth_sd, strict=True)
        n = n.to("cuda:0")
        n.engine()
        return n
    lazy, now = build("4"), build("0")
    assert lazy.engine().capture_after == 4 and now.engine().capture_after == 0
    h, w, sc = 34, 46, (3.3, 2.5)
    for n in (lazy, now):
        n.set_scale(sc)
    for k in range(7):                                           # frames 0-3 eager, frame 4 captured, 5-6 replayed
        lq = synth.synth_clip(7, 3, h, w, seed=20 + k).to("cuda:0")
        assert torch.equal(lazy(lq), now(lq)), k
    st = lazy.engine().host_stats
    assert st["eager_frames"] == 4 and st["captures"] == 1, st
    clips = [synth.synth_clip(7, 3, h, w, seed=40 + k)[0].to("cuda:0") for k in range(18)]
    for k0 in range(0, 18, 6):                                   # 3 streams x 2 clips per call: each stream's context is visited 6 times
        a = lazy.forward_many(clips[k0:k0 + 6], [sc] * 6)
        b = now.forward_many(clips[k0:k0 + 6], [sc] * 6)
        torch.cuda.synchronize()
        assert all(torch.equal(x, y) for x, y in zip(a, b)), k0
    st = lazy.engine().host_stats
    assert st["eager_frames"] == 4 + 3 * 4 and st["captures"] == 1 + 3, st


def test_clip_batched_launch_sequence_bitwise(synth_sd, monkeypatch):
    """SAVSR_CLIP_BATCH: clips of one (shape, scale) run as ONE launch sequence (every named buffer holds a copy per clip, every conv / OSConv
    descriptor goes out once per clip inside the same batched launch, the per-clip kernels are looped).  The convs of a batched launch are
    independent, so with ONE conv form everywhere (SAVSR_CONV_WY=0) every clip's output equals the unbatched stream's bit for bit -- groups of
    3 and 2, a lone clip, odd sizes, an expanded-table scale, captured and replayed; with the product's per-launch form choice the form is a function of the flow and of
    SAVSR_CLIP_BATCH (`form_nb`), so the clip_batch = 3 engine agrees with the clip_batch = 1 engine to the forms' rounding, and with ITSELF bit for
    bit under any grouping (alone, pairs, the batched stream)."""
    import random
    import savsr_amd

    def build(cb, wy):
        monkeypatch.setenv("SAVSR_CLIP_BATCH", cb)
        monkeypatch.setenv("SAVSR_CLIP_BATCH_MAX_PX", str(1 << 30))
        monkeypatch.setenv("SAVSR_CONV_WY", wy)
        n = savsr_amd.build_network(dict(type="SAVSR")).eval()
        n.load_state_dict(synth_sd, strict=True)
        n = n.to("cuda:0")
        n.engine()
        return n
    cases = [(33, 46, (3.3, 2.5), 5), (24, 40, (4.0, 4.0), 3), (30, 44, (1.5, 4.0), 1), (21, 50, (3.7, 3.7), 4)]      # (h, w, scale, clips): groups 3+2, 3, 1, 3+1
    clips, scales = [], []
    for h, w, sc, n in cases:
        for k in range(n):
            clips.append(synth.synth_clip(7, 3, h, w, seed=100 * h + k)[0].to("cuda:0"))
            scales.append(sc)
    order = list(range(len(clips)))
    random.Random(1).shuffle(order)                              # equal (shape, scale) clips need not be adjacent
    clips, scales = [clips[i] for i in order], [scales[i] for i in order]
    for wy, tol in (("0", 0.0), ("1", 3e-5)):
        one, three = build("1", wy), build("3", wy)
        assert one.engine().clip_batch == 1 and three.engine().clip_batch == 3
        for rep in range(3):                                     # capture, replay, replay
            a = three.forward_many(clips, scales)
            b = one.forward_many(clips, scales)
            torch.cuda.synchronize()
            for i in range(len(clips)):
                assert tuple(a[i].shape) == tuple(b[i].shape)
                err = float((a[i] - b[i]).abs().max())
                assert err <= tol, (wy, rep, i, tuple(clips[i].shape), scales[i], err)
        assert three.engine().host_stats["captures"] < one.engine().host_stats["captures"]      # fewer, fatter launch sequences
        # With ONE setting of SAVSR_CLIP_BATCH a clip's result does not depend on its group (the form rule counts `form_nb` clips, not the
        # nb at hand): the batched stream against every clip alone and against pairs, bit for bit, in both form settings
        for i in range(len(clips)):
            alone = three.forward_many([clips[i]], [scales[i]])[0]
            torch.cuda.synchronize()
            assert torch.equal(alone, a[i]), (wy, i, tuple(clips[i].shape), scales[i])
        for i in range(0, len(clips) - 1, 2):
            pair = three.forward_many(clips[i:i + 2], scales[i:i + 2])
            torch.cuda.synchronize()
            assert torch.equal(pair[0], a[i]) and torch.equal(pair[1], a[i + 1]), (wy, i)
