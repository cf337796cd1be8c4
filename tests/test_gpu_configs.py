"""BASELINE configs 3, 4, 5 at their working sizes (the arbitrary-scale purpose of the model): every phase-table regime
of the SATU HR stage (4 ... 24 360 distinct coordinate pairs at 180x320), the UDM10 asymmetric shapes, and a seeded sample
of the Vimeo90K training (shape, scale) list -- shape per the reference's get_HW, finiteness, bitwise rerun, and max-abs
against the CPU oracle (config 5: 8 seeded draws + every third entry of the 60-entry list).  Tolerance: 5e-5 max-abs on outputs of magnitude ~1 (fp32 re-association + split-bf16 products;
measured <= 1.5e-5), far inside north_star's 1e-3 dB PSNR."""
import pytest
import torch

from oracle import savsr_oracle as O
from savsr_amd.engine import get_hw, satu_axis_tables
from savsr_amd.utils import synth, workloads

pytestmark = pytest.mark.gpu

TOL = 5e-5


@pytest.fixture(scope="module")
def net(synth_sd):
    import savsr_amd
    n = savsr_amd.build_network(dict(type="SAVSR")).eval()
    n.load_state_dict(synth_sd, strict=True)
    return n.to("cuda:0")


def _run(net, lq, sc):
    net.set_scale(sc)
    a = net(lq.to("cuda:0"))
    b = net(lq.to("cuda:0"))
    torch.cuda.synchronize()
    h, w = lq.shape[-2:]
    H, W = round(h * sc[0]), round(w * sc[1])                   # savsr_arch.py:745-751 (Python round)
    assert tuple(a.shape) == (1, 3, H, W) == (1, 3) + get_hw(h, w, sc)
    assert bool(torch.isfinite(a).all())
    assert torch.equal(a, b), "bitwise identical reruns"
    return a.cpu()


# scale -> compare with the oracle?  (the oracle costs seconds per 180x320 frame on the box's host cores)
# Against the oracle at full size: every 5th scale of the 30-scale list (x1.1, 1.6, 2.1, 2.6, 3.1, 3.6) plus x1.5 / x2.5 / x3.7 (phase-table
# regimes) -- 9 of 30; x4 is test_full_size_config2; the other scales run the property checks (test_config3_all_30_scales_...).
CONFIG3 = [((1.1, 1.1), True), ((1.5, 1.5), True), ((1.6, 1.6), True), ((2, 2), False), ((2.1, 2.1), True), ((2.5, 2.5), True), ((2.6, 2.6), True),
           ((3, 3), False), ((3.1, 3.1), True), ((3.6, 3.6), True), ((3.7, 3.7), True), ((3.9, 3.9), False)]


@pytest.mark.parametrize("sc,vs_oracle", CONFIG3)
def test_config3_vid4_sweep_full_size(net, synth_sd, sc, vs_oracle):
    lq = synth.synth_clip(7, 3, 180, 320, seed=0)
    out = _run(net, lq, sc)
    if vs_oracle:
        with torch.no_grad():
            ref = O.forward(synth_sd, lq, sc)
        err = float((out - ref).abs().max())
        print("config3", sc, "max-abs vs oracle", err)
        assert err < TOL


@pytest.mark.parametrize("sc", workloads.VID4_ASYM_SCALES)
def test_vid4_asymmetric_yaml_scales_half_size(net, synth_sd, sc):
    """The 12 asymmetric pairs of the shipped YAMLs (Vid4.yml:518-826), at half the Vid4 LR size (90 x 160) so the suite stays inside the
    driver's budget; all 42 YAML pairs at 180 x 320 and all 60 training pairs at their sizes: tools/scale_list_sweep.py ->
    profiles/r05_scale_lists_vs_oracle.log."""
    lq = synth.synth_clip(7, 3, 90, 160, seed=12)
    out = _run(net, lq, sc)
    with torch.no_grad():
        ref = O.forward(synth_sd, lq, sc)
    err = float((out - ref).abs().max())
    print("vid4 asymmetric", sc, "max-abs vs oracle", err)
    assert err < TOL


def test_config3_table_regimes_are_covered():
    """The sweep above really spans the phase-table sizes of the 30-scale list (host arithmetic only)."""
    import numpy as np
    n = {}
    for sc, _ in CONFIG3:
        H, W = get_hw(180, 320, sc)
        n[sc[0]] = len(np.unique(satu_axis_tables(H, 180, sc[0])[0])) * len(np.unique(satu_axis_tables(W, 320, sc[1])[0]))
    assert n[2] == 4 and n[1.5] > 256 and n[1.1] > 1500 and n[3.7] > 20000 and n[3.9] > 24000, n


@pytest.mark.parametrize("gt_hw,sc", workloads.CONFIG4_CASES)
def test_config4_udm10_shapes_full_size(net, synth_sd, gt_hw, sc):
    """720x1272 GT: x(1.5, 4) -> LR 480x318 -> 720x1272; x(3.5, 2) -> GT crop 714x1272, LR 204x636 -> 714x1272."""
    h, w = workloads.lr_shape(gt_hw, sc)
    assert (h, w) == {(1.5, 4.0): (480, 318), (3.5, 2.0): (204, 636)}[sc]
    lq = synth.synth_clip(7, 3, h, w, seed=4)
    out = _run(net, lq, sc)
    assert tuple(out.shape[-2:]) == {(1.5, 4.0): (720, 1272), (3.5, 2.0): (714, 1272)}[sc]
    with torch.no_grad():
        ref = O.forward(synth_sd, lq, sc)
    err = float((out - ref).abs().max())
    print("config4", sc, (h, w), "max-abs vs oracle", err)
    assert err < TOL


@pytest.mark.parametrize("h,w,sc", workloads.config5_cases(8, seed=0))
def test_config5_vimeo_training_shapes(net, synth_sd, h, w, sc):
    lq = synth.synth_clip(7, 3, h, w, seed=5)
    out = _run(net, lq, sc)
    with torch.no_grad():
        ref = O.forward(synth_sd, lq, sc)
    err = float((out - ref).abs().max())
    print("config5", (h, w), sc, "max-abs vs oracle", err)
    assert err < TOL


_C5_SEEDED = set(workloads.config5_cases(8, seed=0))
_C5_SYSTEMATIC = [(workloads.lr_shape(workloads.VIMEO_GT, sc) + (sc,)) for sc in workloads.TRAIN_SCALES[::3]]


@pytest.mark.parametrize("h,w,sc", [c for c in _C5_SYSTEMATIC if c not in _C5_SEEDED])
def test_config5_every_third_training_scale(net, synth_sd, h, w, sc):
    """beside the 8 seeded draws above, every third entry of the 60-entry training scale list (10 symmetric, 10 asymmetric
    pairs; LR 64 x 112 ... 230 x 400) against the oracle: up to 28 of the 60 (shape, scale) pairs at their working sizes."""
    lq = synth.synth_clip(7, 3, h, w, seed=7)
    out = _run(net, lq, sc)
    with torch.no_grad():
        ref = O.forward(synth_sd, lq, sc)
    err = float((out - ref).abs().max())
    print("config5 (systematic)", (h, w), sc, "max-abs vs oracle", err)
    assert err < TOL


def test_mixed_scale_stream_reuses_engine(net):
    """Config 5 is a STREAM of mixed (shape, scale) clips through one engine: results must not depend on what ran before
    (per-(shape, scale) graphs / buffers / tables are cached and evicted, never shared wrongly)."""
    cases = workloads.config5_cases(6, seed=1)
    first = {}
    for h, w, sc in cases + cases[::-1]:
        lq = synth.synth_clip(7, 3, h, w, seed=h * 1000 + w)
        net.set_scale(sc)
        out = net(lq.to("cuda:0")).cpu()
        key = (h, w, sc)
        if key in first:
            assert torch.equal(first[key], out)
        first[key] = out


def test_live_graphs_keep_their_satu_tables(synth_sd, monkeypatch):
    """A captured hipGraph bakes in the device pointers of its (size, scale)'s SATU tables.  More live (shape, scale) graphs
    than the table LRU holds (here 3 shapes x 2 scales = 6 graphs, table cap max(shapes, scales) = 4) must not let a table
    be freed under a graph that can still be replayed: revisiting the oldest pairs replays their graphs bitwise."""
    import savsr_amd
    monkeypatch.setenv("SAVSR_CACHE_SHAPES", "4")
    monkeypatch.setenv("SAVSR_CACHE_SCALES", "2")
    n = savsr_amd.build_network(dict(type="SAVSR")).eval()
    n.load_state_dict(synth_sd, strict=True)
    n = n.to("cuda:0")
    shapes = [(24, 40), (32, 48), (40, 56)]
    scales = [(2.5, 2.5), (3.7, 3.7)]             # one small-table and one expanded-table regime
    first = {}
    for h, w in shapes:
        lq = synth.synth_clip(7, 3, h, w, seed=h)
        for sc in scales:
            n.set_scale(sc)
            first[(h, w, sc)] = n(lq.to("cuda:0")).cpu()
    eng = n.engine()
    assert eng.cache_stats()["scales"] == 6 and eng.cache_stats()["axes"] <= 4
    lq4 = synth.synth_clip(7, 3, 48, 64, seed=9)  # a fourth shape: its tables land in whatever the evicted ones freed
    for sc in [(1.7, 1.7), (3.3, 3.3)]:
        n.set_scale(sc)
        n(lq4.to("cuda:0"))
    for h, w in shapes:
        lq = synth.synth_clip(7, 3, h, w, seed=h)
        for sc in scales:
            n.set_scale(sc)
            assert torch.equal(n(lq.to("cuda:0")).cpu(), first[(h, w, sc)]), (h, w, sc)


def _eviction_stream(synth_sd, monkeypatch, env):
    """A forward_many stream of 12 distinct (shape, scale) pairs, each visited three times in shuffled order, through an engine whose caches are
    far smaller than the working set: contexts are evicted while the sibling streams replay theirs, and re-captured on the next visit."""
    import random
    import savsr_amd
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = savsr_amd.build_network(dict(type="SAVSR")).eval()
    n.load_state_dict(synth_sd, strict=True)
    n = n.to("cuda:0")
    scs = [workloads.TRAIN_SCALES[i] for i in (0, 9, 19, 29, 31, 36, 41, 44, 49, 52, 55, 58)]       # 4 symmetric + 8 asymmetric, x1.1 ... x4
    pairs = [(22 + 2 * i, 30 + 3 * i, sc) for i, sc in enumerate(scs)]                                # 12 distinct LR shapes (odd widths too)
    assert len({(h, w) for h, w, _ in pairs}) == 12
    rng = random.Random(3)
    visits = []
    for _ in range(3):
        order = pairs[:]
        rng.shuffle(order)
        visits += order
    clip = lambda h, w: synth.synth_clip(7, 3, h, w, seed=h * 100 + w)[0]
    first = {}
    for i in range(0, len(visits), 6):                                                              # 6 clips per call: 2 per stream
        chunk = visits[i:i + 6]
        outs = n.forward_many([clip(h, w).to("cuda:0") for h, w, _ in chunk], [sc for _, _, sc in chunk])
        torch.cuda.synchronize()
        for key, o in zip(chunk, outs):
            o = o.cpu()
            assert tuple(o.shape) == (3,) + get_hw(key[0], key[1], key[2]) and bool(torch.isfinite(o).all())
            if key in first:
                assert torch.equal(first[key], o), ("a re-captured context differs from its first visit", key)
            first[key] = o
    return n, pairs, first, clip


def test_eviction_by_count_while_sibling_streams_replay(synth_sd, monkeypatch):
    """SAVSR_CACHE_SHAPES=3 against 12 shapes, 3 streams.  Every output bitwise equal to its first visit, a sample of
    them against the CPU oracle, and the engines really evicted."""
    n, pairs, first, clip = _eviction_stream(synth_sd, monkeypatch, {"SAVSR_CACHE_SHAPES": "3", "SAVSR_STREAMS": "3"})
    eng = n.engine()
    st = eng.cache_stats()
    assert st["shapes"] <= 3 and st["evictions"] >= 12, st
    for h, w, sc in pairs[::4]:
        with torch.no_grad():
            ref = O.forward(synth_sd, clip(h, w).unsqueeze(0), sc)
        err = float((first[(h, w, sc)] - ref[0]).abs().max())
        print("eviction stream", (h, w), sc, "max-abs vs oracle", err)
        assert err < TOL


def test_eviction_by_byte_budget(synth_sd, monkeypatch):
    """The byte budget (SAVSR_CACHE_GB, shared by the stream engines) is what evicts by default: 0.2 GB holds two or three of these small
    contexts (one 64-MiB arena chunk each); same bitwise / oracle checks, and the account never exceeds budget + the context in use."""
    n, pairs, first, clip = _eviction_stream(synth_sd, monkeypatch, {"SAVSR_CACHE_GB": "0.2", "SAVSR_STREAMS": "3"})
    eng = n.engine()
    st = eng.cache_stats()
    assert st["evictions"] >= 12 and st["budget_limit"] == int(0.2 * (1 << 30)), st
    per_ctx = 80 << 20                                          # (arena chunk + graph I/O of one small context)
    assert st["budget_used"] <= st["budget_limit"] + 3 * per_ctx, st
    total = sum(e.cache_stats()["bytes"] for e in [eng] + eng._siblings)
    assert total == st["budget_used"], (total, st)
    h, w, sc = pairs[5]
    with torch.no_grad():
        ref = O.forward(synth_sd, clip(h, w).unsqueeze(0), sc)
    assert float((first[(h, w, sc)] - ref[0]).abs().max()) < TOL


def test_failed_first_frame_drops_the_half_made_buffer_plan(synth_sd):
    """an exception inside a shape's FIRST frame (its buffer liveness plan is being made) must not leave a partly consumed free
    list behind -- the context is dropped, and the retry equals an undisturbed engine's output bit for bit."""
    import savsr_amd
    from savsr_amd import engine as E
    n = savsr_amd.build_network(dict(type="SAVSR")).eval()
    n.load_state_dict(synth_sd, strict=True)
    n = n.to("cuda:0")
    n.set_scale((2.5, 2.5))
    lq = synth.synth_clip(7, 3, 30, 44, seed=11).to("cuda:0")
    eng = n.engine()
    real = eng.rcab
    calls = {"n": 0}

    def failing(*a, **k):
        calls["n"] += 1
        if calls["n"] == 5:
            raise RuntimeError("injected failure in the middle of the first frame")
        return real(*a, **k)
    eng.rcab = failing
    with pytest.raises(RuntimeError, match="injected"):
        n(lq)
    torch.cuda.synchronize()
    assert eng.cache_stats()["shapes"] == 0 and eng.cache_stats()["budget_used"] == 0
    eng.rcab = real
    out = n(lq).cpu()
    m = savsr_amd.build_network(dict(type="SAVSR")).eval()
    m.load_state_dict(synth_sd, strict=True)
    m = m.to("cuda:0")
    m.set_scale((2.5, 2.5))
    assert torch.equal(out, m(lq).cpu())


def test_config3_all_30_scales_shape_finite_bitwise(net):
    """Every one of the 30 symmetric Vid4 scales (x1.1 ... x4.0, Vid4.yml) at the working LR size: output shape per the
    reference's get_HW, finite, bitwise identical rerun (the oracle comparison runs on the regime-covering subset above)."""
    lq = synth.synth_clip(7, 3, 180, 320, seed=0)
    assert len(workloads.CONFIG3_SCALES) == 30
    for sc in workloads.CONFIG3_SCALES:
        _run(net, lq, sc)


def test_large_frames_1080p_vs_oracle_and_4k(net, synth_sd):
    """Maximum sizes: an LR clip of 540 x 960 (9 x the pixels of the headline shape; 16-row conv tiles 34 x 30 per image, byte offsets of the
    128-channel pair buffers at 265 MB) to 1080 x 1920 at x2 against the CPU oracle, and to 2160 x 3840 (4K: 9 planes of 33 MB between the SATU
    HR stage and the tail) at x4 through the property checks -- shape, finiteness, bitwise rerun -- plus the x2 / x4 agreement of the LR-resolution
    body (the same clip: the x4 output, box-averaged 2 x 2, stays close to the x2 output: a loose sanity bound, not parity)."""
    lq = synth.synth_clip(7, 3, 540, 960, seed=5)
    out2 = _run(net, lq, (2.0, 2.0))
    with torch.no_grad():
        ref = O.forward(synth_sd, lq, (2.0, 2.0))
    err = float((out2 - ref).abs().max())
    print("540x960 x2 max-abs vs oracle", err)
    assert err < TOL
    out4 = _run(net, lq, (4.0, 4.0))
    assert tuple(out4.shape) == (1, 3, 2160, 3840)
    down = torch.nn.functional.avg_pool2d(out4, 2)
    assert float((down - out2).abs().mean()) < 0.05


def test_forget_recycles_a_finished_unit(synth_sd, monkeypatch):
    """cache.forget((h, w), scale): what the YAML job calls when a rank is done with a (dataset, folder) unit -- under memory pressure (here:
    forced) the unit's contexts leave every engine of the budget through the limbo, the account stays exact, other units stay resident, and a
    later visit re-captures to the same bits."""
    import savsr_amd
    monkeypatch.setenv("SAVSR_STREAMS", "3")
    n = savsr_amd.build_network(dict(type="SAVSR")).eval()
    n.load_state_dict(synth_sd, strict=True)
    n = n.to("cuda:0")
    eng = n.engine()
    a = [synth.synth_clip(7, 3, 24, 40, seed=k)[0].to("cuda:0") for k in range(6)]
    b = [synth.synth_clip(7, 3, 30, 36, seed=10 + k)[0].to("cuda:0") for k in range(6)]
    oa = [o.clone() for o in n.forward_many(a, [(2.5, 2.5)] * 6)]
    ob = [o.clone() for o in n.forward_many(b, [(3.0, 2.0)] * 6)]
    torch.cuda.synchronize()
    engs = [eng] + eng._siblings
    assert eng.forget((24, 40), (2.5, 2.5)) == 0, "far below the budget: nothing is forgotten"
    before = sum(e.cache_stats()["scales"] for e in engs)
    dropped = eng.forget((24, 40), (2.5, 2.5), always=True)
    assert dropped >= 1 and sum(e.cache_stats()["scales"] for e in engs) == before - dropped
    assert all((24, 40) != tuple(k[-2:]) for e in engs for k in e._ctx) and any((30, 36) == tuple(k[-2:]) for e in engs for k in e._ctx)
    st = eng.cache_stats()
    assert st["budget_used"] == sum(e.cache_stats()["bytes"] for e in engs) and st["forgotten"] == dropped and st["evictions"] == 0
    oa2 = n.forward_many(a, [(2.5, 2.5)] * 6)            # re-captured
    ob2 = n.forward_many(b, [(3.0, 2.0)] * 6)            # still resident
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(oa, oa2)) and all(torch.equal(x, y) for x, y in zip(ob, ob2))
