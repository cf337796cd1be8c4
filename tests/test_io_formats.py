"""Checkpoint / image formats: savsr_amd/io.py against the reference's conventions
(base_model.py:285-319, data_util.py:29-60, img_util.py:114-153, video_base_model.py:79-92)."""
import os

import numpy as np
import pytest
import torch

import savsr_amd
from savsr_amd import io
from savsr_amd.metrics import tensor2img
from savsr_amd.utils import synth


def test_png_roundtrip_and_read_img_seq(tmp_path):
    frames = [synth.synth_gt(3, 37, 50, seed=i) for i in range(3)]
    for i, f in enumerate(frames):
        io.imwrite(tensor2img(f), str(tmp_path / "clip" / f"{i:08d}.png"))          # BGR uint8, parent dir created
    (tmp_path / "clip" / ".hidden").write_text("x")                                # scandir skips hidden files
    back = io.imread(str(tmp_path / "clip" / "00000001.png"))
    assert back.dtype == np.uint8 and np.array_equal(back, tensor2img(frames[1]))  # lossless, BGR in = BGR out
    seq, names = io.read_img_seq(str(tmp_path / "clip"), return_imgname=True)
    assert names == ["00000000", "00000001", "00000002"] and tuple(seq.shape) == (3, 3, 37, 50) and seq.dtype == torch.float32
    for i, f in enumerate(frames):                                                 # RGB, [0, 1], quantised to 8 bits
        q = (f.clamp(0, 1) * 255.0).round() / 255.0
        assert float((seq[i] - q).abs().max()) < 1e-7
    crop = io.read_img_seq(str(tmp_path / "clip"), require_as_mod_crop=True, scale=(3.5, 2))
    assert tuple(crop.shape[-2:]) == (35, 50) and torch.equal(crop, seq[..., :35, :50])
    with pytest.raises(IOError):
        io.imwrite(np.zeros((4, 4, 3), np.float32), str(tmp_path / "x.png"))


def test_load_network_variants(tmp_path, synth_sd):
    net = savsr_amd.build_network(dict(type="SAVSR")).eval()
    net.load_state_dict(synth_sd, strict=True)
    p = str(tmp_path / "models" / "net_g.pth")
    io.save_network(net, p)
    assert set(torch.load(p).keys()) == {"params"}
    fresh = savsr_amd.build_network(dict(type="SAVSR")).eval()
    io.load_network(fresh, p, strict=True, param_key="params_ema")                 # falls back to 'params'
    for (k, a), (_, b) in zip(fresh.state_dict().items(), net.state_dict().items()):
        assert torch.equal(a, b), k
    # DataParallel-style 'module.' prefixes and a wrapped target
    sd = {"params": {"module." + k: v for k, v in net.state_dict().items()}}
    p2 = str(tmp_path / "wrapped.pth")
    torch.save(sd, p2)

    class Wrap(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.module = m
    fresh2 = savsr_amd.build_network(dict(type="SAVSR")).eval()
    io.load_network(Wrap(fresh2), p2)
    assert torch.equal(fresh2.state_dict()["tail.weight"], net.state_dict()["tail.weight"])
    # non-strict: a tensor of a different size is ignored instead of raising; strict raises
    bad = dict(net.state_dict())
    bad["tail.bias"] = torch.zeros(5)
    p3 = str(tmp_path / "bad.pth")
    torch.save({"params": bad}, p3)
    fresh3 = savsr_amd.build_network(dict(type="SAVSR")).eval()
    before = fresh3.state_dict()["tail.bias"].clone()
    io.load_network(fresh3, p3, strict=False)
    assert torch.equal(fresh3.state_dict()["tail.bias"], before)
    with pytest.raises(RuntimeError):
        io.load_network(fresh3, p3, strict=True)


def test_result_img_path():
    assert io.result_img_path("results/x/visualization", "Vid4", "calendar", "datasets/Vid4/BIx4/calendar/00000003.png", "SAVSR_x4") == \
        os.path.join("results/x/visualization", "Vid4", "calendar", "00000003_SAVSR_x4.png")
    assert io.result_img_path("v", "Vimeo90K", "f", "a/00001/0266/im4.png", "n", suffix="s") == os.path.join("v", "Vimeo90K", "f", "00001_0266_im4_s.png")


def test_frame_store_decodes_each_file_once(tmp_path):
    """io.FrameStore (host side; the device side is covered by tests/test_gpu_run_test.py): files requested ahead are decoded
    on the pool, every later reader -- another dataset over the same dataroot_gt -- gets the cached array."""
    import numpy as np
    from savsr_amd import io as sio
    rng = np.random.RandomState(0)
    paths = []
    for i in range(6):
        p = str(tmp_path / f"{i:03d}.png")
        sio.imwrite(rng.randint(0, 256, (9 + i, 12, 3)).astype(np.uint8), p)
        paths.append(p)
    st = sio.FrameStore(host_bytes=1 << 20, device_bytes=1 << 20, workers=3)
    st.request(paths[:4])
    st.request(paths[:4])                                                     # already in flight: nothing new
    for p in paths:
        a = st.host(p)
        assert a.dtype == np.uint8 and np.array_equal(a[:, :, ::-1], sio.imread(p))
        assert st.host_shape(p) == a.shape[:2]
    assert st.stats["decoded"] == 6
    for p in paths:                                                           # a second dataset over the same files
        st.host(p)
    assert st.stats["decoded"] == 6 and st.stats["host_hits"] >= 6
    small = sio.FrameStore(host_bytes=700, device_bytes=0, workers=1)         # byte cap: old frames are dropped, never the one in hand
    for p in paths:
        small.host(p)
    assert small._host_bytes <= 700 + 14 * 12 * 3


def test_frame_store_host_cap_counts_prefetched_files(tmp_path):
    """SAVSR_DECODE_CACHE_GB is enforced for prefetched files too: a decode started by request() is counted when it completes (not only
    when somebody waits for it), the eviction loop steps over decodes still in flight, a failed decode does not stay cached, and
    request() keeps a bounded number of decodes in flight."""
    import time
    rng = np.random.RandomState(3)
    paths = []
    for i in range(12):
        p = str(tmp_path / f"{i:03d}.png")
        io.imwrite(rng.randint(0, 255, (32, 48, 3), dtype=np.uint8), p)
        paths.append(p)
    one = 32 * 48 * 3
    store = io.FrameStore(host_bytes=3 * one, device_bytes=1 << 30, workers=2)
    store.request(paths)
    from concurrent.futures import Future
    for _ in range(500):                                   # the pool finishes without anybody calling host()
        with store._lock:
            busy = any(isinstance(v, Future) for v in store._host.values())
        if store.stats["decoded"] == len(paths) and not busy:
            break
        time.sleep(0.01)
    assert store.stats["decoded"] == len(paths)
    assert store._host_bytes <= 3 * one and store._host_bytes == sum(v.nbytes for v in store._host.values() if isinstance(v, np.ndarray))
    assert len(store._host) <= 3
    a = store.host(paths[-1])                              # still there (newest) or decoded again: same pixels either way
    assert a.shape == (32, 48, 3)
    bad = str(tmp_path / "broken.png")
    open(bad, "wb").write(b"not a png")
    store.request([bad])
    for _ in range(500):
        if store._key(bad) not in store._host:
            break
        time.sleep(0.01)
    assert store._key(bad) not in store._host              # the failed decode was dropped ...
    with pytest.raises(Exception):
        store.host(bad)                                    # ... and asking for the file raises (again) instead of serving a poisoned entry
    assert store._key(bad) not in store._host
    big = io.FrameStore(host_bytes=1 << 30, device_bytes=1 << 30, workers=1)
    big.request(paths)                                     # at most 8 per worker in flight; the rest waits in the queue ...
    assert big._inflight <= 8 and len(big._host) <= 8
    for _ in range(500):                                   # ... and is decoded BY THE POOL as decodes finish: nobody calls host() here
        if big.stats["decoded"] == len(paths) and big._inflight == 0:
            break
        time.sleep(0.01)
    assert big.stats["decoded"] == len(paths) and not big._pending and big._inflight == 0
    assert all(big.host(p).shape == (32, 48, 3) for p in paths) and big.stats["decoded"] == len(paths)      # all host hits
    big.request(paths)                                     # already decoded: nothing queued
    assert not big._pending and big._inflight == 0


def test_frame_store_clear_during_prefetch_keeps_the_inflight_account(tmp_path):
    """clear() while decodes are running or queued: the queue is dropped, the running decodes are un-counted when they finish (their table
    entries are gone), and the pool keeps accepting work afterwards -- `_inflight` returns to 0 instead of sticking at the cap, which would
    push every later decode onto the consumer thread."""
    import time
    rng = np.random.RandomState(5)
    paths = []
    for i in range(24):
        p = str(tmp_path / f"{i:03d}.png")
        io.imwrite(rng.randint(0, 255, (64, 64, 3), dtype=np.uint8), p)
        paths.append(p)
    store = io.FrameStore(host_bytes=1 << 30, device_bytes=1 << 30, workers=1)
    for rep in range(3):
        store.request(paths)
        assert store._inflight >= 1
        store.clear()
        assert not store._pending and not store._host
    for _ in range(1000):
        if store._inflight == 0:
            break
        time.sleep(0.01)
    assert store._inflight == 0 and not store._counted
    store.request(paths)                                  # the pool still takes work: everything is decoded by it, nobody calls host()
    for _ in range(1000):
        if store._inflight == 0 and not store._pending and len(store._host) == len(paths):
            break
        time.sleep(0.01)
    assert store._inflight == 0 and len(store._host) == len(paths)
    assert all(isinstance(v, np.ndarray) for v in store._host.values())
