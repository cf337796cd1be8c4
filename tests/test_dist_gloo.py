"""N > 1 path on CPU: world_size-2 gloo run of the harness's frame sharding + metric gather
(the only cross-rank step of the path, video_base_model.py:50,108-113)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from savsr_amd import harness


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeNet:
    """Stands in for the GPU network on CPU: nearest-neighbour x2 'super-resolution'."""

    def set_scale(self, s):
        self.s = s

    def __call__(self, win):
        c = win[:, win.shape[1] // 2]
        return torch.nn.functional.interpolate(c, scale_factor=2, mode="nearest")


def _worker(rank, world, port, n, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    lq = torch.rand(n, 3, 12, 14, generator=g)
    gt = [torch.rand(3, 24, 28, generator=g) for _ in range(n)]
    rows = harness.validate_folder(_FakeNet(), lq, gt, (2, 2), rank, world)
    ret[rank] = rows
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 6])
def test_two_rank_gather_matches_single(n):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, n, ret), nprocs=2, join=True)
    g = torch.Generator().manual_seed(0)
    lq = torch.rand(n, 3, 12, 14, generator=g)
    gt = [torch.rand(3, 24, 28, generator=g) for _ in range(n)]
    single = harness.validate_folder(_FakeNet(), lq, gt, (2, 2))
    assert torch.equal(ret[0], ret[1])
    assert torch.allclose(ret[0], single, rtol=0, atol=0)


def test_frame_and_window_indices():
    assert harness.frame_indices(7, 1, 3) == [1, 4]
    assert harness.window_indices(0, 41, 7) == [3, 2, 1, 0, 1, 2, 3]
    assert harness.window_indices(40, 41, 7) == [37, 38, 39, 40, 39, 38, 37]
    assert harness.window_indices(0, 10, 5, "replicate") == [0, 0, 0, 1, 2]
    assert harness.window_indices(0, 10, 5, "reflection_circle") == [4, 3, 0, 1, 2]
    assert harness.window_indices(0, 10, 5, "circle") == [3, 4, 0, 1, 2]


# ---- the YAML flow's cross-rank logic (models.VideoBaseModel.dist_validation): per-folder block partition, ONE gather per
# ---- dataset, folder grouping -> per-folder means -> mean over folders (video_base_model.py:50,108-113,125-167) ----------
FOLDERS = [("calendar", 7), ("city", 4), ("walk", 6)]              # uneven: 17 frames over 2 ranks -> 8 + 9, blocks of 3/4, 2/2, 3/3


def _frame_rows(n):
    g = torch.Generator().manual_seed(3)
    return torch.rand(n, 2, generator=g, dtype=torch.float64) * 40.0


def _agg_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = [n for _, n in FOLDERS]
    n = sum(sizes)
    owners = [harness.block_partition(sizes, r, world) for r in range(world)]
    mine = owners[rank]
    rows = _frame_rows(n)[mine]                                       # this rank "computed" only its own frames
    allrows = harness.gather_rows(rows, n, rank, world, owners)
    folders = [f for f, k in FOLDERS for _ in range(k)]
    ret[rank] = (allrows, harness.aggregate_rows(allrows, [("psnr_y", 0), ("ssim_y", 1)], folders, "Vid4_x4", (4, 4)), mine)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_block_partition_gather_and_aggregation():
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_agg_worker, args=(2, port, ret), nprocs=2, join=True)
    n = sum(k for _, k in FOLDERS)
    ref = _frame_rows(n)
    (a0, g0, m0), (a1, g1, m1) = ret[0], ret[1]
    assert len(m0) != len(m1) and sorted(m0 + m1) == list(range(n))          # uneven split, every frame exactly once
    assert torch.equal(a0, ref) and torch.equal(a1, ref)                      # frame order restored on every rank
    # the aggregation against a plain restatement: float32 table, per-folder means, mean over folders
    t = ref.to(torch.float32)
    means, base = {}, 0
    for f, k in FOLDERS:
        means[f] = t[base:base + k].mean(0)
        base += k
    for g in (g0, g1):
        assert list(g["folders"]) == [f for f, _ in FOLDERS]
        for f, _ in FOLDERS:
            assert g["folders"][f]["psnr_y"] == means[f][0].item() and g["folders"][f]["ssim_y"] == means[f][1].item()
            assert g["frames"][f].shape == (dict(FOLDERS)[f], 2)
        assert g["metrics"]["psnr_y"] == sum(means[f][0].item() for f, _ in FOLDERS) / 3
        assert g["metrics"]["psnr_y"] != float(t[:, 0].mean())                # mean over folders, not over frames
    assert g0["metrics"] == g1["metrics"]


def test_block_partition_and_needed_frames():
    """Per-rank work of the sharded datasets: a rank reads about 1/world of a folder + the window reach, never all of it."""
    for sizes in ([41, 34, 49, 47], [32, 32, 32, 32], [7, 5]):
        for world in (1, 2, 3, 8):
            owners = [harness.block_partition(sizes, r, world) for r in range(world)]
            assert sorted(i for o in owners for i in o) == list(range(sum(sizes)))
            base = 0
            for n in sizes:
                total = 0
                for r in range(world):
                    local = [g - base for g in owners[r] if base <= g < base + n]
                    assert local == list(range(local[0], local[0] + len(local))) if local else True      # contiguous
                    assert len(local) in (n // world, n // world + 1)
                    need = harness.needed_frames(local, n, 7, "reflection")
                    assert set(local) <= set(need) and len(need) <= min(n, len(local) + 6)               # block + 3-frame halo each side
                    total += len(need)
                assert total <= n + 6 * world
                base += n
    assert harness.needed_frames([0], 41) == [0, 1, 2, 3] and harness.needed_frames([], 41) == []
