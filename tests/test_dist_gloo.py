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
