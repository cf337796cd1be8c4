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


# ---- the job plan of a whole YAML (harness.plan_job): (dataset, folder) units, folder-major, cost-balanced -----------------------------
VID4 = [("calendar", 576, 720, 41), ("city", 576, 704, 34), ("foliage", 480, 720, 49), ("walk", 480, 720, 47)]


def _yaml_units(scales, folders=VID4):
    from savsr_amd.resize_gpu import as_mod_crop_hw
    units = []
    for d, sc in enumerate(scales):
        base = 0
        for name, H, W, n in folders:
            hr = as_mod_crop_hw(H, W, tuple(float(v) for v in sc))
            lr = (round(hr[0] / sc[0]), round(hr[1] / sc[1]))
            units.append(dict(dataset=d, folder=name, group=("gt", name), base=base, frames=n, cost=harness.frame_cost(lr, hr)))
            base += n
    return units


def _check_plan(units, world, plan):
    n_ds = 1 + max(u["dataset"] for u in units)
    sizes = [sum(u["frames"] for u in units if u["dataset"] == d) for d in range(n_ds)]
    for d in range(n_ds):                                      # every frame of every dataset exactly once
        assert sorted(g for r in range(world) for g in plan["owners"][d][r]) == list(range(sizes[d]))
    for r in range(world):                                     # segments = the owners lists, contiguous pieces of ONE unit each
        got = {}
        for d, folder, lo, hi in plan["segments"][r]:
            assert lo < hi
            u = next(u for u in units if u["dataset"] == d and u["folder"] == folder)
            assert u["base"] <= lo and hi <= u["base"] + u["frames"]
            got.setdefault(d, []).extend(range(lo, hi))
        for d in range(n_ds):
            assert sorted(got.get(d, [])) == plan["owners"][d][r]


def test_plan_job_on_the_vid4_yaml_dimensions():
    """The shipped Vid4 YAML (42 scales x 4 folders of 41 / 34 / 49 / 47 frames = 168 units, 7 182 frames) over 8 ranks: every frame once,
    a rank owns ~21 units -- at most two of them partial --, sees at most two folders, and the modelled cost is balanced to a few per cent
    (round 5's per-folder block partition: 126+ contexts and all four folders on every rank)."""
    from savsr_amd.utils import workloads
    units = _yaml_units(workloads.YAML_SCALES)
    assert len(units) == 168 and sum(u["frames"] for u in units) == 42 * 171
    for world in (1, 2, 4, 8):
        plan = harness.plan_job(units, world)
        _check_plan(units, world, plan)
        mean = sum(plan["cost"]) / world
        assert max(plan["cost"]) <= 1.04 * mean, (world, plan["cost"])
        for r in range(world):
            segs = plan["segments"][r]
            ctx = {(d, f) for d, f, _, _ in segs}
            whole = sum(1 for d, f, lo, hi in segs if hi - lo == next(u["frames"] for u in units if u["dataset"] == d and u["folder"] == f))
            assert len(segs) - whole <= 2, (world, r)                                  # partial units only at the ends of a rank's piece
            assert len({f for _, f in ctx}) <= (4 if world == 1 else 3 if world <= 4 else 2), (world, r)       # (city is short: a piece of a quarter can span it)
            if world == 8:
                assert len(ctx) <= 44 and len(segs) == len(ctx), (r, len(ctx))        # (heavy low-scale units: few; light x4 units: many)
        if world == 8:
            assert sum(len(s) for s in plan["segments"]) <= 168 + 7                    # at most one extra context per cut
    one = harness.plan_job(units, 1)
    assert [(d, f) for d, f, _, _ in one["segments"][0]] == [(units[i]["dataset"], units[i]["folder"]) for i in one["order"]]
    assert [f for _, f, _, _ in one["segments"][0]] == sorted((f for _, f, _, _ in one["segments"][0]), key=[v[0] for v in VID4].index)   # folder-major


def test_plan_job_few_units_and_tiny_jobs():
    """Fewer units than ranks: the folders are cut like contiguous blocks; a job of a handful of frames still hands every frame out once."""
    units = _yaml_units([(4, 4)], [("a", 576, 720, 32), ("b", 576, 720, 32), ("c", 576, 720, 32), ("d", 576, 720, 32)])
    plan = harness.plan_job(units, 8)
    _check_plan(units, 8, plan)
    assert [sum(hi - lo for _, _, lo, hi in s) for s in plan["segments"]] == [16] * 8
    tiny = _yaml_units([(2, 2)], [("a", 64, 64, 7)])
    plan = harness.plan_job(tiny, 8)
    _check_plan(tiny, 8, plan)
    assert sum(1 for s in plan["segments"] if s) >= 6                                     # (7 frames: nearly every rank gets one)
    plan = harness.plan_job(_yaml_units([(4, 4), (2, 2)], [("a", 128, 128, 9), ("b", 96, 128, 5)]), 3)
    _check_plan(_yaml_units([(4, 4), (2, 2)], [("a", 128, 128, 9), ("b", 96, 128, 5)]), 3, plan)
    assert harness.plan_job([], 4)["segments"] == [[], [], [], []]


def test_chunk_block_rules():
    """How a rank's block of one folder goes to forward_many: streams x clip_batch frames per call for small frames (a short block whole),
    the one-clip-per-stream rule for large ones, never a lone leftover frame."""
    assert harness.chunk_block(41, 3, 3) == [(0, 9), (9, 18), (18, 27), (27, 36), (36, 41)]
    assert harness.chunk_block(17, 3, 3) == [(0, 17)] and harness.chunk_block(5, 3, 3) == [(0, 5)]
    assert harness.chunk_block(19, 3, 3) == [(0, 9), (9, 19)]                            # 9 + 9 + 1 -> the lone frame joins the previous call
    assert harness.chunk_block(5, 3, 1) == [(i, i + 1) for i in range(5)]                # large frames, short block: one at a time
    assert harness.chunk_block(7, 3, 1) == [(0, 2), (2, 4), (4, 7)]
    assert harness.chunk_block(13, 3, 1) == [(0, 3), (3, 6), (6, 9), (9, 13)]
    assert harness.chunk_block(0, 3, 3) == []


JOB = [("Vid4_x4", (4, 4)), ("Vid4_x2", (2, 2)), ("Vid4_x1.5_x4", (1.5, 4))]
JOB_FOLDERS = [("calendar", 96, 120, 7), ("city", 96, 112, 4), ("walk", 80, 120, 6)]


def _job_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    units = _yaml_units([sc for _, sc in JOB], JOB_FOLDERS)
    plan = harness.plan_job(units, world)
    n = sum(k for _, _, _, k in JOB_FOLDERS)
    offs = [d * n for d in range(len(JOB) + 1)]
    truth = _frame_rows(len(JOB) * n)                                   # row of (dataset d, frame g) = truth[d * n + g]
    mine = [offs[d] + g for d in range(len(JOB)) for g in plan["owners"][d][rank]]
    local = truth[mine]                                                 # this rank "computed" only its own frames
    owners = [[offs[d] + g for d in range(len(JOB)) for g in plan["owners"][d][r]] for r in range(world)]
    allrows = harness.gather_rows(local, len(JOB) * n, rank, world, owners)          # ONE collective for the whole job
    folders = [f for f, _, _, k in JOB_FOLDERS for _ in range(k)]
    tables = [harness.aggregate_rows(allrows[offs[d]:offs[d + 1]], [("psnr_y", 0), ("ssim_y", 1)], folders, name, sc) for d, (name, sc) in enumerate(JOB)]
    ret[rank] = (allrows, tables, mine)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_job_plan_one_gather_identical_tables():
    """World size 2 on the job plan: three datasets over three folders, each rank its segments, ONE gather of all rows, per-dataset tables
    identical on both ranks and identical to world size 1 (the rows in frame order, then the float32 table / folder means / mean over folders)."""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_job_worker, args=(2, port, ret), nprocs=2, join=True)
    n = sum(k for _, _, _, k in JOB_FOLDERS)
    truth = _frame_rows(len(JOB) * n)
    (a0, t0, m0), (a1, t1, m1) = ret[0], ret[1]
    assert sorted(m0 + m1) == list(range(len(JOB) * n)) and m0 and m1
    assert torch.equal(a0, truth) and torch.equal(a1, truth)
    folders = [f for f, _, _, k in JOB_FOLDERS for _ in range(k)]
    for d, (name, sc) in enumerate(JOB):
        ref = harness.aggregate_rows(truth[d * n:(d + 1) * n], [("psnr_y", 0), ("ssim_y", 1)], folders, name, sc)
        for t in (t0[d], t1[d]):
            assert t["metrics"] == ref["metrics"] and t["folders"] == ref["folders"] and t["dataset"] == name
            assert all(torch.equal(t["frames"][f], ref["frames"][f]) for f in ref["frames"])
