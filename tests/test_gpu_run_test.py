"""YAML surface end to end on the GPU: a YAML in the reference's format -> DATASET_REGISTRY /
MODEL_REGISTRY -> PNG folders -> GPU LR synthesis -> SAVSR.forward -> GPU PSNR-Y / SSIM-Y -> metric table + result PNGs,
checked against the host metric path and the CPU oracle; plus the RCCL path on one rank."""
import os
import socket

import numpy as np
import pytest
import torch

from savsr_amd import io as sio
from savsr_amd import metrics as M
from savsr_amd.options import parse_test_options
from savsr_amd.resize_gpu import as_mod_crop_hw
from savsr_amd.utils import synth

pytestmark = pytest.mark.gpu

YAML = """
name: test_SAVSR_synth
model_type: ASVSRModel
num_gpu: 1
manual_seed: 0
datasets:
  test_01:
    name: Vid4_x2
    type: ASVideoTestDataset
    dataroot_gt: {root}/GT
    dataroot_lq: {root}/unused
    io_backend:
      type: disk
    cache_data: false
    num_frame: 7
    padding: reflection
    use_arbitrary_scale_downsampling: true
    downsampling_scale: !!python/tuple [2, 2]
    downsampling_mode: torch
  test_02:
    name: Vid4_x1.5_x2.5
    type: ASVideoTestDataset
    dataroot_gt: {root}/GT
    dataroot_lq: {root}/unused
    io_backend:
      type: disk
    cache_data: false
    num_frame: 7
    padding: reflection
    use_arbitrary_scale_downsampling: true
    downsampling_scale: !!python/tuple [1.5, 2.5]
    downsampling_mode: torch
network_g:
  type: SAVSR
  num_in_ch: 3
  num_feat: 64
  num_frame: 7
  slid_win: 3
  fusion_win: 5
  interval: 0
  w1_num_block: 4
  w2_num_block: 2
  n_resgroups: 4
  n_resblocks: 8
  center_frame_idx: ~
path:
  pretrain_network_g: {root}/savsr_synth.pth
  strict_load_g: true
  resume_state: ~
  results_root: {root}/results
val:
  save_img: true
  suffix: ~
  metrics:
    psnr_y:
      type: calculate_psnr
      crop_border: 0
      test_y_channel: true
    ssim_y:
      type: calculate_ssim
      crop_border: 0
      test_y_channel: true
"""

FOLDERS = {"calendar": (12, 45, 62), "city": (4, 45, 62)}       # name -> (frames, H, W); a 7-frame reflection window needs >= 4 frames


@pytest.fixture(scope="module")
def workdir(tmp_path_factory, synth_sd):
    root = str(tmp_path_factory.mktemp("savsr_run"))
    for name, (n, H, W) in FOLDERS.items():
        for i in range(n):
            img = synth.synth_gt(3, H, W, seed=17 * len(name) + i)                     # [3, H, W] RGB in [0, 1]
            sio.imwrite(M.tensor2img(img), os.path.join(root, "GT", name, f"{i:08d}.png"))
    sd = {"params": {k: v.clone() for k, v in synth_sd.items()}}
    torch.save(sd, os.path.join(root, "savsr_synth.pth"))
    return root


def _opt(root):
    return parse_test_options(YAML.format(root=root), root_path=root)


def test_run_test_yaml_to_metric_table(workdir, synth_sd):
    from oracle import savsr_oracle as O
    from savsr_amd.test import run_test
    opt = _opt(workdir)
    results = run_test(opt)
    assert [r["dataset"] for r in results] == ["Vid4_x2", "Vid4_x1.5_x2.5"]
    for r, ds in zip(results, opt["datasets"].values()):
        sc = ds["downsampling_scale"]
        assert r["scale"] == sc and set(r["metrics"]) == {"psnr_y", "ssim_y"} and set(r["folders"]) == set(FOLDERS)
        per_folder = []
        for name, (n, H, W) in FOLDERS.items():
            Hc, Wc = as_mod_crop_hw(H, W, sc)
            rows = r["frames"][name]
            assert tuple(rows.shape) == (n, 2) and bool(torch.isfinite(rows).all())
            for i in range(n):
                # the reference's layout: results/<name>/visualization/<dataset>/<folder>/<img>_<name>.png (video_base_model.py:79-92)
                p = os.path.join(workdir, "results", opt["name"], "visualization", r["dataset"], name, f"{i:08d}_{opt['name']}.png")
                assert os.path.isfile(p), p
                sr = sio.imread(p)
                gt = sio.imread(os.path.join(workdir, "GT", name, f"{i:08d}.png"))[:Hc, :Wc]
                assert sr.shape == gt.shape
                # host metric path on the SAME quantised images == the GPU kernel's rows (stored as float32 like the reference's)
                assert abs(M.calculate_psnr(sr, gt, 0, test_y_channel=True) - float(rows[i, 0])) < 1e-4
                assert abs(M.calculate_ssim(sr, gt, 0, test_y_channel=True) - float(rows[i, 1])) < 1e-6
            per_folder.append(rows.mean(0))
            assert abs(r["folders"][name]["psnr_y"] - float(rows[:, 0].mean())) < 1e-5
        total = torch.stack(per_folder).mean(0)                                          # mean over folders of per-folder means (:132-146)
        assert abs(r["metrics"]["psnr_y"] - float(total[0])) < 1e-4 and abs(r["metrics"]["ssim_y"] - float(total[1])) < 1e-6
    # one frame against the CPU oracle fed with torch-CPU LR synthesis (the arithmetic the reference's torchvision call runs)
    sc = (1.5, 2.5)
    name, (n, H, W) = "city", FOLDERS["city"]
    Hc, Wc = as_mod_crop_hw(H, W, sc)
    gt = sio.read_img_seq(os.path.join(workdir, "GT", name), require_as_mod_crop=True, scale=sc)
    lq = torch.nn.functional.interpolate(gt, size=(round(Hc / sc[0]), round(Wc / sc[1])), mode="bicubic", align_corners=False, antialias=True)
    from savsr_amd.harness import window_indices
    with torch.no_grad():
        ref = O.forward(synth_sd, lq[window_indices(1, n, 7)].unsqueeze(0), sc)
    want = M.tensor2img(ref[0])
    got = sio.imread(os.path.join(workdir, "results", opt["name"], "visualization", "Vid4_x1.5_x2.5", name, f"{1:08d}_{opt['name']}.png"))
    assert want.shape == got.shape
    d = np.abs(want.astype(np.int32) - got.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 5e-3, (d.max(), (d > 0).mean())      # only round-half ties may flip a level


def test_frames_in_flight_equal_one_at_a_time(workdir, monkeypatch):
    """The validation loop keeps SAVSR_STREAMS launch units in flight on HIP streams; per-frame results are bitwise those of one stream
    carrying the frames one launch unit after the other (the reference's flow, video_base_model.py:51-53, is one frame at a time): every
    frame takes the engine's throughput flow, whose result does not depend on the grouping or the stream count."""
    from savsr_amd.test import run_test
    opt = _opt(workdir)
    opt["val"]["save_img"] = False
    multi = run_test(opt)
    monkeypatch.setenv("SAVSR_STREAMS", "1")
    opt1 = _opt(workdir)
    opt1["val"]["save_img"] = False
    single = run_test(opt1)
    for a, b in zip(multi, single):
        assert a["metrics"] == b["metrics"]
        for f in a["frames"]:
            assert torch.equal(a["frames"][f], b["frames"][f])


def test_job_plan_equals_dataset_by_dataset(workdir, monkeypatch):
    """run_test's default (models.validate_job: all datasets of the YAML as one job, folder-major segments, ONE gather) returns exactly the
    tables of the reference's dataset-by-dataset loop (lbasicsr/test.py:37-48; SAVSR_JOB_PLAN=0), and so does every rank's share of an
    emulated world of three put together (rows a rank does not own stay zero in an emulated run)."""
    from savsr_amd.test import run_test
    opt = _opt(workdir)
    opt["val"]["save_img"] = False
    job = run_test(opt)
    monkeypatch.setenv("SAVSR_JOB_PLAN", "0")
    loop = run_test(_opt(workdir) | {"val": dict(opt["val"])})
    monkeypatch.delenv("SAVSR_JOB_PLAN")
    for a, b in zip(job, loop):
        assert a["dataset"] == b["dataset"] and a["scale"] == b["scale"] and a["metrics"] == b["metrics"] and a["folders"] == b["folders"]
        assert all(torch.equal(a["frames"][f], b["frames"][f]) for f in a["frames"])
    acc = None
    for r in range(3):
        o = _opt(workdir)
        o["val"]["save_img"] = False
        o["rank"], o["world_size"], o["dist"], o["emulate_world"] = r, 3, False, True
        part = run_test(o)
        acc = [{f: t["frames"][f].clone() for f in t["frames"]} for t in part] if acc is None else \
            [{f: x[f] + t["frames"][f] for f in x} for x, t in zip(acc, part)]
    for a, x in zip(job, acc):
        assert all(torch.equal(a["frames"][f], x[f]) for f in x)


def test_post_resize_when_output_and_gt_differ(workdir):
    """sr_model.py:290-294: bicubic + antialias resize of the output to the GT size (GPU kernel vs torch CPU)."""
    from savsr_amd.models import build_model
    opt = _opt(workdir)
    model = build_model(opt)
    g = torch.Generator().manual_seed(5)
    out = torch.rand(1, 3, 41, 57, generator=g)
    model.lq = torch.zeros(1, 7, 3, 8, 8, device="cuda")
    model.gt = torch.zeros(1, 3, 38, 60, device="cuda")
    model.output = out.cuda()
    vis = model.get_current_visuals()
    ref = torch.nn.functional.interpolate(out, size=(38, 60), mode="bicubic", align_corners=False, antialias=True)
    assert tuple(vis["result"].shape) == (1, 3, 38, 60)
    assert float((vis["result"].cpu() - ref).abs().max()) < 2e-6


def test_run_test_with_colour_metrics(workdir):
    """`test_y_channel: false` for both metrics (psnr_ssim.py:12,85): the rows of the GPU kernel against the host restatement on the
    saved, quantised images."""
    from savsr_amd.test import run_test
    opt = _opt(workdir)
    for m in opt["val"]["metrics"].values():
        m["test_y_channel"] = False
    opt["datasets"] = {"test_01": opt["datasets"]["test_01"]}
    r = run_test(opt)[0]
    sc = opt["datasets"]["test_01"]["downsampling_scale"]
    name, (n, H, W) = "city", FOLDERS["city"]
    Hc, Wc = as_mod_crop_hw(H, W, sc)
    rows = r["frames"][name]
    for i in range(n):
        sr = sio.imread(os.path.join(workdir, "results", opt["name"], "visualization", r["dataset"], name, f"{i:08d}_{opt['name']}.png"))
        gt = sio.imread(os.path.join(workdir, "GT", name, f"{i:08d}.png"))[:Hc, :Wc]
        assert abs(M.calculate_psnr(sr, gt, 0, test_y_channel=False) - float(rows[i, 0])) < 1e-4
        assert abs(M.calculate_ssim(sr, gt, 0, test_y_channel=False) - float(rows[i, 1])) < 1e-6
        assert abs(M.calculate_psnr(sr, gt, 0, test_y_channel=True) - float(rows[i, 0])) > 1e-3      # not the luma numbers


def test_unsupported_metric_config_raises(workdir):
    from savsr_amd.models import build_model
    from savsr_amd.datasets import build_dataset
    opt = _opt(workdir)
    opt["val"]["metrics"]["psnr_y"]["test_y_channel"] = False
    model = build_model(opt)
    with pytest.raises(NotImplementedError):
        model.validation(build_dataset(opt["datasets"]["test_01"]), "x", None, False)


def test_rccl_single_rank_gather(workdir):
    """The RCCL branch on the one GPU available here: process group "nccl" with world_size 1, the metric rows go through
    dist.all_gather_into_tensor (harness.gather_rows) and the table equals the non-distributed run's."""
    import torch.distributed as dist
    from savsr_amd.test import run_test
    base = run_test(_opt(workdir))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from savsr_amd import harness
        calls = []
        orig = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            opt = _opt(workdir)
            opt["dist"] = True
            opt["val"]["save_img"] = False
            got = run_test(opt)
        finally:
            dist.all_gather_into_tensor = orig
        assert len(calls) == 1, "ONE collective for the job (both datasets' rows in one all_gather_into_tensor)"
        for a, b in zip(base, got):
            assert a["metrics"] == b["metrics"]
            for f in a["frames"]:
                assert torch.equal(a["frames"][f], b["frames"][f])
    finally:
        dist.destroy_process_group()


def test_frame_store_device_path_matches_read_img_seq(workdir):
    """io.FrameStore.frames_chw_f32 (uint8 upload + 256-entry table on the device) is bit for bit read_img_seq
    (data_util.py:29-60: float32 / 255 on the host), with and without the arbitrary-scale mod crop."""
    store = sio.FrameStore()
    name, (n, H, W) = "calendar", FOLDERS["calendar"]
    paths = sorted(sio.scandir(os.path.join(workdir, "GT", name), full_path=True))
    dev = torch.device("cuda", 0)
    assert torch.equal(store.frames_chw_f32(paths, dev).cpu(), sio.read_img_seq(paths))
    sc = (1.5, 2.5)
    crop = as_mod_crop_hw(H, W, sc)
    assert torch.equal(store.frames_chw_f32(paths, dev, crop).cpu(), sio.read_img_seq(paths, require_as_mod_crop=True, scale=sc))
    assert store.stats["decoded"] == n and store.stats["uploaded"] == n and store.stats["device_hits"] == n


def test_sharded_datasets_read_their_block_only(workdir, monkeypatch):
    """Two ranks of one dataset (contiguous per-folder blocks + window reach): every owned item is bitwise the unsharded
    dataset's, each rank decodes / uploads fewer files than the folder holds, and a second dataset over the same
    dataroot_gt decodes nothing."""
    from savsr_amd import harness
    from savsr_amd.datasets import build_dataset
    root = os.path.join(workdir, "GT7")
    for i in range(12):                                   # one 12-frame folder: blocks of 6 + a 3-frame reach = 9 of 12 files
        sio.imwrite(M.tensor2img(synth.synth_gt(3, 40, 48, seed=90 + i)), os.path.join(root, "walk", f"{i:08d}.png"))
    dopt = dict(_opt(workdir)["datasets"]["test_02"])
    dopt["dataroot_gt"] = root
    monkeypatch.setattr(sio, "_STORE", sio.FrameStore())
    full = build_dataset(dict(dopt))
    ref = [full[i] for i in range(len(full))]
    assert sio.frame_store().stats["decoded"] == 12
    for rank in range(2):
        monkeypatch.setattr(sio, "_STORE", sio.FrameStore())
        ds = build_dataset(dict(dopt))
        mine = ds.shard(rank, 2)
        assert mine == harness.block_partition([12], rank, 2) == list(range(6 * rank, 6 * rank + 6))
        for i in mine:
            it = ds[i]
            assert torch.equal(it["lq"], ref[i]["lq"]) and torch.equal(it["gt"], ref[i]["gt"]) and it["idx"] == ref[i]["idx"]
        st = sio.frame_store().stats
        assert st["decoded"] == st["uploaded"] == 9, st
        ds2 = build_dataset(dict(dopt, downsampling_scale=(2, 2), name="Vid4_x2"))      # another scale, same files
        ds2.shard(rank, 2)
        ds2[mine[0]]
        assert sio.frame_store().stats["decoded"] == 9 and sio.frame_store().stats["uploaded"] == 9


def test_two_ranks_on_one_gpu(workdir, tmp_path):
    """The N > 1 flow end to end on the hardware at hand: TWO ranks (fresh processes from torch.distributed.run, both on cuda:0,
    gloo group because RCCL refuses two ranks on one device) run run_test(opt) over the PNG tree -- the job plan of harness.plan_job
    ((dataset, folder) units cut over the ranks), every rank reading only its segments + window reach, ONE gather of the job's rows,
    aggregation -- and both return exactly the single-process table."""
    import subprocess
    import sys
    from savsr_amd.test import run_test
    opt = _opt(workdir)
    opt["val"]["save_img"] = False
    base = run_test(opt)
    ypath = str(tmp_path / "two_rank.yml")
    open(ypath, "w").write(YAML.format(root=workdir))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), "two_rank_worker.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), helper, ypath, workdir, out], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    tabs = [torch.load(f"{out}.{k}.pt", weights_only=False) for k in range(2)]
    n_files = sum(n for n, _, _ in FOLDERS.values())
    for t in tabs:
        assert [x["dataset"] for x in t["results"]] == [b["dataset"] for b in base]
        for x, b in zip(t["results"], base):
            assert x["metrics"] == b["metrics"] and x["folders"] == b["folders"]
            for f in b["frames"]:
                assert torch.equal(x["frames"][f], b["frames"][f])
        assert 0 < t["store"]["decoded"] <= n_files            # (tiny folders: block + window reach may cover a whole folder)


def test_bench_two_ranks_share_the_gpu():
    """bench.py's N > 1 path on a 1-GPU box: `--gpus 2` starts its own two ranks (SAVSR_BENCH_SHARE_GPU=1: both on GPU 0, gloo
    group), rank 0 prints ONE JSON line with n_gpus 2 and the whole-job value; a test of the code path, not a performance figure."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SAVSR_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--clips-per-step", "3",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["parallelism"] == "clip-sharded dp2"
    assert abs(d["value"] - 2 * 2 * 3 * 0.9216 / (d["ms_per_step"] * 2 / 1e3)) < 0.05 * d["value"]      # whole-job: all ranks' clips / max time


def test_bench_emulate_world_predicts_strong_scaling():
    """`bench.py --config run_test --emulate-world 2`: the whole job and both ranks of a world-size-2 run of the YAML flow, each in a fresh child
    process on the one GPU over one PNG tree; one JSON line with the predicted strong-scaling efficiency and where a rank's cold pass goes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "run_test", "--emulate-world", "2", "--frames-per-folder", "4"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["bench_config"] == "run_test --emulate-world" and d["emulated_world"] == 2 and d["n_gpus"] == 1
    assert d["world1"]["frames"] == 4 * 4 * 6 and sum(x["frames"] for x in d["ranks"]) == 4 * 4 * 6 and all(x["frames"] > 0 for x in d["ranks"])
    assert d["ranks_timed"] == [0, 1] and all(len(x["folders"]) <= 3 for x in d["ranks"])      # (the job plan: folder-major pieces)
    for k in ("predicted_strong_scaling_eff_cold", "predicted_strong_scaling_eff_steady"):
        assert 0.05 < d[k] < 2.5, (k, d[k])      # (a 96-frame job: the whole-job process pays more one-off costs than a rank -- plumbing, not a measurement)
    loss = d["loss_breakdown_cold_slowest_rank"]
    assert loss["captures"] >= 1 and loss["rank_wall_s"] > 0 and loss["png_decoded"] >= 8
