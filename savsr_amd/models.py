"""Test-time model wrapper under the reference's registry name: `model_type: ASVSRModel` of
options/test/SAVSR/*.yml resolves here through MODEL_REGISTRY.

What is mirrored (inference only -- no optimisers, losses, EMA, loggers or training state):
  SRModel.__init__            lbasicsr/models/sr_model.py:27-41   build_network(opt['network_g']) -> device -> load_network
  BaseModel.feed_data         sr_model.py:91-98                  lq / gt onto the device
  ASVSRModel.test             asvsr_model.py:31-61               set_scale(opt['scale']); eval; no_grad forward
  VideoBaseModel.dist_validation / _log_validation_metric_values
                              video_base_model.py:18-118,125-167 frame round-robin over ranks, per-frame metrics, per-folder
                                                                 means, mean over folders
  SRModel.get_current_visuals sr_model.py:290-294                bicubic-antialias post-resize when output and GT sizes differ

MI355X-first placement: the post-resize (csrc/resize.hip), the uint8 quantisation + BT.601 luma + PSNR-Y / SSIM-Y
(csrc/metrics.hip) run on the GPU on the un-copied output; the metric rows stay in HBM until ONE collective per dataset
(RCCL all_gather_into_tensor of the [n_frames, n_metrics] rows -- the reference issues one dist.reduce per folder).
Images are written by the host (PNG encode is host work) only when `val.save_img` asks for them.
"""
from __future__ import annotations

from collections import OrderedDict
from copy import deepcopy
from typing import Dict, Optional

import torch

from . import io as sio
from .archs import build_network
from .harness import aggregate_rows, block_partition, chunk_block, frame_indices, gather_rows, plan_job
from .metrics import tensor2img
from .registry import METRIC_REGISTRY, MODEL_REGISTRY
from .resize_gpu import resize_bicubic_aa


def _gpu_metric_plan(metrics_opt):
    """The fused GPU kernel produces (PSNR, SSIM) of one crop_border and one test_y_channel setting in one pass.  Returns
    (crop_border, test_y_channel) when every requested metric is calculate_psnr / calculate_ssim with common settings (every
    shipped YAML: test_y_channel true), else raises: there is deliberately no host fallback inside the product path."""
    crop = ych = None
    for name, m in metrics_opt.items():
        if m["type"] not in ("calculate_psnr", "calculate_ssim") or m.get("input_order", "HWC") != "HWC":
            raise NotImplementedError(f"metric '{name}' ({dict(m)}): the GPU metric kernel implements calculate_psnr / calculate_ssim "
                                      "on HWC images (the metrics of options/test/SAVSR/*.yml)")
        if m["type"] not in METRIC_REGISTRY:
            raise KeyError(m["type"])
        c, y = int(m.get("crop_border", 0)), bool(m.get("test_y_channel", False))
        if (crop is not None and c != crop) or (ych is not None and y != ych):
            raise NotImplementedError("metrics with different crop_border / test_y_channel values in one run are not supported by the fused kernel")
        crop, ych = c, y
    return crop, ych


class BaseModel:
    """The slice of lbasicsr/models/base_model.py the test flow touches."""

    def __init__(self, opt):
        self.opt = opt
        if opt.get("num_gpu", 1) == 0 or not torch.cuda.is_available():
            raise RuntimeError("savsr_amd models run on an AMD GPU only (num_gpu: 0 / CPU mode is not available: no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.is_train = opt.get("is_train", False)

    def validation(self, dataloader, current_iter, tb_logger=None, save_img=False):
        """base_model.py:45-57."""
        if self.opt.get("dist"):
            return self.dist_validation(dataloader, current_iter, tb_logger, save_img)
        return self.nondist_validation(dataloader, current_iter, tb_logger, save_img)


@MODEL_REGISTRY.register()
class VideoBaseModel(BaseModel):
    """video_base_model.py + the parts of sr_model.py it inherits, inference only."""

    def __init__(self, opt):
        super().__init__(opt)
        if self.is_train:
            raise NotImplementedError("savsr_amd implements the test flow only (is_train: false)")
        self.net_g = build_network(opt["network_g"]).eval()
        load_path = opt["path"].get("pretrain_network_g")
        if load_path is not None:                                     # sr_model.py:36-39
            sio.load_network(self.net_g, load_path, opt["path"].get("strict_load_g", True), opt["path"].get("param_key_g", "params"))
        self.net_g.to(self.device)
        self.metric_results: Dict[str, torch.Tensor] = {}
        self.last_validation: Optional[dict] = None

    # ---- one frame -------------------------------------------------------------------------------------------------
    def feed_data(self, data):
        self.lq = data["lq"].to(self.device)
        if "gt" in data:
            self.gt = data["gt"].to(self.device)

    def test(self):
        with torch.no_grad():
            self.output = self.net_g(self.lq)

    def get_current_visuals(self):
        """sr_model.py:277-307: arbitrary-scale outputs whose size differs from the ground truth's are resized to it with
        torchvision's BICUBIC + antialias (here: csrc/resize.hip, the same ATen arithmetic)."""
        if hasattr(self, "gt") and self.output.shape[-2:] != self.gt.shape[-2:]:
            self.output = resize_bicubic_aa(self.output, tuple(self.gt.shape[-2:]))
        out = OrderedDict(lq=self.lq, result=self.output)
        if hasattr(self, "gt"):
            out["gt"] = self.gt
        return out

    # ---- one dataset -----------------------------------------------------------------------------------------------
    def _group_scale(self, net):
        """Scale the grouped (forward_many) path runs with: what `test()` would use for the same frame.  VideoBaseModel.test
        does not touch the network's scale (video_base_model.py / sr_model.py:200-212): whatever was set last stays."""
        return tuple(net.scale)

    def _dataset_prologue(self, dataset):
        """What dist_validation does before its frame loop (video_base_model.py:19-37)."""
        if dataset.opt.get("downsampling_scale", 0) != 0:             # :21-22
            self.opt["scale"] = dataset.opt["downsampling_scale"]
        metrics_opt = self.opt["val"].get("metrics")
        with_metrics = metrics_opt is not None
        names = list(metrics_opt.keys()) if with_metrics else []
        crop, ych = _gpu_metric_plan(metrics_opt) if with_metrics else (None, True)
        return metrics_opt, with_metrics, names, crop, ych

    def _run_frames(self, dataset, mine, rows, save_img=False, next_folder=None):
        """The frame loop of video_base_model.py:50-98 over the global frame indices `mine` of `dataset` (this rank's share, folder by
        folder); the PSNR / SSIM rows go to rows[k] for mine[k] and stay in HBM.

        Frames are independent units (the hidden state restarts per window, savsr_arch.py:705-706): instead of one frame at a time
        (video_base_model.py:51-53) they go through the network several at a time, each launch unit on its own HIP stream
        (SAVSR.forward_many) -- the launch-latency-bound parts of one frame run under another's convolutions.  Every frame takes the
        throughput flow of the engine (HipEngine._set_flow), whose result for a frame does not depend on the frames that came with it:
        metric tables and saved images are identical for every world size and partition.
        next_folder: (dataset, folder) the caller runs next -- its files are decoded in the background meanwhile."""
        from .metrics_gpu import psnr_ssim_y
        metrics_opt, with_metrics, names, crop, ych = self._dataset_prologue(dataset)
        dataset_name = dataset.opt["name"]
        net = self.net_g.module if hasattr(self.net_g, "module") else self.net_g
        many = hasattr(net, "forward_many")
        eng = net.engine() if many else None
        streams = max(1, int(getattr(eng, "n_streams", 1))) if many else 1
        folders_all = dataset.data_info["folder"]
        my_folders = list(dict.fromkeys(folders_all[i] for i in mine))
        if hasattr(dataset, "prefetch") and my_folders:
            dataset.prefetch(my_folders[0])
        if many:
            net.eval()                                                # once per call (a recursive walk over ~300 modules: 3 ms)
        timing = bool(self.opt.get("profile_gpu_time"))
        ev = []                                                        # (start, end) HIP events around the device work of a group
        k = 0
        while k < len(mine):
            # a block: this rank's consecutive frames of ONE folder (a folder is one LR shape = one set of captured graphs per stream)
            f0 = folders_all[mine[k]]
            k1 = k
            while k1 < len(mine) and folders_all[mine[k1]] == f0:
                k1 += 1
            nxt = my_folders.index(f0) + 1
            if hasattr(dataset, "prefetch"):                          # the pool decodes the NEXT folder meanwhile
                if nxt < len(my_folders):
                    dataset.prefetch(my_folders[nxt])
                elif next_folder is not None:
                    next_folder[0].prefetch(next_folder[1])
            first = dataset[mine[k]]
            h, w = int(first["lq"].shape[-2]), int(first["lq"].shape[-1])
            self._last_lr_hw = (h, w)
            # clips per launch sequence only where the engine batches them (small frames); large frames keep the one-clip-per-stream rule
            if many and hasattr(eng, "streams_for"):
                streams = max(1, int(eng.streams_for(h * w)))             # (fewer launch units in flight for large frames)
            unit = max(1, int(getattr(eng, "clip_batch", 1))) if (many and h * w <= int(getattr(eng, "clip_batch_max_px", 0))
                                                                  and eng.cfg.get("interval", 0) == 0) else 1
            for a, b in (chunk_block(k1 - k, streams, unit) if many else [(i, i + 1) for i in range(k1 - k)]):
                k0, k_end = k + a, k + b
                vals = [first if idx == mine[k] else dataset[idx] for idx in mine[k0:k_end]]
                if timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                if many:               # (a lone frame too: forward_many's result for a frame does not depend on the frames that came with it)
                    outs = net.forward_many([v["lq"] for v in vals], [self._group_scale(net)] * len(vals))
                else:
                    outs = None
                for j, val in enumerate(vals):
                    val["lq"] = val["lq"].unsqueeze(0)
                    val["gt"] = val["gt"].unsqueeze(0)
                    self.feed_data(val)
                    if outs is None:
                        self.test()
                    else:
                        self.output = outs[j].unsqueeze(0)
                    vis = self.get_current_visuals()
                    if save_img:
                        path = sio.result_img_path(self.opt["path"]["visualization"], dataset_name, val["folder"], val["lq_path"],
                                                   self.opt["name"], self.opt["val"].get("suffix"))
                        sio.imwrite_async(tensor2img(vis["result"]), path)
                    if with_metrics:
                        psnr_ssim_y(vis["result"][0], vis["gt"][0], crop, out=rows[k0 + j], test_y_channel=ych)
                    del self.lq, self.output, self.gt
                if timing:
                    e1.record()
                    ev.append((e0, e1))
            k = k1
        if timing and ev:
            self._gpu_events = getattr(self, "_gpu_events", []) + ev

    def _settle_gpu_time(self):
        ev = getattr(self, "_gpu_events", None)
        if ev:
            torch.cuda.synchronize()
            self.gpu_ms = getattr(self, "gpu_ms", 0.0) + sum(a.elapsed_time(b) for a, b in ev)
            self._gpu_events = []

    def dist_validation(self, dataloader, current_iter, tb_logger=None, save_img=False):
        """ONE dataset (video_base_model.py:18-118): this rank's frames, the gather of the metric rows, the table.  A whole YAML goes
        through `validate_job`, which partitions all its datasets at once."""
        dataset = getattr(dataloader, "dataset", dataloader)
        metrics_opt, with_metrics, names, crop, ych = self._dataset_prologue(dataset)
        rank, world = self.opt.get("rank", 0), self.opt.get("world_size", 1)
        n = len(dataset)
        # Frame partition (:50).  The reference deals frames round-robin, which makes every rank read every file; a dataset
        # that can shard itself hands out contiguous per-folder blocks instead and then loads only its block + window reach.
        if hasattr(dataset, "shard"):
            mine = dataset.shard(rank, world)
            owners = [block_partition(dataset.folder_sizes(), r, world) for r in range(world)]
        else:
            mine = frame_indices(n, rank, world)
            owners = None
        rows = torch.zeros(len(mine), 2, dtype=torch.float64, device=self.device)
        self._run_frames(dataset, mine, rows, save_img)
        if save_img:
            sio.flush_writes()
        self._settle_gpu_time()
        if not with_metrics:
            return None
        if self.opt.get("emulate_world"):
            # bench.py --emulate-world: this process runs rank `rank`'s share of a world-size-N run ALONE (no process group) to time it; the
            # other ranks' rows stay zero and the metric table of such a run means nothing
            allrows = torch.zeros(n, 2, dtype=rows.dtype, device=rows.device)
            if len(mine):
                allrows[torch.as_tensor(list(mine), device=rows.device)] = rows
        else:
            allrows = gather_rows(rows, n, rank, world, owners)       # the one collective of the dataset (:108-113)
        return self._table(dataset, allrows, metrics_opt, names)

    def _table(self, dataset, allrows, metrics_opt, names):
        cols = [(m, 0 if metrics_opt[m]["type"] == "calculate_psnr" else 1) for m in names]
        scale = dataset.opt["downsampling_scale"] if dataset.opt.get("downsampling_scale", 0) != 0 else self.opt.get("scale")
        self.last_validation = aggregate_rows(allrows, cols, dataset.data_info["folder"], dataset.opt["name"], scale)   # :125-167
        self.metric_results = self.last_validation["frames"]
        return self.last_validation

    FORGET_UNITS = 64      # a rank that walks at least this many (dataset, folder) units releases each unit's engine contexts when it is done with it (cache.forget)

    def validate_job(self, datasets, current_iter=None, tb_logger=None, save_img=False):
        """Every dataset of a YAML as ONE job (what lbasicsr/test.py:37-48 loops over, dataset by dataset).  The (dataset, folder) units of
        all datasets are cut over the ranks by harness.plan_job -- folder-major, cost-balanced, whole units wherever possible --, a rank runs
        its segments back to back (one folder decoded once, one graph capture per (folder, scale) it owns, ~40 frames per capture instead
        of ~5), and the metric rows of ALL datasets travel in ONE collective at the end (RCCL all_gather_into_tensor of the padded
        [rows, 2] block; the reference reduces once per folder, :108-113).  Returns the per-dataset tables in dataset order, identical
        on every rank and identical to what world size 1 returns."""
        rank, world = self.opt.get("rank", 0), self.opt.get("world_size", 1)
        units = [u for d, ds in enumerate(datasets) for u in ds.units(d)]
        plan = plan_job(units, world)
        self.last_plan = plan
        sizes = [len(ds) for ds in datasets]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        mine_d = [plan["owners"][d][rank] if d < len(plan["owners"]) else [] for d in range(len(datasets))]
        for ds, mine in zip(datasets, mine_d):
            ds.shard_frames(mine)
        pos = [{g: k for k, g in enumerate(m)} for m in mine_d]
        rows = [torch.zeros(len(m), 2, dtype=torch.float64, device=self.device) for m in mine_d]
        segs = plan["segments"][rank]
        any_metrics = self.opt["val"].get("metrics") is not None
        for si, (d, folder, lo, hi) in enumerate(segs):
            ds = datasets[d]
            nxt = (datasets[segs[si + 1][0]], segs[si + 1][1]) if si + 1 < len(segs) and hasattr(datasets[segs[si + 1][0]], "prefetch") else None
            k0 = pos[d][lo]
            self._run_frames(ds, list(range(lo, hi)), rows[d][k0:k0 + (hi - lo)], save_img, next_folder=nxt)
            if hasattr(ds, "release_resident") and (nxt is None or nxt[0] is not ds):
                ds.release_resident()
            # this rank is done with the unit's (LR size, scale): under memory pressure its engine contexts are recycled now (cache.forget)
            net = self.net_g.module if hasattr(self.net_g, "module") else self.net_g
            hw = getattr(self, "_last_lr_hw", None)
            if hw is not None and hasattr(net, "engine") and hasattr(net.engine(), "forget") and (si + 1 == len(segs) or segs[si + 1][:2] != (d, folder)):
                net.engine().forget(hw, tuple(net.scale), always=len(segs) >= self.FORGET_UNITS)
        if save_img:
            sio.flush_writes()
        self._settle_gpu_time()
        if not any_metrics:
            return [None] * len(datasets)
        # ONE collective for the job: global row index = dataset offset + frame index
        local = torch.cat(rows, 0) if rows else torch.zeros(0, 2, dtype=torch.float64, device=self.device)
        total = offs[-1]
        if self.opt.get("emulate_world"):                             # (bench.py --emulate-world: no process group, the others' rows stay zero)
            allrows = torch.zeros(total, 2, dtype=local.dtype, device=local.device)
            idx = [offs[d] + g for d in range(len(datasets)) for g in mine_d[d]]
            if idx:
                allrows[torch.as_tensor(idx, device=local.device)] = local
        else:
            owners = [[offs[d] + g for d in range(len(datasets)) for g in plan["owners"][d][r]] for r in range(world)]
            allrows = gather_rows(local, total, rank, world, owners)
        results = []
        for d, ds in enumerate(datasets):
            metrics_opt, with_metrics, names, crop, ych = self._dataset_prologue(ds)
            results.append(self._table(ds, allrows[offs[d]:offs[d + 1]], metrics_opt, names))
        return results

    def nondist_validation(self, dataloader, current_iter, tb_logger=None, save_img=False):
        return self.dist_validation(dataloader, current_iter, tb_logger, save_img)       # :120-123


@MODEL_REGISTRY.register()
class ASVSRModel(VideoBaseModel):
    """asvsr_model.py:12-61."""

    def test(self):
        net = self.net_g.module if hasattr(self.net_g, "module") else self.net_g
        net.set_scale(self.opt["scale"])
        if net.training:                         # (asvsr_model.py:58 calls eval() per frame; the recursive walk over ~300 modules costs 3 ms a time)
            net.eval()
        with torch.no_grad():
            self.output = self.net_g(self.lq)

    def _group_scale(self, net):
        net.set_scale(self.opt["scale"])          # asvsr_model.py:54-57: the dataset's scale, set before every forward
        return tuple(net.scale)


def build_model(opt):
    """lbasicsr/models/__init__.py build_model: MODEL_REGISTRY lookup by `model_type`."""
    opt = deepcopy(opt)
    return MODEL_REGISTRY.get(opt["model_type"])(opt)
