"""Thin test-time harness around the hot path: what `VideoBaseModel.dist_validation` +
`ASVSRModel.test` do for one dataset (lbasicsr/models/video_base_model.py:18-118,
asvsr_model.py:31-61), on in-memory clips.

    for idx in range(rank, n_frames, world):            # frame round-robin   (:50)
        window  = 7 frames around idx, reflection pad    (data_util.py:63-112)
        net.set_scale(scale); out = net(window)          (asvsr_model.py:54-60)
        psnr_y, ssim_y  vs GT                            (:94-98)
    one collective per dataset: gather of the [n_frames, 2] metric rows (:108-113)

Frames are independent units (hidden state restarts per window), so there is no data-path
collective; with RCCL (backend "nccl") the gather rides xGMI, with gloo it runs on CPU tensors.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch

from .metrics import calculate_psnr, calculate_ssim, tensor2img


def frame_indices(n_frames: int, rank: int, world: int) -> List[int]:
    """Frames owned by `rank` (video_base_model.py:50)."""
    return list(range(rank, n_frames, world))


def window_indices(crt_idx: int, max_frame_num: int, num_frames: int = 7, padding: str = "reflection") -> List[int]:
    """Index list of a `num_frames` window centred on `crt_idx` (generate_frame_indices,
    data_util.py:63-112).  e.g. idx 0 of 41 frames -> [3, 2, 1, 0, 1, 2, 3]."""
    if num_frames % 2 != 1:
        raise AssertionError("num_frames should be an odd number.")
    if padding not in ("replicate", "reflection", "reflection_circle", "circle"):
        raise AssertionError(f"Wrong padding mode: {padding}.")
    last = max_frame_num - 1
    half = num_frames // 2
    out = []
    for i in range(crt_idx - half, crt_idx + half + 1):
        if i < 0:
            j = {"replicate": 0, "reflection": -i, "reflection_circle": crt_idx + half - i, "circle": num_frames + i}[padding]
        elif i > last:
            j = {"replicate": last, "reflection": 2 * last - i, "reflection_circle": (crt_idx - half) - (i - last),
                 "circle": i - num_frames}[padding]
        else:
            j = i
        out.append(j)
    return out


def block_partition(folder_sizes: Sequence[int], rank: int, world: int) -> List[int]:
    """Frames owned by `rank` when every folder is cut into `world` CONTIGUOUS blocks (global indices, folders laid out one
    after the other as in `data_info`).  Same units and same independence as the reference's round-robin
    (video_base_model.py:50; frames are independent: hidden state restarts per window), but a rank then needs only its block
    plus the window reach at both ends (`needed_frames`) -- 1/world of a folder's decode / upload / LR-synthesis work instead
    of all of it.  Block r of an n-frame folder is [r*n // world, (r+1)*n // world): sizes differ by at most one."""
    out, base = [], 0
    for n in folder_sizes:
        out.extend(range(base + (rank * n) // world, base + ((rank + 1) * n) // world))
        base += n
    return out


# ---- the job plan of a whole YAML: (dataset, folder) units cut over the ranks ---------------------------------------------------------
# A shipped YAML is MANY datasets over one set of folders (test_SAVSR_Vid4_asBI.yml: 42 datasets = 42 scales x the 4 Vid4 folders = 168
# (dataset, folder) units of 34-49 frames).  Cutting every folder of every dataset into `world` blocks (block_partition, round 5) makes every
# rank visit every (folder, scale) context: 126 graph-capture contexts for ~860 frames at world 8, 4-6 frames each, every folder decoded on
# every rank.  The plan below lays the units out FOLDER-MAJOR on one line (all datasets of folder 0, then of folder 1, ...), weighs every
# frame with a cost model and cuts the line into `world` contiguous pieces of equal cost: a rank owns ~units / world WHOLE units (at most
# two partial ones, at the ends of its piece), sees one or two folders, captures a context for ~40 frames instead of ~5 and decodes only the
# folders its piece touches.  Frames stay the independent unit (video_base_model.py:50: hidden state restarts per window), results are
# identical under any partition, and the only collective is the gather of the metric rows at the end of the job.

LR_PX_FLOOR = 12000.0        # below this many LR pixels a frame is launch-latency-bound: its cost stops falling with its size
HR_PX_WEIGHT = 0.012         # cost of one HR pixel (SATU HR stage + tail + metrics) in LR-pixel units: 84 us / 0.92 MP against 7.2 ms / 57.6 kpx
UNIT_FIXED_LR_PX = 150000.0  # what entering a (folder, scale) context costs (buffer plan, table upload, graph capture: ~20 ms) in LR-pixel units
MIN_SPLIT_FRAMES = 4         # a cut closer than this to a unit's edge moves to the edge (a 2-frame splinter is not worth its capture)


def frame_cost(lr_hw: Tuple[int, int], hr_hw: Tuple[int, int]) -> float:
    """Relative GPU cost of one frame of a unit.  The network body is LR-sized (146 ns per LR pixel at 180x320 and at 480x318 alike,
    profiles/r05_bench_config3.json / config4.json), SATU's HR stage + the tail + the metrics are HR-sized and small."""
    return max(float(lr_hw[0] * lr_hw[1]), LR_PX_FLOOR) + HR_PX_WEIGHT * float(hr_hw[0] * hr_hw[1])


def plan_job(units: Sequence[dict], world: int, min_split: int = MIN_SPLIT_FRAMES) -> dict:
    """units: one dict per (dataset, folder) in dataset-major `data_info` order --
         {"dataset": d, "folder": name, "group": hashable identity of the folder's files (equal across the datasets that read them),
          "base": global index of the folder's first frame inside dataset d, "frames": n, "cost": cost of one frame (frame_cost)}.
    Returns {"segments": [per rank: ordered [(d, folder, lo, hi)] with GLOBAL frame indices [lo, hi) of dataset d],
             "owners":   [per dataset: [per rank: sorted global frame indices]],
             "cost":     [per rank: planned cost], "order": the folder-major unit order (indices into units)}.
    Deterministic and rank-independent: every rank computes the same plan from the same YAML."""
    if world < 1:
        raise ValueError("world >= 1")
    seen: dict = {}
    for u in units:
        seen.setdefault(u["group"], len(seen))
    order = sorted(range(len(units)), key=lambda i: (seen[units[i]["group"]], units[i]["dataset"], units[i]["base"]))
    n_ds = 1 + max((u["dataset"] for u in units), default=-1)
    segments: List[List[tuple]] = [[] for _ in range(world)]
    owners: List[List[List[int]]] = [[[] for _ in range(world)] for _ in range(n_ds)]
    cost = [0.0] * world
    total = sum(u["frames"] * u["cost"] + UNIT_FIXED_LR_PX for u in units if u["frames"] > 0)
    if total <= 0:
        return {"segments": segments, "owners": owners, "cost": cost, "order": order}
    share = total / world
    cum = 0.0
    for i in order:
        u = units[i]
        n = int(u["frames"])
        if n <= 0:
            continue
        # rank of every frame by the midpoint of its cost interval on the line; the unit's fixed cost is spread over its frames (every piece
        # of a split unit pays its own capture: charging the first piece alone would hand it fewer frames than the second)
        fc = u["cost"] + UNIT_FIXED_LR_PX / n
        ranks = [min(world - 1, int((cum + (k + 0.5) * fc) / share)) for k in range(n)]
        cum += n * fc
        # cuts inside the unit: positions where the rank changes; a cut within min_split frames of an edge (or of the previous cut) snaps away
        cuts = [k for k in range(1, n) if ranks[k] != ranks[k - 1]]
        ms = max(1, min(min_split, int(share / fc / 2)))       # (a rank's whole share is a few frames: small pieces are all there is)
        pieces, lo = [], 0
        for c in cuts:
            if c - lo < ms or n - c < ms:
                continue
            pieces.append((lo, c, ranks[(lo + c - 1) // 2]))
            lo = c
        pieces.append((lo, n, ranks[(lo + n - 1) // 2]))
        for a, b, r in pieces:
            segments[r].append((u["dataset"], u["folder"], u["base"] + a, u["base"] + b))
            owners[u["dataset"]][r].extend(range(u["base"] + a, u["base"] + b))
            cost[r] += (b - a) * u["cost"] + UNIT_FIXED_LR_PX * (b - a) / n
    for d in range(n_ds):
        for r in range(world):
            owners[d][r].sort()
    return {"segments": segments, "owners": owners, "cost": cost, "order": order}


def chunk_block(n_f: int, streams: int, clip_batch: int) -> List[Tuple[int, int]]:
    """How a block of n_f consecutive frames of ONE folder goes to forward_many: [(start, end)] offsets into the block.
    clip_batch > 1 (frames small enough to share launch sequences): every call hands over streams x clip_batch frames, a block shorter than
    two such calls goes whole (forward_many cuts balanced units: 4 -> 2 + 2).  clip_batch == 1 (large frames: one clip per launch sequence):
    `streams` frames per call, fewer for short blocks -- every stream's engine captures its own graphs, which only pays back over enough
    frames (n_f < 6: one at a time).  A lone leftover frame joins the previous call."""
    if n_f <= 0:
        return []
    if clip_batch > 1:
        g = streams * clip_batch
        g = g if n_f >= 2 * g else n_f
    else:
        g = streams if n_f >= 4 * streams else (min(streams, 2) if n_f >= 6 else 1)
    out = [(a, min(a + g, n_f)) for a in range(0, n_f, g)]
    if g > 1 and len(out) >= 2 and out[-1][1] - out[-1][0] == 1:
        out[-2:] = [(out[-2][0], out[-1][1])]
    return out


def needed_frames(owned_local: Sequence[int], n: int, num_frames: int = 7, padding: str = "reflection") -> List[int]:
    """Frames of an n-frame folder that the windows of `owned_local` read (sorted): the block + a halo of num_frames // 2
    on each side, folded back inside the folder by the padding rule."""
    need = set()
    for i in owned_local:
        need.update(window_indices(i, n, num_frames, padding))
    return sorted(need)


def gather_rows(local: torch.Tensor, n_total: int, rank: int, world: int, owners: Optional[Sequence[Sequence[int]]] = None) -> torch.Tensor:
    """All ranks' per-frame rows in frame order.  local: [frames owned by `rank`, k], in the order of its index list.
    owners[r] = global frame indices of rank r (every rank passes the same lists; default: the reference's round-robin
    `frame_indices`).  One padded all_gather (RCCL all_gather_into_tensor on GPU tensors; list all_gather on gloo)."""
    import torch.distributed as dist
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local                     # no process group: nothing to gather (a single-rank GROUP still runs the collective)
    if owners is None:
        owners = [frame_indices(n_total, r, world) for r in range(world)]
    assert len(owners) == world and sum(len(o) for o in owners) == n_total and local.shape[0] == len(owners[rank])
    per = max(1, max(len(o) for o in owners))
    k = local.shape[1]
    padded = torch.zeros(per, k, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    if local.is_cuda and dist.get_backend() == "gloo":
        # a gloo group over GPU ranks (several ranks sharing ONE GPU in a test -- RCCL refuses duplicate devices): the rows take
        # the host path of the collective and come back to the device
        lst = [torch.empty(per, k, dtype=local.dtype) for _ in range(world)]
        dist.all_gather(lst, padded.cpu())
        parts = torch.stack(lst, 0).to(local.device)
    elif local.is_cuda:
        buf = torch.empty(world * per, k, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, padded)
        parts = buf.view(world, per, k)
    else:
        lst = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(lst, padded)
        parts = torch.stack(lst, 0)
    out = torch.empty(n_total, k, dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = list(owners[r])
        if idx:
            out[torch.as_tensor(idx, device=local.device)] = parts[r, : len(idx)]
    return out


def aggregate_rows(allrows: torch.Tensor, metric_cols: Sequence[Tuple[str, int]], folders: Sequence[str], dataset_name: str = "",
                   scale=None) -> dict:
    """What the reference does with the reduced metric tensor (video_base_model.py:108-113, 125-167), device-free:
    allrows [n_frames, k] in frame order -> float32 table of the requested metrics (the reference accumulates in float32
    tensors, :36-37) -> per-folder frame tables -> per-folder means -> the mean over FOLDERS (not over frames).
    metric_cols: (metric name, column of allrows); folders[i] = folder of frame i."""
    names = [m for m, _ in metric_cols]
    table = allrows[:, [c for _, c in metric_cols]].to(torch.float32).cpu()
    frames = {}
    for f in dict.fromkeys(folders):
        frames[f] = table[[i for i, g in enumerate(folders) if g == f]]
    avg = {f: torch.mean(t, dim=0) for f, t in frames.items()}
    total = {m: 0.0 for m in names}
    for t in avg.values():
        for i, m in enumerate(names):
            total[m] += t[i].item()
    for m in names:
        total[m] /= max(1, len(avg))
    return {"dataset": dataset_name, "scale": scale, "metrics": total,
            "folders": {f: {m: t[i].item() for i, m in enumerate(names)} for f, t in avg.items()},
            "frames": {f: frames[f].clone() for f in avg}}


def validate_folder(net: Callable, lq_frames: torch.Tensor, gt_frames: Sequence[torch.Tensor], scale: Tuple[float, float],
                    rank: int = 0, world: int = 1, num_frame: int = 7, padding: str = "reflection",
                    device: Optional[torch.device] = None) -> torch.Tensor:
    """PSNR-Y / SSIM-Y of every frame of one folder.  lq_frames: [N, 3, h, w]; gt_frames[i]: [3, H, W].
    Returns the gathered [N, 2] rows (identical on every rank)."""
    n = lq_frames.shape[0]
    mine = frame_indices(n, rank, world)
    rows = torch.zeros(len(mine), 2, dtype=torch.float64)
    net.set_scale(scale)
    on_gpu = device is not None and device.type == "cuda"
    if on_gpu:                                        # metrics stay on the device until the per-dataset gather
        from .metrics_gpu import psnr_ssim_y
        from .resize_gpu import resize_bicubic_aa
        rows = rows.to(device)
    for k, idx in enumerate(mine):
        win = lq_frames[window_indices(idx, n, num_frame, padding)].unsqueeze(0)
        if device is not None:
            win = win.to(device)
        out = net(win)
        if on_gpu:
            gt = gt_frames[idx].to(device)
            if out.shape[-2:] != gt.shape[-2:]:       # arbitrary-scale post-resize to the GT size (sr_model.py:290-294)
                out = resize_bicubic_aa(out, tuple(gt.shape[-2:]))
            psnr_ssim_y(out[0], gt, 0, out=rows[k])
        else:
            # Host metrics (the reference's own numpy formulation, savsr_amd/metrics.py).  Only reached with a CPU `net`,
            # i.e. by the world-size-2 gloo test of the sharding / gather logic (tests/test_dist_gloo.py) -- the real
            # network refuses CPU tensors, so the product flow never computes here.
            sr, gt = tensor2img(out[0]), tensor2img(gt_frames[idx])
            rows[k, 0] = calculate_psnr(sr, gt, 0, test_y_channel=True)
            rows[k, 1] = calculate_ssim(sr, gt, 0, test_y_channel=True)
    if on_gpu:
        return gather_rows(rows, n, rank, world).cpu()
    return gather_rows(rows, n, rank, world)


def validate_folder_from_gt(net: Callable, gt_frames: torch.Tensor, scale: Tuple[float, float], rank: int = 0, world: int = 1,
                            num_frame: int = 7, padding: str = "reflection", device: Optional[torch.device] = None) -> torch.Tensor:
    """The reference's test flow with `use_arbitrary_scale_downsampling` (asvideo_test_dataset: GT -> as_mod_crop ->
    arbitrary_scale_downsample -> windows -> net -> metrics), with every step on the GPU when `device` is one:
    gt_frames [N, 3, H, W] RGB in [0, 1] (any size: the arbitrary-scale mod crop is applied here, transforms.py:48-69),
    LR synthesis by savsr_amd.resize_gpu (data_util.py:371-420), PSNR-Y / SSIM-Y by savsr_amd.metrics_gpu.
    Returns the gathered [N, 2] rows."""
    from .resize_gpu import arbitrary_scale_downsample, as_mod_crop_hw
    H, W = as_mod_crop_hw(gt_frames.shape[-2], gt_frames.shape[-1], tuple(scale))
    gt = gt_frames[..., :H, :W].contiguous()
    if device is None or device.type != "cuda":
        raise RuntimeError("validate_folder_from_gt synthesises the LR frames on the GPU (csrc/resize.hip); there is no CPU fallback")
    gt = gt.to(device)
    lq = arbitrary_scale_downsample(gt, tuple(scale))
    return validate_folder(net, lq, list(gt), scale, rank, world, num_frame, padding, device)
