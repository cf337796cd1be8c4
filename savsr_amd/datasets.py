"""Test datasets under the reference's registry names: `type: ASVideoTestDataset` /
`type: VideoTestDataset` of options/test/SAVSR/*.yml resolve here through DATASET_REGISTRY.

Contract mirrored from lbasicsr/data/video_test_dataset.py:12-141 (VideoTestDataset) and :273-328
(ASVideoTestDataset): the folder scan, `data_info` (lq_path / gt_path / folder / idx 'i/n' / border), the
7-frame window with reflection padding (data_util.py:63-112 = harness.window_indices), the arbitrary-scale mod
crop of the ground truth (transforms.py:48-69) and the 'BI' LR synthesis (data_util.py:371-420).

MI355X-first differences (results identical, work placed differently):
  * PNG decode is host work (PIL); everything after it is on the GPU.  Files go through the process-wide `io.FrameStore`:
    a file is decoded ONCE per process (on a small thread pool that runs a folder ahead of the GPU) and uploaded ONCE as
    uint8, however many of the YAML's datasets (42 scales over one `dataroot_gt`) read it; the mod crop is a view and the
    uint8 -> fp32 conversion a table lookup on the device, bit for bit `astype(float32) / 255`.  Each frame's LR version is
    synthesised ONCE per dataset by csrc/resize.hip -- the reference re-reads and re-resizes the 7 frames of every window on
    the CPU (7x the decode + resize work per output frame).
  * Sharding (`shard(rank, world)`): every folder is cut into `world` contiguous blocks (harness.block_partition), so a
    rank decodes / uploads / synthesises only its block plus the window reach (num_frame // 2 frames at each end) instead
    of the whole folder (the reference's round-robin, video_base_model.py:50, makes every rank read every frame).
  * `__getitem__` returns DEVICE tensors ('lq' [t, c, h, w], 'gt' [c, h, w]); the DataLoader is not used (the
    reference's validation loop indexes the dataset directly too, video_base_model.py:51).
There is no CPU path: constructing a dataset without a GPU raises.
"""
from __future__ import annotations

import glob
from collections import OrderedDict
from os import path as osp
from typing import Dict, List

import torch

from . import io as sio
from ._xfer import h2d
from .harness import block_partition, needed_frames, window_indices
from .registry import DATASET_REGISTRY
from .resize_gpu import arbitrary_scale_downsample, as_mod_crop_hw

_SUPPORTED = ("vid4", "reds4", "redsofficial", "udm10")
_MAX_CACHED_FOLDERS = 2      # folders whose GT / LR frames stay resident in HBM (the validation loop walks folder by folder)


def _device(opt) -> torch.device:
    dev = opt.get("device")
    if dev is None:
        if not torch.cuda.is_available():
            raise RuntimeError("savsr_amd datasets synthesise LR frames on an AMD GPU; there is no CPU fallback")
        dev = torch.device("cuda", torch.cuda.current_device())
    return torch.device(dev)


@DATASET_REGISTRY.register()
class VideoTestDataset:
    """video_test_dataset.py:12-141.  LR frames are read from `dataroot_lq` (no synthesis)."""

    def __init__(self, opt):
        self.opt = opt
        self.cache_data = opt["cache_data"]
        self.gt_root, self.lq_root = opt["dataroot_gt"], opt.get("dataroot_lq")
        self.data_info = {"lq_path": [], "gt_path": [], "folder": [], "idx": [], "border": []}
        if opt["io_backend"]["type"] == "lmdb":
            raise AssertionError("No need to use lmdb during validation/test.")
        self.as_down = "use_arbitrary_scale_downsampling" in opt
        if self.as_down:
            self.scale = opt["downsampling_scale"]
        self.device = _device(opt)
        self.imgs_lq: Dict[str, List[str]] = {}
        self.imgs_gt: Dict[str, List[str]] = {}
        self._resident: "OrderedDict[str, dict]" = OrderedDict()
        if "meta_info_file" in opt:
            with open(opt["meta_info_file"], "r") as fin:
                subfolders = [line.split(" ")[0].strip() for line in fin]
            sub_gt = [osp.join(self.gt_root, k) for k in subfolders]
        else:
            sub_gt = sorted(glob.glob(osp.join(self.gt_root, "*")))
        # the LR folders are only consulted when LR frames are read from disk (the reference lists them in any case and
        # then never opens them in the arbitrary-scale flow: 'dataroot_lq ... not needed', Vid4.yml:15)
        lq_from_disk = not self._synthesise()
        sub_lq = [osp.join(self.lq_root, osp.basename(p)) for p in sub_gt] if lq_from_disk else sub_gt
        if opt["name"].lower().split("_")[0] not in _SUPPORTED:
            raise ValueError(f'Non-supported video test dataset: {type(opt["name"])}')
        for f_lq, f_gt in zip(sub_lq, sub_gt):
            name = osp.basename(f_gt)
            p_gt = sorted(sio.scandir(f_gt, full_path=True))
            p_lq = sorted(sio.scandir(f_lq, full_path=True)) if lq_from_disk else p_gt
            n = len(p_lq)
            assert n == len(p_gt), f"Different number of images in lq ({n}) and gt folders ({len(p_gt)})"
            self.data_info["lq_path"].extend(p_lq)
            self.data_info["gt_path"].extend(p_gt)
            self.data_info["folder"].extend([name] * n)
            self.data_info["idx"].extend(f"{i}/{n}" for i in range(n))
            border = [0] * n
            for i in range(opt["num_frame"] // 2):
                border[i] = 1
                border[n - i - 1] = 1
            self.data_info["border"].extend(border)
            self.imgs_lq[name], self.imgs_gt[name] = p_lq, p_gt

    def _synthesise(self) -> bool:
        return False

    # ---- sharding ---------------------------------------------------------------------------------------------------
    def folder_sizes(self) -> List[int]:
        return [len(v) for v in self.imgs_gt.values()]               # insertion order = data_info order

    def shard(self, rank: int = 0, world: int = 1) -> List[int]:
        """Global frame indices owned by `rank` (contiguous per-folder blocks) -- and from here on this dataset object
        only ever decodes / uploads / synthesises the frames those windows read."""
        owned = block_partition(self.folder_sizes(), rank, world)
        if world <= 1:                                   # the whole dataset again (undoes an earlier shard): every frame loadable
            self._need = None
            self._resident.clear()
            return owned
        self._need = {}
        base = 0
        for name, paths in self.imgs_gt.items():
            n = len(paths)
            local = [g - base for g in owned if base <= g < base + n]
            self._need[name] = needed_frames(local, n, self.opt["num_frame"], self.opt["padding"])
            base += n
        self._resident.clear()
        return owned

    def shard_frames(self, owned: List[int]) -> List[int]:
        """Like shard(), for ANY set of owned global frame indices (the job plan of a whole YAML, harness.plan_job, hands a rank whole
        folders of some datasets and nothing of others): from here on this object only decodes / uploads / synthesises the frames the
        windows of `owned` read."""
        owned = sorted(int(g) for g in owned)
        self._need = {}
        base = 0
        for name, paths in self.imgs_gt.items():
            n = len(paths)
            local = [g - base for g in owned if base <= g < base + n]
            self._need[name] = needed_frames(local, n, self.opt["num_frame"], self.opt["padding"])
            base += n
        self._resident.clear()
        return owned

    def release_resident(self) -> None:
        """Drop the folders this object keeps in HBM (the job plan walks (dataset, folder) segments: a dataset is not revisited soon)."""
        self._resident.clear()

    def _unit_sizes(self, folder: str, H: int, W: int):
        """(LR size, HR size) of a frame of `folder` whose GT file is H x W (for the job plan's cost model only)."""
        if self.imgs_lq[folder] is not self.imgs_gt[folder] and self.imgs_lq[folder]:
            return sio.image_size(self.imgs_lq[folder][0]), (H, W)
        return (H, W), (H, W)

    def units(self, d: int = 0) -> List[dict]:
        """The (dataset, folder) units of this dataset for harness.plan_job: folder, first global frame index, frame count, cost of a frame
        (from the image headers; nothing is decoded) and `group` = the identity of the folder's files, equal for every dataset of a YAML
        that reads the same `dataroot_gt` folder."""
        from .harness import frame_cost
        out, base = [], 0
        for name, paths in self.imgs_gt.items():
            n = len(paths)
            if n:
                H, W = sio.image_size(paths[0])
                lr, hr = self._unit_sizes(name, H, W)
                out.append(dict(dataset=d, folder=name, group=(osp.abspath(self.gt_root), name), base=base, frames=n, cost=frame_cost(lr, hr)))
            base += n
        return out

    def needed(self, folder: str) -> List[int]:
        """Frames of `folder` this object loads (all of them until shard() narrows it)."""
        need = getattr(self, "_need", None)
        return list(range(len(self.imgs_gt[folder]))) if need is None else need[folder]

    def _read_lists(self, folder: str) -> List[List[str]]:
        """The path lists _load_folder really reads for `folder` (GT always; the LR files only when the LR frames come from disk)."""
        lists = [self.imgs_gt[folder]]
        if not self._synthesise() and self.imgs_lq[folder] is not self.imgs_gt[folder]:
            lists.append(self.imgs_lq[folder])
        return lists

    def prefetch(self, folder: str) -> None:
        """Start decoding the needed files of `folder` in the background (the validation loop calls this one folder ahead)."""
        store = sio.frame_store()
        for paths in self._read_lists(folder):
            store.request([paths[i] for i in self.needed(folder)])

    # ---- per-folder residency -----------------------------------------------------------------------------------
    def _folder(self, folder: str) -> dict:
        """GT frames (mod-cropped when the flow asks for it) and LR frames of one folder, resident on the GPU:
        {'gt': [k, 3, H, W], 'lq': [k, 3, h, w], 'pos': frame index -> row} for the k needed frames."""
        ent = self._resident.get(folder)
        if ent is None:
            ent = self._load_folder(folder)
            pos = {f: i for i, f in enumerate(self.needed(folder))}
            ent["pos"] = pos
            # window rows of every frame whose window is resident, as ONE device index table per folder: __getitem__ then gathers
            # a window with index_select on a row view -- indexing with a Python list builds a host tensor and copies it to the
            # device through the stream, i.e. waits for every frame still in flight (5 ms per item under cProfile)
            n = len(self.imgs_gt[folder])
            win = torch.zeros(n, self.opt["num_frame"], dtype=torch.int64)
            ok = [False] * n
            for i in range(n):
                sel = window_indices(i, n, self.opt["num_frame"], padding=self.opt["padding"])
                if all(j in pos for j in sel):
                    win[i] = torch.tensor([pos[j] for j in sel], dtype=torch.int64)
                    ok[i] = True
            ent["win"], ent["win_ok"] = h2d(win, self.device), ok
            self._resident[folder] = ent
            while len(self._resident) > _MAX_CACHED_FOLDERS:
                self._resident.popitem(last=False)
        else:
            self._resident.move_to_end(folder)
        return ent

    def _read(self, paths: List[str], folder: str, crop_scale) -> torch.Tensor:
        """read_img_seq over the needed frames of a folder, through the frame store."""
        store = sio.frame_store()
        sel = [paths[i] for i in self.needed(folder)]
        store.request(sel)
        out = []
        for p in sel:                                   # (frames of one folder may differ in size only in theory; crop per file)
            crop = None
            if crop_scale is not None:
                h, w = store.host_shape(p)
                crop = as_mod_crop_hw(h, w, crop_scale)
            out.append(store.frames_chw_f32([p], self.device, crop)[0])
        return torch.stack(out, 0)

    def _load_folder(self, folder: str) -> dict:
        # video_test_dataset.py:101-119: only the cache_data branch mod-crops (LR and GT alike); the per-item branch (:133-137)
        # reads both as they are
        crop = self.scale if (self.cache_data and self.as_down) else None
        return {"gt": self._read(self.imgs_gt[folder], folder, crop), "lq": self._read(self.imgs_lq[folder], folder, crop)}

    def __getitem__(self, index):
        folder = self.data_info["folder"][index]
        idx, max_idx = (int(v) for v in self.data_info["idx"][index].split("/"))
        ent = self._folder(folder)
        sel = window_indices(idx, max_idx, self.opt["num_frame"], padding=self.opt["padding"])
        if max(sel) >= max_idx or min(sel) < 0:      # a folder shorter than the padding reach (the reference fails the same way, on the host)
            raise IndexError(f"folder '{folder}' has {max_idx} frames: too few for a {self.opt['num_frame']}-frame '{self.opt['padding']}' window")
        if not ent["win_ok"][idx]:
            raise IndexError(f"frame {idx} of '{folder}' is outside this rank's shard (shard() narrowed the resident frames)")
        out = {"lq": ent["lq"].index_select(0, ent["win"][idx]), "gt": ent["gt"][ent["pos"][idx]], "folder": folder, "idx": self.data_info["idx"][index],
               "border": self.data_info["border"][index], "lq_path": self.data_info["lq_path"][index]}
        if "scale" in self.opt:
            out["scale"] = self.opt["scale"]
        return out

    def __len__(self):
        return len(self.data_info["gt_path"])


@DATASET_REGISTRY.register()
class ASVideoTestDataset(VideoTestDataset):
    """video_test_dataset.py:273-328: LR frames = arbitrary_scale_downsample(as_mod_crop(GT), downsampling_scale)."""

    def __init__(self, opt):
        super().__init__(opt)
        if "downsampling_scale" in self.opt.keys():
            self.opt["scale"] = self.opt["downsampling_scale"]
        if self._synthesise() and self.opt.get("downsampling_mode", "torch") not in ("torch", "core"):
            raise ValueError(f"downsampling_mode '{self.opt['downsampling_mode']}' (data_util.py:408-412 knows 'torch' and 'core')")

    def _synthesise(self) -> bool:
        # cache_data=True synthesises unconditionally (video_test_dataset.py:303-306); otherwise the YAML switch decides
        # (:311-312: `self.opt['use_arbitrary_scale_downsampling']` -- a KeyError when the key is absent, mirrored here)
        if self.opt.get("cache_data"):
            return True
        return bool(self.opt["use_arbitrary_scale_downsampling"])

    def _unit_sizes(self, folder: str, H: int, W: int):
        scale = self.opt["scale"]
        scale = tuple(scale) if isinstance(scale, (tuple, list)) else (scale, scale)
        crop = (not self.cache_data) or self.as_down
        hr = as_mod_crop_hw(H, W, scale) if crop else (H, W)
        lr = (round(hr[0] / scale[0]), round(hr[1] / scale[1])) if self._synthesise() else hr
        return lr, hr

    def _read_lists(self, folder: str) -> List[List[str]]:
        return [self.imgs_gt[folder]]                    # GT only: without synthesis the GT frames ARE the network input (video_test_dataset.py:308-313)

    def _load_folder(self, folder: str) -> dict:
        scale = self.opt["scale"]
        scale = tuple(scale) if isinstance(scale, (tuple, list)) else scale
        # GT: the per-item branch always mod-crops (:310,:313); the cache_data branch serves the parent's cache, which is
        # cropped only when the YAML carries `use_arbitrary_scale_downsampling` (:101-110)
        crop = scale if (not self.cache_data or self.as_down) else None
        gt = self._read(self.imgs_gt[folder], folder, crop)
        lq = arbitrary_scale_downsample(gt, scale, self.opt.get("downsampling_mode", "torch")) if self._synthesise() else gt     # :308-312
        return {"gt": gt, "lq": lq}


def build_dataset(dataset_opt):
    """lbasicsr/data/__init__.py build_dataset: DATASET_REGISTRY lookup by `type`."""
    return DATASET_REGISTRY.get(dataset_opt["type"])(dataset_opt)
