"""Test datasets under the reference's registry names: `type: ASVideoTestDataset` /
`type: VideoTestDataset` of options/test/SAVSR/*.yml resolve here through DATASET_REGISTRY.

Contract mirrored from lbasicsr/data/video_test_dataset.py:12-141 (VideoTestDataset) and :273-328
(ASVideoTestDataset): the folder scan, `data_info` (lq_path / gt_path / folder / idx 'i/n' / border), the
7-frame window with reflection padding (data_util.py:63-112 = harness.window_indices), the arbitrary-scale mod
crop of the ground truth (transforms.py:48-69) and the 'BI' LR synthesis (data_util.py:371-420).

MI355X-first differences (results identical, work placed differently):
  * PNG decode is host work (PIL); everything after it is on the GPU.  A folder's mod-cropped GT frames are
    uploaded ONCE and stay in HBM (a 1280x720 frame is 11 MB; 288 GB holds any test folder), and each frame's LR
    version is synthesised ONCE by csrc/resize.hip -- the reference re-reads and re-resizes the 7 frames of every
    window on the CPU (7x the decode + resize work per output frame).
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
from .harness import window_indices
from .registry import DATASET_REGISTRY
from .resize_gpu import arbitrary_scale_downsample

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

    # ---- per-folder residency -----------------------------------------------------------------------------------
    def _folder(self, folder: str) -> dict:
        """GT frames (mod-cropped when the flow asks for it) and LR frames of one folder, resident on the GPU."""
        ent = self._resident.get(folder)
        if ent is None:
            ent = self._load_folder(folder)
            self._resident[folder] = ent
            while len(self._resident) > _MAX_CACHED_FOLDERS:
                self._resident.popitem(last=False)
        else:
            self._resident.move_to_end(folder)
        return ent

    def _load_folder(self, folder: str) -> dict:
        gt = sio.read_img_seq(self.imgs_gt[folder], require_as_mod_crop=self.as_down, scale=self.scale if self.as_down else None)
        lq = sio.read_img_seq(self.imgs_lq[folder], require_as_mod_crop=self.as_down, scale=self.scale if self.as_down else None)
        return {"gt": gt.to(self.device), "lq": lq.to(self.device)}

    def __getitem__(self, index):
        folder = self.data_info["folder"][index]
        idx, max_idx = (int(v) for v in self.data_info["idx"][index].split("/"))
        ent = self._folder(folder)
        sel = window_indices(idx, max_idx, self.opt["num_frame"], padding=self.opt["padding"])
        if max(sel) >= max_idx or min(sel) < 0:      # a folder shorter than the padding reach (the reference fails the same way, on the host)
            raise IndexError(f"folder '{folder}' has {max_idx} frames: too few for a {self.opt['num_frame']}-frame '{self.opt['padding']}' window")
        out = {"lq": ent["lq"][sel], "gt": ent["gt"][idx], "folder": folder, "idx": self.data_info["idx"][index],
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
        mode = self.opt.get("downsampling_mode", "torch")
        if self._synthesise() and mode != "torch":
            raise NotImplementedError(f"downsampling_mode '{mode}': only 'torch' (bicubic, antialias) is implemented -- the mode of "
                                      "every shipped test YAML (data_util.py:405-412)")

    def _synthesise(self) -> bool:
        # cache_data=True synthesises unconditionally (video_test_dataset.py:303-306); otherwise the YAML switch decides (:311-312)
        return bool(self.opt.get("cache_data")) or bool(self.opt.get("use_arbitrary_scale_downsampling"))

    def _load_folder(self, folder: str) -> dict:
        scale = self.opt["scale"]
        gt = sio.read_img_seq(self.imgs_gt[folder], require_as_mod_crop=True, scale=scale).to(self.device)
        if self._synthesise():
            lq = arbitrary_scale_downsample(gt, tuple(scale) if isinstance(scale, (tuple, list)) else scale)
        else:
            lq = gt              # :308-310 without the switch: the mod-cropped GT frames are the network input
        return {"gt": gt, "lq": lq}


def build_dataset(dataset_opt):
    """lbasicsr/data/__init__.py build_dataset: DATASET_REGISTRY lookup by `type`."""
    return DATASET_REGISTRY.get(dataset_opt["type"])(dataset_opt)
