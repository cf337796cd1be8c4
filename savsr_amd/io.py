"""Checkpoint and image formats on either side of the path: host code, PIL + torch only.

  load_network        lbasicsr/models/base_model.py:293-319 (+ _print_different_keys_loading :285-291):
                      `{'params': OrderedDict}` torch pickles, params_ema fallback, 'module.' prefixes, strict / non-strict
  read_img_seq        lbasicsr/data/data_util.py:29-60: sorted folder (or list) of images -> [t, c, h, w] RGB fp32 in [0, 1]
  imwrite / imread    lbasicsr/utils/img_util.py:114-153 (cv2 BGR uint8 HWC convention; PNG is lossless, so PIL's decode of
                      the same file is the same array with the channel order reversed)
  result_img_path     lbasicsr/models/video_base_model.py:79-92: results/<...>/visualization/<dataset>/<folder>/<img>_<suffix>.png
"""
from __future__ import annotations

import os
import os.path as osp
from collections import OrderedDict
from copy import deepcopy
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from PIL import Image

from .resize_gpu import as_mod_crop_hw


def load_network(net: torch.nn.Module, load_path: str, strict: bool = True, param_key: Optional[str] = "params") -> None:
    """base_model.py:293-319.  `net` may be wrapped (DataParallel / DDP): the bare `.module` is loaded."""
    net = net.module if hasattr(net, "module") and isinstance(net.module, torch.nn.Module) else net
    load_net = torch.load(load_path, map_location=lambda storage, loc: storage)
    if param_key is not None:
        if param_key not in load_net and "params" in load_net:
            param_key = "params"                      # 'Loading: params_ema does not exist, use params.'
        load_net = load_net[param_key]
    for k, v in deepcopy(load_net).items():           # remove unnecessary 'module.'
        if k.startswith("module."):
            load_net[k[7:]] = v
            load_net.pop(k)
    if not strict:                                    # :285-291: tensors of a different size are ignored, not loaded
        crt = net.state_dict()
        for k in set(crt.keys()) & set(load_net.keys()):
            if crt[k].size() != load_net[k].size():
                load_net[k + ".ignore"] = load_net.pop(k)
    net.load_state_dict(load_net, strict=strict)


def save_network(net: torch.nn.Module, save_path: str, param_key: str = "params") -> None:
    """base_model.py:222-260 (the format only): `{param_key: OrderedDict of CPU tensors}`, 'module.' stripped."""
    net = net.module if hasattr(net, "module") and isinstance(net.module, torch.nn.Module) else net
    sd = OrderedDict((k[7:] if k.startswith("module.") else k, v.detach().cpu()) for k, v in net.state_dict().items())
    os.makedirs(osp.dirname(osp.abspath(save_path)), exist_ok=True)
    torch.save({param_key: sd}, save_path)


def imread(path: str, float32: bool = False) -> np.ndarray:
    """cv2.imread(path) semantics for 8-bit colour files: HWC, BGR, uint8 (float32=True: / 255 as fp32)."""
    img = np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1]
    img = np.ascontiguousarray(img)
    return img.astype(np.float32) / 255.0 if float32 else img


def imwrite(img: np.ndarray, file_path: str, auto_mkdir: bool = True) -> None:
    """img_util.py:135-153: HWC BGR (or HW grey) uint8 -> file; the extension picks the format (PNG in the test flow)."""
    if auto_mkdir:
        os.makedirs(osp.abspath(osp.dirname(file_path)), exist_ok=True)
    if img.dtype != np.uint8:
        raise IOError("Failed in writing images.")    # cv2.imwrite would return False for a float array
    pil = Image.fromarray(img if img.ndim == 2 else np.ascontiguousarray(img[:, :, ::-1]))
    pil.save(file_path)


def scandir(dir_path: str, suffix: Optional[Union[str, Tuple[str, ...]]] = None, full_path: bool = False) -> List[str]:
    """utils/misc.py:52-94 (non-recursive): regular, non-hidden files, optionally filtered by suffix; sorted by the caller."""
    out = []
    for entry in os.scandir(dir_path):
        if entry.name.startswith(".") or not entry.is_file():
            continue
        if suffix is None or entry.path.endswith(suffix):
            out.append(entry.path if full_path else osp.relpath(entry.path, dir_path))
    return out


def read_img_seq(path: Union[str, Sequence[str]], require_as_mod_crop: bool = False, scale=None, return_imgname: bool = False):
    """data_util.py:29-60: -> Tensor [t, c, h, w], RGB, fp32 in [0, 1] (files in sorted order)."""
    img_paths = list(path) if isinstance(path, (list, tuple)) else sorted(scandir(path, full_path=True))
    imgs = [imread(v, float32=True) for v in img_paths]
    if require_as_mod_crop:
        cropped = []
        for im in imgs:
            h, w = as_mod_crop_hw(im.shape[0], im.shape[1], scale)
            cropped.append(im[:h, :w, ...])
        imgs = cropped
    t = torch.stack([torch.from_numpy(np.ascontiguousarray(im[:, :, ::-1].transpose(2, 0, 1))) for im in imgs], 0)    # bgr2rgb, HWC -> CHW
    if return_imgname:
        return t, [osp.splitext(osp.basename(p))[0] for p in img_paths]
    return t


def result_img_path(visualization_root: str, dataset_name: str, folder: str, lq_path: str, name: str, suffix: Optional[str] = None) -> str:
    """video_base_model.py:79-92."""
    if "vimeo" in dataset_name.lower():
        sp = lq_path.split("/")
        img_name = f'{sp[-3]}_{sp[-2]}_{sp[-1].split(".")[0]}'
    else:
        img_name = osp.splitext(osp.basename(lq_path))[0]
    return osp.join(visualization_root, dataset_name, folder, f"{img_name}_{suffix if suffix else name}.png")
