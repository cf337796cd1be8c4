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
from collections import OrderedDict, deque
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


# ----------------------------------------------------------------------------------------------------------------------
# Decoded-frame store: one per process, shared by every dataset.  The shipped test YAMLs define up to 42 datasets over ONE
# `dataroot_gt` (options/test/SAVSR/test_SAVSR_Vid4_asBI.yml:8-517: the same frames at 42 scales); the reference re-reads
# and re-decodes every window's 7 PNGs per output frame and per dataset (video_test_dataset.py:297-328).  Here a file is
# decoded ONCE (host, PIL, on a small thread pool that runs ahead of the GPU), uploaded ONCE as uint8 (1/4 of the fp32
# bytes over PCIe) and stays in HBM (a 1280x720 frame is 2.8 MB: 288 GB holds any test set); every dataset then takes its
# mod crop as a view and converts to fp32 on the device.
# ----------------------------------------------------------------------------------------------------------------------
import threading                                      # noqa: E402
from concurrent.futures import Future, ThreadPoolExecutor  # noqa: E402
from typing import Dict, Iterable                     # noqa: E402


def image_size(path: str) -> Tuple[int, int]:
    """(H, W) of an image file from its header (nothing is decoded: PIL reads lazily)."""
    with Image.open(path) as im:
        w, h = im.size
    return int(h), int(w)


def decode_rgb_u8(path: str) -> np.ndarray:
    """One image file -> HWC RGB uint8 (what cv2.imread returns, channel order reversed)."""
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB")))


class FrameStore:
    def __init__(self, host_bytes: Optional[int] = None, device_bytes: Optional[int] = None, workers: Optional[int] = None):
        gib = 1 << 30
        self.host_cap = int(float(os.environ.get("SAVSR_DECODE_CACHE_GB", "8")) * gib) if host_bytes is None else host_bytes
        self.dev_cap = int(float(os.environ.get("SAVSR_GT_CACHE_GB", "32")) * gib) if device_bytes is None else device_bytes
        n = workers if workers is not None else int(os.environ.get("SAVSR_DECODE_THREADS", "0"))
        if n <= 0:
            n = max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
        self.workers = n
        self._pool = ThreadPoolExecutor(max_workers=n, thread_name_prefix="savsr-decode")
        self._lock = threading.Lock()
        self._host: "OrderedDict[tuple, object]" = OrderedDict()      # key -> ndarray | Future
        self._host_bytes = 0
        self._dev: "OrderedDict[tuple, torch.Tensor]" = OrderedDict()
        self._dev_bytes = 0
        self._dev_files: Dict[tuple, int] = {}                        # file key -> devices holding it
        self._shape: Dict[tuple, Tuple[int, int]] = {}                # file key -> (H, W)
        self._lut: Dict[str, torch.Tensor] = {}
        self.stats = {"decoded": 0, "host_hits": 0, "uploaded": 0, "device_hits": 0}
        self._inflight = 0                                            # decodes submitted to the pool and not settled yet
        self._pending: "deque[tuple]" = deque()                       # (key, path) requested beyond the in-flight cap, oldest first
        self._counted: set = set()                                    # the Futures `_inflight` counts

    @staticmethod
    def _key(path: str) -> tuple:
        st = os.stat(path)
        return (osp.abspath(path), st.st_mtime_ns, st.st_size)

    def _decode(self, path: str) -> np.ndarray:
        img = decode_rgb_u8(path)
        with self._lock:
            self.stats["decoded"] += 1
        return img

    def _evict_host(self, keep) -> None:
        """(under the lock) Drop decoded arrays, oldest first, until the host cap holds; in-flight decodes are skipped, not a stop sign."""
        if self._host_bytes <= self.host_cap:
            return
        for ok in list(self._host.keys()):
            if self._host_bytes <= self.host_cap:
                break
            ov = self._host[ok]
            if ok == keep or isinstance(ov, Future):
                continue
            self._host.pop(ok)
            self._host_bytes -= ov.nbytes

    def _settle(self, k, fut: Future) -> None:
        """A decode has finished (worker thread or the waiting caller, whoever comes first): the array replaces its Future and is
        counted against SAVSR_DECODE_CACHE_GB; a failed decode is dropped so that the next request tries again.  The pool is then topped
        up from the pending queue.  Every Future `_submit` counted is un-counted exactly once -- also when its table entry has been
        replaced or cleared meanwhile (clear() during a prefetch): `_inflight` cannot stick at the cap."""
        with self._lock:
            if fut in self._counted:
                self._counted.discard(fut)
                self._inflight -= 1
            if self._host.get(k) is fut:
                if fut.cancelled() or fut.exception() is not None:
                    self._host.pop(k, None)
                else:
                    img = fut.result()
                    self._host[k] = img
                    self._host_bytes += img.nbytes
                    self._shape[k] = (int(img.shape[0]), int(img.shape[1]))
                    self._evict_host(k)
        self._top_up()

    def _submit(self, k, path: str) -> Future:
        """(under the lock) One decode into the pool; the caller attaches the done callback outside the lock."""
        fut = self._pool.submit(self._decode, path)
        self._host[k] = fut
        self._inflight += 1
        self._counted.add(fut)
        return fut

    def _top_up(self) -> None:
        """Move pending requests into the pool while fewer than 8 decodes per worker are in flight: a folder longer than the cap (100-frame
        REDS folders, or any folder with few decode threads) keeps the pool full instead of falling back to one file at a time on the
        consumer thread."""
        started = []
        with self._lock:
            while self._pending and self._inflight < 8 * self.workers:
                k, p = self._pending.popleft()
                if k in self._host or self._dev_files.get(k, 0) > 0:
                    continue
                started.append((k, self._submit(k, p)))
        for k, fut in started:
            fut.add_done_callback(lambda f, k=k: self._settle(k, f))

    def request(self, paths: Iterable[str]) -> None:
        """Start decoding `paths` in the background (no-op for files already decoded, in flight or queued).  At most 8 files per worker
        are in flight at a time -- undecoded work and arrays nobody has counted yet stay bounded --; the rest waits in a queue that every
        finished decode tops the pool up from."""
        with self._lock:
            queued = {k for k, _ in self._pending}
            for p in paths:
                k = self._key(p)
                if k in self._host or k in queued or self._dev_files.get(k, 0) > 0:
                    continue
                self._pending.append((k, p))
                queued.add(k)
        self._top_up()

    def host(self, path: str) -> np.ndarray:
        """HWC RGB uint8 of one file (blocks until its decode is done; decodes here when nobody requested it)."""
        k = self._key(path)
        with self._lock:
            ent = self._host.get(k)
            if ent is None:                             # not requested, or still in the pending queue: decode now (the queue entry is skipped later)
                ent = self._submit(k, path)
            elif not isinstance(ent, Future):
                self.stats["host_hits"] += 1
                self._host.move_to_end(k)
                self._shape[k] = (int(ent.shape[0]), int(ent.shape[1]))
                return ent
        try:
            img = ent.result()
        except Exception:
            self._settle(k, ent)                        # (drops the failed entry)
            raise
        self._settle(k, ent)
        self._shape[k] = (int(img.shape[0]), int(img.shape[1]))
        return img

    def host_shape(self, path: str) -> Tuple[int, int]:
        """(H, W) of a file that is resident on a device or decoded (decodes it otherwise)."""
        k = self._key(path)
        hw = self._shape.get(k)
        if hw is None:
            a = self.host(path)
            hw = (int(a.shape[0]), int(a.shape[1]))
        return hw

    def device(self, path: str, device: torch.device) -> torch.Tensor:
        """[H, W, 3] RGB uint8 on `device` (uploaded once per file and device)."""
        k = (self._key(path), str(device))
        t = self._dev.get(k)
        if t is not None:
            self.stats["device_hits"] += 1
            self._dev.move_to_end(k)
            return t
        from ._xfer import h2d
        arr = self.host(path)
        if not arr.flags.writeable:                # (PIL hands out read-only buffers; torch wants to be told it may not write)
            arr = arr.copy()
        t = h2d(torch.from_numpy(arr), device)
        self.stats["uploaded"] += 1
        self._dev[k] = t
        self._dev_bytes += t.numel()
        self._dev_files[k[0]] = self._dev_files.get(k[0], 0) + 1
        while self._dev_bytes > self.dev_cap and len(self._dev) > 1:
            ok, ov = self._dev.popitem(last=False)
            self._dev_bytes -= ov.numel()
            self._dev_files[ok[0]] -= 1
        with self._lock:                          # the host copy has served its purpose once the frame is resident in HBM
            ent = self._host.pop(k[0], None)
            if ent is not None and not isinstance(ent, Future):
                self._host_bytes -= ent.nbytes
        return t

    def frames_chw_f32(self, paths: Sequence[str], device: torch.device, crop_hw: Optional[Tuple[int, int]] = None) -> torch.Tensor:
        """read_img_seq (data_util.py:29-60) for files through the store: [t, 3, h, w] RGB fp32 in [0, 1] on `device`, bit for bit
        the host path's `uint8.astype(float32) / 255` (a 256-entry table built with that very numpy expression is indexed on the
        device), optionally cropped to the top-left crop_hw (the arbitrary-scale mod crop, transforms.py:48-69)."""
        dk = str(device)
        lut = self._lut.get(dk)
        if lut is None:
            from ._xfer import h2d
            lut = h2d(torch.from_numpy(np.arange(256, dtype=np.uint8).astype(np.float32) / 255.0), device)
            self._lut[dk] = lut
        out = []
        for p in paths:
            u8 = self.device(p, device)
            if crop_hw is not None:
                u8 = u8[: crop_hw[0], : crop_hw[1]]
            idx = u8.permute(2, 0, 1).reshape(-1).to(torch.int64)
            out.append(torch.index_select(lut, 0, idx).view(3, u8.shape[0], u8.shape[1]))
        return torch.stack(out, 0)

    def clear(self):
        """Forget everything decoded, queued or uploaded.  Decodes still running in the pool finish into nowhere: their Futures stay counted
        until they settle (`_settle` un-counts a Future it finds replaced), so the in-flight cap keeps meaning what it says."""
        with self._lock:
            self._host.clear()
            self._host_bytes = 0
            self._pending.clear()
        self._dev.clear()
        self._dev_files.clear()
        self._dev_bytes = 0


_STORE: Optional[FrameStore] = None


def frame_store() -> FrameStore:
    """The process-wide store (created on first use)."""
    global _STORE
    if _STORE is None:
        _STORE = FrameStore()
    return _STORE


_WRITER: Optional[ThreadPoolExecutor] = None
_PENDING: List[Future] = []


def imwrite_async(img: np.ndarray, file_path: str) -> None:
    """imwrite on a small thread pool (PNG encode is zlib work that releases the GIL): the validation loop hands the uint8
    frame over and goes on launching; flush_writes() at the end of a dataset waits for the files and re-raises a failure."""
    global _WRITER
    if _WRITER is None:
        _WRITER = ThreadPoolExecutor(max_workers=max(1, min(4, frame_store().workers)), thread_name_prefix="savsr-png")
    _PENDING.append(_WRITER.submit(imwrite, img, file_path))
    if len(_PENDING) >= 64:                           # bound the queue (a 720p frame is 2.8 MB)
        _PENDING.pop(0).result()


def flush_writes() -> None:
    while _PENDING:
        _PENDING.pop(0).result()
