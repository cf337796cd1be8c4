"""Host -> device copies that do not queue behind the compute stream.

`tensor.to(device)` from pageable host memory is enqueued on the CURRENT stream and blocks the host until it has run, i.e.
until every frame still in flight on that stream has finished (~5 ms each): a validation loop that uploads a PNG, a window
index table or a SATU coordinate table per step then runs host and GPU in lock-step (4.4 s of 7.6 s in a cProfile of the
YAML workflow's cold pass).  `h2d` issues the copy on a dedicated copy stream: it waits for nothing but earlier copies, and it
has completed when the call returns (PyTorch synchronises the copy stream after a pageable copy), so the result can be used
on any stream right away.
"""
from __future__ import annotations

from typing import Dict

import torch

_COPY_STREAMS: Dict[str, "torch.cuda.Stream"] = {}


def h2d(t: torch.Tensor, device, dtype=None) -> torch.Tensor:
    device = torch.device(device)
    if device.type != "cuda" or t.is_cuda:
        return t.to(device, dtype) if dtype is not None else t.to(device)
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("host -> device copy requested inside a hipGraph capture")
    key = str(device)
    s = _COPY_STREAMS.get(key)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _COPY_STREAMS[key] = s
    with torch.cuda.stream(s):
        d = t.to(device, dtype) if dtype is not None else t.to(device)
    d.record_stream(torch.cuda.current_stream(device))     # the consumer's stream (allocator bookkeeping only)
    return d
