"""Every run-time switch of the engine in ONE place.

`EngineConfig.from_env()` reads the `SAVSR_*` environment variables once, when an engine is built (tests set them with
monkeypatch before building a network); nothing else in the engine looks at the environment.  `knobs()` lists the values that
differ from the product defaults -- bench.py prints them into its JSON line (`config.knobs`), so a measurement names the
configuration it was taken in.  The product path is the default of every field; the others exist for A/B measurements and
diagnostics (DESIGN.md section 9 says what each one showed).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, fields
from typing import Optional


def _flag(name: str, default: bool) -> bool:
    v = os.environ.get(name)
    return default if v is None else v != "0"


@dataclass
class EngineConfig:
    # ---- flow ------------------------------------------------------------------------------------------------------------------
    streams: int = 3                       # SAVSR_STREAMS: launch units of a batch / of forward_many in flight on separate HIP streams (small frames)
    streams_large: int = 2                 # SAVSR_STREAMS_LARGE: ... when the LR frames of a call average at least `streams_large_px` pixels
    streams_large_px: int = 40000          # SAVSR_STREAMS_LARGE_PX
    clip_batch: int = 4                    # SAVSR_CLIP_BATCH: clips of one (shape, scale) per launch sequence (capped by the library's batch limits: 24 convs / 8 OSConvs per launch)
    clip_batch_max_px: int = 70400         # SAVSR_CLIP_BATCH_MAX_PX: LR frames up to this many pixels share launch sequences.  200 000 reads +2 % on the steady pass of bench.py's 6-scale job (154.6 -> 158.2 frames/s) but makes 5-7 GB contexts of the 300-340 x 450-510 frames of a shipped YAML's low scales: their first-touch allocation inside the capture costs 150-250 ms a context and the byte budget starts evicting -- a rank of eight holding such units: cold pass 12.8 -> 10.3 s, second pass 10.9 -> 8.9 s with 70 400 (profiles/r06_rank4_*.json)
    graphs: bool = True                    # SAVSR_GRAPHS: replay captured hipGraphs (0: issue every launch from Python -- diagnostics)
    capture_after: int = 0                 # SAVSR_CAPTURE_AFTER: a context's first n frames run eagerly, then the graphs are captured
    # ---- kernel forms ----------------------------------------------------------------------------------------------------------
    conv_wy: bool = True                   # SAVSR_CONV_WY: static 3x3 convs (and the OSConv dynamic convs) in the Winograd F(2,3)-along-y form where a launch fills the chip
    wy_min_tiles: int = 200                # SAVSR_WY_MIN_TILES: 16-row tiles a launch needs for the Winograd form (latency flow) ...
    wy_min_tiles_tp: int = 100             # SAVSR_WY_MIN_TILES_TP: ... and in the throughput flow
    satu_q: bool = True                    # SAVSR_SATU_Q: SATU HR stage in the row-summed tail form (9 planes + seams); 0: the 27-plane form
    osconv_fused: bool = False             # SAVSR_OSCONV_FUSED: OSConv weight generation as one launch (measured slower: DESIGN.md section 9)
    reuse_buffers: bool = True             # SAVSR_REUSE_BUFFERS: liveness-planned LR buffers; 0: every name its own memory
    # ---- caches ----------------------------------------------------------------------------------------------------------------
    cache_shapes: int = 256                # SAVSR_CACHE_SHAPES: LR shapes resident per engine (count cap; the byte budget normally decides)
    cache_scales: int = 48                 # SAVSR_CACHE_SCALES: scales resident per shape
    cache_gb: Optional[float] = None       # SAVSR_CACHE_GB: byte budget of the resident contexts, all streams; None = half of the free HBM at engine build
    hr_plans: bool = True                  # SAVSR_HR_PLANS: use savsr_amd/hr_plans.json (the HR stage's measured launch plan per scale)
    # ---- diagnostics -----------------------------------------------------------------------------------------------------------
    hr_tile: Optional[str] = None          # SAVSR_HR_TILE "rows,cols32": force the HR stage's tile
    hr_variant: Optional[int] = None       # SAVSR_HR_VARIANT: force the HR stage's wave split
    hr_static: bool = False                # SAVSR_HR_STATIC: static tile order instead of the tile queue
    hr_print_plans: bool = False           # SAVSR_HR_PRINT_PLANS: print what every feasible HR plan measured
    profile_capture: bool = False          # SAVSR_PROFILE_CAPTURE: print the host time of every graph capture
    poison: bool = False                   # SAVSR_POISON: fresh arena chunks are filled with NaN (a read of a never-written value becomes visible)

    @classmethod
    def from_env(cls) -> "EngineConfig":
        e = os.environ.get
        gb = e("SAVSR_CACHE_GB")
        var = e("SAVSR_HR_VARIANT")
        return cls(
            streams=max(1, int(e("SAVSR_STREAMS", "3"))),
            streams_large=max(1, int(e("SAVSR_STREAMS_LARGE", e("SAVSR_STREAMS", "2")))),
            streams_large_px=int(e("SAVSR_STREAMS_LARGE_PX", "40000")),
            clip_batch=max(1, int(e("SAVSR_CLIP_BATCH", "4"))),
            clip_batch_max_px=int(e("SAVSR_CLIP_BATCH_MAX_PX", "70400")),
            graphs=_flag("SAVSR_GRAPHS", True),
            capture_after=max(0, int(e("SAVSR_CAPTURE_AFTER", "0"))),
            conv_wy=_flag("SAVSR_CONV_WY", True),
            wy_min_tiles=int(e("SAVSR_WY_MIN_TILES", "200")),
            wy_min_tiles_tp=int(e("SAVSR_WY_MIN_TILES_TP", "100")),
            satu_q=_flag("SAVSR_SATU_Q", True),
            osconv_fused=_flag("SAVSR_OSCONV_FUSED", False),
            reuse_buffers=_flag("SAVSR_REUSE_BUFFERS", True),
            cache_shapes=max(1, int(e("SAVSR_CACHE_SHAPES", "256"))),
            cache_scales=max(1, int(e("SAVSR_CACHE_SCALES", "48"))),
            cache_gb=None if gb is None else float(gb),
            hr_plans=_flag("SAVSR_HR_PLANS", True),
            hr_tile=e("SAVSR_HR_TILE") or None,
            hr_variant=None if not var else int(var),
            hr_static=e("SAVSR_HR_STATIC") == "1",
            hr_print_plans=bool(e("SAVSR_HR_PRINT_PLANS")),
            profile_capture=bool(e("SAVSR_PROFILE_CAPTURE")),
            poison=e("SAVSR_POISON") == "1",
        )

    def knobs(self) -> dict:
        """The fields that differ from the product defaults (empty for the product configuration)."""
        d = EngineConfig()
        return {f.name: getattr(self, f.name) for f in fields(self) if getattr(self, f.name) != getattr(d, f.name)}
