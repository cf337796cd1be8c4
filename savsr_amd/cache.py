"""Device-memory contexts of the engine: per LR shape (the ~200 named channel-last feature maps, bump-allocated from arena chunks and
shared by a liveness plan) and, inside a shape, per scale (HR-sized buffers, the captured hipGraphs with their static input / output).

`ContextCache` is the part of `HipEngine` that owns them: LRU order, the byte budget shared by the engines of all HIP streams, eviction
(global LRU; dropped contexts wait in a LIMBO list until the replays that may still touch their memory have completed) and the
rate-limited trim.  The reference frees everything per frame (video_base_model.py:72-74); an arbitrary-scale sweep here keeps a bounded
working set instead.
"""
from __future__ import annotations

import weakref
from typing import Dict, List, Optional, Tuple

import torch


class ContextCache:
    """Mixin of HipEngine: (shape, scale) buffer contexts, byte budget, eviction, named buffers."""

    # Device memory is cached per LR clip shape (the ~200 named channel-last feature maps) and, inside a shape, per scale
    # (HR-sized buffers, the captured hipGraphs with their static input / output).  Both levels are small LRU caches: an
    # arbitrary-scale sweep over many (shape, scale) pairs (BASELINE configs 3 / 5) keeps a bounded working set instead of
    # pinning every size it has ever seen (the reference frees everything per frame, video_base_model.py:72-74).
    def _init_caches(self):
        from collections import OrderedDict
        # Count caps (secondary: the byte budget below is what normally decides) ...
        self.max_shapes, self.max_scales = self.knobs.cache_shapes, self.knobs.cache_scales
        # ... and the BYTE budget of everything this engine and its sibling engines (one per HIP stream) keep resident per (shape, scale):
        # arena chunks of the LR / HR buffers, the graphs' static input / output, per-pixel SATU tables.  SAVSR_CACHE_GB, default half of the
        # HBM that is free when the engine is built: a Vimeo-shaped stream (51 LR shapes x 3 streams x ~0.3 GB) stays resident, a 540x960
        # stream (5 GB per shape and stream) keeps what fits -- one knob for both instead of a shape count that suits one of them.
        gb = self.knobs.cache_gb
        if gb is not None:
            limit = int(float(gb) * (1 << 30))
        else:
            try:
                limit = int(0.5 * torch.cuda.mem_get_info(self.dev)[0])
            except RuntimeError:
                limit = 64 << 30
        # (shared with the sibling engines, clone_for_stream: the account, the use counter of the global LRU order, the engines that draw on it,
        # and the LIMBO of dropped contexts -- see _drop)
        self._budget = {"limit": max(limit, 1), "used": 0, "evictions": 0, "trim": False, "tick": 0, "engines": [weakref.ref(self)], "limbo": [], "last_trim": 0.0,
                        "limbo_peak": 0}
        self._ctx: "OrderedDict[tuple, dict]" = OrderedDict()
        self._axes: "OrderedDict[tuple, dict]" = OrderedDict()
        self._default_ctx = dict(bufs={}, scales=OrderedDict(), bytes=0, untracked=True)      # direct kernel-level calls (tests, tools) outside a forward
        self._default_sc = dict(bufs={}, graphs=None, chunk=0, bytes=0, untracked=True)
        self.hr_sched = torch.zeros(16, dtype=torch.int32, device=self.dev)     # tile-queue scratch of the SATU HR kernel (one per engine = per stream)
        self._cur, self._cur_sc = self._default_ctx, self._default_sc
        self._cur_key = None

    def _charge(self, owner: dict, nbytes: int) -> None:
        """Account `nbytes` of device memory to a (shape) or (shape, scale) context and to the shared budget."""
        owner["bytes"] = owner.get("bytes", 0) + int(nbytes)
        if not owner.get("untracked"):
            self._budget["used"] += int(nbytes)

    @staticmethod
    def _ctx_bytes(ctx: dict) -> int:
        return ctx.get("bytes", 0) + sum(sc.get("bytes", 0) for sc in ctx["scales"].values())

    def _to_limbo(self, holder: dict) -> None:
        """Eviction safety as a property.  A dropped context's graphs may still be replaying (the host runs frames ahead of the device) and its
        arena chunks / graph pools go back to the caching allocator the moment the last reference dies -- from where a sibling stream's next arena
        chunk, or this stream's next capture, can be carved while the old replay is still reading and writing them.  So nothing dies here: the
        context moves to the limbo list together with an event recorded on every stream it was used on, and is only let go once those events
        have COMPLETED (host-side query at a later _select; no synchronisation).  Everything the replays touch stays allocated until they are over."""
        evs = []
        for st in holder.pop("streams", {}).values():
            e = torch.cuda.Event()
            e.record(st)
            evs.append(e)
        limbo = self._budget["limbo"]
        limbo.append((evs, holder))
        self._budget["limbo_peak"] = max(self._budget["limbo_peak"], len(limbo))

    def _reap_limbo(self) -> None:
        """Let go of the dropped contexts whose last replays have finished (event.query(): non-blocking)."""
        limbo = self._budget["limbo"]
        if limbo:
            limbo[:] = [(evs, h) for evs, h in limbo if not all(e.query() for e in evs)]

    def _drop(self, skey: tuple) -> None:
        """Evict one shape context of THIS engine: off the account at once, its memory kept until its replays are over (_to_limbo)."""
        ctx = self._ctx.pop(skey)
        self._budget["used"] -= self._ctx_bytes(ctx)
        self._budget["evictions"] += 1
        self._budget["trim"] = True
        for sc in ctx["scales"].values():            # (the scale contexts carry the streams they replayed on)
            ctx.setdefault("streams", {}).update(sc.get("streams", {}))
        self._to_limbo(ctx)

    def _drop_scale(self, ctx: dict, ckey: tuple) -> None:
        sc = ctx["scales"].pop(ckey)
        self._budget["used"] -= sc.get("bytes", 0)
        self._budget["evictions"] += 1
        self._budget["trim"] = True
        self._to_limbo(sc)

    def _evict_to_budget(self, keep: tuple) -> None:
        """Least recently used contexts go until the shared budget holds -- GLOBALLY over the engines that share it (one per HIP stream): whole shape
        contexts first (never the one being entered, never one a sibling is inside), then, inside the current shape, its least recently used
        scales.  (Round 5 evicted this engine's own contexts only: with the bytes held by a sibling, every fresh context dropped all of its own
        and the account still stood above the limit.)"""
        b = self._budget
        while b["used"] > b["limit"]:
            victim, owner, tick = None, None, None
            for ref in b["engines"]:
                e = ref()                  # (weak: the shared account must not keep an engine -- and the graphs of its contexts -- alive in a reference cycle)
                if e is None:
                    continue
                for k, c in e._ctx.items():
                    if (e is self and k == keep) or c is e._cur:
                        continue
                    if tick is None or c.get("tick", 0) < tick:
                        victim, owner, tick = k, e, c.get("tick", 0)
            if victim is None:
                break
            owner._drop(victim)
        cur = self._ctx.get(keep)
        while cur is not None and b["used"] > b["limit"] and len(cur["scales"]) > 1:
            self._drop_scale(cur, next(iter(cur["scales"])))

    def _select(self, shape: tuple, scale) -> dict:
        """Make (clip shape, scale) the current buffer context; evicts least recently used ones beyond the byte budget / the count caps."""
        skey = tuple(int(v) for v in shape)
        ctx = self._ctx.get(skey)
        fresh = ctx is None
        if fresh:
            from collections import OrderedDict
            ctx = dict(bufs={}, scales=OrderedDict(), bytes=0)
            self._ctx[skey] = ctx
            while len(self._ctx) > self.max_shapes:
                self._drop(next(iter(self._ctx)))
        else:
            self._ctx.move_to_end(skey)
        ckey = (float(scale[0]), float(scale[1]))
        sc = ctx["scales"].get(ckey)
        if sc is None:
            fresh = True
            sc = dict(bufs={}, graphs=None, chunk=0, bytes=0)        # (a scale context holds the HR-sized buffers: exact-size allocations)
            ctx["scales"][ckey] = sc
            while len(ctx["scales"]) > self.max_scales:
                self._drop_scale(ctx, next(iter(ctx["scales"])))
        else:
            ctx["scales"].move_to_end(ckey)
        b = self._budget
        b["tick"] += 1
        ctx["tick"] = sc["tick"] = b["tick"]
        cs = torch.cuda.current_stream()
        sc.setdefault("streams", {})[cs.cuda_stream] = cs          # every stream this context's launches / replays were enqueued on (_to_limbo)
        self._cur, self._cur_sc, self._cur_key = ctx, sc, (skey, ckey)
        self._reap_limbo()
        if fresh:
            self._evict_to_budget(skey)
            self._maybe_trim()
        return sc

    TRIM_MIN_INTERVAL_S = 2.0

    def _maybe_trim(self) -> None:
        """Hand cached-but-unused device memory back to the driver after evictions -- only when the allocator holds more than the budget
        beyond what the process keeps outside the engine's account (the frame store's decoded-frame cache, io.FrameStore), at most once per
        TRIM_MIN_INTERVAL_S (empty_cache() is device-synchronous), and never with a capture under way.  Dropped contexts still in limbo are not
        freed by it; they go when their replays are over."""
        b = self._budget
        if not b["trim"] or torch.cuda.is_current_stream_capturing():
            return
        import time as _time
        now = _time.monotonic()
        if now - b["last_trim"] < self.TRIM_MIN_INTERVAL_S:
            return
        from . import io as _sio
        other = _sio._STORE._dev_bytes if _sio._STORE is not None else 0
        if torch.cuda.memory_reserved(self.dev) - other > b["limit"]:
            torch.cuda.empty_cache()
            b["last_trim"] = now
        b["trim"] = False

    FORGET_ABOVE = 0.9          # forget() without `always`: only when the shared account stands above this share of its limit

    def forget(self, hw: Tuple[int, int], scale, always: bool = False) -> int:
        """The caller is DONE with (LR frame size, scale) -- a finished (dataset, folder) unit of a YAML job, which no later dataset revisits
        (every scale has its own LR size).  With `always` (the job walks many units: models.validate_job) or while the shared account stands above
        FORGET_ABOVE of its limit, the contexts of that unit are dropped on every engine of the budget -- into the limbo, i.e. released once their
        last replays are over.  Why: a capture gets slower with the number of contexts ALIVE in the process -- the one-process pass over the
        shipped Vid4 YAML (168 units, ~640 graph sets alive by the end) spent 19-22 of 83-88 s in 640 captures of 30-35 ms; with finished units
        forgotten a capture costs 6.5 ms as in a rank of eight and the pass takes 73 s (profiles/r06_emu_world1_forget.json).  Small jobs keep
        everything resident (a second pass over them replays).  Returns the contexts dropped."""
        b = self._budget
        if not always and b["used"] <= self.FORGET_ABOVE * b["limit"]:
            return 0
        ckey = (float(scale[0]), float(scale[1]))
        n = 0
        for ref in b["engines"]:
            e = ref()
            if e is None:
                continue
            for skey in [k for k in e._ctx if tuple(k[-2:]) == (int(hw[0]), int(hw[1]))]:
                ctx = e._ctx[skey]
                if ckey in ctx["scales"]:
                    if len(ctx["scales"]) == 1:
                        e._drop(skey)
                    else:
                        e._drop_scale(ctx, ckey)
                    b["evictions"] -= 1          # (a planned release, not an eviction under pressure)
                    b["forgotten"] = b.get("forgotten", 0) + 1
                    n += 1
                    if e._cur is ctx:
                        e._cur, e._cur_sc, e._cur_key = e._default_ctx, e._default_sc, None
        if n:
            self._reap_limbo()
        return n

    def cache_stats(self) -> dict:
        """Resident contexts of THIS engine; `bytes` = device memory they hold (arena chunks + graph I/O + per-pixel tables), `budget_*` = the
        account shared with the sibling engines."""
        return {"shapes": len(self._ctx), "scales": sum(len(c["scales"]) for c in self._ctx.values()), "axes": len(self._axes),
                "bytes": sum(self._ctx_bytes(c) for c in self._ctx.values()),
                "budget_used": self._budget["used"], "budget_limit": self._budget["limit"], "evictions": self._budget["evictions"],
                "limbo": len(self._budget["limbo"]), "limbo_peak": self._budget["limbo_peak"], "forgotten": self._budget.get("forgotten", 0)}

    ARENA_CHUNK = 64 << 20      # bytes per arena chunk (larger requests get a chunk of their own)

    def _get_buf(self, owner: dict, name: str, shape: tuple) -> torch.Tensor:
        """Named fp32 buffer of a context, carved out of the context's ARENA: the ~110 feature maps of a clip shape are never
        freed one by one (the context is dropped as a whole), so they are bump-allocated from a few large device allocations
        instead of one allocator round trip each -- a new LR shape (every folder x scale of the YAML sweep is one) costs a
        handful of hipMallocs, not a hundred.  256-byte aligned (the kernels ask for 16)."""
        store = owner["bufs"]
        key = (name,) + tuple(shape)
        t = store.get(key)
        if t is None:
            n = 1
            for d in shape:
                n *= int(d)
            nbytes1 = (4 * n + 255) & ~255
            nbytes = nbytes1 * self.nb                    # (nb copies: clip b's lives nbytes1 * b further on)
            free = owner.get("free", {}).get(nbytes) if not owner.get("sealed") else None
            if free:
                raw = free.pop()                          # a slot whose previous owner's last reader is already enqueued (release())
            else:
                arena = owner.setdefault("arena", [])
                if not arena or arena[-1][1] + nbytes > arena[-1][0].numel():
                    # (nb clips per launch sequence: nb x the chunk, so that a batched context costs the same handful of allocations -- 26 64-MiB
                    # hipMallocs inside a capture were 40 ms of a 45 ms capture)
                    arena.append([torch.empty(max(nbytes, owner.get("chunk", self.ARENA_CHUNK * self.nb)), device=self.dev, dtype=torch.uint8), 0])
                    self._charge(owner, arena[-1][0].numel())
                    if self.knobs.poison:      # diagnostics: every fresh chunk full of NaN (inside a capture: a memset node in front of the frame's
                        arena[-1][0].view(torch.float32).fill_(float("nan"))      # kernels, replayed with it) -- a read of a never-written float shows
                chunk, off = arena[-1]
                raw = chunk[off:off + nbytes]
                arena[-1][1] = off + nbytes
            t = raw[:4 * n].view(torch.float32).view(shape)
            store[key] = t
            owner.setdefault("raw", {})[t.data_ptr()] = raw
            if self.nb > 1:
                self._bstride[t.data_ptr()] = nbytes1
            else:
                self._bstride.pop(t.data_ptr(), None)     # (an address an evicted batched context used to own)
        return t

    # Buffer liveness.  The launch sequence of a clip shape is static, so the assignment of named buffers to memory is decided ONCE, on the
    # context's first frame: release(x) there returns x's slot to a per-size free list (every reader of x has been enqueued on the one
    # stream of this engine, and the stream is in-order, so a later writer cannot overtake them), and the next new name of that size takes
    # it.  After the first frame the context is sealed: names keep their slots (captured hipGraphs hold the pointers), release() does
    # nothing, and a name first seen later gets fresh memory.  What is released, and where: the network pieces below.
    def release(self, *xs) -> None:
        owner = self._cur
        if owner.get("sealed") or not self.reuse_buffers:
            return
        raws = owner.get("raw", {})
        for x in xs:
            t = x if (x is None or isinstance(x, torch.Tensor)) else x.t          # (a launch.Src slice, or the tensor itself)
            raw = raws.get(t.data_ptr()) if t is not None else None
            if raw is not None and not any(raw.data_ptr() == r.data_ptr() for r in owner.setdefault("free", {}).setdefault(raw.numel(), [])):
                owner["free"][raw.numel()].append(raw)

    def seal_buffers(self) -> None:
        """End of a context's first frame: the name -> memory assignment is final."""
        self._cur["sealed"] = True
        self._cur.pop("free", None)

    def _abort_frame(self) -> None:
        """A frame's launch sequence raised (allocation failure, a capture error, ...).  While a shape's buffer plan is still being made (first
        frame, not sealed) a partly consumed free list would hand live memory to the next new name on a retry -- the plan is all or nothing:
        the whole shape context goes.  A sealed shape keeps its plan; only the half-built (shape, scale) context is dropped."""
        if self._cur_key is None:
            return
        skey, ckey = self._cur_key
        ctx = self._ctx.get(skey)
        if ctx is not None:
            if not ctx.get("sealed"):
                self._drop(skey)
            elif ckey in ctx["scales"] and not ctx["scales"][ckey].get("graphs"):
                self._drop_scale(ctx, ckey)
        self._cur, self._cur_sc, self._cur_key = self._default_ctx, self._default_sc, None

    def buf(self, name: str, *shape: int) -> torch.Tensor:
        """Named LR-sized buffer of the current clip shape."""
        return self._get_buf(self._cur, name, shape)

    def sbuf(self, name: str, *shape: int) -> torch.Tensor:
        """Named buffer whose size depends on the scale (HR-sized), owned by the current (shape, scale) context."""
        return self._get_buf(self._cur_sc, name, shape)
