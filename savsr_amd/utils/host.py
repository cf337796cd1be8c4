"""Host facts shared by bench.py and the test session."""
from __future__ import annotations

import os


def effective_cpus() -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (a 16-core quota on a 128-core host would otherwise
    start 128 ATen threads that take turns on 16 cores)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"
