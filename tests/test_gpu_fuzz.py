"""Randomised whole-network parity (tools/fuzz_network.py with a fixed seed): 32 random LR sizes (odd / even, down to 4 x 5) and
scale pairs in [1.05, 4.3] against the CPU oracle, max-abs < 5e-5, eager == captured == replayed bits.  A 200-case sweep of the same
generator ran clean in round 3 (worst 1.09e-5)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_sizes_and_scales_vs_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_network.py"), "--cases", "32", "--seed", "3", "--max-side", "48"],
                       capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.strip().splitlines()[-5:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "32 cases, worst max-abs" in tail
