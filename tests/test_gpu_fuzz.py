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


def test_random_sizes_and_scales_vs_oracle_winograd_everywhere():
    """The same sweep (another seed) with EVERY eligible conv launch forced into the Winograd F(2,3)-along-y form (SAVSR_WY_MIN_TILES=1: at
    these small sizes the engine would otherwise pick the direct kernel for all of them): the static 3x3 convs and the OSConv dynamic convs
    through conv_wy.hip / osconv_aggregate_wy_kernel on ragged shapes -- tiles with rows below the image, partial columns, one-chunk and
    20-chunk convs."""
    env = dict(os.environ, SAVSR_WY_MIN_TILES="1", SAVSR_WY_MIN_TILES_TP="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_network.py"), "--cases", "20", "--seed", "11", "--max-side", "48"],
                       capture_output=True, text=True, timeout=900, env=env)
    tail = "\n".join(r.stdout.strip().splitlines()[-5:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "20 cases, worst max-abs" in tail


def test_random_medium_frames_in_the_throughput_flow_vs_oracle():
    """tools/fuzz_network.py --throughput: 16 random frames of 33..200 x 64..352 (widths mostly not multiples of 32, a quarter of the heights
    leaving <= 2 row pairs: strip tiles in the Winograd-y launches) at random scale pairs, two clips per forward_many call -- clip 0 against the
    oracle (< 5e-5), the repeated call and the lone-clip call bit-identical.  250 cases of the same generator ran clean in round 6 (worst 1.34e-5)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_network.py"), "--throughput", "--cases", "16", "--seed", "5"],
                       capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.strip().splitlines()[-5:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "16 cases, worst max-abs" in tail


def test_every_scale_pair_of_the_reference_lists_vs_oracle():
    """tools/scale_list_sweep.py in the suite (the lists were sampled, not closed): all 42 scale pairs of the shipped YAMLs at
    LR 180 x 320 and all 60 Vimeo90K training pairs at their LR sizes, HIP forward against the CPU oracle -- shape, max-abs < 5e-5, |dPSNR-Y| <= 1e-3 dB,
    |dSSIM-Y| <= 1e-4 against one synthetic GT, bitwise rerun.  ~170 s on the box's 16 host cores (the oracle is the cost)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_list_sweep.py")], capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.strip().splitlines()[-4:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "# 102 cases, 0 failed" in tail, tail
