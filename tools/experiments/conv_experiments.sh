#!/bin/bash
# ARCHIVED with the knobs it drives: apply tools/experiments/conv_mfma_switches.patch first (git apply), then
# gpurun -- 'bash tools/experiments/conv_experiments.sh'.  Timing experiments on the PRODUCT conv kernel:
# builds conv_mfma.hip with -DCONV_EXP=8 + {1: no staging, 2: no fragment reads, 4: no epilogue body} into a SEPARATE
# object directory and library (savsr_amd/csrc/exp/, libsavsr_hip_exp.so -- the product objects and libsavsr_hip.so are
# never touched) and prints the launch time next to the per-workgroup s_memtime total.  Cycles, not microseconds, are the
# comparable figure: the variants draw different power and the shader clock moves between ~1.45 and ~2.1 GHz with it.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd "$ROOT"
export SAVSR_LIB_PATH="$ROOT/savsr_amd/csrc/libsavsr_hip_exp.so"      # diagnostics-only override honoured by savsr_amd/_lib.py
for e in ${CONV_EXPS:-8 9 10 11 12 15 8}; do
  EXTRA_FLAGS="-DCONV_EXP=$e ${CONV_XFLAGS:-}" OBJDIR=exp OUT=libsavsr_hip_exp.so bash savsr_amd/csrc/build.sh >/dev/null || exit 1
  echo "== CONV_EXP=$e"
  python3 tools/bench_kernels.py conv 128 64 3 --batch 6 --iters 30 --cycles 2>&1 | tail -2
  python3 tools/bench_kernels.py conv 64 64 3 --batch 1 --iters 50 --cycles 2>&1 | tail -2
done
