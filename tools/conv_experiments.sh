#!/bin/bash
# Timing experiments on the PRODUCT conv kernel, run on the GPU box (gpurun -- 'bash tools/conv_experiments.sh'):
# rebuilds conv_mfma.hip with -DCONV_EXP=8 + {1: no staging, 2: no fragment reads, 4: no epilogue body} and prints the
# launch time next to the per-workgroup s_memtime total.  Cycles, not microseconds, are the comparable figure: the
# variants draw different power and the shader clock moves between ~1.45 and ~2.1 GHz with it.
# Leaves the product library (CONV_EXP=0) behind.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}/savsr_amd/csrc"
build() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DCONV_EXP=$1 ${CONV_XFLAGS:-} -c conv_mfma.hip -o conv_mfma.o 2>&1 | grep -m1 error
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsavsr_hip.so conv_mfma.o osconv.o elementwise.o satu.o tail.o metrics.o resize.o api.o
}
for e in ${CONV_EXPS:-8 9 10 11 12 15 8}; do
  build $e
  echo "== CONV_EXP=$e"
  (cd ../.. && python3 tools/bench_kernels.py conv 128 64 3 --batch 6 --iters 30 --cycles 2>&1 | tail -2; python3 tools/bench_kernels.py conv 64 64 3 --batch 1 --iters 50 --cycles 2>&1 | tail -2)
done
build 0
