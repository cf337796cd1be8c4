cd $GRAFT_REPO_ROOT/savsr_amd/csrc
export TMPDIR=/tmp
for e in 0 1 2 3 0; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DLR_EXP=$e -c satu.hip -o satu.o 2>&1 | grep -m1 error
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsavsr_hip.so conv_mfma.o osconv.o elementwise.o satu.o tail.o metrics.o resize.o api.o
  cd ../..
  rm -rf gpurun_out/lrx
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lrx -- python3 tools/bench_kernels.py satu --iters 40 > gpurun_out/lrx.log 2>&1
  f=$(find gpurun_out/lrx -name "*kernel_stats.csv" | head -1)
  echo "== LR_EXP=$e"; grep "satu_lr" $f | cut -d, -f1-7
  rm -rf gpurun_out/lrx
  cd savsr_amd/csrc
done
