#!/bin/bash
# PMC passes over the SATU launches (gpurun -- 'bash tools/pmc_satu.sh OUTDIR').  Counter passes are collected in their own runs
# with --kernel-trace only, as the pool requires; HBM bytes per the guide: FETCH_SIZE x 2 for 16-B/lane streaming reads, WRITE_SIZE as read.
set -u
OUT=${1:-gpurun_out/pmc_satu}
export SAVSR_HR_VARIANT=${SAVSR_HR_VARIANT:-1} SAVSR_HR_TILE=${SAVSR_HR_TILE:-20,2}     # the plan the engine measures fastest at 180x320 x4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  if [ -z "${PMC_TARGET:-}" ]; then
    # SATU: the product's LR / HR launches on frame-shaped tensors, with the HR launch plan FORCED (SAVSR_HR_VARIANT / SAVSR_HR_TILE:
    # the engine would otherwise time every feasible plan once, and those launches would pollute the per-launch means)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$R/$OUT/p$i" -- python3 "$R/tools/time_satu.py" --iters 8 --reps 1 > "$R/$OUT/p$i.log" 2>&1 || echo "pass $i ($set) failed" >> "$R/$OUT/failed.log"
  else
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$R/$OUT/p$i" -- python3 "$R/tools/bench_kernels.py" $PMC_TARGET --iters 8 > "$R/$OUT/p$i.log" 2>&1 || echo "pass $i ($set) failed" >> "$R/$OUT/failed.log"
  fi
done
cd "$R"
python3 tools/pmc_summary.py $OUT/p* > $OUT/summary.csv
# keep only the summary (the per-dispatch CSVs are large)
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
