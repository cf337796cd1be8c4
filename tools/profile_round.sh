#!/bin/bash
# The round's evidence set in one gpurun call:  gpurun -- 'bash tools/profile_round.sh gpurun_out/r06_prof'
#   bench_stats.csv      rocprofv3 --kernel-trace --stats of the DRIVER's command (python3 bench.py --gpus 1 --steps 20 --warmup 5)
#   satu_stats.csv       ... of the SATU launches alone (tools/time_satu.py, HR plan forced)
#   satu_pmc.csv         7 PMC passes over the SATU launches (tools/pmc_satu.sh)
#   conv_wy_pmc.csv      7 PMC passes over the dominant conv launch (6 x 128->64) in the form the product runs (Winograd-y)
#   conv_pmc.csv         ... and in the direct form
#   satu_traffic.json    profiles/satu_traffic.json regenerated from satu_pmc.csv, stamped with the library's SATU source hash
#   bench_trace_busy.log tools/trace_busy.py over the kernel trace of the driver's command (union-busy, kernels in flight, idle gaps)
#   bench_line.json      the bench line of the same lease, un-profiled
set -u
OUT=${1:-gpurun_out/r06_prof}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$R/$OUT"
cd "$R"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
export SAVSR_HR_VARIANT=${SAVSR_HR_VARIANT:-1} SAVSR_HR_TILE=${SAVSR_HR_TILE:-20,2}
python3 tools/time_satu.py --iters 50 --reps 3 --warm-frames 40 > "$OUT/satu_events.log" 2>&1
unset SAVSR_HR_VARIANT SAVSR_HR_TILE
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/bench_prof" -- python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$R/$OUT/bench_prof.json" 2> "$R/$OUT/bench_prof.err"
SAVSR_HR_VARIANT=1 SAVSR_HR_TILE=20,2 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/satu_prof" -- python3 "$R/tools/time_satu.py" --iters 50 --reps 3 --warm-frames 40 > "$R/$OUT/satu_prof.log" 2>&1
cd "$R"
python3 tools/trace_busy.py "$OUT/bench_prof" --last-s 6.0 > "$OUT/bench_trace_busy.log" 2>&1      # GPU-busy fraction / kernels in flight over the timed regions
for d in bench_prof satu_prof; do
  f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${d%_prof}_stats.csv"
  rm -rf "$OUT/$d"
done
bash tools/pmc_satu.sh "$OUT/pmc_satu" && cp "$OUT/pmc_satu/summary.csv" "$OUT/satu_pmc.csv"
PMC_TARGET="conv 128 64 3 --batch 6 --distinct --wy" bash tools/pmc_satu.sh "$OUT/pmc_conv_wy" && cp "$OUT/pmc_conv_wy/summary.csv" "$OUT/conv_wy_pmc.csv"
PMC_TARGET="conv 128 64 3 --batch 6 --distinct" bash tools/pmc_satu.sh "$OUT/pmc_conv" && cp "$OUT/pmc_conv/summary.csv" "$OUT/conv_pmc.csv"
python3 tools/make_satu_traffic.py "$OUT/satu_pmc.csv" --source "profiles/r06_satu_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/time_satu.py with the HR plan forced, tools/pmc_satu.sh; FETCH_SIZE x 2 per MI355X_MICROARCH.md; KiB)" > "$OUT/satu_traffic.json"
ls -la "$OUT"
