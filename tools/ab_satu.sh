#!/bin/bash
# SATU LR / HR launches of several library builds timed on ONE lease, interleaved rounds (tools/time_satu.py, HIP events, HR plan forced).
#   bash tools/ab_satu.sh <out.log> <rounds> lib1.so lib2.so ...      (paths relative to savsr_amd/csrc/)
set -u
OUT=$1; ROUNDS=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
if [ "${HR_PLAN:-forced}" = "forced" ]; then export SAVSR_HR_VARIANT=${SAVSR_HR_VARIANT:-1} SAVSR_HR_TILE=${SAVSR_HR_TILE:-20,2}; fi     # HR_PLAN=auto: the engine times its plans
: > "$OUT"
for r in $(seq 1 "$ROUNDS"); do
  for lib in "$@"; do
    echo "== round $r $lib" >> "$OUT"
    SAVSR_LIB_PATH=savsr_amd/csrc/$lib python3 tools/time_satu.py --iters ${ITERS:-100} --reps ${REPS:-2} ${TIME_SATU_ARGS:-} 2>&1 | grep -v "^tail_gather" >> "$OUT"
  done
done
