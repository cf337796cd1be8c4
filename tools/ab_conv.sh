#!/bin/bash
# Build conv-kernel variants HERE or on the GPU box and A/B them in one process:  AB_ONLY=conv_wy.hip bash tools/ab_conv.sh "-DWY_VALU=4" "-DWY_STAMPS=1"
# Each argument is one variant's EXTRA_FLAGS for conv_mfma.hip (own object dir savsr_amd/csrc/exp_vN, own libsavsr_hip_exp_vN.so; the
# product library is variant 0 and is never touched).  AB_BUILD_ONLY=1: build only (the .so files travel to the GPU box with gpurun).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
LIBS="savsr_amd/csrc/libsavsr_hip.so"
i=0
for fl in "$@"; do
  i=$((i+1))
  if [ ! -f "savsr_amd/csrc/libsavsr_hip_exp_v$i.so" ] || [ "${AB_REBUILD:-1}" = "1" ]; then
    EXTRA_FLAGS="$fl" EXTRA_ONLY="${AB_ONLY:-conv_mfma.hip}" OBJDIR=exp_v$i OUT=libsavsr_hip_exp_v$i.so bash savsr_amd/csrc/build.sh >/dev/null || exit 1
  fi
  echo "variant v$i: $fl"
  LIBS="$LIBS savsr_amd/csrc/libsavsr_hip_exp_v$i.so"
done
[ "${AB_BUILD_ONLY:-0}" = "1" ] && exit 0
python3 tools/ab_conv.py --libs $LIBS ${AB_ARGS:-}
