#!/bin/bash
# Builds libsavsr_hip.so for gfx950 in-tree (the .so is git-ignored but travels with gpurun).
#
# An object is rebuilt when its source, a shared header or the compile flags changed: the flags (incl. EXTRA_FLAGS,
# which the experiment scripts use for -D switches) are hashed into <obj>.flags next to each object, so an object
# left behind by an experiment build is never linked into the product library.  A failed compile removes its object
# and fails the script (every background job is waited for by PID).
#   OUT=<name>.so    library to link (default libsavsr_hip.so; experiment builds use their own name)
#   OBJDIR=<dir>     where objects go (default: this directory)
#   EXTRA_ONLY="a.hip b.hip"   apply EXTRA_FLAGS to these sources only (the others keep the base flags and their objects)
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
BASE_FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function"
# SAVSR_DIAG=1: the INSTRUMENTED library (DIAG kernel instantiations + the savsr_debug_* entry points of the header's
# SAVSR_DIAG section) as libsavsr_hip_diag.so with its own object directory; tools load it through SAVSR_LIB_PATH.
if [ "${SAVSR_DIAG:-0}" = "1" ]; then
  EXTRA_FLAGS="${EXTRA_FLAGS:-} -DSAVSR_DIAG"
  OUT=${OUT:-libsavsr_hip_diag.so}
  OBJDIR=${OBJDIR:-diag_obj}
fi
OUT=${OUT:-libsavsr_hip.so}
OBJDIR=${OBJDIR:-.}
mkdir -p "$OBJDIR"
# savsr_source_hash(): sha256 over the kernel sources + headers + flags this library is built from (first 16 hex digits), compiled into
# api.cpp, so that a measurement file (profiles/satu_traffic.json) can name the build it was taken on and bench.py can tell a stale one
SRC_HASH=$( (cat conv_mfma.hip conv_wy.hip osconv.hip elementwise.hip satu.hip tail.hip metrics.hip resize.hip api.cpp common.hpp conv_common.hpp ../../include/savsr_hip.h; printf '%s' "$BASE_FLAGS ${EXTRA_FLAGS:-} ${EXTRA_ONLY:-}") | sha256sum | cut -c1-16)
# the SATU + tail kernels alone (not the header: it changes with every other kernel's interface) -- with the EXTRA_FLAGS that reach them: an
# experiment / DIAG build that changes these kernels through -D switches must not report the product's hash (profiles/satu_traffic.json and
# savsr_amd/hr_plans.json are attached to a library by this stamp)
SATU_EXTRA=""
if [ -n "${EXTRA_FLAGS:-}" ] && { [ -z "${EXTRA_ONLY:-}" ] || [[ " ${EXTRA_ONLY} " == *" satu.hip "* ]] || [[ " ${EXTRA_ONLY} " == *" tail.hip "* ]]; }; then SATU_EXTRA=" ${EXTRA_FLAGS}"; fi
SATU_HASH=$( (cat satu.hip tail.hip common.hpp; printf '%s' "$BASE_FLAGS -fno-slp-vectorize$SATU_EXTRA") | sha256sum | cut -c1-16)

# per-file flags: satu.hip and conv_wy.hip keep their scalar fp32 arithmetic scalar -- packed fp32 instructions (v_pk_mul / v_pk_fma) are an
# anti-lever beside MFMAs on gfx950 (MI355X_MICROARCH.md); the explicit 2-vector code of the HR stage is unaffected
file_flags() { case "$1" in satu.hip|conv_wy.hip) echo "-fno-slp-vectorize";; api.cpp) echo "-DSAVSR_SOURCE_HASH=\"$SRC_HASH\" -DSAVSR_SATU_HASH=\"$SATU_HASH\"";; *) echo "";; esac; }
flags_for() {  # flags_for <src>
  local f="$BASE_FLAGS $(file_flags "$1")"
  if [ -z "${EXTRA_ONLY:-}" ] || [[ " ${EXTRA_ONLY} " == *" $1 "* ]]; then echo "$f ${EXTRA_FLAGS:-}"; else echo "$f"; fi
}
sig_for() { printf '%s' "$HIPCC $(flags_for "$1")" | sha1sum | cut -d' ' -f1; }

stale() {   # stale <src> <obj>
  local src=$1 obj=$2
  [ ! -f "$obj" ] || [ ! -f "$obj.flags" ] || [ "$(cat "$obj.flags")" != "$(sig_for "$src")" ] ||
    [ "$src" -nt "$obj" ] || [ common.hpp -nt "$obj" ] || [ conv_common.hpp -nt "$obj" ] || [ ../../include/savsr_hip.h -nt "$obj" ]
}

compile() { # compile <src> <obj> [extra hipcc args]
  local src=$1 obj=$2
  shift 2
  rm -f "$obj" "$obj.flags"
  if $HIPCC $(flags_for "$src") "$@" -c "$src" -o "$obj.tmp"; then
    mv "$obj.tmp" "$obj"
    printf '%s' "$(sig_for "$src")" > "$obj.flags"
  else
    rm -f "$obj.tmp"
    return 1
  fi
}

OBJS=()
PIDS=()
for f in conv_mfma.hip conv_wy.hip osconv.hip elementwise.hip satu.hip tail.hip metrics.hip resize.hip; do
  o="$OBJDIR/${f%.hip}.o"
  if stale "$f" "$o"; then
    compile "$f" "$o" &
    PIDS+=($!)
  fi
  OBJS+=("$o")
done
if stale api.cpp "$OBJDIR/api.o"; then
  compile api.cpp "$OBJDIR/api.o" -x hip &
  PIDS+=($!)
fi
rc=0
for p in "${PIDS[@]:-}"; do
  [ -z "$p" ] && continue
  wait "$p" || rc=1
done
if [ $rc -ne 0 ]; then
  echo "build.sh: a compile failed; $OUT not linked" >&2
  exit 1
fi
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${OBJS[@]}" "$OBJDIR/api.o"
echo "built $(pwd)/$OUT"
