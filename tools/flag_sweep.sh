#!/bin/bash
# Scratch: build librsrt variants with different compiler flags HERE (no GPU needed), then time each on the
# GPU box:  tools/flag_sweep.sh build   (local)  /  tools/flag_sweep.sh run [kernel [KNOB values]] (inside gpurun)
# Variants live under build_exp/ (git-ignored, travels with gpurun).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/rsoderh-raytracing_amd/csrc/hip/rsrt_api.hip"
BASE="--offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fno-vectorize -I $ROOT/include"
mkdir -p "$ROOT/build_exp"
if [ "$1" = build ]; then
  i=0
  while IFS= read -r flags; do
    [ -z "$flags" ] && continue
    /opt/rocm/bin/hipcc $BASE $flags -o "$ROOT/build_exp/v$i.so" "$SRC" && echo "v$i: $flags" | tee -a "$ROOT/build_exp/index.txt"
    i=$((i+1))
  done
else
  K="${2:-2}"; KNOB="${3:-RSRT_DUMMY}"; VALS="${4:-0}"
  for so in "$ROOT"/build_exp/v*.so; do
    echo "== $(basename $so): $(grep "^$(basename $so .so):" "$ROOT/build_exp/index.txt")"
    RSRT_LIB="$so" python "$ROOT/tools/knob_sweep.py" "$KNOB" "$VALS" "$K" 64 | grep "round 2"
  done
fi
