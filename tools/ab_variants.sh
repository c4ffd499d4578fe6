#!/bin/bash
# A/B of several builds of the library on ONE GPU box in ONE gpurun call (boxes of the pool differ by +-4 %, two builds on one
# box agree to 0.2 %: never compare numbers across calls).
#
# Here (CPU, cross-compiles):   bash tools/ab_variants.sh build  base=  ldlt=-DPDS_ROLL_LDLT  wpe3="-DPDS_ROLL_WPE=3"
#     -> tools/variants/<name>.bin, one library per NAME=EXTRA-flags pair (EXTRA is the Makefile's knob variable)
# On the box (inside gpurun):   bash tools/ab_variants.sh run "python tools/rolling_only.py" [rounds]
#     -> runs the command once per variant and round with that variant installed, restores the default library afterwards
# Both:                         gpurun --timeout 600 -- 'bash tools/ab_variants.sh run "python tools/rolling_only.py" 2'
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/polars_ds_extension_amd/csrc"
LIB="$CSRC/libpds_lstsq_hip.so"
VAR="$ROOT/tools/variants"
case "${1:-}" in
  build)
    shift
    mkdir -p "$VAR"
    rm -f "$VAR"/*.bin
    cp "$LIB" /tmp/pds_default_lib.so
    for pair in "$@"; do
      name="${pair%%=*}"; flags="${pair#*=}"
      echo "== $name: EXTRA='$flags'"
      find "$CSRC" -maxdepth 1 \( -name '*.hip' -o -name '*.cpp' \) -exec touch {} +
      make -C "$CSRC" -j16 EXTRA="$flags" 2>&1 | grep -E "error|spill|scratch" || true
      cp "$LIB" "$VAR/$name.bin"
    done
    # back to the default build (objects are stale on purpose: force them)
    find "$CSRC" -maxdepth 1 \( -name '*.hip' -o -name '*.cpp' \) -exec touch {} +
    make -C "$CSRC" -j16 2>&1 | grep -E "error" || true
    ls -la "$VAR"
    ;;
  run)
    cmd="${2:?command}"; rounds="${3:-2}"
    cp "$LIB" /tmp/pds_default_lib.so
    trap 'cp /tmp/pds_default_lib.so "$LIB"' EXIT
    for r in $(seq 1 "$rounds"); do
      for v in "$VAR"/*.bin; do
        cp "$v" "$LIB"
        echo "=== round $r  variant $(basename "$v" .bin)"
        timeout -k 5 300 bash -c "$cmd" 2>&1 | grep -v "amdgpu.ids" || echo "(command failed or timed out)"
      done
    done
    ;;
  *) sed -n 2,12p "$0"; exit 2 ;;
esac
