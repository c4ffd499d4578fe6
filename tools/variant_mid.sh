#!/bin/bash
# quick A/B builds that only differ in moments_mid.hip: bash tools/variant_mid.sh NAME "EXTRA flags" -> tools/variants/NAME.bin
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; CSRC="$ROOT/polars_ds_extension_amd/csrc"; mkdir -p "$ROOT/tools/variants"
touch "$CSRC/moments_mid.hip" "$CSRC/grouped_mid.hip"; make -C "$CSRC" -j8 EXTRA="$2" 2>&1 | grep -E "error|Error" || true
cp "$CSRC/libpds_lstsq_hip.so" "$ROOT/tools/variants/$1.bin"
