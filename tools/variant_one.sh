#!/bin/bash
# A/B builds that differ in ONE source file: bash tools/variant_one.sh NAME file.hip "EXTRA flags" -> tools/variants/NAME.bin
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; CSRC="$ROOT/polars_ds_extension_amd/csrc"; mkdir -p "$ROOT/tools/variants"
touch "$CSRC/$2"; make -C "$CSRC" -j8 EXTRA="$3" 2>&1 | grep -E "error|Error" || true
cp "$CSRC/libpds_lstsq_hip.so" "$ROOT/tools/variants/$1.bin"
touch "$CSRC/$2"
