#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the paired grouped stream (17 / 24 / 32 features) beside the 16-feature fused kernel, one counter set per run
# (rocprofv3 --kernel-trace --pmc only).  Output: gpurun_out/pmc_gmid/sq_<k>.csv -> python tools/summarize_pmc_gmid.py
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_gmid; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gmid_run.py <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
import torch
import polars_ds_extension_amd as pds
dev = torch.device("cuda", 0)
ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream())
G, R = 1_000_000, 100
N = G * R
g = torch.Generator(device=dev); g.manual_seed(1)
xs = [torch.randn(N, dtype=torch.float64, device=dev, generator=g) for _ in range(32)]
y = torch.randn(N, dtype=torch.float64, device=dev, generator=g)
off = torch.arange(0, N + 1, R, dtype=torch.int64, device=dev)
for p in (16, 17, 24, 32):
    for _ in range(2): pds.lin_reg_by(*xs[:p], target=y, group_offsets=off, add_bias=False, ctx=ctx)
torch.cuda.synchronize()
PY
k=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  k=$((k+1))
  rm -rf /tmp/pg && timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pg -o s -- python -u /tmp/gmid_run.py $ROOT > $OUT/run_$k.log 2>&1
  f=$(find /tmp/pg -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/sq_$k.csv
done
ls -la $OUT
