#!/bin/bash
# Round 5: fresh counters of the f32 wide Gram at config 5 (1e7 x 512 f32) -- the 256 x 256-tile bf16-split kernel that is the default
# (moments_wide_split256_kernel) and the 128 x 128 / tail kernels beside it.  Separate passes, --kernel-trace only.
#   bash tools/pmc_wide_r05.sh      -> gpurun_out/pmc_wide/r05_pmc_wide.json  (summarised by hand into profiles/r05_pmc_wide.json)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_wide; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export _WIDE_CHILD=1 PDS_WIDE_F32_NATIVE=0
k=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  k=$((k+1))
  rm -rf /tmp/pg; timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pg -o g -- python -u $ROOT/tools/wide_split_ab.py time > $OUT/run_$k.log 2>&1
  f=$(find /tmp/pg -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "no counters for: $set"; tail -5 $OUT/run_$k.log; continue; }
  head -1 $f > $OUT/pmc_$k.csv; grep "moments_wide" $f >> $OUT/pmc_$k.csv
done
rm -rf /tmp/pg; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o g -- python -u $ROOT/tools/wide_split_ab.py time > $OUT/run_stats.log 2>&1
cp $(find /tmp/pg -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python - <<'PY'
import csv, glob, collections, json, os
out=collections.defaultdict(dict)
root=os.environ.get("GRAFT_REPO_ROOT", ".")+"/gpurun_out/pmc_wide"
for f in sorted(glob.glob(root+"/pmc_*.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn=r["Kernel_Name"]
        if "moments_wide" in kn:
            acc[(kn.split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn,c),v in acc.items():
        out[kn][c]={"mean": sum(v)/len(v), "launches": len(v)}
st={}
for r in csv.DictReader(open(root+"/kernel_stats.csv")):
    if "moments_wide" in r["Name"]: st[r["Name"].split("(")[0][-60:]]={"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"])/1e3}
json.dump({"counters": out, "kernel_stats": st}, open(root+"/r05_pmc_wide.json","w"), indent=1)
print(json.dumps({"counters": out, "kernel_stats": st}, indent=1)[:1500])
PY
