# PMC passes (separate runs, --kernel-trace only) on the rolling kernel at C4: instruction mix and wait fractions
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pg; timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pg -o g -- python -u $GRAFT_REPO_ROOT/tools/rolling_bench.py c4 > /tmp/pg.log 2>&1
  f=$(find /tmp/pg -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "rolling_seg_kernel<double, 8, 0" in r["Kernel_Name"]:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
done
