# PMC passes (separate runs, --kernel-trace only) on the f32 wide Gram kernel at config 5: instruction mix and wait fractions
# usage: bash tools/pmc_wide.sh [0|1]   (PDS_WIDE_F32_NATIVE)
cd /tmp && export TMPDIR=/tmp
export _WIDE_CHILD=1 PDS_WIDE_F32_NATIVE=${1:-0}
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf /tmp/pg; timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pg -o g -- python -u $GRAFT_REPO_ROOT/tools/wide_split_ab.py time > /tmp/pg.log 2>&1
  f=$(find /tmp/pg -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "no counters for: $set"; tail -5 /tmp/pg.log; continue; }
  python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "moments_wide_kernel<float, 2" in r["Kernel_Name"]:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
done
