"""gpurun_out/pmc_gmid/sq_<k>.csv (tools/pmc_grouped_mid.sh) -> profiles/<round>_pmc_grouped_mid.json: per kernel, every counter averaged over its
launches, and the derived fractions (GRBM_GUI_ACTIVE is summed over the 8 XCCs: launch clocks = GRBM / 8; matrix_pipe_busy =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch clocks)).   python tools/summarize_pmc_gmid.py r06"""
import csv, json, sys
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
acc = defaultdict(lambda: defaultdict(list))
for f in sorted((ROOT / "gpurun_out" / "pmc_gmid").glob("sq_*.csv")):
    per = defaultdict(lambda: defaultdict(float))  # (kernel, dispatch) -> counter -> sum over XCC / SE rows
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pds::" not in k or ("grouped_stream_kernel" not in k and "grouped_mid_stream_kernel" not in k): continue
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), cs in per.items():
        for c, v in cs.items(): acc[k][c].append(v)
out = {"note": "tools/pmc_grouped_mid.sh: rocprofv3 --kernel-trace --pmc, one counter set per run, averaged over the launches of a kernel; 1e6 groups x 100 rows at "
               "16 / 17 / 24 / 32 f64 features, device-resident offsets.  launch clocks = GRBM_GUI_ACTIVE / 8; matrix_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x launch clocks); "
               "issue / wait fractions are of SQ_WAVE_CYCLES.", "kernels": {}}
for k, cs in acc.items():
    name = k.replace("void pds::(anonymous namespace)::", "").split("(")[0]
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    der = {}
    if "GRBM_GUI_ACTIVE" in d:
        der["launch_clocks"] = round(d["GRBM_GUI_ACTIVE"] / 8)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d: der["matrix_pipe_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE"] / 8), 3)
    if "SQ_WAVE_CYCLES" in d:
        for c, n in (("SQ_ACTIVE_INST_ANY", "issue_frac"), ("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac")):
            if c in d: der[n] = round(d[c] / d["SQ_WAVE_CYCLES"], 3)
    if "SQ_INSTS_VALU_MFMA_F64" in d and d["SQ_INSTS_VALU_MFMA_F64"]: der["mfma_busy_clk_per_inst"] = round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / d["SQ_INSTS_VALU_MFMA_F64"], 1)
    d["derived"] = der
    out["kernels"][name] = d
p = ROOT / "profiles" / f"{rnd}_pmc_grouped_mid.json"
p.write_text(json.dumps(out, indent=1))
for k, d in out["kernels"].items(): print(k[:70], d["derived"], "VALU", int(d.get("SQ_INSTS_VALU", 0)), "SALU", int(d.get("SQ_INSTS_SALU", 0)), "LDS", int(d.get("SQ_INSTS_LDS", 0)), "VMEM", int(d.get("SQ_INSTS_VMEM_RD", 0)), "MFMA", int(d.get("SQ_INSTS_VALU_MFMA_F64", 0)))
