"""Builds profiles/<round>_* from what tools/profile_round.sh left in gpurun_out/prof/ (run in the repo, no GPU)."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out" / "prof"
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
OUT = ROOT / "profiles"
OUT.mkdir(exist_ok=True)


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def stats(path, only_pds=True):
    rows = []
    for r in csv.DictReader(open(path)):
        n = short(r["Name"])
        if only_pds and not n.startswith("pds::"):
            continue
        rows.append((n, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return rows


def pmc(path):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        n = short(r["Kernel_Name"])
        if n.startswith("pds::"):
            acc[n].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


line = json.loads((SRC / "bench_line.json").read_text().strip().splitlines()[-1])
(OUT / f"{rnd}_bench_line.json").write_text(json.dumps(line, indent=1) + "\n")
(OUT / f"{rnd}_bench_kernel_stats.csv").write_text((SRC / "bench_kernel_stats.csv").read_text())
(OUT / f"{rnd}_extra_kernel_stats.csv").write_text((SRC / "extra_kernel_stats.csv").read_text())
extra = json.loads((SRC / "bench_extra.json").read_text())
(OUT / f"{rnd}_bench_extra.json").write_text(json.dumps(extra, indent=1) + "\n")

fetch, write = pmc(SRC / "pmc_FETCH_SIZE.csv"), pmc(SRC / "pmc_WRITE_SIZE.csv")
for extra_pmc in ("roll",):  # rolling kernel at C4 (tools/rolling_bench.py c4), same two counters
    fp, wp = SRC / f"pmc_{extra_pmc}_FETCH_SIZE.csv", SRC / f"pmc_{extra_pmc}_WRITE_SIZE.csv"
    if fp.exists() and wp.exists():
        for k, v in pmc(fp).items():
            fetch.setdefault(k, v)
        for k, v in pmc(wp).items():
            write.setdefault(k, v)


def steady(path, warm_launches):
    """Per-kernel duration statistics from the kernel TRACE with the first `warm_launches` launches of every kernel dropped
    (the --stats table averages the warm-ups in: round 1's CSV and bench line disagreed by 7 % for that reason)."""
    per = defaultdict(list)
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        n = short(r["Kernel_Name"])
        if n.startswith("pds::"):
            per[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for n, v in per.items():
        w = v[warm_launches:] if len(v) > warm_launches + 2 else v
        out[n] = {"launches": len(w), "avg_us": sum(w) / len(w), "min_us": min(w), "max_us": max(w), "dropped_warmup": len(v) - len(w)}
    return out
traffic = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `python bench.py --steps 2 "
                   "--warmup 1 --no-cpu`; read bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction, MI355X_MICROARCH.md HBM section), "
                   "write bytes = WRITE_SIZE x 1024", "kernels": {}}
for k in fetch:
    rd = 2.0 * fetch[k][0] * 1024
    wr = write.get(k, (0.0, 0))[0] * 1024
    traffic["kernels"][k] = {"fetch_size_KiB": fetch[k][0], "write_size_KiB": write.get(k, (0.0, 0))[0], "read_bytes_corrected": rd,
                             "write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "launches_sampled": fetch[k][1]}
(OUT / f"{rnd}_traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")

md = [f"# profiles/{rnd} -- one MI355X (gfx950), ROCm 7.2", "",
      f"* `{rnd}_bench_line.json` -- the `python bench.py` JSON line.",
      f"* `{rnd}_bench_kernel_stats.csv` -- `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu --no-extras`.",
      f"* `{rnd}_traffic.json` -- HBM bytes per launch from `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (two separate passes).",
      f"* `{rnd}_bench_extra.json` / `{rnd}_extra_kernel_stats.csv` -- `python tools/bench_extra.py`: the other BASELINE configs (single OLS + "
      "report, rolling, recursive, elastic net) and the host-buffer rate, and the rocprofv3 kernel stats of that run.",
      "* produced by `tools/profile_round.sh` (GPU box) + `tools/summarize_profiles.py`.", "",
      "## Headline step (bench.py): kernel time, rocprofv3 --stats", "", "| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
for n, c, us, pct in stats(SRC / "bench_kernel_stats.csv"):
    md.append(f"| `{n}` | {c} | {us:.1f} | {pct:.2f} |")
rf = line["roofline"]
trace = SRC / "bench_kernel_trace.csv"
if trace.exists():
    st = steady(trace, 3)
    (OUT / f"{rnd}_bench_kernel_steady.json").write_text(json.dumps(st, indent=1) + "\n")
    md += ["", "Same run, from the kernel trace, warm-up launches (the first 3 of each kernel) dropped -- the figure to hold "
           f"against the bench line (`{rnd}_bench_kernel_steady.json`):", "", "| kernel | launches | avg us | min | max |", "|---|---|---|---|---|"]
    for n, v in sorted(st.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"]):
        md.append(f"| `{n}` | {v['launches']} | {v['avg_us']:.1f} | {v['min_us']:.1f} | {v['max_us']:.1f} |")
md += ["", f"HIP-event timing inside bench.py for the same kernels: grouped fused {rf['avg_launch_ms']:.4f} ms per launch "
       f"({rf['achieved']:.0f} GB/s algorithmic, frac {rf['frac']:.3f} of 8 TB/s); single-OLS Gram "
       f"{line['gram_build']['avg_launch_ms']:.4f} ms ({line['gram_build']['achieved_GBps']:.0f} GB/s, frac {line['gram_build']['frac_of_hbm_peak']:.3f}).", "",
       "## HBM traffic (PMC, per launch)", "", "| kernel | read MB (2 x FETCH_SIZE KiB x 1024) | write MB |", "|---|---|---|"]
for k, v in traffic["kernels"].items():
    md.append(f"| `{k}` | {v['read_bytes_corrected'] / 1e6:.1f} | {v['write_bytes'] / 1e6:.1f} |")
md += ["", f"Algorithmic bytes per launch of the fused grouped kernel: {rf['algorithmic_bytes_per_launch'] / 1e6:.1f} MB "
       "(1e8 rows x 17 f64 columns + offsets + coefficients).", "",
       "## Other configs (tools/bench_extra.py): kernel time, rocprofv3 --stats", "", "| kernel | calls | avg us |", "|---|---|---|"]
for n, c, us, pct in stats(SRC / "extra_kernel_stats.csv"):
    md.append(f"| `{n}` | {c} | {us:.1f} |")
md += ["", "```json", json.dumps({k: v for k, v in extra.items()}, indent=1), "```", ""]
# ---- round 3: shuffled keys (partition route) + grouped pred, mid-width reports
for tag, title, cmd in (("keyed", "Keys in any row order (C3 frame, shuffled) + grouped pred", "python tools/ab_quick.py keyed pred"),
                        ("wide", "Mid-width single regressions and reports (2e7 x 20 / 32 / 64 f64)", "python tools/wide_report_prof.py")):
    f = SRC / f"{tag}_kernel_stats.csv"
    if not f.exists():
        continue
    (OUT / f"{rnd}_{tag}_kernel_stats.csv").write_text(f.read_text())
    md += [f"## {title}: `rocprofv3 --kernel-trace --stats -- {cmd}`", "", "| kernel | calls | avg us |", "|---|---|---|"]
    for n, c, us, pct in stats(f):
        if us * c > 200.0:
            md.append(f"| `{n}` | {c} | {us:.1f} |")
    log = SRC / f"{tag}_run.log"
    if log.exists():
        keep = [l for l in log.read_text().splitlines() if l.startswith(("keyed", "n=", "ordered")) or " ms" in l and "rocprof" not in l and "amdgpu" not in l]
        md += ["", "```", *keep[:40], "```", ""]
gm = SRC / "grouped_mid.log"
if gm.exists():
    lines = [l for l in gm.read_text().splitlines() if "groups x" in l]
    (OUT / f"{rnd}_grouped_mid_width.txt").write_text("# python tools/grouped_mid_width.py: grouped OLS with 17 .. 64 features, wall ms per call (Gram records + solves)\n" + "\n".join(lines) + "\n")
    md += ["## Grouped fits with 17 .. 64 features (`tools/grouped_mid_width.py`)", "", "```", *lines, "```", ""]
kf, kw = SRC / "pmc_keyed_FETCH_SIZE.csv", SRC / "pmc_keyed_WRITE_SIZE.csv"
if kf.exists() and kw.exists():
    f2, w2 = pmc(kf), pmc(kw)
    md += ["HBM traffic of the shuffled-keys route (PMC, per launch):", "", "| kernel | read MB | write MB |", "|---|---|---|"]
    kt = {}
    for k in f2:
        rd, wr = 2.0 * f2[k][0] * 1024, w2.get(k, (0.0, 0))[0] * 1024
        if rd + wr > 5e7:
            md.append(f"| `{k}` | {rd / 1e6:.1f} | {wr / 1e6:.1f} |")
            kt[k] = {"read_bytes_corrected": rd, "write_bytes": wr, "launches_sampled": f2[k][1]}
    (OUT / f"{rnd}_keyed_traffic.json").write_text(json.dumps(kt, indent=1) + "\n")
    md.append("")
# ---- round 4: traffic of the mid-width kernels / grouped stream / grouped pred against their algorithmic bytes, the width cliff, SQ counters
mid = {}
for tag in ("wide", "gmid", "pred"):
    fp, wp = SRC / f"pmc_{tag}_FETCH_SIZE.csv", SRC / f"pmc_{tag}_WRITE_SIZE.csv"
    if fp.exists() and wp.exists():
        f2, w2 = pmc(fp), pmc(wp)
        for k in f2:
            rd, wr = 2.0 * f2[k][0] * 1024, w2.get(k, (0.0, 0))[0] * 1024
            if rd + wr > 2e8:
                mid[f"{tag}: {k}"] = {"read_bytes_corrected": rd, "write_bytes": wr, "launches_averaged": f2[k][1]}
if mid:
    (OUT / f"{rnd}_mid_traffic.json").write_text(json.dumps({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, averaged over ALL launches of a kernel in "
                                                              "tools/wide_report_prof.py (2e7 rows x 64 / 32 / 20 f64: 10.4 / 5.28 / 3.36 GB per stream), tools/grouped_mid_width.py, "
                                                              "tools/ab_quick.py pred (C3 frame: 1e8 x 9 f64 in, pred + resid out); read = 2 x FETCH_SIZE KiB x 1024", "kernels": mid}, indent=1) + "\n")
    md += ["## Round-4 counters: mid-width kernels, grouped stream, grouped pred (PMC averages over the launches of each tool run)", "",
           "| run: kernel | read MB | write MB | launches |", "|---|---|---|---|"]
    for k, v in mid.items():
        md.append(f"| `{k}` | {v['read_bytes_corrected'] / 1e6:.1f} | {v['write_bytes'] / 1e6:.1f} | {v['launches_averaged']} |")
    md.append("")
wc = SRC / "width_cliff.log"
if wc.exists():
    lines = [l for l in wc.read_text().splitlines() if l.startswith(("#", "1e6"))]
    md += ["## The 16 -> 17 feature cliff (`tools/grouped_width_cliff.py`; second block: `PDS_GROUPED_MID_FUSED=0`)", "", "```", *lines, "```", ""]
sq = {}
for f in sorted(SRC.glob("pmc_sq_*.csv")):
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        for tagk, key in (("grouped_stream_kernel<double, 16", "grouped_stream_kernel<double,16> (headline)"), ("moments_small_kernel<double, 0", "moments_small_kernel<double,0> (Gram, control)")):
            if tagk in r["Kernel_Name"]:
                sq.setdefault(key, defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
if sq:
    table = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in sq.items()}
    (OUT / f"{rnd}_pmc_grouped.json").write_text(json.dumps(table, indent=1) + "\n")
    names = sorted({c for d in table.values() for c in d})
    md += ["## SQ counters per launch: the fused grouped kernel against the Gram kernel (`bench.py --steps 3 --warmup 1 --no-cpu --no-extras`, one counter set per pass)", "",
           "| counter | " + " | ".join(table) + " |", "|---|" + "---|" * len(table)]
    for c in names:
        md.append(f"| {c} | " + " | ".join(f"{table[k].get(c, float('nan')):.4g}" for k in table) + " |")
    md.append("")
# ---- round 6: the HC2 / HC3 report route at C2, kernel resources of the product build, the wide Gram's counters
hs = SRC / "hc_kernel_stats.csv"
if hs.exists():
    (OUT / f"{rnd}_hc_kernel_stats.csv").write_text(hs.read_text())
    hf, hw = SRC / "pmc_hc_FETCH_SIZE.csv", SRC / "pmc_hc_WRITE_SIZE.csv"
    f2, w2 = (pmc(hf), pmc(hw)) if hf.exists() and hw.exists() else ({}, {})
    md += ["## HC2 / HC3 reports at C2 (`rocprofv3 --kernel-trace --stats -- python tools/report_hc_quick.py`: 1e8 rows x 16 / 12 / 9 / 8 f64 + intercept, "
           "every std_err type; HBM bytes per launch from separate `--pmc FETCH_SIZE` / `WRITE_SIZE` passes, averaged over the launches of a kernel)", "",
           "| kernel | calls | avg us | read MB | write MB |", "|---|---|---|---|---|"]
    for n, c, us, pct in stats(hs):
        if us * c < 500.0:
            continue
        rd = 2.0 * f2[n][0] * 1024 / 1e6 if n in f2 else float("nan")
        wr = w2.get(n, (0.0, 0))[0] * 1024 / 1e6 if n in w2 else float("nan")
        md.append(f"| `{n}` | {c} | {us:.1f} | {rd:.1f} | {wr:.1f} |")
    log = SRC / "report_hc_run.log"
    if log.exists():
        md += ["", "```", *[l for l in log.read_text().splitlines() if l.startswith("p =")], "```", ""]
pw = SRC / "pmc_wide.json"
if pw.exists():
    (OUT / f"{rnd}_pmc_wide_raw.json").write_text(pw.read_text())
import subprocess
kr = subprocess.run([sys.executable, str(ROOT / "tools" / "kernel_resources.py")], capture_output=True, text=True).stdout
if kr.strip():
    (OUT / f"{rnd}_kernel_resources.txt").write_text("# python tools/kernel_resources.py: registers / spills / scratch of every kernel of the product build (code-object notes)\n" + kr)
    rows = [l for l in kr.splitlines()[1:] if l.split() and (int(l.split()[4]) > 0 or int(l.split()[6]) > 0)]
    md += [f"## Kernel resources (`{rnd}_kernel_resources.txt`): kernels with spilled VGPRs or scratch: {len(rows)}", "", "```", kr.splitlines()[0], *rows, "```", ""]
    sg = sorted((l for l in kr.splitlines()[1:] if l.split()), key=lambda l: -int(l.split()[5]))[:12]
    md += ["Largest spilled-SGPR counts (v_writelane / v_readlane into spare VGPR lanes, no memory traffic):", "", "```", kr.splitlines()[0], *sg, "```", ""]
(OUT / f"{rnd}_summary.md").write_text("\n".join(md))
print("wrote", sorted(p.name for p in OUT.glob(f"{rnd}_*")))
