"""stdin: bench.py's JSON line -> one short line (A/B runs)."""
import json
import sys

for ln in sys.stdin:
    ln = ln.strip()
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    r, g, p8 = d["roofline"], d.get("gram_build") or {}, d.get("grouped_p8") or {}
    print(f"step {d['ms_per_step']:.3f} ms | fused p16 {r['avg_launch_ms']:.3f} ms frac {r['frac']:.3f} | gram {g.get('avg_launch_ms')} ms "
          f"frac {g.get('frac_of_hbm_peak')} | p8 {p8.get('ms_per_step')} ms frac {p8.get('frac_of_hbm_peak')}")
