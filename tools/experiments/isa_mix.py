"""Instruction mix per basic block of a gfx950 assembly listing (hipcc -S --cuda-device-only): the largest blocks of each kernel
with their VALU / f64 arithmetic / DPP / plain-move counts, plus VGPR and spill counts.   python tools/experiments/isa_mix.py file.s"""
import re
import sys

s = open(sys.argv[1]).read()
meta = re.findall(r"\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", s, flags=re.S)
for name, vg, sp in meta:
    print(f"{name}: {vg} VGPRs, {sp} spills")
for k in re.split(r"\n(?=_Z\w+:\s)", s)[1:]:
    name = k.split(":", 1)[0]
    end = k.find("s_endpgm")
    if end < 0:
        continue
    rows = []
    parts = re.split(r"\n(\.LBB\d+_\d+):", k[:end])
    for i in range(1, len(parts), 2):
        ins = [ln.strip().split()[0] for ln in parts[i + 1].split("\n") if ln.strip() and not ln.strip().startswith((".", ";", "//"))]
        rows.append((parts[i], len(ins), sum(x.startswith("v_") for x in ins), sum(bool(re.match(r"v_(fma|mul|add|fmac)_f64", x)) for x in ins),
                     sum("dpp" in x for x in ins), sum((x.startswith("v_mov") and "dpp" not in x) or x.startswith("v_accvgpr") for x in ins),
                     sum(x.startswith(("global_load", "global_store", "ds_", "scratch_")) for x in ins)))
    print(name)
    for r in sorted(rows, key=lambda t: -t[1])[:6]:
        print("   %s: %d instr, %d VALU (%d f64 arith, %d dpp, %d plain mov), %d memory" % r)
