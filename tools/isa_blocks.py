"""Per-basic-block instruction mix of one kernel in a `hipcc -save-temps` .s file: where the SGPR spill traffic (v_writelane / v_readlane),
the matrix instructions and the memory instructions sit.    python tools/isa_blocks.py file.s 'kernel name substring (demangled)' [min_instrs]"""
import re, subprocess, sys
from collections import Counter

path, want = sys.argv[1], sys.argv[2]
min_i = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().splitlines()
# kernel bodies: "<mangled>:" ... ".Lfunc_end"
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s*(;.*)?$", l)]
names = subprocess.run(["c++filt"], input="\n".join(lines[i].split(":")[0] for i in starts), capture_output=True, text=True).stdout.splitlines()
sel = [s for s, n in zip(starts, names) if want in n]
if not sel:
    sys.exit("no kernel matches; have:\n" + "\n".join(n[:160] for n in names))
s = sel[0]
e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
print(names[starts.index(s)][:200])
blocks, cur, name = [], [], "entry"
for l in lines[s + 1:e]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1)
        continue
    t = l.strip()
    if t and not t.startswith((";", ".", "//")):
        cur.append(t)
blocks.append((name, cur))
keys = ("v_readlane", "v_writelane", "v_mfma", "global_load", "ds_read", "ds_write", "s_load", "global_store", "global_atomic", "s_waitcnt", "s_cbranch", "s_branch", "v_readfirstlane", "s_sleep")
tot = Counter()
print(f"{'block':>12} {'n':>5} " + " ".join(f"{k[-9:]:>9}" for k in keys) + "  branches-to")
for name, ins in blocks:
    c = Counter()
    for t in ins:
        op = t.split()[0]
        for k in keys:
            if op.startswith(k):
                c[k] += 1
    tot.update(c); tot["n"] += len(ins)
    if len(ins) >= min_i or c["v_mfma"]:
        tg = [t.split()[-1] for t in ins if t.startswith(("s_cbranch", "s_branch"))]
        print(f"{name:>12} {len(ins):5d} " + " ".join(f"{c[k]:9d}" for k in keys) + "  " + ",".join(tg))
print(f"{'TOTAL':>12} {tot['n']:5d} " + " ".join(f"{tot[k]:9d}" for k in keys))
