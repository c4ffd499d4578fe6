"""Register / spill / scratch / LDS figures of every kernel in the built objects (csrc/build/*.o), from the code objects' metadata
notes (`llvm-readelf --notes` of the gfx950 bundle entry).   python tools/kernel_resources.py [substring ...]   -> a table on stdout
(profiles/r06_kernel_resources.txt is this tool's output for the product build)."""
import re, subprocess, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "polars_ds_extension_amd" / "csrc" / "build"
LLVM = Path("/opt/rocm/lib/llvm/bin")
want = sys.argv[1:]
rows = []
for obj in sorted(BUILD.glob("*.o")):
    with tempfile.TemporaryDirectory() as td:
        out, fat = Path(td) / "dev.co", Path(td) / "fat.bin"
        r = subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(obj)], capture_output=True, text=True)
        if r.returncode != 0 or not fat.exists():
            continue  # (host-only object)
        r = subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--output={out}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
        if r.returncode != 0 or not out.exists() or out.stat().st_size == 0:
            continue
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(out)], capture_output=True, text=True).stdout
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", notes, re.S):
        body = m.group(2)
        g = lambda k: int(re.search(k + r":\s+(\d+)", body).group(1)) if re.search(k + r":\s+(\d+)", body) else -1
        rows.append((obj.stem, m.group(1), g(r"\.vgpr_count"), g(r"\.agpr_count"), g(r"\.sgpr_count"), g(r"\.vgpr_spill_count"),
                     g(r"\.sgpr_spill_count"), g(r"\.private_segment_fixed_size"), g(r"\.group_segment_fixed_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[1] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'object':16s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>7s}  kernel")
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n.replace("void ", ""))
    n = n.split("(")[0]
    if want and not any(w in n for w in want):
        continue
    print(f"{r[0]:16s} {r[2]:5d} {r[3]:5d} {r[4]:5d} {r[5]:6d} {r[6]:6d} {r[7]:7d} {r[8]:7d}  {n}")
