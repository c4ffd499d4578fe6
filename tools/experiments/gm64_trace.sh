#!/bin/bash
# On the GPU box: kernel stats of the 33 .. 64-feature grouped route (100 000 groups x 100 rows x 64 f64; 10 000 x 1000 x 64; 100 000 x 100 x 48).
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gm64.py <<'PY'
import sys; sys.path.insert(0, sys.argv[1])
import torch, polars_ds_extension_amd as pds
dev = torch.device("cuda", 0); ctx = pds.Context(0); ctx.set_stream(torch.cuda.current_stream(dev))
g = torch.Generator(device=dev); g.manual_seed(1)
for G, R, p in ((100_000, 100, 64), (10_000, 1000, 64), (100_000, 100, 48)):
    N = G * R
    xs = [torch.randn(N, dtype=torch.float64, device=dev, generator=g) for _ in range(p)]
    y = torch.randn(N, dtype=torch.float64, device=dev, generator=g)
    off = torch.arange(0, N + 1, R, dtype=torch.int64, device=dev)
    for _ in range(3): co, nu = pds.lin_reg_by(*xs, target=y, group_offsets=off, ctx=ctx)
    torch.cuda.synchronize(); print(G, R, p, "nulls", int(nu.sum()))
PY
rm -rf /tmp/g64 && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/g64 -o t -- python /tmp/gm64.py $GRAFT_REPO_ROOT 2>&1 | grep -v "rocprofv3\|amdgpu" | tail -4
f=$(find /tmp/g64 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pds::" in r["Name"]]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms total  {int(r['Calls']):4d} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:120]}")
PY
