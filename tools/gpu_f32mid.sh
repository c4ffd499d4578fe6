#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_f32_contract.py tests/test_reference_suite.py tests/test_baseline_sizes.py -m gpu -q -x -k "f32 or moments or wide or report or glm or GLM" -p no:cacheprovider 2>&1 | tail -8
python - <<'PY'
import time, torch, sys
sys.path.insert(0, ".")
import polars_ds_extension_amd as pds
pds.config.LIN_REG_EXPR_F64 = False
dev = torch.device("cuda", 0)
import os
for n, p in ((20_000_000, 32), (20_000_000, 64), (20_000_000, 20)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    xs = [torch.randn(n, dtype=torch.float32, device=dev, generator=g) for _ in range(p)]
    y = torch.randn(n, dtype=torch.float32, device=dev, generator=g)
    for env in ("1", "0"):
        os.environ["PDS_MID_GRAM"] = env
        pds.gram_moments(*xs, target=y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): pds.gram_moments(*xs, target=y)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
        gb = n * (p + 1) * 4 / 1e9
        print(f"f32 Gram {n:.0e} x {p}: {'stream' if env == '1' else 'tiled '} {t * 1e3:.2f} ms  ({gb / t / 1e3:.2f} TB/s)", flush=True)
    del xs, y
PY
