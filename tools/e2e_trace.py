"""pl_lr_by on the headline host frame with PDS_TRACE=1: where the wall clock of the plugin call goes (stage marks serialise the
pipeline -- a diagnostic, not a benchmark)."""
import os
import sys
import time
from pathlib import Path

os.environ["PDS_TRACE"] = "1"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402
import pyarrow as pa  # noqa: E402
import torch  # noqa: E402

import plugin_harness as ph  # noqa: E402
from polars_ds_extension_amd import _lib  # noqa: E402

lib = _lib.load()
G, R, P = 1_000_000, 100, 16
N = G * R
rng = np.random.default_rng(0)
host = [("y", pa.array(rng.normal(size=N)))] + [(f"x{j + 1}", pa.array(rng.normal(size=N))) for j in range(P)]
key = ("key", pa.array(np.repeat(np.arange(G, dtype=np.int64), R)))
kw = {"bias": False, "null_policy": "raise", "l1_reg": 0.0, "l2_reg": 0.0, "solver": "qr", "tol": 1e-5, "max_iter": 200,
      "weighted": False, "positive": False, "singular_x_tol": 1e-12}
for i in range(3):
    t0 = time.perf_counter()
    _, res = ph.call_plugin(lib, sys.argv[1] if len(sys.argv) > 1 else "pl_lr_by", [key] + host, kw)
    print(f"call {i}: {1e3 * (time.perf_counter() - t0):.1f} ms wall, {len(res)} rows out", file=sys.stderr, flush=True)
