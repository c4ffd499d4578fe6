"""
A CPU rehearsal of `bench.py --gpus N` for N > 1 (the driver launches it on an 8-GPU node at round end; no such node exists here or on
the gpurun box): PDS_BENCH_DRYRUN=1 swaps RCCL for gloo, device tensors for CPU tensors and the kernels for injected numpy compute --
the control flow (plan construction and piece-count agreement, peer sends / grouped receives, barriers, the MAX all-reduce of the
elapsed time, the scatter leg, the one JSON line from rank 0) runs exactly as written.  Launched the way the driver launches it.
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,scaling", [(8, "strong"), (2, "weak")])
def test_bench_multi_rank_control_flow(world, scaling):
    env = dict(os.environ, PDS_BENCH_DRYRUN="1", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--groups", "4003", "--rows-per-group", "12", "--feats", "4", "--scaling", scaling]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=540, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == scaling
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "DRY RUN" in d["data"]
    total = 4003 if scaling == "strong" else 4003 * world
    assert d["config"]["groups_total"] == total
    if scaling == "strong":
        assert d["config"]["gather_chunks"] >= 1 and d["config"]["gather"] == "p2p"
        assert d["scatter"] and "error" not in d["scatter"], d["scatter"]  # the scatter leg ran on all ranks
