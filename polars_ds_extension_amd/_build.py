"""In-tree build of libpds_lstsq_hip.so (hipcc, --offload-arch=gfx950).  No GPU needed to compile."""
from __future__ import annotations

import os
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"


def build(jobs: int | None = None, force: bool = False) -> Path:
    jobs = jobs or min(8, os.cpu_count() or 1)
    if force:
        subprocess.check_call(["make", "-C", str(CSRC), "clean"])
    subprocess.check_call(["make", "-C", str(CSRC), f"-j{jobs}", "-s"])
    lib = CSRC / "libpds_lstsq_hip.so"
    if not lib.exists():
        raise RuntimeError("hipcc build did not produce libpds_lstsq_hip.so")
    return lib
