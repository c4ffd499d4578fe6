#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export PDS_PROBE_ONLY=8
echo "== default"; python tools/grouped_second_pass_cost.py 2>&1 | grep "p=8"
echo "== PDS_DBG_SECOND=1 (sus_tol set, no host work behind the kernel)"; PDS_DBG_SECOND=1 python tools/grouped_second_pass_cost.py 2>&1 | grep "p=8"
echo "== PDS_DBG_SECOND=2 (choleskey too runs collect + count readback)"; PDS_DBG_SECOND=2 python tools/grouped_second_pass_cost.py 2>&1 | grep "p=8"
echo "== PDS_DBG_SUS=1e300 (sus on, threshold unreachable)"; PDS_DBG_SUS=1e300 python tools/grouped_second_pass_cost.py 2>&1 | grep "p=8"
echo "== PDS_DBG_SUS=0 (sus off for both, second pass host code runs for qr)"; PDS_DBG_SUS=0 python tools/grouped_second_pass_cost.py 2>&1 | grep "p=8"
