#!/bin/bash
# round 2, final validation: smoke(), the whole -m gpu suite, the default bench line
mkdir -p gpurun_out
{
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4
python bench.py 2>/dev/null | python tools/bench_brief.py
} > gpurun_out/r02_final.log 2>&1
cat gpurun_out/r02_final.log
