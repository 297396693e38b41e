#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4rec; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -n 3 $O/tests.log
cp gpurun_out/parity_margins.json gpurun_out/narrow_parity.json $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
