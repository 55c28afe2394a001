#!/bin/bash
# round-end pass on one box: the whole GPU suite, smoke(), then bench line + rocprofv3 stats + PMC passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/final
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > gpurun_out/final/pytest.log
tail -3 gpurun_out/final/pytest.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -3 ) > gpurun_out/final/smoke.log
tail -2 gpurun_out/final/smoke.log
bash tools/collect_evidence.sh 2>&1 | tail -45
