#!/bin/bash
# round 3, GPU call G: the whole -m gpu suite, smoke(), then the evidence set of the final code
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 ) | tee gpurun_out/pytest_gpu_final.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -4 ) | tee gpurun_out/smoke_final.log
bash tools/collect_evidence_r03.sh 2>&1 | grep -v "^ *\"" | tail -40
