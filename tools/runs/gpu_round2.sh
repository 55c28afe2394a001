#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/pytest.log
tail -4 $O/pytest.log
cp gpurun_out/parity.json $O/parity.json
for v in 1 2; do
  VGEN_TEMPORAL_MINB=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae --no-roofline > $O/bench_minb$v.json 2>/dev/null
  python -c "import json;d=json.loads(open('$O/bench_minb$v.json').read().strip().splitlines()[-1]);print('MINB=$v', d['value'], d['ms_per_step'])"
done
( timeout 900 python tools/autotune_gemm.py --skip 24 --greedy 40 2>&1 | grep -E "greedy|A/B" | tail -50 ) > $O/autotune2.log
tail -45 $O/autotune2.log
cp gpurun_out/autotune.json $O/autotune2.json
echo ROUND2_DONE
