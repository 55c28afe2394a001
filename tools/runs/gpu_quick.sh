#!/bin/bash
# quick GPU iteration: kernel + model parity subset, then the in-model per-shape timing and the step rate
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-q}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests -m gpu -x -q -k "${2:-tapgemm or groupnorm or layernorm or attention or unet_tiny or full_size or session or deterministic}" 2>&1 | tail -8 ) > $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --dump-shapes --no-cpu-baseline --no-vae > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("steps/s", d["value"], "ms/step", d["ms_per_step"], "tapgemm ms", d["roofline"]["tapgemm_ms_per_step"], "frac", d["roofline"]["frac"], d.get("hbm_kernels"))
PY
cp $R/gpurun_out/tapgemm_shapes_t2v.json $R/gpurun_out/other_shapes_t2v.json $O/ 2>/dev/null
echo QUICK_DONE
