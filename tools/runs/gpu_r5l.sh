#!/bin/bash
# r05 call L: the FINAL code (K = 320 + K = 640 panels, stagger 64) — the whole -m gpu suite, smoke() (with its panel launch),
# the default bench line with per-shape timings.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05l; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --maxfail=8 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
cp gpurun_out/parity.json $O/parity.json 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -5 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 --dump-shapes > $O/bench.json 2> $O/bench.err
cp gpurun_out/tapgemm_shapes_t2v_fp16_mixed.json $O/ 2>/dev/null
tail -c 1200 $O/bench.json
