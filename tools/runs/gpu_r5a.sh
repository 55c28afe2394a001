#!/bin/bash
# r05 call A (prepared at the end of r04, whose GPU budget was spent before the committed default could be timed):
# the three mixed rules in ONE process order on ONE box — "mixed:e0d0:all" (r03: level 0, every kind), the r04 GPU-measured
# rule ("mixed:e0d0:noextra": level 0 minus ff1 / ff2 / q2), and the committed default (that
# plus conv2 / proj_in / proj_out at level 1) — the default with its in-run parity over the three t2v fixtures; then the
# full-width fixture tests under the default (emulated: DESIGN §4.1; test_full_width_* and the vcomposer test).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05a; mkdir -p $O
A="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-scaling-model"
timeout 100 python bench.py $A --no-parity --precision mixed:e0d0:all > $O/bench_rule_r03.json 2> /dev/null
timeout 100 python bench.py $A --no-parity --precision mixed:e0d0:noextra > $O/bench_rule_r04l.json 2> /dev/null
timeout 120 python bench.py $A --precision mixed > $O/bench_rule_default.json 2> /dev/null
python - <<PY
import json
for f in ("bench_rule_r03", "bench_rule_r04l", "bench_rule_default"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % f).read().splitlines() if l.startswith('{"metric"')][-1])
        print(f, d["value"], d["ms_per_step"], (d.get("parity") or {}).get("fixtures"))
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "full_width or vcomposer or t2v_full" 2>&1 | tail -5 | tee $O/pytest_full_width.log
# the composer trunk's cheaper rules (emulated 1.027e-3 / 9.01e-4 / 8.58e-4): error and forward time on the GPU
timeout 600 python - <<PY 2>&1 | tee $O/vcomposer_rules.log
import sys, time, torch
sys.path.insert(0, "tests")
import full_cases as fc
g = fc.load("vcomposer")
for pr in ("mixed", "mixed:e01d01:all", "high"):
    m = fc.build("vcomposer", g, pr, "cuda")
    out = fc.forward("vcomposer", m, g, "cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        fc.forward("vcomposer", m, g, "cuda")
    torch.cuda.synchronize()
    print(pr, "rel-L2 %.4e" % fc.error(out, g)[0], "eager forward %.1f ms" % ((time.perf_counter() - t0) / 3 * 1e3), flush=True)
    del m, out
    torch.cuda.empty_cache()
PY
