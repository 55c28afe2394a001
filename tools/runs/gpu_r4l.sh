#!/bin/bash
# r04 call L (the round's last GPU minutes): the mixed rule without the FeedForward / cross-q two-term weights (new default)
# against the r03 rule ("mixed:e0d0:all"), same box, the new one with its in-run parity over the three t2v fixtures.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04l; mkdir -p $O
A="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-scaling-model"
timeout 100 python bench.py $A --no-parity --precision mixed:e0d0:all > $O/bench_rule_r03.json 2> /dev/null
timeout 120 python bench.py $A --precision mixed > $O/bench_rule_r04.json 2> /dev/null
python - <<PY
import json
for f in ("bench_rule_r03", "bench_rule_r04"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % f).read().splitlines() if l.startswith('{"metric"')][-1])
        print(f, d["value"], d["ms_per_step"], (d.get("parity") or {}).get("fixtures"))
    except Exception as e:
        print(f, "FAILED", e)
PY
