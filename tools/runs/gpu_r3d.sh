#!/bin/bash
# round 3, GPU call D: mid-round evidence (default bench line, rocprofv3 kernel stats, PMC passes) + the other BASELINE shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== mixed-mode full-size test"
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "mixed_precision" -p no:cacheprovider 2>&1 | tail -3
bash tools/collect_evidence_r03.sh 2>&1 | tail -70
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== other BASELINE shapes"
: > gpurun_out/configs_r03.jsonl
for spec in "i2vgen mixed" "i2vgen high" "i2vgen fast" "sr600 mixed" "tft2v896 mixed" "tft2v32f mixed" "videolcm mixed"; do
  set -- $spec
  timeout 600 python bench.py --config $1 --precision $2 --steps 5 --warmup 1 --no-cpu-baseline --no-vae --no-roofline 2>/dev/null | tail -1 >> gpurun_out/configs_r03.jsonl
  tail -1 gpurun_out/configs_r03.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['name'], d['config']['precision'], d['ms_per_step'], 'ms', d['value'], d['unit'], d.get('inversion'))"
done
