#!/bin/bash
# round 3, GPU call I: environment-switch A/Bs (GroupNorm path selection, column-statistics threshold), mixed-level variants,
# the other BASELINE shapes on the final code
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; : > gpurun_out/ab_env.jsonl
ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-parity"
run() {  # tag, env assignments...
  tag=$1; shift
  env "$@" timeout 300 python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'tag': '$tag', 'ms_per_step': d['ms_per_step']}))" | tee -a gpurun_out/ab_env.jsonl
}
for r in 1 2; do
  run base X=1
  run gn_fused24 VGEN_GN_FUSED_MAX_MB=24
  run gn_fused0 VGEN_GN_FUSED_MAX_MB=0
  run gn_regs0 VGEN_GN_REGS=0
  run cs3584 VGEN_COLSTATS_MIN_ROWS=3584
  run noshare VGEN_SHARED_PREFIX=0
done
echo "== mixed level variants (same process)"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae --no-roofline --variants "fp16/mixed:d0t0,fp16/mixed:e0d0t1,fp16/fast" 2>/dev/null | tail -1 > gpurun_out/bench_levels.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_levels.json").read())
print("mixed", d["ms_per_step"], d["parity"]["unet_rel_l2"])
for k, v in d["variants"].items(): print(k, v["ms_per_step"], v.get("unet_rel_l2"))
PY
echo "== other BASELINE shapes, final code"
: > gpurun_out/configs_r03_final.jsonl
for spec in "i2vgen mixed" "i2vgen fast" "sr600 mixed" "tft2v896 mixed" "tft2v32f mixed" "videolcm mixed"; do
  set -- $spec
  timeout 600 python bench.py --config $1 --precision $2 --steps 5 --warmup 1 --no-cpu-baseline --no-vae --no-roofline 2>/dev/null | tail -1 >> gpurun_out/configs_r03_final.jsonl
  tail -1 gpurun_out/configs_r03_final.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['name'], d['config']['precision'], d['ms_per_step'], 'ms', d['value'], d['unit'], d.get('inversion'))"
done
