#!/bin/bash
# r05 call G: head start of waves 0-3 over their SIMD partners in the panel kernel (VGEN_PANEL_STAGGER x 64 cycles), per
# shape, then the whole step at the two best settings.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05g; mkdir -p $O
export VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_tuning.so
timeout 400 python tools/panel_probe.py $O/panel_stagger.json --panel-only --stagger-scan 2>&1 | tee $O/panel_stagger.log
A="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-scaling-model --no-parity"
for sg in 0 48 96 0 48 96; do
  VGEN_PANEL_STAGGER=$sg timeout 200 python bench.py $A --precision mixed 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'stagger': $sg, 'ms_per_step': d['ms_per_step']}))" | tee -a $O/ab_stagger.jsonl
done
