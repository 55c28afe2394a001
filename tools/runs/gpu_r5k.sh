#!/bin/bash
# r05 call K: the panel kernel's fp32 residual tile by quad-contiguous loads + ds_bpermute (PANEL_RES_QUAD): parity of the
# residual cases, per-shape A/B (flags 0 / 1), whole step A/B; smoke() with its new panel launch.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "panel" 2>&1 | tail -5 | tee $O/pytest_panel.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $O/smoke.log
export VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_tuning.so
timeout 300 python tools/panel_probe.py $O/panel_flags.json --panel-only --flags-scan --only=o-proj --only="L1 o-proj" 2>&1 | tee $O/panel_flags.log
A="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-scaling-model --no-parity"
for r in 1 2; do
  for fl in 0 1; do
    VGEN_PANEL_FLAGS=$fl timeout 200 python bench.py $A --precision mixed 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'panel_flags': $fl, 'round': $r, 'ms_per_step': d['ms_per_step']}))" | tee -a $O/ab_flags.jsonl
  done
done
