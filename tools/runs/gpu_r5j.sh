#!/bin/bash
# r05 call J: K = 640 launches (the 16 x 28 level's q/k/v, out-projections, cross-q) on the panel shape (80-column single-pass
# panels): parity cases, per-shape probe (panel vs streaming), whole step with VGEN_PANEL_K640 = 0 / 1 (tuning library).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "k640" 2>&1 | tail -5 | tee $O/pytest_k640.log
export VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_tuning.so
timeout 300 python tools/panel_probe.py $O/panel_probe_k640.json --only=L1 2>&1 | tee $O/panel_probe_k640.log
A="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-scaling-model --no-parity"
for r in 1 2; do
  for k6 in 0 1; do
    VGEN_PANEL_K640=$k6 timeout 200 python bench.py $A --precision mixed 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'panel_k640': $k6, 'round': $r, 'ms_per_step': d['ms_per_step']}))" | tee -a $O/ab_k640.jsonl
  done
done
