#!/bin/bash
# r05 call B: first run of the W-panel-resident tap-GEMM shape (csrc/panelgemm.hip).  (1) its parity cases + the existing
# K = 320 cases that now dispatch to it, on the PRODUCT library; (2) tools/panel_probe.py: per-shape time panel vs streaming
# shapes in one process (tuning library, VGEN_TAPGEMM_PANEL flipped); (3) whole-step A/B of the t2v bench, tuning library,
# panel off / on, two interleaved rounds, in-run parity on the "on" arm.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "panel or lin_300x320 or dw_lin_bigM_qkv or dw_lin_geglu_res or plain_torch" 2>&1 | tail -15 | tee $O/pytest_panel.log
export VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_tuning.so
timeout 400 python tools/panel_probe.py $O/panel_probe.json 2>&1 | tee $O/panel_probe.log
A="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-scaling-model"
for r in 1 2; do
  for pm in 0 1; do
    par="--no-parity"; [ $pm = 1 ] && [ $r = 1 ] && par=""
    for prec in mixed fast; do
      VGEN_TAPGEMM_PANEL=$pm timeout 200 python bench.py $A $par --precision $prec 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'panel': $pm, 'round': $r, 'precision': '$prec', 'ms_per_step': d['ms_per_step'], 'parity': (d.get('parity') or {}).get('fixtures')}))" | tee -a $O/ab_panel.jsonl
    done
  done
done
