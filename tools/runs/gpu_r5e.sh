#!/bin/bash
# r05 call E: the panel shape with its A operand staged per wave through LDS-DMA (instead of direct MFMA-layout loads).
# parity cases on the product library; per-shape probe + stamps (tuning library); whole-step A/B panel off / on.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "panel or lin_300x320 or dw_lin_bigM_qkv or dw_lin_geglu_res" 2>&1 | tail -8 | tee $O/pytest_panel.log
export VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_tuning.so
timeout 400 python tools/panel_probe.py $O/panel_probe.json --stamps 2>&1 | tee $O/panel_probe.log
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
