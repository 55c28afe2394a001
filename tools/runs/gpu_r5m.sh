#!/bin/bash
# r05 call M: split-K partials reduced inside the launch (last block of a tile to arrive; no splitk_reduce_kernel).
#   1. every tap-GEMM parity case + the 12-round interleaved split-K repeat test (product library: the planner's own splits,
#      pp128 and pp shapes), then the repeat test again on the tuning library with the dual shape forced (BN 128 / 160, split 2)
#   2. model level: determinism, the tiny golden, session graph replay, the full-size mixed parity test
#   3. smoke()
#   4. whole-step A/B, previous library (HEAD~1 sources: separate reducer launch) vs this one, two interleaved rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "tapgemm or splitk" > $O/pytest_tapgemm.log 2>&1; tail -4 $O/pytest_tapgemm.log
for plan in 1,128,2 1,160,2; do
  VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_tuning.so VGEN_TAPGEMM_PLAN=$plan timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu \
    -p no:cacheprovider -k "splitk_reduces" > $O/pytest_splitk_forced_$plan.log 2>&1; echo "forced $plan: $(tail -1 $O/pytest_splitk_forced_$plan.log)"
done
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -p no:cacheprovider \
  -k "deterministic or unet_tiny_vs or graph_replay or t2v_full_size_mixed" > $O/pytest_model.log 2>&1; tail -4 $O/pytest_model.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
A="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-scaling-model --no-parity"
for r in 1 2; do
  for lib in libvgen_hip_prev.so libvgen_hip.so; do
    VGEN_HIP_LIB=$PWD/vgen_amd/$lib timeout 200 python bench.py $A --precision mixed 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'lib': '$lib', 'round': $r, 'ms_per_step': d['ms_per_step']}))" | tee -a $O/ab_splitk.jsonl
  done
done
echo CALL_M_DONE
