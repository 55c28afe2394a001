#!/bin/bash
# r05 call M2: the in-launch split-K reduction WITHOUT cache-wide fences (call M: correct, but +13 % on the step — every
# wave's agent-scope release / acquire is a buffer_wbl2 + buffer_inv over its XCD's whole L2): partials stored / loaded with
# scope bits (write-through / read-through), a relaxed ticket.  Split-K parity + 12-round repeat test (product library,
# tuning library with the dual shape forced, the agent-scope-only variant), model-level determinism / parity, then the
# whole-step A/B: previous library (separate reducer launch) | system-scope accesses | agent-scope accesses.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05m2; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "splitk" > $O/pytest_splitk.log 2>&1; tail -2 $O/pytest_splitk.log
for plan in 1,128,2 1,160,2; do
  VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_tuning.so VGEN_TAPGEMM_PLAN=$plan timeout 200 python -m pytest tests/test_gpu_kernels.py -q -m gpu \
    -p no:cacheprovider -k "splitk_reduces" > $O/pytest_splitk_forced_$plan.log 2>&1; echo "forced $plan: $(tail -1 $O/pytest_splitk_forced_$plan.log)"
done
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_sc1.so timeout 200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider \
  -k "splitk" > $O/pytest_splitk_sc1.log 2>&1; echo "sc1 variant: $(tail -1 $O/pytest_splitk_sc1.log)"
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -p no:cacheprovider \
  -k "deterministic or unet_tiny_vs or graph_replay or t2v_full_size_mixed" > $O/pytest_model.log 2>&1; tail -2 $O/pytest_model.log
A="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-scaling-model --no-parity"
for r in 1 2; do
  for lib in libvgen_hip_prev.so libvgen_hip.so libvgen_hip_sc1.so; do
    VGEN_HIP_LIB=$PWD/vgen_amd/$lib timeout 200 python bench.py $A --precision mixed 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'lib': '$lib', 'round': $r, 'ms_per_step': d['ms_per_step']}))" | tee -a $O/ab_splitk.jsonl
  done
done
echo CALL_M2_DONE
