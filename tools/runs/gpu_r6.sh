#!/bin/bash
# r06 GPU calls, one script, stages chosen by STAGES (space separated):
#   shapes   parity of the pp256 / q128 tap-GEMM shapes (forced through the plan table)
#   tune     tools/autotune_gemm.py: every (shape, BN, split-K) candidate incl. the r06 shapes, timed in-model -> autotune.json
#   calpar   the calibrated-mode parity tests on ALL full-width fixtures (VGEN_GPU_SLOW=1) -> parity_calibrated.json
#   evidence tools/collect_evidence.sh (bench line in the headline mode, rocprofv3 kernel stats + PMC passes of that mode)
#   ab       tools/ab_env.py: same-process A/B of a tuning-build switch on the whole step
#   suite    the whole -m gpu suite with durations
#   bench    the driver's bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG:-r6}
mkdir -p $O
cd $R
for st in ${STAGES:-shapes tune calpar evidence}; do
  echo "=== stage $st $(date +%T)"
  case $st in
    shapes)
      timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "r06 or splitk" -p no:cacheprovider 2>&1 | tail -15 > $O/pytest_shapes.log; tail -5 $O/pytest_shapes.log ;;
    tune)
      timeout 900 python tools/autotune_gemm.py --reps ${REPS:-3} > $O/autotune.log 2>&1; tail -40 $O/autotune.log
      cp gpurun_out/autotune.json $O/ 2>/dev/null ;;
    calpar)
      rm -f gpurun_out/parity_calibrated.json
      VGEN_GPU_SLOW=1 timeout 2400 python -m pytest tests/test_zz_calibrated_full.py tests/test_gpu_model.py -m gpu -q -k "calibrated" \
        --durations=12 -p no:cacheprovider > $O/pytest_calibrated.log 2>&1; tail -25 $O/pytest_calibrated.log
      cp gpurun_out/parity_calibrated.json $O/ 2>/dev/null ;;
    ab)   # AB_VAR / AB_VALUES: tools/ab_env.py (tuning build, one process, interleaved rounds)
      mkdir -p gpurun_out; rm -f gpurun_out/ab_env.jsonl
      timeout 1500 python tools/ab_env.py "$AB_VAR" "$AB_VALUES" --rounds ${AB_ROUNDS:-3} ${AB_EXTRA:-} > $O/ab_env.log 2>&1; tail -30 $O/ab_env.log
      cp gpurun_out/ab_env.jsonl gpurun_out/ab_env_shapes.json $O/ 2>/dev/null ;;
    evidence)
      SKIP_SUITE=1 SKIP_SMOKE=1 ${EVIDENCE_ENV:-} bash tools/collect_evidence.sh 2>&1 | tail -60 ;;
    suite)
      timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=25 > $O/pytest_gpu.log 2>&1; tail -40 $O/pytest_gpu.log
      cp gpurun_out/parity.json gpurun_out/parity_calibrated.json $O/ 2>/dev/null ;;
    bench)
      timeout 900 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -3 $O/bench.err ;;
  esac
done
echo R6_DONE
