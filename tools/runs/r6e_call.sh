#!/bin/bash
# r06 call E: the other BASELINE shapes in the headline (calibrated) mode, one bench.py run each, own timeout each
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r6e; mkdir -p $O
A="--no-cpu-baseline --no-roofline --variants="
for c in i2vgen sr600 tft2v896 tft2v32f; do
  echo "=== $c $(date +%T)"
  timeout 700 python bench.py --config $c --steps 10 --warmup 2 $A --no-vae > $O/bench_$c.json 2> $O/bench_$c.err
  echo "rc $?"; tail -c 700 $O/bench_$c.json; echo; tail -2 $O/bench_$c.err
done
echo "=== videolcm $(date +%T)"
timeout 700 python bench.py --config videolcm --steps 32 --warmup 8 $A > $O/bench_videolcm.json 2> $O/bench_videolcm.err
echo "rc $?"; tail -c 900 $O/bench_videolcm.json; echo; tail -2 $O/bench_videolcm.err
echo "=== two-stage $(date +%T)"
timeout 1500 python bench.py --config tft2v_sr600 --steps 50 $A > $O/bench_two_stage.json 2> $O/bench_two_stage.err
echo "rc $?"; tail -c 1200 $O/bench_two_stage.json; echo; tail -3 $O/bench_two_stage.err
echo R6E_DONE
