#!/bin/bash
# r05 call R (the round's last GPU seconds): the fourth PMC pass of the final code that call N dropped for time —
# wave-cycle / stall counters of one eager step (rocprofv3 --pmc only, nothing else traced) -> profiles/r05r_pmc_wave_cycles.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --precision mixed --no-graph --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-parity --no-scaling-model --variants="
rm -rf /tmp/prof_w
timeout 55 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/prof_w -- $B --steps 1 --warmup 1 > /dev/null 2>&1
f=$(ls /tmp/prof_w/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python $R/tools/pmc_classes.py $O/pmc_wave_cycles.json WAVE=$f | head -30; else echo "no counter file"; fi
