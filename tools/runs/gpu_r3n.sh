#!/bin/bash
# round 3, GPU call N: the default bench line and the rocprofv3 kernel stats of the final code (branch-free regather)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 230 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
B="python $R/bench.py --precision mixed --no-graph --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-parity --variants="
rm -rf /tmp/prof_kt
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- $B --steps 5 --warmup 2 > $O/prof_bench.json 2> /dev/null
python $R/tools/rocprof_summary.py $(ls /tmp/prof_kt/*/*kernel_stats.csv | head -1) $O/kernel_stats_summary.csv | head -12
