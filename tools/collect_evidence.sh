#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line + rocprofv3 kernel stats + HBM counters.
# Outputs under gpurun_out/evidence/; copy the summaries you want judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/evidence
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 2000 $O/bench.json
# per-kernel time of the same step (eager submission: same kernels, same durations as the graph replay)
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- \
    python $R/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-vae --no-roofline > $O/prof_bench.json 2> /dev/null
python $R/tools/rocprof_summary.py $(ls /tmp/prof_kt/*/*kernel_stats.csv | head -1) $O/kernel_stats_summary.csv | head -25
# HBM traffic counters, one pass each (FETCH_SIZE takes 3 of the 4 TCC slots)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/prof_$c -- \
      python $R/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-vae --no-roofline > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(ls /tmp/prof_$c/*/*counter_collection.csv | head -1) > $O/pmc_$c.txt
  grep -A2 "tapgemm_kernel" $O/pmc_$c.txt | head -40
done
