#!/bin/bash
# r04 call G: for the record on the final code — rocprofv3 kernel stats of an i2vgen-xl CFG step (where flash attention is a
# quarter of the step) and of the VAE decode, the tft2v step shapes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_i2v
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_i2v -- python $R/bench.py --config i2vgen --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-roofline > $O/prof_i2vgen_bench.json 2> /dev/null
python $R/tools/rocprof_summary.py $(ls /tmp/prof_i2v/*/*kernel_stats.csv | head -1) $O/kernel_stats_i2vgen.csv | head -14
cd $R
for c in tft2v896 tft2v32f; do
  timeout 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-roofline > $O/bench_$c.json 2> $O/bench_$c.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_$c.json").read().splitlines() if l.startswith('{"metric"')][-1]); print("$c", d["value"], d["ms_per_step"])
except Exception as e: print("$c FAILED", e)
PY
done
echo R4G_DONE
