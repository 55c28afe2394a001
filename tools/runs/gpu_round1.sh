#!/bin/bash
# GPU round 1 of r02: parity tests, public-API bench (fp16 + bf16), other configs, kernel trace.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --dump-shapes > $O/bench_fp16.json 2> $O/bench_fp16.err
tail -c 1500 $O/bench_fp16.json
timeout 200 python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline --no-vae > $O/bench_bf16.json 2> $O/bench_bf16.err
tail -c 600 $O/bench_bf16.json
timeout 200 python bench.py --steps 10 --warmup 3 --partition --no-cpu-baseline --no-vae --no-roofline > $O/bench_partition.json 2> $O/bench_partition.err
tail -c 600 $O/bench_partition.json
for c in videolcm tft2v32f tft2v896 i2vgen sr600; do
  timeout 400 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-vae --dump-shapes > $O/bench_$c.json 2> $O/bench_$c.err
  tail -c 700 $O/bench_$c.json; tail -3 $O/bench_$c.err
done
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --vae-size 720x1280 > $O/bench_vae720.json 2> $O/bench_vae720.err
tail -c 400 $O/bench_vae720.json
cd /tmp
rm -rf /tmp/prof_kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- \
    python $R/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-vae --no-roofline > $O/prof_bench.json 2> /dev/null
python $R/tools/rocprof_summary.py $(ls /tmp/prof_kt/*/*kernel_stats.csv | head -1) $O/kernel_stats_summary.csv | head -30
cp $R/gpurun_out/parity.json $O/parity.json 2>/dev/null
cp $R/gpurun_out/*shapes*.json $O/ 2>/dev/null
echo ROUND1_DONE
