#!/bin/bash
# round 3, GPU call A: dual-W kernel parity first (fast fail), the whole -m gpu suite, the default bench line, rocprof stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== dual-W + tapgemm kernel tests"; 
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "tapgemm" -p no:cacheprovider 2>&1 | tail -25 | tee gpurun_out/pytest_tapgemm.log
echo "== full gpu suite"
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== bench default"
timeout 900 python bench.py --steps 20 --warmup 5 --dump-shapes > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 3000 gpurun_out/bench_default.json; tail -5 gpurun_out/bench_default.err
echo "== rocprof stats (high)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_high" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --variants '' --no-cpu-baseline --no-vae --no-roofline --no-parity > "$GRAFT_REPO_ROOT/gpurun_out/prof_high.log" 2>&1
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof_high -name "*kernel_stats*" | head; 
f=$(find gpurun_out/prof_high -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
# keep only the stats csv (the trace is large)
find gpurun_out/prof_high -type f ! -name "*stats*" -delete
