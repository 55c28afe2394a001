#!/bin/bash
# r05 call N (last of the round, ~8 GPU-minutes left): the FINAL code — the K = 320 + K = 640 panels, the two-launch split-K
# (the in-launch reductions of calls M / M2 were removed again).  The split-K parity cases + the 12-round repeat test on the
# library that ships, then tools/collect_evidence.sh without the 11-minute suite and the partition benches (both measured
# earlier this round on code that differs from this only in the K = 640 panels: profiles/r05_pytest_gpu.log,
# r05_bench_partition_*.json): default bench line with per-shape timings, rocprofv3 kernel stats, three PMC passes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05n
timeout 200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "splitk" > gpurun_out/r05n/pytest_splitk.log 2>&1; tail -2 gpurun_out/r05n/pytest_splitk.log
SKIP_SUITE=1 SKIP_PARTITION=1 SKIP_SMOKE=1 PMC_PASSES=3 bash tools/collect_evidence.sh
