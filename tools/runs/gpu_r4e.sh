#!/bin/bash
# r04 call E: why did the forced-collective partition runs of the evidence script print nothing?  (stack dump after 70 s)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04e; mkdir -p $O
P="--steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-parity --no-scaling-model --variants= --partition"
VGEN_BENCH_WATCHDOG=70 VGEN_FORCE_COLLECTIVE=1 timeout 150 python bench.py $P > $O/eager_rccl.json 2> $O/eager_rccl.err; echo rc=$?; tail -c 400 $O/eager_rccl.json; grep -v "amdgpu.ids\|hostname" $O/eager_rccl.err | tail -40
VGEN_BENCH_WATCHDOG=70 VGEN_FORCE_COLLECTIVE=1 timeout 150 python bench.py $P --graph-collective > $O/graph_rccl.json 2> $O/graph_rccl.err; echo rc=$?; tail -c 400 $O/graph_rccl.json; grep -v "amdgpu.ids\|hostname" $O/graph_rccl.err | tail -40
