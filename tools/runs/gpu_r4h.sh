#!/bin/bash
# r04 call H: the bench line of the final code exactly as the driver runs it (JSON must be the LAST stdout line), N = 1 and the
# N = 2 launch line on one device (gloo), and the forced-RCCL partition run whose banner used to trail the JSON.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04h; mkdir -p $O
timeout 400 python bench.py > $O/bench_default.out 2> $O/bench_default.err; echo "rc=$? last line starts with: $(tail -n 1 $O/bench_default.out | cut -c1-60)"
VGEN_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --backend gloo --steps 4 --warmup 1 --no-cpu-baseline --no-vae --no-roofline --no-e2e > $O/bench_2ranks.out 2> $O/bench_2ranks.err; echo "rc=$? last line starts with: $(tail -n 1 $O/bench_2ranks.out | cut -c1-60)"
VGEN_FORCE_COLLECTIVE=1 timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-parity --no-scaling-model --variants= --partition --graph-collective > $O/bench_graph_rccl.out 2> $O/bench_graph_rccl.err; echo "rc=$? last line starts with: $(tail -n 1 $O/bench_graph_rccl.out | cut -c1-60)"; wc -l $O/*.out
