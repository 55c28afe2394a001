#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: the whole -m gpu suite, the default bench line, rocprofv3 kernel stats
# of the same step (eager launches, csv) and the PMC counters in SEPARATE --pmc-only passes (never combined with tracing),
# the multi-GPU step path on one device (eager gather / update vs the captured step with the RCCL all-gather inside).
# Outputs under gpurun_out/evidence/; tools/evidence_to_profiles.py <tag> turns them into the profiles/<tag>_* files.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/evidence
mkdir -p $O
cd $R
# SKIP_SUITE=1 / SKIP_PARTITION=1: a short evidence call (bench line, kernel stats, counters) when the GPU budget does not
# hold the 11-minute suite again — the driver runs it at round end either way
if [ -z "${SKIP_SUITE:-}" ]; then
  timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --maxfail=8 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
  cp gpurun_out/parity.json $O/parity.json 2>/dev/null
fi
if [ -z "${SKIP_SMOKE:-}" ]; then timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -4 $O/smoke.log; fi
cd /tmp && export TMPDIR=/tmp
PREC=${PREC:-calibrated}
# the profiled passes load the calibrated weights the bench run below saves (one pack-time pass for all of them)
CALF=/tmp/vgen_evidence_t2v.cal
rm -f $CALF
B="python $R/bench.py --precision $PREC --calibration-file $CALF --no-graph --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-parity --no-scaling-model --variants="
timeout 900 python $R/bench.py --steps 20 --warmup 5 --dump-shapes --precision $PREC --calibration-file $CALF > $O/bench.json 2> $O/bench.err
cp $R/gpurun_out/tapgemm_shapes_t2v_fp16_$PREC.json $R/gpurun_out/other_shapes_t2v_fp16_$PREC.json $O/ 2>/dev/null
tail -c 1500 $O/bench.json
if [ -z "${SKIP_PARTITION:-}" ]; then
P="--steps 20 --warmup 5 --precision $PREC --calibration-file $CALF --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-parity --no-scaling-model --variants= --partition"
timeout 200 python $R/bench.py $P > $O/bench_partition_eager.json 2> $O/bench_partition_eager.err
VGEN_FORCE_COLLECTIVE=1 timeout 200 python $R/bench.py $P > $O/bench_partition_eager_rccl.json 2> $O/bench_partition_eager_rccl.err
VGEN_FORCE_COLLECTIVE=1 timeout 200 python $R/bench.py $P --graph-collective > $O/bench_partition_graph_rccl.json 2> $O/bench_partition_graph_rccl.err
python - <<PY
import json
for f in ("bench_partition_eager", "bench_partition_eager_rccl", "bench_partition_graph_rccl"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % f).read().splitlines() if l.startswith('{"metric"')][-1]); print(f, d["value"], d["ms_per_step"], d["config"]["hipgraph"])
    except Exception as e:
        print(f, "FAILED", e)
PY
# the N > 1 launch line of bench.py itself, two ranks sharing the one device over gloo (functional: RCCL refuses two ranks per GPU)
VGEN_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  $R/bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-roofline --no-e2e > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks_one_device.err
tail -c 600 $O/bench_2ranks_one_device.json; tail -2 $O/bench_2ranks_one_device.err
fi
rm -rf /tmp/prof_kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- $B --steps 5 --warmup 2 > $O/prof_bench.json 2> /dev/null
python $R/tools/rocprof_summary.py $(ls /tmp/prof_kt/*/*kernel_stats.csv | head -1) $O/kernel_stats_summary.csv | head -24
SPECS=""
# PMC_PASSES=3 drops the wave-cycle pass (the traffic and MFMA-busy figures the bench line quotes need the first three)
PASSES=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY")
for c in "${PASSES[@]:0:${PMC_PASSES:-4}}"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/prof_$tag
  timeout 400 rocprofv3 --pmc $c --output-format csv -d /tmp/prof_$tag -- $B --steps 1 --warmup 1 > /dev/null 2>&1
  f=$(ls /tmp/prof_$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then SPECS="$SPECS $tag=$f"; else echo "no counter file for $c"; fi
  PREC=$PREC python $R/tools/pmc_classes.py $O/pmc_classes.json $SPECS > /dev/null    # after every pass: a call cut short keeps what it has
done
PREC=$PREC python $R/tools/pmc_classes.py $O/pmc_classes.json $SPECS | head -40
echo EVIDENCE_DONE
