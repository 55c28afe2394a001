#!/bin/bash
# r04 call C: the fused d = 512 mid attention (kernel parity, fused vs GEMM path, VAE goldens with producer column
# statistics), VAE numbers, the two-stage pipeline at its real step counts, the other BASELINE shapes on the final code.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -p no:cacheprovider --maxfail=8 -k "attention_d512 or vae or groupnorm or kernels_vs_plain" > $O/pytest_vae.log 2>&1; tail -8 $O/pytest_vae.log
timeout 300 python bench.py --steps 10 --warmup 3 --variants= --no-cpu-baseline --no-roofline --dump-shapes > $O/bench_vae.json 2> $O/bench_vae.err; python - <<PY
import json
d=json.loads(open("$O/bench_vae.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["vae"], d["e2e"])
PY
cp gpurun_out/vae_shapes_*.json $O/ 2>/dev/null
timeout 600 python bench.py --config tft2v_sr600 --steps 50 > $O/bench_two_stage.json 2> $O/bench_two_stage.err; tail -c 1600 $O/bench_two_stage.json; tail -3 $O/bench_two_stage.err
# (i2vgen / sr600 lines: measured in the first run of this script, profiles/r04c_bench_{i2vgen,sr600}.json)
echo R4C_DONE
