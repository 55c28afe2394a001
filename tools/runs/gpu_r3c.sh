#!/bin/bash
# round 3, GPU call C: new kernels / fixtures under test, mixed-precision candidates timed + parity-checked in one process
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== tests: gauss (fused solver step), attention, full-size VAE fixtures, RCCL paths"
timeout 1200 python -m pytest tests/test_gauss.py tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_parallel.py -q -m gpu \
   -k "gauss or attention or plain_torch or vae_full_size or rccl or i2vgen_full or world2_on_one_gpu" -p no:cacheprovider 2>&1 | tail -12
echo "== bench: headline high + mixed candidates + fast, same process"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae --no-roofline \
   --variants "fp16/mixed:e0d01,fp16/mixed:e01d01,fp16/mixed:e0d0t1,fp16/fast" > gpurun_out/bench_mixed.json 2> gpurun_out/bench_mixed.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_mixed.json").read().strip().splitlines()[-1])
print("high", d["ms_per_step"], d["parity"]["unet_rel_l2"])
for k, v in d["variants"].items():
    print(k, v["ms_per_step"], v.get("unet_rel_l2"))
PY
tail -3 gpurun_out/bench_mixed.err
