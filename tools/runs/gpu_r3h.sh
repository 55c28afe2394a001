#!/bin/bash
# round 3, GPU call H: split_out epilogue + two-term raw copy from the GroupNorm pass: kernel parity, model parity, step time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu \
   -k "tapgemm or groupnorm or plain_torch or t2v_full or i2vgen_full or tiny_vs_reference or blocks" -p no:cacheprovider 2>&1 | tail -12
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae --no-roofline --variants "fp16/high,fp16/fast" > gpurun_out/bench_h.json 2> gpurun_out/bench_h.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_h.json").read().strip().splitlines()[-1])
print("mixed", d["ms_per_step"], d["parity"]["unet_rel_l2"])
for k, v in d["variants"].items():
    print(k, v["ms_per_step"], v.get("unet_rel_l2"))
PY
tail -3 gpurun_out/bench_h.err
