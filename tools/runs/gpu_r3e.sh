#!/bin/bash
# round 3, GPU call E: two-term activations + glue kernels under test; mixed (new) vs high vs mixed:e0d01 vs fast in one process
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_session.py tests/test_gauss.py -q -m gpu \
   -k "cast_split or glue or t2v_full or i2vgen_full or session or pinned or gauss or tiny_vs_reference" -p no:cacheprovider 2>&1 | tail -15
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae --no-roofline \
   --variants "fp16/high,fp16/mixed:e0d01,fp16/mixed:e0d0t1,fp16/fast" > gpurun_out/bench_mixed2.json 2> gpurun_out/bench_mixed2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_mixed2.json").read().strip().splitlines()[-1])
print("mixed", d["ms_per_step"], d["parity"]["unet_rel_l2"])
for k, v in d["variants"].items():
    print(k, v["ms_per_step"], v.get("unet_rel_l2"))
PY
tail -3 gpurun_out/bench_mixed2.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/parity.json"))
for k, v in d.items():
    if "full" in k: print(k, v)
PY
