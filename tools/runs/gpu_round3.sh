#!/bin/bash
# other BASELINE configs through bench.py, VAE at 720p, bf16 + partition lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
for c in videolcm tft2v32f tft2v896 i2vgen sr600; do
  timeout 500 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-vae > $O/bench_$c.json 2> $O/bench_$c.err
  python -c "
import json
try:
    d=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], 'steps/s', d['ms_per_step'], 'ms', d['model_tflops_per_s'], 'TF/s', 'finite', d['finite'], d['roofline']['frac'], d['hbm_kernels'])
except Exception as e: print('$c FAILED', e)
"
  tail -2 $O/bench_$c.err
done
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --vae-size 720x1280 > $O/bench_vae720.json 2> $O/bench_vae720.err
timeout 200 python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 200 python bench.py --steps 20 --warmup 5 --partition --no-cpu-baseline --no-vae --no-roofline > $O/bench_partition.json 2> $O/bench_partition.err
python - <<PY
import json
for f in ("bench_vae720","bench_bf16","bench_partition"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d.get("vae"), d.get("e2e"))
    except Exception as e: print(f, "FAILED", e)
PY
echo ROUND3_DONE
