#!/bin/bash
# r04 call B: the whole -m gpu suite on the ABI-5 library (new full-width fixtures in mixed, session re-bind, VAE chunk
# graphs, fused partition step under RCCL capture), then the default bench line, the VAE shape dump and the two new
# bench configurations (videolcm whole videos; 2-step smoke of the two-stage pipeline).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04b; mkdir -p $O
timeout 1100 python -m pytest tests -q -m gpu -p no:cacheprovider --maxfail=8 > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
cp gpurun_out/parity.json $O/parity.json 2>/dev/null
timeout 400 python bench.py --steps 20 --warmup 5 --dump-shapes > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json; tail -3 $O/bench.err
cp gpurun_out/*shapes*t2v_fp16_mixed.json gpurun_out/vae_shapes_*.json $O/ 2>/dev/null
timeout 300 python bench.py --config videolcm --steps 16 --warmup 4 > $O/bench_videolcm.json 2> $O/bench_videolcm.err; tail -c 1500 $O/bench_videolcm.json; tail -3 $O/bench_videolcm.err
timeout 400 python bench.py --config tft2v_sr600 --steps 2 > $O/bench_two_stage_smoke.json 2> $O/bench_two_stage_smoke.err; tail -c 1500 $O/bench_two_stage_smoke.json; tail -5 $O/bench_two_stage_smoke.err
echo R4B_DONE
