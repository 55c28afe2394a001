cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
STAGES="ab" TAG=r6b AB_VAR=VGEN_TAPGEMM_STAGGER AB_VALUES="0,0 50,50,2 50,0,2 0,50,2 25,25,2 100,100,3 50,50,4 75,75,2" AB_ROUNDS=3 AB_EXTRA="--shapes" bash tools/runs/gpu_r6.sh
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -k "vae_full_size_calibrated or pp256_f32_res_splitk3 or vae_full_size_decode_high" -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/r6b/pytest.log
cp gpurun_out/parity.json gpurun_out/r6b/ 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --variants= > gpurun_out/r6b/bench.json 2> gpurun_out/r6b/bench.err; tail -c 2500 gpurun_out/r6b/bench.json; tail -3 gpurun_out/r6b/bench.err
