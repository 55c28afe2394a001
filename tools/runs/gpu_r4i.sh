#!/bin/bash
# r04 call I: fp16 pack2 as ONE v_cvt_pk_f16_f32 (was cvt + cvt_sdwa + or per pair) in every 16-bit epilogue: all kernel parity
# cases, the model tests with bit-reproducibility, and a same-box A/B against the library built before the change
# (vgen_amd/libvgen_hip_pack3.so); the in-run parity values must be IDENTICAL (the conversion is RNE either way).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04i; mkdir -p $O; rm -f gpurun_out/ab.jsonl
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider --maxfail=5 > $O/pytest_kernels.log 2>&1; tail -3 $O/pytest_kernels.log
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-scaling-model --precision mixed" bash tools/ab_libs.sh 2 vgen_amd/libvgen_hip_pack3.so vgen_amd/libvgen_hip.so
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-scaling-model --precision fast" bash tools/ab_libs.sh 1 vgen_amd/libvgen_hip_pack3.so vgen_amd/libvgen_hip.so
AB_ARGS="--config i2vgen --steps 4 --warmup 1 --no-cpu-baseline --no-vae --no-roofline" bash tools/ab_libs.sh 1 vgen_amd/libvgen_hip_pack3.so vgen_amd/libvgen_hip.so
cp gpurun_out/ab.jsonl $O/ab_pack2.jsonl
timeout 500 python -m pytest tests/test_gpu_model.py -q -m gpu -p no:cacheprovider --maxfail=5 -k "tiny or block_alone or t2v_full_size or vae_full_size_decode or session" > $O/pytest_model.log 2>&1; tail -3 $O/pytest_model.log
echo R4I_DONE
