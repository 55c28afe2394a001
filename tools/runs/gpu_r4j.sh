#!/bin/bash
# r04 call J: flash_kernel's softmax denominators on the matrix pipe (ones-row MFMA) — attention parity cases, model tests,
# same-box A/B against the library before the change (vgen_amd/libvgen_hip_lsumvalu.so).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04j; mkdir -p $O; rm -f gpurun_out/ab.jsonl
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -p no:cacheprovider --maxfail=5 -k "attention or kernels_vs_plain or unet_tiny or block_alone or t2v_full_size_mixed or clip or i2vgen_full" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-scaling-model --precision mixed" bash tools/ab_libs.sh 2 vgen_amd/libvgen_hip_lsumvalu.so vgen_amd/libvgen_hip.so
AB_ARGS="--config i2vgen --steps 4 --warmup 1 --no-cpu-baseline --no-vae --no-roofline" bash tools/ab_libs.sh 1 vgen_amd/libvgen_hip_lsumvalu.so vgen_amd/libvgen_hip.so
cp gpurun_out/ab.jsonl $O/ab_lsum.jsonl
cp gpurun_out/parity.json $O/parity.json
echo R4J_DONE
