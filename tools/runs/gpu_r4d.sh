#!/bin/bash
# r04 call D: V^T fragments through gfx950's LDS transpose read (flash_kernel, temporal_kernel): attention parity cases, the
# model tests that exercise them, and a same-box A/B of the whole step (t2v mixed; i2vgen, where flash attention is a quarter
# of the step) against the library with the r03 attention kernels (vgen_amd/libvgen_hip_oldattn.so, built in the container).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04d; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -p no:cacheprovider --maxfail=8 -k "attention or kernels_vs_plain or tiny or block_alone or t2v_full_size_mixed or clip" > $O/pytest_attn.log 2>&1; tail -5 $O/pytest_attn.log
rm -f gpurun_out/ab.jsonl
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-scaling-model --precision mixed" bash tools/ab_libs.sh 2 vgen_amd/libvgen_hip_oldattn.so vgen_amd/libvgen_hip.so
AB_ARGS="--config i2vgen --steps 4 --warmup 1 --no-cpu-baseline --no-vae --no-roofline" bash tools/ab_libs.sh 1 vgen_amd/libvgen_hip_oldattn.so vgen_amd/libvgen_hip.so
cp gpurun_out/ab.jsonl $O/ab_attention.jsonl
echo R4D_DONE
