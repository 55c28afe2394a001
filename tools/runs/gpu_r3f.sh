#!/bin/bash
# round 3, GPU call F: A/B of the lgkmcnt-after-barrier variant (correctness first)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/ab.jsonl
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_lgkm.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "tapgemm" -p no:cacheprovider 2>&1 | tail -3
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-parity --precision mixed" bash tools/ab_libs.sh 2 vgen_amd/libvgen_hip.so vgen_amd/libvgen_hip_lgkm.so
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-parity --precision fast" bash tools/ab_libs.sh 2 vgen_amd/libvgen_hip.so vgen_amd/libvgen_hip_lgkm.so
