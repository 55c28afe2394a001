#!/bin/bash
# round 3, GPU call K: DMA pieces deferred from the read phase into the matrix phase (-DVGEN_MPH=2/3/4): parity, then same-box A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/ab.jsonl
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_mph3.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "test_tapgemm" -p no:cacheprovider 2>&1 | tail -6
L="vgen_amd/libvgen_hip.so vgen_amd/libvgen_hip_mph2.so vgen_amd/libvgen_hip_mph3.so vgen_amd/libvgen_hip_mph4.so"
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-parity --precision fast" bash tools/ab_libs.sh 2 $L
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --precision mixed" bash tools/ab_libs.sh 1 $L
