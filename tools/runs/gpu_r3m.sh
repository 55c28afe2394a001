#!/bin/bash
# round 3, GPU call M: branch-free A-pointer regather (-DVGEN_GATHER2) and the pointer step on the read-phase side (-DVGEN_ADV_R)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/ab.jsonl
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_g2r.so timeout 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "test_tapgemm" -p no:cacheprovider 2>&1 | tail -4
L="vgen_amd/libvgen_hip.so vgen_amd/libvgen_hip_g2.so vgen_amd/libvgen_hip_advr.so vgen_amd/libvgen_hip_g2r.so"
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-parity --precision fast" bash tools/ab_libs.sh 1 $L
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --precision mixed" bash tools/ab_libs.sh 1 $L
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_stamp_g2r.so timeout 120 python tools/stamp_probe.py g2r 2>&1 | grep -v amdgpu.ids | tail -9
