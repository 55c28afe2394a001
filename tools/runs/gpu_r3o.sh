#!/bin/bash
# round 3, GPU call O: buffer-resource LDS-DMA (-DVGEN_BUFDMA: SGPR base + scalar K offset + 32-bit lane offsets, zero rows
# as out-of-range offsets): model-level parity + same-box A/B first, then the K-step probe, then the kernel parity cases
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/ab.jsonl
L="vgen_amd/libvgen_hip.so vgen_amd/libvgen_hip_buf.so"
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --precision mixed" bash tools/ab_libs.sh 1 $L
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --precision fast" bash tools/ab_libs.sh 1 $L
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_stamp_buf.so timeout 60 python tools/stamp_probe.py buf 2>&1 | grep -v amdgpu.ids | tail -9
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_buf.so timeout 150 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "test_tapgemm and not dualw" -p no:cacheprovider 2>&1 | tail -4
