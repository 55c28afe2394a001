#!/bin/bash
# round 3, GPU call J: the one-wave-per-SIMD (W4) tap-GEMM loop: kernel parity under the variant library, then same-box A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/ab.jsonl
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_w4.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "test_tapgemm and not dualw" -p no:cacheprovider 2>&1 | tail -6
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-parity --precision fast" bash tools/ab_libs.sh 2 vgen_amd/libvgen_hip.so vgen_amd/libvgen_hip_w4.so
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --precision mixed" bash tools/ab_libs.sh 1 vgen_amd/libvgen_hip.so vgen_amd/libvgen_hip_w4.so
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_w4.so timeout 300 python bench.py --steps 10 --warmup 3 --precision fast --variants= --no-cpu-baseline --no-vae --no-parity --dump-shapes > /dev/null 2>&1
cp gpurun_out/tapgemm_shapes_t2v_fp16_fast.json gpurun_out/tapgemm_shapes_t2v_fp16_fast_w4.json
