#!/bin/bash
# First GPU call of the next round: decide the buffer-resource LDS-DMA (-DVGEN_BUFDMA).
# BEFORE gpurun, in the build container (variant libraries are git-ignored, they travel with the snapshot):
#   python -m vgen_amd.build --variant=buf -DVGEN_BUFDMA
#   python -m vgen_amd.build --variant=stamp -DVGEN_STAMP
#   python -m vgen_amd.build --variant=stamp_buf -DVGEN_STAMP -DVGEN_BUFDMA
#   python -m vgen_amd.build --variant=bm224 -DVGEN_BM224                     (224-row dual tiles)
#   python -m vgen_amd.build --variant=bm224t -DVGEN_BM224 -DVGEN_TUNING      (to FORCE the new shape in the parity cases)
#   python -m vgen_amd.build --variant=buf224 -DVGEN_BM224 -DVGEN_BUFDMA
# ~6 GPU-minutes: model-level parity + same-box A/B first, K-step probe, then every tap-GEMM kernel parity case.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/ab.jsonl
L="vgen_amd/libvgen_hip.so vgen_amd/libvgen_hip_buf.so vgen_amd/libvgen_hip_bm224.so vgen_amd/libvgen_hip_buf224.so"
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --precision mixed" bash tools/ab_libs.sh 2 $L
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --precision fast" bash tools/ab_libs.sh 2 $L
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_stamp.so timeout 60 python tools/stamp_probe.py base 2>&1 | grep -v amdgpu.ids | tail -9
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_stamp_buf.so timeout 60 python tools/stamp_probe.py buf 2>&1 | grep -v amdgpu.ids | tail -9
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_buf.so timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "test_tapgemm" -p no:cacheprovider 2>&1 | tail -4
# the 224-row shapes forced on every parity case they are legal for (plan "3,<bn>,1" = dual224, "4,320,<split-K>" = the
# 224 x 320 ping-pong tile; illegal cases fall back to the model's plan)
for plan in 3,160,1 3,128,1 3,64,1 4,320,1 4,320,2; do
  VGEN_TAPGEMM_PLAN="$plan" VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_bm224t.so timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "test_tapgemm and not dualw" -p no:cacheprovider 2>&1 | tail -2
done
