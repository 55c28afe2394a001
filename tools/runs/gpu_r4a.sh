#!/bin/bash
# r04 call A: decide -DVGEN_BUFDMA (buffer-resource LDS-DMA) and -DVGEN_BM224 (224-row tiles) — DESIGN 3.1 / 8.
# Variant libraries are built in the container first (git-ignored, they travel with the snapshot):
#   python -m vgen_amd.build --variant=buf -DVGEN_BUFDMA;  --variant=bm224 -DVGEN_BM224;
#   --variant=bm224t -DVGEN_BM224 -DVGEN_TUNING;  --variant=buf224 -DVGEN_BM224 -DVGEN_BUFDMA
# Same-box A/B of the whole step first (with in-run parity), then the tap-GEMM parity cases under the variants.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/ab.jsonl
L="vgen_amd/libvgen_hip.so vgen_amd/libvgen_hip_buf.so vgen_amd/libvgen_hip_bm224.so vgen_amd/libvgen_hip_buf224.so"
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --precision mixed" bash tools/ab_libs.sh 2 $L
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --precision fast" bash tools/ab_libs.sh 1 $L
cp gpurun_out/ab.jsonl gpurun_out/r04a_ab_libs.jsonl
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_buf.so timeout 240 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "test_tapgemm" -p no:cacheprovider 2>&1 | tail -4
# the 224-row shapes forced on every parity case they are legal for (plan "3,<bn>,1" = dual224, "4,320,<split-K>" = the
# 224 x 320 ping-pong tile; illegal cases fall back to the model's plan)
for plan in 3,160,1 3,64,1 4,320,1; do
  VGEN_TAPGEMM_PLAN="$plan" VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_bm224t.so timeout 240 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "test_tapgemm and not dualw" -p no:cacheprovider 2>&1 | tail -2
done
echo R4A_DONE
