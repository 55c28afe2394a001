#!/bin/bash
# r04 call F: block size of the LDS-resident single-launch GroupNorm (73 launches per step at ~14 us: latency-bound) — same-box
# A/B of the whole step, 256 / 512 (product) / 1024 threads; GroupNorm parity cases under the variants.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04f; mkdir -p $O; rm -f gpurun_out/ab.jsonl
AB_ARGS="--steps 30 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-scaling-model --no-parity --precision mixed" bash tools/ab_libs.sh 3 vgen_amd/libvgen_hip.so vgen_amd/libvgen_hip_gnf256.so vgen_amd/libvgen_hip_gnf1024.so
cp gpurun_out/ab.jsonl $O/ab_gnf_threads.jsonl
for v in gnf256 gnf1024; do VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_$v.so timeout 200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -k "groupnorm or kernels_vs_plain" 2>&1 | tail -1; done
