#!/bin/bash
# round 3, GPU call B: flash kernel change correctness, A/B of experiment libs, fast-mode per-shape table of this box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; rm -f gpurun_out/ab.jsonl
echo "== attention + full-size tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -k "attention or plain_torch or t2v_full or clip" -p no:cacheprovider 2>&1 | tail -8
echo "== A/B high mode"
bash tools/ab_libs.sh 2 vgen_amd/libvgen_hip.so vgen_amd/libvgen_hip_setprio.so vgen_amd/libvgen_hip_flashpk.so
echo "== A/B fast mode"
AB_ARGS="--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-parity --precision fast" bash tools/ab_libs.sh 2 vgen_amd/libvgen_hip.so vgen_amd/libvgen_hip_setprio.so vgen_amd/libvgen_hip_flashpk.so
echo "== fast-mode shapes on this box"
timeout 300 python bench.py --steps 10 --warmup 3 --precision fast --variants= --no-cpu-baseline --no-vae --no-parity --dump-shapes | tail -c 600
