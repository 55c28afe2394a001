#!/bin/bash
# round 3, GPU call L: s_memtime segment sums of the ping-pong K-step (-DVGEN_STAMP), product schedule and the MPH=3 one
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_stamp.so timeout 240 python tools/stamp_probe.py base 2>&1 | grep -v amdgpu.ids | tail -12
VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_stamp_mph3.so timeout 240 python tools/stamp_probe.py mph3 2>&1 | grep -v amdgpu.ids | tail -12
