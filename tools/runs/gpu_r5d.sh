#!/bin/bash
# r05 call D: tools/probes/vmem_probe — what a CU's vector-memory path sustains for the panel kernel's access patterns
# (row-quad = MFMA layout straight from row-major memory, full-line = 8 rows x 128 B, linear), loads / stores / both.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05d; mkdir -p $O
timeout 120 tools/probes/vmem_probe 2>&1 | tee $O/vmem_probe.txt
