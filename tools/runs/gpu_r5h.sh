#!/bin/bash
# r05 call H: vmem_probe — the streaming shapes' DMA piece (8 rows x 128 B) against 16 rows x 64 B pieces
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05h; mkdir -p $O
timeout 120 tools/probes/vmem_probe 112 quick h 2>&1 | tee $O/vmem_probe_pieces.txt
