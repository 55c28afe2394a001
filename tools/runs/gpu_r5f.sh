#!/bin/bash
# r05 call F: vmem_probe with the quad-contiguous renumbering of the C-fragment pieces (pattern 3), at the GEGLU-sized and
# the o-proj-sized (outputs stay in the Infinity Cache) footprints.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05f; mkdir -p $O
(timeout 120 tools/probes/vmem_probe 112 quick; timeout 120 tools/probes/vmem_probe 14 quick) 2>&1 | tee $O/vmem_probe_quad.txt
