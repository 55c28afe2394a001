#!/bin/bash
# Same-box A/B of library variants (vgen_amd/build.py build(variant=..., defines=...)): the headline step of bench.py,
# interleaved rounds, one JSON line per run into gpurun_out/ab_<tag>.jsonl.   usage: ab_libs.sh ROUNDS lib1.so lib2.so ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ROUNDS=$1; shift
ARGS=${AB_ARGS:-"--steps 20 --warmup 5 --variants= --no-cpu-baseline --no-vae --no-roofline --no-parity"}
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    VGEN_HIP_LIB=$PWD/$lib timeout 300 python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'lib': '$tag', 'round': $r, 'ms_per_step': d['ms_per_step'], 'precision': d['config']['precision'], 'dtype': d['dtype'], 'parity_rel_l2': (d.get('parity') or {}).get('unet_rel_l2')}))" | tee -a gpurun_out/ab.jsonl
  done
done
