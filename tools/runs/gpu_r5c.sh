#!/bin/bash
# r05 call C: where a panel wave's cycles go — s_memtime stamps per segment of the slice loop (tuning build) and an SQ
# counter pass over the panel kernels alone; plus the i2vgen-xl 704p step with the panel shape off / on.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c; mkdir -p $O
export VGEN_HIP_LIB=$PWD/vgen_amd/libvgen_hip_tuning.so
timeout 300 python tools/panel_probe.py $O/panel_stamps.json --stamps --panel-only 2>&1 | tee $O/panel_stamps.log
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/prof_panel -- python $R/tools/panel_probe.py --panel-only --only=geglu --only=qkv --only=o-proj > /dev/null 2>&1
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/prof_panel/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "panel_kernel" not in k: continue
        k = k.split("panel_kernel")[1][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
out = {k: {c: v / max(cnt[k], 1) for c, v in d.items()} | {"launches": cnt[k]} for k, d in agg.items()}
json.dump(out, open("$R/$O/panel_pmc.json", "w"), indent=1)
for k, d in out.items():
    w = d.get("SQ_WAVE_CYCLES", 1)
    print(k, {c: round(v / w, 3) for c, v in d.items() if c != "launches"}, d["launches"])
PY
cd $R
A="--steps 6 --warmup 2 --variants= --no-cpu-baseline --no-vae --no-roofline --no-e2e --no-scaling-model --no-parity --config i2vgen"
for pm in 0 1; do
  VGEN_TAPGEMM_PANEL=$pm timeout 300 python bench.py $A 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'panel': $pm, 'config': 'i2vgen', 'ms_per_step': d['ms_per_step']}))" | tee -a $O/ab_panel_i2vgen.jsonl
done
