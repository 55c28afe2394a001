"""gpurun_out/evidence/* (tools/collect_evidence.sh) -> profiles/<tag>_{bench_evidence.json, kernel_stats_summary.csv,
pmc_classes.json, tapgemm_traffic.json}.  The traffic file is what bench.py quotes under roofline.committed.

    python tools/evidence_to_profiles.py r03 [precision] [dtype]
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(ROOT, "gpurun_out", "evidence")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
prec = sys.argv[2] if len(sys.argv) > 2 else "mixed"
dtype = sys.argv[3] if len(sys.argv) > 3 else "fp16"
P = os.path.join(ROOT, "profiles")
def bench_line(path):
    """the bench's JSON line (older runs: RCCL's banner could follow it on stdout)"""
    return json.loads([ln for ln in open(path).read().splitlines() if ln.startswith('{"metric"')][-1])


bench = bench_line(os.path.join(E, "bench.json"))
json.dump(bench, open(os.path.join(P, f"{tag}_bench_evidence.json"), "w"), indent=1)
shutil.copy(os.path.join(E, "kernel_stats_summary.csv"), os.path.join(P, f"{tag}_kernel_stats_summary.csv"))
pmc = json.load(open(os.path.join(E, "pmc_classes.json")))
json.dump(pmc, open(os.path.join(P, f"{tag}_pmc_classes.json"), "w"), indent=1)
t = pmc["tapgemm"]
# FETCH_SIZE / WRITE_SIZE are in KB; gfx950 tallies 128-B requests at 64 B -> FETCH doubled (MI355X_MICROARCH.md, HBM section)
fetch = 2.0 * t["FETCH_SIZE_per_launch"] * 1024.0
write = t["WRITE_SIZE_per_launch"] * 1024.0
busy = t["SQ_VALU_MFMA_BUSY_CYCLES_total"] / (t["GRBM_GUI_ACTIVE_total"] * 32.0 * 4.0)
out = {"source": "rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE | "
                 "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY) on `bench.py --steps 1 --warmup 1 --no-graph` "
                 f"({dtype}, {prec}, t2v), tools/collect_evidence.sh + tools/pmc_classes.py; FETCH_SIZE doubled (gfx950 "
                 "tallies 128-B requests at 64 B), KB -> bytes x1024",
       "precision": prec, "dtype": dtype, "launches": t["launches"], "fetch_bytes_per_launch": fetch,
       "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
       "algorithmic_bytes_per_launch": bench.get("roofline", {}).get("algorithmic_bytes_per_launch"),
       "mfma_busy_frac": round(busy, 4)}
json.dump(out, open(os.path.join(P, f"{tag}_tapgemm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
for f, dst in (("pytest_gpu.log", "pytest_gpu.log"), ("smoke.log", "smoke.log"), ("parity.json", "parity.json"),
               ("bench_partition_eager.json", "bench_partition_eager.json"),
               ("bench_partition_eager_rccl.json", "bench_partition_eager_rccl.json"),
               ("bench_partition_graph_rccl.json", "bench_partition_graph_rccl.json"),
               ("bench_2ranks_one_device.json", "bench_2ranks_one_device.json")):
    if os.path.exists(os.path.join(E, f)):
        if f.startswith("bench_"):
            json.dump(bench_line(os.path.join(E, f)), open(os.path.join(P, f"{tag}_{dst}"), "w"), indent=1)
        else:
            shutil.copy(os.path.join(E, f), os.path.join(P, f"{tag}_{dst}"))
