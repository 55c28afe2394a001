#!/bin/bash
# r05 call A: (1) the HEAD default `mixed` rule's GPU parity on ALL eight full-width fixtures, values not pass/fail
# (VERDICT r04 weak #1a) -> gpurun_out/r05a/parity.json; (2) the default bench line with per-shape tap-GEMM timings of the
# default rule (the r05 kernel work's baseline: profiles/r05a_tapgemm_shapes_t2v_mixed.json).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05a; mkdir -p $O
timeout 600 python - <<PY 2>&1 | tee $O/parity.log
import sys, json, time, torch
sys.path.insert(0, "tests")
import full_cases as fc
res = {}
for name in ("t2v", "t2v_b", "t2v_c", "videolcm", "tft2v", "i2vgen", "sr600", "vcomposer"):
    g = fc.load(name)
    for pr in (("mixed", "high") if name == "vcomposer" else ("mixed",)):
        t0 = time.perf_counter()
        m = fc.build(name, g, pr, "cuda")
        out = fc.forward(name, m, g, "cuda")
        torch.cuda.synchronize()
        e, nr = fc.error(out, g)
        res[name + ":" + pr] = {"rel_l2": e, "norm_ratio": nr}
        print(name, pr, "rel-L2 %.4e" % e, "norm ratio %.5f" % nr, "(%.0f s)" % (time.perf_counter() - t0), flush=True)
        del m, out
        torch.cuda.empty_cache()
json.dump(res, open("$O/parity.json", "w"), indent=1)
PY
timeout 300 python bench.py --steps 20 --warmup 5 --variants=fp16/fast --no-cpu-baseline --no-vae --no-e2e --dump-shapes > $O/bench_default.json 2> $O/bench_default.err
cp gpurun_out/tapgemm_shapes_t2v_fp16_mixed.json gpurun_out/other_shapes_t2v_fp16_mixed.json $O/ 2>/dev/null
tail -c 1500 $O/bench_default.json
