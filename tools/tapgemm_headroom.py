"""Where the tap-GEMM class stands against its floors, shape by shape (a reading of a committed per-shape profile — no GPU):
for every (mode, M, N, K, epilogue, output) of profiles/<shapes>.json the measured microseconds per launch next to
  * the matrix floor: executed FLOP (x 2 for a dual-W launch) at the dense 2.5 PFLOP/s, and at the 62 % in-loop MFMA
    occupancy the ping-pong K-step reaches on its best shape,
  * the HBM floor: A's source rows + the weight terms + the output, once each, at 5 TB/s (what the streaming norm kernels
    reach; a fp32 residual read, where a launch has one, is not in the profile's key and is NOT counted: the floor is low),
and the same sums by (resolution level, conv | linear).

    python tools/tapgemm_headroom.py [profiles/r04b_tapgemm_shapes_t2v_mixed.json [out.json]]   -> profiles/r04_tapgemm_headroom.json
"""
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK, OCC, HBM = 2.5e15, 0.62, 5.0e12


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04b_tapgemm_shapes_t2v_mixed.json")
    rows = json.load(open(src))["rows"]
    shapes, by = [], defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for (mode, M, N, K, epi, od), n, ms, _tf in rows:
        dw, f32 = "+dw" in od, "float32" in od
        taps = {0: 1, 1: 9, 2: 3}[mode]
        n_out = N // 2 if epi == 1 else N
        flop = 2.0 * M * N * K
        byts = 2.0 * M * K / taps + 2.0 * N * K * (2 if dw else 1) + M * n_out * (4 if f32 else 2)
        us = ms / n * 1e3
        mfma_us = flop * (2 if dw else 1) / PEAK * 1e6
        hbm_us = byts / HBM * 1e6
        floor_us = max(mfma_us / OCC, hbm_us)
        shapes.append({"shape": [mode, M, N, K, epi, od], "launches": n, "us": round(us, 1), "mfma_peak_us": round(mfma_us, 1),
                       "mfma_62pct_us": round(mfma_us / OCC, 1), "hbm_5TBs_us": round(hbm_us, 1),
                       "x_floor": round(us / floor_us, 2), "ms_above_floor": round(ms - floor_us * n / 1e3, 3)})
        b = by[(M, "conv3x3" if mode == 1 else "linear / temporal conv")]
        b[0] += n; b[1] += ms; b[2] += flop * n; b[3] += floor_us * n / 1e3
    shapes.sort(key=lambda r: -r["ms_above_floor"])
    levels = [{"rows": M, "kind": k, "launches": v[0], "ms": round(v[1], 2), "algorithmic_TFLOPs": round(v[2] / v[1] / 1e9),
               "floor_ms": round(v[3], 2)} for (M, k), v in sorted(by.items(), key=lambda kv: (-kv[0][0], kv[0][1]))]
    out = {"source": os.path.relpath(src, ROOT), "floors": {"mfma_peak": PEAK, "in_loop_occupancy": OCC, "hbm_Bps": HBM},
           "total_ms": round(sum(r[2] for r in rows), 2), "floor_ms": round(sum(v[3] for v in by.values()), 2),
           "by_level": levels, "shapes": shapes}
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r04_tapgemm_headroom.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("total %.2f ms, floor %.2f ms" % (out["total_ms"], out["floor_ms"]))
    for l in levels:
        print(l)
    for r in shapes[:12]:
        print(r)


if __name__ == "__main__":
    main()
