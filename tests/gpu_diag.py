"""Not a pytest file: runs every kernel case on the GPU and dumps statistics (+ coarse error maps
for the tap-GEMM) to gpurun_out/diag.json so one GPU call yields a full picture."""
import json
import os
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kernel_cases as kc  # noqa: E402
from vgen_amd import ops  # noqa: E402


def main():
    dev = "cuda:0"
    ops.set_backend(None)
    be = ops.backend()
    out = {}

    def run(name, fn):
        t0 = time.time()
        try:
            out[name] = fn()
            torch.cuda.synchronize()
        except Exception as e:  # noqa
            out[name] = {"error": repr(e), "tb": traceback.format_exc()[-1500:]}
        out[name]["_sec"] = round(time.time() - t0, 3) if isinstance(out[name], dict) else 0
        worst = max([v.get("rel_l2", 0) for v in out[name].values() if isinstance(v, dict)] + [0])
        print(f"{name:50s} worst rel_l2 {worst:.3e} {'ERROR ' + out[name]['error'] if 'error' in out[name] else ''}", flush=True)

    for dtn, dt in kc.DTS.items():
        for c in kc.GN_CASES:
            run(f"gn/{dtn}/nb{c[0]}_S{c[1]}_C{c[2]}+{c[3]}", lambda c=c: kc.case_groupnorm(be, dev, dt, *c))
        for c in kc.LN_CASES:
            run(f"ln/{dtn}/M{c[0]}_d{c[1]}", lambda c=c: kc.case_layernorm(be, dev, dt, *c))
        for name, spec in kc.tapgemm_cases(dt).items():
            run(f"tapgemm/{dtn}/{name}", lambda spec=spec: kc.case_tapgemm(be, dev, spec, want_map=True))
        for name, spec in kc.attn_cases(dt).items():
            run(f"attn/{dtn}/{name}", lambda spec=spec: kc.case_attention(be, dev, spec))
        run(f"softmax/{dtn}", lambda: kc.case_softmax_rows(be, dev, dt, 70, 200, 256))
        run(f"act_cast/{dtn}", lambda: kc.case_act_cast(be, dev, dt, 5000, 1))
        run(f"temb/{dtn}", lambda: kc.case_timestep_embedding(be, dev, dt, 320))
        run(f"im2col_bcfhw/{dtn}", lambda: kc.case_im2col(be, dev, dt, "bcfhw"))
        run(f"im2col_rows/{dtn}", lambda: kc.case_im2col(be, dev, dt, "rows"))
    run("pointwise", lambda: kc.case_pointwise(be, dev))
    run("gaussian", lambda: kc.case_gaussian(be, dev))
    for mt in (0, 1, 2):
        for eta in (0.0, 0.7):
            run(f"cfg_ddim/mt{mt}_eta{eta}", lambda: kc.case_cfg_ddim(be, dev, mt, eta))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
