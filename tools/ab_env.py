"""Same-process, same-box A/B of tuning-build switches on the whole t2v CFG step (hipGraph replay through the public sampler
API): the model is built ONCE, every value of the environment variable gets a fresh sampling session (its graph captures the
launches as the tuning library plans them under that value), rounds are interleaved.

    python tools/ab_env.py VGEN_TAPGEMM_STAGGER "0,0 50,50,2 100,100,2" [--rounds 3] [--steps 20] [--shapes]
-> gpurun_out/ab_env.jsonl (one line per measurement) and a summary table on stdout.  --shapes: also one instrumented eager
pass per value with the per-signature tap-GEMM times (which launches moved).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VGEN_HIP_LIB", os.path.join(ROOT, "vgen_amd", "libvgen_hip_tuning.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("var")
    ap.add_argument("values")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--shapes", action="store_true")
    ap.add_argument("--config", default="t2v")
    args = ap.parse_args()
    import torch
    from bench import CONFIGS, DDIM, build_model, conditioning
    from vgen_amd import ops
    from vgen_amd.calibrate import calibrate_single_pass, calibration_batch
    from vgen_amd.diffusion import DiffusionDDIM
    dev = torch.device("cuda", 0)
    ops.set_backend(None)
    # the headline mode's launches without its host factorisations: two-term pack, every weight to nearest (k_max = 0)
    model = build_model(args.config, dev, "fp16", "high")
    xc, tc, yc = calibration_batch(CONFIGS[args.config]["latent"], n=1, device=dev)
    kwc = conditioning(args.config, model, 1, dev, torch.Generator(device=dev).manual_seed(1))[0]
    kwc["y"] = yc
    calibrate_single_pass(model, xc, tc, k_max=0, **kwc)
    g = torch.Generator(device=dev).manual_seed(8888)
    C, F, H, W = CONFIGS[args.config]["latent"]
    xt0 = torch.randn(1, C, F, H, W, generator=g, device=dev)
    kw = conditioning(args.config, model, 1, dev, g)
    t = torch.full((1,), 981, dtype=torch.long, device=dev)
    values = args.values.split()

    def step_ms(val):
        os.environ[args.var] = val
        d = DiffusionDDIM(**DDIM)
        d.rng_parity = False
        x = xt0
        for _ in range(4):
            x = d.ddim_sample(x, t, model, kw, guide_scale=9.0, ddim_timesteps=50, eta=0.0)[0]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            x = d.ddim_sample(x, t, model, kw, guide_scale=9.0, ddim_timesteps=50, eta=0.0)[0]
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) / args.steps * 1e3
        d.sessions.clear()
        return ms, bool(torch.isfinite(x).all())

    out = open(os.path.join(ROOT, "gpurun_out", "ab_env.jsonl"), "a")
    res = {v: [] for v in values}
    for r in range(args.rounds):
        for v in values:
            ms, fin = step_ms(v)
            res[v].append(ms)
            out.write(json.dumps({"var": args.var, "value": v, "round": r, "ms_per_step": round(ms, 4), "finite": fin}) + "\n")
            out.flush()
    base = min(res[values[0]])
    for v in values:
        print(f"{args.var}={v:14s} best {min(res[v]):7.3f} ms  median {sorted(res[v])[len(res[v]) // 2]:7.3f}  "
              f"vs first {100 * (min(res[v]) / base - 1):+5.1f} %", flush=True)
    if args.shapes:
        d0 = DiffusionDDIM(**DDIM)
        d0.rng_parity = False
        d0.sessions = None
        per = {}
        for v in values:
            os.environ[args.var] = v
            d0.ddim_sample(xt0, t, model, kw, guide_scale=9.0, ddim_timesteps=50, eta=0.0)
            acc = {}
            for _ in range(2):
                ops.KERNEL_PROFILE = []
                torch.cuda.synchronize()
                torch.cuda._sleep(int(2e8))
                d0.ddim_sample(xt0, t, model, kw, guide_scale=9.0, ddim_timesteps=50, eta=0.0)
                torch.cuda.synchronize()
                for rec in ops.KERNEL_PROFILE:
                    if rec[0] == "tapgemm":
                        a = acc.setdefault(rec[4][:6], [0, 0.0])
                        a[0] += 1
                        a[1] += rec[1].elapsed_time(rec[2])
                ops.KERNEL_PROFILE = None
            per[v] = {str(k): (n // 2, round(ms / 2, 4)) for k, (n, ms) in acc.items()}
        keys = sorted(per[values[0]], key=lambda k: -per[values[0]][k][1])[:40]
        rows = [[k, per[values[0]][k][0]] + [per[v].get(k, (0, 0))[1] for v in values] for k in keys]
        json.dump({"values": values, "cols": ["signature", "launches"] + [f"ms @ {v}" for v in values], "rows": rows,
                   "total_ms": {v: round(sum(x[1] for x in per[v].values()), 3) for v in values}},
                  open(os.path.join(ROOT, "gpurun_out", "ab_env_shapes.json"), "w"), indent=0)
        print("tap-GEMM class ms per value:", {v: round(sum(x[1] for x in per[v].values()), 3) for v in values})


if __name__ == "__main__":
    main()
