"""bench.py — denoise-steps/sec of the t2v sampling hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one classifier-free-guidance DDIM denoise step of the t2v configuration
(BASELINE.json configs[1]: latent [1,4,16,32,56] = 16 frames 448x256, 77x1024 context, guide 9,
UNetSD_T2VBase 1411 M parameters): two UNet forwards (evaluated as one batch of 2 units) + the
fused CFG/DDIM update, with all inputs resident in HBM.  Synthetic latents/context, seeded
random-init weights (no checkpoints offline).

N > 1 (weak scaling): P = N prompts in flight -> 2N units spread over the ranks (unit u -> rank
u % N), ONE all-gather of the unit outputs per step (RCCL), every rank applies the cheap update for
all prompts.  value = prompts * steps / max-over-ranks time.  The local units' forward is replayed as a
hipGraph (vgen_amd.graph.GraphedForward); the all-gather and the update stay eager.  `--partition`
runs that same code path at N = 1.

Extra objects on the JSON line:
  roofline     — dominant kernel (tap-GEMM, MFMA-bound): algorithmic FLOP per launch / average launch
                 duration, measured with HIP events on the launch stream in an instrumented pass of
                 the same step; peak = 2.5 PFLOP/s dense 16-bit MFMA.
                 `traffic` = HBM-side bytes per launch from the committed rocprofv3 PMC passes of this
                 command (profiles/r01_tapgemm_traffic.json; tools/collect_evidence.sh regenerates them).
  cpu_baseline — the oracle (CPU port of the reference forward, oracle/torch_ref.py) timed on the
                 host cores on a bounded sample (full-size UNet, 4-frame latent, 32 threads), scaled to
                 a full step; rank 0 at N=1.
  vae          — AutoencoderKL decode frames/s at 256x448 (decoder_bs = 2 like t2v_infer.yaml).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

UNET_T2V = dict(in_dim=4, dim=320, y_dim=1024, context_dim=1024, out_dim=4, dim_mult=[1, 2, 4, 4],
                num_heads=8, head_dim=64, num_res_blocks=2, attn_scales=[1.0, 0.5, 0.25], dropout=0.1,
                temporal_attention=True, temporal_attn_times=1, use_checkpoint=False,
                use_fps_condition=False, use_sim_mask=False)
VAE_SD = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
              ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
DDIM = dict(schedule="cosine", schedule_param=dict(num_timesteps=1000, cosine_s=0.008, zero_terminal_snr=True),
            mean_type="v", loss_type="mse", var_type="fixed_small", rescale_timesteps=False)
UNET_FWD_TFLOP = 8.665      # one forward, B=1, [1,4,16,32,56], ctx 77 (FlopCounterMode)
VAE_DEC_TFLOP = 1.092       # per 256x448 frame
PEAK_TFLOPS = 2500.0        # dense bf16/fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md


def randomize_(module, seed):
    """Seeded non-degenerate weights on the device (zero-init layers included, SURVEY §8c trap)."""
    g = torch.Generator(device=next(module.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g, device=p.device) * (0.8 / fan_in ** 0.5))
            elif name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g, device=p.device))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g, device=p.device))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--partition", action="store_true",
                    help="use the multi-GPU code path (UnitPartition + graphed local forward + eager gather/update) "
                         "even at --gpus 1")
    ap.add_argument("--dump-shapes", action="store_true", help="write per-shape tap-GEMM timings to gpurun_out/")
    args = ap.parse_args()

    import torch.distributed as dist
    from vgen_amd import ops
    from vgen_amd.diffusion import DiffusionDDIM
    from vgen_amd.parallel import UnitPartition
    from vgen_amd.unet import UNetSD_T2VBase

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ops.set_backend(None)
    assert ops.backend().name == "hip"

    with torch.device(dev):
        model = UNetSD_T2VBase(**UNET_T2V, compute_dtype=args.dtype)
    model.eval()
    randomize_(model, 0)
    model.pack()
    for p in model.parameters():                      # fp32 masters are not needed for sampling
        p.data = torch.empty(0, device=dev)
    torch.cuda.empty_cache()

    diff = DiffusionDDIM(**DDIM)
    diff.rng_parity = False
    P = world                                           # prompts in flight (weak scaling)
    g = torch.Generator(device=dev).manual_seed(8888)
    xt = torch.randn(P, 4, 16, 32, 56, generator=g, device=dev)
    y_c = torch.randn(P, 77, 1024, generator=g, device=dev)
    y_u = torch.randn(P, 77, 1024, generator=g, device=dev)
    kw = [dict(y=y_c), dict(y=y_u)]
    steps_all = (1 + torch.arange(0, 1000, 20)).clamp(0, 999).flip(0).tolist()
    part = UnitPartition() if (world > 1 or args.partition) else None
    diff.partition = part
    step_model = model
    if part is not None and not args.no_graph:
        # the RCCL all-gather stays outside the graph: replay the local units' forward, then gather + update
        from vgen_amd.graph import GraphedForward
        step_model = GraphedForward(model, warmup=1)

    state = {"xt": xt}
    t_buf = torch.full((P,), steps_all[0], dtype=torch.long, device=dev)

    def one_step():
        state["xt"], _ = diff.ddim_sample(state["xt"], t_buf, step_model, kw, guide_scale=9.0,
                                          ddim_timesteps=50, eta=0.0)

    use_graph = (not args.no_graph) and part is None
    graph = None
    static_in = None
    if use_graph:
        static_in = xt.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                diff.ddim_sample(static_in, t_buf, model, kw, guide_scale=9.0, ddim_timesteps=50, eta=0.0)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out, _ = diff.ddim_sample(static_in, t_buf, model, kw, guide_scale=9.0,
                                             ddim_timesteps=50, eta=0.0)

        def one_step():  # noqa: F811
            graph.replay()
            static_in.copy_(static_out)

    def set_t(i):
        t_buf.fill_(steps_all[i % len(steps_all)])

    if step_model is not model:      # untimed: eager pass + capture of the local forward (like the N=1 capture above)
        for _ in range(2):
            one_step()
        state["xt"] = xt

    for i in range(args.warmup):
        set_t(i)
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        set_t(args.warmup + i)
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt_s], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_s = float(tt.item())
    xt_end = state["xt"] if not use_graph else static_in
    finite = bool(torch.isfinite(xt_end).all())
    xt_absmax = float(xt_end.float().abs().nan_to_num(nan=float("inf")).max())

    steps_per_s = P * args.steps / dt_s
    res = {
        "metric": "denoise_steps_per_sec", "value": round(steps_per_s, 4), "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt_s / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "t2v 16x448x256 latent [1,4,16,32,56], DDIM CFG step (2 UNetSD_T2VBase fwd "
                               "+ fused update), guide 9, 77x1024 ctx, random-init 1411M params",
                   "prompts_in_flight": P, "units_per_step": 2 * P,
                   "parallelism": "single GPU" if world == 1 else f"unit partition over {world} ranks, 1 all-gather/step",
                   "hipgraph": "whole step" if use_graph else ("local forward" if step_model is not model else False)},
        "finite": finite, "latent_absmax_after_timed_steps": xt_absmax,
        "model_tflops_per_s": round(2 * UNET_FWD_TFLOP * steps_per_s, 2),
    }

    # ---- roofline of the dominant kernel (instrumented eager pass, same step) ----------------------
    if rank == 0 and not args.no_roofline:
        ops.KERNEL_PROFILE = []
        xs = xt[:1].clone()
        kw1 = [dict(y=y_c[:1]), dict(y=y_u[:1])]
        diff.partition = None
        torch.cuda.synchronize()
        torch.cuda._sleep(int(4e8))     # let the host run ahead so event pairs bracket GPU time only
        diff.ddim_sample(xs, t_buf[:1], model, kw1, guide_scale=9.0, ddim_timesteps=50, eta=0.0)
        torch.cuda.synchronize()
        allrecs = ops.KERNEL_PROFILE
        ops.KERNEL_PROFILE = None
        recs = [r for r in allrecs if r[0] == "tapgemm"]
        diff.partition = part
        ms = [r[1].elapsed_time(r[2]) for r in recs]
        fl = [r[3] for r in recs]
        if args.dump_shapes:
            other = {}
            for r in allrecs:
                if r[0] == "tapgemm":
                    continue
                a = other.setdefault((r[0],) + tuple(r[4]), [0, 0.0, 0.0])
                a[0] += 1
                a[1] += r[1].elapsed_time(r[2])
                a[2] += r[3]
            orows = sorted(([list(k), v[0], round(v[1], 4), round(v[2] / (v[1] * 1e-3) / 1e9, 1)] for k, v in other.items()),
                           key=lambda r: -r[2])
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "other_shapes.json"), "w") as f:
                json.dump({"cols": ["(op,shape...)", "launches", "ms", "GB/s or GFLOP/s"], "rows": orows}, f, indent=0)
            agg = {}
            for r, m in zip(recs, ms):
                a = agg.setdefault(r[4], [0, 0.0, 0.0])
                a[0] += 1
                a[1] += m
                a[2] += r[3]
            rows = sorted(([list(k), v[0], round(v[1], 4), round(v[2] / (v[1] * 1e-3) / 1e12, 1)] for k, v in agg.items()),
                          key=lambda r: -r[2])
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "tapgemm_shapes.json"), "w") as f:
                json.dump({"cols": ["(mode,M,N,K,epi,out)", "launches", "ms", "TFLOP/s"], "rows": rows}, f, indent=0)
        tot_ms, tot_fl = sum(ms), sum(fl)
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        res["roofline"] = {"kernel": "tapgemm_kernel", "bound": "mfma", "achieved": round(ach, 2),
                           "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS, 4),
                           "traffic": None, "launches_per_step": len(recs),
                           "avg_launch_us": round(1e3 * tot_ms / max(len(recs), 1), 2),
                           "avg_gflop_per_launch": round(tot_fl / max(len(recs), 1) / 1e9, 3),
                           "tapgemm_ms_per_step": round(tot_ms, 3)}
        # HBM-side bytes per launch cannot be read from inside the process: they come from the committed
        # rocprofv3 PMC passes of this same command (tools/collect_evidence.sh -> profiles/)
        tpath = os.path.join(ROOT, "profiles", "r01_tapgemm_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            res["roofline"]["traffic"] = round(tj["hbm_bytes_per_launch"])
            res["roofline"]["traffic_unit"] = "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, profiles/r01_tapgemm_traffic.json)"

    # ---- VAE decode frames/s ---------------------------------------------------------------------------
    if rank == 0 and not args.no_vae:
        from vgen_amd.vae import AutoencoderKL
        with torch.device(dev):
            vae = AutoencoderKL(ddconfig=VAE_SD, embed_dim=4, compute_dtype=args.dtype)
        vae.eval()
        randomize_(vae, 1)
        z = torch.randn(2, 4, 32, 56, device=dev) / 0.18215 * 0.2
        for _ in range(2):
            vae.decode(z)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nrep = 4
        for _ in range(nrep):
            vae.decode(z)
        torch.cuda.synchronize()
        fps = 2 * nrep / (time.perf_counter() - t1)
        res["vae"] = {"decode_frames_per_sec": round(fps, 2), "frame": "256x448", "decoder_bs": 2,
                      "tflops_per_s": round(fps * VAE_DEC_TFLOP, 2)}
        del vae

    # ---- CPU baseline: the oracle on host cores, bounded sample -------------------------------------------
    # Sample = the full-size UNetSD_T2VBase (1411 M params) on a 4-frame latent [1,4,4,32,56] — 1/4 of the
    # 16-frame forward (conv/linear FLOPs are proportional to F) — scaled x4 to one forward, x2 to one
    # CFG step.  32 threads: more threads made the oracle slower on the 256-core host (363 s per full
    # forward with 256 threads in an earlier run).
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import torch_ref
        gold = torch.load(os.path.join(ROOT, "tests", "golden", "unet_t2v_full.pt"), map_location="cpu",
                          weights_only=False)
        sd = torch_ref.synth_state_dict(gold["shapes"], seed=0)
        cores = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(cores)
        gen = torch.Generator("cpu").manual_seed(8888)
        x = torch.randn(1, 4, 16, 32, 56, generator=gen)[:, :, :4].contiguous()
        yy = torch.randn(1, 77, 1024, generator=gen)
        with torch.no_grad():
            t1 = time.perf_counter()
            torch_ref.unet_forward(sd, x, torch.tensor([981]), yy, 320)
            cpu_s = time.perf_counter() - t1
        fwd_s = 4.0 * cpu_s
        res["cpu_baseline"] = {"value": round(1.0 / (2 * fwd_s), 5), "unit": "steps/s", "cores": cores,
                               "kind": "port",
                               "sample": "oracle (oracle/torch_ref.py, fp32) full-size UNet on a 4-frame latent "
                                         f"[1,4,4,32,56]: {cpu_s:.1f} s, x4 frames x2 CFG branches per step"}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
