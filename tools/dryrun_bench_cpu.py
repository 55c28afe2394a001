"""TEST TOOLING (imports oracle/): runs bench.py's main() END TO END on the CPU, full-size t2v model, with the ABI emulator as
the op backend — a dry run of the script's host logic (headline measurement, in-run parity, roofline pass, the calibrated
model's pack-time calibration, the JSON line; VGEN_DRYRUN_TINY=1: the same on the tiny golden model in about a minute) for code
paths whose first GPU execution is the driver's.  Nothing here is a measurement: the "times" are the emulator's CPU seconds
and the fake launch events below; the parity values ARE meaningful (the emulator is within 1 % of the GPU on every fixture
both have run).  ~20 minutes on 8 cores.

    python tools/dryrun_bench_cpu.py [extra bench.py flags]        # -> the JSON line on stdout
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

os.environ["VGEN_BENCH_DEVICE"] = "cpu"
os.environ["VGEN_GRAPH"] = "0"
from oracle.abi_emulator import EmuBackend  # noqa: E402
from vgen_amd import ops  # noqa: E402
import bench  # noqa: E402


class _FakeEvent:
    def __init__(self, t):
        self.t = t

    def elapsed_time(self, other):
        return max(1e3 * (other.t - self.t), 1e-3)


class DryBackend(EmuBackend):
    """The emulator under the HIP backend's name, leaving the launch records the HIP backend leaves (ops.KERNEL_PROFILE)."""
    name = "hip"

    def _rec(self, name, work, meta, extra, t0):
        if ops.KERNEL_PROFILE is not None:
            ops.KERNEL_PROFILE.append((name, _FakeEvent(t0), _FakeEvent(time.perf_counter()), work, meta, extra))

    def tapgemm(self, g):
        t0 = time.perf_counter()
        out = super().tapgemm(g)
        dw = getattr(g.W, "vgen_dw", None)
        K = g.taps * g.C1 + g.C2
        meta = (g.mode, g.M, g.N, K, g.epilogue, str(g.out_dtype) + ("+dw" if dw is not None else ""))
        self._rec("tapgemm", 2.0 * g.M * g.N * (g.alg_k or K), meta,
                  (2.0 * g.M * (g.C1 + g.C2) + 2.0 * g.N * K, 2.0 * g.M * g.N * K * (2 if dw is not None else 1)), t0)
        return out

    def groupnorm(self, x1, x2, nb, S, groups, *a, **k):
        t0 = time.perf_counter()
        out = super().groupnorm(x1, x2, nb, S, groups, *a, **k)
        C = x1.shape[1] + (0 if x2 is None else x2.shape[1])
        self._rec("groupnorm", nb * S * C * 10, (nb, S, C, 0), None, t0)
        return out

    def layernorm(self, x, *a, **k):
        t0 = time.perf_counter()
        out = super().layernorm(x, *a, **k)
        self._rec("layernorm", x.numel() * 6, tuple(x.shape), None, t0)
        return out


def main():
    be = DryBackend()
    ops.set_backend(be)
    real_set = ops.set_backend
    ops.set_backend = lambda b=None: real_set(be if b is None else b)   # bench.py asks for the HIP backend (None): the dry one
    for fn in ("set_device", "synchronize", "empty_cache", "_sleep"):
        setattr(torch.cuda, fn, lambda *a, **k: None)
    if os.environ.get("VGEN_DRYRUN_TINY") == "1":
        # host-logic smoke of main() in a minute: the tiny golden model in place of the 1411 M one (no full-size parity)
        tiny = os.path.join(ROOT, "tests", "golden", "unet_tiny.pt")
        g = torch.load(tiny, map_location="cpu", weights_only=False)
        bench.GOLDEN_T2V = tiny
        bench.CONFIGS["t2v"] = dict(bench.CONFIGS["t2v"], cfg=dict(g["cfg"]), latent=tuple(g["x"].shape[1:]))
        real_cond = bench.conditioning

        def cond(name, model, P, dev, gen):
            kw = real_cond(name, model, P, dev, gen)
            for d in kw:
                d["y"] = torch.randn(P, g["y"].shape[1], g["y"].shape[2], generator=gen, device=dev)
            return kw
        bench.conditioning = cond
        from vgen_amd import calibrate as cal
        real_batch = cal.calibration_batch
        cal.calibration_batch = lambda shape, n=None, **k: real_batch(shape, n=4, **dict(k, context=tuple(g["y"].shape[1:])))
        sys.argv = [sys.argv[0], "--steps", "2", "--warmup", "1", "--no-vae", "--no-cpu-baseline", "--no-parity",
                    "--variants", "fp16/mixed,fp16/fast"] + sys.argv[1:]
    else:
        sys.argv = [sys.argv[0], "--steps", "1", "--warmup", "0", "--no-vae", "--no-cpu-baseline", "--no-scaling-model",
                    "--variants", "fp16/mixed"] + sys.argv[1:]
    bench.main()


if __name__ == "__main__":
    main()
