"""Known-answer vectors for the latent-consistency multistep sampler, derived BY HAND from the published algorithm —
pure-Python float64, no torch, no vgen_amd, no oracle import — so that vgen_amd/lcm.py (a restatement of the un-vendored
`diffusers.LCMScheduler`) is pinned to something outside this repository's own code.

Sources of every formula below:
  [LCM]  Luo et al. 2023, "Latent Consistency Models", eq. (9) + Appendix: f(x, t) = c_skip(t) x + c_out(t) F(x, t) with
         c_skip(t) = sd^2 / ((s t)^2 + sd^2), c_out(t) = s t / sqrt((s t)^2 + sd^2), sd = 0.5, timestep scaling s = 10;
         Algorithm 3 (multistep consistency sampling): x <- sqrt(abar_prev) f(x, t) + sqrt(1 - abar_prev) z.
  [SD]   the "scaled linear" beta schedule of latent diffusion: beta_i = (sqrt(b0) + i / (T - 1) (sqrt(b1) - sqrt(b0)))^2,
         b0 = 0.00085, b1 = 0.012, T = 1000 (the constructor arguments of inference_videolcm_entrance.py:171).
  [ZSNR] Lin et al. 2023, "Common Diffusion Noise Schedules and Sample Steps are Flawed", Algorithm 1: shift and scale
         sqrt(abar) so that sqrt(abar_T) = 0 and sqrt(abar_0) is unchanged.
  [v]    Salimans & Ho 2022, v-parameterisation: x0 = sqrt(abar_t) x - sqrt(1 - abar_t) v.
  schedule: the LCM "skipping step" grid t_k = k' (T / N0) - 1, k' = 1..N0 (N0 = 50 training-time DDIM steps -> 19, 39, ...,
         999), taken in descending order at indices floor(i N0 / n), i = 0..n-1 (n = 4 -> indices 0, 12, 25, 37).

    python tests/golden/make_lcm_kat.py     # rewrites tests/golden/lcm_kat.json
"""
import json
import math
import os

T, N0, B0, B1, SD, S = 1000, 50, 0.00085, 0.012, 0.5, 10.0


def abar_table():
    beta = [(math.sqrt(B0) + i / (T - 1) * (math.sqrt(B1) - math.sqrt(B0))) ** 2 for i in range(T)]
    ab, p = [], 1.0
    for b in beta:
        p *= 1.0 - b
        ab.append(p)
    rt = [math.sqrt(a) for a in ab]                      # [ZSNR] Algorithm 1
    r0, rT = rt[0], rt[-1]
    rt = [(r - rT) * r0 / (r0 - rT) for r in rt]
    return [r * r for r in rt]


def schedule(n):
    grid = [k * (T // N0) - 1 for k in range(1, N0 + 1)][::-1]
    return [grid[int(math.floor(i * N0 / n))] for i in range(n)]


def boundary(t):
    st = S * t
    return SD ** 2 / (st ** 2 + SD ** 2), st / math.sqrt(st ** 2 + SD ** 2)


def step(ab, x, v, t, t_prev, z):
    a = ab[t]
    c_skip, c_out = boundary(t)
    out_den, out_prev = [], []
    for xi, vi, zi in zip(x, v, z):
        x0 = math.sqrt(a) * xi - math.sqrt(1.0 - a) * vi                     # [v]
        den = c_out * x0 + c_skip * xi                                       # [LCM] eq. (9)
        out_den.append(den)
        out_prev.append(den if t_prev is None else math.sqrt(ab[t_prev]) * den + math.sqrt(1.0 - ab[t_prev]) * zi)
    return out_den, out_prev


def main():
    ab = abar_table()
    ts = schedule(4)
    assert ts == [999, 759, 499, 259] and schedule(2) == [999, 499] and schedule(8)[:3] == [999, 879, 759]
    x = [1.0, -2.0, 0.5, 3.0]
    v = [0.25, -0.5, 1.0, 2.0]
    z = [0.1, -0.2, 0.3, 0.4]
    kat = {"timesteps_4": ts, "timesteps_2": schedule(2), "timesteps_8": schedule(8),
           "alphas_cumprod": {str(t): ab[t] for t in (0, 259, 499, 759, 998, 999)},
           "boundary": {str(t): list(boundary(t)) for t in ts},
           "x": x, "v": v, "z": z, "steps": []}
    cur = x
    for i, t in enumerate(ts):                           # the whole 4-step loop with a FIXED model output v and noise z
        t_prev = ts[i + 1] if i + 1 < len(ts) else None
        den, prev = step(ab, cur, v, t, t_prev, z)
        kat["steps"].append({"t": t, "denoised": den, "prev_sample": prev})
        cur = prev
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lcm_kat.json")
    json.dump(kat, open(path, "w"), indent=1)
    print(json.dumps(kat, indent=1)[:1500])


if __name__ == "__main__":
    main()
