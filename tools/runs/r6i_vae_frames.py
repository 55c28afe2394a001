"""r06 call I: does the calibrated VAE's decode parity improve with more calibration frames / other damping?
(tests/golden/vae_sd_full.pt, the fixture of tests/test_gpu_model.py::test_vae_full_size_calibrated)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import gold, rel_l2
from oracle import torch_ref
from vgen_amd import calibrate as cal, ops
from vgen_amd.vae import AutoencoderKL
ops.set_backend(None)
DEV = "cuda:0"
g, g2 = gold("vae_sd_full.pt"), gold("vae_sd_full2.pt")
sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
z = torch.randn(1, 4, 32, 56, generator=torch.Generator("cpu").manual_seed(g["input_seed"])).to(DEV)
img = torch.randn(1, 3, 256, 448, generator=torch.Generator("cpu").manual_seed(g2["enc256_seed"])).clamp(-1, 1).to(DEV)
out = {}
for n, damp in ((8, 0.01), (16, 0.01), (32, 0.01), (16, 0.003), (16, 0.03)):
    v = AutoencoderKL(ddconfig=g["ddconfig"], embed_dim=4, compute_dtype="fp16", precision="high").eval()
    v.load_state_dict(sd, strict=True)
    v = v.to(DEV)
    gen = torch.Generator("cpu").manual_seed(424244)
    zc = torch.randn(n, 4, 32, 56, generator=gen).to(DEV)
    xc = torch.randn(n, 3, 256, 448, generator=gen).clamp(-1, 1).to(DEV)
    t0 = time.time()
    rep = cal.calibrate_vae(v, zc, xc, damp=damp)
    dt = time.time() - t0
    e_dec = rel_l2(v.decode(z)[:, :, ::4, ::4], g["dec_sub"])
    e_enc = rel_l2(v.encode(img).parameters, g2["enc256_moments"])
    out[f"n={n},damp={damp}"] = dict(decode=e_dec, encode=e_enc, seconds=round(dt, 1), calibrated=rep["calibrated"])
    print(n, damp, f"decode {e_dec:.4e} encode {e_enc:.4e} {dt:.0f}s", flush=True)
    del v
    torch.cuda.empty_cache()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r6i_vae_frames.json"), "w"), indent=1)
