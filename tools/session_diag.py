"""GPU diagnostic: where does the session path differ from the step-by-step path? (tiny UNet)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import gold
from oracle import torch_ref
from vgen_amd import ops
from vgen_amd.unet import UNetSD_T2VBase
from vgen_amd.diffusion import DiffusionDDIM
from vgen_amd.session import UnitSession

DEV = "cuda:0"
g = gold("unet_tiny.pt")
sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
m = UNetSD_T2VBase(**g["cfg"], compute_dtype="bf16", precision="fast").eval()
m.load_state_dict(sd, strict=True)
m = m.to(DEV)
x, y = g["x"].to(DEV), g["y"].to(DEV)
kw = [dict(y=y), dict(y=torch.roll(y, 1, 1))]
t = torch.full((x.shape[0],), 601, dtype=torch.long, device=DEV)
m.pack()
tab = m.time_embedding_table(1000)
e = m._embed(t.repeat(2), None, 4, DEV)
print("emb table row == per-step row:", torch.equal(tab[t.repeat(2)], e), float((tab[t.repeat(2)] - e).abs().max()))
a = m.forward_units(x, t, kw)
b = m.forward_units(x, t, kw)
print("forward_units reproducible:", all(torch.equal(p, q) for p, q in zip(a, b)))
os.environ["VGEN_GRAPH"] = "1"
s = UnitSession(m, tuple(x.shape), DEV, kw, torch.long, 1000)
for i in range(4):
    o = s.eval(x, t)
    print(f"session.eval call {i}: == forward_units:", [bool(torch.equal(p, q)) for p, q in zip(o, a)],
          [float((p - q).abs().max()) for p, q in zip(o, a)])
# body with the per-step emb / kv vs session's
prep = m._prepare_units(tuple(x.shape), x.device, kw)
kv = m._context_kv(prep["ctx"], DEV)
print("kv equal:", torch.equal(kv, s.kv))
xs = x.float().repeat(2, 1, 1, 1, 1)
print("x_units equal:", torch.equal(xs, s.x_units))
o1 = m._body(xs, e, kv, prep["ctx"].shape[1], False)
o2 = m._body(s.x_units, tab[t.repeat(2)].contiguous(), s.kv, s.Lctx, False)
o3 = m._body(xs, e, kv, prep["ctx"].shape[1], False)
print("body(per-step inputs) vs body(session inputs):", torch.equal(o1, o2), " body reproducible:", torch.equal(o1, o3))
print("body vs forward_units:", torch.equal(o1[:2], a[0]), torch.equal(o1[2:], a[1]))
d0 = DiffusionDDIM(schedule="cosine", schedule_param=dict(num_timesteps=1000, cosine_s=0.008, zero_terminal_snr=True), mean_type="v", var_type="fixed_small")
ct = d0._coef_table(DEV, "ddim", 20, 0.0)
cr = d0._coef_rows(d0._table(DEV), t, "ddim", 20, 0.0)
print("coef table rows == per-step rows:", torch.equal(ct[t], cr), (ct[t] - cr).abs().max().item())
