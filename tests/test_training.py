"""Scope row f-3 (training-side operators): backward of alpha compositing and the hash-decay regulariser.
Fixtures `grad_composite_*` are autograd through the REFERENCE's compute_alpha_weights + volumetric_rendering."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden
from oracle import nlr_oracle as orc

T = torch.from_numpy
TAGS = ["opaque", "transparent"]


def _loss(r, w, g):
    return sum((r[k] * T(g["cot_" + k]).to(w.device)).sum() for k in ("rgb", "depth", "semantic", "intensity", "acc")) + \
        (w * T(g["cot_weights"]).to(w.device)).sum()


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_composite_gradients_match_reference(tag):
    g = golden(f"grad_composite_{tag}")
    dens, rgbs, sem, inten = (T(g[k]).clone().requires_grad_(True) for k in ("density", "rgbs", "sem", "intensity"))
    w = orc.compute_alpha_weights(dens, T(g["tdist"]), T(g["dirs"]), bool(g["opaque"]))
    r = orc.volumetric_rendering(rgbs, w, T(g["tdist"]), 1.0, torch.full((dens.shape[0], 1), 2.5), True, semantic=sem, intensity=inten)
    gd, gr, gs, gi = torch.autograd.grad(_loss(r, w, g), [dens, rgbs, sem, inten])
    for got, key in ((gd, "g_density"), (gr, "g_rgbs"), (gs, "g_sem"), (gi, "g_intensity")):
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-5, atol=1e-6, err_msg=key)
    # sem_detach: the semantic / intensity cotangents must not reach the density
    w2 = orc.compute_alpha_weights(dens, T(g["tdist"]), T(g["dirs"]), bool(g["opaque"]))
    r2 = orc.volumetric_rendering(rgbs, w2, T(g["tdist"]), 1.0, torch.full((dens.shape[0], 1), 2.5), False, semantic=sem, intensity=inten)
    (g0,) = torch.autograd.grad((r2["semantic"] * T(g["cot_semantic"])).sum() + (r2["intensity"] * T(g["cot_intensity"])).sum(), [dens],
                                allow_unused=True)
    assert g0 is None or float(g0.abs().max()) == 0.0


def test_oracle_hash_decay_is_segment_mean():
    rng = np.random.default_rng(0)
    off = np.array([0, 8, 24, 56, 120], np.int32)
    e = torch.from_numpy(rng.normal(size=(120, 2)).astype(np.float32))
    idx = np.repeat(np.arange(4), np.diff(off))
    want = np.mean([[float((e[idx == l, c] ** 2).mean()) for c in range(2)] for l in range(4)])  # segment_coo(..., 'mean').mean()
    assert abs(float(orc.hash_decay_loss(e, off)) - want) < 1e-6


# ---- GPU ----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_composite_forward_backward_match_reference(tag):
    from nerflidar_hip import training
    g = golden(f"grad_composite_{tag}")
    cu = lambda k: T(g[k]).cuda()
    dens, rgbs, sem, inten = (cu(k).clone().requires_grad_(True) for k in ("density", "rgbs", "sem", "intensity"))
    r = training.volumetric_render(dens, cu("tdist"), cu("dirs"), rgbs, sem, inten, opaque_background=bool(g["opaque"]), bg=1.0)
    for k in ("rgb", "depth", "semantic", "intensity", "acc"):
        np.testing.assert_allclose(r[k].detach().cpu().numpy(), g["out_" + k], rtol=2e-5, atol=5e-6, err_msg=k)
    np.testing.assert_allclose(r["weights"].detach().cpu().numpy(), g["weights"], rtol=1e-5, atol=2e-6)
    _loss(r, r["weights"], g).backward()
    for t, key in ((dens, "g_density"), (rgbs, "g_rgbs"), (sem, "g_sem"), (inten, "g_intensity")):
        got, want = t.grad.cpu().numpy(), g[key]
        scale = np.abs(want).max()
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-6 * max(scale, 1.0), err_msg=key)
    assert float(dens.grad[:, -1].abs().max()) == 0.0 if bool(g["opaque"]) else True  # opaque last interval: no gradient


@pytest.mark.gpu
def test_composite_backward_partial_cotangents_and_errors():
    from nerflidar_hip import training
    g = golden("grad_composite_opaque")
    cu = lambda k: T(g[k]).cuda()
    dens = cu("density").clone().requires_grad_(True)
    rgbs = cu("rgbs").clone().requires_grad_(True)
    r = training.volumetric_render(dens, cu("tdist"), cu("dirs"), rgbs)  # no semantic / intensity
    assert set(r) == {"rgb", "depth", "acc", "weights"}
    (r["depth"] * cu("cot_depth")).sum().backward()                      # only one output used
    d2 = T(g["density"]).clone().requires_grad_(True)
    w = orc.compute_alpha_weights(d2, T(g["tdist"]), T(g["dirs"]), True)
    ro = orc.volumetric_rendering(T(g["rgbs"]), w, T(g["tdist"]), 1.0, torch.full((d2.shape[0], 1), 2.5), False)
    (gd,) = torch.autograd.grad((ro["depth"] * T(g["cot_depth"])).sum(), [d2])
    np.testing.assert_allclose(dens.grad.cpu().numpy(), gd.numpy(), rtol=2e-4, atol=1e-6)
    assert float(rgbs.grad.abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        training.volumetric_render(T(g["density"]), T(g["tdist"]), T(g["dirs"]), T(g["rgbs"]))


@pytest.mark.gpu
def test_hash_decay_loss_and_gradient():
    from nerflidar_hip import training
    from nerflidar_hip.gridencoder import GridEncoder
    torch.manual_seed(0)
    encs = [GridEncoder(input_dim=3, num_levels=6, level_dim=c, base_resolution=16, desired_resolution=512, log2_hashmap_size=12).cuda()
            for c in (1, 4)]
    for e in encs:
        with torch.no_grad():
            e.embeddings.normal_(0, 0.3)
    loss = training.hash_decay_loss(encs, mult=0.7)
    loss.backward()
    want = 0.0
    refs = []
    for e in encs:
        p = e.embeddings.detach().cpu().clone().requires_grad_(True)
        refs.append(p)
        want = want + orc.hash_decay_loss(p, e._offsets_host.numpy(), 0.7)
    want.backward()
    assert abs(float(loss.detach()) - float(want.detach())) <= 1e-6 * max(1.0, abs(float(want.detach())))
    for e, p in zip(encs, refs):
        np.testing.assert_allclose(e.embeddings.grad.cpu().numpy(), p.grad.numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
def test_training_slice_decreases_loss():
    """GridEncoder (HIP fwd/bwd) -> torch MLP -> compositing (HIP fwd/bwd) + hash decay: a few Adam steps on a toy depth target."""
    from nerflidar_hip import training
    from nerflidar_hip.gridencoder import GridEncoder
    torch.manual_seed(1)
    N, S = 256, 32
    enc = GridEncoder(input_dim=3, num_levels=8, level_dim=2, base_resolution=16, desired_resolution=256, log2_hashmap_size=14).cuda()
    mlp = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.Linear(64, 4)).cuda()
    opt = torch.optim.Adam(list(enc.parameters()) + list(mlp.parameters()), lr=2e-2)
    o = torch.zeros(N, 3, device="cuda")
    d = torch.nn.functional.normalize(torch.randn(N, 3, device="cuda"), dim=-1)
    tdist = torch.linspace(0.05, 1.0, S + 1, device="cuda")[None].repeat(N, 1)
    pts = o[:, None] + 0.5 * (tdist[:, 1:] + tdist[:, :-1])[..., None] * d[:, None]
    target = torch.full((N,), 0.6, device="cuda")
    losses = []
    for _ in range(30):
        h = mlp(enc(pts.reshape(-1, 3), bound=1)).reshape(N, S, 4)
        r = training.volumetric_render(torch.nn.functional.softplus(h[..., 0] + 1), tdist, d, torch.sigmoid(h[..., 1:]))
        loss = ((r["depth"] - target) ** 2).mean() + training.hash_decay_loss([enc], 0.1)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.2 * losses[0], losses[::6]


@pytest.mark.gpu
def test_trainable_level_matches_oracle_forward_and_gradients():
    """cast/contract op + HIP grid fwd/bwd + torch Linear stack + HIP compositing fwd/bwd against the oracle (torch CPU autograd
    for the Linear parameters, the C grid oracle's backward for the table)."""
    from nerflidar_hip import config as nconfig, lidar as nlidar, training, weights as nweights
    mc = nconfig.workload("C2", 12)
    sd = nweights.synth_state_dict(mc, seed=5, trained_like=True)
    b = nlidar.synthetic_sweep(width=4, seed=5, beams=nlidar.LIDAR_ANGLES[::8])
    N, S = b["origins"].shape[0], 32
    rng = np.random.default_rng(0)
    tdist = np.sort(rng.uniform(0.01, 1.5, (N, S + 1)).astype(np.float32), axis=-1)
    cot = {k: rng.normal(size=sh).astype(np.float32) for k, sh in dict(rgb=(N, 3), depth=(N,), semantic=(N, 19), intensity=(N,)).items()}
    lvl = training.TrainableNerfLevel(mc.nerf_mlp).load_reference(sd).cuda()
    batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    store, orig = {}, training.encode_features

    def keep_grad(*a, **k):
        f = orig(*a, **k)
        f.retain_grad()
        store["f"] = f
        return f
    training.encode_features = keep_grad
    try:
        r, o = lvl.render(batch, torch.from_numpy(tdist).cuda())
        loss = sum((r[k] * torch.from_numpy(cot[k]).cuda()).sum() for k in cot)
        loss.backward()
    finally:
        training.encode_features = orig
    # oracle: same level with torch CPU autograd on the Linear parameters
    enc = orc.make_encoders(sd, mc)["nerf_mlp"]
    sd_t = {k: v.clone().requires_grad_(True) for k, v in orc.to_torch_sd(sd).items() if k.startswith("nerf_mlp.")}
    bt = {k: torch.from_numpy(v) for k, v in b.items()}
    means, stds = orc.cast_rays(torch.from_numpy(tdist), bt["origins"], bt["directions"], bt["radii"], bt["base_x"], bt["base_y"], n=7, m=3,
                                std_scale=0.35)
    res = orc.mlp_forward(sd_t, "nerf_mlp", mc.nerf_mlp, enc, means, stds, bt["viewdirs"])
    w = orc.compute_alpha_weights(res["density"], torch.from_numpy(tdist), bt["directions"], True)
    ro = orc.volumetric_rendering(res["rgb"], w, torch.from_numpy(tdist), 1.0, bt["far"], False, semantic=res["semantic"], intensity=res["intensity"])
    lo = sum((ro[k] * torch.from_numpy(cot[k])).sum() for k in cot)
    lo.backward()
    for k in ("rgb", "depth", "semantic", "intensity"):
        np.testing.assert_allclose(r[k].detach().cpu().numpy(), ro[k].detach().numpy(), rtol=2e-3, atol=2e-4, err_msg=k)
    for name, p in lvl.named_parameters():
        if name == "encoder.embeddings":
            continue
        want = sd_t["nerf_mlp." + name].grad.numpy()
        got = p.grad.cpu().numpy()
        scale = max(np.abs(want).max(), 1e-6)
        assert np.abs(got - want).max() <= 5e-3 * scale, (name, np.abs(got - want).max(), scale)
    g_tab = lvl.encoder.embeddings.grad.cpu().numpy()
    assert np.isfinite(g_tab).all() and np.abs(g_tab).max() > 0
    # table gradient: dL/dfeatures pushed through (mean over 7, erf weights) and the C oracle of kernel_grid_backward
    cfg = mc.nerf_mlp
    L, Cc = cfg.grid_num_levels, cfg.grid_level_dim
    m_h, s_h = training.cast_contract(batch, torch.from_numpy(tdist).cuda())
    gs = torch.from_numpy(sd["nerf_mlp.encoder.grid_sizes"]).float()
    w = torch.erf(1 / torch.clamp(torch.sqrt(8 * s_h.cpu()[..., None] ** 2 * gs ** 2), min=1e-10))          # [N,S,7,L]
    g_raw = store["f"].grad.cpu().reshape(N, S, 1, L, Cc) * w[..., None] / 7.0                                 # [N,S,7,L,C]
    gl = g_raw.reshape(-1, L, Cc).permute(1, 0, 2).contiguous().numpy()
    x01 = ((m_h.cpu().reshape(-1, 3) + 1) / 2).numpy()
    from nerflidar_hip.weights import grid_layout
    offsets, _, pls = grid_layout(cfg)
    want_tab, _ = orc.grid_backward_c(gl, x01, offsets, g_tab.shape[0], Cc, float(np.log2(pls)), cfg.grid_base_resolution)
    scale = np.abs(want_tab).max()
    assert np.abs(g_tab - want_tab).max() <= 2e-3 * scale, (np.abs(g_tab - want_tab).max(), scale)
    # means/stds operator against the oracle's cast_rays + contraction
    m_hip, s_hip = training.cast_contract(batch, torch.from_numpy(tdist).cuda())
    cm, cs = orc.contract_mean_std(means.reshape(-1, 3), stds.reshape(-1))
    cm, cs = cm.reshape(means.shape), cs.reshape(stds.shape)
    np.testing.assert_allclose(m_hip.cpu().numpy(), (cm / 2).numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(s_hip.cpu().numpy(), (cs / 2).numpy(), rtol=2e-5, atol=1e-9)
