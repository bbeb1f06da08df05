"""Scope row f-3 (training-side operators): backward of alpha compositing and the hash-decay regulariser.
Fixtures `grad_composite_*` are autograd through the REFERENCE's compute_alpha_weights + volumetric_rendering."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden
from oracle import nlr_oracle as orc
from nerflidar_hip import training as ntrain

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


def _bf16_ste(t):
    """round to bf16 in value, identity in gradient: what a bf16 MFMA operand is to the f32 chain around it"""
    return t + (t.to(torch.bfloat16).float() - t).detach()


def _level_forward_bf16_operands(lvl, batch, tdist):
    """TrainableNerfLevel.forward with every Linear's input and weight rounded to bf16 (f32 accumulation, f32 bias): the arithmetic
    of the fused kernels, written with torch ops so that torch autograd provides the gradients to compare with."""
    from nerflidar_hip import training
    from nerflidar_hip.objects import _pos_enc
    F = torch.nn.functional
    cfg = lvl.cfg
    lin = lambda m, x: F.linear(_bf16_ste(x), _bf16_ste(m.weight), m.bias)
    means, stds = training.cast_contract(batch, tdist)
    f = training.encode_features(lvl.encoder, means, stds, cfg.re_weights)
    x = lin(lvl.density_layer[2], F.relu(lin(lvl.density_layer[0], f)))
    out = {"density": F.softplus(x[..., 0] + cfg.density_bias)}
    if cfg.use_semantic:
        out["semantic"] = torch.softmax(lin(lvl.sem_layer[2], F.relu(lin(lvl.sem_layer[0], x))), -1)
    if cfg.use_intensity:
        out["intensity"] = lin(lvl.intensity_layer[2], F.relu(lin(lvl.intensity_layer[0], x)))[..., 0]
    enc = _pos_enc(batch["viewdirs"].reshape(x.shape[0], 3).float(), cfg.deg_view)
    h = torch.cat([x, enc[:, None, :].expand(-1, x.shape[1], -1)], dim=-1)
    inputs = h
    for i in range(cfg.net_depth_viewdirs):
        h = F.relu(lin(getattr(lvl, f"lin_second_stage_{i}"), h))
        if i == cfg.skip_layer_dir:
            h = torch.cat([h, inputs], dim=-1)
    rgb = torch.sigmoid(cfg.rgb_premultiplier * lin(lvl.rgb_layer, h) + cfg.rgb_bias)
    out["rgb"] = rgb * (1 + 2 * cfg.rgb_padding) - cfg.rgb_padding
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("wl,S", [("C2", 32), ("REF", 32), ("P_W128I", 64), ("P_NOSEM", 32), ("P_D3", 32)])
def test_fused_training_mlp_matches_torch_autograd(wl, S):
    """f-3: nlr_mlp_train_forward / nlr_mlp_train_backward (bf16 MFMA chains, f32 accumulation) against torch autograd through the
    SAME level's Linear stack evaluated with bf16-rounded operands (the kernels' arithmetic; against an f32 forward a bf16
    forward flips ~1 % of the ReLU masks, which alone is a 10 % gradient difference): outputs, the gradient of every Linear
    weight and bias, and the gradient that reaches the hash table through the feature gradient.  What remains is the bf16
    rounding of the activation gradients between the layers of the backward chain: 2e-2 of each tensor's norm."""
    from nerflidar_hip import config as nconfig, lidar as nlidar, training, weights as nweights
    mc = nconfig.workload(wl, 12)
    sd = nweights.synth_state_dict(mc, seed=5, trained_like=False)
    # a livelier random scene than the reference init, without the x1500 density gain of trained_like (bf16 operands)
    for k in sd:
        if k.endswith("encoder.embeddings"):
            sd[k] = (sd[k] * 3e3).astype(np.float32)
    b = nlidar.synthetic_sweep(width=6, seed=5, beams=nlidar.LIDAR_ANGLES[::8])
    N = b["origins"].shape[0]
    rng = np.random.default_rng(0)
    tdist = torch.from_numpy(np.sort(rng.uniform(0.01, 1.5, (N, S + 1)).astype(np.float32), axis=-1)).cuda()
    batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    cfg = mc.nerf_mlp
    ref = training.TrainableNerfLevel(cfg).load_reference(sd).cuda()
    fus = training.TrainableNerfLevel(cfg, fused_mlp=True).load_reference(sd).cuda()
    cot = {"density": rng.normal(size=(N, S)), "rgb": rng.normal(size=(N, S, 3))}
    if cfg.use_semantic:
        cot["semantic"] = rng.normal(size=(N, S, cfg.class_num))
    if cfg.use_intensity:
        cot["intensity"] = rng.normal(size=(N, S))
    cot = {k: torch.from_numpy(v.astype(np.float32)).cuda() for k, v in cot.items()}
    fus._keep_debug = True
    outs = {}
    for name, lvl in (("ref", ref), ("fus", fus)):
        o = _level_forward_bf16_operands(lvl, batch, tdist) if name == "ref" else lvl(batch, tdist)
        sum((o[k] * cot[k]).sum() for k in cot).backward()
        outs[name] = {k: v.detach().float().cpu().numpy() for k, v in o.items()}

    def close(name, got, want, rel):
        err = float(np.linalg.norm(got - want)) / max(float(np.linalg.norm(want)), 1e-30)
        assert err <= rel, f"{wl} {name}: relative norm error {err:.3e} > {rel}"

    # (1) forward: the kernels' arithmetic written in torch
    for k in cot:
        close(k, outs["fus"][k], outs["ref"][k], 2e-3)

    # (2) backward kernel, layer by layer, against the chain rule in float64 ON THE ACTIVATIONS THE FORWARD KERNEL SAVED (same ReLU
    # masks by construction) with bf16-rounded weights and the bf16 rounding of each gradient tile between layers that the kernel
    # applies: what is left is f32 accumulation order and the rounding of the stored value itself
    torch.set_grad_enabled(False)
    d = fus._dbg
    M = N * S
    W, WB, D, K = cfg.net_width_viewdirs, cfg.bottleneck_width, cfg.net_depth_viewdirs, (cfg.class_num if cfg.use_semantic else 0)
    HH = (64 if K else 0) + (64 if cfg.use_intensity else 0)
    c_hid, c_hbe, c_q, c_x, aw = 0, 64, 64 + WB, 64 + WB + HH, 64 + WB + HH + D * W
    A = d["acts"].double().cpu()
    G = d["gacts"].double().cpu()
    r16 = lambda t: t.float().to(torch.bfloat16).double()
    wt = lambda m: r16(m.weight.detach().cpu())
    p_ = cfg.rgb_padding
    g_rgb = cot["rgb"].reshape(M, 3).double().cpu()
    sg = (d["rgb"].double().cpu().t() + p_) / (1 + 2 * p_)
    d_o = g_rgb * (1 + 2 * p_) * sg * (1 - sg) * cfg.rgb_premultiplier
    want = {}
    dz = (A[:, c_x + (D - 1) * W:c_x + D * W] > 0) * (r16(d_o) @ wt(fus.rgb_layer))
    want[c_x + (D - 1) * W] = dz
    for l in range(D - 1, 1, -1):
        dz = (A[:, c_x + (l - 1) * W:c_x + l * W] > 0) * (r16(dz) @ wt(getattr(fus, f"lin_second_stage_{l}")))
        want[c_x + (l - 1) * W] = dz
    dz1 = r16(dz)
    W1, W0 = wt(fus.lin_second_stage_1), wt(fus.lin_second_stage_0)
    dz0 = (A[:, c_x:c_x + W] > 0) * (dz1 @ W1[:, :W])
    want[c_x] = dz0
    dhbe = dz1 @ W1[:, W:W + WB] + r16(dz0) @ W0[:, :WB]
    if HH:
        dlo = torch.zeros(M, 32, dtype=torch.float64)
        if K:
            pr = d["sem"].double().cpu().t()
            gs = cot["semantic"].reshape(M, K).double().cpu()
            dlo[:, :K] = pr * (gs - (pr * gs).sum(-1, keepdim=True))
        if cfg.use_intensity:
            dlo[:, K] = cot["intensity"].reshape(M).double().cpu()
        np.testing.assert_allclose(G[:, aw:aw + 32].numpy(), r16(dlo).numpy(), rtol=1e-2, atol=1e-6 * float(dlo.abs().max()))
        h1 = torch.cat(([wt(fus.sem_layer[0])] if K else []) + ([wt(fus.intensity_layer[0])] if cfg.use_intensity else []), 0)
        h2 = torch.zeros(32, HH, dtype=torch.float64)
        r0 = 0
        if K:
            h2[:K, :64] = wt(fus.sem_layer[2])
            r0 = 64
        if cfg.use_intensity:
            h2[K, r0:r0 + 64] = wt(fus.intensity_layer[2])[0]
        dq = (A[:, c_q:c_q + HH] > 0) * (r16(dlo) @ h2)
        want[c_q] = dq
        dhbe = dhbe + r16(dq) @ h1
    dr = cot["density"].reshape(M).double().cpu() * (1 - torch.exp(-d["density"].double().cpu()))
    dhbe[:, 0] += dr
    want[c_hbe] = dhbe
    dhid = (A[:, c_hid:c_hid + 64] > 0) * (r16(dhbe) @ wt(fus.density_layer[2]))
    want[c_hid] = dhid
    dfeat = r16(dhid) @ wt(fus.density_layer[0])
    for c0, w in want.items():
        close(f"gacts[{c0}:{c0 + w.shape[1]}]", G[:, c0:c0 + w.shape[1]].numpy(), w.numpy(), 4e-3)
    close("d_feat", d["d_feat"].double().cpu().numpy(), dfeat.numpy(), 4e-3)
    torch.set_grad_enabled(True)

    # (3) end to end against autograd through the bf16-operand torch model: on top of (2), the two forwards differ in f32
    # accumulation order, a few activations round to the other bf16 neighbour, and ~3e-4 of the ReLU masks flip
    for (name, p), (_, pf) in zip(ref.named_parameters(), fus.named_parameters()):
        if p.grad is None:  # a layer the configuration builds but does not read (sem_layer without use_semantic)
            assert pf.grad is None, name
            continue
        assert pf.grad is not None, name
        close("grad " + name, pf.grad.float().cpu().numpy(), p.grad.float().cpu().numpy(), 4e-2)
    # and it trains: a few Adam steps on the fused level lower a depth + rgb loss
    opt = torch.optim.Adam(fus.parameters(), lr=2e-3)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        r, _ = fus.render(batch, tdist)
        loss = ((r["depth"] - 0.7) ** 2).mean() + ((r["rgb"] - 0.25) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses


# ---- the whole model: forward consistency with the fused inference path, and the training step of train.py:272-459 ------------------
def _ref_scene(log2_hashmap=12, width=64, seed=0):
    from nerflidar_hip import config as nconfig, lidar as nlidar, weights as nweights
    mc = nconfig.workload("REF", log2_hashmap)
    sd = nweights.synth_state_dict(mc, seed=seed, trained_like=True)
    b = nlidar.synthetic_sweep(width=width, seed=seed)
    return mc, sd, {k: torch.from_numpy(v).cuda() for k, v in b.items()}


@pytest.mark.gpu
def test_trainable_model_computes_what_the_fused_inference_path_computes():
    """`TrainableModel.forward` (autograd graph over the HIP operators) and `Model.forward` (the fused inference path, pinned on the
    reference's fixtures) on the same weights and rays: same sample positions, same weights, same renderings."""
    from nerflidar_hip.models import Model
    mc, sd, batch = _ref_scene()
    tm = ntrain.TrainableModel(mc).cuda().load_reference(sd)
    with torch.no_grad():
        rend, hist = tm(batch, train_frac=1.0)
    ref_r, ref_h = Model(mc, sd, precision=0)(False, batch, train_frac=1.0, compute_extras=True)
    for li in range(3):
        # level 0 resamples the same [0, 1] interval: identical positions; later levels invert a CDF built from the previous
        # level's weights, which differ by the summation order of the density MLP (torch GEMM here, fused kernel there)
        # (where the previous level's CDF is flat a 1e-6 change of a weight moves a sample by 1e-4: a handful of positions)
        ds = (hist[li]["sdist"] - ref_h[li]["sdist"]).abs()
        if li == 0:
            assert float(ds.max()) == 0.0
        else:
            assert float(ds.mean()) <= 1e-6 and float((ds > 1e-4).float().mean()) <= 1e-3 and float(ds.max()) <= 5e-3, (li, float(ds.mean()), float(ds.max()))
        dw = (hist[li]["weights"] - ref_h[li]["weights"]).abs()
        # (a surface sample that moved takes its weight along: rare, large, and invisible in the composited outputs below)
        assert float(dw.mean()) <= 2e-5 and float((dw > 1e-3).float().mean()) <= 1e-3, (li, float(dw.mean()), float(dw.max()))
        assert float((rend[li]["depth"] - ref_r[li]["depth"]).abs().mean()) <= 1e-4
    for k in ("rgb", "depth", "semantic", "acc"):   # the two paths agree to summation-order noise, amplified on a few edge rays
        d = (rend[-1][k] - ref_r[-1][k]).abs()
        assert float(d.mean()) <= 2e-4 and float((d > 1e-3).float().mean()) <= 2e-2 and float(d.max()) <= 1e-1, (k, float(d.mean()), float(d.max()))
    assert float((rend[-1]["semantic"].argmax(-1) == ref_r[-1]["semantic"].argmax(-1)).float().mean()) >= 0.995
    # and the trained parameters go back into the inference path unchanged
    back = tm.reference_state_dict()
    for k, v in sd.items():
        if not k.endswith(("encoder.offsets", "encoder.grid_sizes", "encoder.idx")):
            np.testing.assert_array_equal(back[k], np.asarray(v, np.float32), err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_training_step_of_the_whole_model(fused):
    """train.py:272-459 on the fused path: a student with the reference's init learns a teacher's sweep (colour, depth, labels);
    every loss term is finite, the interlevel term reaches the proposal networks, hash tables of all three levels get gradients,
    and the loss goes down."""
    from nerflidar_hip.models import Model
    from nerflidar_hip import weights as nweights
    mc, sd, batch = _ref_scene(width=32)
    with torch.no_grad():
        teach = Model(mc, sd, precision=0).render_rays(batch)[0]
    batch = dict(batch, rgb=teach["rgb"].clone(), depth=teach["depth"].clone(), semantic=teach["semantic"].argmax(-1))
    torch.manual_seed(0)
    tm = ntrain.TrainableModel(mc, fused_mlp=fused).cuda()
    tm.load_reference(nweights.synth_state_dict(mc, seed=5, trained_like=True))   # a different scene: something to unlearn
    opt = torch.optim.Adam(tm.parameters(), lr=2e-3, eps=1e-15)
    kw = dict(depth_lam=0.4, sem_lam=0.04, interlevel_mult=0.0, anti_interlevel_mult=0.01, distortion_mult=0.005)
    first = ntrain.training_step(tm, opt, batch, randomized=True, tv_weight=1e-7, grad_max_norm=1.0, **kw)
    assert set(first) >= {"data", "depth", "sem", "interlevel", "distortion", "hash_decay", "loss"}
    assert all(np.isfinite(v) for v in first.values()), first
    # gradients of that first step reached every trainable part (checked on a fresh backward, the step cleared nothing)
    for name in ("prop_mlp_0.density_layer.0.weight", "prop_mlp_1.encoder.embeddings", "nerf_mlp.encoder.embeddings",
                 "nerf_mlp.lin_second_stage_1.weight", "nerf_mlp.sem_layer.2.weight"):
        g = dict(tm.named_parameters())[name].grad
        assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0, name
    hist = [first["loss"]]
    for _ in range(40):
        hist.append(ntrain.training_step(tm, opt, batch, randomized=True, grad_max_norm=1.0, **kw)["loss"])
    assert np.isfinite(hist).all()
    assert np.mean(hist[-5:]) < 0.7 * np.mean(hist[:3]), (hist[:3], hist[-5:])
    # the trained state dict renders through the fused inference path
    m = Model(mc, tm.reference_state_dict(), precision=2)
    r = m.render_rays(batch)[0]
    assert bool(torch.isfinite(r["depth"]).all())


@pytest.mark.gpu
def test_training_step_reads_nothing_back_until_its_terms_are_asked_for():
    """A step of the static model (forward, `losses.total_loss` with every masked term, hash decay, backward, clip, Adam) under
    `torch.cuda.set_sync_debug_mode("error")`: any device-to-host read inside it (a boolean-mask index, `float(tensor)`, `nonzero`) raises.
    `training_step(..., as_tensors=True)` hands the terms back as device scalars; a training loop reads them only when it logs."""
    from nerflidar_hip import scene as nscene
    mc, sd, _ = _ref_scene()
    mc.config.use_intensity = True
    mc.__post_init__()
    from nerflidar_hip import weights as nweights
    tm = ntrain.TrainableModel(mc, fused_mlp=True).cuda().load_reference(nweights.synth_state_dict(mc, seed=0, trained_like=True))
    opt = torch.optim.Adam(tm.parameters(), lr=1e-3, eps=1e-15)
    batch = nscene.supervise(nscene.random_lidar_rays(2048, 0, 1, torch.device("cuda")))
    first = ntrain.training_step(tm, opt, batch)                      # warm-up: lazy initialisations may read back
    assert set(first) >= {"data", "depth", "sem", "int", "loss"} and all(np.isfinite(v) for v in first.values())
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = ntrain.training_step(tm, opt, batch, as_tensors=True)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert all(isinstance(v, torch.Tensor) and v.is_cuda and v.dim() == 0 for v in out.values())
    assert np.isfinite(float(out["loss"]))


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_whole_training_step_matches_the_reference_step(fused):
    """The reference's WHOLE step on one batch - `model(...)` in training mode, the loss assembly of train.py:283-453 (executed from the
    reference's file by tests/golden/make_golden.py:gen_train_step), `.backward()`, nan_to_num_ - against `TrainableModel` +
    `losses.nusc_masks` + `losses.total_loss` + `training.clip_gradients`: every loss term, the total, and the gradients of twelve
    named parameters of the three MLPs incl. their hash tables (missing 3 / weak 7).
    Tolerances: unfused (torch fp32 Linear stack on the HIP operators) - terms to 2e-4 relative, gradients to 5e-3 of their norm
    (measured <= 2.3e-3, cosine 1.00000; the density carries a x1500 gain: summation order in the trunk is 1e-4 of a density, and the interlevel / distortion terms
    differentiate step functions of it).  Fused bf16 MLP (8 mantissa bits per layer of the view MLP, bf16 trunk in training): terms
    to 3e-2, gradients to 6e-2 of their norm with a cosine of at least 0.998 (measured <= 3.7e-2, >= 0.9993)."""
    from nerflidar_hip import losses as nl, weights as nweights, lidar as nlidar2, config as ncfg
    g = golden("train_step_REF")
    mc = ncfg.workload("REF", int(g["log2_hashmap"]))
    mc.config.use_intensity = True
    mc.__post_init__()
    sd = nweights.synth_state_dict(mc, seed=int(g["seed"]), trained_like=True)
    b = nlidar2.synthetic_sweep(width=int(g["width"]), seed=int(g["seed"]), beams=list(g["beams"]))
    batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    for k in ("rgb", "depth", "intensity", "semantic", "mask", "patch_mask", "lidar_mask"):
        batch[k] = torch.from_numpy(g["sup_" + k]).cuda()
    masks = nl.nusc_masks(batch, lidar_supervision=True)
    assert torch.equal(masks["mask_rgb"].cpu(), torch.from_numpy(g["mask_rgb"]))   # train.py:288-324 incl. its `mask == 0` quirk
    batch.update(masks)
    tm = ntrain.TrainableModel(mc, fused_mlp=fused).cuda().load_reference(sd)
    rend, hist = tm(batch, train_frac=float(g["train_frac"]), randomized=False)
    terms = nl.total_loss(rend, hist, batch, data_kind=str(g["data_loss_type"]), charb_padding=float(g["charb_padding"]),
                          data_coarse_mult=float(g["data_coarse_mult"]), data_mult=float(g["data_mult"]),
                          interlevel_mult=float(g["inter_mult"]), anti_interlevel_mult=float(g["anti_mult"]),
                          pulse_width=g["pulse_width"].tolist(), distortion_mult=float(g["dist_mult"]), depth_lam=0.1, sem_lam=0.01)
    loss = sum(terms.values())
    loss.backward()
    ntrain.clip_gradients(tm)
    rt = 3e-2 if fused else 2e-4
    assert set(terms) == {k[5:] for k in g if k.startswith("loss_")}, (sorted(terms), [k for k in g if k.startswith("loss_")])
    for k, v in terms.items():
        np.testing.assert_allclose(float(v.detach()), float(g["loss_" + k]), rtol=rt, atol=1e-7 if not fused else 2e-6, err_msg=k)
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=rt)
    np.testing.assert_allclose(rend[-1]["depth"].detach().cpu().numpy(), g["out_depth"], atol=2e-4 if not fused else 5e-3, rtol=0)
    named = dict(tm.named_parameters())
    report = []
    for k in [k[5:] for k in g if k.startswith("grad_")]:
        want = torch.from_numpy(g["grad_" + k]).double()
        got = named[k].grad.detach().cpu().double()
        rel = float((got - want).norm() / want.norm())
        cos = float((got * want).sum() / (got.norm() * want.norm()))
        ok = (rel <= 6e-2 and cos >= 0.998) if fused else (rel <= 5e-3 and cos >= 0.99999)
        report.append(f"{'ok ' if ok else 'BAD'} {k}: rel {rel:.2e} cos {cos:.5f}")
    print("\n".join(report))
    assert not [r for r in report if r.startswith("BAD")], "\n".join(report)


@pytest.mark.gpu
def test_whole_training_step_with_dynamic_objects_matches_the_reference_step():
    """The SHIPPED configuration's step (`Config.instance_obj = True`, nuscenes_single.gin:13): the reference's
    `model(...)` in training mode with the dynamic-object branch (models.py:401-477), train.py:283-453 executed from the reference's file
    (mask logic with instance_obj, `latent_reg`, `obj_mask` in the interlevel term), `.backward()` - tests/golden/make_golden.py:
    gen_train_step_obj - against `TrainableModel(mc, tracks=, class_names=)`: every loss term to 2e-4, the owner maps of all three levels
    exactly, and the gradients of ObjMLP parameters (trunk, view MLP, rgb layer, hash table), of the latent codes of all tracks and of
    static parameters to 5e-3 of their norm (unfused torch Linear stacks on the HIP operators)."""
    from nerflidar_hip import losses as nl, weights as nweights, lidar as nlidar2, config as ncfg
    g = golden("train_step_OBJ")
    lg, seed = int(g["log2_hashmap"]), int(g["seed"])
    mc = ncfg.workload("REF", lg)
    mc.config.instance_obj, mc.config.latent_size = True, 128
    mc.__post_init__()
    names = {13: "vehicle.car", 14: "vehicle.truck", 15: "vehicle.bus.rigid", 11: "human.pedestrian.adult"}
    class_names = [names[int(c)] for c in g["class_ids"]]
    sd = nweights.synth_state_dict(mc, seed=seed, trained_like=True)
    cids = sorted(set(int(c) for c in g["class_ids"]))
    sd.update(nweights.synth_object_state_dict({c: ncfg.obj_mlp_config(c, latent_size=128, log2_hashmap=lg) for c in cids}, len(class_names), seed=seed))
    b = nlidar2.synthetic_sweep(width=int(g["width"]), seed=seed, beams=list(g["beams"]))
    batch = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    batch["timestamp"] = torch.from_numpy(g["timestamp"]).cuda()
    for k in ("rgb", "depth", "semantic", "mask", "patch_mask", "lidar_mask"):
        batch[k] = torch.from_numpy(g["sup_" + k]).cuda()
    masks = nl.nusc_masks(batch, lidar_supervision=True, instance_obj=True)
    assert torch.equal(masks["mask_rgb"].cpu(), torch.from_numpy(g["mask_rgb"]))
    batch.update(masks)
    tm = ntrain.TrainableModel(mc, tracks=g["tracks"], class_names=class_names, obj_log2_hashmap=lg).cuda().load_reference(sd)
    rend, hist = tm(batch, train_frac=float(g["train_frac"]), randomized=False)
    for lvl, h in enumerate(hist):
        assert torch.equal(h["obj_mask"].cpu(), torch.from_numpy(g[f"hist{lvl}_obj_mask"])), f"owner map of level {lvl}"
    terms = nl.total_loss(rend, hist, batch, depth_lam=0.1, sem_lam=0.01)          # defaults = configs.py, as the fixture's Config()
    terms["latent_reg"] = tm.latent_reg(float(g["latent_reg"]))
    loss = sum(terms.values())
    loss.backward()
    ntrain.clip_gradients(tm)
    assert set(terms) == {k[5:] for k in g if k.startswith("loss_")}, (sorted(terms), [k for k in g if k.startswith("loss_")])
    for k, v in terms.items():
        np.testing.assert_allclose(float(v.detach()), float(g["loss_" + k]), rtol=2e-4, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=2e-4)
    np.testing.assert_allclose(rend[-1]["depth"].detach().cpu().numpy(), g["out_depth"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(rend[-1]["rgb"].detach().cpu().numpy(), g["out_rgb"], atol=2e-4, rtol=0)
    named = dict(tm.named_parameters())
    report = []
    got_lat = torch.stack([torch.zeros(128) if named[f"latent_vector_dict.obj_latent_{t}"].grad is None else
                           named[f"latent_vector_dict.obj_latent_{t}"].grad.cpu() for t in range(len(class_names))])
    checks = [(k[5:], named[k[5:]].grad.detach().cpu(), torch.from_numpy(g[k])) for k in g if k.startswith("grad_") and k != "grad_latents"]
    checks.append(("latent_vector_dict (all tracks)", got_lat, torch.from_numpy(g["grad_latents"])))
    for k, got, want in checks:
        got, want = got.double(), want.double()
        rel = float((got - want).norm() / want.norm())
        cos = float((got * want).sum() / (got.norm() * want.norm()))
        ok = rel <= 5e-3 and cos >= 0.99999
        report.append(f"{'ok ' if ok else 'BAD'} {k}: rel {rel:.2e} cos {cos:.5f}")
    print("\n".join(report))
    assert not [r for r in report if r.startswith("BAD")], "\n".join(report)
    # the proposal levels hand the object networks no gradient (models.py:447-449), the static field does get one
    assert named["prop_mlp_0.encoder.embeddings"].grad.abs().sum() > 0
    # and the trained parameters go back into the fused inference path (objects.DynamicModel) under the reference's names
    from nerflidar_hip.objects import DynamicModel
    dm = DynamicModel(mc, tm.reference_state_dict(), g["tracks"], class_names, device="cuda:0", obj_log2_hashmap=lg)
    r, _ = dm.render_rays({k: v for k, v in batch.items() if k in ("origins", "directions", "viewdirs", "radii", "near", "far", "base_x", "base_y", "timestamp")})
    dd = np.abs(r["depth"].cpu().numpy() - rend[-1]["depth"].detach().cpu().numpy())   # fused bf16 kernels against the torch fp32 stacks on a
    assert np.median(dd) < 1e-3 and np.mean(dd > 1e-2) <= 0.10, (np.median(dd), dd.max())   # white-noise scene: a few rays flip their surface


@pytest.mark.gpu
@pytest.mark.parametrize("F,M", [(6, 64 * 64 * 7 + 13), (8, 4096), (16, 100), (1, 64)])
def test_fused_prop_density_network_matches_torch(F, M):
    """`nlr_prop_mlp_forward` / `_backward` (PropMLP density_layer, ZI/models.py:887-889) against the same two nn.Linear in torch:
    values, feature gradient and all four parameter gradients; M not a multiple of the wave size, F from 1 to 16."""
    torch.manual_seed(F)
    dev = "cuda"
    lin0, lin2 = torch.nn.Linear(F, 64).to(dev), torch.nn.Linear(64, 1).to(dev)
    feats = torch.randn(M, F, device=dev, requires_grad=True)
    cot = torch.randn(M, device=dev)
    ref = lin2(torch.relu(lin0(feats)))[:, 0]
    g_ref = torch.autograd.grad((ref * cot).sum(), [feats, lin0.weight, lin0.bias, lin2.weight, lin2.bias])
    got = ntrain._PropDensity.apply(feats, lin0.weight, lin0.bias, lin2.weight, lin2.bias)
    g_got = torch.autograd.grad((got * cot).sum(), [feats, lin0.weight, lin0.bias, lin2.weight, lin2.bias])
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    for name, a, b in zip(("feats", "w1", "b1", "w2", "b2"), g_got, g_ref):
        assert a.shape == b.shape, name
        scale = float(b.abs().max()) + 1e-12
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=2e-5 * scale + 1e-6, err_msg=name)


@pytest.mark.gpu
@pytest.mark.parametrize("which,rand", [("nerf", False), ("prop0", False), ("nerf", True)])
def test_fused_encode_features_matches_the_torch_chain(which, rand):
    """`nlr_encode_features_forward` / `_backward` (cast -> contract -> grid -> erf re-weight -> mean, ZI/models.py:965-979, one
    kernel each way) against `cast_contract` + `encode_features` (HIP cast kernel, grid operator with its own backward, torch ops for
    the re-weighting and the mean): features and the table gradient."""
    from nerflidar_hip.gridencoder import GridEncoder
    mc, sd, batch = _ref_scene(log2_hashmap=14, width=16)
    cfg = mc.nerf_mlp if which == "nerf" else mc.prop_cfg(0)
    enc = GridEncoder(input_dim=3, num_levels=cfg.grid_num_levels, level_dim=cfg.grid_level_dim, base_resolution=cfg.grid_base_resolution,
                      desired_resolution=cfg.grid_disired_resolution, log2_hashmap_size=cfg.grid_log2_hashmap_size, init_std=0.5).cuda()
    n, S = batch["origins"].shape[0], 48
    gen = torch.Generator(device="cuda").manual_seed(3)
    td = torch.sort(torch.rand(n, S + 1, device="cuda", generator=gen) * 1.2 + 0.01, dim=-1)[0]
    rd = torch.rand(n, S, 7, device="cuda", generator=gen) if rand else None
    cot = torch.randn(n, S, enc.output_dim, device="cuda", generator=gen)
    means, stds = ntrain.cast_contract(batch, td, rand_deg=rd)
    ref = ntrain.encode_features(enc, means, stds, True)
    (g_ref,) = torch.autograd.grad((ref * cot).sum(), [enc.embeddings])
    got = ntrain.encode_features_fused(enc, batch, td, rand_deg=rd, re_weights=True)
    (g_got,) = torch.autograd.grad((got * cot).sum(), [enc.embeddings])
    assert got.shape == ref.shape
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().cpu().numpy(), atol=2e-6, rtol=2e-5)
    scale = float(g_ref.abs().max())
    assert scale > 0
    np.testing.assert_allclose(g_got.cpu().numpy(), g_ref.cpu().numpy(), rtol=0, atol=2e-5 * scale)


@pytest.mark.gpu
def test_encode_kernels_refuse_a_sample_count_beyond_their_lane_index():
    """8 lanes per sample with a 32-bit lane index: N * S >= 2^29 samples is an enumerated error (NLR_ERR_INVALID with a message), not a
    wrapped grid.  Nothing is launched: the tensors handed in are far smaller than the claimed shape."""
    import ctypes as C
    from nerflidar_hip import _lib
    from nerflidar_hip.gridencoder import GridEncoder
    mc, sd, batch = _ref_scene(log2_hashmap=12, width=8)
    enc = GridEncoder(input_dim=3, num_levels=6, level_dim=1, base_resolution=16, desired_resolution=512, log2_hashmap_size=12).cuda()
    n = batch["origins"].shape[0]
    from nerflidar_hip.models import _RAY_KEYS
    keep = [batch[k].reshape(n, -1).contiguous().float() for k in _RAY_KEYS]
    rays, gd = ntrain._EncodeFeatures._descs(enc, keep, enc.embeddings.detach().contiguous())
    td = torch.zeros(64, device="cuda")
    out = torch.zeros(64, device="cuda")
    rc = _lib.lib().nlr_encode_features_forward(C.byref(rays), _lib.ptr(td), 1 << 23, 64, 7, 3, 0.35, None, C.byref(gd), 1, _lib.ptr(out), None)
    assert rc == -1 and b"32-bit lane index" in _lib.lib().nlr_last_error()
