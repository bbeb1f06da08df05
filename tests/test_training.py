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
        losses.append(float(loss))
    assert losses[-1] < 0.2 * losses[0], losses[::6]
