"""CPU tests of the `render_image` driver (ZI/models.py:1379-1507 contract): chunk plan, zero-ray padding, rank shares,
gather and reassembly -- with a stand-in model (a pure per-ray function), one process and two gloo processes.
The real renderer behind the same driver is covered by tests/test_hip_parity.py::test_render_image_driver_and_labels."""
import os
import socket

import numpy as np
import torch

from nerflidar_hip import config as nconfig
from nerflidar_hip.models import _chunk_plan, _rows, render_image


class FakeModel:
    """Per-ray pure function with the return structure of Model.forward: two levels, ray_* bundles on both."""
    training = True

    def eval(self):
        self.training = False

    def train(self):
        self.training = True

    def __call__(self, rand, batch, train_frac, compute_extras, zero_glo=True):
        o, d = batch["origins"], batch["directions"]
        n = o.shape[0]
        depth = (d * torch.tensor([1.0, -2.0, 0.5])).sum(-1) + o[:, 0]
        sem = torch.stack([depth * (c + 1) for c in range(4)], -1)
        sd = [torch.linspace(0, 1, S + 1).repeat(n, 1) * (1 + depth[:, None]) for S in (8, 4)]
        w = [torch.ones(n, S) * depth[:, None] for S in (8, 4)]
        rends = [dict(depth=depth * 0.5, ray_sdist=sd[0], ray_weights=w[0], ray_rgbs=w[0][..., None].repeat(1, 1, 3)),
                 dict(rgb=d.abs(), depth=depth, semantic=sem, acc=torch.ones(n), ray_sdist=sd[1], ray_weights=w[1],
                      ray_rgbs=w[1][..., None].repeat(1, 1, 3))]
        hist = [dict(weights=w[0]), dict(weights=w[1])]
        return rends, hist


def _batch(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    return dict(origins=torch.rand(n, 3, generator=g), directions=torch.rand(n, 3, generator=g) - 0.5,
                radii=torch.rand(n, 1, generator=g), lossmult=None)


def test_chunk_plan_covers_every_ray_once():
    for total, chunk, world in ((100, 40, 1), (100, 40, 3), (7, 16, 2), (64, 16, 4), (33, 8, 5)):
        seen = np.zeros(total, int)
        for rank in range(world):
            for first, count, lo, hi in _chunk_plan(total, chunk, world, rank):
                assert (hi - lo) * world >= count and (hi - lo) * world - count < world  # minimal padding
                real = np.arange(first + lo, first + min(hi, count))
                seen[real] += 1
        assert (seen == 1).all()


def test_rows_pads_with_zero_rays():
    flat = {"origins": torch.arange(30.0).reshape(10, 3)}
    part = _rows(flat, 4, 5, 3, 6)["origins"]  # chunk rays 4..8, share rows 3..5 -> 2 real rays + 1 zero ray
    assert part.shape == (3, 3) and torch.equal(part[:2], flat["origins"][7:9]) and (part[2] == 0).all()
    assert _rows(flat, 4, 5, 6, 9)["origins"].abs().sum() == 0  # a share that lies entirely in the padding


def test_render_image_one_process_chunked_equals_one_shot():
    torch.manual_seed(0)
    b = _batch(6 * 7)
    img = {k: (v.reshape(6, 7, -1) if v is not None else None) for k, v in b.items()}
    cfg = nconfig.Config(render_chunk_size=16)
    m = FakeModel()
    out = render_image(m, None, img, False, cfg)
    assert m.training  # restored (models.py:1505)
    ref = m(False, {k: v for k, v in b.items() if v is not None}, 1, True)[0][-1]
    assert out["depth"].shape == (6, 7) and out["rgb"].shape == (6, 7, 3) and out["semantic"].shape == (6, 7, 4)
    for k in ("depth", "rgb", "semantic", "acc"):
        assert torch.equal(out[k].reshape(ref[k].shape), ref[k])
    assert len(out["ray_sdist"]) == 2 and out["ray_sdist"][0].shape == (min(cfg.vis_num_rays, 42), 9)
    flat = render_image(m, None, {k: v for k, v in b.items()}, False, cfg, image=False, return_weights=True)
    assert flat["depth"].shape == (42, 1) and flat["weights"].shape == (42, 4)  # image=False: [N, -1] (models.py:1491)


def _worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Acc:  # the three members of accelerate.Accelerator that render_image touches
        process_index, num_processes = rank, world
        calls = 0

        def gather(self, t):
            Acc.calls += 1
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            return torch.cat(parts)

    b = _batch(45, seed=3)  # 45 rays, chunks of 16: the last chunk (13 rays) needs one zero ray of padding
    cfg = nconfig.Config(render_chunk_size=16)
    m = FakeModel()
    ref = m(False, {k: v for k, v in b.items() if v is not None}, 1, True)
    outs = {}
    for packed in (True, False):
        Acc.calls = 0
        torch.manual_seed(5)    # the random subset of the ray bundles (models.py:1495-1503) is drawn after the chunks
        out = outs[packed] = render_image(m, Acc(), dict(b), False, cfg, image=False, return_weights=True, packed_gather=packed)
        for k in ("depth", "rgb", "semantic", "acc"):
            assert torch.equal(out[k].reshape(ref[0][-1][k].shape), ref[0][-1][k]), k
        assert torch.equal(out["weights"], ref[1][-1]["weights"])
        chunks = 3                                        # 45 rays in chunks of 16
        # one collective per chunk, against one per key (4 per-ray keys + 3 bundles x 2 levels + weights = 11) per chunk
        assert Acc.calls == (chunks if packed else chunks * 11), (packed, Acc.calls)
    for k in outs[True]:                                   # the two call patterns assemble the same rendering, bit for bit
        a, c = outs[True][k], outs[False][k]
        if isinstance(a, list):
            assert all(torch.equal(x, y) for x, y in zip(a, c)), k
        else:
            assert torch.equal(a, c), k
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("1")


def test_render_image_two_processes_gloo(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
