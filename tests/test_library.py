"""CPU-side checks of the product's host code: the C-ABI library loads and exports every symbol the header
declares, host helpers agree with torch / the oracle, the gin-lite reader and config names, and the
multi-process (gloo, world_size 2) path of the azimuth-sharded driver.  No GPU compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from nerflidar_hip import _lib, config as nconfig, lidar as nlidar, weights as nweights


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "nerflidar_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)   # declarations only: the comments name kernels and Python helpers too
    declared = set(re.findall(r"\b(nlr_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/nerflidar_hip.h but not exported"
    assert set(_lib.EXPORTS) == declared
    assert L.nlr_version() >= 100


def test_struct_sizes_match_header():
    """ctypes mirrors must have the C layout (pointer/uint32 packing)."""
    assert C.sizeof(_lib.NlrLinear) == 24
    assert C.sizeof(_lib.NlrGridDesc) == 56
    assert C.sizeof(_lib.NlrRays) == 64
    assert C.sizeof(_lib.NlrLevelOut) == 112
    assert C.sizeof(_lib.NlrOut) == 12 * 8 + 8 + 4 * 112
    assert C.sizeof(_lib.NlrRenderCfg) == 16 + 2 * 32 + 8


def test_sample_u_matches_torch_linspace():
    L = _lib.lib()
    eps = float(torch.finfo(torch.float32).eps)
    for n in (2, 7, 32, 64, 128, 256):
        for rand in (0, 1):
            u = np.zeros(n, np.float32)
            mj = C.c_float(0)
            L.nlr_sample_u(n, rand, u.ctypes.data, C.addressof(mj))
            if not rand:
                pad = 1 / (2 * n)
                ref = torch.linspace(pad, 1. - pad - eps, n).numpy()
            else:
                u_max = eps + (1 - eps) / n
                ref = torch.linspace(0, 1 - u_max, n).numpy()
                assert np.float32((1 - u_max) / (n - 1) - eps) == np.float32(mj.value)
            # ATen's vectorised linspace and its scalar formula differ by at most one ulp
            np.testing.assert_allclose(u, ref, atol=1.2e-7, rtol=0)
            assert (np.diff(u) > 0).all()


def test_level_scale_matches_oracle():
    from oracle import nlr_oracle as orc
    L = _lib.lib()
    for S, H, n in ((1.0, 16, 10), (0.5849625, 16, 12), (1.0, 16, 8)):
        sc, rs = np.zeros(n, np.float32), np.zeros(n, np.uint32)
        L.nlr_level_scale(n, S, H, sc.ctypes.data, rs.ctypes.data)
        osc, ors = orc.level_scale(n, S, H)
        np.testing.assert_array_equal(sc, osc)
        np.testing.assert_array_equal(rs, ors)


def test_gin_lite_reads_shipped_bindings():
    text = """
    Config.use_semantic = True
    Config.no_sem_layer = False
    Model.raydist_fn = 'power_transformation'
    Model.opaque_background = True
    Model.num_nerf_samples = 128
    Model.num_prop_samples = (256, 64)   # render_video.py:130
    PropMLP.disable_rgb = True
    PropMLP.grid_level_dim = 1
    NerfMLP.net_depth_viewdirs = 8
    NerfMLP.net_width_viewdirs = 256
    ObjMLP.bottleneck_width = 64         # ignored scope
    Config.unknown_field = 3             # ignored field
    """
    mc = nconfig.parse_gin_bindings(text)
    assert mc.num_nerf_samples == 128 and mc.num_prop_samples == (256, 64)
    assert mc.nerf_mlp.net_depth_viewdirs == 8 and mc.nerf_mlp.use_semantic and not mc.nerf_mlp.no_sem_layer
    assert mc.level_samples() == [256, 64, 128]


def test_mac_counts_match_survey():
    """Algorithmic MACs/sample used by bench.py's roofline."""
    from nerflidar_hip.flops import macs_per_sample, flops_per_ray
    assert macs_per_sample(nconfig.workload("C2").nerf_mlp) == 657408
    # C1 (4x128 + semantic head): 18944 + 36224 + 52608 + 2*16384 + 384 + 17600 (SURVEY 8d quotes 117632, which
    # does not correspond to the layer shapes of ZI/models.py:939-957; the shapes are authoritative)
    assert macs_per_sample(nconfig.workload("C1").nerf_mlp) == 158528
    assert macs_per_sample(nconfig.workload("REF").prop_cfg(0)) == 448
    assert macs_per_sample(nconfig.workload("REF").prop_cfg(1)) == 576
    # REF: the shipped gin has use_intensity=False -> 247744 MACs/sample for the NerfMLP (SURVEY counts the
    # intensity head in too: 264192); 2*(64*448 + 64*576 + 32*247744)
    assert flops_per_ray(nconfig.workload("REF")) == 2 * (64 * 448 + 64 * 576 + 32 * 247744)
    assert flops_per_ray(nconfig.workload("C2")) == 2 * (64 * 448 + 64 * 576 + 128 * 657408)
    assert abs(flops_per_ray(nconfig.workload("C2S")) / 1e6 - 168.3) < 0.05


def test_azimuth_sector_partition():
    b = nlidar.synthetic_sweep(width=20, seed=0, beams=nlidar.LIDAR_ANGLES[:4])
    parts = [nlidar.azimuth_sector(b, 4, 20, r, 3) for r in range(3)]
    wp = parts[0][1]
    assert wp == 7
    img = np.concatenate([p[0]["directions"].reshape(4, wp, 3) for p in parts], axis=1)[:, :20]
    np.testing.assert_array_equal(img.reshape(-1, 3), b["directions"])
    # viewdirs keep the full-sweep Frobenius norm (quirk), so a sector renders identically to the full sweep
    np.testing.assert_array_equal(parts[1][0]["viewdirs"][0], b["viewdirs"][7])


def _shard_worker(rank, world, port, tmp, W=22):
    import torch.distributed as dist
    from nerflidar_hip import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H = 4
    full = nlidar.synthetic_sweep(width=W, seed=1, beams=nlidar.LIDAR_ANGLES[:H])

    def fake_render(b, packed=None):  # deterministic per-ray function standing in for the HIP renderer
        d, o = b["directions"], b["origins"]
        depth = (d * torch.tensor([1.0, 2.0, 3.0])).sum(-1) + o[:, 0]
        return dict(depth=depth, intensity=depth * 0.5, acc=torch.ones_like(depth), rgb=d.abs(),
                    labels=(depth.abs() * 7).to(torch.int32) % 19)

    one_r = fake_render({k: torch.from_numpy(v) for k, v in full.items()})
    one = sharding.pack_tile(one_r, H, W)  # azimuth-major [W, H, 7]
    g = sharding.SweepGatherer(H, W, "cpu")
    assert g.wp == -(-W // world) and g.images[0].shape[0] == g.wp * world  # the gathered buffer carries the padded columns
    for i in range(3):  # double buffering: three sweeps through two buffers
        img = sharding.render_sweep_sharded(fake_render, full, H, W, "cpu", gatherer=g, index=i)
        assert img.shape == (W, H, 7)
        assert torch.equal(img, one), "gathered range image differs from the single-process image"
    u = sharding.unpack_image(img)
    assert torch.equal(u["labels"].reshape(-1), one_r["labels"]) and torch.equal(u["depth"].reshape(-1), one_r["depth"])
    assert torch.equal(sharding.as_hw(img)[..., 3:6].reshape(-1, 3), one_r["rgb"])
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("1")


def test_sharded_sweep_world2_gloo(tmp_path):
    """N>1 path on CPU: azimuth-sector partition + ONE all-gather reassembles the one-process image bit for bit."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_sharded_sweep_world4_padded_gloo(tmp_path):
    """A width that really pads: 22 columns over 4 ranks = 6 per rank, 2 padded columns in the last rank's tile, which the
    gathered image drops (sharding.SweepGatherer.image -> images[b][:width])."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_shard_worker, args=(4, port, str(tmp_path), 22), nprocs=4, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(4))


def test_bench_selftest_reference_sweep_8_ranks():
    """The reference's real sweep shape, 32 x 1100 (ZI/lidar_utils.py:122-134), over 8 CPU ranks: 138 columns per rank, 4 padded."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-cpu", "--steps", "2", "--gpus", "8",
                        "--selftest-hw", "32", "1100"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["image_equal"] and d["ranks_seen"] == 8 and d["columns_per_rank"] == 138 and d["padded_columns"] == 4
    assert d["shape"] == [1100, 32, 7]


@pytest.mark.parametrize("extra", [["--gpus", "2"], ["--gpus", "2", "--scaling", "weak"], ["--gpus", "1", "--force-dist"]])
def test_bench_self_launch_cpu(extra):
    """`python bench.py --gpus N` without a launcher environment starts its own N ranks (torch.distributed.run children) and
    prints ONE JSON line; here on CPU ranks (gloo) with the stand-in renderer of --selftest-cpu."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-cpu", "--steps", "3"] + extra,
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["selftest"] and d["image_equal"] and d["n_gpus"] == int(extra[1]) and d["ranks_seen"] == int(extra[1])
    assert d["shape"] == [64 * (2 if "weak" in extra else 1), 4, 7]


def test_binary_carries_its_source_hash_and_stale_builds_lose_traffic(tmp_path, monkeypatch):
    """The source hash is compiled INTO libnerflidar_hip.so (`nlr_build_sha`), bench.py stamps its line with the binary's value and
    quotes PMC traffic only while binary == sources == profile."""
    import importlib.util
    import json
    from nerflidar_hip import buildinfo
    b, s_ = buildinfo.binary_sha(), buildinfo.kernel_source_sha()
    assert len(b) == 64 and b == s_ and buildinfo.stale() is None, "libnerflidar_hip.so is stale against the sources: run make -C nerf-lidar_amd"
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    prof = tmp_path / "pmc.json"
    json.dump({"kernel_source_sha": b, "kernels": {"void nlr_mlp_kernel<8>": {"hbm_read_bytes_corrected": 100.0, "hbm_write_bytes": 5.0}}}, open(prof, "w"))
    t, note = bench.pmc_traffic(str(prof), b, s_, "nlr_mlp_kernel")
    assert t == 105.0 and "bytes per launch" in note
    # a source file edited after the build: the tree's hash moves, the binary's does not -> traffic withheld, mismatch named
    extra = tmp_path / "tampered.hip"
    extra.write_text("// edited after the build\n")
    monkeypatch.setattr(buildinfo, "kernel_source_files", lambda f=buildinfo.kernel_source_files: f() + [str(extra)])
    s2 = buildinfo.kernel_source_sha()
    assert s2 != b and "stale" in buildinfo.stale() and b[:12] in buildinfo.stale() and s2[:12] in buildinfo.stale()
    t, note = bench.pmc_traffic(str(prof), b, s2, "nlr_mlp_kernel")
    assert t is None and "stale binary" in note and b[:12] in note and s2[:12] in note
    # a profile of another build
    json.dump({"kernel_source_sha": "0" * 64, "kernels": {}}, open(prof, "w"))
    t, note = bench.pmc_traffic(str(prof), b, b, "nlr_mlp_kernel")
    assert t is None and "was measured on kernel source 000000000000" in note


def test_fast_level_body_envelope_is_what_32_bit_arithmetic_needs():
    """`nlr_level_fast_ok` just inside / outside each of its limits (host arithmetic only, no table is touched).  A level of
    2^28 sixteen-byte entries is 4 GiB = 2^32 bytes: the last size whose byte offsets fit; 2^29 is out.  Smoothstep, align_corners and
    the tiled grid type (non power-of-two wrap) are out; resolutions far beyond round 3's 2^14 limit are in."""
    import numpy as np
    from nerflidar_hip import _lib
    L_ = _lib.lib()

    def fast(offsets, Lv, C, dtype=0, gridtype=0, align=0, interp=0, S=1.0, H=16):
        off = np.ascontiguousarray(offsets, np.int32)
        return L_.nlr_grid_fast_path(off.ctypes.data, Lv, C, float(S), H, dtype, gridtype, align, interp)

    def table(Lv, log2, H=16):
        sizes = [min(2 ** log2, (H * 2 ** l + 1) ** 3) for l in range(Lv)]
        sizes = [-(-s_ // 8) * 8 for s_ in sizes]
        return np.concatenate([[0], np.cumsum(sizes)])

    assert fast(table(10, 21), 10, 4) == 1                    # the NerfMLP grid
    assert fast(table(16, 19), 16, 2) == 1                    # 16 levels to resolution 524 288 (round 3: generic)
    assert fast(table(10, 21), 10, 4, interp=1) == 0          # smoothstep
    assert fast(table(10, 21), 10, 4, align=1) == 0
    assert fast(table(10, 21), 10, 4, gridtype=1) == 0        # tiled: wrap by modulo of a non power of two
    # byte offsets: int32 offsets cap a whole table at 2^31 entries, so the bound is met through the entry size: 2^28 entries x 16 B in,
    # 2^28 x 32 B (C = 8, fp32) out
    big = np.array([0, 2 ** 28], np.int64)
    assert fast(big, 1, 4, H=2 ** 12) == 1
    assert fast(big, 1, 8, H=2 ** 12) == 0
    assert fast(big, 1, 8, dtype=1, H=2 ** 12) == 1           # fp16: 16-byte entries again
