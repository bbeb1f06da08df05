"""Train a model on the analytic scene (`nerflidar_hip.scene`) with LiDAR supervision, the way ZI/train.py trains on nuScenes sweeps,
and leave a checkpoint in the reference's format.

    python -m nerflidar_hip.train_scene --workload REFI --log2-hashmap 13 --steps 3000 --out ckpt_dir

The loop is train.py:180-190,272-459,559-566 without the dataset / accelerate / logging plumbing: per step the learning rate of
`training.create_optimizer` is written into the optimiser, a batch of LiDAR rays is drawn (`scene.random_lidar_rays`, the batch
contract of ZI/lidar_utils.py:8-33) and supervised by ray casting (`scene.supervise`: depth, intensity, semantic, rgb),
`training.training_step` takes the step (forward with jitter through the HIP kernels, the loss dictionary of train.py:326-446, backward
through the HIP backward kernels, Adam), and `checkpoints.save_checkpoint` writes `checkpoint_<step>.ckpt` at the end.  Afterwards the
checkpoint is read back through `checkpoints.model_from_checkpoint` and a held-out sweep is rendered on the fused inference path and
compared with the analytic ground truth (metres, label accuracy), so that a run reports whether the field it leaves is a scene.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np
import torch

from . import checkpoints as nckpt
from . import config as nconfig
from . import lidar as nlidar
from . import scene as nscene
from . import training as ntrain


def evaluate(model, scale_factor: float, sweep_idx: int = 100, width: int = 1024, seed: int = 0):
    """Held-out sweep (a sensor position no training ray starts from) on the fused inference path against the analytic scene."""
    b = nlidar.synthetic_sweep(width=width, seed=seed, scale_factor=scale_factor, sweep_idx=sweep_idx)
    tb = {k: torch.from_numpy(v).to(model.device) for k, v in b.items()}
    gt = nscene.cast(tb["origins"], tb["directions"], nlidar.seeded_rotation(seed), scale_factor)
    r, _ = model.render_rays(tb, scale_factor=scale_factor)
    err_m = (r["depth"] - gt["depth"]).abs() / scale_factor
    out = dict(rays=int(err_m.numel()), depth_err_m_median=float(err_m.median()), depth_err_m_mean=float(err_m.mean()),
               depth_err_m_p90=float(err_m.quantile(0.9)), label_accuracy=float((r["labels"].long() == gt["semantic"]).float().mean()))
    if "intensity" in r:
        out["intensity_mae"] = float((r["intensity"] - gt["intensity"]).abs().mean())
    out["rgb_mae"] = float((r["rgb"] - gt["rgb"]).abs().mean())
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--workload", default="REFI", help="architecture (nerflidar_hip.config.workload): REF, REFI, C2, ...")
    ap.add_argument("--log2-hashmap", type=int, default=None, help="hash-map size of all three grids (default: the gin value 21)")
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--rays", type=int, default=16384, help="rays per step (Config.batch_size is 65 536, configs.py:29)")
    ap.add_argument("--fused", type=int, default=1, help="1: fused bf16 MFMA NerfMLP forward / backward; 0: torch Linear modules in fp32")
    ap.add_argument("--lr-init", type=float, default=0.01)
    ap.add_argument("--lr-final", type=float, default=0.001)
    ap.add_argument("--lr-delay-steps", type=int, default=None, help="default: a fifth of the run (5 000 of 25 000 in configs.py:87)")
    ap.add_argument("--depth-lam", type=float, default=0.1, help="train.py:331-332: 0.1 before Config.end_step, 0.4 after")
    ap.add_argument("--sem-lam", type=float, default=0.01, help="train.py:409-410: 0.01 before Config.end_step, 0.04 after")
    ap.add_argument("--hash-decay", type=float, default=0.1)
    ap.add_argument("--scale-factor", type=float, default=1.0 / 250.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", required=True, help="checkpoint directory")
    ap.add_argument("--log-every", type=int, default=200)
    ap.add_argument("--eval-every", type=int, default=0, help="also render the held-out sweep on the fused inference path every this many steps")
    a = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("train_scene needs a GPU: the training operators have no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(a.seed)
    mc = nconfig.workload(a.workload, a.log2_hashmap)
    tm = ntrain.TrainableModel(mc, fused_mlp=bool(a.fused)).to(dev)
    delay = a.lr_delay_steps if a.lr_delay_steps is not None else max(a.steps // 5, 1)
    opt, lr_fn = ntrain.create_optimizer(tm, a.lr_init, a.lr_final, a.steps, delay)
    t0 = time.time()
    log = []
    for step in range(1, a.steps + 1):                                     # train.py:180: steps count from 1
        for g in opt.param_groups:
            g["lr"] = lr_fn(step)                                          # train.py:187-190
        batch = nscene.supervise(nscene.random_lidar_rays(a.rays, a.seed, step, dev, a.seed, a.scale_factor), a.seed, a.scale_factor)
        train_frac = float(np.clip((step - 1) / max(a.steps - 1, 1), 0, 1))  # train.py:184
        terms = ntrain.training_step(tm, opt, batch, train_frac=train_frac, randomized=True, hash_decay_mult=a.hash_decay,
                                     depth_lam=a.depth_lam, sem_lam=a.sem_lam, as_tensors=True)
        if step % a.log_every == 0 or step == 1 or step == a.steps:          # the only host read of the loop
            torch.cuda.synchronize()
            rec = dict(step=step, lr=lr_fn(step), elapsed_s=round(time.time() - t0, 1), **{k: round(float(v), 6) for k, v in terms.items()})
            log.append(rec)
            print(json.dumps(rec), flush=True)
        if a.eval_every and step % a.eval_every == 0 and step < a.steps:
            from .models import Model
            ev = evaluate(Model(mc, tm.reference_state_dict(), device=dev), a.scale_factor, seed=a.seed)
            print(json.dumps(dict(step=step, held_out_sweep=ev)), flush=True)
    torch.cuda.synchronize()
    train_s = time.time() - t0
    path = nckpt.save_checkpoint(a.out, tm.reference_state_dict(), a.steps)
    # the round trip a user of the reference takes: file -> state_dict -> Model (weights packed for the fused kernels)
    model, step, ignored = nckpt.model_from_checkpoint(a.out, base=mc)
    ev = evaluate(model, a.scale_factor, seed=a.seed)
    summary = dict(workload=a.workload, log2_hashmap=a.log2_hashmap, steps=a.steps, rays_per_step=a.rays, fused=bool(a.fused),
                   train_seconds=round(train_s, 1), train_rays_per_s=round(a.steps * a.rays / train_s), checkpoint=path,
                   checkpoint_bytes=os.path.getsize(path), restored_step=step, held_out_sweep=ev, final_terms=log[-1])
    print(json.dumps(summary), flush=True)
    with open(os.path.join(a.out, "train_summary.json"), "w") as f:
        json.dump(dict(summary=summary, log=log), f, indent=1)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
