"""End-to-end learning check on a multi-view-consistent synthetic scene (the lego / fern datasets are not available offline).

An analytic scene (a shaded sphere in front of a white background, seen by a ring of cameras with the reference's camera
convention -- rays come from nerf_hip_rays, so quirk Q2 is honoured) is rendered to images; a NeRFRunner trains on the
training views and is evaluated on held-out views.  Reports PSNR before/after and the trainer's rays/s.
Usage (GPU box):  python scripts/teacher_student.py [iterations] [bf16|f32|split] [seed] [batch_ray]  ->  one JSON line (copied to profiles/ by hand).
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_tiny_amd as P  # noqa: E402


def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 10.0 * np.log10(1.0 / max(mse, 1e-12))


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    bf16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
    split = len(sys.argv) > 2 and sys.argv[2] == "split"  # the opt-in split-fp32 train step (model.split_train)
    if len(sys.argv) > 3:  # same initial weights (and whatever else draws from torch's default generators) for A/B runs
        torch.manual_seed(int(sys.argv[3]))
        torch.cuda.manual_seed_all(int(sys.argv[3]))
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
    dev = torch.device("cuda:0")
    H = W = 64
    scene = P.data.analytic_sphere_scene(n_pic=24, H=H, W=W, seed=5, device=dev)
    poses, imgs = scene.poses_bounds, scene.all_pix.view(24, H, W, 3)
    test_idx = np.arange(0, 24, 6)
    train_idx = np.setdiff1d(np.arange(24), test_idx)
    train = P.data.ArrayDataset(imgs[train_idx], poses[train_idx])
    test = P.data.ArrayDataset(imgs[test_idx], poses[test_idx])
    out_dir = os.path.join(ROOT, "gpurun_out", "teacher_student") + "/"
    run = P.NeRFRunner(gpu=0, img_dir="", results_path=out_dir, ckpt_path=out_dir + "ck/", low_res=1, total_iter=iters, batch_ray=batch,
                       learning=3e-4, lr_gamma=0.1, lr_milestone=[10, 200], n_coarse=64, n_fine=128, data_type="sync", step=10 ** 9,
                       decay_end=10 * iters, sched="EXP", continue_=False, datasets={"train": train, "val": train, "test": test},
                       log_every=max(iters // 10, 1), bf16_mlp=bf16, split_train=split, on_resample_fault="warn")
    before = psnr(torch.from_numpy(run.display(save=False)), imgs[test_idx])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.trainer("train")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pred = torch.from_numpy(run.display(save=True))
    after = psnr(pred, imgs[test_idx])
    run.disp_rays = run.train_rays  # same renderer on the training views
    after_train = psnr(torch.from_numpy(run.display(save=False)), imgs[train_idx])
    print(json.dumps({"scene": f"analytic sphere, 20 train / 4 held-out views of 64x64, {batch}-ray batches, 64+128 samples, " + ("bf16 MLP" if bf16 else "split-fp32 train step" if split else "fp32"),
                      "iterations": iters, "psnr_heldout_before_db": round(before, 2), "psnr_heldout_after_db": round(after, 2), "psnr_train_views_after_db": round(after_train, 2),
                      "trainer_rays_per_s": round(iters * batch / dt, 1), "train_seconds": round(dt, 1)}))


if __name__ == "__main__":
    main()
