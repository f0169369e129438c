"""Does the REFERENCE's training recipe itself run away on the teacher/student scene, or only the device path?
The same small problem (analytic sphere, 20 training views of 32 x 32, 256-ray batches, 16 + 32 samples, lr 3e-4, Adam eps 1e-7, the
EXP schedule of nerf.py:426, sum-of-squares loss) trained two ways from the same initial weights on the SAME sequence of batches:
  oracle   oracle/nerf_oracle.py (the bit-identical CPU restatement of the reference's forward) + torch autograd + torch.optim.Adam
  gpu      NeRFModel (libnerf_hip) + train.FusedAdam, fp32 or bf16-MLP
One line per run: the loss every `every` steps and whether the run ended in the dead state (loss stuck at the all-background value).
Usage:  [LR=1e-3] [THREADS=2] [STOP=800] [EVERY=100] python tests/tools/collapse_stats.py oracle|gpu|gpu_bf16 ITERS SEED [SEED ...]
"""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nerf_oracle as O  # noqa: E402

H = W = 32
N_PIC, B, NC, NF = 24, 256, 16, 32
LR, GAMMA = float(os.environ.get("LR", "3e-4")), 0.1  # (LR=1e-3: the reference runner's default, nerf.py:358)


def scene():
    """scripts/teacher_student.py's scene (data.analytic_sphere_scene) on the CPU: rays from the oracle (= the reference's camera model)."""
    rng = np.random.default_rng(5)
    focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
    K_inv = O.make_K_inv(H, W, focal)
    rr, cc = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    row, col = rr.reshape(-1), cc.reshape(-1)
    light = torch.tensor([0.5, 0.3, 0.8]).double()
    light = light / light.norm()
    poses, imgs = [], torch.ones(N_PIC, H, W, 3)
    for i in range(N_PIC):
        ang = 2 * math.pi * i / N_PIC + rng.uniform(-0.05, 0.05)
        elev = rng.uniform(0.2, 0.6)
        o = 4.0 * np.array([math.cos(ang) * math.cos(elev), math.sin(ang) * math.cos(elev), math.sin(elev)])
        fwd = -o / np.linalg.norm(o)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        c2w = np.stack([right, up, -fwd, o], axis=1)  # camera looks down -z (Blender convention)
        pr = O.pose_row(c2w, H, W, focal, 2.0, 6.0)
        poses.append(pr)
        pb = torch.from_numpy(np.tile(pr, (H * W, 1))).float()
        d_wrd = O.world_dirs(O.poses_extract(pb)[0], O.camera_dirs(row, col, K_inv)).double()
        d_wrd = d_wrd / d_wrd.norm(dim=1, keepdim=True)
        oo = torch.from_numpy(o).double()
        b = (d_wrd * oo).sum(1)
        disc = b * b - ((oo * oo).sum() - 1.0)
        t = -b - torch.sqrt(disc.clamp_min(0))
        p = oo + t[:, None] * d_wrd
        colour = (0.5 + 0.5 * p) * (p * light).sum(1).clamp_min(0.15)[:, None]
        imgs[i] = torch.where((disc > 0)[:, None], colour, torch.ones_like(colour)).float().reshape(H, W, 3)
    return np.stack(poses), imgs, K_inv


def batches(seed, n_train_pix, iters):
    g = torch.Generator().manual_seed(1000 + seed)
    out = []
    while len(out) < iters:
        perm = torch.randperm(n_train_pix, generator=g)
        for s in range(0, n_train_pix - B + 1, B):
            out.append(perm[s:s + B])
            if len(out) == iters:
                break
    return out


def main():
    if os.environ.get("THREADS"):
        torch.set_num_threads(int(os.environ["THREADS"]))
    mode, iters, seeds = sys.argv[1], int(sys.argv[2]), [int(s) for s in sys.argv[3:]]
    every = int(os.environ.get("EVERY", max(iters // 10, 1)))
    poses, imgs, K_inv = scene()
    train_idx = np.setdiff1d(np.arange(N_PIC), np.arange(0, N_PIC, 6))
    pix = imgs[train_idx].reshape(-1, 3)
    pr = torch.from_numpy(poses[train_idx]).float()
    hit = float((pix.sum(1) < 2.999).float().mean())
    dead = None
    for seed in seeds:
        torch.manual_seed(seed)
        w0 = O.make_weights(seed)  # U(-1/sqrt(fan_in), 1/sqrt(fan_in)): the nn.Linear default the reference starts from
        bl = batches(seed, pix.shape[0], iters)[:int(os.environ.get("STOP", iters))]  # (STOP: end early, the lr schedule stays the one of ITERS)
        lam = lambda it: GAMMA ** (it / (10 * iters))
        t0 = time.perf_counter()
        losses, q7 = [], None
        if mode == "oracle":
            params = {k: v.clone().requires_grad_(True) for k, v in w0.items()}
            opt = torch.optim.Adam(list(params.values()), lr=LR, betas=(0.9, 0.999), eps=1e-7)
            sch = torch.optim.lr_scheduler.LambdaLR(opt, lam)
            for it, idx in enumerate(bl):
                pic, rem = idx // (H * W), idx % (H * W)
                opt.zero_grad(set_to_none=True)
                try:
                    Cc, Cf = O.render(params, rem // W, rem % W, pr[pic], K_inv, NC, NF, check=True)
                except O.ResampleIndexError:
                    q7 = it  # a ray whose coarse weights are all zero: the reference prints its banner and exit(0)s here (nerf.py:251-253)
                    break
                loss = O.ray_loss(Cc, Cf, pix[idx])
                loss.backward()
                opt.step()
                sch.step()
                if (it + 1) % every == 0:
                    losses.append(round(float(loss.detach()), 1))
        else:
            import nerf_tiny_amd as P

            dev = torch.device("cuda:0")
            m = P.NeRFModel(NC, NF, B)
            m.load_state_dict(w0)
            m = m.to(dev)
            m.bf16_mlp = mode == "gpu_bf16"
            opt = P.train.FusedAdam([{"params": list(m.network.parameters()), "initial_lr": LR}], lr=LR, betas=(0.9, 0.999), eps=1e-7)
            sch = torch.optim.lr_scheduler.LambdaLR(opt, lam)
            pix_d, pr_d = pix.to(dev), pr.to(dev)
            for it, idx in enumerate(bl):
                idx = idx.to(dev)
                pic, rem = idx // (H * W), idx % (H * W)
                opt.zero_grad(set_to_none=True)
                if q7 is None:
                    m.check_resample = True  # (one host sync per step, like the reference) until the reference's exit condition shows
                    try:
                        Cc, Cf = m(rem // W, rem % W, pr_d[pic], K_inv)
                    except P.nerf.ResampleIndexError:
                        q7 = it
                        m.check_resample = False
                if q7 is not None:
                    Cc, Cf = m(rem // W, rem % W, pr_d[pic], K_inv)  # the device path clamps the index and goes on: what follows is beyond the reference's run
                loss = m.ray_loss(Cc, Cf, pix_d[idx])
                loss.backward()
                opt.step()
                sch.step()
                if (it + 1) % every == 0:
                    losses.append(round(float(loss.detach()), 1))
        # the dead state: every ReLU path off -> the same colour for every ray; on this problem its loss is ~1000 per 256-ray batch while a
        # learning run is below 300 after the first hundred steps
        stuck = len(losses) >= 3 and float(np.median(losses[-3:])) > 600.0
        print(f"{mode} seed {seed}: {' '.join(str(x) for x in losses)} | {'COLLAPSED' if stuck else 'learning'} | reference exit(0) condition (resample index out of range) {'never' if q7 is None else 'at iteration ' + str(q7)} | {time.perf_counter() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
