"""Debug tool (GPU box): real training (the teacher/student scene of scripts/teacher_student.py) with EVERY step's weight / bias
gradients re-computed in torch from the workspace's own G / save / dz buffers (dW = G^T X, fp32 matmuls) and compared with what
libnerf_hip wrote.  A rare wrong launch of the weight-gradient kernels (a race, a stale operand) shows up as one step with a
large relative error; rounding differences stay below 1e-3.
Usage:  [BATCH=400] python tests/tools/train_dw_check.py [iterations] [seed]      (BATCH: rays per step, default 4096)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nerf_tiny_amd as P  # noqa: E402
from nerf_tiny_amd import _abi  # noqa: E402


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    if len(sys.argv) > 2:
        torch.manual_seed(int(sys.argv[2]))
    torch.backends.cuda.matmul.allow_tf32 = False
    dev = torch.device("cuda:0")
    H = W = 64
    B, Nc, Nf = int(os.environ.get("BATCH", "4096")), 64, 128
    scene = P.data.analytic_sphere_scene(n_pic=24, H=H, W=W, seed=5, device=dev)
    poses, imgs = scene.poses_bounds, scene.all_pix.view(24, H, W, 3)
    test_idx = np.arange(0, 24, 6)
    train_idx = np.setdiff1d(np.arange(24), test_idx)
    train = P.data.ArrayDataset(imgs[train_idx], poses[train_idx])
    test = P.data.ArrayDataset(imgs[test_idx], poses[test_idx])
    out_dir = os.path.join(ROOT, "gpurun_out", "train_dw_check") + "/"
    run = P.NeRFRunner(gpu=0, img_dir="", results_path=out_dir, ckpt_path=out_dir + "ck/", low_res=1, total_iter=iters, batch_ray=B,
                       learning=3e-4, lr_gamma=0.1, lr_milestone=[10, 200], n_coarse=Nc, n_fine=Nf, data_type="sync", step=10 ** 9,
                       decay_end=10 * iters, sched="EXP", continue_=False, datasets={"train": train, "val": train, "test": test},
                       log_every=10 ** 9)
    Mtot = B * (Nc + Nf)
    MS = Mtot + 64
    it, worst_all, bad = 0, 0.0, 0
    while it < iters:
        for row, col, pix, pb, pic in run.train_rays.epoch(B):
            run.optimizer.zero_grad(set_to_none=True)
            run.model.train()
            Cc, Cf = run.model(row, col, pb, run.K_inv)
            loss = run.model.ray_loss(Cc, Cf, pix)
            loss.backward()
            view = lambda name, shape, dt=None: _abi.ws_view(run.model.last_workspace, B, Nc, Nf, _abi.SAVE_FOR_BACKWARD, name, shape, dt)
            save, G, dz = view("save", (10, MS, 256)), view("G", (9, MS, 256)), view("dz", (Mtot, 4))
            X = lambda t: save[t, :Mtot]
            Gt = lambda t: G[t, :Mtot]
            g = {k: p.grad for k, p in run.model.network.named_parameters()}
            chk = [("point_layer.0.0.weight", g["point_layer.0.0.weight"], (Gt(0).T @ X(9))[:, :60]),
                   ("point_layer.0.0.bias", g["point_layer.0.0.bias"], Gt(0).sum(0))]
            for l in range(1, 8):
                full = Gt(l).T @ X(l - 1)
                if l == 4:
                    full = torch.cat((full, (Gt(4).T @ X(9))[:, :60]), 1)
                chk.append((f"point_layer.{l}.0.weight", g[f"point_layer.{l}.0.weight"], full))
                chk.append((f"point_layer.{l}.0.bias", g[f"point_layer.{l}.0.bias"], Gt(l).sum(0)))
            # point_info folded into dir_info (csrc/common.h SEG_FOLD)
            sd = run.model.network.state_dict()
            Wd, Wp, bp = sd["dir_info.0.weight"], sd["point_info.weight"], sd["point_info.bias"]
            Mfold, dbd = (Gt(8).T @ X(7))[:128], Gt(8).sum(0)[:128]
            chk.append(("point_info.weight", g["point_info.weight"], Wd[:, 24:].T @ Mfold))
            chk.append(("point_info.bias", g["point_info.bias"], Wd[:, 24:].T @ dbd))
            chk.append(("dir_info.0.weight[:,24:]", g["dir_info.0.weight"][:, 24:], Mfold @ Wp.T + torch.outer(dbd, bp)))
            chk.append(("dir_info.0.bias", g["dir_info.0.bias"], dbd))
            gd9 = G[8]
            raysum = gd9[:B * Nc].view(B, Nc, 256)[:, :, :128].sum(1) + gd9[B * Nc:Mtot].view(B, Nf, 256)[:, :, :128].sum(1)
            chk.append(("per-ray sums of dpre_dir (sbuf)", view("sbuf", (2, B, 128)).sum(0), raysum))
            chk.append(("dir_info.0.weight[:,:24]", g["dir_info.0.weight"][:, :24], raysum.T @ view("gdbuf", (B, 24))))
            chk.append(("color_layer.0.weight", g["color_layer.0.weight"], dz[:, :3].T @ X(8)[:, :128]))
            chk.append(("color_layer.0.bias", g["color_layer.0.bias"], dz[:, :3].sum(0)))
            chk.append(("sigma_layer.0.weight", g["sigma_layer.0.weight"], dz[:, 3:4].T @ X(7)))
            chk.append(("sigma_layer.0.bias", g["sigma_layer.0.bias"], dz[:, 3].sum().reshape(1)))
            worst, wname = 0.0, ""
            for name, got, want in chk:
                e = rel(got.reshape(want.shape), want)
                if not np.isfinite(e) or e > worst:
                    worst, wname = e, name
            worst_all = max(worst_all, worst) if np.isfinite(worst) else float("inf")
            if not np.isfinite(worst) or worst > 2e-3:
                bad += 1
                print(f"iter {it}: {wname} lib vs torch(G^T X) rel {worst:.3e}  loss {float(loss.detach()):.1f}", flush=True)
            if it % 100 == 0:
                print(f"iter {it} loss {float(loss.detach()):.1f} worst rel this step {worst:.2e} ({wname}); worst so far {worst_all:.2e}", flush=True)
            run.optimizer.step()
            run.scheduler.step()
            it += 1
            if it >= iters:
                break
    print(f"{it} training steps checked, {bad} steps above 2e-3, worst {worst_all:.3e}")


if __name__ == "__main__":
    main()
