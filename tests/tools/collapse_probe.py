"""Debug tool (GPU box): train the teacher/student scene of scripts/teacher_student.py step by step; when the loss jumps,
go back to the state before the jump and compare the library's gradients of that very batch with the oracle's autograd
(coarse-only loss: the well-conditioned comparison, 1e-4 per tensor on random weights).  Prints per-tensor relative errors.
Usage:  python tests/tools/collapse_probe.py [max_iters] [check_every]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nerf_oracle as O  # noqa: E402
import nerf_tiny_amd as P  # noqa: E402


def l2_rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def compare(model, batch, K_inv, tag):
    row, col, pix, pb, _ = batch
    dev = row.device
    w = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    inputs = (row.cpu(), col.cpu(), pb.cpu(), K_inv.cpu(), pix.cpu())
    for coarse_only in (True, False):
        m = P.NeRFModel(64, 128, row.shape[0])
        m.load_state_dict(model.state_dict())
        m = m.to(dev)
        Cc, Cf = m(row, col, pb, K_inv)
        loss = torch.sum(torch.square(Cc - pix)) if coarse_only else m.ray_loss(Cc, Cf, pix)
        loss.backward()
        p = {k: v.clone().requires_grad_(True) for k, v in w.items()}
        oCc, oCf = O.render(p, inputs[0], inputs[1], inputs[2], inputs[3], 64, 128)
        oloss = torch.sum(torch.square(oCc - inputs[4])) if coarse_only else O.ray_loss(oCc, oCf, inputs[4])
        oloss.backward()
        print(f"[{tag}] coarse_only={coarse_only}: loss lib {float(loss):.4f} oracle {float(oloss):.4f}  C_coarse rel {l2_rel(Cc, oCc):.2e} C_fine rel {l2_rel(Cf, oCf):.2e}")
        for (k, ref), q in zip(p.items(), m.network.parameters()):
            e = l2_rel(q.grad, ref.grad)
            flag = "  <-----" if (e > (1e-3 if coarse_only else 0.3) or not np.isfinite(e)) else ""
            print(f"    {k:28s} |g| lib {float(q.grad.norm()):.4e} oracle {float(ref.grad.norm()):.4e} rel {e:.2e}{flag}")


def main():
    max_iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    dev = torch.device("cuda:0")
    H = W = 64
    scene = P.data.analytic_sphere_scene(n_pic=24, H=H, W=W, seed=5, device=dev)
    poses, imgs = scene.poses_bounds, scene.all_pix.view(24, H, W, 3)
    test_idx = np.arange(0, 24, 6)
    train_idx = np.setdiff1d(np.arange(24), test_idx)
    train = P.data.ArrayDataset(imgs[train_idx], poses[train_idx])
    test = P.data.ArrayDataset(imgs[test_idx], poses[test_idx])
    out_dir = os.path.join(ROOT, "gpurun_out", "collapse_probe") + "/"
    run = P.NeRFRunner(gpu=0, img_dir="", results_path=out_dir, ckpt_path=out_dir + "ck/", low_res=1, total_iter=3000, batch_ray=4096,
                       learning=3e-4, lr_gamma=0.1, lr_milestone=[10, 200], n_coarse=64, n_fine=128, data_type="sync", step=10 ** 9,
                       decay_end=30000, sched="EXP", continue_=False, datasets={"train": train, "val": train, "test": test},
                       log_every=10 ** 9)
    it, hist = 0, []
    prev_state, prev_batch, ring = None, None, []
    while it < max_iters:
        for batch in run.train_rays.epoch(run.batch_ray):
            row, col, pix, pb, pic = batch
            if it >= 900:
                prev_state = {k: v.detach().clone() for k, v in run.model.state_dict().items()}
                prev_batch = batch
                ring.append((it, prev_state, prev_batch))
                ring[:] = ring[-12:]
            run.optimizer.zero_grad(set_to_none=True)
            run.model.train()
            Cc, Cf = run.model(row, col, pb, run.K_inv)
            loss = run.model.ray_loss(Cc, Cf, pix)
            loss.backward()
            if it >= 900:
                lv = float(loss.detach())
                gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in run.model.parameters())))
                hist.append(lv)
                med = float(np.median(hist[-50:]))
                if it % 50 == 0:
                    print(f"iter {it} loss {lv:.1f} |grad| {gn:.3e}", flush=True)
                if len(hist) > 20 and (lv > 3 * med or not np.isfinite(lv) or not np.isfinite(gn)):
                    print(f"JUMP at iter {it}: loss {lv:.1f} vs median {med:.1f}, |grad| {gn:.3e}; last losses {[round(x, 1) for x in hist[-6:]]}")
                    compare(run.model, batch, run.K_inv, "state AT the jump (weights that produced the large loss)")
                    it0, st0, b0 = ring[0]
                    run.model.load_state_dict(st0)
                    compare(run.model, b0, run.K_inv, f"state of iteration {it0} (before the rise)")
                    return
            run.optimizer.step()
            run.scheduler.step()
            it += 1
            if it >= max_iters:
                break
    print("no jump up to iteration", it)
    compare(run.model, prev_batch, run.K_inv, "final state")


if __name__ == "__main__":
    main()
