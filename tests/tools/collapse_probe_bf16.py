"""Debug tool (GPU box): the bf16-MLP trainer on the teacher/student scene step by step (same runner arguments as
scripts/teacher_student.py); per step: loss, |grad|, |Adam update|, share of layer-7 units that are dead on the batch.  When the loss jumps
the last steps are printed and the gradient of that very batch is recomputed three ways on the weights BEFORE the jump: bf16 kernels
(again), fp32 kernels, and the same comparison a few steps earlier -- a kernel fault shows as a bf16 / fp32 disagreement far above the
variant's usual 1e-2; an optimisation blow-up shows both agreeing on a large gradient.
Usage:  python tests/tools/collapse_probe_bf16.py [seed] [max_iters] [total_iter_for_the_lr_schedule]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nerf_tiny_amd as P  # noqa: E402


def grads_of(state, batch, K_inv, bf16, dev):
    row, col, pix, pb, _ = batch
    m = P.NeRFModel(64, 128, row.shape[0])
    m.load_state_dict(state)
    m = m.to(dev)
    m.bf16_mlp = bf16
    Cc, Cf = m(row, col, pb, K_inv)
    loss = m.ray_loss(Cc, Cf, pix)
    loss.backward()
    return float(loss), [p.grad.detach().double().clone() for p in m.network.parameters()]


def report(state, batch, K_inv, dev, tag):
    lb, gb = grads_of(state, batch, K_inv, True, dev)
    lb2, gb2 = grads_of(state, batch, K_inv, True, dev)
    lf, gf = grads_of(state, batch, K_inv, False, dev)
    nb = float(torch.sqrt(sum((g ** 2).sum() for g in gb)))
    nf = float(torch.sqrt(sum((g ** 2).sum() for g in gf)))
    d = float(torch.sqrt(sum(((a - b) ** 2).sum() for a, b in zip(gb, gf))))
    rep = float(torch.sqrt(sum(((a - b) ** 2).sum() for a, b in zip(gb, gb2))))
    cos = float(sum((a * b).sum() for a, b in zip(gb, gf)) / max(nb * nf, 1e-300))
    worst = max((float((a - b).norm() / b.norm().clamp_min(1e-300)), i) for i, (a, b) in enumerate(zip(gb, gf)))
    print(f"[{tag}] loss bf16 {lb:.2f} fp32 {lf:.2f} | |g| bf16 {nb:.4e} fp32 {nf:.4e} | bf16 vs fp32 L2-rel {d / max(nf, 1e-300):.3e} cosine {cos:.6f} | "
          f"worst tensor #{worst[1]} rel {worst[0]:.3e} | bf16 run-to-run {rep / max(nb, 1e-300):.1e} | finite {all(torch.isfinite(g).all() for g in gb)}", flush=True)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 13
    max_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    total = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    dev = torch.device("cuda:0")
    H = W = 64
    scene = P.data.analytic_sphere_scene(n_pic=24, H=H, W=W, seed=5, device=dev)
    poses, imgs = scene.poses_bounds, scene.all_pix.view(24, H, W, 3)
    test_idx = np.arange(0, 24, 6)
    train_idx = np.setdiff1d(np.arange(24), test_idx)
    train = P.data.ArrayDataset(imgs[train_idx], poses[train_idx])
    test = P.data.ArrayDataset(imgs[test_idx], poses[test_idx])
    out_dir = os.path.join(ROOT, "gpurun_out", "collapse_probe") + "/"
    run = P.NeRFRunner(gpu=0, img_dir="", results_path=out_dir, ckpt_path=out_dir + "ck/", low_res=1, total_iter=total, batch_ray=4096,
                       learning=3e-4, lr_gamma=0.1, lr_milestone=[10, 200], n_coarse=64, n_fine=128, data_type="sync", step=10 ** 9,
                       decay_end=10 * total, sched="EXP", continue_=False, datasets={"train": train, "val": train, "test": test},
                       log_every=10 ** 9, bf16_mlp=True)
    run.display(save=False)  # (teacher_student.py renders the held-out views before training: same generator state afterwards)
    it, hist, ring = 0, [], []
    while it < max_iters:
        for batch in run.train_rays.epoch(run.batch_ray):
            row, col, pix, pb, pic = batch
            state = {k: v.detach().clone() for k, v in run.model.state_dict().items()}
            run.optimizer.zero_grad(set_to_none=True)
            run.model.train()
            Cc, Cf = run.model(row, col, pb, run.K_inv)
            loss = run.model.ray_loss(Cc, Cf, pix)
            loss.backward()
            lv = float(loss.detach())
            gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in run.model.network.parameters())))
            before = [p.detach().clone() for p in run.model.network.parameters()]
            run.optimizer.step()
            run.scheduler.step()
            un = float(torch.sqrt(sum(((p.detach() - b).double() ** 2).sum() for p, b in zip(run.model.network.parameters(), before))))
            wn = float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in run.model.network.parameters())))
            rec = (it, lv, gn, un, wn)
            ring.append((rec, state, batch))
            ring[:] = ring[-40:]
            hist.append(lv)
            if it % 100 == 0:
                print(f"iter {it} loss {lv:.1f} |grad| {gn:.3e} |update| {un:.3e} |w| {wn:.3f}", flush=True)
            med = float(np.median(hist[-50:]))
            if len(hist) > 100 and (lv > 4 * med or not np.isfinite(lv) or not np.isfinite(gn)):
                print(f"JUMP at iter {it}: loss {lv:.1f} vs median {med:.1f}")
                for (i, l, g, u, w), _, _ in ring:
                    print(f"   iter {i} loss {l:.1f} |grad| {g:.3e} |update| {u:.3e} |w| {w:.4f}")
                for k in (-1, -2, -3, -6, -20):
                    (i, *_), st, b = ring[k]
                    report(st, b, run.K_inv, dev, f"weights before step {i}, batch of step {i}")
                return
            it += 1
            if it >= max_iters:
                break
    print("no jump up to iteration", it)


if __name__ == "__main__":
    main()
