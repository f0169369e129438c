"""Why do the device trainer and the oracle meet the reference's exit(0) condition (nerf.py:251-253) on DIFFERENT seeds and iterations from
identical weights and batches (profiles/r03_training_stability.json: device seed 4 at iteration 19, the oracle never; oracle seed 12 at
265, the device never)?  Step by step, on tests/tools/collapse_stats.py's problem (lr 1e-3, the reference runner's default):
  teacher-forced   at the DEVICE trainer's current weights, the oracle (autograd through the CPU restatement of the reference) evaluates the
                   same batch: loss, all 24 gradients, and the exit condition.  The full-loss gradient is ill-conditioned (sort ties and
                   ReLU kinks on the t_fine path), so the device's distance is printed beside the ORACLE'S OWN: its
                   gradient at the same weights moved by a seeded relative 1e-6 (two seeds, the larger distance);
  free-running     the oracle + torch.optim.Adam trained on its own from the same start: how fast two CORRECT trainers drift apart.
One JSON line per step, then a summary line.  Test infrastructure (imports oracle/): runs on the GPU box.
Usage:  python tests/tools/collapse_gradient_diff.py [SEED=4] [STEPS=25]
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import collapse_stats as CS  # noqa: E402  (scene, batches, sizes; puts the repo root and oracle/ on the path)

O = CS.O
ITERS_OF_THE_SCHEDULE = 800  # the runs of profiles/r03_training_stability.json: "first 800 iterations"
LR = 1e-3


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    import nerf_tiny_amd as P

    torch.manual_seed(seed)
    poses, imgs, K_inv = CS.scene()
    import numpy as np

    train_idx = np.setdiff1d(np.arange(CS.N_PIC), np.arange(0, CS.N_PIC, 6))
    pix = imgs[train_idx].reshape(-1, 3)
    pr = torch.from_numpy(poses[train_idx]).float()
    w0 = O.make_weights(seed)
    bl = CS.batches(seed, pix.shape[0], ITERS_OF_THE_SCHEDULE)[:steps]
    lam = lambda it: CS.GAMMA ** (it / (10 * ITERS_OF_THE_SCHEDULE))  # noqa: E731

    dev = torch.device("cuda:0")
    m = P.NeRFModel(CS.NC, CS.NF, CS.B)
    m.load_state_dict(w0)
    m = m.to(dev)
    names = [k for k, _ in m.named_parameters()]  # "network.point_layer.0.0.weight", ...: the oracle's keys (oracle.make_weights)
    opt = P.train.FusedAdam([{"params": list(m.network.parameters()), "initial_lr": LR}], lr=LR, betas=(0.9, 0.999), eps=1e-7)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lam)
    pix_d, pr_d = pix.to(dev), pr.to(dev)

    free = {k: v.clone().requires_grad_(True) for k, v in w0.items()}
    fopt = torch.optim.Adam(list(free.values()), lr=LR, betas=(0.9, 0.999), eps=1e-7)
    fsch = torch.optim.lr_scheduler.LambdaLR(fopt, lam)
    free_dead = None

    worst_grad, worst_loss, first_dev_fault, first_tf_fault = 0.0, 0.0, None, None
    ratios = []
    for it, idx in enumerate(bl):
        pic, rem = idx // (CS.H * CS.W), idx % (CS.H * CS.W)
        row, col = rem // CS.W, rem % CS.W
        # ---- the device step (its fault flag read every step, like the reference checks every forward)
        opt.zero_grad(set_to_none=True)
        Cc, Cf = m(row.to(dev), col.to(dev), pr_d[pic.to(dev)], K_inv)
        loss = m.ray_loss(Cc, Cf, pix_d[idx.to(dev)])
        loss.backward()
        dev_fault = bool(m.resample_fault())
        g_dev = [p.grad.detach().cpu().clone() for p in m.network.parameters()]
        w_dev = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        # ---- teacher-forced: the oracle at the device's weights, same batch
        tf = {k: v.clone().requires_grad_(True) for k, v in w_dev.items()}
        tf_fault = False
        try:
            with torch.no_grad():
                O.render(tf, row, col, pr[pic], K_inv, CS.NC, CS.NF, check=True)
        except O.ResampleIndexError:
            tf_fault = True
        Cc_o, Cf_o = O.render(tf, row, col, pr[pic], K_inv, CS.NC, CS.NF, check=False)
        loss_o = O.ray_loss(Cc_o, Cf_o, pix[idx])
        g_o = torch.autograd.grad(loss_o, [tf[n] for n in names])
        # the oracle's own sensitivity: the same evaluation at weights moved by a relative 1e-6 (another correct fp32 evaluation of the step)
        own = 0.0
        for js in (1, 2):
            gj = torch.Generator().manual_seed(100 * it + js)
            tj = {k: (v * (1.0 + 1e-6 * (2.0 * torch.rand(v.shape, generator=gj) - 1.0))).requires_grad_(True) for k, v in w_dev.items()}
            Cc_j, Cf_j = O.render(tj, row, col, pr[pic], K_inv, CS.NC, CS.NF, check=False)
            g_j = torch.autograd.grad(O.ray_loss(Cc_j, Cf_j, pix[idx]), [tj[n] for n in names])
            own = max(own, rel(torch.cat([g.reshape(-1) for g in g_j]), torch.cat([g.reshape(-1) for g in g_o])))
        flat_d, flat_o = torch.cat([g.reshape(-1) for g in g_dev]), torch.cat([g.reshape(-1) for g in g_o])
        per = {n: rel(a, b) for n, a, b in zip(names, g_dev, g_o) if float(b.norm()) > 0}
        wn = max(per, key=per.get) if per else None
        # ---- free-running oracle trainer (distance of the two weight sets BEFORE this step's updates)
        drift = rel(torch.cat([w_dev[k].reshape(-1) for k in w_dev]), torch.cat([free[k].detach().reshape(-1) for k in w_dev]))
        free_loss = None
        if free_dead is None:
            fopt.zero_grad(set_to_none=True)
            try:
                Cc_f, Cf_f = O.render(free, row, col, pr[pic], K_inv, CS.NC, CS.NF, check=True)
                fl = O.ray_loss(Cc_f, Cf_f, pix[idx])
                fl.backward()
                fopt.step()
                fsch.step()
                free_loss = float(fl.detach())
            except O.ResampleIndexError:
                free_dead = it
        rec = {"it": it, "loss_device": round(float(loss.detach()), 4), "loss_oracle_at_device_weights": round(float(loss_o.detach()), 4),
               "loss_rel": abs(float(loss.detach()) - float(loss_o.detach())) / max(abs(float(loss_o.detach())), 1e-30),
               "grad_l2_rel_all": rel(flat_d, flat_o), "oracle_own_grad_l2_rel_under_1e-6_jitter": own, "grad_worst_tensor": wn, "grad_worst_tensor_rel": per.get(wn) if wn else None,
               "exit_condition_device": dev_fault, "exit_condition_oracle_at_device_weights": tf_fault,
               "loss_oracle_free_running": free_loss, "weights_rel_distance_device_vs_free_oracle": drift}
        print(json.dumps(rec), flush=True)
        worst_grad, worst_loss = max(worst_grad, rec["grad_l2_rel_all"]), max(worst_loss, rec["loss_rel"])
        ratios.append(rec["grad_l2_rel_all"] / max(own, 1e-30))
        if dev_fault and first_dev_fault is None:
            first_dev_fault = it
        if tf_fault and first_tf_fault is None:
            first_tf_fault = it
        opt.step()
        sch.step()
    print(json.dumps({"summary": {"seed": seed, "steps": steps, "lr": LR, "worst_grad_l2_rel_all": worst_grad, "worst_loss_rel": worst_loss,
                                  "device_distance_over_oracle_own_distance": {"median": float(np.median(ratios)), "max": float(max(ratios))},
                                  "first_exit_condition_device": first_dev_fault, "first_exit_condition_oracle_at_device_weights": first_tf_fault,
                                  "free_running_oracle_met_exit_condition_at": free_dead}}), flush=True)


if __name__ == "__main__":
    main()
