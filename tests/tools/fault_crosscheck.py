"""Debug tool: is the device path's "resample index out of range" flag the reference's own condition on the SAME weights and batch?
  gpu    (GPU box)  trains collapse_stats.py's problem with the library until the flag is raised, saves weights + batch of that step
  oracle (anywhere) loads them and evaluates the oracle's forward with the reference's check (nerf.py:251-253)
Usage:  LR=1e-3 python tests/tools/fault_crosscheck.py gpu SEED   ->  gpurun_out/fault_SEED.pt ;   python tests/tools/fault_crosscheck.py oracle SEED
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import collapse_stats as CS  # noqa: E402
import nerf_oracle as O  # noqa: E402


def main():
    mode, seed = sys.argv[1], int(sys.argv[2])
    path = os.path.join(ROOT, "gpurun_out", f"fault_{seed}.pt")
    poses, imgs, K_inv = CS.scene()
    train_idx = np.setdiff1d(np.arange(CS.N_PIC), np.arange(0, CS.N_PIC, 6))
    pix = imgs[train_idx].reshape(-1, 3)
    pr = torch.from_numpy(poses[train_idx]).float()
    H, W, B, NC, NF = CS.H, CS.W, CS.B, CS.NC, CS.NF
    if mode == "gpu":
        import nerf_tiny_amd as P

        iters = 3000
        dev = torch.device("cuda:0")
        m = P.NeRFModel(NC, NF, B)
        m.load_state_dict(O.make_weights(seed))
        m = m.to(dev)
        opt = P.train.FusedAdam([{"params": list(m.network.parameters()), "initial_lr": CS.LR}], lr=CS.LR, betas=(0.9, 0.999), eps=1e-7)
        sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: CS.GAMMA ** (it / (10 * iters)))
        pix_d, pr_d = pix.to(dev), pr.to(dev)
        for it, idx in enumerate(CS.batches(seed, pix.shape[0], 800)):
            idd = idx.to(dev)
            pic, rem = idd // (H * W), idd % (H * W)
            opt.zero_grad(set_to_none=True)
            Cc, Cf = m(rem // W, rem % W, pr_d[pic], K_inv)
            if m.resample_fault():
                ws = m.last_workspace
                from nerf_tiny_amd import _abi
                w_c = _abi.ws_view(ws, B, NC, NF, _abi.SAVE_FOR_BACKWARD, "w_c", (B, NC)).cpu()
                torch.save({"it": it, "idx": idx, "weights": {k: v.detach().cpu() for k, v in m.state_dict().items()}, "w_c": w_c,
                            "C_fine": Cf.detach().cpu()}, path)
                print(f"gpu seed {seed}: flag at iteration {it}; rays with all-zero coarse weights: {int((w_c.abs().sum(1) == 0).sum())}, "
                      f"smallest row sum {float(w_c.sum(1).min()):.3e}")
                return
            m.ray_loss(Cc, Cf, pix_d[idd]).backward()
            opt.step()
            sch.step()
        print("no flag in 800 iterations")
    else:
        d = torch.load(path, weights_only=False)
        idx = d["idx"]
        pic, rem = idx // (H * W), idx % (H * W)
        params = {k: v.clone() for k, v in d["weights"].items()}
        stages = {}
        try:
            Cc, Cf = O.render(params, rem // W, rem % W, pr[pic], K_inv, NC, NF, check=True, stages=stages)
            print(f"oracle on the device run's weights of iteration {d['it']}: NO exit condition; max |C_fine - device| = {float((Cf - d['C_fine']).abs().max()):.3e}")
        except O.ResampleIndexError as e:
            print(f"oracle on the device run's weights of iteration {d['it']}: the reference's exit condition IS met ({e})")
        w_dev = d["w_c"]
        print(f"device coarse weights: rows with sum == 0: {int((w_dev.sum(1) == 0).sum())}; smallest row sums {sorted(w_dev.sum(1).tolist())[:3]}")


if __name__ == "__main__":
    main()
