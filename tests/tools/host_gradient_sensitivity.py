import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import nerf_oracle as O
from conftest import load_golden, golden_inputs
torch.set_num_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
name = sys.argv[1]
g = load_golden(name); row, col, pb, K, Ct = golden_inputs(g)
w = O.make_weights(int(g["seed"]), bool(g["sharp"]))
Cc, Cf, loss, grads = O.loss_and_grads(w, row, col, pb, K, Ct, int(g["Nc"]), int(g["Nf"]))
print("threads", torch.get_num_threads(), "C_f max-rel vs golden", float((Cf - torch.from_numpy(g["C_fine"])).abs().max() / np.abs(g["C_fine"]).max()), "loss", float(loss), float(g["loss"]))
for k, v in list(grads.items())[:2] + list(grads.items())[14:18] + list(grads.items())[-2:]:
    print(f"{k:40s} |g| {float(v.double().norm()):.4e} golden {float(g['gnorm_' + k]):.4e}")
