"""A/B: register-resident vs LDS-tile field kernels on the same inputs (forward values and gradients)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import nerf_oracle as O
import nerf_tiny_amd as P
dev = torch.device("cuda:0")
B, Nc, Nf = 256, 64, 128
row, col, pb, K, Ct = O.fern_inputs(B, seed=3)
w = O.make_weights(2, sharp=True)
outs = []
for tile in (False, True):
    m = P.NeRFModel(Nc, Nf, B); m.load_state_dict(w); m = m.to(dev); m.force_tile_kernel = tile
    Cc, Cf = m(row, col, pb, K)
    loss = torch.sum(torch.square(Cc - Ct.to(dev)))
    loss.backward()
    outs.append((Cc.detach(), Cf.detach(), [p.grad.clone() for p in m.network.parameters()]))
a, b = outs
print("C_c max diff", float((a[0]-b[0]).abs().max()), "C_f", float((a[1]-b[1]).abs().max()))
for (k, _), x, y in zip(w.items(), a[2], b[2]):
    print(f"{k:40s} reg-vs-tile l2rel {float((x-y).norm()/y.norm()):.3e}")
