"""Development aid: the split-fp32 train step's intermediate gradients against the exact fp32 path's, tensor by tensor (GPU).
usage: python tests/tools/split_train_debug.py [fixture] [rays]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nerf_oracle as O  # noqa: E402
import nerf_tiny_amd as P  # noqa: E402
from conftest import golden_inputs, load_golden  # noqa: E402
from nerf_tiny_amd import _abi  # noqa: E402
from test_gpu_split import BG_KS, _decode_pieces, _wave_blocks  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "small_16_32"
g = load_golden(name)
inp = golden_inputs(g)
rays = int(sys.argv[2]) if len(sys.argv) > 2 else inp[0].shape[0]
row, col, pb, K, Ct = (x[:rays] if (torch.is_tensor(x) and x.dim() > 0 and x.shape[0] > 3) else x for x in inp)
Nc, Nf = int(g["Nc"]), int(g["Nf"])
dev = torch.device("cuda:0")
w = O.make_weights(int(g["seed"]), bool(g["sharp"]))
M, Mc = rays * (Nc + Nf), rays * Nc


def run(split):
    m = P.NeRFModel(Nc, Nf, rays)
    m.load_state_dict(w)
    m = m.to(dev)
    m.split_train = split
    Cc, Cf, loss = m.train_step(row, col, pb, K, Ct)
    torch.cuda.synchronize()
    return m, float(loss)


m0, l0 = run(False)
m1, l1 = run(True)
print("loss", l0, l1)
f0, f1 = _abi.SAVE_FOR_BACKWARD, _abi.SAVE_FOR_BACKWARD | _abi.SPLIT_MLP
v0 = lambda n, sh, dt=None: _abi.ws_view(m0.last_workspace, rays, Nc, Nf, f0, n, sh, dt).cpu()
v1 = lambda n, sh, dt=None: _abi.ws_view(m1.last_workspace, rays, Nc, Nf, f1, n, sh, dt)
for n, sh in (("dsig_f", (rays, Nf)), ("drgb_f", (rays, Nf, 3)), ("dt_f", (rays, Nf)), ("dsig_c", (rays, Nc)), ("drgb_c", (rays, Nc, 3)), ("spre", (M,))):
    a, b = v0(n, sh), v1(n, sh).cpu()
    print(f"{n:8s} rel diff {float((a - b).norm() / a.norm().clamp_min(1e-30)):.3e}")
G = v0("G", (9, M + 64, 256))
wb_c, wb_tot = _wave_blocks(rays, Nc), _wave_blocks(rays, Nc) + _wave_blocks(rays, Nf)
nb = wb_tot * sum(BG_KS) * 1024
gh, gm = v1("bG", (nb,), torch.uint8), v1("bG2", (nb,), torch.uint8)
for t in range(9):
    cols = 256 if t < 8 else 128
    for nm, wb0, nwb, r0, rows in (("coarse", 0, (Mc + 31) // 32, 0, Mc), ("fine", wb_c, (M - Mc + 31) // 32, Mc, M - Mc)):
        got = (_decode_pieces(gh, wb_tot, BG_KS, t, wb0, nwb) + _decode_pieces(gm, wb_tot, BG_KS, t, wb0, nwb))[:rows, :cols]
        ref = G[t, r0:r0 + rows, :cols]
        print(f"G tensor {t} {nm:6s}: rel diff {float((got - ref).norm() / ref.norm().clamp_min(1e-30)):.3e}   |ref| {float(ref.norm()):.3e}")
for (k, p0), p1 in zip(m0.network.named_parameters(), m1.network.parameters()):
    print(f"{k:34s} {float((p0.grad - p1.grad).norm() / p0.grad.norm().clamp_min(1e-30)):.3e}")
