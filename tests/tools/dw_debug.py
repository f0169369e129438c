"""Debug: separate chain errors from weight-gradient-GEMM errors.  Recomputes every dW = G^T X in torch from the workspace's own
G / save buffers and compares with (a) what libnerf_hip wrote and (b) the oracle's autograd (coarse-only loss)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import nerf_oracle as O
import nerf_tiny_amd as P
from nerf_tiny_amd import _abi
dev = torch.device("cuda:0")
B, Nc, Nf = int(os.environ.get("DBG_B", "256")), 64, 128
row, col, pb, K, Ct = O.lego_inputs(B, seed=0)
w = O.make_weights(0)
m = P.NeRFModel(Nc, Nf, B); m.load_state_dict(w); m = m.to(dev)
Cc, Cf = m(row, col, pb, K)
loss = torch.sum(torch.square(Cc - Ct.to(dev)))
loss.backward()
torch.cuda.synchronize()
F = _abi.SAVE_FOR_BACKWARD
Mtot = B * (Nc + Nf); MS = Mtot + 64
view = lambda name, shape, dt=None: _abi.ws_view(m.last_workspace, B, Nc, Nf, F, name, shape, dt)
save = view("save", (10, MS, 256)); G = view("G", (9, MS, 256)); dz = view("dz", (Mtot, 4))
print("finite: save", bool(torch.isfinite(save[:, :Mtot]).all()), "G", bool(torch.isfinite(G[:, :Mtot]).all()), "dz", bool(torch.isfinite(dz).all()))
names = [k for k, _ in m.network.named_parameters()]
grads = {k: p.grad for k, p in m.network.named_parameters()}
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
X = lambda t: save[t, :Mtot].double(); Gt = lambda t: G[t, :Mtot].double()
chk = []
chk.append(("point_layer.0.0.weight", (Gt(0).T @ X(9))[:, :60]))
chk.append(("point_layer.0.0.bias", Gt(0).sum(0)))
for l in range(1, 8):
    full = Gt(l).T @ X(l - 1)
    if l == 4: full = torch.cat((full, (Gt(4).T @ X(9))[:, :60]), 1)
    chk.append((f"point_layer.{l}.0.weight", full)); chk.append((f"point_layer.{l}.0.bias", Gt(l).sum(0)))
# point_info folded into dir_info (csrc/common.h SEG_FOLD): M = dpre_dir^T h7, dW_pi = W_dir[:, 24:]^T M, dW_dir[:, 24:] = M W_pi^T + db_dir (x) b_pi
Wd, Wp, bp = w["network.dir_info.0.weight"].double().to(dev), w["network.point_info.weight"].double().to(dev), w["network.point_info.bias"].double().to(dev)
Mfold = (Gt(8).T @ X(7))[:128]; dbd = Gt(8).sum(0)[:128]
chk.append(("point_info.weight", Wd[:, 24:].T @ Mfold)); chk.append(("point_info.bias", Wd[:, 24:].T @ dbd))
chk.append(("dir_info.0.weight[:,24:]", Mfold @ Wp.T + torch.outer(dbd, bp))); chk.append(("dir_info.0.bias", dbd))
chk.append(("color_layer.0.weight", dz[:, :3].double().T @ X(8)[:, :128])); chk.append(("color_layer.0.bias", dz[:, :3].double().sum(0)))
chk.append(("sigma_layer.0.weight", (dz[:, 3:4].double().T @ X(7)))); chk.append(("sigma_layer.0.bias", dz[:, 3].double().sum().reshape(1)))
p = {k: v.clone().requires_grad_(True) for k, v in w.items()}
Ec, Ef = O.render(p, row, col, pb, K, Nc, Nf)
torch.sum(torch.square(Ec - Ct)).backward()
for name, want in chk:
    key = "network." + name.split("[")[0]
    got = grads[key.replace("network.", "")] if key.replace("network.", "") in grads else None
    got = grads[name.split("[")[0]]
    if "[:,24:]" in name: got = got[:, 24:]
    ref = p[key].grad
    if "[:,24:]" in name: ref = ref[:, 24:]
    print(f"{name:28s} lib vs torch(G^T X) {rel(got.reshape(want.shape), want):.2e}   torch(G^T X) vs oracle {rel(want.cpu(), ref.reshape(want.shape)):.2e}   lib vs oracle {rel(got.cpu().reshape(want.shape), ref.reshape(want.shape)):.2e}")
