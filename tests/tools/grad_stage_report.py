import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nerf_oracle as O
import nerf_tiny_amd as P
from nerf_tiny_amd import _abi
import test_gpu_backward as T
from conftest import load_golden, golden_inputs
dev = torch.device("cuda:0")
name = sys.argv[1]; nr = int(sys.argv[2])
g = load_golden(name); inputs = golden_inputs(g)
inputs = tuple(x[:nr] if (torch.is_tensor(x) and x.dim() > 0 and x.shape[0] == inputs[0].shape[0]) else x for x in inputs)
Nc, Nf = int(g["Nc"]), int(g["Nf"]); B = inputs[0].shape[0]
w = O.make_weights(int(g["seed"]), bool(g["sharp"]))
p, st, _ = T._oracle_with_grads(O, w, inputs, Nc, Nf)
std = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in st.items()}
m, l = T._train_step(P, O, dev, w, inputs, Nc, Nf, ref_stages=std)
view = T._views(P, m, B, Nc, Nf)
def v(name, shape): return view(name, shape).cpu()
def rel(a, b): return float((a - b).norm() / b.norm())
print("dsig_f", rel(v("dsig_f", (B, Nf)), st["sig_f"].grad))
print("drgb_f", rel(v("drgb_f", (B, Nf, 3)), st["rgb_f"].grad))
print("dt_f  ", rel(v("dt_f", (B, Nf)), st["t_f"].grad))
print("dsig_c", rel(v("dsig_c", (B, Nc)), st["sig_c"].grad))
print("drgb_c", rel(v("drgb_c", (B, Nc, 3)), st["rgb_c"].grad))
a, b = v("dsig_c", (B, Nc)), st["sig_c"].grad
perray = ((a - b).norm(dim=1) / b.norm(dim=1))
print("dsig_c per-ray rel: median", float(perray.median()), "max", float(perray.max()), "n>1e-2", int((perray > 1e-2).sum()), "of", B)
r = int(perray.argmax()); print("worst ray", r, a[r, :8], b[r, :8])
a2, b2 = v("dt_f", (B, Nf)), st["t_f"].grad
pr2 = ((a2 - b2).norm(dim=1) / b2.norm(dim=1)); print("dt_f worst ray", int(pr2.argmax()), float(pr2.max()), "median", float(pr2.median()))
print("|dt_f| mean", float(b2.abs().mean()), " |dsig_c| mean", float(b.abs().mean()))
for (k, gg), pp in zip(p.items(), m.network.parameters()):
    print(f"{k:40s} l2rel {float((pp.grad.cpu()-gg.grad).norm()/gg.grad.norm()):.3e} |g| {float(gg.grad.norm()):.2e}")
