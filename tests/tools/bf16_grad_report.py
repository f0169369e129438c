"""Measured accuracy of the bf16-MLP variant against its emulation (oracle.mlp_bf16) and against fp32, for DESIGN.md section 7.
Usage (GPU box):  python tests/tools/bf16_grad_report.py [golden case]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nerf_oracle as O  # noqa: E402
import nerf_tiny_amd as P  # noqa: E402
from conftest import golden_inputs, load_golden  # noqa: E402


def rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30)), float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg1_lego_crop32"
    g = load_golden(name)
    row, col, pb, K, Ct = golden_inputs(g)
    Nc, Nf, B = int(g["Nc"]), int(g["Nf"]), row.shape[0]
    dev = torch.device("cuda:0")
    w = O.make_weights(int(g["seed"]), bool(g["sharp"]))
    m = P.NeRFModel(Nc, Nf, B)
    m.load_state_dict(w)
    m = m.to(dev)
    m.bf16_mlp = True
    for coarse_only in (True, False):
        for p in m.network.parameters():
            p.grad = None
        Cc, Cf = m(row.to(dev), col.to(dev), pb.to(dev), K)
        loss = torch.sum(torch.square(Cc - Ct.to(dev))) if coarse_only else m.ray_loss(Cc, Cf, Ct.to(dev))
        loss.backward()
        out = {}
        for label, mlp in (("emulation", O.mlp_bf16), ("fp32", O.mlp)):
            p = {k: v.clone().requires_grad_(True) for k, v in w.items()}
            Ec, Ef = O.render(p, row, col, pb, K, Nc, Nf, mlp=mlp, check=False)
            el = torch.sum(torch.square(Ec - Ct)) if coarse_only else O.ray_loss(Ec, Ef, Ct)
            el.backward()
            rs = [rel(pm.grad, pe.grad) for pe, pm in zip(p.values(), m.network.parameters())]
            out[label] = dict(C_c=float((Cc.detach().cpu() - Ec.detach()).abs().max() / Ec.detach().abs().max()),
                              C_f=float((Cf.detach().cpu() - Ef.detach()).abs().max() / Ef.detach().abs().max()),
                              loss_rel=abs(float(loss.detach()) - float(el.detach())) / abs(float(el.detach())),
                              grad_rel_max=max(r[0] for r in rs), grad_rel_median=sorted(r[0] for r in rs)[len(rs) // 2],
                              grad_cos_min=min(r[1] for r in rs))
        print(name, "coarse-only loss" if coarse_only else "full loss", {k: {a: float(f"{b:.3g}") for a, b in v.items()} for k, v in out.items()})


if __name__ == "__main__":
    main()
