"""Pins oracle/nerf_oracle.py against the imported reference and writes golden fixtures.

Run ONLY in the build container (it needs /root/reference; the GPU box has neither the
reference nor any need for this script):

    cd /tmp && python /root/repo/tests/golden/make_golden.py

The reference is imported unmodified (stub modules only for the absent ``imageio`` /
``tensorboard``, which are off the hot path), ``nerf.device`` is set to CPU, and the
weights of oracle.make_weights() are loaded with load_state_dict.  Fixtures hold inputs
and the REFERENCE's outputs (data only, no reference code).
"""
import os
import sys
import types

sys.modules["imageio"] = types.ModuleType("imageio")
_tb = types.ModuleType("torch.utils.tensorboard")
_tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _tb
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

import nerf  # the reference (seed_everything(624) runs at import, nerf.py:50)

nerf.device = torch.device("cpu")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import nerf_oracle as O  # noqa: E402

torch.set_num_threads(8)


def run_reference(params, row, col, pb, K_inv, C_true, Nc, Nf, grads=True):
    B = row.shape[0]
    m = nerf.NeRFModel(num_coarse=Nc, num_fine=Nf, batch_ray=B)
    m.load_state_dict(params, strict=True)
    if grads:
        Cc, Cf = m(row, col, pb, K_inv)
        loss = m.ray_loss(Cc, Cf, C_true)
        loss.backward()
        g = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        return Cc.detach(), Cf.detach(), loss.detach(), g
    with torch.no_grad():
        Cc, Cf = m(row, col, pb, K_inv)
    return Cc, Cf, None, None


def reference_stages(params, row, col, pb, K_inv, Nc, Nf):
    """Re-run the reference's own methods stage by stage to capture intermediates."""
    B = row.shape[0]
    m = nerf.NeRFModel(num_coarse=Nc, num_fine=Nf, batch_ray=B)
    m.load_state_dict(params, strict=True)
    st = {}
    with torch.no_grad():
        Kd = K_inv
        c2w, _, _, _, near, far = nerf.poses_extract(pb.to(torch.float))
        t_c = torch.tensor(np.linspace(tuple(near), tuple(far), Nc)).transpose(0, 1)
        # capture pts / dirs by wrapping the encoder
        cap = {}
        enc_fwd = m.encoder.forward

        def spy(num_points, point, dir):
            cap.setdefault("pts", []).append(point.clone())
            cap.setdefault("dir", []).append(dir.clone())
            out = enc_fwd(num_points, point, dir)
            cap.setdefault("gp", []).append(out[0].clone())
            cap.setdefault("gd", []).append(out[1].clone())
            return out

        m.encoder.forward = spy
        rgb_c, sig_c = m.net_out(t_c, row, col, c2w, Kd, Nc)
        delta_c = ((far - near) / Nc).unsqueeze(1).repeat(1, Nc)
        w_c = m.get_density(delta_c, sig_c.squeeze())
        t_f = m.resample(t_c, w_c)
        rgb_f, sig_f = m.net_out(t_f, row, col, c2w, Kd, Nf)
        st.update(t_c=t_c, pts_c=cap["pts"][0], d_wrd=cap["dir"][0][:, 0, :], gp_c=cap["gp"][0].flatten(2),
                  gd=cap["gd"][0][:, 0].flatten(1), rgb_c=rgb_c, sig_c=sig_c.squeeze(-1), w_c=w_c, t_f=t_f,
                  pts_f=cap["pts"][1], rgb_f=rgb_f, sig_f=sig_f.squeeze(-1))
    return st


def check_equal(name, a, b):
    a, b = np.asarray(a), np.asarray(b)
    same = np.array_equal(a, b)
    md = float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0
    print(f"  {name:10s} bit-identical={same}  max|diff|={md:.3e}")
    return same


def case(name, inputs, seed, sharp, Nc, Nf, store_full_grads, stage_rays=8):
    row, col, pb, K_inv, C_true = inputs
    params = O.make_weights(seed, sharp)
    print(f"[{name}] B={row.shape[0]} Nc={Nc} Nf={Nf} sharp={sharp}")
    Cc, Cf, loss, g = run_reference(params, row, col, pb, K_inv, C_true, Nc, Nf)
    oCc, oCf, oloss, og = O.loss_and_grads(params, row, col, pb, K_inv, C_true, Nc, Nf)
    ok = check_equal("C_coarse", Cc, oCc) & check_equal("C_fine", Cf, oCf) & check_equal("loss", loss, oloss)
    gmax = 0.0
    for k in g:
        rel = float((g[k] - og[k]).norm() / (g[k].norm() + 1e-30))
        gmax = max(gmax, rel)
    print(f"  grads      max L2-rel over 24 tensors = {gmax:.3e}")
    st = reference_stages(params, row, col, pb, K_inv, Nc, Nf)
    ost = {}
    with torch.no_grad():
        O.render(params, row, col, pb, K_inv, Nc, Nf, stages=ost)
    for k in ("t_c", "pts_c", "d_wrd", "gd", "sig_c", "rgb_c", "w_c", "t_f", "pts_f", "sig_f", "rgb_f"):
        ok &= check_equal(k, st[k], ost[k])
    ok &= check_equal("gp_c", st["gp_c"], O.encode(ost["pts_c"], O.frequencies()[0]))
    assert ok, f"oracle restatement is NOT bit-identical to the reference on case {name}"
    assert gmax < 1e-5
    out = dict(meta_host=np.array(O.host_fingerprint()), row=row.numpy(), col=col.numpy(), poses_bound=pb.numpy(), K_inv=K_inv.numpy(), C_true=C_true.numpy(),
               seed=np.int64(seed), sharp=np.bool_(sharp), Nc=np.int64(Nc), Nf=np.int64(Nf),
               C_coarse=Cc.numpy(), C_fine=Cf.numpy(), loss=loss.numpy())
    s = slice(0, stage_rays)
    for k in ("t_c", "pts_c", "d_wrd", "gd", "sig_c", "rgb_c", "w_c", "t_f", "pts_f", "sig_f", "rgb_f", "gp_c"):
        out["st_" + k] = st[k][s].numpy()
    for k, v in g.items():
        out["gnorm_" + k] = np.float64(v.double().norm().item())
        if store_full_grads or v.numel() <= 4096:
            out["grad_" + k] = v.numpy()
        else:
            out["gslice_" + k] = v.flatten()[::97].numpy().copy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {path} ({os.path.getsize(path)/1024:.0f} KiB)")


if __name__ == "__main__":
    f_p, f_d = O.frequencies()
    print("point f_l bits:", " ".join(f"{x:08x}" for x in f_p.numpy().view(np.uint32)))
    print("dir   f_l bits:", " ".join(f"{x:08x}" for x in f_d.numpy().view(np.uint32)))
    # cfg1: 32x32 centre crop, lego-like; full grads stored
    case("cfg1_lego_crop32", O.lego_inputs(1024, seed=0, crop=32), seed=0, sharp=False, Nc=64, Nf=128, store_full_grads=True)
    case("cfg1_lego_crop32_sharp", O.lego_inputs(1024, seed=0, crop=32), seed=1, sharp=True, Nc=64, Nf=128, store_full_grads=False)
    # cfg2: 4096 random rays
    case("cfg2_lego_rand4096", O.lego_inputs(4096, seed=0), seed=0, sharp=False, Nc=64, Nf=128, store_full_grads=False)
    # cfg4-like: fern, per-ray near/far (Q6)
    case("cfg4_fern_rand512", O.fern_inputs(512, seed=3), seed=2, sharp=True, Nc=64, Nf=128, store_full_grads=False)
    # small odd shapes (generality of Nc/Nf)
    case("small_16_32", O.lego_inputs(64, seed=5), seed=3, sharp=True, Nc=16, Nf=32, store_full_grads=False)
