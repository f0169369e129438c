"""Golden fixtures that pin the BACKWARD to the reference's own gradients.

Run ONLY in the build container (it needs /root/reference):

    cd /tmp && python /root/repo/tests/golden/make_decisions_golden.py

The reference's gradient is discontinuous in its fp32 MLP outputs (per-channel sorts, ReLU kinks on the un-detached t_fine path:
tests/test_gpu_backward.py's module docstring), so a stored gradient can only be compared sharply when the DISCRETE DECISIONS of the run
that produced it are stored with it.  For a 32-ray batch of cfg1 and of cfg4 this script runs the unmodified reference (forward, ray_loss,
backward: nerf.py:470-473) and records, from inside that same run:

  * the `torch.sort` indices of nerf.py:308 ([B, 192, 5], the five independent channel sorts, quirk Q1),
  * `index_fine` of nerf.py:248 ([B, 128], the resampling bins),
  * the ReLU sign bits of nerf.py:107-119 -- point_layer[0..7] (256 bits per sample and layer), dir_info (128 bits) -- and the sign inside
    sigma_layer's |.| (nerf.py:94,115), bit-packed (numpy.packbits, bitorder "little"), coarse pass then fine pass,
  * C_coarse, C_fine, the loss and all 24 gradients in full.

Capture is by observation only: forward hooks on the reference's own modules and recording wrappers around `torch.sort` /
`torch.searchsorted` that return the original results untouched.  Fixtures hold inputs and the REFERENCE's outputs (data, no code).
The oracle is run beside it and must reproduce every decision bit for bit (that is what pins the oracle's `stages`).
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

import nerf  # the reference

nerf.device = torch.device("cpu")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import nerf_oracle as O  # noqa: E402

torch.set_num_threads(8)
RAYS = 32


def reference_run_with_decisions(params, row, col, pb, K_inv, C_true, Nc, Nf):
    B = row.shape[0]
    m = nerf.NeRFModel(num_coarse=Nc, num_fine=Nf, batch_ray=B)
    m.load_state_dict(params, strict=True)
    rec = {"relu": [[] for _ in range(8)], "relu_dir": [], "sigma_sign": [], "sort": [], "index_fine": []}
    hooks = []
    for i in range(8):
        hooks.append(m.network.point_layer[i].register_forward_hook(lambda mod, inp, out, i=i: rec["relu"][i].append((out.detach() > 0).clone())))
    hooks.append(m.network.dir_info.register_forward_hook(lambda mod, inp, out: rec["relu_dir"].append((out.detach() > 0).clone())))
    hooks.append(m.network.sigma_layer[0].register_forward_hook(lambda mod, inp, out: rec["sigma_sign"].append((out.detach() >= 0).clone())))
    orig_sort, orig_ss = torch.sort, torch.searchsorted

    def sort_spy(*a, **k):
        r = orig_sort(*a, **k)
        rec["sort"].append(r[1].detach().clone())
        return r

    def ss_spy(*a, **k):
        r = orig_ss(*a, **k)
        rec["index_fine"].append((r.detach() - 1).clone())
        return r

    torch.sort, torch.searchsorted = sort_spy, ss_spy
    try:
        Cc, Cf = m(row, col, pb, K_inv)
        loss = m.ray_loss(Cc, Cf, C_true)
        loss.backward()
    finally:
        torch.sort, torch.searchsorted = orig_sort, orig_ss
        for h in hooks:
            h.remove()
    assert len(rec["sort"]) == 1 and len(rec["index_fine"]) == 1 and all(len(r) == 2 for r in rec["relu"])
    g = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    return Cc.detach(), Cf.detach(), loss.detach(), g, rec


def pack(bits):
    return np.packbits(bits.numpy().astype(np.uint8), axis=-1, bitorder="little")


def case(name, inputs, seed, sharp, Nc=64, Nf=128):
    row, col, pb, K_inv, C_true = [x[:RAYS] if (torch.is_tensor(x) and x.dim() > 0 and x.shape[0] != 3) else x for x in inputs]
    assert row.shape[0] == RAYS and K_inv.shape == (3, 3)
    params = O.make_weights(seed, sharp)
    Cc, Cf, loss, g, rec = reference_run_with_decisions(params, row, col, pb, K_inv, C_true, Nc, Nf)
    # the oracle beside it: same outputs, same decisions
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    st = {}
    oCc, oCf = O.render(p, row, col, pb, K_inv, Nc, Nf, stages=st)
    assert torch.equal(oCc.detach(), Cc) and torch.equal(oCf.detach(), Cf)
    assert torch.equal(st["perm"], rec["sort"][0]), "oracle sort decisions differ from the reference's"
    assert torch.equal(st["k"], rec["index_fine"][0]), "oracle resampling bins differ from the reference's"
    f_p, _ = O.frequencies()
    with torch.no_grad():
        for pts, n, which in ((st["pts_c"], Nc, 0), (st["pts_f"], Nf, 1)):
            _, _, hidden, _, c = O.mlp(params, O.encode(pts.detach(), f_p), st["gd"][:, None, :].expand(-1, n, -1), return_hidden=True)
            for i in range(8):
                assert torch.equal(hidden[i] > 0, rec["relu"][i][which]), (i, which)
            assert torch.equal(c > 0, rec["relu_dir"][which])
    print(f"[{name}] B={RAYS}: oracle reproduces the reference's outputs, sort indices, bins and ReLU masks bit for bit; loss {float(loss):.6f}")
    out = dict(meta_host=np.array(O.host_fingerprint()), row=row.numpy(), col=col.numpy(), poses_bound=pb.numpy(), K_inv=K_inv.numpy(),
               C_true=C_true.numpy(), seed=np.int64(seed), sharp=np.bool_(sharp), Nc=np.int64(Nc), Nf=np.int64(Nf),
               C_coarse=Cc.numpy(), C_fine=Cf.numpy(), loss=loss.numpy(),
               sort_index=rec["sort"][0].numpy().astype(np.uint8),          # [B, N, 5]: sorted position -> original index (< 192)
               index_fine=rec["index_fine"][0].numpy().astype(np.uint8),    # [B, Nf]
               relu_c=np.stack([pack(rec["relu"][i][0]) for i in range(8)]),   # [8, B, Nc, 32] bytes = 256 bits, little bit order
               relu_f=np.stack([pack(rec["relu"][i][1]) for i in range(8)]),   # [8, B, Nf, 32]
               relu_dir_c=pack(rec["relu_dir"][0]), relu_dir_f=pack(rec["relu_dir"][1]),  # [B, n, 16]
               sigma_nonneg_c=pack(rec["sigma_sign"][0].squeeze(-1)), sigma_nonneg_f=pack(rec["sigma_sign"][1].squeeze(-1)))  # [B, n/8]
    assert int(rec["sort"][0].max()) < 256 and int(rec["index_fine"][0].max()) < 256 and int(rec["index_fine"][0].min()) >= 0
    for k, v in g.items():
        out["grad_" + k] = v.numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    case("dec_cfg1_lego_crop32_r32", O.lego_inputs(1024, seed=0, crop=32), seed=1, sharp=True)
    case("dec_cfg4_fern_r32", O.fern_inputs(512, seed=3), seed=2, sharp=True)
