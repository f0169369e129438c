"""GPU: the split-fp32 inference mode (NERF_HIP_SPLIT_MLP, model.split_mlp; csrc/field_fwd_split.hip) -- the fp32 MLP evaluated on bf16
MFMA with every fp32 operand split into two bf16 parts (hi + mid) and three MFMAs per product, fp32 accumulation.  It is held to the
SAME bar as the exact-fp32 default: <= 1e-4 max-rel against the reference's own outputs (golden fixtures) and against the oracle."""
import numpy as np
import pytest
import torch

from conftest import elementwise_rel, golden_inputs, load_golden, max_rel

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model(pkg, oracle, g, dev, B):
    w = oracle.make_weights(int(g["seed"]), bool(g["sharp"]))
    m = pkg.NeRFModel(int(g["Nc"]), int(g["Nf"]), B)
    m.load_state_dict(w)
    return w, m.to(dev)


@pytest.mark.parametrize("name", ["cfg1_lego_crop32", "cfg1_lego_crop32_sharp", "cfg2_lego_rand4096", "cfg4_fern_rand512", "small_16_32"])
def test_split_forward_matches_the_reference_outputs(oracle, pkg, dev, name):
    """end to end against what /root/reference/nerf.py itself returned for these inputs and weights (tests/golden/make_golden.py)"""
    g = load_golden(name)
    row, col, pb, K, _ = golden_inputs(g)
    w, m = _model(pkg, oracle, g, dev, row.shape[0])
    with torch.no_grad():
        Cc0, Cf0 = m(row, col, pb, K)          # exact-fp32 default
        m.split_mlp = True
        Cc, Cf = m(row, col, pb, K)
    ec, ef = max_rel(Cc, g["C_coarse"]), max_rel(Cf, g["C_fine"])
    print(f"{name}: split vs reference  C_coarse {ec:.2e}  C_fine {ef:.2e}   (exact-fp32 kernels: {max_rel(Cc0, g['C_coarse']):.2e} / {max_rel(Cf0, g['C_fine']):.2e})")
    assert ec < TOL and ef < TOL
    assert elementwise_rel(Cf, g["C_fine"]) < 5 * TOL
    mse = float(((Cf.cpu().double() - torch.from_numpy(g["C_fine"]).double()) ** 2).mean())
    assert 10 * np.log10(1.0 / max(mse, 1e-300)) > 80.0          # PSNR vs ref
    assert not torch.equal(Cf, Cf0)                                # the flag really selects another kernel


@pytest.mark.parametrize("B,Nc,Nf", [(7, 5, 3), (33, 100, 200), (130, 31, 65), (3, 1024, 1024)])
def test_split_forward_ragged_and_maximum_sizes(oracle, pkg, dev, B, Nc, Nf):
    """sample counts that are no multiple of the 32-sample wave tile / the 128-sample workgroup, tiles straddling rays, maximum sizes"""
    row, col, pb, K, _ = oracle.fern_inputs(B, seed=11)
    w = oracle.make_weights(5, sharp=True)
    m = pkg.NeRFModel(Nc, Nf, B)
    m.load_state_dict(w)
    m = m.to(dev)
    m.split_mlp = True
    with torch.no_grad():
        Cc, Cf = m(row, col, pb, K)
        oc, of = oracle.render(w, row, col, pb, K, Nc, Nf, check=False)
    assert torch.isfinite(Cc).all() and torch.isfinite(Cf).all()
    assert max_rel(Cc, oc) < TOL and max_rel(Cf, of) < TOL


def test_split_mlp_alone_leaves_training_exact(oracle, pkg, dev):
    """`split_mlp` is the INFERENCE switch: a forward that records a graph runs the exact-fp32 training kernels whatever it says
    (bit-identical loss and gradients); only `split_train` puts the training forward on the split kernel."""
    g = load_golden("small_16_32")
    row, col, pb, K, Ct = golden_inputs(g)
    w, m = _model(pkg, oracle, g, dev, row.shape[0])

    def step():
        for p in m.network.parameters():
            p.grad = None
        Cc, Cf = m(row, col, pb, K)
        loss = m.ray_loss(Cc, Cf, Ct.to(dev))
        loss.backward()
        return float(loss.detach()), [p.grad.clone() for p in m.network.parameters()]

    l0, g0 = step()
    m.split_mlp = True
    l1, g1 = step()
    assert l0 == l1 and all(torch.equal(a, b) for a, b in zip(g0, g1))
    m.split_train = True
    l2, g2 = step()
    assert l2 != l0 and abs(l2 - l0) <= 1e-5 * abs(l0)  # another forward kernel, the same loss to 1e-5


def _saved(pkg, m, B, Nc, Nf, flags):
    from nerf_tiny_amd import _abi

    ws = m.last_workspace
    M = B * (Nc + Nf)
    tiles = (B * Nc + 63) // 64 + (B * Nf + 63) // 64
    save = _abi.ws_view(ws, B, Nc, Nf, flags, "save", (10, M + 64, 256)).clone()
    masks = _abi.ws_view(ws, B, Nc, Nf, flags, "masks", (8, tiles, 4, 256), torch.int16).clone()
    spre = _abi.ws_view(ws, B, Nc, Nf, flags, "spre", (M,)).clone()
    return save[:, :M], masks, spre


@pytest.mark.parametrize("name,rays", [("cfg1_lego_crop32_sharp", 256), ("cfg4_fern_rand512", 200), ("small_16_32", 50)])
def test_split_training_forward_saves_what_the_exact_forward_saves(oracle, pkg, dev, name, rays):
    """NERF_HIP_SPLIT_MLP | NERF_HIP_SAVE_FOR_BACKWARD: the split kernel leaves the fp32 rows of gamma_p / h0..h7 / c, the ReLU mask words
    and the sigma pre-activation exactly where k_field_fwd_reg<SAVE> leaves them (ragged row counts included: 200 rays x 64 / 128 samples
    and 50 x 16 / 32 are no multiples of the 128-sample workgroup).  gamma_p is bit-identical (same fp32 encodings); the activations agree
    to the split arithmetic's 2^-16 per product; mask bits differ only where a pre-activation sits within that distance of zero."""
    from nerf_tiny_amd import _abi

    g = load_golden(name)
    row, col, pb, K, Ct = (x[:rays] if (torch.is_tensor(x) and x.dim() > 0 and x.shape[0] > 3) else x for x in golden_inputs(g))
    Nc, Nf = int(g["Nc"]), int(g["Nf"])
    w, m = _model(pkg, oracle, g, dev, rays)
    Cc0, Cf0 = m(row, col, pb, K)
    s0, k0, p0 = _saved(pkg, m, rays, Nc, Nf, _abi.SAVE_FOR_BACKWARD)
    m.split_train = True
    Cc1, Cf1 = m(row, col, pb, K)
    s1, k1, p1 = _saved(pkg, m, rays, Nc, Nf, _abi.SAVE_FOR_BACKWARD | _abi.SPLIT_MLP)
    assert max_rel(Cc1, Cc0) < 2e-5 and max_rel(Cf1, Cf0) < TOL
    # compared on the COARSE pass's rows (first B * Nc rows / whole 64-sample mask tiles of every buffer): its inputs are identical in both
    # runs; the fine pass starts from t_fine, which already carries the coarse pass's 1e-5 through a 3217 rad/unit encoding
    Mc, tc = rays * Nc, (rays * Nc) // 64
    assert torch.equal(s1[9, :Mc, :64], s0[9, :Mc, :64])  # gamma_p (tensor S_GP, 64 columns): the same fp32 encodings, bit for bit
    for t in range(8):  # h0..h7
        assert float((s1[t, :Mc] - s0[t, :Mc]).abs().max()) <= 1e-4 * float(s0[t, :Mc].abs().max()) + 1e-6, t
    assert float((s1[8, :Mc, :128] - s0[8, :Mc, :128]).abs().max()) <= 1e-4 * float(s0[8, :Mc].abs().max()) + 1e-6  # c
    assert float((p1[:Mc] - p0[:Mc]).abs().max()) <= 1e-4 * float(p0[:Mc].abs().max()) + 1e-6
    diff = (k1[:, :tc] ^ k0[:, :tc]).contiguous().view(torch.uint8)
    flipped = sum(int(((diff >> b) & 1).sum()) for b in range(8))
    total = k0[:, :tc].numel() * 16
    print(f"{name}: split training forward: {flipped} of {total} coarse-pass mask bits differ from the exact forward's ({flipped / total:.2e})")
    assert flipped <= 2e-4 * total
    # the fine pass: same buffers, finite and of the same size as the exact forward's (its end-to-end check is test_split_train_step_end_to_end)
    assert torch.isfinite(s1).all() and abs(float(s1[7].abs().mean()) / float(s0[7].abs().mean()) - 1.0) < 1e-2


@pytest.mark.parametrize("name", ["cfg1_lego_crop32", "cfg1_lego_crop32_sharp", "cfg4_fern_rand512", "small_16_32", "cfg2_lego_rand4096"])
def test_split_train_step_end_to_end(oracle, pkg, dev, name):
    """VERDICT round 4 item 8 (stretch, opt-in): the train step with its forward on the split-fp32 kernel (model.split_train), the backward on the
    exact-fp32 kernels.  Held to what the exact path is held to in tests/test_gpu_backward.py::test_train_step_end_to_end: loss to 1e-5, every
    gradient tensor within twice the oracle's OWN shift under a seeded 1e-6 relative weight perturbation (floor 1e-3) -- the reference's
    gradient moves 10-17 % under such a jitter, a 1e-5-accurate forward sits far inside."""
    from test_gpu_backward import _oracle_grads, _sensitivity_band

    from conftest import l2_rel

    g = load_golden(name)
    inputs = golden_inputs(g)
    row, col, pb, K, Ct = inputs
    Nc, Nf = int(g["Nc"]), int(g["Nf"])
    w, m = _model(pkg, oracle, g, dev, row.shape[0])
    m.split_train = True
    Cc, Cf, loss = m.train_step(row, col, pb, K, Ct)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert max_rel(Cc, g["C_coarse"]) < TOL and max_rel(Cf, g["C_fine"]) < TOL
    _, g0 = _oracle_grads(oracle, w, inputs, Nc, Nf)
    band = _sensitivity_band(oracle, w, inputs, Nc, Nf, g0, seeds=(1,) if name.startswith("cfg2") else (1, 2))
    worst = (0.0, 1.0, "")
    for k, q in m.named_parameters():
        key = k if k.startswith("network.") else "network." + k
        assert torch.isfinite(q.grad).all(), k
        e = l2_rel(q.grad, g0[key])
        bar = max(2.0 * band[key], 1e-3)
        if e / bar > worst[0] / worst[1]:
            worst = (e, bar, k)
        assert e < bar, (k, e, bar)
    print(f"{name}: split-forward train step: closest to its bar: {worst[2]} L2-rel {worst[0]:.2e} (bar {worst[1]:.2e})")
    # the autograd surface takes the same path: identical loss and gradients
    grads = [p.grad.clone() for p in m.network.parameters()]
    for p in m.network.parameters():
        p.grad = None
    Cc2, Cf2 = m(row, col, pb, K)
    l2 = m.ray_loss(Cc2, Cf2, Ct.to(dev))
    l2.backward()
    assert float(l2.detach()) == float(loss)
    assert all(torch.equal(a, p.grad) for a, p in zip(grads, m.network.parameters()))
