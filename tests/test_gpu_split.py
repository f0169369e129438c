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
    (bit-identical loss and gradients); only `split_train` puts the train step on the split kernels."""
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
    assert l2 != l0 and abs(l2 - l0) <= 1e-5 * abs(l0)  # other kernels, the same loss to 1e-5


def _decode_pieces(buf, wb_tot, ks_list, tensor, wb0, nwb):
    """fragment layout (csrc/bf16_common.h) -> [nwb * 32 samples, 16 ks features] fp32: piece (wb, ks), lane (j, h), slot s holds feature
    16 ks + 4 h + (s & 3) + 8 (s >> 2) of sample wb * 32 + j"""
    ks_t = ks_list[tensor]
    start = wb_tot * 1024 * sum(ks_list[:tensor])
    raw = buf[start:start + wb_tot * ks_t * 1024].view(torch.bfloat16).view(wb_tot, ks_t, 2, 32, 8)[wb0:wb0 + nwb].float().cpu()
    out = torch.zeros(nwb, 32, ks_t * 16)
    for ks in range(ks_t):
        for h in range(2):
            for sl in range(8):
                out[:, :, 16 * ks + 4 * h + (sl & 3) + 8 * (sl >> 2)] = raw[:, ks, h, :, sl]
    return out.reshape(nwb * 32, ks_t * 16)


BS_KS = [4] + [16] * 8 + [8, 2]       # gamma_p, h0..h7, c, gamma_d   (csrc/bf16_common.h)
BG_KS = [16] * 8 + [8, 2]             # dpre0..7, dpre_dir, (dz, dspre)


def _wave_blocks(B, N):
    return ((B * N + 255) // 256) * 8


@pytest.mark.parametrize("name,rays", [("cfg1_lego_crop32_sharp", 256), ("small_16_32", 50)])
def test_split_training_forward_saves_hi_plus_mid(oracle, pkg, dev, name, rays):
    """Stage 1 of the split-fp32 train step: the forward leaves every layer input as TWO bf16 fragment-layout tensors whose sum is the fp32
    value to 2^-16 (hi = bf16(x), mid = bf16(x - hi)).  Compared on the COARSE pass (identical inputs in both runs) with the exact forward's fp32
    rows: gamma_p / h0..h7 / c to 1e-4 of the tensor's size; the sigma pre-activation likewise; ragged row counts included (50 x 16 samples)."""
    from nerf_tiny_amd import _abi

    g = load_golden(name)
    row, col, pb, K, Ct = (x[:rays] if (torch.is_tensor(x) and x.dim() > 0 and x.shape[0] > 3) else x for x in golden_inputs(g))
    Nc, Nf = int(g["Nc"]), int(g["Nf"])
    M, Mc = rays * (Nc + Nf), rays * Nc
    w, m = _model(pkg, oracle, g, dev, rays)
    Cc0, Cf0 = m(row, col, pb, K)
    f0 = _abi.SAVE_FOR_BACKWARD
    save = _abi.ws_view(m.last_workspace, rays, Nc, Nf, f0, "save", (10, M + 64, 256)).clone().cpu()
    spre0 = _abi.ws_view(m.last_workspace, rays, Nc, Nf, f0, "spre", (M,)).clone().cpu()
    m.split_train = True
    Cc1, Cf1 = m(row, col, pb, K)
    f1 = _abi.SAVE_FOR_BACKWARD | _abi.SPLIT_MLP
    wb_c, wb_tot = _wave_blocks(rays, Nc), _wave_blocks(rays, Nc) + _wave_blocks(rays, Nf)
    nbytes = wb_tot * sum(BS_KS) * 1024
    hi = _abi.ws_view(m.last_workspace, rays, Nc, Nf, f1, "bsave", (nbytes,), torch.uint8)
    mid = _abi.ws_view(m.last_workspace, rays, Nc, Nf, f1, "bsave2", (nbytes,), torch.uint8)
    spre1 = _abi.ws_view(m.last_workspace, rays, Nc, Nf, f1, "spre", (M,)).clone().cpu()
    assert max_rel(Cc1, Cc0) < 2e-5 and max_rel(Cf1, Cf0) < TOL
    nwb = (Mc + 31) // 32
    # (tensor index in the fragment buffers, tensor index in the fp32 row buffer, columns)
    for tb, tf, cols in [(0, 9, 60)] + [(1 + l, l, 256) for l in range(8)] + [(9, 8, 128)]:
        got = (_decode_pieces(hi, wb_tot, BS_KS, tb, 0, nwb) + _decode_pieces(mid, wb_tot, BS_KS, tb, 0, nwb))[:Mc, :cols]
        ref = save[tf, :Mc, :cols]
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        assert err <= (3e-5 if tb == 0 else 1e-4) * scale + 1e-7, (tb, err, scale)
    assert float((spre1[:Mc] - spre0[:Mc]).abs().max()) <= 1e-4 * float(spre0[:Mc].abs().max()) + 1e-6


@pytest.mark.parametrize("name", ["cfg1_lego_crop32", "cfg1_lego_crop32_sharp", "cfg4_fern_rand512", "small_16_32", "cfg2_lego_rand4096"])
def test_split_train_step_end_to_end(oracle, pkg, dev, name):
    """VERDICT round 4 item 8 (stretch, opt-in): the whole train step in split-fp32 arithmetic (model.split_train).  Loss to 1e-5 of the
    reference's; every gradient tensor against the oracle inside the band the exact path is held to -- twice the oracle's OWN shift under a
    seeded relative weight perturbation (floor 1e-3) -- with the perturbation at the mode's accuracy class (1e-5 instead of 1e-6: its products
    carry 16 significant bits); and close to the exact device path's own gradients."""
    from test_gpu_backward import _oracle_grads, _sensitivity_band

    from conftest import l2_rel

    g = load_golden(name)
    inputs = golden_inputs(g)
    row, col, pb, K, Ct = inputs
    Nc, Nf = int(g["Nc"]), int(g["Nf"])
    w, m = _model(pkg, oracle, g, dev, row.shape[0])
    Cc, Cf, loss0 = m.train_step(row, col, pb, K, Ct)
    exact = [p.grad.clone() for p in m.network.parameters()]
    m.split_train = True
    Cc, Cf, loss = m.train_step(row, col, pb, K, Ct)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert max_rel(Cc, g["C_coarse"]) < TOL and max_rel(Cf, g["C_fine"]) < TOL
    _, g0 = _oracle_grads(oracle, w, inputs, Nc, Nf)
    # three seeds where the oracle is cheap: the reference's gradient is piecewise -- on small_16_32 ONE discrete decision moves
    # layer 0's weight gradient by 74 % and flips under one seeded 1e-6 jitter out of three; two evaluations of the same weights may land on either side
    band = _sensitivity_band(oracle, w, inputs, Nc, Nf, g0, seeds=(1,) if name.startswith("cfg2") else (1, 2, 3), rel=1e-5)
    worst = (0.0, 1.0, "")
    rows = []
    for (k, q), ex in zip(m.named_parameters(), exact):
        key = k if k.startswith("network.") else "network." + k
        assert torch.isfinite(q.grad).all(), k
        e = l2_rel(q.grad, g0[key])
        bar = max(2.0 * band[key], 1e-3)
        rows.append((k, e, bar, l2_rel(q.grad, ex)))
        if e / bar > worst[0] / worst[1]:
            worst = (e, bar, k)
    for k, e, bar, ex in rows:
        print(f"  {k:34s} vs oracle {e:.2e} (bar {bar:.2e})   vs the exact device path {ex:.2e}")
    for k, e, bar, ex in rows:
        assert e < bar, (k, e, bar)
    print(f"{name}: split-fp32 train step: closest to its bar: {worst[2]} L2-rel {worst[0]:.2e} (bar {worst[1]:.2e})")
    # the autograd surface takes the same path: identical loss and gradients
    grads = [p.grad.clone() for p in m.network.parameters()]
    for p in m.network.parameters():
        p.grad = None
    Cc2, Cf2 = m(row, col, pb, K)
    l2 = m.ray_loss(Cc2, Cf2, Ct.to(dev))
    l2.backward()
    assert float(l2.detach()) == float(loss)
    assert all(torch.equal(a, p.grad) for a, p in zip(grads, m.network.parameters()))


@pytest.mark.parametrize("B,Nc,Nf", [(50, 24, 40), (37, 16, 32), (130, 31, 65)])
def test_split_train_step_on_ragged_sizes(oracle, pkg, dev, B, Nc, Nf):
    """Row counts that fill neither the 128-sample workgroups nor the 256-sample groups the fragment-layout buffers are counted in (50 x 24 = 1,200
    coarse samples: 9.4 workgroups, 37.5 wave blocks in a buffer of 40): every wave block the weight-gradient products read must have been written
    (lanes past the end: copies of the last sample forward, exact zeros in the chain).  On the coarse-only loss (no sorts, no position path: the
    well-conditioned case of tests/test_gpu_backward.py) every gradient tensor is held to 1e-3 against autograd; on the full loss the step equals the
    exact device step within the reference's own discontinuity (loss to 1e-5, every tensor finite)."""
    from conftest import l2_rel

    inputs = oracle.fern_inputs(B, seed=9)
    row, col, pb, K, Ct = inputs
    w = oracle.make_weights(6, sharp=True)
    p = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    Cc, Cf = oracle.render(p, row, col, pb, K, Nc, Nf)
    oloss = torch.sum(torch.square(Cc - Ct))
    oloss.backward()
    m = pkg.NeRFModel(Nc, Nf, B)
    m.load_state_dict(w)
    m = m.to(dev)
    m.split_train = True
    for rep in range(2):  # the second pass runs on a workspace whose padding blocks hold the first pass's data
        for q in m.network.parameters():
            q.grad = None
        Dc, Df = m(row, col, pb, K)
        loss = torch.sum(torch.square(Dc - Ct.to(dev)))
        loss.backward()
        assert abs(float(loss) - float(oloss)) <= 1e-5 * float(oloss)
        errs = {k: l2_rel(q.grad, p["network." + k if not k.startswith("network.") else k].grad) for k, q in m.named_parameters()}
        worst = max(errs, key=errs.get)
        assert errs[worst] < 1e-3, (rep, worst, errs[worst])
    print(f"{B} x ({Nc} + {Nf}): split-fp32 train step, coarse-only loss: worst gradient L2-rel {errs[worst]:.2e} ({worst})")
    Dc, Df, loss = m.train_step(row, col, pb, K, Ct)
    _, _, ol, _ = oracle.loss_and_grads(w, row, col, pb, K, Ct, Nc, Nf)
    assert abs(float(loss) - float(ol)) <= 1e-5 * float(ol) and all(torch.isfinite(q.grad).all() for q in m.network.parameters())


def test_split_train_step_is_bit_reproducible(oracle, pkg, dev):
    """Like the exact step (tests/test_gpu_backward.py::test_train_step_is_bit_reproducible): every sum of the split-fp32 train step runs in a fixed
    order (slab reduce in workgroup order, no float atomics) -- the same weights and batch give the same loss and 24 gradients bit for bit."""
    g = load_golden("cfg1_lego_crop32")
    row, col, pb, K, Ct = golden_inputs(g)
    w, m = _model(pkg, oracle, g, dev, row.shape[0])
    m.split_train = True
    ref = None
    for rep in range(3):
        Cc, Cf, loss = m.train_step(row, col, pb, K, Ct)
        cur = (float(loss), [p.grad.detach().clone() for p in m.network.parameters()])
        if ref is None:
            ref = cur
            continue
        assert cur[0] == ref[0]
        for (k, _), a, b in zip(m.network.named_parameters(), cur[1], ref[1]):
            assert torch.equal(a, b), (rep, k, float((a - b).abs().max()))
