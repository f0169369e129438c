"""GPU parity of the train step (forward + ray_loss + backward through the C ABI).

What can be asserted, and why (measurements in DESIGN.md section 6, all reproducible on the CPU oracle):
  * The reference's gradient is DISCONTINUOUS in the fp32 MLP outputs in two places: the five independent
    channel sorts (quirk Q1: two samples whose sigma / r / g / b differ by less than GEMM rounding noise swap
    sorted positions and with them the gradients they receive) and the ReLU kinks on the un-detached t_fine
    path (quirk Q9 x a 3217 rad/unit encoding).  A 1e-6 relative perturbation of the reference's own head
    outputs moves its trunk gradients by 2-14 % L2-rel.
  * On top of that the position gradient d loss/d t_fine is a cancelling sum of terms weighted by f_l <= 3217:
    the reference's own fp32 value is 0.7-0.9 % (L2-rel) away from an fp64 evaluation of the same graph.
So: (1) every well-conditioned piece is held to 1e-4 (merge backward, resampling backward, the whole dX chain and
weight-gradient GEMMs on a coarse-only loss, the colour / direction / feature branches of the full loss);
(2) with the reference's discrete decisions replayed (sort order, ReLU masks) the full gradient must sit inside the
reference's own fp32 noise band; (3) the untouched product path must give the loss to 1e-5 and every gradient tensor
within twice the reference's own shift under a 1e-6 relative weight perturbation, computed per case and per tensor in the test;
(4) with the decisions replayed every tensor is compared with a float64 evaluation of the same branch: the device may not be
further from it than 2.5x the reference's own fp32 gradient is (measured 0.8-1.8x).

Which bar is pinned to what:
  * test_full_gradient_given_the_references_decisions -- pinned to the REFERENCE itself, no oracle in the loop: the fixtures
    tests/golden/dec_*.npz (make_decisions_golden.py) hold, from ONE run of /root/reference/nerf.py on a 32-ray batch, the `torch.sort` indices of
    nerf.py:308, `index_fine` of nerf.py:248, the bit-packed ReLU sign bits of nerf.py:107-119 and all 24 gradients of nerf.py:473.  The test replays the
    stored sort indices and ReLU masks into the workspace and compares the device gradients with the stored ones at NOISE_BAND.
  * (1), test_full_gradient_given_reference_decisions, (4) -- against the oracle's autograd run in this process (the oracle is pinned to the
    reference bit for bit on the generating host: tests/test_oracle_golden.py), decisions taken from that in-process run.
  * (3) -- against the in-process oracle AND the reference's stored gradients (tests/golden/cfg*.npz: `grad_*` / `gslice_*`), both inside the
    sensitivity band, because without replay the discrete decisions of two fp32 evaluations differ.
"""
import numpy as np
import pytest
import torch

from conftest import golden_inputs, l2_rel, load_golden

pytestmark = pytest.mark.gpu
GTOL = 1e-4          # well-conditioned gradients
NOISE_BAND = 2e-2    # full gradient with the reference's discrete decisions replayed (reference fp32-vs-fp64: ~1e-2)


def _relu_mask_image(hidden, tiles):
    """ReLU masks in the field kernels' accumulator layout: [8][tiles][4][256] int16, bit r of entry
    (f*2+st, tid = wv*64 + h*32 + j) <-> feature wv*64 + f*32 + 8(r>>2) + 4h + (r&3) of sample st*32 + j."""
    f, st, wv, h, j, r = torch.meshgrid(torch.arange(2), torch.arange(2), torch.arange(4), torch.arange(2), torch.arange(32),
                                        torch.arange(16), indexing="ij")
    feat = wv * 64 + f * 32 + 8 * (r >> 2) + 4 * h + (r & 3)
    samp = st * 32 + j
    out = []
    for H in hidden:  # [M, 256]
        M = H.shape[0]
        Hp = torch.zeros(tiles * 64, 256, dtype=torch.bool)
        Hp[:M] = H > 0
        Hp = Hp.view(tiles, 64, 256)
        bits = Hp[:, samp, feat].to(torch.int32)  # [tiles, f, st, wv, h, j, r]
        word = (bits << torch.arange(16, dtype=torch.int32)).sum(-1)
        word = word - 65536 * (word >= 32768).to(torch.int32)
        out.append(word.reshape(tiles, 4, 256).to(torch.int16))
    return torch.stack(out)  # [8, tiles, 4, 256]


def _views(pkg, m, B, Nc, Nf):
    from nerf_tiny_amd import _abi

    ws = m.last_workspace
    return lambda name, shape, dt=None: _abi.ws_view(ws, B, Nc, Nf, _abi.SAVE_FOR_BACKWARD, name, shape, dt)


def _train_step(pkg, oracle, dev, w, inputs, Nc, Nf, ref_stages=None, coarse_only=False):
    """forward (saving) -> [optionally replay the reference's discrete decisions] -> loss -> backward."""
    row, col, pb, K, Ct = inputs
    B = row.shape[0]
    m = pkg.NeRFModel(Nc, Nf, B)
    m.load_state_dict(w)
    m = m.to(dev)
    Cc, Cf = m(row, col, pb, K)
    if ref_stages is not None:
        view = _views(pkg, m, B, Nc, Nf)
        N = Nc + Nf
        vals = torch.cat((torch.cat((view("t_c", (B, Nc)), view("t_f", (B, Nf))), 1).unsqueeze(2),
                          torch.cat((view("rgb_c", (B, Nc, 3)), view("rgb_f", (B, Nf, 3))), 1),
                          torch.cat((view("sig_c", (B, Nc)), view("sig_f", (B, Nf))), 1).unsqueeze(2)), dim=2)  # [B,N,5]
        perm = ref_stages["perm"].to(dev)  # [B,N,5] sorted position -> original index
        view("bundle", (B, N, 5)).copy_(torch.gather(vals, 1, perm))  # device values in the reference's order
        view("perm", (B, 5, N), torch.int16).copy_(perm.permute(0, 2, 1).to(torch.int16))
        f_p, _ = oracle.frequencies()
        gd = ref_stages["gd"]
        tiles_c, tiles_f = (B * Nc + 63) // 64, (B * Nf + 63) // 64
        imgs = []
        for pts, n, tl in ((ref_stages["pts_c"], Nc, tiles_c), (ref_stages["pts_f"], Nf, tiles_f)):
            with torch.no_grad():
                _, _, hidden, _, _ = oracle.mlp(w, oracle.encode(pts, f_p), gd[:, None, :].expand(-1, n, -1), return_hidden=True)
            imgs.append(_relu_mask_image([h.reshape(-1, 256) for h in hidden], tl))
        img = torch.cat(imgs, dim=1)
        view("masks", tuple(img.shape), torch.int16).copy_(img.to(dev))
    if coarse_only:
        loss = torch.sum(torch.square(Cc - Ct.to(dev)))  # d loss / d C_fine = 0
    else:
        loss = m.ray_loss(Cc, Cf, Ct.to(dev))
    loss.backward()
    return m, loss


def _oracle_with_grads(oracle, w, inputs, Nc, Nf, coarse_only=False):
    row, col, pb, K, Ct = inputs
    p = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    st = {}
    Cc, Cf = oracle.render(p, row, col, pb, K, Nc, Nf, stages=st)
    for k in ("sig_c", "rgb_c", "sig_f", "rgb_f", "t_f", "pts_f"):
        st[k].retain_grad()
    loss = torch.sum(torch.square(Cc - Ct)) if coarse_only else oracle.ray_loss(Cc, Cf, Ct)
    loss.backward()
    return p, st, loss.detach()


CASES = ["cfg1_lego_crop32", "cfg1_lego_crop32_sharp", "cfg4_fern_rand512", "small_16_32"]


def _case(oracle, name, max_rays=None):
    g = load_golden(name)
    inputs = golden_inputs(g)
    if max_rays is not None:
        inputs = tuple(x[:max_rays] if (torch.is_tensor(x) and x.dim() > 0 and x.shape[0] == inputs[0].shape[0]) else x for x in inputs)
    return g, inputs, int(g["Nc"]), int(g["Nf"]), oracle.make_weights(int(g["seed"]), bool(g["sharp"]))


# ---------------------------------------------------------------------------------------------------------------
# (1) well-conditioned pieces at 1e-4
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", CASES)
def test_chain_and_weight_gradients_on_coarse_only_loss(oracle, pkg, dev, name):
    """loss = sum (C_coarse - C*)^2: no sort and no position path, so the fused dX chain, the split-M weight-gradient
    GEMMs, the bias column sums and the thin heads must reproduce autograd to 1e-4 per tensor."""
    g, inputs, Nc, Nf, w = _case(oracle, name, max_rays=256)
    p, st, oloss = _oracle_with_grads(oracle, w, inputs, Nc, Nf, coarse_only=True)
    m, loss = _train_step(pkg, oracle, dev, w, inputs, Nc, Nf, coarse_only=True)
    assert abs(float(loss) - float(oloss)) <= 1e-5 * float(oloss)
    worst = 0.0
    for (k, ref), q in zip(p.items(), m.network.parameters()):
        e = l2_rel(q.grad, ref.grad)
        worst = max(worst, e)
        assert e < GTOL, (k, e)
    print(f"{name}: coarse-only loss, worst grad L2-rel {worst:.2e}")


@pytest.mark.parametrize("rays", [701, 333, 99, 7])
def test_weight_gradients_on_ragged_row_counts(oracle, pkg, dev, rays):
    """Row counts that do not fill the weight-gradient kernels' row ranges: 701 rays leave a ragged last range in all three kernels
    (128 x 128 and 128 x 64 blocks and the thin colour-head product), 333 rays = 63,936 rows in every MFMA-bound
    product shape (the per-row tail loop with its clamped rows, incl. the column-sum and sigma-head duties), 99 rays in the
    256 x 256 ones only, 7 rays leave most of the 256 workgroups without rows.  Bar 3e-4 per tensor: layer 0's weight gradient
    (the ill-conditioned one, see _sensitivity_band) sits at 1.5e-4 for 333 rays in fp32 on either side, while ONE dropped or doubled row
    of 63,936 would move a gradient by ~1 / sqrt(rows) = 4e-3."""
    g, inputs, Nc, Nf, w = _case(oracle, "cfg1_lego_crop32", max_rays=rays)
    assert inputs[0].shape[0] == rays
    p, st, oloss = _oracle_with_grads(oracle, w, inputs, Nc, Nf, coarse_only=True)
    m, loss = _train_step(pkg, oracle, dev, w, inputs, Nc, Nf, coarse_only=True)
    assert abs(float(loss) - float(oloss)) <= 1e-5 * float(oloss)
    errs = {k: l2_rel(q.grad, ref.grad) for (k, ref), q in zip(p.items(), m.network.parameters())}
    print(f"{rays} rays: worst grad L2-rel {max(errs.values()):.2e} ({max(errs, key=errs.get)})")
    for k, e in errs.items():
        assert e < 3e-4, (k, e)


def test_train_step_is_bit_reproducible(oracle, pkg, dev):
    """Every sum of the fp32 train step runs in a fixed order (slab reduce, column sums, the gamma_d columns' two-step sum; no float
    atomics): the same weights and batch give the same 24 gradients bit for bit, launch after launch."""
    g, inputs, Nc, Nf, w = _case(oracle, "cfg1_lego_crop32")
    ref = None
    for rep in range(4):
        m, loss = _train_step(pkg, oracle, dev, w, inputs, Nc, Nf)
        grads = [q.grad.detach().clone() for q in m.network.parameters()]
        if ref is None:
            ref, lref = grads, float(loss.detach())
            continue
        assert float(loss.detach()) == lref
        for (k, _), a, b in zip(m.network.named_parameters(), grads, ref):
            assert torch.equal(a, b), (rep, k, float((a - b).abs().max()))


@pytest.mark.parametrize("name", CASES)
def test_merge_backward_given_reference_sort_order(oracle, pkg, dev, name):
    """rows a8/a9 backward: d sigma, d rgb of both passes and the merge part of d t_fine (workspace buffers after
    backward) against autograd, with the reference's permutation replayed."""
    g, inputs, Nc, Nf, w = _case(oracle, name, max_rays=256)
    B = inputs[0].shape[0]
    p, st, _ = _oracle_with_grads(oracle, w, inputs, Nc, Nf)
    std = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in st.items()}
    m, _ = _train_step(pkg, oracle, dev, w, inputs, Nc, Nf, ref_stages=std)
    view = _views(pkg, m, B, Nc, Nf)
    assert l2_rel(view("dsig_f", (B, Nf)), st["sig_f"].grad) < 3e-4
    assert l2_rel(view("drgb_f", (B, Nf, 3)), st["rgb_f"].grad) < 3e-4
    # colour / feature / direction branches of the FULL loss do not see the ill-conditioned paths
    for k, q in m.named_parameters():
        if any(s in k for s in ("color_layer", "dir_info", "point_info")):
            assert l2_rel(q.grad, p[k].grad) < 3e-4, k


@pytest.mark.parametrize("name", CASES)
def test_resample_backward_stage(oracle, pkg, dev, name):
    """row a7 backward on its own: feed the oracle's d loss/d t_fine and d C_coarse to nerf_hip_coarse_composite_backward."""
    g, inputs, Nc, Nf, w = _case(oracle, name, max_rays=256)
    row, col, pb, K, Ct = inputs
    st = {}
    with torch.no_grad():
        oracle.render(w, row, col, pb, K, Nc, Nf, stages=st)
    sig = st["sig_c"].clone().requires_grad_(True)
    rgb = st["rgb_c"].clone().requires_grad_(True)
    delta_c = ((st["far"] - st["near"]) / Nc)[:, None].expand(-1, Nc)
    w_c = oracle.weights_from_sigma(delta_c, sig)
    t_f, _ = oracle.resample(st["t_c"], w_c, Nf)
    C_c = oracle.composite(w_c, rgb)
    gen = torch.Generator().manual_seed(3)
    g_t = torch.randn(t_f.shape, generator=gen) * 0.01
    g_C = torch.randn(C_c.shape, generator=gen)
    (t_f * g_t).sum().add((C_c * g_C).sum()).backward()
    delta0 = float(st["t_c"][0, 1] - st["t_c"][0, 0])
    d = lambda x: x.contiguous().to(dev)
    dsig, drgb = pkg.ops.coarse_composite_backward(d(st["t_c"]), d(st["sig_c"]), d(st["rgb_c"]), st["near"], st["far"], delta0,
                                                   d(g_C), d(g_t))
    assert l2_rel(drgb, rgb.grad) < 1e-5
    assert l2_rel(dsig, sig.grad) < GTOL


# ---------------------------------------------------------------------------------------------------------------
# (2) full gradient with the reference's decisions replayed: inside the reference's own fp32 noise band
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", CASES + ["cfg2_lego_rand4096"])  # incl. once at cfg2's own size (4096 rays)
def test_full_gradient_given_reference_decisions(oracle, pkg, dev, name):
    """Reference = the oracle's autograd run in THIS process (the golden gradients were produced on another host whose
    BLAS rounds differently, i.e. with other discrete decisions; they are compared in test_train_step_end_to_end)."""
    g, inputs, Nc, Nf, w = _case(oracle, name)
    p, st, oloss = _oracle_with_grads(oracle, w, inputs, Nc, Nf)
    std = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in st.items()}
    m, loss = _train_step(pkg, oracle, dev, w, inputs, Nc, Nf, ref_stages=std)
    assert abs(float(loss) - float(oloss)) <= 1e-5 * abs(float(oloss))
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    worst = 0.0
    for (k, ref), q in zip(p.items(), m.network.parameters()):
        e = l2_rel(q.grad, ref.grad)
        worst = max(worst, e)
        assert e < NOISE_BAND, (k, e)
    print(f"{name}: worst grad L2-rel vs autograd (reference decisions replayed) {worst:.2e}")


def _unpack_bits(a, n):
    return torch.from_numpy(np.unpackbits(a, axis=-1, bitorder="little")[..., :n].astype(np.uint8))


@pytest.mark.parametrize("name", ["dec_cfg1_lego_crop32_r32", "dec_cfg4_fern_r32"])
def test_full_gradient_given_the_references_decisions(pkg, dev, name):
    """Row a11 pinned to the reference's OWN gradients.  The fixture holds what ONE run of /root/reference/nerf.py took
    and produced on these 32 rays: sort indices (nerf.py:308), ReLU sign bits of the eight trunk layers (nerf.py:107-111), index_fine
    (nerf.py:248), outputs, loss and the 24 gradients (nerf.py:473).  Forward with the library, write the reference's sort order (device values
    gathered in that order) and ReLU masks into the workspace, run the library's backward, compare with the STORED gradients.  No oracle
    function runs here: the fixture is data, the replay is index arithmetic.  (dir_info's ReLU and the sign of sigma are taken by the device
    from its own saved values; index_fine is checked against the bins the device's own coarse weights give.)"""
    from nerf_tiny_amd import _abi  # noqa: F401

    g = load_golden(name)
    row, col, pb, K, Ct = golden_inputs(g)
    B, Nc, Nf = row.shape[0], int(g["Nc"]), int(g["Nf"])
    N = Nc + Nf
    # the fixture stores the weight generator's seed; bench.synth_weights is the product-side restatement of that generator
    import importlib.util
    import os

    from conftest import ROOT

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sd = bench.synth_weights(int(g["seed"]), bool(g["sharp"])).state_dict()
    m = pkg.NeRFModel(Nc, Nf, B)
    m.load_state_dict(sd)
    m = m.to(dev)
    Cc, Cf = m(row, col, pb, K)
    assert float((Cc.detach().cpu() - torch.from_numpy(g["C_coarse"])).abs().max()) <= 1e-4 * float(np.abs(g["C_coarse"]).max())
    assert float((Cf.detach().cpu() - torch.from_numpy(g["C_fine"])).abs().max()) <= 1e-4 * float(np.abs(g["C_fine"]).max())
    view = _views(pkg, m, B, Nc, Nf)
    # resampling bins: the device's own coarse weights through nerf.py:228-248's arithmetic (torch ops on device values, not the oracle)
    w_c = view("w_c", (B, Nc)).cpu()
    cdf = torch.cumsum(w_c, dim=1)
    lo, hi = cdf[:, 0], cdf[:, -1]
    u = torch.arange(1, Nf + 1, dtype=torch.float32)[None, :] * ((hi - lo) / np.float32(Nf + 1))[:, None] + lo[:, None]
    k_dev = torch.searchsorted(cdf.contiguous(), u.contiguous()) - 1
    agree = float((k_dev.numpy() == g["index_fine"].astype(np.int64)).mean())
    assert agree > 0.995, agree
    # replay: the reference's five per-channel sort permutations and the eight trunk layers' ReLU masks
    perm = torch.from_numpy(g["sort_index"].astype(np.int64)).to(dev)  # [B, N, 5]
    vals = torch.cat((torch.cat((view("t_c", (B, Nc)), view("t_f", (B, Nf))), 1).unsqueeze(2),
                      torch.cat((view("rgb_c", (B, Nc, 3)), view("rgb_f", (B, Nf, 3))), 1),
                      torch.cat((view("sig_c", (B, Nc)), view("sig_f", (B, Nf))), 1).unsqueeze(2)), dim=2)
    view("bundle", (B, N, 5)).copy_(torch.gather(vals, 1, perm))
    view("perm", (B, 5, N), torch.int16).copy_(perm.permute(0, 2, 1).to(torch.int16))
    imgs = []
    for key, n in (("relu_c", Nc), ("relu_f", Nf)):
        bits = _unpack_bits(g[key], 256)  # [8, B, n, 256]
        imgs.append(_relu_mask_image([bits[i].reshape(-1, 256) for i in range(8)], (B * n + 63) // 64))
    img = torch.cat(imgs, dim=1)
    view("masks", tuple(img.shape), torch.int16).copy_(img.to(dev))
    loss = m.ray_loss(Cc, Cf, Ct.to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    errs = {}
    for k, q in m.named_parameters():
        key = k if k.startswith("network.") else "network." + k
        errs[key] = l2_rel(q.grad, torch.from_numpy(g["grad_" + key]))
    worst = max(errs, key=errs.get)
    print(f"{name}: device gradients vs the REFERENCE's stored gradients with its decisions replayed: worst L2-rel {errs[worst]:.2e} ({worst}); "
          f"resampling bins agree on {agree * 100:.2f} %")
    for k, e in errs.items():
        assert e < NOISE_BAND, (k, e)


def test_position_gradient_is_as_accurate_as_the_reference(oracle, pkg, dev):
    """d loss/d t_fine through the field (the ill-conditioned part): compare BOTH the device value and the reference's
    fp32 value with an fp64 evaluation of the same graph; the device must not be further from fp64 than 1.5x the reference."""
    g, inputs, Nc, Nf, w = _case(oracle, "cfg1_lego_crop32_sharp", max_rays=128)
    B = inputs[0].shape[0]
    p, st, _ = _oracle_with_grads(oracle, w, inputs, Nc, Nf)
    std = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in st.items()}
    m, _ = _train_step(pkg, oracle, dev, w, inputs, Nc, Nf, ref_stages=std)
    view = _views(pkg, m, B, Nc, Nf)
    dw = std["d_wrd"]
    ref32_field = (st["pts_f"].grad * dw[:, None, :]).sum(-1)
    merge_part = st["t_f"].grad - ref32_field
    dev_field = view("dt_f", (B, Nf)).cpu() - merge_part
    # fp64 evaluation with the same upstream gradients
    w64 = {k: v.double() for k, v in w.items()}
    pts = std["pts_f"].double().requires_grad_(True)
    f_p, _ = oracle.frequencies()
    rgb, sig = oracle.mlp(w64, oracle.encode(pts, f_p.double()), std["gd"].double()[:, None, :].expand(-1, Nf, -1))
    (rgb * st["rgb_f"].grad.double()).sum().add((sig * st["sig_f"].grad.double()).sum()).backward()
    truth = (pts.grad * dw[:, None, :].double()).sum(-1)
    e_ref = float((ref32_field.double() - truth).norm() / truth.norm())
    e_dev = float((dev_field.double() - truth).norm() / truth.norm())
    print(f"d t_fine (field part) vs fp64: reference fp32 {e_ref:.2e}, device {e_dev:.2e}")
    assert e_dev < max(1.5 * e_ref, 2e-3)


def _fp64_grads_with_decisions(oracle, w, inputs, Nc, Nf, st):
    """The branch of the reference's piecewise-smooth graph that its fp32 run took, evaluated in float64: every discrete
    decision -- ReLU masks and the sign inside |.|, the resampling bin k, the five per-channel sort permutations -- comes from
    the fp32 stages `st`.  Returns {name: gradient (float64)} of the reference's loss."""
    import torch.nn.functional as F

    row, col, pb, K, Ct = inputs
    p = {k: v.double().clone().requires_grad_(True) for k, v in w.items()}
    f_p32, _ = oracle.frequencies()
    f_p = f_p32.double()
    R, o, near, far = [x.double() for x in oracle.poses_extract(pb)]
    d_cam = st["d_cam"].double()
    gd32, gd = st["gd"], st["gd"].double()

    def field(pts, pts32, n):
        with torch.no_grad():
            _, _, hid32, _, c32 = oracle.mlp(w, oracle.encode(pts32, f_p32), gd32[:, None, :].expand(-1, n, -1), return_hidden=True)
            pre32 = F.linear(hid32[7], w["network.sigma_layer.0.weight"], w["network.sigma_layer.0.bias"]).squeeze(-1)
        gp = oracle.encode(pts, f_p)
        h = gp
        for i in range(8):
            inp = torch.cat((h, gp), dim=-1) if i == 4 else h
            h = F.linear(inp, p[f"network.point_layer.{i}.0.weight"], p[f"network.point_layer.{i}.0.bias"]) * (hid32[i] > 0)
        sigma = F.linear(h, p["network.sigma_layer.0.weight"], p["network.sigma_layer.0.bias"]).squeeze(-1) * torch.sign(pre32)
        feat = F.linear(h, p["network.point_info.weight"], p["network.point_info.bias"])
        c = F.linear(torch.cat((gd[:, None, :].expand(-1, n, -1), feat), dim=-1), p["network.dir_info.0.weight"],
                     p["network.dir_info.0.bias"]) * (c32 > 0)
        return torch.sigmoid(F.linear(c, p["network.color_layer.0.weight"], p["network.color_layer.0.bias"])), sigma

    t_c = st["t_c"].double()
    rgb_c, sig_c = field(oracle.sample_points(R, o, d_cam, t_c), st["pts_c"], Nc)
    w_c = oracle.weights_from_sigma(((far - near) / Nc)[:, None].expand(-1, Nc), sig_c)
    cdf = torch.cumsum(w_c, dim=1)
    hi, lo = cdf[:, -1].detach(), cdf[:, 0].detach()  # max / min of a non-decreasing sequence
    slope = torch.cat(((t_c[0, 1] - t_c[0, 0]) / (w_c[:, 1:] + 1e-7), torch.zeros(t_c.shape[0], 1, dtype=torch.float64)), dim=1)
    u = torch.arange(1, Nf + 1, dtype=torch.float64)[None, :] * ((hi - lo) / (Nf + 1))[:, None] + lo[:, None]
    k = st["k"]
    t_f = torch.gather(t_c, 1, k) + (u - torch.gather(cdf, 1, k)) * torch.gather(slope, 1, k)
    rgb_f, sig_f = field(oracle.sample_points(R, o, d_cam, t_f), st["pts_f"], Nf)
    bundle = torch.cat((torch.cat((t_c, t_f), 1).unsqueeze(2), torch.cat((rgb_c, rgb_f), 1), torch.cat((sig_c, sig_f), 1).unsqueeze(2)), dim=2)
    sb = torch.gather(bundle, 1, st["perm"])
    t_s, rgb_s, sig_s = sb[:, :, 0], sb[:, :, 1:4], sb[:, :, 4]
    delta = torch.cat((t_s[:, 1:] - t_s[:, :-1], torch.full((t_s.shape[0], 1), 1e-4, dtype=torch.float64)), dim=1)
    C_c, C_f = oracle.composite(w_c, rgb_c), oracle.composite(oracle.weights_from_sigma(delta, sig_s), rgb_s)
    oracle.ray_loss(C_c, C_f, Ct.double()).backward()
    return {k_: v.grad for k_, v in p.items()}


@pytest.mark.parametrize("name", ["cfg1_lego_crop32", "cfg1_lego_crop32_sharp", "cfg4_fern_rand512"])
def test_every_gradient_is_as_accurate_as_the_reference(oracle, pkg, dev, name):
    """All 24 tensors, full loss, the reference's decisions replayed: the device gradient must not be further from the
    float64 evaluation of the same branch than 2.5x what the reference's own fp32 gradient is (floor 1e-4; measured: the
    device sits at 0.8-1.8x -- its dot products are strict k-ordered fma chains, the host BLAS sums in blocks).  This is the
    statement a flat tolerance cannot make: where the fp32 reference itself is 1 % away from fp64, so may the device be;
    where it is 1e-6 away, the device is held to 1e-4."""
    g, inputs, Nc, Nf, w = _case(oracle, name, max_rays=256)
    p, st, _ = _oracle_with_grads(oracle, w, inputs, Nc, Nf)
    std = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in st.items()}
    truth = _fp64_grads_with_decisions(oracle, w, inputs, Nc, Nf, std)
    m, _ = _train_step(pkg, oracle, dev, w, inputs, Nc, Nf, ref_stages=std)
    rows = []
    for (k, ref), q in zip(p.items(), m.network.parameters()):
        e_ref, e_dev = l2_rel(ref.grad, truth[k]), l2_rel(q.grad, truth[k])
        rows.append((k, e_ref, e_dev))
    for k, e_ref, e_dev in rows:
        print(f"  {k:40s} reference fp32 vs fp64 {e_ref:.2e}   device vs fp64 {e_dev:.2e}")
    for k, e_ref, e_dev in rows:
        assert e_dev < max(2.5 * e_ref, 1e-4), (k, e_ref, e_dev)
    worst = max(rows, key=lambda r: r[2] / max(r[1], 1e-30))
    print(f"{name}: vs fp64 -- reference fp32 {min(r[1] for r in rows):.1e}..{max(r[1] for r in rows):.1e}, device "
          f"{min(r[2] for r in rows):.1e}..{max(r[2] for r in rows):.1e}; largest device/reference ratio {worst[2] / max(worst[1], 1e-30):.2f} ({worst[0]})")


# ---------------------------------------------------------------------------------------------------------------
# (3) the untouched product path
# ---------------------------------------------------------------------------------------------------------------
def _oracle_grads(oracle, w, inputs, Nc, Nf):
    row, col, pb, K, Ct = inputs
    _, _, loss, g = oracle.loss_and_grads(w, row, col, pb, K, Ct, Nc, Nf)
    return loss, g


def _sensitivity_band(oracle, w, inputs, Nc, Nf, g0, seeds, rel=1e-6):
    """The reference's OWN gradient shift when its weights move by a seeded relative 1e-6 -- the size of the difference
    between two correct fp32 evaluations of the MLP (GEMM summation order: 1.3e-6 on C_coarse).
    Per tensor, L2-rel, maximum over the seeds."""
    band = {k: 0.0 for k in g0}
    for sd in seeds:
        gen = torch.Generator().manual_seed(sd)
        wp = {k: v * (1.0 + rel * torch.randn(v.shape, generator=gen)) for k, v in w.items()}
        _, g1 = _oracle_grads(oracle, wp, inputs, Nc, Nf)
        for k in g0:
            band[k] = max(band[k], l2_rel(g1[k], g0[k]))
    return band


@pytest.mark.parametrize("name", CASES + ["cfg2_lego_rand4096"])
def test_train_step_end_to_end(oracle, pkg, dev, name):
    """The untouched product path against the oracle's autograd run in THIS process.  The reference's gradient is discontinuous
    in its own MLP outputs (module docstring), so the bar is not a flat number: per case and per tensor it is twice the
    reference's own gradient shift under a seeded 1e-6 relative weight perturbation (floor 1e-3), and the loss itself 1e-5.
    The golden gradients (another host's BLAS = other discrete decisions) are held to the same band."""
    g = load_golden(name)
    inputs = golden_inputs(g)
    Nc, Nf = int(g["Nc"]), int(g["Nf"])
    w = oracle.make_weights(int(g["seed"]), bool(g["sharp"]))
    m, loss = _train_step(pkg, oracle, dev, w, inputs, Nc, Nf)
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    _, g0 = _oracle_grads(oracle, w, inputs, Nc, Nf)
    band = _sensitivity_band(oracle, w, inputs, Nc, Nf, g0, seeds=(1,) if name.startswith("cfg2") else (1, 2))
    worst = (0.0, 0.0, "")
    for k, q in m.named_parameters():
        key = "network." + k if not k.startswith("network.") else k
        got = q.grad.detach().cpu().double()
        assert torch.isfinite(got).all(), k
        e = l2_rel(got, g0[key])
        bar = max(2.0 * band[key], 1e-3)
        if e / bar > worst[0] / max(worst[1], 1e-30):
            worst = (e, bar, k)
        assert e < bar, (k, e, band[key])
        # the golden file's gradients of this tensor (full, or every 97th element), produced on another host
        gn = float(g["gnorm_" + key])
        assert abs(float(got.norm()) - gn) <= max(2.0 * band[key], 1e-3) * gn + 1e-12, (k, float(got.norm()), gn)
        if "grad_" + key in g:
            e2 = l2_rel(got, torch.from_numpy(g["grad_" + key]))
        else:
            ref = torch.from_numpy(g["gslice_" + key]).double()
            e2 = float((got.flatten()[::97] - ref).norm() / max(float(ref.norm()), 1e-30))
        assert e2 < max(3.0 * band[key], 2e-3), (k, e2, band[key])
    print(f"{name}: closest to its bar: {worst[2]} L2-rel {worst[0]:.2e} (bar {worst[1]:.2e} = 2 x the reference's own 1e-6 sensitivity)")


def test_odd_sizes_coarse_only(oracle, pkg, dev):
    """ragged sizes: B*N not a multiple of the 64-sample tile, Nc/Nf not multiples of 64."""
    B, Nc, Nf = 50, 24, 40
    inputs = oracle.fern_inputs(B, seed=9)
    w = oracle.make_weights(6, sharp=True)
    p, st, oloss = _oracle_with_grads(oracle, w, inputs, Nc, Nf, coarse_only=True)
    m, loss = _train_step(pkg, oracle, dev, w, inputs, Nc, Nf, coarse_only=True)
    assert abs(float(loss) - float(oloss)) <= 1e-5 * float(oloss)
    for (k, ref), q in zip(p.items(), m.network.parameters()):
        assert l2_rel(q.grad, ref.grad) < GTOL, k
    # and the full loss inside the noise band with the reference's decisions
    p, st, oloss = _oracle_with_grads(oracle, w, inputs, Nc, Nf)
    std = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in st.items()}
    m, loss = _train_step(pkg, oracle, dev, w, inputs, Nc, Nf, ref_stages=std)
    assert abs(float(loss) - float(oloss)) <= 1e-5 * float(oloss)
    for (k, ref), q in zip(p.items(), m.network.parameters()):
        assert l2_rel(q.grad, ref.grad) < NOISE_BAND, k


def test_ray_loss_and_grad(oracle, pkg, dev):
    torch.manual_seed(0)
    Cc = torch.rand(300, 3, requires_grad=True)
    Cf = torch.rand(300, 3, requires_grad=True)
    Ct = torch.rand(300, 3)
    ol = oracle.ray_loss(Cc, Cf, Ct)
    ol.backward()
    m = pkg.NeRFModel(64, 128, 300)
    a = Cc.detach().to(dev).requires_grad_(True)
    b = Cf.detach().to(dev).requires_grad_(True)
    l = m.ray_loss(a, b, Ct.to(dev))
    l.backward()
    assert abs(float(l) - float(ol)) < 1e-5 * float(ol)
    assert torch.allclose(a.grad.cpu(), Cc.grad, rtol=1e-6, atol=1e-7)
    assert torch.allclose(b.grad.cpu(), Cf.grad, rtol=1e-6, atol=1e-7)


def test_sum_of_shard_gradients_equals_full_batch(oracle, pkg, dev):
    """multi-GPU contract checked on one GPU: the loss is a SUM over rays, so the gradients of two
    half batches (global ray 0 forwarded for quirk Q6) add up to the full-batch gradient -- same kernels, same
    per-ray arithmetic, hence the same discrete decisions."""
    B = 512
    row, col, pb, K, Ct = oracle.fern_inputs(B, seed=21)
    w = oracle.make_weights(7, sharp=True)

    def grads(sl, ray0):
        n = sl.stop - sl.start
        m = pkg.NeRFModel(64, 128, n)
        m.load_state_dict(w)
        m = m.to(dev)
        m.ray0_near_far = ray0
        Cc, Cf = m(row[sl], col[sl], pb[sl], K)
        m.ray_loss(Cc, Cf, Ct[sl].to(dev)).backward()
        return [q.grad.detach().clone() for q in m.network.parameters()]

    full = grads(slice(0, B), None)
    r0 = (float(pb[0, 15].float()), float(pb[0, 16].float()))
    a = grads(slice(0, B // 2), r0)
    b = grads(slice(B // 2, B), r0)
    for f, x, y in zip(full, a, b):
        # relative to the size of the summands: the two halves may cancel (e.g. the scalar sigma bias)
        scale = float(x.double().norm() + y.double().norm())
        assert float((x.double() + y.double() - f.double()).norm()) < 2e-5 * scale
