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
reference's own fp32 noise band; (3) the untouched product path must give the loss to 1e-5, the well-conditioned
gradients to 1e-3 and the rest inside the reference's sensitivity band.
"""
import numpy as np
import pytest
import torch

from conftest import golden_inputs, l2_rel, load_golden

pytestmark = pytest.mark.gpu
GTOL = 1e-4          # well-conditioned gradients
NOISE_BAND = 2e-2    # full gradient with the reference's discrete decisions replayed (reference fp32-vs-fp64: ~1e-2)
CHAOS_BAND = 0.3     # full gradient with the device's own decisions (reference under a 1e-6 perturbation: up to 0.14)


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
@pytest.mark.parametrize("name", CASES)
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


# ---------------------------------------------------------------------------------------------------------------
# (3) the untouched product path
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", CASES + ["cfg2_lego_rand4096"])
def test_train_step_end_to_end(oracle, pkg, dev, name):
    g = load_golden(name)
    inputs = golden_inputs(g)
    w = oracle.make_weights(int(g["seed"]), bool(g["sharp"]))
    m, loss = _train_step(pkg, oracle, dev, w, inputs, int(g["Nc"]), int(g["Nf"]))
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    worst = 0.0
    for k, q in m.named_parameters():
        got = q.grad.detach().cpu().double()
        assert torch.isfinite(got).all()
        gn = float(g["gnorm_" + k])
        assert abs(float(got.norm()) - gn) <= 0.1 * gn, (k, float(got.norm()), gn)
        if "grad_" + k in g:
            ref = torch.from_numpy(g["grad_" + k]).double()
            e = float((got - ref).norm() / ref.norm())
        else:
            ref = torch.from_numpy(g["gslice_" + k]).double()
            e = float((got.flatten()[::97] - ref).norm() / max(float(ref.norm()), 1e-30))
        worst = max(worst, e)
        assert e < (1e-3 if "color_layer" in k else CHAOS_BAND), (k, e)
    print(f"{name}: worst grad L2-rel (own decisions) {worst:.2e}")


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
