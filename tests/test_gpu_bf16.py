"""GPU checks of the bf16-MLP variant (BASELINE.json cfg3 "bf16 MLP / fp32 composite", flag NERF_HIP_BF16_MLP).

The reference has no reduced precision, so there is no reference output to be identical to: the variant is SPECIFIED by
``oracle.mlp_bf16`` (bf16-rounded weights and layer inputs, fp32 accumulation / biases / activations) and checked
  (a) against that emulation -- differences come only from the fp32 summation order and 1-ulp encoding differences, which
      now and then flip a bf16 rounding of a hidden unit (one flip = 2^-8 relative on one of 256 inputs);
  (b) against the fp32 oracle, to state how far cfg3 is from the 1e-4 bar it cannot meet (SURVEY.md section 7 hard-6:
      3.8e-3 / 1.1e-2 measured for bf16 weights + activations).
Tolerances are written next to each assertion.
"""
import numpy as np
import pytest
import torch

from conftest import golden_inputs, load_golden, max_rel

pytestmark = pytest.mark.gpu


def _params_dev(oracle, seed, sharp, dev):
    p = oracle.make_weights(seed, sharp)
    return p, [v.to(dev).contiguous() for v in p.values()]


@pytest.mark.parametrize("name,N", [("cfg1_lego_crop32", 64), ("cfg4_fern_rand512", 128), ("small_16_32", 16), ("small_16_32", 37)])
def test_field_bf16_against_emulation(oracle, pkg, dev, name, N):
    g = load_golden(name)
    row, col, pb, K, _ = golden_inputs(g)
    B = min(row.shape[0], 256)
    row, col, pb = row[:B], col[:B], pb[:B]
    params, pd = _params_dev(oracle, int(g["seed"]), bool(g["sharp"]), dev)
    R, o, near, far = oracle.poses_extract(pb)
    gen = torch.Generator().manual_seed(7)
    t = near[:, None] + (far - near)[:, None] * torch.rand(B, N, generator=gen)
    rgb, sig = pkg.ops.field_bf16(pd, row.to(dev), col.to(dev), pb.float().to(dev), K, t.to(dev))
    d_cam = oracle.camera_dirs(row, col, K)
    fp, fd = oracle.frequencies()
    ogp = oracle.encode(oracle.sample_points(R, o, d_cam, t), fp)
    ogd = oracle.encode(oracle.world_dirs(R, d_cam), fd)[:, None, :].expand(-1, N, -1)
    with torch.no_grad():
        ergb, esig = oracle.mlp_bf16(params, ogp, ogd)
        frgb, fsig = oracle.mlp(params, ogp, ogd)
    # (a) against the emulation: 5e-3 of the largest value (rounding flips, see module docstring; measured <= 2.3e-3)
    assert max_rel(sig, esig) < 5e-3, max_rel(sig, esig)
    assert float((rgb.cpu() - ergb).abs().max()) < 5e-3
    # (b) against fp32: 3e-2 (bf16 has 8 significant bits; 10 layers deep)
    assert max_rel(sig, fsig) < 3e-2
    assert float((rgb.cpu() - frgb).abs().max()) < 3e-2
    # most outputs agree with the emulation far better than the bound
    assert float(((sig.cpu() - esig).abs() / esig.abs().max()).median()) < 1e-4


@pytest.mark.parametrize("name", ["cfg1_lego_crop32", "small_16_32"])
def test_forward_bf16(oracle, pkg, dev, name):
    """whole forward with the flag: against the emulated render and against fp32"""
    g = load_golden(name)
    row, col, pb, K, _ = golden_inputs(g)
    Nc, Nf = int(g["Nc"]), int(g["Nf"])
    params, _ = _params_dev(oracle, int(g["seed"]), bool(g["sharp"]), dev)
    m = pkg.NeRFModel(Nc, Nf, row.shape[0]).to(dev)
    m.load_state_dict(params)
    with torch.no_grad():
        C_c32, C_f32 = m(row.to(dev), col.to(dev), pb.to(dev), K)
        m.bf16_mlp = True
        C_c, C_f = m(row.to(dev), col.to(dev), pb.to(dev), K)
        E_c, E_f = oracle.render(params, row, col, pb, K, Nc, Nf, mlp=oracle.mlp_bf16, check=False)
    assert max_rel(C_c, E_c) < 2e-3, max_rel(C_c, E_c)
    assert max_rel(C_f, E_f) < 1e-2, max_rel(C_f, E_f)      # the fine pass amplifies (inverse-CDF resampling + sort)
    assert max_rel(C_c, C_c32) < 1e-2, max_rel(C_c, C_c32)
    assert max_rel(C_f, C_f32) < 5e-2, max_rel(C_f, C_f32)
    assert not torch.equal(C_c, C_c32)                      # the flag really selects another kernel


# ---------------------------------------------------------------------------------------------------------------
# training (forward with saving + backward chain + weight-gradient GEMMs, all on bf16 MFMA)
# ---------------------------------------------------------------------------------------------------------------
BS_KS = [4] + [16] * 8 + [8, 2]         # bf16_common.h: gamma_p, h0..h7, c, gamma_d (point_info is folded into dir_info: no feat)
BG_KS = [16] * 8 + [8, 2]               # dpre0..7, dpre_dir, (dz, dspre)


def _wave_blocks(B, N):
    return ((B * N + 255) // 256) * 8


def _decode(buf, wb_tot, ks_list, tensor, wb0, nwb):
    """fragment layout -> [nwb*32 samples, 16*ks features] fp32: piece (wb, ks), lane (j, h), slot s holds feature
    16ks + 4h + (s&3) + 8(s>>2) of sample wb*32 + j"""
    ks_t = ks_list[tensor]
    start = wb_tot * 1024 * sum(ks_list[:tensor])
    raw = buf[start:start + wb_tot * ks_t * 1024].view(torch.bfloat16).view(wb_tot, ks_t, 2, 32, 8)[wb0:wb0 + nwb].float().cpu()
    out = torch.zeros(nwb, 32, ks_t * 16)
    for ks in range(ks_t):
        for h in range(2):
            for sl in range(8):
                out[:, :, 16 * ks + 4 * h + (sl & 3) + 8 * (sl >> 2)] = raw[:, ks, h, :, sl]
    return out.reshape(nwb * 32, ks_t * 16)


def _bf16_model(pkg, params, Nc, Nf, B, dev):
    m = pkg.NeRFModel(Nc, Nf, B)
    m.load_state_dict(params)
    m = m.to(dev)
    m.bf16_mlp = True
    return m


@pytest.mark.parametrize("name", ["small_16_32", "cfg1_lego_crop32"])
def test_bf16_saved_activations_and_gradients_coarse_only(oracle, pkg, dev, name):
    """coarse-only loss (d loss / d C_fine = 0: no sort, no resampling in the gradient path -> well conditioned).
    Saved layer inputs against the emulation; weight gradients against autograd THROUGH the emulation.  The kernels
    additionally round every gradient that enters an MFMA to bf16 (standard mixed precision): 2^-9 per element and
    layer, random -> L2-rel <= 3e-2 per tensor (measured: see assertion messages), cosine > 0.999."""
    from nerf_tiny_amd import _abi

    g = load_golden(name)
    row, col, pb, K, Ct = golden_inputs(g)
    Nc, Nf, B = int(g["Nc"]), int(g["Nf"]), row.shape[0]
    params, _ = _params_dev(oracle, int(g["seed"]), bool(g["sharp"]), dev)
    m = _bf16_model(pkg, params, Nc, Nf, B, dev)
    Cc, Cf = m(row.to(dev), col.to(dev), pb.to(dev), K)
    flags = _abi.SAVE_FOR_BACKWARD | _abi.BF16_MLP
    ws = m.last_workspace
    wb_c, wb_tot = _wave_blocks(B, Nc), _wave_blocks(B, Nc) + _wave_blocks(B, Nf)
    bsave = _abi.ws_view(ws, B, Nc, Nf, flags, "bsave", (wb_tot * sum(BS_KS) * 1024,), torch.uint8)

    # emulation with autograd
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    st = {}
    Ec, Ef = oracle.render(p, row, col, pb, K, Nc, Nf, stages=st, mlp=oracle.mlp_bf16, check=False)
    f_p, _ = oracle.frequencies()
    with torch.no_grad():
        _, _, hidden, feat, cc = oracle.mlp_bf16(params, oracle.encode(st["pts_c"], f_p), st["gd"][:, None, :].expand(-1, Nc, -1), return_hidden=True)
    M = B * Nc
    for tensor, ref in [(1 + l, hidden[l]) for l in (0, 3, 7)] + [(9, cc)]:
        got = _decode(bsave, wb_tot, BS_KS, tensor, 0, wb_c)[:M]
        ref = ref.reshape(M, -1)
        d = (got[:, :ref.shape[1]] - ref).abs()
        scale = float(ref.abs().max())
        assert float(d.max()) < 3e-2 * scale, (tensor, float(d.max()), scale)          # a flipped rounding upstream shows here
        assert float((d > 1e-6 * scale).float().mean()) < 0.05, tensor                 # ... but most entries are identical
    loss = torch.sum(torch.square(Cc - Ct.to(dev)))
    loss.backward()
    eloss = torch.sum(torch.square(Ec - Ct))
    eloss.backward()
    assert abs(float(loss.detach()) - float(eloss.detach())) < 5e-3 * abs(float(eloss.detach()))
    for (k, pe), pm in zip(p.items(), m.network.parameters()):
        ge, gm = pe.grad.double().flatten(), pm.grad.double().flatten().cpu()
        assert torch.isfinite(gm).all(), k
        rel = float((gm - ge).norm() / ge.norm().clamp_min(1e-30))
        cos = float(torch.dot(gm, ge) / (gm.norm() * ge.norm()).clamp_min(1e-30))
        assert rel < 3e-2 and cos > 0.999, (k, rel, cos)


@pytest.mark.parametrize("name", ["small_16_32", "cfg1_lego_crop32"])
def test_bf16_full_loss_gradients(oracle, pkg, dev, name):
    """whole train step with both colours in the loss.  The full gradient is ill-conditioned already in fp32
    (tests/test_gpu_backward.py: per-channel sort ties, ReLU kinks on the t_fine path); here it must be finite, give the
    emulation's loss and stay inside the same band against autograd through the emulation."""
    g = load_golden(name)
    row, col, pb, K, Ct = golden_inputs(g)
    Nc, Nf, B = int(g["Nc"]), int(g["Nf"]), row.shape[0]
    params, _ = _params_dev(oracle, int(g["seed"]), bool(g["sharp"]), dev)
    m = _bf16_model(pkg, params, Nc, Nf, B, dev)
    Cc, Cf = m(row.to(dev), col.to(dev), pb.to(dev), K)
    loss = m.ray_loss(Cc, Cf, Ct.to(dev))
    loss.backward()
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    Ec, Ef = oracle.render(p, row, col, pb, K, Nc, Nf, mlp=oracle.mlp_bf16, check=False)
    eloss = oracle.ray_loss(Ec, Ef, Ct)
    eloss.backward()
    assert abs(float(loss.detach()) - float(eloss.detach())) < 2e-2 * abs(float(eloss.detach()))
    # The bar, per tensor (as tests/test_gpu_backward.py::test_train_step_end_to_end does for fp32, here THROUGH the bf16 emulation): twice
    # the emulation's OWN gradient shift when every value it rounds to bf16 is first moved by a seeded relative 1e-6 (oracle.
    # mlp_bf16_jittered: another correct evaluation of the same specification -- a few flipped bf16 roundings, which then flip sort
    # ties and ReLU kinks on the t_fine path), maximum over two seeds; floor 3e-2 = the bar of the well-conditioned coarse-only gradients
    # above (the kernels additionally round every gradient that enters an MFMA to bf16).
    g0 = {k: pe.grad.detach().double().flatten() for k, pe in p.items()}
    band = {k: 0.0 for k in g0}
    for sd in (1, 2):
        pj = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        Jc, Jf = oracle.render(pj, row, col, pb, K, Nc, Nf, mlp=oracle.mlp_bf16_jittered(sd), check=False)
        oracle.ray_loss(Jc, Jf, Ct).backward()
        for k in band:
            band[k] = max(band[k], float((pj[k].grad.double().flatten() - g0[k]).norm() / g0[k].norm().clamp_min(1e-30)))
    worst = (0.0, 1.0, "")
    for (k, ge), pm in zip(g0.items(), m.network.parameters()):
        gm = pm.grad.double().flatten().cpu()
        assert torch.isfinite(gm).all(), k
        e = float((gm - ge).norm() / ge.norm().clamp_min(1e-30))
        bar = max(2.0 * band[k], 3e-2)
        if e / bar > worst[0] / worst[1]:
            worst = (e, bar, k)
        assert e < bar, (k, e, band[k])
    print(f"{name}: bf16 full-loss gradient closest to its bar: {worst[2]} L2-rel {worst[0]:.2e} (bar {worst[1]:.2e}; the emulation's own "
          f"1e-6 jitter bands span {min(band.values()):.1e} .. {max(band.values()):.1e})")


@pytest.mark.parametrize("B,Nc,Nf", [(7, 5, 3), (33, 100, 200), (3, 1024, 1024), (130, 31, 65)])
def test_bf16_ragged_and_maximum_sizes_forward(oracle, pkg, dev, B, Nc, Nf):
    """inference with sizes that are not multiples of the 256-sample workgroups, wave blocks that straddle rays, and the
    largest Nc / Nf, against the emulated render (bars as in test_forward_bf16, fine pass looser for long rays)"""
    row, col, pb, K, _ = oracle.fern_inputs(B, seed=B)
    w = oracle.make_weights(8, sharp=True)
    m = _bf16_model(pkg, w, Nc, Nf, B, dev)
    with torch.no_grad():
        Cc, Cf = m(row, col, pb, K)
        ec, ef = oracle.render(w, row, col, pb, K, Nc, Nf, mlp=oracle.mlp_bf16, check=False)
    assert torch.isfinite(Cc).all() and torch.isfinite(Cf).all()
    assert max_rel(Cc, ec) < 5e-3, max_rel(Cc, ec)
    assert max_rel(Cf, ef) < 3e-2, max_rel(Cf, ef)


def test_bf16_inference_forms(oracle, pkg, dev):
    """inference runs on the 16x16x32 MFMA form, training on 32x32x16: the two agree to summation order (a few flipped
    bf16 roundings); with force_tile_kernel the inference uses the training form and equals a training forward bit for bit"""
    g = load_golden("cfg1_lego_crop32")
    row, col, pb, K, _ = golden_inputs(g)
    Nc, Nf, B = int(g["Nc"]), int(g["Nf"]), row.shape[0]
    params, _ = _params_dev(oracle, int(g["seed"]), bool(g["sharp"]), dev)
    m = _bf16_model(pkg, params, Nc, Nf, B, dev)
    Tc, Tf = m(row.to(dev), col.to(dev), pb.to(dev), K)          # training form (grad enabled)
    with torch.no_grad():
        Ic, If = m(row.to(dev), col.to(dev), pb.to(dev), K)      # 16x16x32
        m.force_tile_kernel = True
        Fc, Ff = m(row.to(dev), col.to(dev), pb.to(dev), K)      # 32x32x16, inference variant
    assert torch.equal(Fc, Tc.detach()) and torch.equal(Ff, Tf.detach())
    assert not torch.equal(Ic, Fc)
    assert max_rel(Ic, Fc) < 2e-3 and max_rel(If, Ff) < 1e-2


def test_bf16_ragged_sizes(oracle, pkg, dev):
    """pass sizes that are not multiples of the 256-sample workgroup / 32-sample wave block (B*Nc = 132, B*Nf = 220):
    the padded lanes must contribute nothing to any gradient"""
    B, Nc, Nf = 11, 12, 20
    row, col, pb, K, Ct = oracle.lego_inputs(B, seed=5)
    params = oracle.make_weights(3)
    m = _bf16_model(pkg, params, Nc, Nf, B, dev)
    Cc, Cf = m(row.to(dev), col.to(dev), pb.to(dev), K)
    loss = torch.sum(torch.square(Cc - Ct.to(dev)))
    loss.backward()
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    Ec, Ef = oracle.render(p, row, col, pb, K, Nc, Nf, mlp=oracle.mlp_bf16, check=False)
    torch.sum(torch.square(Ec - Ct)).backward()
    assert max_rel(Cc.detach(), Ec.detach()) < 2e-3 and max_rel(Cf.detach(), Ef.detach()) < 2e-2
    for (k, pe), pm in zip(p.items(), m.network.parameters()):
        ge, gm = pe.grad.double().flatten(), pm.grad.double().flatten().cpu()
        assert torch.isfinite(gm).all(), k
        rel = float((gm - ge).norm() / ge.norm().clamp_min(1e-30))
        assert rel < 3e-2, (k, rel)


def test_bf16_training_learns(pkg, dev):
    """a few fused-Adam steps in bf16 mode on a synthetic scene reduce the loss"""
    ds = pkg.data.synthetic_scene(n_pic=4, H=32, W=32, seed=3)
    rays = pkg.data.DeviceRays(ds, dev, seed=1)
    m = pkg.NeRFModel(32, 64, 1024).to(dev)
    m.bf16_mlp = True
    opt = pkg.train.FusedAdam(list(m.network.parameters()), lr=5e-4)
    K = torch.tensor([[1.0, 0.0, -16.0], [0.0, -1.0, 16.0], [0.0, 0.0, -float(ds.focal)]]).t()
    losses = []
    idx = torch.randperm(rays.num_pix, device=dev, generator=rays.gen)[:1024]
    row, col, pix, pb, _ = rays.gather(idx)
    for _ in range(30):
        opt.zero_grad()
        Cc, Cf = m(row, col, pb, K)
        loss = m.ray_loss(Cc, Cf, pix)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.6 * losses[0], (losses[0], losses[-1])


def test_bf16_training_trajectory_follows_the_emulation(oracle, pkg, dev):
    """SURVEY.md section 7 hard-6's acceptance for cfg3 -- "<= 2e-2 max-rel, and loss-curve agreement" -- as the bf16 twin of
    tests/test_gpu_train.py::test_training_trajectory_follows_the_oracle_with_torch_adam: 30 iterations of the reference's trainer loop
    (nerf.py:468-475; Adam betas (0.9, 0.999), eps 1e-7, the EXP schedule of nerf.py:425-426 with both branches walked), the bf16
    EMULATION (oracle.mlp_bf16 through the oracle's renderer, autograd) + torch.optim.Adam on the CPU against NeRFModel(bf16_mlp) +
    FusedAdam on the GPU, same ray batches and start weights.
    (a) Teacher-forced: at every step the device evaluates the loss at the EMULATION's current weights: within 1e-3 relative along the
        whole trajectory (a tenth of the bf16 forward bar -- C_fine's 1e-2 in test_forward_bf16; measured 7e-6: the loss is a sum over
        rays in which the flipped roundings of single hidden units average out).
    (b) Free-running: inside a band computed from the emulation's OWN drift -- its run with every to-be-rounded value jittered by a seeded
        relative 1e-6 (oracle.mlp_bf16_jittered) against its unjittered run: three times the maximum over two seeds per step (floor: the
        forward bar), for the loss curve and for the final weights.
    (c) The fp32-vs-bf16 gap of the loss curve is asserted: the bf16 device run ends within 10 % of the fp32 device run's final loss on
        the same batches and both learn -- reduced precision in the MLP does not change what the trainer converges towards."""
    B, Nc, Nf, steps, lr0, gamma, decay_end = 64, 16, 32, 30, 3e-4, 0.1, 20
    FWD_BAR = 1e-2
    lam = lambda it: gamma ** (it / decay_end) if it < decay_end else gamma * lr0  # nerf.py:426
    batches = [oracle.lego_inputs(B, seed=100 + s) for s in range(steps)]
    w0 = oracle.make_weights(3, sharp=False)

    def cpu_run(mlp_of_step, keep=False):
        params = {k: v.clone().requires_grad_(True) for k, v in w0.items()}
        opt = torch.optim.Adam([{"params": list(params.values()), "initial_lr": lr0}], lr=lr0, betas=(0.9, 0.999), eps=1e-7)
        sch = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lam)
        losses, snaps = [], []
        for s, (row, col, pb, K, Ct) in enumerate(batches):
            if keep:
                snaps.append({k: v.detach().clone() for k, v in params.items()})
            opt.zero_grad(set_to_none=True)
            Ec, Ef = oracle.render(params, row, col, pb, K, Nc, Nf, mlp=mlp_of_step(s), check=False)
            loss = oracle.ray_loss(Ec, Ef, Ct)
            loss.backward()
            opt.step()
            sch.step()
            losses.append(float(loss.detach()))
        return losses, {k: v.detach().clone() for k, v in params.items()}, snaps

    ref_l, ref_w, snaps = cpu_run(lambda s: oracle.mlp_bf16, keep=True)
    band_l, band_w = [0.0] * steps, {k: 0.0 for k in w0}
    for sd in (1, 2):
        pl, pw, _ = cpu_run(lambda s: oracle.mlp_bf16_jittered(1000 * sd + s))
        band_l = [max(b, abs(a - r)) for b, a, r in zip(band_l, pl, ref_l)]
        for k in band_w:
            band_w[k] = max(band_w[k], float((pw[k] - ref_w[k]).norm() / ref_w[k].norm()))

    def device_model(w, bf16):
        m = pkg.NeRFModel(Nc, Nf, B)
        m.load_state_dict(w)
        m = m.to(dev)
        m.bf16_mlp = bf16
        return m

    # (a) teacher-forced
    tf = device_model(w0, True)
    worst_tf = 0.0
    with torch.no_grad():
        for s, (row, col, pb, K, Ct) in enumerate(batches):
            tf.load_state_dict(snaps[s])
            tf.force_tile_kernel = True  # the 32x32x16 form, i.e. the forward half of a training call
            Cc, Cf = tf(row, col, pb, K)
            e = abs(float(tf.ray_loss(Cc, Cf, Ct.to(dev))) - ref_l[s]) / ref_l[s]
            worst_tf = max(worst_tf, e)
            assert e <= 0.1 * FWD_BAR, (s, e)

    # (b) free-running, bf16; (c) the fp32 device run on the same batches
    def device_run(bf16):
        m = device_model(w0, bf16)
        opt = pkg.FusedAdam([{"params": list(m.network.parameters()), "initial_lr": lr0}], lr=lr0, betas=(0.9, 0.999), eps=1e-7)
        sch = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lam)
        out = []
        m.train()
        for row, col, pb, K, Ct in batches:
            opt.zero_grad(set_to_none=True)
            Cc, Cf = m(row, col, pb, K)
            loss = m.ray_loss(Cc, Cf, Ct.to(dev))
            loss.backward()
            opt.step()
            sch.step()
            out.append(float(loss.detach()))
        return out, m

    dev_l, m = device_run(True)
    worst = 0.0
    for s, (a, r, b) in enumerate(zip(dev_l, ref_l, band_l)):
        bar = max(3.0 * b, FWD_BAR * abs(r))
        worst = max(worst, abs(a - r) / bar)
        assert abs(a - r) <= bar, (s, a, r, b)
    for (k, v), q in zip(ref_w.items(), m.network.parameters()):
        e = float((q.detach().cpu() - v).norm() / v.norm())
        assert e <= max(3.0 * band_w[k], 1e-3), (k, e, band_w[k])
    f32_l, _ = device_run(False)
    gap = abs(dev_l[-1] - f32_l[-1]) / f32_l[-1]
    curve_gap = max(abs(a - b) / b for a, b in zip(dev_l, f32_l))
    assert dev_l[-1] < 0.6 * dev_l[0] and ref_l[-1] < 0.6 * ref_l[0] and f32_l[-1] < 0.6 * f32_l[0]  # all three trainers learn
    assert gap < 0.10 and curve_gap < 0.15, (gap, curve_gap)
    print(f"bf16 trajectory: {steps} steps, loss {ref_l[0]:.3f} -> {ref_l[-1]:.3f} (emulation) / {dev_l[-1]:.3f} (device bf16) / {f32_l[-1]:.3f} (device fp32); "
          f"teacher-forced worst rel {worst_tf:.1e} (bar {0.1 * FWD_BAR:.0e}); free-running largest |dev - emulation| / bar = {worst:.2f}; the emulation's own "
          f"jitter drift at the last step {band_l[-1] / ref_l[-1]:.1e} rel; fp32-vs-bf16 final-loss gap {gap:.1e}, largest along the curve {curve_gap:.1e}")


def _variant(tmp_path, name, env_extra, B=96):
    import os
    import subprocess
    import sys

    from conftest import ROOT

    env = {k: v for k, v in os.environ.items() if k not in ("NERF_PREP_BF16", "NERF_DW_BF16_MULTI", "NERF_PAIR_BF16", "NERF_BF16_4WAVE", "NERF_DW_BF16_SMALLGROUP", "NERF_FUSE_RAYS")}
    env.update(env_extra)
    out = str(tmp_path / (name + ".pt"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "bf16_variant_dump.py"), out, str(B)], capture_output=True, text=True,
                       env=env, timeout=400)
    assert r.returncode == 0 and "DUMP-OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    return torch.load(out, weights_only=False)


@pytest.mark.timeout(900)
def test_one_launch_preparation_equals_separate_launches(tmp_path):
    """prep_bf16.hip: fold + packed image(s) + ray records of a bf16-MLP call in ONE launch, the packers of folded fragments waiting for the
    fold blocks inside the launch (agent-scope release / acquire on a {token, count} word).  Against the round-3 structure (fold, pack, rays
    as separate launches, the backward packing its own image: NERF_PREP_BF16=0, a process of its own -- the switch is read once): the packed
    images, ray records and coarse depths byte for byte, the outputs and -- with the same weight-gradient structure on both sides -- every
    gradient bit for bit; no timeout reported; the frozen rendering loop (rays only, image reused) gives the inference call's pixels."""
    new = _variant(tmp_path, "prep1", {"NERF_DW_BF16_MULTI": "1"})
    old = _variant(tmp_path, "prep0", {"NERF_PREP_BF16": "0", "NERF_DW_BF16_MULTI": "1"})
    assert new["sticky"] == 0 and old["sticky"] == 0
    for k in ("train_packed_bf", "infer_packed_bf", "rayf", "t_c", "Cc", "Cf", "Ic", "If", "Fc", "Ff", "Sc", "Sf"):
        assert torch.equal(new[k], old[k]), k
    assert new["loss"] == old["loss"]
    for a, b in zip(new["grads"], old["grads"]):
        assert torch.equal(a, b)
    assert torch.equal(new["Fc"], new["Ic"]) and torch.equal(new["Ff"], new["If"])


@pytest.mark.timeout(900)
def test_weight_gradients_in_one_launch_equal_a_launch_per_product(tmp_path):
    """dw_bf16.hip k_dw_bf16_multi (small batches: every product of the step in ONE launch, the workgroups dealt out by bytes) against a
    launch per product (NERF_DW_BF16_MULTI=0): the same sums over the same samples split into a different number of slabs -- equal to fp32
    summation order (1e-5 of each tensor's norm), forward untouched."""
    for B in (96, 333):  # 333 rays: more wave blocks than workgroups for some products, fewer for others
        one = _variant(tmp_path, f"multi1_{B}", {"NERF_DW_BF16_MULTI": "1"}, B)
        per = _variant(tmp_path, f"multi0_{B}", {"NERF_DW_BF16_MULTI": "0"}, B)
        assert torch.equal(one["Cc"], per["Cc"]) and torch.equal(one["Cf"], per["Cf"]) and one["loss"] == per["loss"]
        for i, (a, b) in enumerate(zip(one["grads"], per["grads"])):
            assert torch.isfinite(a).all()
            assert float((a.double() - b.double()).norm()) <= 1e-5 * float(b.double().norm()) + 1e-12, (B, i)


@pytest.mark.timeout(900)
def test_pair_kernel_equals_separate_launches(tmp_path):
    """field_fwd_bf16x.hip k_render_pair_bf16x (small bf16-MLP inference batches, Nc = 64 / Nf = 128): both field passes, the coarse
    composite + resampling and the merge + channel sorts + composite of a ray PAIR in one workgroup and ONE launch, against the four
    separate launches (NERF_PAIR_BF16=0, a process of its own) -- ray records made in the kernel too, so a rendering loop is ONE launch per
    call.  Same functions (ray_record, bx_field_pass, ray_parts.h), so the same bits: the two colours, the ray records and every per-sample
    buffer of the workspace, the status word; an odd batch (the last workgroup holds ONE ray), a shard that is
    handed the global ray 0's near / far, a ray that meets the reference's exit(0) condition."""
    for B in (97, 256):
        one = _variant(tmp_path, f"pair1_{B}", {"NERF_PAIR_BF16": "1"}, B)
        sep = _variant(tmp_path, f"pair0_{B}", {"NERF_PAIR_BF16": "0"}, B)
        for k in ("Ic", "If", "Fc", "Ff", "Sc", "Sf", "infer_rayf", "infer_t_c", "infer_sig_c", "infer_rgb_c", "infer_w_c", "infer_t_f", "infer_sig_f",
                  "infer_rgb_f"):
            assert torch.equal(one[k], sep[k]), (B, k)
        # the status word: stamped with the call's generation in the pair kernel (no kernel in front of it to zero a word), legacy otherwise
        assert one["S_fault"] is True and sep["S_fault"] is True
        assert one["S_fault_after_healthy"] is False and sep["S_fault_after_healthy"] is False
        assert one["sticky"] & 2 == 0


@pytest.mark.timeout(900)
def test_four_wave_training_workgroups_equal_eight_wave_ones(tmp_path):
    """field_fwd_bf16.hip / field_bwd_bf16.hip: a SMALL pass runs 4-wave workgroups (128 samples, one wave per SIMD) so that every CU gets one
    (the coarse pass of a 512-ray batch is 128 workgroups of 256 samples); NERF_BF16_4WAVE=0 keeps the 8-wave ones.  The same waves on the
    same wave blocks with the same fragment stream: outputs, saved activations and therefore every gradient bit for bit."""
    for B in (96, 333):
        four = _variant(tmp_path, f"w4_{B}", {"NERF_DW_BF16_MULTI": "1"}, B)
        eight = _variant(tmp_path, f"w8_{B}", {"NERF_BF16_4WAVE": "0", "NERF_DW_BF16_MULTI": "1"}, B)
        for k in ("Cc", "Cf", "train_packed_bf"):
            assert torch.equal(four[k], eight[k]), (B, k)
        assert four["loss"] == eight["loss"]
        for i, (a, b) in enumerate(zip(four["grads"], eight["grads"])):
            assert torch.equal(a, b), (B, i)


@pytest.mark.timeout(900)
def test_ray_stages_fused_into_the_field_launches_equal_separate_launches(tmp_path):
    """Small bf16 training batches (Nc = 64, Nf = 128): k_coarse / k_merge run as EPILOGUES of the coarse / fine forward launch and
    k_merge_bwd / k_coarse_bwd as PROLOGUES of the fine / coarse chain launch, on the workgroup's own rays (kernels.h FwdFuse / BwdFuse)
    instead of four launches of their own (NERF_FUSE_RAYS=0, a process of its own).  The same functions (ray_parts.h, ray_parts_bwd.h) on the
    same buffers: colours, loss, the resampled depths and every gradient bit for bit; an odd batch (a workgroup with a ray behind the
    batch) and one with several workgroup rounds."""
    for B in (97, 256, 400):
        fused = _variant(tmp_path, f"fuse1_{B}", {"NERF_FUSE_RAYS": "1", "NERF_DW_BF16_MULTI": "1"}, B)
        sep = _variant(tmp_path, f"fuse0_{B}", {"NERF_FUSE_RAYS": "0", "NERF_DW_BF16_MULTI": "1"}, B)
        for k in ("Cc", "Cf", "train_t_f", "train_w_c", "train_bundle", "train_perm"):
            assert torch.equal(fused[k], sep[k]), (B, k)
        assert fused["loss"] == sep["loss"]
        for i, (a, b) in enumerate(zip(fused["grads"], sep["grads"])):
            assert torch.equal(a, b), (B, i)


def test_preparation_timeout_is_loud(oracle, pkg, dev, monkeypatch):
    """VERDICT round 4 item 4(ii) / ADVICE: the one-launch preparation's in-launch wait is bounded, and a timeout must SURFACE.  With
    NERF_PREP_FAULT_INJECT=1 (the fold blocks publish nobody's token, the bound is 2^10 polls) every waiting block times out: the packed
    fragments that needed the fold are NaN, so are C_coarse / C_fine / loss / gradients; nerf_hip_read_status reports PREP_TIMEOUT for that
    call AND for a later call that reuses the image (NERF_HIP_WEIGHTS_UNCHANGED); read_status() / render() / resample_fault_since() raise.
    A healthy call afterwards packs a clean image and reports nothing."""
    import ctypes as C

    from nerf_tiny_amd import _abi

    B, Nc, Nf = 96, 64, 128
    row, col, pb, K, Ct = oracle.lego_inputs(B, seed=11)
    w = oracle.make_weights(5, sharp=True)
    m = pkg.NeRFModel(Nc, Nf, B)
    m.load_state_dict(w)
    m = m.to(dev)
    m.bf16_mlp = True

    def status():
        ws = m.last_workspace
        st = C.c_uint32(0)
        _abi.check(_abi.lib().nerf_hip_read_status(ws.data_ptr(), ws.numel(), C.byref(st), torch.cuda.current_stream(dev).cuda_stream))
        return st.value

    with torch.no_grad():
        Cc, Cf = m(row, col, pb, K)
    assert torch.isfinite(Cf).all() and status() & _abi.STATUS_PREP_TIMEOUT == 0
    monkeypatch.setenv("NERF_PREP_FAULT_INJECT", "1")
    # inference: poisoned image -> NaN pixels, flagged
    with torch.no_grad():
        with m.frozen_weights():
            Cc, Cf = m(row, col, pb, K)
            assert torch.isnan(Cf).all() and torch.isnan(Cc).all()
            assert status() & _abi.STATUS_PREP_TIMEOUT
            Cc2, Cf2 = m(row, col, pb, K)  # reuses the poisoned image (no packing in this call): still NaN, still flagged
            assert torch.isnan(Cf2).all() and status() & _abi.STATUS_PREP_TIMEOUT
        with pytest.raises(_abi.NerfHipError):
            m.read_status()
        with pytest.raises(_abi.NerfHipError):
            m.render(row, col, pb, K)
    # training: NaN loss and gradients, the runner's logging-point check raises
    Cc, Cf, loss = m.train_step(row, col, pb, K, Ct)
    assert torch.isnan(loss) and any(bool(torch.isnan(p.grad).any()) for p in m.network.parameters())
    with pytest.raises(_abi.NerfHipError):
        m.resample_fault_since(clear=True)
    monkeypatch.delenv("NERF_PREP_FAULT_INJECT")
    with torch.no_grad():
        Cc, Cf = m(row, col, pb, K)
    assert torch.isfinite(Cf).all() and torch.isfinite(Cc).all() and status() & _abi.STATUS_PREP_TIMEOUT == 0
    Cc, Cf, loss = m.train_step(row, col, pb, K, Ct)
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in m.network.parameters())
    m.read_status()
    assert m.resample_fault_since(clear=True) in (False, True)
