"""GPU parity: libnerf_hip.so (through the C ABI) against the CPU oracle and the committed golden vectors.

Tolerances (BASELINE.json north_star: <= 1e-4 rel fp32):
  * ray generation, coarse depths, sample points: BIT-exact (the fine pass is ill-conditioned in them);
  * encodings: 2e-6 abs (device sinf/cosf vs SLEEF, <= 2 ulp of values in [-1, 1]);
  * field outputs, weights, colours: max-rel <= 1e-4 (measured floor ~1e-6 .. 2e-5).
"""
import numpy as np
import pytest
import torch

from conftest import elementwise_rel, golden_inputs, load_golden, max_rel

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _params_dev(oracle, seed, sharp, dev):
    p = oracle.make_weights(seed, sharp)
    return p, [v.to(dev).contiguous() for v in p.values()]


@pytest.mark.parametrize("name", ["cfg1_lego_crop32", "cfg4_fern_rand512", "small_16_32"])
def test_rays_bit_exact(oracle, pkg, dev, name):
    g = load_golden(name)
    row, col, pb, K, _ = golden_inputs(g)
    Nc = int(g["Nc"])
    d_cam, d_wrd, t_c = pkg.ops.rays(row.to(dev), col.to(dev), pb.float().to(dev), K, Nc)
    R, o, near, far = oracle.poses_extract(pb)
    od = oracle.camera_dirs(row, col, K)
    assert torch.equal(d_cam.cpu(), od)
    assert torch.equal(d_wrd.cpu(), oracle.world_dirs(R, od))
    assert torch.equal(t_c.cpu(), oracle.coarse_depths(near, far, Nc))
    n = g["st_t_c"].shape[0]
    assert np.array_equal(t_c[:n].cpu().numpy(), g["st_t_c"])
    assert np.array_equal(d_wrd[:n].cpu().numpy(), g["st_d_wrd"])


@pytest.mark.parametrize("name,N", [("cfg1_lego_crop32", 64), ("cfg4_fern_rand512", 128), ("small_16_32", 16), ("small_16_32", 37)])
def test_field_against_oracle(oracle, pkg, dev, name, N):
    """points bit-exact, encoding ~ulp, rgb/sigma within 1e-4 (coarse depths or random depths)."""
    g = load_golden(name)
    row, col, pb, K, _ = golden_inputs(g)
    B = min(row.shape[0], 256)
    row, col, pb = row[:B], col[:B], pb[:B]
    params, pd = _params_dev(oracle, int(g["seed"]), bool(g["sharp"]), dev)
    R, o, near, far = oracle.poses_extract(pb)
    gen = torch.Generator().manual_seed(7)
    t = near[:, None] + (far - near)[:, None] * torch.rand(B, N, generator=gen)
    rgb, sig, pts, gp = pkg.ops.field(pd, row.to(dev), col.to(dev), pb.float().to(dev), K, t.to(dev), debug=True)
    d_cam = oracle.camera_dirs(row, col, K)
    opts = oracle.sample_points(R, o, d_cam, t)
    assert torch.equal(pts.cpu(), opts)
    fp, fd = oracle.frequencies()
    ogp = oracle.encode(opts, fp)
    assert float((gp.cpu() - ogp).abs().max()) < 2e-6
    ogd = oracle.encode(oracle.world_dirs(R, d_cam), fd)
    with torch.no_grad():
        orgb, osig = oracle.mlp(params, ogp, ogd[:, None, :].expand(-1, N, -1))
    assert max_rel(sig, osig) < TOL
    assert max_rel(rgb, orgb) < TOL


def test_field_matches_golden_stage_vectors(oracle, pkg, dev):
    g = load_golden("cfg1_lego_crop32_sharp")
    row, col, pb, K, _ = golden_inputs(g)
    n = g["st_t_c"].shape[0]
    params, pd = _params_dev(oracle, int(g["seed"]), bool(g["sharp"]), dev)
    t = torch.from_numpy(g["st_t_c"])
    rgb, sig, pts, gp = pkg.ops.field(pd, row[:n].to(dev), col[:n].to(dev), pb[:n].float().to(dev), K, t.to(dev), debug=True)
    assert np.array_equal(pts.cpu().numpy(), g["st_pts_c"])
    assert float(np.abs(gp.cpu().numpy() - g["st_gp_c"]).max()) < 2e-6
    assert max_rel(sig, g["st_sig_c"]) < TOL
    assert max_rel(rgb, g["st_rgb_c"]) < TOL


@pytest.mark.parametrize("name", ["cfg1_lego_crop32_sharp", "cfg4_fern_rand512", "small_16_32"])
def test_coarse_composite_and_merge(oracle, pkg, dev, name):
    """Feed the ORACLE's sigma/rgb to the composite kernels: isolates rows a6-a9 from the MLP."""
    g = load_golden(name)
    row, col, pb, K, _ = golden_inputs(g)
    Nc, Nf = int(g["Nc"]), int(g["Nf"])
    params = oracle.make_weights(int(g["seed"]), bool(g["sharp"]))
    st = {}
    with torch.no_grad():
        Cc, Cf = oracle.render(params, row, col, pb, K, Nc, Nf, stages=st)
    d = lambda x: x.contiguous().to(dev)
    delta0 = float(st["t_c"][0, 1] - st["t_c"][0, 0])
    w_c, C_c, t_f, status = pkg.ops.coarse_composite(d(st["t_c"]), d(st["sig_c"]), d(st["rgb_c"]), st["near"], st["far"], delta0, Nf)
    assert status == 0
    assert max_rel(w_c, st["w_c"]) < 1e-5
    assert max_rel(C_c, Cc) < 1e-5
    # t_f is continuous in the CDF: device expf vs SLEEF moves samples by ulps, never across the tolerance
    assert float((t_f.cpu() - st["t_f"]).abs().max()) < 1e-5 * float(st["t_f"].abs().max())
    bundle, w, C_f = pkg.ops.merge_composite(d(st["t_c"]), d(st["t_f"]), d(st["sig_c"]), d(st["sig_f"]), d(st["rgb_c"]), d(st["rgb_f"]))
    assert torch.equal(bundle[:, :, 0].cpu(), st["t_s"])
    assert torch.equal(bundle[:, :, 1:4].cpu(), st["rgb_s"])
    assert torch.equal(bundle[:, :, 4].cpu(), st["sig_s"])
    assert max_rel(w, st["w"]) < 1e-5
    assert max_rel(C_f, Cf) < 1e-5


def test_resample_status_flag(oracle, pkg, dev):
    """quirk Q7: zero density -> the nerf.py:251 condition is reported as a status bit, never an abort."""
    B, Nc, Nf = 8, 64, 128
    t_c = torch.linspace(2, 6, Nc).repeat(B, 1)
    z = torch.zeros(B, Nc)
    rgb = torch.full((B, Nc, 3), 0.5)
    near, far = torch.full((B,), 2.0), torch.full((B,), 6.0)
    *_, status = pkg.ops.coarse_composite(t_c.to(dev), z.to(dev), rgb.to(dev), near, far, 4.0 / 63, Nf)
    assert status & pkg._abi.STATUS_RESAMPLE_INDEX


@pytest.mark.parametrize("name", ["cfg1_lego_crop32", "cfg1_lego_crop32_sharp", "cfg2_lego_rand4096", "cfg4_fern_rand512", "small_16_32"])
def test_forward_matches_golden(oracle, pkg, dev, name):
    """End to end through NeRFModel.__call__ (the reference's call surface) against the reference's outputs."""
    g = load_golden(name)
    row, col, pb, K, _ = golden_inputs(g)
    B = row.shape[0]
    m = pkg.NeRFModel(int(g["Nc"]), int(g["Nf"]), B)
    m.load_state_dict(oracle.make_weights(int(g["seed"]), bool(g["sharp"])))
    m = m.to(dev)
    m.check_resample = True
    with torch.no_grad():
        Cc, Cf = m(row, col, pb, K)
    ec, ef = max_rel(Cc, g["C_coarse"]), max_rel(Cf, g["C_fine"])
    # the same bar read element-wise: |a - b| / max(|b|, 1e-6) for every colour value (colours live in [0.08, 0.52] here)
    xc, xf = elementwise_rel(Cc, g["C_coarse"]), elementwise_rel(Cf, g["C_fine"])
    print(f"{name}: max-rel C_coarse {ec:.2e}  C_fine {ef:.2e};  element-wise rel C_coarse {xc:.2e}  C_fine {xf:.2e}")
    assert ec < TOL and ef < TOL
    assert xc < TOL and xf < TOL, (xc, xf)
    mse = float(((Cf.cpu() - torch.from_numpy(g["C_fine"])) ** 2).mean())
    assert 10 * np.log10(1.0 / max(mse, 1e-30)) > 80.0  # "PSNR vs ref" 


def test_forward_linearity_property_full_size(oracle, pkg, dev):
    """Full cfg2 size (4096 x (64+128)) without the oracle: rays are independent, so rendering a batch in two
    halves (with the global ray 0 passed for quirk Q6) must reproduce the one-shot result bit for bit."""
    row, col, pb, K, _ = oracle.lego_inputs(4096, seed=11)
    w = oracle.make_weights(5, sharp=True)
    m = pkg.NeRFModel(64, 128, 4096)
    m.load_state_dict(w)
    m = m.to(dev)
    with torch.no_grad():
        Cc, Cf = m(row, col, pb, K)
    h = pkg.NeRFModel(64, 128, 2048)
    h.load_state_dict(w)
    h = h.to(dev)
    h.ray0_near_far = (float(pb[0, 15]), float(pb[0, 16]))
    with torch.no_grad():
        a = h(row[:2048], col[:2048], pb[:2048], K)
        a = (a[0].clone(), a[1].clone())
        b = h(row[2048:], col[2048:], pb[2048:], K)
    assert torch.equal(torch.cat((a[0], b[0])), Cc)
    assert torch.equal(torch.cat((a[1], b[1])), Cf)
    assert torch.isfinite(Cf).all() and float(Cf.min()) >= 0.0


@pytest.mark.parametrize("B,Nc,Nf", [(7, 5, 3), (33, 100, 200), (3, 1024, 1024), (130, 31, 65)])
def test_ragged_and_maximum_sizes(oracle, pkg, dev, B, Nc, Nf):
    """sizes that are not multiples of the 32/64-sample tiles, tiles that straddle rays, and the largest Nc/Nf."""
    row, col, pb, K, _ = oracle.fern_inputs(B, seed=B)
    w = oracle.make_weights(8, sharp=True)
    m = pkg.NeRFModel(Nc, Nf, B)
    m.load_state_dict(w)
    m = m.to(dev)
    with torch.no_grad():
        Cc, Cf = m(row, col, pb, K)
        oc, of = oracle.render(w, row, col, pb, K, Nc, Nf)
    assert max_rel(Cc, oc) < TOL and max_rel(Cf, of) < TOL


def test_tile_and_register_kernels_agree(oracle, pkg, dev):
    """the LDS-tile kernels (NERF_HIP_FORCE_TILE_KERNEL) and the register-resident kernels run the same MFMA sequence:
    forward values to 1e-6, gradients to 1e-5."""
    B, Nc, Nf = 200, 64, 128
    row, col, pb, K, Ct = oracle.fern_inputs(B, seed=4)
    w = oracle.make_weights(9, sharp=True)
    outs = []
    for tile in (False, True):
        m = pkg.NeRFModel(Nc, Nf, B)
        m.load_state_dict(w)
        m = m.to(dev)
        m.force_tile_kernel = tile
        with torch.no_grad():
            inf = m(row, col, pb, K)
        Cc, Cf = m(row, col, pb, K)
        torch.sum(torch.square(Cc - Ct.to(dev))).backward()
        outs.append((inf, (Cc.detach(), Cf.detach()), [p.grad.clone() for p in m.network.parameters()]))
    a, b = outs
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        assert max_rel(x, y) < 1e-6
    assert max_rel(a[0][1], a[1][1]) < 1e-6  # inference and training forward agree
    for x, y in zip(a[2], b[2]):
        assert float((x - y).norm() / y.norm()) < 1e-5


def test_check_resample_raises_like_the_reference_exits(oracle, pkg, dev):
    """quirk Q7 through the model surface: zero density -> ResampleIndexError when check_resample is on, silent otherwise."""
    B = 16
    row, col, pb, K, _ = oracle.lego_inputs(B, seed=0)
    w = oracle.make_weights(0)
    w["network.sigma_layer.0.weight"].zero_()
    w["network.sigma_layer.0.bias"].zero_()
    m = pkg.NeRFModel(64, 128, B)
    m.load_state_dict(w)
    m = m.to(dev)
    with torch.no_grad():
        Cc, Cf = m(row, col, pb, K)  # no abort, finite output
        assert torch.isfinite(Cf).all()
        m.check_resample = True
        with pytest.raises(pkg.nerf.ResampleIndexError):
            m(row, col, pb, K)


def test_render_rows_sharded_frame_with_tail(oracle, pkg, dev):
    """frame rendering in shards: contiguous ray ranges per rank on the reference's batch grid, tail batch padded and cropped
    (the reference would drop it, nerf.py:442); two 'ranks' rendered one after the other on this GPU cover the list exactly
    once and every batch uses ITS OWN ray 0 for the resampling slope like the reference's display loop (quirk Q6).
    (Full-size frames: tests/test_gpu_configs.py.)"""
    n, Bm = 1000, 256
    row, col, pb, K, _ = oracle.fern_inputs(n, seed=12)  # per-ray near/far: the batch's ray 0 matters
    w = oracle.make_weights(3, sharp=True)
    m = pkg.NeRFModel(64, 128, Bm)
    m.load_state_dict(w)
    m = m.to(dev)
    m.ray0_near_far = (3.0, 4.0)  # a caller's setting survives the call
    out = torch.zeros(n, 3, device=dev)
    spans = []
    for rank in range(2):
        lo, hi, C = pkg.parallel.render_rows_sharded(m, row.to(dev), col.to(dev), pb.to(dev), K, rank, 2, out=out)
        spans.append((lo, hi))
        assert C.shape == (hi - lo, 3)
    assert spans == [(0, 512), (512, 1000)] and m.ray0_near_far == (3.0, 4.0)
    with torch.no_grad():
        for s in range(0, n, Bm):  # the reference's batches
            e = min(s + Bm, n)
            oc, of = oracle.render(w, row[s:e], col[s:e], pb[s:e], K, 64, 128)
            assert max_rel(out[s:e], of) < TOL, s
    # a ray split that ignores the batch grid returns the same pixels
    out2 = torch.zeros(n, 3, device=dev)
    for rank in range(3):
        pkg.parallel.render_rows_sharded(m, row.to(dev), col.to(dev), pb.to(dev), K, rank, 3, out=out2, align_to_batches=False)
    assert torch.equal(out, out2)


def test_packed_weights_are_reused_only_inside_a_frozen_section(oracle, pkg, dev):
    """rendering loops skip the weight re-packing (NERF_HIP_WEIGHTS_UNCHANGED) only inside `with model.frozen_weights():`;
    outside it every call repacks, so ANY parameter write is seen -- also the ones no version stamp can see (`p.data.add_()`,
    raw-pointer writes of the fused Adam kernel)"""
    B, Nc, Nf = 64, 16, 32
    row, col, pb, K, Ct = oracle.lego_inputs(B, seed=2)
    w = oracle.make_weights(5)
    m = pkg.NeRFModel(Nc, Nf, B)
    m.load_state_dict(w)
    m = m.to(dev)

    def check_against_oracle(out):
        oc, of = oracle.render({k: v.detach().cpu() for k, v in m.state_dict().items()}, row, col, pb, K, Nc, Nf)
        assert max_rel(out[0], oc) < TOL and max_rel(out[1], of) < TOL

    with torch.no_grad():
        a = m(row, col, pb, K)
        with m.frozen_weights():
            b = m(row, col, pb, K)  # packs
            b2 = m(row, col, pb, K)  # reuses the packed image
            assert (m.last_workspace.data_ptr(), 0) in m._packed
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b2[1]) and not m._packed
        m.network.point_info.bias.data.add_(0.05)  # invisible to torch's version counter
        c = m(row, col, pb, K)
        assert not torch.equal(a[0], c[0])
        check_against_oracle(c)
        with m.frozen_weights():  # a new section never trusts an older image
            c2 = m(row, col, pb, K)
        assert torch.equal(c[0], c2[0]) and torch.equal(c[1], c2[1])
    # update through the C ABI (fused Adam)
    opt = pkg.train.FusedAdam(list(m.network.parameters()), lr=1e-2)
    Cc, Cf = m(row, col, pb, K)
    m.ray_loss(Cc, Cf, Ct.to(dev)).backward()
    opt.step()
    with torch.no_grad():
        d = m(row, col, pb, K)
    assert not torch.equal(d[0], c[0])
    check_against_oracle(d)


def test_training_and_inference_workspaces_coexist(oracle, pkg, dev):
    """one workspace slot per flag set: an inference call between a training forward and its backward neither reallocates
    nor disturbs the saved activations (validate-while-training loops)"""
    B, Nc, Nf = 64, 16, 32
    row, col, pb, K, Ct = oracle.lego_inputs(B, seed=2)
    m = pkg.NeRFModel(Nc, Nf, B)
    m.load_state_dict(oracle.make_weights(5))
    m = m.to(dev)
    Cc, Cf = m(row, col, pb, K)
    m.ray_loss(Cc, Cf, Ct.to(dev)).backward()
    g0 = [p.grad.clone() for p in m.network.parameters()]

    def same(a, b):  # (bit-identical since round 2's deterministic sums; the relative bar is kept: it is what this test is about)
        return float((a.double() - b.double()).norm()) <= 1e-5 * float(b.double().norm())

    ws_train = m.last_workspace
    for p in m.network.parameters():
        p.grad = None
    Cc, Cf = m(row, col, pb, K)
    assert m.last_workspace.data_ptr() == ws_train.data_ptr()
    with torch.no_grad():
        m(row, col, pb, K)  # inference in between: its own slot
    assert m.last_workspace.data_ptr() != ws_train.data_ptr() and len(m._ws) == 2
    m.ray_loss(Cc, Cf, Ct.to(dev)).backward()
    for p, g in zip(m.network.parameters(), g0):
        assert same(p.grad, g)
    # a second TRAINING forward on the same slot invalidates the first one's backward
    C1 = m(row, col, pb, K)
    m(row, col, pb, K)
    with pytest.raises(RuntimeError):
        m.ray_loss(C1[0], C1[1], Ct.to(dev)).backward()


# ---------------------------------------------------------------------------------------------------------------
# the two documented divergences, fenced
# ---------------------------------------------------------------------------------------------------------------
def test_exact_ties_in_the_sorted_channels_do_not_change_the_forward(oracle, pkg, dev):
    """quirk Q1: five independent channel sorts.  torch.sort is not stable on the CPU, the kernel breaks ties by original
    index -- with EXACT ties in sigma / r / g / b (and in t) the sorted values, the weights and C_fine must still be
    bit-identical to a torch.sort of the same bundle, because equal values are interchangeable in the forward."""
    B, Nc, Nf = 64, 32, 64
    gen = torch.Generator().manual_seed(0)
    t_c = torch.sort(torch.rand(B, Nc, generator=gen) * 4 + 2, dim=1).values
    t_f = t_c[:, torch.randint(0, Nc, (Nf,), generator=gen)].clone()            # every fine depth duplicates a coarse one
    t_f[:, ::3] += 0.01
    sig_c = torch.randint(0, 4, (B, Nc), generator=gen).float() * 0.5            # four levels: masses of exact ties
    sig_f = torch.randint(0, 4, (B, Nf), generator=gen).float() * 0.5
    rgb_c = torch.randint(0, 3, (B, Nc, 3), generator=gen).float() * 0.25 + 0.25
    rgb_f = torch.randint(0, 3, (B, Nf, 3), generator=gen).float() * 0.25 + 0.25
    d = lambda x: x.to(dev)
    bundle, w, C_f = pkg.ops.merge_composite(d(t_c), d(t_f), d(sig_c), d(sig_f), d(rgb_c), d(rgb_f))
    ob = torch.cat((torch.cat((t_c, t_f), 1).unsqueeze(2), torch.cat((rgb_c, rgb_f), 1), torch.cat((sig_c, sig_f), 1).unsqueeze(2)), dim=2)
    sb, _ = torch.sort(ob, dim=1)
    assert torch.equal(bundle.cpu(), sb)
    delta = torch.cat((sb[:, 1:, 0] - sb[:, :-1, 0], torch.full((B, 1), 1e-4)), dim=1)
    ow = oracle.weights_from_sigma(delta, sb[:, :, 4])
    assert max_rel(w, ow) < 1e-5 and max_rel(C_f, oracle.composite(ow, sb[:, :, 1:4])) < 1e-5


@pytest.mark.parametrize("Nc,Nf", [(64, 128), (32, 64), (16, 32)])
def test_merge_sort_network_on_adversarial_values(oracle, pkg, dev, Nc, Nf):
    """The five per-channel sorts against torch.sort on values the renderer never produces but a sort must order anyway: negative
    numbers, both zeros, runs of duplicates, huge and tiny magnitudes.  64 + 128 samples run the register network (256 slots, keys =
    order-preserving unsigned images of the floats), the smaller counts the LDS network."""
    B = 96
    gen = torch.Generator().manual_seed(Nc)
    N = Nc + Nf

    def channel():
        x = torch.randn(B, N, generator=gen) * torch.tensor([1e-30, 1.0, 1e30])[torch.randint(0, 3, (B, N), generator=gen)]
        x = torch.where(torch.rand(B, N, generator=gen) < 0.2, torch.round(x.clamp(-3, 3)), x)          # duplicates, +0 and -0
        x = torch.where(torch.rand(B, N, generator=gen) < 0.05, -torch.zeros(B, N), x)
        return x

    ch = [channel() for _ in range(5)]
    t_all, r, g, b, sg = ch
    d = lambda x: x.contiguous().to(dev)
    rgb = torch.stack((r, g, b), dim=2)
    bundle, w, C_f = pkg.ops.merge_composite(d(t_all[:, :Nc]), d(t_all[:, Nc:]), d(sg[:, :Nc]), d(sg[:, Nc:]), d(rgb[:, :Nc]), d(rgb[:, Nc:]))
    ob = torch.cat((t_all.unsqueeze(2), rgb, sg.unsqueeze(2)), dim=2)
    sb, _ = torch.sort(ob, dim=1)
    got = bundle.cpu()
    assert torch.equal(got, sb)  # (== on values: a -0 and a +0 may trade places)
    assert bool(torch.isfinite(got).all())
    # NaNs (diverged weights can produce them): torch.sort puts every NaN last, whatever its sign bit
    nan_ch = [c.clone() for c in ch]
    for c in nan_ch:
        m = torch.rand(B, N, generator=gen) < 0.03
        c[m] = float("nan")
        c[torch.rand(B, N, generator=gen) < 0.01] = -torch.tensor(float("nan"))
    t_all, r, g, b, sg = nan_ch
    rgb = torch.stack((r, g, b), dim=2)
    bundle, w, C_f = pkg.ops.merge_composite(d(t_all[:, :Nc]), d(t_all[:, Nc:]), d(sg[:, :Nc]), d(sg[:, Nc:]), d(rgb[:, :Nc]), d(rgb[:, Nc:]))
    sb, _ = torch.sort(torch.cat((t_all.unsqueeze(2), rgb, sg.unsqueeze(2)), dim=2), dim=1)
    got = bundle.cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(sb))
    assert torch.equal(torch.nan_to_num(got, nan=0.0), torch.nan_to_num(sb, nan=0.0))


def test_near_equal_far_is_flagged_where_the_reference_exits(oracle, pkg, dev):
    """near == far (nerf.py:288: numpy.linspace with step == 0).  numpy then evaluates (i / div) * delta for EVERY ray of the
    batch instead of i * step -- one ulp different for the other rays -- which the kernels do not reproduce; it cannot be
    observed: the degenerate ray has zero weights, its resampling index is -1 and the reference exit(0)s on the whole batch
    (nerf.py:251-253, quirk Q7).  The library flags exactly that batch (status bit / ResampleIndexError), returns finite
    colours, and a batch WITHOUT such a ray is unaffected."""
    B = 32
    row, col, pb, K, _ = oracle.fern_inputs(B, seed=4)
    w = oracle.make_weights(2, sharp=True)
    m = pkg.NeRFModel(64, 128, B)
    m.load_state_dict(w)
    m = m.to(dev)
    with torch.no_grad():
        m.check_resample = True
        good = m(row, col, pb, K)                    # healthy batch: no flag
        oc, of = oracle.render(w, row, col, pb, K, 64, 128)
        assert max_rel(good[0], oc) < TOL and max_rel(good[1], of) < TOL
        pb2 = pb.clone()
        pb2[5, 16] = pb2[5, 15]                      # ray 5: far = near
        with pytest.raises(oracle.ResampleIndexError):
            oracle.render(w, row, col, pb2, K, 64, 128)
        with pytest.raises(pkg.nerf.ResampleIndexError):
            m(row, col, pb2, K)
        m.check_resample = False
        Cc, Cf = m(row, col, pb2, K)                 # no abort: finite output, nothing composited along the degenerate ray
        assert torch.isfinite(Cc).all() and torch.isfinite(Cf).all() and float(Cc[5].abs().max()) == 0.0
        d_cam, d_wrd, t_c = pkg.ops.rays(row.to(dev), col.to(dev), pb2.float().to(dev), K, 64)
        assert torch.equal(t_c[5].cpu(), torch.full((64,), float(pb2[5, 15].float())))  # numpy.linspace gives start everywhere, too


@pytest.mark.parametrize("mode", ["f32", "bf16", "split"])
def test_render_fuses_batches_bit_identically(oracle, pkg, dev, mode):
    """NeRFModel.render: the reference's batches [g*Bm, (g+1)*Bm) whose ray 0 has the same (near, far) share kernel calls (quirk Q6 is
    the only cross-ray term) -- every ray must get exactly the bits the per-batch `forward` gives it, whatever the kernel variant;
    where the pair changes inside the list the calls split; the tail batch is rendered with its own ray 0."""
    n, Bm = 2500, 400
    row, col, pb, K, _ = oracle.lego_inputs(n, seed=21)
    pb = pb.clone()
    pb[1300:, 15] = 2.5   # a second "picture" with its own near / far from ray 1300 on: batch 3 (rays 1200..1599) still starts in the first
    pb[1300:, 16] = 5.5
    w = oracle.make_weights(5, sharp=True)
    m = pkg.NeRFModel(64, 128, Bm)
    m.load_state_dict(w)
    m = m.to(dev)
    m.bf16_mlp, m.split_mlp = mode == "bf16", mode == "split"
    rd, cd, pd = row.to(dev), col.to(dev), pb.to(dev)
    nf0 = pb[::Bm, 15:17].float().tolist()
    plan = pkg.nerf.fuse_plan(nf0, n, Bm)
    assert [(s, e) for s, e, _, _ in plan] == [(0, 1600), (1600, 2500)]  # batches 0-3 (ray 0 in picture one), batches 4-6 (tail included)
    Cc, Cf = m.render(rd, cd, pd, K)
    Cc1, Cf1 = m.render(rd, cd, pd, K, fuse_rays=Bm)  # one call per batch
    assert torch.equal(Cf, Cf1) and torch.equal(Cc, Cc1)
    with torch.no_grad():
        for s in range(0, n - Bm + 1, Bm):  # the reference's own loop over the full batches
            c, f = m(rd[s:s + Bm], cd[s:s + Bm], pd[s:s + Bm], K)
            assert torch.equal(Cf[s:s + Bm], f) and torch.equal(Cc[s:s + Bm], c), s
    # a sub-range that starts and ends off the batch grid: the same pixels
    _, Cf2 = m.render(rd, cd, pd, K, 333, 2222)
    assert torch.equal(Cf2, Cf[333:2222])
    if mode == "f32":  # and they are the reference's (oracle) pixels, batch by batch with that batch's ray 0
        for s in (1200, 2400):
            e = min(s + Bm, n)
            _, of = oracle.render(w, row[s:e], col[s:e], pb[s:e], K, 64, 128)
            assert max_rel(Cf[s:e], of) < TOL, s


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_render_one_ray_pieces(oracle, pkg, dev, mode):
    """a plan can hold a 1-ray piece (n % batch_ray == 1 behind a batch of another near / far; a fuse_rays limit one ray
    short of the end; an unaligned shard that starts on a batch's last ray).  The library needs B >= 2 (like the reference, nerf.py:208):
    render() launches such a piece with its ray repeated and crops -- the ray gets the bits the per-batch `forward` gives it."""
    n, Bm = 1001, 250
    row, col, pb, K, _ = oracle.lego_inputs(n, seed=31)
    pb = pb.clone()
    pb[1000:, 15], pb[1000:, 16] = 2.5, 5.5   # the tail batch (one ray) belongs to another picture
    w = oracle.make_weights(7, sharp=True)
    m = pkg.NeRFModel(64, 128, Bm)
    m.load_state_dict(w)
    m = m.to(dev)
    m.bf16_mlp = mode == "bf16"
    rd, cd, pd = row.to(dev), col.to(dev), pb.to(dev)
    assert pkg.nerf.fuse_plan(pb[::Bm, 15:17].float().tolist(), n, Bm)[-1] == (1000, 1001, 2.5, 5.5)
    Cc, Cf = m.render(rd, cd, pd, K)
    assert Cc.shape == (n, 3) and torch.isfinite(Cf).all()
    # the last ray as the first ray of a 2-ray batch of its own (ray 0 = itself: its own near / far)
    m2 = pkg.NeRFModel(64, 128, 2)
    m2.load_state_dict(w)
    m2 = m2.to(dev)
    m2.bf16_mlp = m.bf16_mlp
    with torch.no_grad():
        c2, f2 = m2(rd[[1000, 1000]], cd[[1000, 1000]], pd[[1000, 1000]], K)
    assert torch.equal(Cc[1000], c2[0]) and torch.equal(Cf[1000], f2[0])
    if mode == "f32":
        _, of = oracle.render(w, row[[1000, 1000]], col[[1000, 1000]], pb[[1000, 1000]], K, 64, 128)
        assert max_rel(Cf[1000], of[0]) < TOL
    # a fuse_rays limit one ray short of the end: same pixels as the unlimited plan
    pb_same = pb.clone()
    pb_same[:, 15], pb_same[:, 16] = 2.0, 6.0
    pds = pb_same.to(dev)
    full = m.render(rd, cd, pds, K)[1]
    assert torch.equal(m.render(rd, cd, pds, K, fuse_rays=1000)[1], full)
    # an unaligned shard that starts on the last ray of batch 1 while batch 2's ray 0 has another near / far
    pb3 = pb_same.clone()
    pb3[500:, 15], pb3[500:, 16] = 2.5, 5.5
    pd3 = pb3.to(dev)
    full3 = m.render(rd, cd, pd3, K)[1]
    lo, hi, C = pkg.parallel.render_rows_sharded(m, rd, cd, pd3, K, rank=0, world=1, align_to_batches=False)
    assert (lo, hi) == (0, n) and torch.equal(C, full3)
    assert torch.equal(m.render(rd, cd, pd3, K, 499, 900)[1], full3[499:900])


def test_sticky_resample_status_survives_later_forwards(oracle, pkg, dev):
    """nerf_hip_read_status_sticky (ABI 4): the per-forward status word is cleared by the next forward, the sticky one only by the caller
    -- a loop that looks at its logging points only still learns that some forward in between met nerf.py:251-253's condition."""
    B = 64
    row, col, pb, K, _ = oracle.lego_inputs(B, seed=9)
    w = oracle.make_weights(4, sharp=True)
    m = pkg.NeRFModel(64, 128, B)
    m.load_state_dict(w)
    m = m.to(dev)
    pb_bad = pb.clone()
    pb_bad[5, 16] = pb_bad[5, 15]  # ray 5: far = near -> all coarse weights vanish
    with torch.no_grad():
        m(row, col, pb, K)
        assert not m.resample_fault() and not m.resample_fault_since(clear=False)
        m(row, col, pb_bad, K)
        assert m.resample_fault() and m.resample_fault_since(clear=False)
        m(row, col, pb, K)  # a healthy forward clears the per-forward word, not the sticky one
        assert not m.resample_fault()
        assert m.resample_fault_since(clear=True)
        assert not m.resample_fault_since(clear=True)  # cleared by the caller
        m(row, col, pb, K)
        assert not m.resample_fault_since()
