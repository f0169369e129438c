"""GPU checks of the bf16-MLP variant (BASELINE.json cfg3 "bf16 MLP / fp32 composite", flag NERF_HIP_BF16_MLP).

The reference has no reduced precision, so there is no reference output to be identical to: the variant is SPECIFIED by
``oracle.mlp_bf16`` (bf16-rounded weights and layer inputs, fp32 accumulation / biases / activations) and checked
  (a) against that emulation -- differences come only from the fp32 summation order and 1-ulp encoding differences, which
      now and then flip a bf16 rounding of a hidden unit (one flip = 2^-8 relative on one of 256 inputs);
  (b) against the fp32 oracle, to state how far cfg3 is from the 1e-4 bar it cannot meet (SURVEY.md section 7 hard-6:
      3.8e-3 / 1.1e-2 measured for bf16 weights + activations).
Tolerances are written next to each assertion.
"""
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


def test_bf16_training_is_refused(oracle, pkg, dev):
    g = load_golden("small_16_32")
    row, col, pb, K, _ = golden_inputs(g)
    params, _ = _params_dev(oracle, int(g["seed"]), bool(g["sharp"]), dev)
    m = pkg.NeRFModel(int(g["Nc"]), int(g["Nf"]), row.shape[0]).to(dev)
    m.load_state_dict(params)
    m.bf16_mlp = True
    with pytest.raises(RuntimeError, match="BF16_MLP"):
        m(row.to(dev), col.to(dev), pb.to(dev), K)
