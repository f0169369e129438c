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


def test_split_is_inference_only_and_training_ignores_it(oracle, pkg, dev):
    """a forward that records a graph runs the exact-fp32 training kernels whatever the flag says (bit-identical loss and gradients);
    the C ABI refuses the combination outright"""
    from nerf_tiny_amd import _abi

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
    with pytest.raises(_abi.NerfHipError):
        _abi.check(-1 if _abi.ws_bytes(8, 16, 32, _abi.SPLIT_MLP) <= 0 else _call_forward_with_save_and_split(pkg, m, row, col, pb, K, dev))


def _call_forward_with_save_and_split(pkg, m, row, col, pb, K, dev):
    from nerf_tiny_amd import _abi

    B, Nc, Nf = row.shape[0], m.num_coarse, m.num_fine
    flags = _abi.SAVE_FOR_BACKWARD | _abi.SPLIT_MLP
    ws = torch.empty(_abi.ws_bytes(B, Nc, Nf, flags), dtype=torch.uint8, device=dev)
    ps = list(m.network.parameters())
    Cc, Cf = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
    K9 = _abi.f32_array(K.reshape(-1).tolist())
    return _abi.lib().nerf_hip_forward(_abi.ptr_array(ps), row.to(dev).data_ptr(), col.to(dev).data_ptr(), pb.float().to(dev).data_ptr(), K9, None,
                                       B, Nc, Nf, 1e-4, Cc.data_ptr(), Cf.data_ptr(), ws.data_ptr(), ws.numel(), flags,
                                       torch.cuda.current_stream(dev).cuda_stream)
