"""CPU: the oracle restatement against the golden vectors captured from the imported reference
(tests/golden/make_golden.py).  Forward outputs and stage intermediates must be BIT-identical;
gradients (multi-threaded accumulation order) within 1e-5 L2-rel."""
import numpy as np
import pytest
import torch

from conftest import golden_inputs, l2_rel, load_golden

CASES = ["cfg1_lego_crop32", "cfg1_lego_crop32_sharp", "cfg4_fern_rand512", "small_16_32"]


@pytest.mark.parametrize("name", CASES + ["cfg2_lego_rand4096"])
def test_forward_bit_identical(oracle, name):
    g = load_golden(name)
    row, col, pb, K, _ = golden_inputs(g)
    params = oracle.make_weights(int(g["seed"]), bool(g["sharp"]))
    st = {}
    with torch.no_grad():
        Cc, Cf = oracle.render(params, row, col, pb, K, int(g["Nc"]), int(g["Nf"]), stages=st)
    n = g["st_t_c"].shape[0]
    # upstream of the first GEMM: bit-identical on every host
    for k in ("t_c", "pts_c", "d_wrd", "gd"):
        assert np.array_equal(st[k][:n].numpy(), g["st_" + k]), k
    gp = oracle.encode(st["pts_c"][:n], oracle.frequencies()[0])
    same_host = str(g["meta_host"]) == oracle.host_fingerprint()
    # sin/cos (SLEEF picks a code path per ISA) and everything downstream of a GEMM (BLAS kernel choice) are
    # bit-identical on the generating host class only; elsewhere: ulps on the encodings, 1e-4 on the outputs, and
    # 2e-3 on fine-pass intermediates (which inherit the conditioning of the 3217 rad/unit encoding).
    if same_host:
        assert np.array_equal(gp.numpy(), g["st_gp_c"])
    else:
        assert float(np.abs(gp.numpy() - g["st_gp_c"]).max()) < 2e-6
    for got, want, k in [(Cc, g["C_coarse"], "C_coarse"), (Cf, g["C_fine"], "C_fine")] + \
            [(st[k][:n], g["st_" + k], k) for k in ("sig_c", "rgb_c", "w_c", "t_f", "pts_f", "sig_f", "rgb_f")]:
        if same_host:
            assert np.array_equal(got.numpy(), want), k
        else:
            tol = 1e-4 if k.startswith("C_") else 2e-3
            assert float(np.abs(got.numpy() - want).max()) <= tol * float(np.abs(want).max()), k


@pytest.mark.parametrize("name", ["cfg1_lego_crop32", "small_16_32"])
def test_gradients(oracle, name):
    g = load_golden(name)
    row, col, pb, K, Ct = golden_inputs(g)
    params = oracle.make_weights(int(g["seed"]), bool(g["sharp"]))
    _, _, loss, grads = oracle.loss_and_grads(params, row, col, pb, K, Ct, int(g["Nc"]), int(g["Nf"]))
    same_host = str(g["meta_host"]) == oracle.host_fingerprint()
    # On another host class the forward differs in the last bits and the reference's gradient is discontinuous in them
    # (per-channel sort, ReLU kinks on the t_fine path: DESIGN.md section 6), so only a band can be asserted there.
    tol = 1e-5 if same_host else 0.3
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    for k, v in grads.items():
        assert abs(float(v.double().norm()) - float(g["gnorm_" + k])) <= tol * float(g["gnorm_" + k]), k
        if "grad_" + k in g:
            assert l2_rel(v, g["grad_" + k]) < tol, k
        else:
            assert l2_rel(v.flatten()[::97], g["gslice_" + k]) < tol, k


@pytest.mark.parametrize("name", ["dec_cfg1_lego_crop32_r32", "dec_cfg4_fern_r32"])
def test_oracle_reproduces_the_references_decisions(oracle, name):
    """tests/golden/make_decisions_golden.py: the reference's torch.sort indices (nerf.py:308), index_fine (nerf.py:248) and ReLU sign bits
    (nerf.py:107-119), recorded inside the run that produced the stored gradients.  On the generating host class the oracle takes the same
    decisions bit for bit and lands on the same gradients to 1e-5; elsewhere (another BLAS) nearly all decisions and a band."""
    g = load_golden(name)
    row, col, pb, K, Ct = golden_inputs(g)
    Nc, Nf = int(g["Nc"]), int(g["Nf"])
    w = oracle.make_weights(int(g["seed"]), bool(g["sharp"]))
    p = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    st = {}
    Cc, Cf = oracle.render(p, row, col, pb, K, Nc, Nf, stages=st)
    loss = oracle.ray_loss(Cc, Cf, Ct)
    loss.backward()
    same_host = str(g["meta_host"]) == oracle.host_fingerprint()
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    perm, k = st["perm"].numpy(), st["k"].numpy()
    f_p, _ = oracle.frequencies()
    with torch.no_grad():
        _, _, hid, _, c = oracle.mlp(w, oracle.encode(st["pts_f"].detach(), f_p), st["gd"][:, None, :].expand(-1, Nf, -1), return_hidden=True)
    relu_f = np.stack([np.packbits((h > 0).numpy().astype(np.uint8), axis=-1, bitorder="little") for h in hid])
    if same_host:
        assert np.array_equal(perm, g["sort_index"]) and np.array_equal(k, g["index_fine"]) and np.array_equal(relu_f, g["relu_f"])
        for kk, v in p.items():
            assert l2_rel(v.grad, g["grad_" + kk]) < 1e-5, kk
    else:
        assert float((perm[:, :, 0] == g["sort_index"][:, :, 0]).mean()) > 0.999  # (the depth channel has no near-ties)
        assert float((k == g["index_fine"]).mean()) > 0.99
        assert float(np.unpackbits(relu_f ^ g["relu_f"]).mean()) < 1e-3
        for kk, v in p.items():
            assert l2_rel(v.grad, g["grad_" + kk]) < 0.3, kk


def test_corrected_mode_of_the_oracle(oracle):
    """the flagged extra (SURVEY.md 8a "Q"; parity unpinned -- the reference has no such mode): joint depth sort + detached t_fine.  With ONE
    permutation per ray the sorted sigma / rgb are the samples' own; the coarse colour is unchanged; with a loss on C_fine alone the gradient
    equals the default mode's whenever nothing flows through t_fine and the channel sorts happen to agree -- here only the structure is
    checked: same C_coarse, another C_fine, finite gradients, the permutation identical across channels."""
    row, col, pb, K, Ct = oracle.lego_inputs(48, seed=2)
    w = oracle.make_weights(1, sharp=True)
    st0, st1 = {}, {}
    with torch.no_grad():
        Cc0, Cf0 = oracle.render(w, row, col, pb, K, 64, 128, stages=st0)
        Cc1, Cf1 = oracle.render(w, row, col, pb, K, 64, 128, stages=st1, corrected=True)
    assert torch.equal(Cc0, Cc1) and not torch.equal(Cf0, Cf1)
    assert all(torch.equal(st1["perm"][:, :, c], st1["perm"][:, :, 0]) for c in range(5))
    assert torch.equal(st1["t_s"], st0["t_s"])  # the depth channel is sorted the same way in both modes
    assert not torch.equal(st1["sig_s"], st0["sig_s"])
    _, _, loss, grads = oracle.loss_and_grads(w, row, col, pb, K, Ct, 64, 128, corrected=True)
    assert torch.isfinite(loss) and all(torch.isfinite(v).all() for v in grads.values())


def test_frequencies_bits(oracle):
    """quirk Q3: non-integer octaves; bit patterns baked into csrc/common.h."""
    fp, fd = oracle.frequencies()
    want_p = "40490fdb 40d928ae 416a8b6c 41fd527b 4288cd33 4313c0fa 439f953c 442c5bef 44ba2881 45490fdb".split()
    want_d = "40490fdb 40fd527a 419f953c 42490fdb".split()
    assert [f"{x:08x}" for x in fp.numpy().view(np.uint32)] == want_p
    assert [f"{x:08x}" for x in fd.numpy().view(np.uint32)] == want_d


def test_resample_guard_fires_on_zero_density(oracle):
    """quirk Q7: all-equal coarse weights -> index condition of nerf.py:251 (the reference exit(0)s)."""
    row, col, pb, K, _ = oracle.lego_inputs(8, seed=0)
    params = oracle.make_weights(0)
    params["network.sigma_layer.0.weight"].zero_()
    params["network.sigma_layer.0.bias"].zero_()
    with pytest.raises(oracle.ResampleIndexError):
        with torch.no_grad():
            oracle.render(params, row, col, pb, K, 64, 128)


def test_param_table_matches_package(oracle, pkg):
    m = pkg.NeRFModel(64, 128, 8)
    sd = m.state_dict()
    assert list(sd.keys()) == list(oracle.PARAM_SHAPES.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(oracle.PARAM_SHAPES[k]), k
    assert sum(v.numel() for v in sd.values()) == 593924
    assert [tuple(p.shape) for p in m.network.parameters()] == [tuple(s) for s in oracle.PARAM_SHAPES.values()]
