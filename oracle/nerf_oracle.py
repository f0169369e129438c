"""CPU oracle for the NeRF-tiny volume-rendering hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch PyTorch-CPU/NumPy *restatement* of the algorithm in the
reference's ``nerf.py`` (``NeRFModel.forward -> render_rays`` and ``ray_loss``).  It is the
checker for the HIP product path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``nerf-tiny_amd/``
imports it, and the product path raises when the HIP library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` (run in the build container, where
``/root/reference`` is importable) checks this restatement bit-for-bit against the imported
reference and writes the golden fixtures in ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` re-checks the restatement against those fixtures everywhere.

Every function cites the reference lines it follows (paths relative to /root/reference).
fp32 throughout; the order of every rounding in ray generation / sample points / phases
is the reference's (SURVEY.md section 8a SPEC), because the fine pass is ill-conditioned in it.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

def host_fingerprint() -> str:
    """CPU model + torch/numpy versions + thread count.  GEMM-dependent golden values are bit-reproducible only on the
    host class that generated them (oneDNN/MKL pick kernels per CPU and thread count); everything upstream of the first
    GEMM (rays, depths, points, encodings) is bit-reproducible everywhere."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return f"{model} | torch {torch.__version__} | numpy {np.__version__} | threads {torch.get_num_threads()}"


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------

L_POINT = 10  # nerf.py:127
L_DIR = 4  # nerf.py:127
WIDTH = 256  # nerf.py:76
POINT_DIM = 3 * 2 * L_POINT  # 60
DIR_DIM = 3 * 2 * L_DIR  # 24
EPSILON = 1e-7  # nerf.py:234
LAST_DELTA = 1e-4  # nerf.py:286

#: state_dict keys and shapes of ``NeRFModel`` (nerf.py:85-99, measured in SURVEY 8b), in
#: ``network.parameters()`` order.
PARAM_SHAPES = OrderedDict(
    [(f"network.point_layer.{i}.0.weight", (WIDTH, POINT_DIM if i == 0 else (WIDTH + POINT_DIM if i == 4 else WIDTH)))
     if j == 0 else (f"network.point_layer.{i}.0.bias", (WIDTH,))
     for i in range(8) for j in range(2)]
    + [
        ("network.sigma_layer.0.weight", (1, WIDTH)),
        ("network.sigma_layer.0.bias", (1,)),
        ("network.point_info.weight", (WIDTH, WIDTH)),
        ("network.point_info.bias", (WIDTH,)),
        ("network.dir_info.0.weight", (WIDTH // 2, WIDTH + DIR_DIM)),
        ("network.dir_info.0.bias", (WIDTH // 2,)),
        ("network.color_layer.0.weight", (3, WIDTH // 2)),
        ("network.color_layer.0.bias", (3,)),
    ]
)


def make_weights(seed: int = 0, sharp: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic weights: per tensor U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the nn.Linear
    default range, nerf.py:85-99) from ``numpy.random.default_rng([seed, tensor_index])``.
    ``sharp`` scales the sigma layer x50 to mimic a trained, peaky density."""
    out = OrderedDict()
    for idx, (name, shape) in enumerate(PARAM_SHAPES.items()):
        fan_in = shape[1] if len(shape) == 2 else PARAM_SHAPES[name.replace("bias", "weight")][1]
        bound = 1.0 / math.sqrt(fan_in)
        rng = np.random.default_rng([seed, idx])
        w = rng.uniform(-bound, bound, size=shape).astype(np.float32)
        if sharp and "sigma_layer" in name:
            w = w * np.float32(50.0)
        out[name] = torch.from_numpy(w)
    return out


def frequencies() -> tuple[torch.Tensor, torch.Tensor]:
    """f_l = fp32(2**e_l) * fp32(pi), e = linspace(0, L, L)  (nerf.py:141-146; quirk Q3:
    the octaves are NOT integers).  Returns (f_point[10], f_dir[4]) as fp32 tensors."""
    e_p = torch.linspace(0, L_POINT, L_POINT)
    e_d = torch.linspace(0, L_DIR, L_DIR)
    return torch.exp2(e_p) * math.pi, torch.exp2(e_d) * math.pi


# --------------------------------------------------------------------------------------
# stages
# --------------------------------------------------------------------------------------

def poses_extract(poses_bound: torch.Tensor):
    """nerf.py:52-67 + nerf.py:338.  [B,17] (any float dtype) -> R[B,3,3], o[B,3], near[B], far[B] (fp32)."""
    pb = poses_bound.to(torch.float32)
    pose = pb[:, :15].reshape(-1, 3, 5)
    return pose[:, :, :3].contiguous(), pose[:, :, 3].contiguous(), pb[:, 15].contiguous(), pb[:, 16].contiguous()


def camera_dirs(row: torch.Tensor, col: torch.Tensor, K_inv: torch.Tensor) -> torch.Tensor:
    """nerf.py:186-197.  p = [x, y, 1] @ K_inv (K_inv already transposed, nerf.py:433), x <- row,
    y <- column (quirk Q2); d_cam = p / max(||p||, 1e-12).  Products and sums individually
    rounded in k order; for the reference's K_inv all of them are exact."""
    x = row.to(torch.float32)
    y = col.to(torch.float32)
    K = K_inv.to(torch.float32)
    p = torch.stack([(x * K[0, j] + y * K[1, j]) + K[2, j] for j in range(3)], dim=1)  # [B,3]
    return F.normalize(p, p=2.0, dim=1)


def world_dirs(R: torch.Tensor, d_cam: torch.Tensor) -> torch.Tensor:
    """nerf.py:211: dir_wrd = R @ d_cam (3x3 matvec, k order, no fma)."""
    return (R[:, :, 0] * d_cam[:, 0:1] + R[:, :, 1] * d_cam[:, 1:2]) + R[:, :, 2] * d_cam[:, 2:3]


def coarse_depths(near: torch.Tensor, far: torch.Tensor, n_coarse: int) -> torch.Tensor:
    """nerf.py:288: numpy.linspace(near, far, Nc) in fp32: t_i = near + i * ((far-near)/(Nc-1)),
    last sample overwritten with far."""
    step = (far - near) / np.float32(n_coarse - 1)
    i = torch.arange(n_coarse, dtype=torch.float32)
    t = i[None, :] * step[:, None] + near[:, None]
    t[:, -1] = far
    return t


def sample_points(R, o, d_cam, t):
    """nerf.py:200-216: v = d_cam * t; pts = ((R[:,0]*v0 + R[:,1]*v1) + R[:,2]*v2) + o.
    Every * and + rounds to fp32 separately (no fma) -- SURVEY 7 hard-1."""
    v = d_cam[:, None, :] * t[:, :, None]  # [B,N,3]
    Rb = R[:, None, :, :]  # [B,1,3,3]
    pts = (Rb[..., 0] * v[..., 0:1] + Rb[..., 1] * v[..., 1:2]) + Rb[..., 2] * v[..., 2:3]
    return pts + o[:, None, :]


def encode(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """nerf.py:135-167 + flatten at nerf.py:103-104: gamma[..., c*2L + 2l + s] = (sin, cos)[s](fp32(x_c * f_l))."""
    phase = x[..., :, None] * freqs  # [..., 3, L]
    g = torch.stack((torch.sin(phase), torch.cos(phase)), dim=-1)  # [..., 3, L, 2]
    return g.flatten(start_dim=-3)


def mlp(params, gp: torch.Tensor, gd: torch.Tensor, return_hidden: bool = False):
    """nerf.py:101-124.  gp [B,N,60], gd [B,N,24] -> rgb [B,N,3], sigma [B,N]."""
    W = lambda n: params[n]
    h = gp
    hidden = []
    for i in range(8):
        inp = torch.cat((h, gp), dim=-1) if i == 4 else h  # hidden first (nerf.py:109)
        h = torch.relu(F.linear(inp, W(f"network.point_layer.{i}.0.weight"), W(f"network.point_layer.{i}.0.bias")))
        hidden.append(h)
    sigma = torch.abs(F.linear(h, W("network.sigma_layer.0.weight"), W("network.sigma_layer.0.bias")))
    feat = F.linear(h, W("network.point_info.weight"), W("network.point_info.bias"))
    c = torch.relu(F.linear(torch.cat((gd, feat), dim=-1), W("network.dir_info.0.weight"), W("network.dir_info.0.bias")))
    rgb = torch.sigmoid(F.linear(c, W("network.color_layer.0.weight"), W("network.color_layer.0.bias")))
    if return_hidden:
        return rgb, sigma.squeeze(-1), hidden, feat, c
    return rgb, sigma.squeeze(-1)


def _bf16(x: torch.Tensor) -> torch.Tensor:
    """round-to-nearest-even to bfloat16, returned as fp32 (autograd: straight-through, like a cast)"""
    return x.to(torch.bfloat16).to(torch.float32)


def mlp_bf16(params, gp: torch.Tensor, gd: torch.Tensor, return_hidden: bool = False, _round_act=None):
    """NOT a restatement of the reference (which has no reduced precision): the arithmetic the build's bf16-MLP variant
    (BASELINE.json cfg3, flag NERF_HIP_BF16_MLP) is specified to perform, emulated in fp32 -- same graph as ``mlp``
    (nerf.py:101-124) with every linear layer's weights AND inputs rounded to bf16 (RNE), fp32 products/accumulation,
    fp32 biases and fp32 activations.  One algebraic step is taken BEFORE rounding: point_info has no activation
    (nerf.py:117), so point_info and the feature columns of dir_info are one linear map of h7; the variant forms
    W_fold = W_dir[:, 24:] @ W_pi and b_fold = W_dir[:, 24:] @ b_pi + b_dir in fp32 and rounds W_fold to bf16 ONCE (the
    intermediate ``feat`` is never rounded -- closer to the fp32 network than rounding both factors and feat).
    Used only to check that variant (tests/test_gpu_bf16.py)."""
    W = lambda n: _bf16(params[n])
    b = lambda n: params[n]
    ra = _round_act or _bf16  # rounding of the ACTIVATIONS (encodings, layer outputs); mlp_bf16_jittered perturbs them before it
    gp, gd = ra(gp), ra(gd)
    h = gp
    hidden = []
    for i in range(8):
        inp = torch.cat((h, gp), dim=-1) if i == 4 else h
        h = ra(torch.relu(F.linear(inp, W(f"network.point_layer.{i}.0.weight"), b(f"network.point_layer.{i}.0.bias"))))
        hidden.append(h)
    sigma = torch.abs(F.linear(h, W("network.sigma_layer.0.weight"), b("network.sigma_layer.0.bias")))
    Wd, Wp = params["network.dir_info.0.weight"], params["network.point_info.weight"]
    n_d = gd.shape[-1]
    W_fold = _bf16(Wd[:, n_d:] @ Wp)
    b_fold = Wd[:, n_d:] @ b("network.point_info.bias") + b("network.dir_info.0.bias")
    feat = F.linear(h, W("network.point_info.weight"), b("network.point_info.bias"))  # (not part of the folded graph; returned for inspection)
    c = ra(torch.relu(F.linear(gd, _bf16(Wd[:, :n_d])) + F.linear(h, W_fold) + b_fold))
    rgb = torch.sigmoid(F.linear(c, W("network.color_layer.0.weight"), b("network.color_layer.0.bias")))
    if return_hidden:
        return rgb, sigma.squeeze(-1), hidden, feat, c
    return rgb, sigma.squeeze(-1)


def mlp_bf16_jittered(seed: int, rel: float = 1e-6):
    """``mlp_bf16`` as ANOTHER correct evaluation of the same specification would compute it: every fp32 value that is about to be
    rounded to bf16 (the encodings, every layer's activated output) is first moved by a seeded relative `rel` -- the size of fp32
    summation-order and 1-ulp sin / cos differences (1.3e-6) -- so that now and then a bf16 rounding flips, which
    is exactly how the device kernels differ from the emulation.  Used to compute the emulation's OWN sensitivity
    band for the gradient and trajectory tests of the bf16 variant (tests/test_gpu_bf16.py); deterministic per (seed, call order)."""
    gen = torch.Generator().manual_seed(seed)

    def round_act(x):
        return _bf16(x * (1.0 + rel * torch.randn(x.shape, generator=gen)))

    def f(params, gp, gd, return_hidden=False):
        return mlp_bf16(params, gp, gd, return_hidden, _round_act=round_act)

    return f


def weights_from_sigma(delta: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
    """nerf.py:263-272: s = sigma*delta; T_i = exp(-sum_{j<=i} s_j) (inclusive, quirk Q4); w = T*(1-exp(-s))."""
    s = delta * sigma
    return torch.exp(-torch.cumsum(s, dim=1)) * (1 - torch.exp(-s))


def composite(w: torch.Tensor, rgb: torch.Tensor) -> torch.Tensor:
    """nerf.py:274-281: C = sum_i w_i rgb_i."""
    return torch.sum(rgb * w.unsqueeze(2), dim=1)


class ResampleIndexError(RuntimeError):
    """The reference prints a banner and calls exit(0) here (nerf.py:251-253, quirk Q7)."""


def resample(t_c: torch.Tensor, w_c: torch.Tensor, n_fine: int, check: bool = True):
    """nerf.py:225-261.  Inverse-CDF sampling of n_fine depths.  u is built on the host from
    detached lo/hi with numpy.linspace in fp32; slope uses delta of RAY 0 for all rays (Q6)."""
    cdf = torch.cumsum(w_c, dim=1).contiguous()
    hi = torch.max(cdf, dim=1)[0].detach()
    lo = torch.min(cdf, dim=1)[0].detach()
    delta0 = (t_c[0, 1] - t_c[0, 0]).detach()
    slope_inv = delta0 / (w_c[:, 1:] + EPSILON)
    step = (hi - lo) / np.float32(n_fine + 1)
    j = torch.arange(1, n_fine + 1, dtype=torch.float32)
    u = j[None, :] * step[:, None] + lo[:, None]
    k = torch.searchsorted(cdf.detach(), u) - 1
    bad = (k > n_fine - 1) | (k < 0)
    if check and bool(bad.any()):
        raise ResampleIndexError("resample index outside [0, Nf-1] (reference would exit(0), nerf.py:251-253)")
    slope = torch.cat((slope_inv, torch.zeros(t_c.shape[0], 1)), dim=1)
    t_f = torch.gather(t_c, 1, k) + (u - torch.gather(cdf, 1, k)) * torch.gather(slope, 1, k)
    return t_f, dict(cdf=cdf, u=u, k=k, bad=bad)


def render(params, row, col, poses_bound, K_inv, n_coarse=64, n_fine=128, last=LAST_DELTA, stages=None, check=True, mlp=None, corrected=False):
    """nerf.py:333-348 + 286-323.  Returns (C_coarse[B,3], C_fine[B,3]).  ``stages`` (a dict) receives
    every intermediate.  Needs B >= 2 like the reference (quirk Q7: B = 1 breaks .squeeze()).
    ``mlp``: the field network; default = the reference's fp32 ``mlp`` (``mlp_bf16`` for the cfg3 variant).
    ``corrected`` (NOT the reference; the flagged extra of SURVEY.md 8a "Q", parity unpinned): t_fine detached (vs nerf.py:259) and ONE
    stable sort by depth that carries rgb / sigma along (vs the five independent channel sorts of nerf.py:307-308)."""
    mlp = mlp or globals()["mlp"]
    f_p, f_d = frequencies()
    R, o, near, far = poses_extract(poses_bound)
    d_cam = camera_dirs(row, col, K_inv)
    d_wrd = world_dirs(R, d_cam)
    gd = encode(d_wrd, f_d)  # identical for all samples of a ray

    t_c = coarse_depths(near, far, n_coarse)
    pts_c = sample_points(R, o, d_cam, t_c)
    rgb_c, sig_c = mlp(params, encode(pts_c, f_p), gd[:, None, :].expand(-1, n_coarse, -1))
    delta_c = ((far - near) / n_coarse)[:, None].expand(-1, n_coarse)  # quirk Q5 (nerf.py:293)
    w_c = weights_from_sigma(delta_c, sig_c)

    t_f, rs = resample(t_c, w_c, n_fine, check=check)
    if corrected:
        t_f = t_f.detach()
    pts_f = sample_points(R, o, d_cam, t_f)
    rgb_f, sig_f = mlp(params, encode(pts_f, f_p), gd[:, None, :].expand(-1, n_fine, -1))

    # nerf.py:302-308: ONE sort over dim=1 of a [B,N,5] bundle => five independent channel sorts (Q1)
    bundle = torch.cat((torch.cat((t_c, t_f), 1).unsqueeze(2), torch.cat((rgb_c, rgb_f), 1),
                        torch.cat((sig_c, sig_f), 1).unsqueeze(2)), dim=2)
    if corrected:
        perm_t = torch.sort(bundle[:, :, 0], dim=1, stable=True)[1]
        perm = perm_t.unsqueeze(2).expand(-1, -1, 5)
        sb = torch.gather(bundle, 1, perm)
    else:
        sb, perm = torch.sort(bundle, dim=1)
    t_s, rgb_s, sig_s = sb[:, :, 0], sb[:, :, 1:4], sb[:, :, 4]
    delta = torch.cat((t_s[:, 1:] - t_s[:, :-1], torch.full((t_s.shape[0], 1), last)), dim=1)
    w = weights_from_sigma(delta, sig_s)
    C_c = composite(w_c, rgb_c)
    C_f = composite(w, rgb_s)
    if stages is not None:
        stages.update(R=R, o=o, near=near, far=far, d_cam=d_cam, d_wrd=d_wrd, gd=gd, t_c=t_c, pts_c=pts_c,
                      rgb_c=rgb_c, sig_c=sig_c, w_c=w_c, t_f=t_f, pts_f=pts_f, rgb_f=rgb_f, sig_f=sig_f,
                      t_s=t_s, rgb_s=rgb_s, sig_s=sig_s, perm=perm, w=w, **rs)
    return C_c, C_f


def ray_loss(C_c, C_f, C_true):
    """nerf.py:325-331: SUM (not mean) of squared errors of both outputs."""
    return torch.sum(torch.square(C_c - C_true)) + torch.sum(torch.square(C_f - C_true))


def loss_and_grads(params, row, col, poses_bound, K_inv, C_true, n_coarse=64, n_fine=128, corrected=False):
    """Forward + autograd backward of the restatement (nerf.py:470-473).  Returns
    (C_c, C_f, loss, grads: OrderedDict name -> tensor)."""
    p = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in params.items())
    C_c, C_f = render(p, row, col, poses_bound, K_inv, n_coarse, n_fine, corrected=corrected)
    loss = ray_loss(C_c, C_f, C_true)
    loss.backward()
    return C_c.detach(), C_f.detach(), loss.detach(), OrderedDict((k, v.grad) for k, v in p.items())


# --------------------------------------------------------------------------------------
# synthetic inputs (literals are synthetic, not from the reference)
# --------------------------------------------------------------------------------------

LEGO_POSE = np.array([[-0.99990219, 0.00419225, -0.01334572, -0.05379832],
                      [-0.01398868, -0.29965907, 0.95394367, 3.84547043],
                      [-4.66e-10, 0.95403719, 0.29968831, 1.20808232]], dtype=np.float64)
LEGO_ANGLE_X = 0.6911112070083618


def make_K_inv(height: float, width: float, focal: float) -> torch.Tensor:
    """nerf.py:433."""
    return torch.tensor([[1.0, 0.0, -0.5 * width], [0.0, -1.0, 0.5 * height], [0.0, 0.0, -focal]]).to(torch.float).transpose(0, 1)


def pose_row(c2w34: np.ndarray, height, width, focal, near, far) -> np.ndarray:
    """One [17] float64 row laid out as loader.py:33 (3x5 [R|o|hwf] flattened, then near, far)."""
    m = np.concatenate((c2w34, np.array([[height], [width], [focal]], dtype=np.float64)), axis=1).flatten()
    return np.concatenate((m, np.array([near, far], dtype=np.float64)))


def lego_inputs(B: int, seed: int = 0, H: int = 400, W: int = 400, crop: int | None = None):
    """cfg1 (crop=32: 32x32 centre crop, B = 1024) / cfg2 (random pixels, one pose, near/far 2/6)."""
    focal = 0.5 * W / np.tan(0.5 * LEGO_ANGLE_X)
    if crop is not None:
        r0, c0 = (W - crop) // 2, (H - crop) // 2
        rr, cc = np.meshgrid(np.arange(r0, r0 + crop), np.arange(c0, c0 + crop), indexing="ij")
        row, col = rr.reshape(-1), cc.reshape(-1)
        assert row.size == B
    else:
        rng = np.random.default_rng(seed)
        row = rng.integers(0, W, size=B)
        col = rng.integers(0, H, size=B)
    pb = np.tile(pose_row(LEGO_POSE, H, W, focal, 2.0, 6.0), (B, 1))
    rng1 = np.random.default_rng(seed + 1)
    C_true = rng1.uniform(0.0, 1.0, size=(B, 3)).astype(np.float32)
    return (torch.from_numpy(row.astype(np.int64)), torch.from_numpy(col.astype(np.int64)),
            torch.from_numpy(pb), make_K_inv(H, W, focal), torch.from_numpy(C_true))


def fern_inputs(B: int, seed: int = 0, H: int = 756, W: int = 1008, n_images: int = 4):
    """cfg4-like: forward-facing poses, per-image near/far (exercises quirk Q6), no NDC (Q11)."""
    rng = np.random.default_rng(seed)
    focal = 0.8 * W
    rows = []
    for _ in range(n_images):
        a = rng.normal(size=3) * 0.05
        Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
        Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
        c2w = np.concatenate((Rx @ Ry, rng.normal(size=(3, 1)) * 0.3), axis=1)
        near = 1.0 + rng.uniform(0, 0.5)
        far = near + 3.0 + rng.uniform(0, 4.0)
        rows.append(pose_row(c2w, H, W, focal, near, far))
    img = rng.integers(0, n_images, size=B)
    pb = np.stack(rows)[img]
    row = rng.integers(0, W, size=B)
    col = rng.integers(0, H, size=B)
    C_true = np.random.default_rng(seed + 1).uniform(0.0, 1.0, size=(B, 3)).astype(np.float32)
    return (torch.from_numpy(row.astype(np.int64)), torch.from_numpy(col.astype(np.int64)),
            torch.from_numpy(pb), make_K_inv(H, W, focal), torch.from_numpy(C_true))
