"""Stage-level calls into libnerf_hip.so (one per row of the hot-path table), used by the parity tests."""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _k9(K_inv):
    return _abi.f32_array(K_inv.detach().to("cpu", torch.float32).reshape(-1).tolist())


def rays(row, col, poses_bound_f32, K_inv, Nc):
    """-> d_cam[B,3], d_wrd[B,3], t_coarse[B,Nc]"""
    B, dev = row.shape[0], row.device
    d_cam = torch.empty(B, 3, device=dev)
    d_wrd = torch.empty(B, 3, device=dev)
    t_c = torch.empty(B, Nc, device=dev)
    _abi.check(_abi.lib().nerf_hip_rays(row.data_ptr(), col.data_ptr(), poses_bound_f32.data_ptr(), _k9(K_inv), B, Nc,
                                        d_cam.data_ptr(), d_wrd.data_ptr(), t_c.data_ptr(), _stream(row)))
    return d_cam, d_wrd, t_c


def field(params, row, col, poses_bound_f32, K_inv, t, debug=False):
    """-> rgb[B,N,3], sigma[B,N] (+ pts[B,N,3], gamma_p[B,N,60] if debug)"""
    B, N = t.shape
    dev = t.device
    rgb = torch.empty(B, N, 3, device=dev)
    sigma = torch.empty(B, N, device=dev)
    pts = torch.empty(B, N, 3, device=dev) if debug else None
    gp = torch.empty(B, N, 60, device=dev) if debug else None
    n = _abi.ws_bytes(max(B, 2), max(N, 2), max(N, 2), 0)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    _abi.check(_abi.lib().nerf_hip_field(_abi.ptr_array(params), row.data_ptr(), col.data_ptr(), poses_bound_f32.data_ptr(),
                                         _k9(K_inv), t.contiguous().data_ptr(), B, N, rgb.data_ptr(), sigma.data_ptr(),
                                         pts.data_ptr() if debug else None, gp.data_ptr() if debug else None,
                                         ws.data_ptr(), ws.numel(), _stream(t)))
    return (rgb, sigma, pts, gp) if debug else (rgb, sigma)


def field_bf16(params, row, col, poses_bound_f32, K_inv, t):
    """field() with the bf16 MLP (NERF_HIP_BF16_MLP) -> rgb[B,N,3], sigma[B,N]"""
    B, N = t.shape
    dev = t.device
    rgb = torch.empty(B, N, 3, device=dev)
    sigma = torch.empty(B, N, device=dev)
    n = _abi.ws_bytes(max(B, 2), max(N, 2), max(N, 2), _abi.BF16_MLP)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    _abi.check(_abi.lib().nerf_hip_field_bf16(_abi.ptr_array(params), row.data_ptr(), col.data_ptr(), poses_bound_f32.data_ptr(),
                                              _k9(K_inv), t.contiguous().data_ptr(), B, N, rgb.data_ptr(), sigma.data_ptr(),
                                              ws.data_ptr(), ws.numel(), _stream(t)))
    return rgb, sigma


def coarse_composite(t_c, sigma_c, rgb_c, near, far, delta0, Nf):
    """-> w_c[B,Nc], C_coarse[B,3], t_f[B,Nf], status(int)"""
    B, Nc = t_c.shape
    dev = t_c.device
    nf = torch.stack((near, far), dim=1).contiguous().to(dev)
    w_c = torch.empty(B, Nc, device=dev)
    C_c = torch.empty(B, 3, device=dev)
    t_f = torch.empty(B, Nf, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    _abi.check(_abi.lib().nerf_hip_coarse_composite(t_c.contiguous().data_ptr(), sigma_c.contiguous().data_ptr(),
                                                    rgb_c.contiguous().data_ptr(), nf.data_ptr(), float(delta0), B, Nc, Nf,
                                                    w_c.data_ptr(), C_c.data_ptr(), t_f.data_ptr(), status.data_ptr(),
                                                    _stream(t_c)))
    return w_c, C_c, t_f, int(status.item())


def merge_composite(t_c, t_f, sigma_c, sigma_f, rgb_c, rgb_f, last=1e-4):
    """-> bundle[B,N,5], w[B,N], C_fine[B,3]"""
    B, Nc = t_c.shape
    Nf = t_f.shape[1]
    dev = t_c.device
    bundle = torch.empty(B, Nc + Nf, 5, device=dev)
    w = torch.empty(B, Nc + Nf, device=dev)
    C_f = torch.empty(B, 3, device=dev)
    _abi.check(_abi.lib().nerf_hip_merge_composite(t_c.contiguous().data_ptr(), t_f.contiguous().data_ptr(),
                                                   sigma_c.contiguous().data_ptr(), sigma_f.contiguous().data_ptr(),
                                                   rgb_c.contiguous().data_ptr(), rgb_f.contiguous().data_ptr(), B, Nc, Nf,
                                                   float(last), bundle.data_ptr(), w.data_ptr(), C_f.data_ptr(), _stream(t_c)))
    return bundle, w, C_f


def coarse_composite_backward(t_c, sigma_c, rgb_c, near, far, delta0, dC_c, dt_f):
    """-> dsig_c[B,Nc], drgb_c[B,Nc,3] (contributions through C_coarse and the resampled depths)"""
    B, Nc = t_c.shape
    Nf = dt_f.shape[1]
    dev = t_c.device
    nf = torch.stack((near, far), dim=1).contiguous().to(dev)
    dsig = torch.zeros(B, Nc, device=dev)
    drgb = torch.zeros(B, Nc, 3, device=dev)
    _abi.check(_abi.lib().nerf_hip_coarse_composite_backward(t_c.contiguous().data_ptr(), sigma_c.contiguous().data_ptr(),
                                                             rgb_c.contiguous().data_ptr(), nf.data_ptr(), float(delta0), B, Nc, Nf,
                                                             dC_c.contiguous().data_ptr(), dt_f.contiguous().data_ptr(),
                                                             dsig.data_ptr(), drgb.data_ptr(), _stream(t_c)))
    return dsig, drgb
