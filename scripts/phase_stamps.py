"""Diagnostic: where a k_field_fwd workgroup spends its cycles (needs `make -C nerf-tiny_amd/csrc stamps`;
run with NERF_HIP_LIB=nerf-tiny_amd/libnerf_hip_stamps.so).  Shares only -- a stamped build is slower."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import nerf_tiny_amd as P
from nerf_tiny_amd import _abi
dev = torch.device("cuda:0")
row, col, pb, K, Ct = bench.synth_inputs(1000)
m = bench.synth_weights(0).to(dev)
m.force_tile_kernel = os.environ.get("NERF_STAMP_TILE") == "1"
row, col, pb = row.to(dev), col.to(dev), pb.float().to(dev)
with torch.no_grad():
    m(row, col, pb, K)
    ws = m.last_workspace
    st = ws[:256].view(torch.int64)
    st[8:16] = 0
    m(row, col, pb, K)
    torch.cuda.synchronize()
    v = st[8:16].cpu().tolist()
tile = os.environ.get("NERF_STAMP_TILE") == "1"  # set together with model.force_tile_kernel
if tile:
    names = ["encode/prologue", "mfma loops", "barrier after mfma", "acc_store", "barrier after store", "heads+rest"]
    ideal = [0, 4608 * 64, 0, 0, 0, 0]
else:
    names = ["prologue (loads, point, encode)", "layers 0..7", "sigma head", "point_info + dir_info", "colour head + stores"]
    ideal = [0, 8320 * 64, 0, 1544 * 64, 0]
n = v[7]; tot = sum(v[:len(names)])
print("tiles", n, "cycles/tile", tot / n)
for nm, x, idl in zip(names, v, ideal):
    extra = f"  (MFMA issue time {idl})" if idl else ""
    print(f"  {nm:32s} {x / n:10.0f} cycles  {100 * x / tot:5.1f} %{extra}")
