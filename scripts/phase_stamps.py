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
flags = _abi.FORCE_TILE_KERNEL if m.force_tile_kernel else 0
with torch.no_grad():
    m(row, col, pb, K)
    # the stamps live in the workspace's 'dbg' area (words 0..31 = forward: phase sums 0..7, tile count at 31), as in train_stamps.py
    dbg = _abi.ws_view(m.last_workspace, bench.B, bench.NC, bench.NF, flags, "dbg", (32,), torch.int64)
    dbg.zero_()
    m(row, col, pb, K)
    torch.cuda.synchronize()
    v = dbg.cpu().tolist()
tile = os.environ.get("NERF_STAMP_TILE") == "1"  # set together with model.force_tile_kernel
if tile:
    names = ["encode/prologue", "mfma loops", "barrier after mfma", "acc_store", "barrier after store", "heads+rest"]
    ideal = [0, 4608 * 64, 0, 0, 0, 0]
else:
    names = ["prologue (loads, point, encode)", "layer 0", "layers 1-3", "layer 4", "layers 5-7", "sigma head", "point_info + dir_info (folded)", "colour head + stores"]
    ideal = [0, 264 * 64, 3096 * 64, 1288 * 64, 3096 * 64, 0, 512 * 64, 0]
n = v[7] if tile else v[31]; tot = sum(v[:len(names)])  # (the LDS-tile kernel keeps its count at word 7)
if n == 0:
    raise SystemExit("no stamps recorded: run with NERF_HIP_LIB=nerf-tiny_amd/libnerf_hip_stamps.so (make -C nerf-tiny_amd/csrc stamps)")
print("tiles", n, "cycles/tile", tot / n)
for nm, x, idl in zip(names, v, ideal):
    extra = f"  (MFMA issue time {idl})" if idl else ""
    print(f"  {nm:32s} {x / n:10.0f} cycles  {100 * x / tot:5.1f} %{extra}")
