"""Diagnostic: cycles per phase of the fp32 TRAINING kernels and per-workgroup times of the weight-gradient kernel
(needs `make -C nerf-tiny_amd/csrc stamps`; run with NERF_HIP_LIB=nerf-tiny_amd/libnerf_hip_stamps.so).  A stamped build
is slower: read shares and spreads, not absolute times."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import nerf_tiny_amd as P
from nerf_tiny_amd import _abi
dev = torch.device("cuda:0")
row, col, pb, K, Ct = bench.synth_inputs(1000)
m = bench.synth_weights(0).to(dev)
row, col, pb, Ct = row.to(dev), col.to(dev), pb.float().to(dev), Ct.to(dev)
F = _abi.SAVE_FOR_BACKWARD
def step():
    for p in m.network.parameters(): p.grad = None
    Cc, Cf = m(row, col, pb, K)
    m.ray_loss(Cc, Cf, Ct).backward()
step(); step()
torch.cuda.synchronize()
dbg = _abi.ws_view(m.last_workspace, bench.B, bench.NC, bench.NF, F, "dbg", (16384,), torch.int64)
dbg.zero_()
step()
torch.cuda.synchronize()
v = dbg.cpu().tolist()
def show(title, vals, n, names, mf):
    tot = sum(vals)
    print(f"{title}: tiles {n}, cycles/tile {tot / n:.0f}")
    for nm, x, k in zip(names, vals, mf):
        print(f"  {nm:34s} {x / n:9.0f} cycles {100 * x / tot:5.1f} %   MFMA issue {k * 64:7d}  ratio {x / n / max(k * 64, 1):.3f}")
show("forward with saves (both passes)", v[0:8], v[31], ["prologue", "layer 0", "layers 1-3", "layer 4", "layers 5-7", "sigma head", "point_info + dir_info (folded)", "colour head"],
     [0, 264, 3096, 1288, 3096, 0, 512, 0])
show("backward chain, coarse", v[32:39], v[63], ["prologue (colour head)", "dir_info + point_info (folded) + sigma", "-", "layers 7-5", "layer 4", "layers 3-1", "epilogue"],
     [0, 520, 0, 3072, 1024, 3072, 0])
show("backward chain, fine", v[40:47], v[62], ["prologue (colour head)", "dir_info + point_info (folded) + sigma", "-", "layers 7-5", "layer 4 (+skip)", "layers 3-1", "layer 0 + d t"],
     [0, 520, 0, 3072, 1280, 3072, 128])
# k_dw4 of layer 1 (a 256 x 256 product): per-wave records
rec = torch.tensor(v[64:64 + 256 * 8 * 4]).view(256, 8, 4)[:, :4]  # k_dw4: four waves per workgroup (record slots 4..7 unused)
t0 = rec[:, :, 0].min()
start = (rec[:, :, 0] - t0).float() / 100.0  # us
end = (rec[:, :, 1] - t0).float() / 100.0
wg_end = end.max(1).values; wg_start = start.min(1).values
print(f"k_dw4 (layer 1, 256 x 256): workgroup start spread {float(wg_start.min()):.1f}..{float(wg_start.max()):.1f} us, end {float(wg_end.min()):.1f}..{float(wg_end.max()):.1f} us, "
      f"mean end {float(wg_end.mean()):.1f} us")
xcc = rec[:, 0, 2] & 0xF
by = collections.defaultdict(list)
for i in range(256): by[int(xcc[i])].append(float(wg_end[i] - wg_start[i]))
for x in sorted(by): print(f"  XCC {x}: {len(by[x])} workgroups, duration min {min(by[x]):.1f} mean {sum(by[x]) / len(by[x]):.1f} max {max(by[x]):.1f} us")
dur = (end - start)
print(f"  wave duration min {float(dur.min()):.1f} mean {float(dur.mean()):.1f} max {float(dur.max()):.1f} us")
