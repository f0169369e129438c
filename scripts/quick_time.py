"""Quick forward timing on the GPU box (development aid; bench.py is the contract)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import nerf_tiny_amd as P

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
bench.B = B
row, col, pb, K, Ct = bench.synth_inputs(0)
m = bench.synth_weights(0).to(dev)
row, col, pb = row.to(dev), col.to(dev), pb.float().to(dev)
m.bf16_mlp = os.environ.get('BF16') == '1'
with torch.no_grad():
    for _ in range(3): m(row, col, pb, K)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n): m(row, col, pb, K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
from nerf_tiny_amd import _abi
_abi.profile_begin(64)
with torch.no_grad():
    m(row, col, pb, K)
torch.cuda.synchronize()
print({k: round(v[0], 4) for k, v in _abi.profile_end().items()})
print(f"B={B}: {dt*1e3:.3f} ms/batch  {B/dt:,.0f} rays/s  ({B/dt*227131392/1e12:.1f} TFLOP/s of 157.3)")
