"""Quick timing on the GPU box (development aid; bench.py is the contract).
usage: [BF16=1] [TRAIN=1] python scripts/quick_time.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import nerf_tiny_amd as P
from nerf_tiny_amd import _abi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
bench.B = B
row, col, pb, K, Ct = bench.synth_inputs(0)
m = bench.synth_weights(0).to(dev)
row, col, pb, Ct = row.to(dev), col.to(dev), pb.float().to(dev), Ct.to(dev)
m.bf16_mlp = os.environ.get('BF16') == '1'
m.split_mlp = os.environ.get('SPLIT') == '1'
m.split_train = os.environ.get('SPLIT') == '1'  # (training calls: the split-fp32 train step, opt-in)
train = os.environ.get('TRAIN') == '1'


def step():
    if train:
        for p in m.network.parameters():
            p.grad = None
        Cc, Cf = m(row, col, pb, K)
        m.ray_loss(Cc, Cf, Ct).backward()
    else:
        with torch.no_grad():
            m(row, col, pb, K)


for _ in range(3):
    step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
_abi.profile_begin(128)
step()
torch.cuda.synchronize()
print({k: round(v[0], 4) for k, v in _abi.profile_end().items()})
flop = 676282368 if train else 227131392
print(f"B={B} bf16={m.bf16_mlp} train={train}: {dt*1e3:.3f} ms/batch  {B/dt:,.0f} rays/s  ({B/dt*flop/1e12:.1f} TFLOP/s)")
