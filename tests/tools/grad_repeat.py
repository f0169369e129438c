"""Debug tool (GPU box): the same train step (same weights, same 4096-ray batch) N times; every gradient must come out
bit-identical each time (every sum of the fp32 train step runs in a fixed order; there are no float atomics).
Usage:  python tests/tools/grad_repeat.py [repeats]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (synthetic cfg2 inputs / weights)
import nerf_tiny_amd as P  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda:0")
    row, col, pb, K, Ct = bench.synth_inputs(0)
    m = bench.synth_weights(0).to(dev)
    row, col, pb, K, Ct = row.to(dev), col.to(dev), pb.to(dev), K.to(dev), Ct.to(dev)
    ref, bad = None, 0
    for i in range(n):
        for p in m.parameters():
            p.grad = None
        Cc, Cf = m(row, col, pb, K)
        loss = m.ray_loss(Cc, Cf, Ct)
        loss.backward()
        g = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        lv = float(loss)
        if ref is None:
            ref, lref = g, lv
            continue
        for k in g:
            a, b = g[k], ref[k]
            d = float((a - b).abs().max())
            if d != 0.0 or lv != lref:
                bad += 1
                print(f"repeat {i}: {k} differs from repeat 0 by {d:.3e} (|g| {float(b.abs().max()):.3e}), loss {lv} vs {lref}")
    print(f"{n} repeats, {bad} differing tensors")


if __name__ == "__main__":
    main()
