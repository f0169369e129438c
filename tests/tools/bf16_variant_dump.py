"""Child process of tests/test_gpu_bf16.py: one bf16-MLP train step and one inference forward on fixed inputs, everything that the
library's launch-structure switches could change dumped to a file.  The switches are environment variables read once per process
(NERF_PREP_BF16: one-launch preparation vs separate fold / pack / rays launches; NERF_DW_BF16_MULTI: all weight-gradient products in one
launch vs a launch per product), so each variant needs a process of its own.

    [NERF_PREP_BF16=0] [NERF_DW_BF16_MULTI=0|1] python tests/tools/bf16_variant_dump.py OUT.pt [B]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import torch

import nerf_oracle as O
import nerf_tiny_amd as P
from nerf_tiny_amd import _abi


def main():
    out = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    Nc, Nf = 64, 128
    dev = torch.device("cuda:0")
    row, col, pb, K, Ct = O.fern_inputs(B, seed=13)
    w = O.make_weights(5, sharp=True)
    m = P.NeRFModel(Nc, Nf, B)
    m.load_state_dict(w)
    m = m.to(dev)
    m.bf16_mlp = True
    res = {}
    # training call: forward image (32x32x16 stream + bias block), transposed image, ray records, gradients
    Cc, Cf = m(row, col, pb, K)
    ws = m.last_workspace
    flags = _abi.SAVE_FOR_BACKWARD | _abi.BF16_MLP

    def region(name, nbytes, f=flags, w_=None):
        off = C.c_size_t(0)
        _abi.check(_abi.lib().nerf_hip_ws_offset(B, Nc, Nf, f, name.encode(), C.byref(off)))
        return (w_ if w_ is not None else ws)[off.value: off.value + nbytes].clone().cpu()

    img_bytes = 16384 + 1056 * 1024
    res["train_packed_bf"] = region("packed_bf", img_bytes)
    res["rayf"] = region("rayf", B * 24 * 4)
    res["t_c"] = region("t_c", B * Nc * 4)
    torch.cuda.synchronize()
    for name, n in (("t_f", B * Nf * 4), ("w_c", B * Nc * 4), ("bundle", B * (Nc + Nf) * 5 * 4), ("perm", B * 5 * (Nc + Nf) * 2)):
        res["train_" + name] = region(name, n)  # what the forward's per-ray stages left in the training workspace
    loss = m.ray_loss(Cc, Cf, Ct.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    res["Cc"], res["Cf"], res["loss"] = Cc.detach().cpu(), Cf.detach().cpu(), float(loss.detach())
    res["grads"] = [p.grad.detach().cpu() for p in m.network.parameters()]
    # inference call: the 16x16x32 image
    with torch.no_grad():
        Ic, If = m(row, col, pb, K)
    wsi = m.last_workspace
    res["infer_packed_bf"] = region("packed_bf", img_bytes, _abi.BF16_MLP, wsi)
    res["Ic"], res["If"] = Ic.cpu(), If.cpu()
    for name, n in (("rayf", B * 24), ("t_c", B * Nc), ("sig_c", B * Nc), ("rgb_c", B * Nc * 3), ("w_c", B * Nc), ("t_f", B * Nf), ("sig_f", B * Nf),
                    ("rgb_f", B * Nf * 3)):
        res["infer_" + name] = region(name, n * 4, _abi.BF16_MLP, wsi)  # the workspace's per-sample buffers of the inference call
    # a shard that does not start at the batch's ray 0 (quirk Q6: the global ray 0's near / far handed in) and the status word
    m.ray0_near_far = (float(pb[0, 15]) * 0.9, float(pb[0, 16]) * 1.1)
    pb_bad = pb.clone()
    pb_bad[B // 2, 16] = pb_bad[B // 2, 15]  # one ray with far = near: the reference's exit(0) condition
    with torch.no_grad():
        Sc, Sf = m(row, col, pb_bad, K)
    res["Sc"], res["Sf"], res["S_fault"] = Sc.cpu(), Sf.cpu(), bool(m.resample_fault())
    m.ray0_near_far = None
    with torch.no_grad():  # a healthy call on the same workspace right behind the faulty one: its status must be clean
        m(row, col, pb, K)
    res["S_fault_after_healthy"] = bool(m.resample_fault())
    # frozen rendering loop: the second call reuses the image and only makes the ray records
    with torch.no_grad(), m.frozen_weights():
        m(row, col, pb, K)
        Fc, Ff = m(row, col, pb, K)
    res["Fc"], res["Ff"] = Fc.cpu(), Ff.cpu()
    st = C.c_uint32(0)
    _abi.check(_abi.lib().nerf_hip_read_status_sticky(ws.data_ptr(), ws.numel(), C.byref(st), 0, torch.cuda.current_stream(dev).cuda_stream))
    res["sticky"] = int(st.value)
    res["env"] = {k: os.environ.get(k) for k in ("NERF_PREP_BF16", "NERF_DW_BF16_MULTI", "NERF_PAIR_BF16", "NERF_BF16_4WAVE", "NERF_FUSE_RAYS")}
    torch.save(res, out)
    print("DUMP-OK", res["env"], flush=True)


if __name__ == "__main__":
    main()
