set -e
cd $GRAFT_REPO_ROOT
NERF_PAIR_BF16=1 python tests/tools/bf16_variant_dump.py gpurun_out/r04e/p1.pt 8 > /dev/null 2>&1
NERF_PAIR_BF16=0 python tests/tools/bf16_variant_dump.py gpurun_out/r04e/p0.pt 8 > /dev/null 2>&1
python - <<'PY'
import torch
a=torch.load('gpurun_out/r04e/p1.pt',weights_only=False); b=torch.load('gpurun_out/r04e/p0.pt',weights_only=False)
B=8
for k,shape in (("infer_sig_c",(B,64)),("infer_rgb_c",(B,64,3)),("infer_w_c",(B,64)),("infer_t_f",(B,128)),("infer_sig_f",(B,128)),("infer_rgb_f",(B,128,3))):
    x=a[k].view(torch.float32).view(*shape); y=b[k].view(torch.float32).view(*shape)
    bad=(x!=y)
    print(k,'mismatch',int(bad.sum()),'of',bad.numel())
    if bad.any():
        d=bad.reshape(B,-1).float()
        print('  per ray:',d.sum(1).tolist())
        r=int(d.sum(1).argmax())
        idx=bad[r].reshape(shape[1],-1).any(1).nonzero().flatten().tolist()
        print('  ray',r,'bad sample idx:',idx[:64])
        print('  pair:',x[r].flatten()[:12].tolist()); print('  sep :',y[r].flatten()[:12].tolist())
print('Ic',a['Ic'][:4].tolist()); print('Ic sep',b['Ic'][:4].tolist())
PY
