# A/B of variant libraries on NeRFModel.train_step (one library call per step, no per-kernel events) -- development aid.
#   usage: LIBS="- plain sc1" BATCHES="512 400 4096" bash scripts/ab_train_step.sh
for B in ${BATCHES:-512}; do
  for L in ${LIBS:--}; do
    if [ "$L" = "-" ]; then lib=$PWD/nerf-tiny_amd/libnerf_hip.so; else lib=$PWD/nerf-tiny_amd/libnerf_hip_$L.so; fi
    for rep in 1 2; do
      echo -n "== $L B=$B : "
      NERF_HIP_LIB=$lib python - $B <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
import bench, nerf_tiny_amd as P
B = int(sys.argv[1]); dev = torch.device("cuda:0")
row, col, pb, K, Ct = bench.synth_inputs(seed=1000)
m = bench.synth_weights(0).to(dev); m.batch_ray = B; m.bf16_mlp = True
inp = (row[:B].to(dev), col[:B].to(dev), pb[:B].float().to(dev), Ct[:B].to(dev))
bucket = P.parallel.GradBucket(m.network.parameters()); m.grad_bucket = bucket
def step():
    m.train_step(inp[0], inp[1], inp[2], K, inp[3]); bucket.consume()
for _ in range(30): step()
torch.cuda.synchronize(); n = 300 if B < 2000 else 60
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n)
print(f"{best*1e3:.4f} ms  {B/best:,.0f} rays/s")
PY
    done
  done
done
