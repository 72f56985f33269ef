#!/bin/bash
# GPU box: M-step kernels at batch scale (25 x 300 frames, R 10, K 10): product vs load-only / no-load ablations of the frames kernel (hipEvent per launch)
cd $GRAFT_REPO_ROOT
cat > /tmp/ms.py <<'PY'
import sys, os, importlib, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
dev = importlib.import_module("disentangled-vae_amd.mcem")
U, n, R, K = 25, 320, 10, 10
N = U * n
g = torch.Generator(device="cuda"); g.manual_seed(0)
X2 = torch.rand(513, N, device="cuda", generator=g) + 0.1
Vs = torch.rand(R, 513, N, device="cuda", generator=g) + 0.1
W = torch.rand(U, 513, K, device="cuda", generator=g) + 0.1; H = torch.rand(K, N, device="cuda", generator=g) + 0.1
gg = torch.ones(N, device="cuda"); Vb = torch.rand(513, N, device="cuda", generator=g) + 0.1
i32 = lambda v: torch.tensor(v, dtype=torch.int32, device="cuda")
ss, sc, ts = i32([u * n for u in range(U)]), i32([300] * U), i32([u for u in range(U) for _ in range(n // 32)])
for _ in range(3): dev.m_step_batch_(X2, Vs, W.clone(), H.clone(), gg.clone(), Vb.clone(), ss, sc, ts)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
Wc, Hc, gc, Vc = W.clone(), H.clone(), gg.clone(), Vb.clone()
e[0].record()
for _ in range(20): dev.m_step_batch_(X2, Vs, Wc, Hc, gc, Vc, ss, sc, ts)
e[1].record(); torch.cuda.synchronize()
print(sys.argv[1], "m-step us", round(e[0].elapsed_time(e[1]) * 1e3 / 20, 1), flush=True)
PY
for r in 1 2; do for v in base mstep_fly5 mstep_fly10; do
  lib=$PWD/disentangled-vae_amd/build/variants/$v.so; [ "$v" = base ] && lib=$PWD/disentangled-vae_amd/libdvae_hip.so
  DVAE_LIB=$lib python /tmp/ms.py $v 2>/dev/null
done; done
