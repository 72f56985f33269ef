"""Experiment: M-step kernels at batch scale (U utterances of 320 frames side by side), time and effective bandwidth."""
import importlib, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = importlib.import_module("disentangled-vae_amd.mcem")
R, K = 10, 10
out = {}
for U in (1, 25, 76):
    n = 320 * U
    X2 = torch.rand(513, n, device="cuda") + 0.01
    Vs = torch.rand(R, 513, n, device="cuda") + 0.1
    W = torch.rand(U, 513, K, device="cuda") + 0.01; H = torch.rand(K, n, device="cuda") + 0.01
    g = torch.ones(n, device="cuda"); Vb = torch.rand(513, n, device="cuda") + 0.1
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device="cuda")
    seg_start, seg_count = i32([320 * u for u in range(U)]), i32([300] * U)
    tile_seg = i32([u for u in range(U) for _ in range(10)])
    for _ in range(2): dev.m_step_batch_(X2, Vs, W, H, g, Vb, seg_start, seg_count, tile_seg)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): dev.m_step_batch_(X2, Vs, W, H, g, Vb, seg_start, seg_count, tile_seg)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    out[f"U{U}"] = dict(us=round(us, 1), vs_MB=round(Vs.numel() * 4 / 1e6, 1), eff_GBs_4reads=round(4 * Vs.numel() * 4 / us / 1e3, 1))
print(json.dumps(out))
