#!/usr/bin/env python3
"""Host-side cost of the drop-in class path at the reference's own batch sizes (train/train_hybrid.py: 32 windows x 61 points):
one optimisation step = zero_grad -> HybridODENN.loss (data + physics + L2) -> backward -> clip_grad_norm_ -> Adam.step, timed
wall-clock without any synchronisation inside the loop, next to the GPU time of the same steps (HIP events)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd"))
import bench  # noqa: E402
from models import HybridODENN  # noqa: E402

dev = torch.device("cuda")
for B in [int(v) for v in sys.argv[1:]] or [32, 256]:
    T = int(os.environ.get("HODE_T", "61"))
    x0, t, meal, tvns = (v.to(dev) for v in bench.synth_cohort(B, 1000))
    t, meal, tvns = t[:T].contiguous(), meal[:, :T].contiguous(), tvns[:, :T].contiguous()
    torch.manual_seed(0)
    m = HybridODENN(device=dev)
    with torch.no_grad():
        off, w = 0, bench.synth_weights(0)
        for p in m.nn_residual.parameters():
            p.copy_(w[off:off + p.numel()].reshape(p.shape)); off += p.numel()
        obs = m(x0, t, {"meal": meal, "tVNS": tvns}) + 0.1 * torch.randn(B, T, 6, device=dev)
    batch = {"initial_state": x0, "observations": obs, "time_points": t, "external_inputs": {"meal": meal, "tVNS": tvns}}
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def step():
        opt.zero_grad()
        loss = m.loss(batch, 1.0, 0.01)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
        opt.step()
        return loss
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        loss = step()
    e1.record()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"B={B} T={T}: {1e3 * wall / n:.2f} ms per step wall ({B * n / wall:.0f} windows/s); host issue time {1e3 * host / n:.2f} ms, "
          f"GPU time between events {e0.elapsed_time(e1) / n:.2f} ms; loss {float(loss):.4f}")
