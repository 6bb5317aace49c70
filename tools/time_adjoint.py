#!/usr/bin/env python3
"""Adjoint timing at the benchmark size (4 096 x 241, fp32): forward with tape once, then the adjoint N times (HIP events).
    python tools/time_adjoint.py [B]            HODE_LIB=<lab library> HODE_BWD=fused for the one-role kernel"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import hode  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda")
x0, t, meal, tvns = (v.to(dev) for v in bench.synth_cohort(B, 1000))
nn, ode = bench.synth_weights(0).to(dev), bench.ODE_DEFAULT.to(dev)
sol = hode.solve_fwd(x0, t, meal, tvns, None, ode, nn, 64, 4, want_tape=True)
gy = torch.randn(B, 241, 6, device=dev, generator=torch.Generator(dev).manual_seed(3)) / (B * 241 * 6)
for want_gode in (False, True):
    g = hode.solve_bwd(sol, gy, want_gode=want_gode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g = hode.solve_bwd(sol, gy, want_gode=want_gode)
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B} adjoint{' + ODE-constant grads' if want_gode else ''}: {e0.elapsed_time(e1) / 5:.3f} ms   |gnn| {float(g[1].norm()):.6e}  lib {os.path.basename(hode.lib_path())} "
          f"HODE_BWD={os.environ.get('HODE_BWD', '')}")
