#!/usr/bin/env python3
"""Timeline of the wave-specialised adjoint (lab library, HODE_WS_DBG bit 1024): shader-clock stamps of every wave of workgroup
(0, 0) over 32 iterations.  Points of a propagation wave: 0 loop top | 1 step headers done | 2 record DMA landed | 3 mechanistic part,
delta_NL | 4, 5, 6 after each transposed matrix | 8 at the barrier | 9 behind it.  Accumulation wave: 0 top | 8 at the barrier | 9 behind.
    HODE_LIB=<lab library> HODE_WS_DBG=1024 python tools/ws_trace.py [B]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import hode  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda")
x0, t, meal, tvns = (v.to(dev) for v in bench.synth_cohort(B, 1000))
nn, ode = bench.synth_weights(0).to(dev), bench.ODE_DEFAULT.to(dev)
sol = hode.solve_fwd(x0, t, meal, tvns, None, ode, nn, 64, 4, want_tape=True)
gy = torch.randn(B, 241, 6, device=dev, generator=torch.Generator(dev).manual_seed(3)) / (B * 241 * 6)
for _ in range(3):
    g = hode.solve_bwd(sol, gy)
torch.cuda.synchronize()
lib = hode.capi.load()
ITS, WAVES, PTS = 32, 16, 10
buf = np.zeros(ITS * WAVES * PTS, dtype=np.uint64)
lib.hode_lab_ws_trace.argtypes = [C.c_void_p, C.c_int]
rc = lib.hode_lab_ws_trace(buf.ctypes.data_as(C.c_void_p), buf.size)
assert rc == buf.size, rc
tr = buf.reshape(ITS, WAVES, PTS).astype(np.int64)
t0 = tr[:, :, 0].min(axis=1)                       # first wave out of the previous barrier
print("iteration length (first loop top -> next): ", np.diff(t0)[:31].tolist())
for it in range(6, 13):
    print(f"-- iteration {200 + it} (stage phase {(200 + it) % 6}); cycles relative to the first wave's loop top")
    for w in range(WAVES):
        r = tr[it, w] - t0[it]
        if w < 8:
            print(f"  P{w}: top {r[0]:5d} hdr {r[1]:5d} dma {r[2]:5d} mech {r[3]:5d} m1 {r[4]:5d} m2 {r[5]:5d} m3 {r[6]:5d} end {r[8]:5d} bar {r[9]:5d}")
        else:
            print(f"  A{w - 8}: top {r[0]:5d}{'':45s} end {r[8]:5d} bar {r[9]:5d}")
d = tr[4:30]
rel = d - d[:, :, :1].min(axis=1, keepdims=True)
print("mean over 26 iterations, P waves: ", {k: float(rel[:, :8, i].mean()) for k, i in
      dict(top=0, hdr=1, dma=2, mech=3, m1=4, m2=5, m3=6, end=8, bar=9).items()})
print("mean over 26 iterations, A waves (by wave): end ", rel[:, 8:, 8].mean(axis=0).round().tolist(), " bar ", float(rel[:, 8:, 9].mean()))
