#!/usr/bin/env python3
"""Where a right-hand side of the forward kernel spends its cycles: shader-clock stamps of ONE wave (workgroup 777 of the 4 096 x 241
benchmark launch, its SIMD partner and the rest of the chip running as usual) at six points of every evaluation.

    tools/build_variant.sh fwdtrace -DHODE_FWD_TRACE=777          (here)
    HODE_LIB=<...>/hode/lab/libhode_fwdtrace.so python tools/fwd_trace.py      (GPU box)

Prints, per segment, the mean cycles over the evaluations of the steady state and the cycles its vector instructions would take
at 4.2 pipe cycles each with two waves sharing the pipe (2 x 4.2 per instruction of THIS wave): the excess is stall."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import hode  # noqa: E402

B = 4096
dev = torch.device("cuda")
x0, t, meal, tvns = (v.to(dev) for v in bench.synth_cohort(B, 1000))
nn, ode = bench.synth_weights(0).to(dev), bench.ODE_DEFAULT.to(dev)
tape = "--tape" in sys.argv
for _ in range(2):
    sol = hode.solve_fwd(x0, t, meal, tvns, None, ode, nn, 64, 4, want_tape=tape)
torch.cuda.synchronize()
lib = C.CDLL(hode.lib_path())
buf = np.zeros(4096 * 8, np.uint64)
cnt = C.c_uint(0)
rc = lib.hode_lab_fwd_trace(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size), C.byref(cnt))
assert rc == 0, rc
rec = buf.reshape(4096, 8).astype(np.int64)
rec = rec[rec[:, 0] > 0]
rec = rec[np.argsort(rec[:, 0])]
gap = np.diff(rec[:, 0])
start = int(np.argmax(gap)) + 1 if gap.size and gap.max() > 1e6 else 0         # the second launch starts after the longest pause
rec = rec[start:]
print(f"lib {os.path.basename(hode.lib_path())}  tape={tape}  evaluations of workgroup 777 in the ring: {len(rec)} of 1448 (second launch)")
rec = rec[8:8 + (len(rec) - 8) // 6 * 6]            # drop the two start-up evaluations and the first step: whole steps from here on
seg = {
    "state broadcast + mechanistic terms (entry -> mech)": rec[:, 6] - rec[:, 0],
    "first layer 9 -> 64 (mech -> h1)": rec[:, 1] - rec[:, 6],
    "hidden layer 1": rec[:, 2] - rec[:, 1],
    "hidden layer 2": rec[:, 3] - rec[:, 2],
    "hidden layer 3": rec[:, 4] - rec[:, 3],
    "output layer + sum (h4 -> return)": rec[:, 5] - rec[:, 4],
    "return -> next entry (stage algebra; every 6th: error norm, controller, output row, next interval)": np.append(rec[1:, 0] - rec[:-1, 5], 0),
}
tot = 0
for name, d in seg.items():
    d = d[:-1]
    if "next entry" in name:
        d = np.append(d, d[-1])
        ok = d < 20000                                   # (a slot that was overwritten leaves a hole in the sequence)
        per = np.array([d[i::6][ok[i::6]].mean() for i in range(6)])
        d = d[ok]
        print(f"  {name}:\n      by position in the step: {np.round(per).astype(int).tolist()}  mean {d.mean():.0f}")
    else:
        print(f"  {name}: mean {d.mean():7.0f}  min {d.min():6d}  p90 {np.percentile(d, 90):7.0f}")
    tot += d.mean()
print(f"  total per evaluation {tot:.0f} cycles  (x 1448 evaluations = {tot * 1448 / 1e6:.2f} M cycles per trajectory)")
# evaluation-to-evaluation time along the trajectory (deciles): is the wave slower at the start or the end of the launch?
ent = rec[:, 0]
d = np.diff(ent)
d = d[d < 20000]
dec = np.array_split(d, 10)
print("  entry-to-entry cycles by decile of the trajectory:", [int(x.mean()) for x in dec])
