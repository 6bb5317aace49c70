#!/usr/bin/env python3
"""Forward solve with and without the stage tape at the benchmark size (4 096 x 241, fp32), then the adjoint that reads the tape:
HIP events.  HODE_LIB=<variant .so> times an experiment build (e.g. the nt-store tape of DESIGN 6.2)."""
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


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


f0 = timed(lambda: hode.solve_fwd(x0, t, meal, tvns, None, ode, nn, 64, 4))
f1 = timed(lambda: hode.solve_fwd(x0, t, meal, tvns, None, ode, nn, 64, 4, want_tape=True, tape=sol.tape))
b = timed(lambda: hode.solve_bwd(sol, gy))
# forward + adjoint back to back (what a training step does: the adjoint reads what the forward has just written)
fb = timed(lambda: hode.solve_bwd(hode.solve_fwd(x0, t, meal, tvns, None, ode, nn, 64, 4, want_tape=True, tape=sol.tape), gy))
print(f"B={B}: forward {f0:.3f} ms | forward + tape {f1:.3f} ms | adjoint {b:.3f} ms | forward + tape, then adjoint {fb:.3f} ms   lib {os.path.basename(hode.lib_path())}")
