import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, hode
from oracle import oracle as O
G = os.path.join(ROOT, "tests/golden/")
w = np.load(G + "g0_weights_h64_l4.npz"); nn = w["nn_flat"]; ode = w["ode"]
g = np.load(G + "g4_t61_rand.npz")
dt = torch.float64
dev = lambda a: None if a is None else torch.as_tensor(np.asarray(a), dtype=dt, device="cuda")
def run(tag, nnw, method, c, T=None, meal=g["meal"], tv=g["tvns"]):
    t = g["t"] if T is None else g["t"][:T]
    m_ = None if meal is None else meal[:2, :len(t)]; v_ = None if tv is None else tv[:2, :len(t)]
    x0 = g["x0"][:2]; c = c[:2, :len(t)]
    ref = O.solve(x0, t, m_, v_, None, ode, nnw, 64, 4, method=method, rtol=1e-8, atol=1e-10, dtype=np.float64, want_tape=True)
    rx, rnn, rode = O.solve_bwd(ref, c)
    s = hode.solve_fwd(dev(x0), dev(t), dev(m_), dev(v_), None, dev(ode), dev(nnw), 64, 4, method=method, rtol=1e-8, atol=1e-10, want_tape=True)
    gx0, gnn, gode = hode.solve_bwd(s, dev(c), want_gode=True)
    print(tag, "nsteps", s.nsteps.cpu().numpy(), ref.nsteps, "ydiff", np.abs(s.y.cpu().numpy()-ref.y).max())
    print("  gx0 kern", gx0.cpu().numpy()[0]); print("  gx0 orac", rx[0])
    print("  gnn relerr", np.linalg.norm(gnn.cpu().numpy()-rnn)/max(np.linalg.norm(rnn),1e-300), " gode kern", gode.cpu().numpy()[:4], "orac", rode[:4])
rng = np.random.default_rng(5)
c = rng.standard_normal((8, 61, 6))
clast = np.zeros_like(c); clast[:, -1] = c[:, -1]
z = np.zeros_like(nn)
run("full dp", nn, 0, c)
run("zeroNN dp", z, 0, c)
run("zeroNN rk4", z, 1, c)
run("zeroNN rk4 T=2 clast", z, 1, clast[:, -2:], T=2)
run("zeroNN dp T=2", z, 0, c, T=2)
run("zeroNN rk4 nomeal", z, 1, c, meal=None, tv=None)
run("full rk4 T=3", nn, 1, c, T=3)
print("---- tape dump")
t = g["t"][:3]
s = hode.solve_fwd(dev(g["x0"][:2]), dev(t), dev(g["meal"][:2,:3]), dev(g["tvns"][:2,:3]), None, dev(ode), dev(z), 64, 4, method=1, want_tape=True)
tp = s.tape.cpu().numpy()
ms = s.max_steps
ent = tp[:2*ms*8*8].view(np.float64).reshape(2, ms, 8)
print("x0", g["x0"][:2]); print("tape", ent); print("seg", tp[2*ms*8*8:].view(np.int32)); print("y", s.y.cpu().numpy())
for dtt in (torch.float32,):
    d32 = lambda a: None if a is None else torch.as_tensor(np.asarray(a), dtype=dtt, device="cuda")
    s = hode.solve_fwd(d32(g["x0"][:2]), d32(t), d32(g["meal"][:2,:3]), d32(g["tvns"][:2,:3]), None, d32(ode), d32(z), 64, 4, method=0, want_tape=True)
    tp = s.tape.cpu().numpy(); ms = s.max_steps
    print("f32 dp tape", tp[:2*ms*8*4].view(np.float32).reshape(2, ms, 8)[:, :4], "nsteps", s.nsteps)
