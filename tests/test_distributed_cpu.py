"""world_size-2 `gloo` test of the patient-sharded data-parallel step (hode/train.py) on CPU.

The HIP kernels cannot run here, so the per-shard compute is INJECTED: the oracle (checker) stands
in for the forward+adjoint of a shard.  What is under test is the N>1 logic itself: contiguous
sharding, the single all-reduce(sum) of the fp64 message [grads | ode grads | loss_sum | n], global-mean scaling,
and that every rank ends with bit-identical parameters equal to the single-process result.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import hode
from hode import train as T
from oracle import oracle as O

H, L = 32, 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    g = np.load(os.path.join(ROOT, "tests", "golden", "g4_batched_t_h32l2.npz"))
    w = np.load(os.path.join(ROOT, "tests", "golden", "g0_weights_h32_l2.npz"))
    rng = np.random.default_rng(0)
    x0 = np.concatenate([g["x0"], g["x0"] * 1.1, g["x0"][:2] * 0.9])           # 10 patients: uneven split 5/5, 4/3/3
    t = np.concatenate([g["t"], g["t"], g["t"][:2]])
    meal = np.concatenate([g["meal"], g["meal"], g["meal"][:2]])
    tv = np.concatenate([g["tvns"], g["tvns"], g["tvns"][:2]])
    obs = rng.standard_normal((10, t.shape[1], 6))
    return x0, t, meal, tv, obs, w["nn_flat"].astype(np.float32), w["ode"]


def _oracle_compute(lo, hi, n_glob):
    x0, t, meal, tv, obs, _, ode = _data()

    def compute(p):
        s = O.solve(x0[lo:hi], t[lo:hi], meal[lo:hi], tv[lo:hi], None, ode, p.numpy(), H, L, rtol=1e-8, atol=1e-10,
                    dtype=np.float64, want_tape=True)
        r = s.y - obs[lo:hi]
        _, gnn, gode = O.solve_bwd(s, 2.0 * r / n_glob)
        return torch.tensor([(r ** 2).sum()], dtype=torch.float64), torch.tensor(gnn, dtype=torch.float32), \
            torch.tensor(gode, dtype=torch.float32), float(r.size)
    return compute


def _cpu_adam(lr):
    def opt(state, g):        # clip_grad_norm_(5.0) + Adam, the semantics hode_adam_step_f32 implements on the GPU
        g = g.clone()
        tn = g.norm()
        g *= torch.clamp(5.0 / (tn + 1e-6), max=1.0)
        state.m.mul_(0.9).add_(g, alpha=0.1)
        state.v.mul_(0.999).addcmul_(g, g, value=0.001)
        bc1, bc2 = 1 - 0.9 ** state.step, 1 - 0.999 ** state.step
        state.p.sub_(lr / bc1 * state.m / (state.v.sqrt() / bc2 ** 0.5 + 1e-8))
    return opt


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x0, t, *_rest, nn0, _ = _data()
    lo, hi = T.shard_bounds(10, rank, world)
    n_glob = 10 * t.shape[1] * 6
    state = T.TrainState(torch.tensor(nn0))
    losses = [float(T.train_step(state, _oracle_compute(lo, hi, n_glob), optimizer=_cpu_adam(1e-2))) for _ in range(3)]
    q.put((rank, state.p.numpy().copy(), losses, (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_data_parallel_step_matches_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # shards tile the cohort without gaps or overlap
    assert [r[3] for r in res][0][0] == 0 and res[-1][3][1] == 10
    assert all(res[i][3][1] == res[i + 1][3][0] for i in range(world - 1))
    # every rank holds bit-identical parameters and saw the same global loss
    for r in res[1:]:
        assert np.array_equal(r[1], res[0][1]) and r[2] == res[0][2]
    # == the single-process step on the whole cohort (summation order differs: fp32 grads, tol 1e-6)
    _, t, *_r, nn0, _ = _data()
    state = T.TrainState(torch.tensor(nn0))
    single = [float(T.train_step(state, _oracle_compute(0, 10, 10 * t.shape[1] * 6), optimizer=_cpu_adam(1e-2)))
              for _ in range(3)]
    assert np.allclose(single, res[0][2], rtol=1e-6)
    assert np.max(np.abs(state.p.numpy() - res[0][1])) < 1e-6
    assert single[-1] < single[0]                      # and the loss goes down


def test_pack_unpack_and_shards():
    g = torch.arange(5, dtype=torch.float32)
    buf = T.pack(g, torch.ones(17), torch.tensor([3.5], dtype=torch.float64), 42)
    assert buf.numel() == 5 + 17 + 2 and buf.dtype == torch.float64
    a, b, c, d = T.unpack(buf, 5)
    assert torch.equal(a.float(), g) and float(b.sum()) == 17 and float(c) == 3.5 and float(d) == 42
    # the tail is not a gradient: BASELINE config 4's element count (8 x 8 192 x 241 x 6) + 1 and a loss sum with 40 significant
    # bits must survive the message and an 8-way sum exactly
    n4 = 8192 * 241 * 6
    msg = sum(T.pack(g, None, torch.tensor([1.0 + 2.0 ** -40], dtype=torch.float64), n4 + (r == 0)) for r in range(8))
    _, _, ls, n = T.unpack(msg, 5)
    assert float(n) == 8 * n4 + 1 and float(ls) == 8.0 + 8 * 2.0 ** -40
    for n, w in [(10, 3), (4096, 8), (65536, 8), (7, 8)]:
        b = [T.shard_bounds(n, r, w) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def _allreduce_worker(rank, world, port, q):
    from models.hybrid_ode_nn import _allreduce_sum
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gnn = torch.arange(6, dtype=torch.float32).reshape(2, 3) * (rank + 1)
    gode = torch.ones(17, dtype=torch.float32) * (rank + 1)
    ss = torch.tensor([1.0 + 2.0 ** -40 * (rank + 1)], dtype=torch.float64)          # needs fp64 to survive the sum
    _allreduce_sum([gnn, gode, ss], None)
    q.put((rank, gnn.numpy().copy(), gode.numpy().copy(), ss.numpy().copy(), str(gnn.dtype), str(ss.dtype)))
    dist.barrier()
    dist.destroy_process_group()


def test_elbo_gradient_allreduce_helper():
    """The ONE collective of a sharded ELBO step (models/hybrid_ode_nn.py:_allreduce_sum): several tensors, one fp64
    buffer, dtypes and shapes restored, identical result on every rank."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_allreduce_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, gnn, gode, ss, d1, d2 in res:
        assert np.array_equal(gnn, np.arange(6, dtype=np.float32).reshape(2, 3) * 6) and (gode == 6).all()
        assert ss[0] == 3.0 + 2.0 ** -40 * 6 and d1 == "torch.float32" and d2 == "torch.float64"
