"""bench.py as the driver launches it for N > 1 -- `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`
-- rehearsed on ONE GPU with the gloo backend (the RCCL path differs only by the backend name: the same all_reduce call on
a device tensor instead of a host-staged one).  Checks the JSON contract of the line rank 0 prints, the collective the
training leg reports, and that the 2-rank step IS the single-process step on the union of the two shards."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_gloo_on_one_gpu_matches_single_process_step():
    import bench
    import hode
    B = 256
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--patients-per-gpu", str(B), "--train-steps", "2"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["unit"] == "patient-trajectories/s" and out["dtype"] == "f32" and out["vs_baseline"] is None
    assert out["config"]["patients_per_gpu"] == B and "workload" in out["config"]
    assert abs(out["value"] - 2 * B * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]      # whole-job aggregate over both ranks
    assert out["roofline"]["bound"] == "valu_fp32" and 0 < out["roofline"]["frac"] < 1
    assert "cpu_baseline" not in out and "vi_step" not in out      # rank-0-at-N=1 legs only
    tr = out["train_step"]
    assert tr["collective"].startswith("1 x all_reduce(sum) of 13 529 fp64")
    # the same step in ONE process on the union of the two shards: first-step loss of the 2-rank run == global mean loss
    dev = torch.device("cuda")
    nn_t, ode = bench.synth_weights(0).to(dev), bench.ODE_DEFAULT.to(dev)
    tot, n = 0.0, 0
    for rank in range(2):
        x0, t, meal, tv = (v.to(dev) for v in bench.synth_cohort(B, 1000 + rank))
        obs, student = bench.train_problem(dev, x0, t, meal, tv, ode, nn_t, rank)
        ls, _, _, _ = hode.train.hip_loss_and_grads(student, ode, x0, t, meal, tv, obs, 64, 4, 2 * B * 241 * 6)
        tot += float(ls)
        n += B * 241 * 6
    assert abs(tr["loss_first_last"][0] - tot / n) < 1e-6 * (tot / n)
    assert np.isfinite(tr["value"]) and tr["value"] > 0


def test_bench_two_ranks_uneven_shards_is_the_single_process_step_on_the_whole_cohort():
    """`--total-patients 37` over two ranks = shards of 19 and 18 (hode.train.shard_bounds) through bench.py's own path: the
    global-mean scaling, the fp64 message and the whole-job value must not assume equal shards."""
    import bench
    import hode
    N = 37
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--total-patients", str(N), "--train-steps", "2", "--no-zscore"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["config"]["patients_total"] == N and out["config"]["patients_per_gpu"] == N / 2 and "zscore_regime" not in out
    assert abs(out["value"] - N * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    dev = torch.device("cuda")
    nn_t, ode = bench.synth_weights(0).to(dev), bench.ODE_DEFAULT.to(dev)
    x0, t, meal, tv = (v.to(dev) for v in bench.synth_cohort(N, 1000))
    tot = 0.0
    for rank in range(2):
        lo, hi = hode.train.shard_bounds(N, rank, 2)
        assert hi - lo == (19 if rank == 0 else 18)
        obs, student = bench.train_problem(dev, x0[lo:hi].contiguous(), t, meal[lo:hi].contiguous(), tv[lo:hi].contiguous(), ode, nn_t, rank)
        ls, _, _, _ = hode.train.hip_loss_and_grads(student, ode, x0[lo:hi].contiguous(), t, meal[lo:hi].contiguous(), tv[lo:hi].contiguous(),
                                                    obs, 64, 4, N * 241 * 6)
        tot += float(ls)
    first = out["train_step"]["loss_first_last"][0]
    assert abs(first - tot / (N * 241 * 6)) < 1e-6 * first
