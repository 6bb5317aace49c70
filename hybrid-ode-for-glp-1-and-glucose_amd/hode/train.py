"""Patient-sharded data-parallel training step on the HIP path.

Trajectories are independent given the parameters, so the cohort is split contiguously over the
ranks (one process per GPU).  Every rank integrates its shard, runs the adjoint, and the ranks
exchange ONE flat fp64 buffer [ MLP grads (P) | ODE-constant grads (17) | loss_sum | n_elements ]
(13 529 values, 108 KB) through a single all-reduce(sum) -- RCCL over xGMI with backend "nccl" --
then apply the identical fused clip+Adam update, so parameters stay bit-identical without a
broadcast.  fp64 because the tail is not a gradient: the squared-error sum is accumulated in fp64
by the MSE kernel and the element count of BASELINE config 4 (8 x 8 192 x 241 x 6 = 94.8 M) is
beyond fp32's 2^24 integers; the gradients ride along (summed in fp64, rounded to fp32 once).
At 108 KB the collective is latency-bound: it is neither bucketed nor overlapped.

The reference has no distributed code at all; the single-rank semantics
mirror train/train_hybrid.py:247-261 (loss, backward, clip_grad_norm_ 5.0, Adam).
"""
from dataclasses import dataclass, field
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from . import _capi as capi

N_ODE = 17


def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous near-equal split of patients [lo, hi) for `rank` (first n_total % world ranks get one more)."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


@dataclass
class TrainState:
    """Flat fp32 MLP parameters + Adam moments (one replica per rank, kept identical)."""
    p: torch.Tensor
    m: torch.Tensor = None
    v: torch.Tensor = None
    step: int = 0
    scratch: torch.Tensor = field(default=None, repr=False)
    tape: torch.Tensor = field(default=None, repr=False)      # reused across steps (several GB)

    def __post_init__(self):
        if self.m is None:
            self.m = torch.zeros_like(self.p)
        if self.v is None:
            self.v = torch.zeros_like(self.p)
        if self.scratch is None and self.p.is_cuda:
            self.scratch = torch.zeros(2, dtype=torch.float32, device=self.p.device)


def pack(gnn: torch.Tensor, gode: Optional[torch.Tensor], loss_sum: torch.Tensor, n_elem: float) -> torch.Tensor:
    """[gnn | gode(17) | loss_sum | n_elem] as one fp64 vector (the only message of a step)."""
    tail = torch.zeros(N_ODE + 2, dtype=torch.float64, device=gnn.device)
    if gode is not None:
        tail[:N_ODE] = gode.double()
    tail[N_ODE] = loss_sum.reshape(()).double()
    tail[N_ODE + 1] = float(n_elem)
    return torch.cat([gnn.double().reshape(-1), tail])


def unpack(buf: torch.Tensor, P: int):
    return buf[:P], buf[P:P + N_ODE], buf[P + N_ODE], buf[P + N_ODE + 1]


def hip_loss_and_grads(p, ode_p, x0, t, meal, tvns, obs, H, L, n_elem_global, rtol=1e-6, atol=1e-8,
                       want_gode=False, state: Optional[TrainState] = None):
    """Local shard: forward solve (tape) -> fused MSE + cotangent -> adjoint.  Returns
    (sum of squared errors fp64[1], gnn, gode|None, solve).  The cotangent is scaled by the GLOBAL
    element count so that summing the ranks' gradients gives the gradient of the global mean."""
    sol = capi.solve_fwd(x0, t, meal, tvns, None, ode_p, p, H, L, rtol=rtol, atol=atol, want_tape=True,
                         tape=None if state is None else state.tape)
    if state is not None:
        state.tape = sol.tape
    loss_sum, gy = capi.mse_fwd_bwd(sol.y, obs, 1.0 / float(n_elem_global))
    _, gnn, gode = capi.solve_bwd(sol, gy, want_gnn=True, want_gode=want_gode)
    return loss_sum, gnn, gode, sol


def train_step(state: TrainState, compute: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], float]],
               lr: float = 1e-3, max_norm: float = 5.0, betas=(0.9, 0.999), eps: float = 1e-8,
               group=None, optimizer: Optional[Callable] = None, host_staged: bool = False) -> float:
    """One optimisation step.  `compute(p)` -> (loss_sum, gnn, gode|None, n_local_elements) for the local shard
    (gradients already scaled for the global mean).  Returns the global mean loss.
    `optimizer(state, g)` overrides the fused HIP Adam (used by the CPU/gloo tests); `host_staged` routes the
    collective through host memory (gloo rehearsal of the multi-rank path on a single GPU)."""
    P = state.p.numel()
    loss_sum, gnn, gode, n_local = compute(state.p)
    if not (dist.is_available() and dist.is_initialized()):
        # no process group (a plain single-GPU run): nothing to exchange, no message to build.  (With a group -- also one of
        # size 1 -- the step goes through the message and the collective: that is how the RCCL path is exercised on one GPU)
        state.step += 1
        if optimizer is not None:
            optimizer(state, gnn.float())
        else:
            capi.adam_step(state.p, gnn.float().contiguous(), state.m, state.v, lr, betas[0], betas[1], eps, state.step,
                           max_norm=max_norm, scratch=state.scratch)
        return loss_sum.reshape(()).double() / float(n_local)
    buf = pack(gnn, gode, loss_sum, n_local)
    if host_staged:                                                   # gloo rehearsal with device tensors
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        buf.copy_(host)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)      # the step's only collective
    g, _gode, lsum, n_tot = unpack(buf, P)
    state.step += 1
    if optimizer is not None:
        optimizer(state, g.float())
    else:
        capi.adam_step(state.p, g.float().contiguous(), state.m, state.v, lr, betas[0], betas[1], eps, state.step,
                       max_norm=max_norm, scratch=state.scratch)
    return lsum / n_tot
