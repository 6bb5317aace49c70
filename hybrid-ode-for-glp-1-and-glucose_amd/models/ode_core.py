"""ODECore -- the 6-state mechanistic GLP-1 / glucose right-hand side as an nn.Module.

Mirror of reference models/ode_core.py (class surface, buffer names and order, defaults:
ode_core.py:34-79; equations: ode_core.py:104-161).  This module is the *definition* of the
17 physiological constants (buffers -> state_dict keys `ode_core.<name>`) and a plain-torch,
autograd-differentiable evaluation used for Jacobians and unit tests.  The batched hot path
(HybridODENN.ode_residual / forward) does not call this forward: it hands the 17 constants to
the HIP kernels (csrc/hode_device.h: rhs_eval).
"""
from typing import Dict, Optional

import torch
import torch.nn as nn

# registration order == layout of the `ode_p[17]` vector of the C ABI (include/hode.h)
ODE_PARAM_DEFAULTS = (
    ("a_GI", 0.0104), ("k_I", 0.025), ("rho", 0.003), ("G_b", 5.0), ("I_b", 60.0),      # insulin
    ("E_max", 0.1), ("EC_50", 50.0), ("Glu_b", 80.0),                                    # glucagon
    ("V_max", 9.0), ("K_m", 7.0), ("k_L", 0.02),                                         # GLP-1
    ("k_GE0", 0.01), ("IGD_50", 1000.0), ("g", 2.0),                                     # gastric emptying
    ("p_7", 0.05), ("p_8", 0.001), ("p_9", 0.01),                                        # FFA
)
ODE_PARAM_NAMES = tuple(n for n, _ in ODE_PARAM_DEFAULTS)


class ODECore(nn.Module):
    """State vector x = [G, I, Glu, GLP1, GE, FFA]; inputs u = {meal, tVNS, GD} (all optional)."""

    def __init__(self, params: Optional[Dict[str, float]] = None):
        super().__init__()
        values = dict(ODE_PARAM_DEFAULTS)
        if params is not None:
            values.update(params)
        for name, value in values.items():     # fp32 0-dim buffers, not Parameters (ode_core.py:78-79)
            self.register_buffer(name, torch.tensor(value, dtype=torch.float32))

    def param_vector(self, dtype=None, device=None) -> torch.Tensor:
        """The 17 constants as one vector in C-ABI order (reads the live attributes, so values
        replaced with setattr -- plots/plot_all.py:179-181, hybrid_ode_nn.py:409-411 -- are seen)."""
        vec = torch.stack([torch.as_tensor(getattr(self, n)).reshape(()).to(dtype=dtype or torch.float32)
                           for n in ODE_PARAM_NAMES])
        return vec if device is None else vec.to(device)

    def forward(self, t: torch.Tensor, state: torch.Tensor,
                external_inputs: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        """dx/dt of the mechanistic model.  `t` and `tVNS` do not enter (ode_core.py:81-166)."""
        single = state.dim() == 1
        x = state.unsqueeze(0) if single else state
        G, I, Glu, GLP1, FFA = x[:, 0], x[:, 1], x[:, 2], x[:, 3], x[:, 5]
        u = external_inputs or {}
        meal = u.get("meal", torch.zeros_like(G))
        GD = u.get("GD", torch.zeros_like(G))

        dI = (1.0 + self.rho * GLP1) * self.a_GI * (G - self.G_b) - self.k_I * (I - self.I_b)
        dGlu = -(self.E_max * (GLP1 / (self.EC_50 + GLP1))) * (Glu - self.Glu_b)
        dGLP1 = self.V_max * (G / (self.K_m + G)) - self.k_L * GLP1
        gd_pow = torch.pow(GD, self.g)
        k_GE = self.k_GE0 * (1.0 - gd_pow / (torch.pow(self.IGD_50, self.g) + gd_pow))
        dFFA = -self.p_7 * FFA - self.p_8 * I * FFA + self.p_9 * G * FFA
        dG = meal - 0.01 * (I - self.I_b) + 0.005 * (Glu - self.Glu_b) - k_GE * G
        dGE = torch.zeros_like(G)

        out = torch.stack([dG.expand_as(G), dI, dGlu, dGLP1, dGE, dFFA], dim=-1)
        return out.squeeze(0) if single else out

    def get_steady_state(self, external_inputs: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        """Basal point [G_b, I_b, Glu_b, 0, 0, 1] (ode_core.py:168-187)."""
        ss = torch.zeros(6)
        ss[0], ss[1], ss[2], ss[5] = self.G_b, self.I_b, self.Glu_b, 1.0
        return ss

    def check_mass_balance(self, state: torch.Tensor, derivatives: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Range checks used for validation (ode_core.py:189-211)."""
        G, I = state[..., 0], state[..., 1]
        return {"non_negative": (state >= 0).all(),
                "glucose_range": (G >= 2.0) & (G <= 30.0),
                "insulin_range": (I >= 0.0) & (I <= 1000.0)}
