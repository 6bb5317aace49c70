"""NNResidual -- the MLP correction g_NN(t, x, GLP1, tVNS) as an nn.Module.

Mirror of reference models/nn_residual.py (constructor :28-98, forward :100-151,
regularization_loss :198-223, get_feature_importance :153-196).  `self.network` is an indexable
nn.Sequential whose Linear layers sit at indices 0,2,4,... so state_dict keys are
`nn_residual.network.{0,2,4,6,8}.{weight,bias}` for the default 4x64 network.
The batched hot path uses the flat parameter vector (`flat_parameters`) in the HIP kernels;
this torch forward is for unit tests / autograd w.r.t. inputs.
"""
import torch
import torch.nn as nn

_ACTIVATIONS = {"relu": nn.ReLU, "tanh": nn.Tanh, "elu": nn.ELU, "leaky_relu": lambda: nn.LeakyReLU(0.1)}
_ACT_CODES = {"relu": 0, "tanh": 1, "elu": 2, "leaky_relu": 3}        # include/hode.h HODE_ACT_*


class NNResidual(nn.Module):
    def __init__(self, input_dim: int = 9, hidden_dim: int = 64, output_dim: int = 6, n_layers: int = 4,
                 activation: str = "relu", dropout: float = 0.0):
        super().__init__()
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.n_layers, self.dropout = n_layers, dropout
        self.activation_name = activation if activation in _ACTIVATIONS else "relu"
        self.activation = _ACTIVATIONS[self.activation_name]()

        layers, width = [], input_dim
        for _ in range(n_layers):
            layers += [nn.Linear(width, hidden_dim), self.activation]
            if dropout > 0:
                layers.append(nn.Dropout(dropout))
            width = hidden_dim
        layers.append(nn.Linear(width, output_dim))
        self.network = nn.Sequential(*layers)
        self._initialize_zero_output()

    def _initialize_zero_output(self):
        """Output layer zero => the hybrid model starts as the pure ODE; hidden layers
        xavier_normal(gain=0.1), zero bias (nn_residual.py:89-98)."""
        linears = [m for m in self.network if isinstance(m, nn.Linear)]
        with torch.no_grad():
            linears[-1].weight.zero_()
            linears[-1].bias.zero_()
        for lin in linears[:-1]:
            nn.init.xavier_normal_(lin.weight, gain=0.1)
            nn.init.zeros_(lin.bias)

    # ---- what the HIP kernels consume -----------------------------------------------------
    def hip_supported(self) -> bool:
        """include/hode.h: no dropout, input 9, output 6, hidden <= 128, 1..8 hidden layers (ReLU up to 64 x 4: the
        register-resident kernels; beyond that, and for tanh / elu / leaky_relu at any shape, the generic streamed-weight
        kernels)."""
        return (self.activation_name in _ACT_CODES and self.dropout == 0 and self.input_dim == 9
                and self.output_dim == 6 and 1 <= self.hidden_dim <= 128 and 1 <= self.n_layers <= 8)

    @property
    def hip_layers(self) -> int:
        """The `L` argument of the C ABI: hidden layers in bits 0..7, activation code in bits 8..15 (HODE_LAYERS)."""
        return self.n_layers | (_ACT_CODES[self.activation_name] << 8)

    def flat_parameters(self) -> torch.Tensor:
        """W1,b1,...,Wout,bout concatenated in parameters() order (differentiable cat)."""
        return torch.cat([p.reshape(-1) for p in self.parameters()])

    # ---- torch evaluation --------------------------------------------------------------------
    @staticmethod
    def _as_batch(v: torch.Tensor, n: int) -> torch.Tensor:
        if v.dim() == 0:
            return v.unsqueeze(0).expand(n)
        if v.dim() == 1 and v.shape[0] == 1:
            return v.expand(n)
        return v

    def forward(self, t: torch.Tensor, state: torch.Tensor, glp1: torch.Tensor, tvns: torch.Tensor) -> torch.Tensor:
        """Input row = [t, G, I, Glu, GLP1, GE, FFA, glp1, tvns] (raw, un-normalised)."""
        single = state.dim() == 1
        x = state.unsqueeze(0) if single else state
        n = x.shape[0]
        t, glp1, tvns = (self._as_batch(v, n) for v in (t, glp1, tvns))
        row = torch.cat([t.unsqueeze(-1), x, glp1.unsqueeze(-1), tvns.unsqueeze(-1)], dim=-1)
        out = self.network(row)
        return out.squeeze(0) if single else out

    def get_feature_importance(self, t, state, glp1, tvns) -> torch.Tensor:
        """Mean |d out_i / d input| averaged over outputs (gradient sensitivity)."""
        x = state.unsqueeze(0) if state.dim() == 1 else state
        t, glp1, tvns = (v.unsqueeze(0) if v.dim() == 0 else v for v in (t, glp1, tvns))
        row = torch.cat([t.unsqueeze(-1), x, glp1.unsqueeze(-1), tvns.unsqueeze(-1)], dim=-1).detach()
        row.requires_grad_(True)
        out = self.network(row)
        importance = torch.zeros(self.input_dim)
        for i in range(self.output_dim):
            (g,) = torch.autograd.grad(out[:, i].sum(), row, retain_graph=True)
            importance += g.abs().mean(dim=0).cpu()
        return importance / self.output_dim

    def regularization_loss(self, l2_weight: float = 1e-4, sparsity_weight: float = 0.0) -> torch.Tensor:
        """l2_weight * sum ||W_l||^2 over Linear weights (biases excluded)."""
        reg = 0.0
        if l2_weight > 0:
            for m in self.network:
                if isinstance(m, nn.Linear):
                    reg = reg + l2_weight * m.weight.pow(2).sum()
        return reg
