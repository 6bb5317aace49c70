"""VariationalInference -- mean-field VI trainer over a HybridODENN, mirror of reference inference/vi.py:19-340.

Same constructor, methods, return values, history / checkpoint layout.  What differs is where the work happens:

* `elbo`: the reference integrates the batch once per Monte-Carlo draw in a Python loop (`forward_with_params`, S x B
  per-patient solve_ivp calls, vi.py:88-100).  Here all S x B trajectories are ONE launch -- the S draws ride in the
  kernel's parameter-set dimension (`HybridODENN.elbo`) -- and KL and likelihood are accumulated in fp64.
* gradients: the reference's likelihood term is detached (SciPy round trip), so its `train_step` only ever
  moves the variational parameters along the KL gradient.  With `model.adjoint = True` (the default) the reparameterised
  gradient of the likelihood reaches mu / log_sigma through the adjoint kernel; `model.adjoint = False` reproduces the
  reference's detached behaviour exactly (tests pin both against values captured from the reference).
* `posterior_predictive`: all draws in one launch (`forward_param_sets`).
* `train()` with a validation loader: the reference stores `self.best_state = self.variational_params.state_dict()`
  (vi.py:236) -- references to the LIVE tensors -- so its closing `load_state_dict(self.best_state)` (:259-260) is a no-op and
  training ends with the last epoch's posterior.  The mirror clones the state and restores the best validation epoch;
  `restore_best=False` reproduces the reference (aliasing and all).  tests/test_host_surface.py pins both.
* `prior_params` is accepted and ignored, as in the reference (vi.py:26-52): the priors live in `model.variational_params`.
"""
import logging
from typing import Dict, List, Optional, Tuple

import torch

logger = logging.getLogger(__name__)


class VariationalInference:
    def __init__(self, model, prior_params: Optional[Dict[str, Dict[str, float]]] = None, learning_rate: float = 1e-3,
                 device: Optional[torch.device] = None, restore_best: bool = True):
        self.model = model
        self.restore_best = restore_best
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.learning_rate = learning_rate
        if not hasattr(model, "variational_params") or model.variational_params is None:
            raise ValueError("Model must be initialized with use_variational=True")
        self.variational_params = model.variational_params
        self.optimizer = torch.optim.Adam(self.variational_params.parameters(), lr=learning_rate)
        self.history = {"elbo": [], "kl": [], "log_likelihood": []}

    # ------------------------------------------------------------------ ELBO
    def elbo(self, batch: Dict[str, torch.Tensor], n_samples: int = 5,
             noise_sigma: float = 1.0) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """ELBO = E_q[log p(x | psi)] - KL[q(psi) || p(psi)] with S = n_samples reparameterised draws shared by the whole
        batch (vi.py:60-118).  Returns (elbo, {'elbo', 'kl', 'log_likelihood'})."""
        elbo, comp = self.model.elbo(batch, n_samples=n_samples, noise_sigma=noise_sigma, return_components=True)
        return elbo, comp

    def train_step(self, batch: Dict[str, torch.Tensor], n_samples: int = 5) -> Dict[str, float]:
        """zero_grad -> -ELBO -> backward -> clip_grad_norm_(5.0) -> Adam on the variational parameters (vi.py:120-155)."""
        self.optimizer.zero_grad()
        elbo, components = self.elbo(batch, n_samples=n_samples)
        loss = -elbo
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.variational_params.parameters(), max_norm=5.0)
        self.optimizer.step()
        return {"loss": loss.item(), "elbo": elbo.item(), "kl": components["kl"].item(),
                "log_likelihood": components["log_likelihood"].item()}

    def _to_device(self, batch):
        for key in batch:
            if isinstance(batch[key], torch.Tensor):
                batch[key] = batch[key].to(self.device)
        return batch

    def train(self, train_loader, val_loader=None, epochs: int = 100, n_samples: int = 5,
              early_stopping_patience: int = 10, verbose: bool = True):
        """Epoch loop with optional validation ELBO and early stopping; keeps the best variational state (vi.py:157-259)."""
        best_val_elbo, patience = -float("inf"), 0
        keys = ("loss", "elbo", "kl", "log_likelihood")
        for epoch in range(epochs):
            self.model.train()
            tr = dict.fromkeys(keys, 0.0)
            it = train_loader
            if verbose:
                try:
                    from tqdm import tqdm
                    it = tqdm(train_loader, desc=f"Epoch {epoch + 1}/{epochs}")
                except ImportError:
                    pass
            for batch in it:
                m = self.train_step(self._to_device(batch), n_samples=n_samples)
                for k in keys:
                    tr[k] += m[k]
                if verbose and hasattr(it, "set_postfix"):
                    it.set_postfix({"ELBO": f"{m['elbo']:.4f}"})
            for k in keys:
                tr[k] /= max(len(train_loader), 1)
            va = None
            if val_loader is not None:
                self.model.eval()
                va = dict.fromkeys(keys, 0.0)
                with torch.no_grad():
                    for batch in val_loader:
                        e, c = self.elbo(self._to_device(batch), n_samples=n_samples)
                        va["loss"] += (-e).item()
                        va["elbo"] += e.item()
                        va["kl"] += c["kl"].item()
                        va["log_likelihood"] += c["log_likelihood"].item()
                for k in keys:
                    va[k] /= max(len(val_loader), 1)
                if va["elbo"] > best_val_elbo:
                    best_val_elbo, patience = va["elbo"], 0
                    sd = self.variational_params.state_dict()
                    self.best_state = {k: v.detach().clone() for k, v in sd.items()} if self.restore_best else sd
                else:
                    patience += 1
                if patience >= early_stopping_patience:
                    logger.info(f"Early stopping at epoch {epoch + 1}")
                    break
            self.history["elbo"].append(tr["elbo"])
            self.history["kl"].append(tr["kl"])
            self.history["log_likelihood"].append(tr["log_likelihood"])
            if verbose and (epoch + 1) % 10 == 0:
                logger.info(f"Epoch {epoch + 1}: Train ELBO={tr['elbo']:.4f}, KL={tr['kl']:.4f}, LL={tr['log_likelihood']:.4f}")
                if va is not None:
                    logger.info(f"  Val ELBO={va['elbo']:.4f}")
        if val_loader is not None and hasattr(self, "best_state"):
            self.variational_params.load_state_dict(self.best_state)

    # ------------------------------------------------------------------ posterior
    def sample_posterior(self, n_samples: int = 100) -> List[Dict[str, torch.Tensor]]:
        return self.variational_params.sample(n_samples)

    def posterior_predictive(self, initial_state: torch.Tensor, time_points: torch.Tensor,
                             external_inputs: Optional[Dict[str, torch.Tensor]] = None,
                             n_samples: int = 100) -> Tuple[torch.Tensor, torch.Tensor]:
        """Mean / (unbiased) std of the trajectories over n_samples posterior draws (vi.py:273-312); the draws are taken
        one `sample(1)` at a time like the reference (same RNG consumption) and integrated in ONE launch."""
        with torch.no_grad():
            draws = [self.variational_params.sample(1)[0] for _ in range(n_samples)]
            preds = self.model.forward_param_sets(draws, initial_state, time_points, external_inputs)
        return preds.mean(dim=0), preds.std(dim=0)

    # ------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, path: str):
        torch.save({"variational_params": self.variational_params.state_dict(), "optimizer": self.optimizer.state_dict(),
                    "history": self.history}, path)
        logger.info(f"Checkpoint saved to {path}")

    def load_checkpoint(self, path: str):
        ck = torch.load(path, map_location=self.device, weights_only=True)      # tensors, lists and numbers only
        self.variational_params.load_state_dict(ck["variational_params"])
        self.optimizer.load_state_dict(ck["optimizer"])
        self.history = ck["history"]
        logger.info(f"Checkpoint loaded from {path}")
