"""Host-side SDE objects: the per-step scalars of the reverse SDE.

Mirrors the interface of the reference's `ccsd/src/sde.py` (VPSDE :345-503, VESDE :506-669,
subVPSDE :672-786) for the parts the sampling path uses.  All arithmetic is fp32 torch on the CPU
in the same op order as the reference, so the tables handed to the device (timestep indices,
sigmas, alphas, G, std) are bit-identical to what the reference computes; the device never
re-derives them.  The tensors the reference broadcasts per sample are identical across the
batch (vec_t = ones(B) * t, solver.py:1127), so one scalar per step suffices.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np
import torch


class SDE:
    """Common surface: N (number of scales), T (= 1), sde(), marginal_prob(), discretize(), priors."""

    kind = "?"

    def __init__(self, N: int):
        self.N = N

    @property
    def T(self) -> int:
        return 1

    # -- priors: drawn on the CPU generator like the reference (sde.py:436, 448-449)
    def prior_sampling(self, shape: Sequence[int]) -> torch.Tensor:
        return torch.randn(*shape)

    def prior_sampling_sym(self, shape: Sequence[int]) -> torch.Tensor:
        z = torch.randn(*shape).triu(1)
        return z + z.transpose(-1, -2)

    def timestep_index(self, t: torch.Tensor) -> torch.Tensor:
        """(t * (N - 1) / T).long()  (sde.py:477, 639; solver.py:753)."""
        return (t * (self.N - 1) / self.T).long()

    def discretize(self, x: torch.Tensor, t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Base-class Euler discretisation (sde.py:107-111); VP and VE override it."""
        dt = 1 / self.N
        drift, diffusion = self.sde(x, t)
        return drift * dt, diffusion * torch.sqrt(torch.tensor(dt))

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(N={self.N}, T={self.T})"


class _BetaSchedule(SDE):
    def __init__(self, beta_min: float = 0.1, beta_max: float = 20.0, N: int = 1000):
        super().__init__(N)
        self.beta_0, self.beta_1 = beta_min, beta_max
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1.0 - self.discrete_betas

    def _beta(self, t):
        return self.beta_0 + t * (self.beta_1 - self.beta_0)

    def _log_mean_coeff(self, t):
        return -0.25 * t**2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0


class VPSDE(_BetaSchedule):
    kind = "VP"

    def __init__(self, beta_min: float = 0.1, beta_max: float = 20.0, N: int = 1000):
        super().__init__(beta_min, beta_max, N)
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)

    def sde(self, x, t):
        beta_t = self._beta(t)
        return -0.5 * beta_t[:, None, None] * x, torch.sqrt(beta_t)

    def marginal_prob(self, x, t):
        lmc = self._log_mean_coeff(t)
        return torch.exp(lmc[:, None, None]) * x, torch.sqrt(1.0 - torch.exp(2.0 * lmc))

    def discretize(self, x, t):
        i = self.timestep_index(t)
        beta, alpha = self.discrete_betas[i], self.alphas[i]
        return torch.sqrt(alpha)[:, None, None] * x - x, torch.sqrt(beta)

    def transition(self, x, t, dt: float):
        c = 0.25 * dt * (2 * self.beta_0 + (2 * t + dt) * (self.beta_1 - self.beta_0))
        return torch.exp(-c[:, None, None]) * x, torch.sqrt(1.0 - torch.exp(2.0 * c))


class subVPSDE(_BetaSchedule):
    kind = "subVP"

    def sde(self, x, t):
        beta_t = self._beta(t)
        discount = 1.0 - torch.exp(-2 * self.beta_0 * t - (self.beta_1 - self.beta_0) * t**2)
        return -0.5 * beta_t[:, None, None] * x, torch.sqrt(beta_t * discount)

    def marginal_prob(self, x, t):
        lmc = self._log_mean_coeff(t)
        return torch.exp(lmc)[:, None, None] * x, 1 - torch.exp(2.0 * lmc)


class VESDE(SDE):
    kind = "VE"

    def __init__(self, sigma_min: float = 0.01, sigma_max: float = 50.0, N: int = 1000):
        super().__init__(N)
        self.sigma_min, self.sigma_max = sigma_min, sigma_max
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(sigma_min), np.log(sigma_max), N))

    def _sigma(self, t):
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def sde(self, x, t):
        g = self._sigma(t) * torch.sqrt(torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min))))
        return torch.zeros_like(x), g

    def marginal_prob(self, x, t):
        return x, self._sigma(t)

    def discretize(self, x, t):
        i = self.timestep_index(t)
        sigma = self.discrete_sigmas[i]
        prev = torch.where(i == 0, torch.zeros_like(t), self.discrete_sigmas[i - 1])
        return torch.zeros_like(x), torch.sqrt(sigma**2 - prev**2)

    def transition(self, x, t, dt: float):
        return x, torch.sqrt(torch.square(self._sigma(t)) - torch.square(self._sigma(t + dt)))


def step_coefficients(sdes, predictor: str, probability_flow: bool, eps: float):
    """[diff_steps][len(sdes)][10] float32 table of (sscale, alpha, pa, pb, pc, m1, s1, d, m2, s2) -- include/ccsd_hip.h.

    Restates, per step and per target, the scalar part of
      get_score_fn(_cc)                      losses.py:157-163, 189-193
      LangevinCorrector alpha                solver.py:752-756
      ReverseDiffusionPredictor + RSDE.discretize   solver.py:430-457, sde.py:329-340
      EulerMaruyamaPredictor + RSDE.sde             solver.py:275-307, sde.py:290-302
    with v_mean = pa * v + pb * net and v = v_mean + pc * z.  predictor == "S4" (S4_solver, solver.py:1266-1352) fills
    m1, s1 = transition(1, t, dt/2); d = -g(t)^2 * sscale * dt (Sdrift * dt); m2, s2 = transition(1, t + dt/2, dt/2),
    with the Langevin alpha taken at sde_x's timestep index for every target (solver.py:1296).
    """
    if predictor not in ("Reverse", "Euler", "S4"):
        raise NotImplementedError(f"Predictor {predictor} not yet supported. Select from [Reverse, Euler].")
    sde_adj = sdes[1]
    steps = sde_adj.N
    timesteps = torch.linspace(sde_adj.T, eps, steps)
    out = np.zeros((steps, 3, 10), dtype=np.float32)
    one = torch.ones(1, 1, 1)
    half = 0.5 if probability_flow else 1.0
    for i in range(steps):
        t = torch.ones(1) * timesteps[i]
        for k, s in enumerate(sdes):
            if isinstance(s, VESDE):
                sscale = torch.ones(1)
                alpha = torch.ones(1)
            else:
                sscale = -1.0 / s.marginal_prob(torch.zeros(1, 1, 1), t)[1]
                alpha = s.alphas[s.timestep_index(t)]
            if predictor == "S4":
                dt = -1.0 / steps                                    # diff_steps = sde_adj.N (solver.py:1281)
                vec_dt = torch.ones(1) * (dt / 2)
                if not isinstance(s, VESDE):                         # alpha index from sde_x for all targets (solver.py:1296)
                    alpha = s.alphas[(t * (sdes[0].N - 1) / sdes[0].T).long()]
                g = s.sde(one, t)[1]
                m1, s1 = s.transition(one, t, vec_dt)
                m2, s2 = s.transition(one, t + vec_dt, vec_dt)
                d = -(g**2) * sscale * dt
                out[i, k] = [float(sscale), float(alpha), 0.0, 0.0, 0.0, float(m1.reshape(-1)[0]), float(s1), float(d),
                             float(m2.reshape(-1)[0]), float(s2)]
                continue
            if predictor == "Reverse":
                f, G = s.discretize(one, t)            # f evaluated at v = 1: v_mean = v - f(v) + G^2 * score
                pa = 1.0 - f.reshape(1)
                pb = G**2 * half * sscale
                pc = torch.zeros_like(G) if probability_flow else G
            else:
                if probability_flow:
                    # the reference's RSDE.sde returns the python float 0.0 as diffusion (sde.py:301),
                    # which EulerMaruyamaPredictor then indexes (solver.py:284)
                    raise TypeError("'float' object is not subscriptable")
                dt = -1.0 / s.N
                drift, g = s.sde(one, t)
                pa = 1.0 + drift.reshape(1) * dt
                pb = -(g**2) * half * dt * sscale
                pc = g * np.sqrt(-dt)
            out[i, k, :5] = [float(sscale), float(alpha), float(pa), float(pb), float(pc)]
    return out
