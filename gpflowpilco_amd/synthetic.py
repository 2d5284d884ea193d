"""Seeded synthetic SVGP dynamics models and input distributions (bench + parity tests).

Recipe of SURVEY.md section 8d / BASELINE.md: Z ~ U[0,1]^{M x d} shared by all L outputs,
lengthscales log-uniform in [0.3, 3], signal variance 0.89^2, noise 1e-2 * variance, targets
drawn from the GP prior (+ noise), q_mu / q_sqrt = the exact posterior at Z (whitened),
mu0 ~ U[0,1]^d, Sigma0 = random correlation scaled to std 0.1 (the reference tests'
``generate_covariance``, ``tests/utils.py:99-121``).

q_mu / q_sqrt are formed in whitened coordinates from (I + L^T L / s2)^-1, positive definite by
construction (no eigendecomposition or eigenvalue clipping in the set-up path).

One deviation, stated in DESIGN.md: with ``stable=True`` the targets are
``-0.5 (z_a - 0.5) + 0.25 * prior draw`` so that a closed d == L rollout stays inside the
data's support; with pure prior draws the state leaves [0,1]^d within a few steps, every q
underflows to zero and the timed kernels would run on all-zero operands.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from .linalg import cholesky

from . import models as gp

F64 = torch.float64


@dataclass
class SyntheticSVGP:
  Z: np.ndarray            # [M, d] shared inducing inputs
  lengthscales: np.ndarray  # [L, d]
  variance: np.ndarray     # [L]
  noise: np.ndarray        # [L]
  q_mu: np.ndarray         # [M, L] (whitened)
  q_sqrt: np.ndarray       # [L, M, M] (whitened)
  mean_c: Optional[np.ndarray] = None
  whiten: bool = True

  @property
  def shape(self):
    return self.lengthscales.shape[0], self.Z.shape[0], self.Z.shape[1]

  def to_model(self, device="cpu") -> gp.SVGP:
    L = self.lengthscales.shape[0]
    kernels = [gp.SquaredExponential(variance=torch.tensor(self.variance[a], dtype=F64, device=device),
                                     lengthscales=torch.tensor(self.lengthscales[a], dtype=F64, device=device))
               for a in range(L)]
    Zt = torch.tensor(self.Z, dtype=F64, device=device)
    iv = gp.SharedIndependentInducingVariables(gp.InducingPoints(Zt))
    mean = gp.Zero() if self.mean_c is None else gp.Constant(torch.tensor(self.mean_c, dtype=F64, device=device))
    return gp.SVGP(kernel=gp.SeparateIndependent(kernels), inducing_variable=iv,
                   q_mu=torch.tensor(self.q_mu, dtype=F64, device=device),
                   q_sqrt=torch.tensor(self.q_sqrt, dtype=F64, device=device),
                   whiten=self.whiten, mean_function=mean, num_latent_gps=L)


def make_svgp(L: int, M: int, d: int, seed: int, stable: bool = True,
              device: str = "cpu", mean_c: bool = False,
              ls_bounds=(0.3, 3.0)) -> SyntheticSVGP:
  """Build the synthetic model in float64 (torch; ``device`` only speeds up the Choleskys)."""
  rng = np.random.default_rng(seed)
  Z = rng.uniform(size=(M, d))
  ls = np.exp(rng.uniform(np.log(ls_bounds[0]), np.log(ls_bounds[1]), size=(L, d)))
  var = np.full(L, 0.89 ** 2)
  noise = 1e-2 * var
  eps_f = rng.standard_normal((L, M))
  eps_n = rng.standard_normal((L, M))
  q_mu = np.empty((M, L))
  q_sqrt = np.empty((L, M, M))
  Zt = torch.tensor(Z, dtype=F64, device=device)
  eye = torch.eye(M, dtype=F64, device=device)
  for a in range(L):
    A = Zt / torch.tensor(ls[a], dtype=F64, device=device)
    d2 = (A * A).sum(-1)[:, None] + (A * A).sum(-1)[None, :] - 2.0 * A @ A.T
    K = var[a] * torch.exp(-0.5 * d2.clamp_min(0.0))
    Lk = cholesky(K + gp.DEFAULT_JITTER * eye)                       # the model's own prior factor (Kuu + jitter)
    draw = Lk @ torch.tensor(eps_f[a], dtype=F64, device=device)
    if stable:
      col = Zt[:, a % d]
      y = -0.5 * (col - 0.5) + 0.25 * draw
    else:
      y = draw
    y = y + np.sqrt(noise[a]) * torch.tensor(eps_n[a], dtype=F64, device=device)
    # Exact posterior of u ~ N(0, Lk Lk^T) given y = u + eps, in whitened coordinates u = Lk v:
    #   cov(v | y) = (I + Lk^T Lk / s2)^-1 =: A^-1,   E[v | y] = A^-1 Lk^T y / s2.
    # A is positive definite with eigenvalues >= 1 BY CONSTRUCTION (no K - K (K + s2 I)^-1 K cancellation, hence no
    # eigenvalue clipping and no eigh in the set-up path).  q_sqrt must be LOWER triangular with
    # q_sqrt q_sqrt^T = A^-1: factor A = U U^T with U UPPER triangular (Cholesky of the index-reversed matrix),
    # then A^-1 = U^-T U^-1 and q_sqrt = U^-T.
    Am = eye + (Lk.T @ Lk) / noise[a]
    Am = 0.5 * (Am + Am.T)
    v = torch.cholesky_solve((Lk.T @ y / noise[a])[:, None], cholesky(Am))
    Rrev = cholesky(torch.flip(Am, (0, 1)))
    U = torch.flip(Rrev, (0, 1))                                     # upper triangular, U U^T = A
    q_sqrt[a] = torch.linalg.solve_triangular(U.T, eye, upper=False).cpu().numpy()
    q_mu[:, a] = v[:, 0].cpu().numpy()
  mc = rng.standard_normal(L) * 0.1 if mean_c else None
  return SyntheticSVGP(Z=Z, lengthscales=ls, variance=var, noise=noise, q_mu=q_mu, q_sqrt=q_sqrt, mean_c=mc)


def generate_covariance(rng, ndims, sample_shape=(), scale=None):
  """Random-eigen covariance prior rescaled to a marginal std (tests/utils.py:99-121)."""
  shape = tuple(sample_shape)
  eigen_vals = -np.log(rng.uniform(size=shape + (1, ndims)))
  A = rng.standard_normal(shape + (ndims, ndims))
  orthog = np.linalg.svd(A, full_matrices=True)[0]
  sqrt_cov = np.sqrt(eigen_vals) * orthog
  cov = sqrt_cov @ np.swapaxes(sqrt_cov, -1, -2)
  if scale is not None:
    istd = 1.0 / np.sqrt(np.diagonal(cov, axis1=-2, axis2=-1))
    cov = (scale ** 2) * cov * istd[..., None] * istd[..., None, :]
  return cov


def make_inputs(B: int, d: int, seed: int, scale: float = 0.1, lo: float = 0.0, hi: float = 1.0):
  """mu0 ~ U[lo,hi]^d, Sigma0 random correlation scaled to std ``scale`` -> numpy float64."""
  rng = np.random.default_rng(seed)
  mu = rng.uniform(lo, hi, size=(B, d))
  Sigma = generate_covariance(rng, d, (B,), scale)
  return mu, Sigma


def make_policy(M: int, d: int, seed: int, scale: float = 0.3, ls_bounds=(0.7, 2.0)) -> SyntheticSVGP:
  """A one-latent SVGP policy regressor on d encoded inputs (the reference's KernelRegressor policy with M = 30
  kernel centres, ``examples/cartpole_swingup/settings.py``): random centres in the unit cube, random q_mu of the
  given scale.  Only its mean is used (models/core.py:60-62), so q_sqrt is the identity."""
  rng = np.random.default_rng(seed)
  Z = rng.uniform(size=(M, d))
  ls = np.exp(rng.uniform(np.log(ls_bounds[0]), np.log(ls_bounds[1]), size=(1, d)))
  var = np.full(1, 0.89 ** 2)
  return SyntheticSVGP(Z=Z, lengthscales=ls, variance=var, noise=1e-2 * var, q_mu=scale * rng.standard_normal((M, 1)),
                       q_sqrt=np.eye(M)[None].copy())


def make_cartpole_like(M_drift: int = 100, M_policy: int = 30, seed: int = 1000, device: str = "cpu"):
  """BASELINE configs[0] wiring (examples/cartpole_swingup/swingup_loops.py:41-91): state x (4) -> trig encoder on the
  angle (dim 1) -> e (5) -> policy u (1) -> drift input d = (e, u) (6) -> dx (4).  Returns (drift, policy) as
  SyntheticSVGP; the drift's action axis (input 5) is stretched to [-2, 2], the range of u = 2 (2 Phi(f) - 1)."""
  drift = make_svgp(4, M_drift, 6, seed=seed, device=device, ls_bounds=(0.8, 3.0))
  drift.Z = drift.Z * np.array([1, 1, 1, 1, 1, 4.0]) - np.array([0, 0, 0, 0, 0, 2.0])
  return drift, make_policy(M_policy, 5, seed + 1)
