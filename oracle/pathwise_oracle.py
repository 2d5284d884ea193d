"""fp64 CPU restatement of the pathwise (decoupled-sampling) rollout (TEST INFRASTRUCTURE ONLY;
SURVEY.md section 8 row f-3, BASELINE.json configs[4]).

**Parity unpinned.**  Every floating-point operation of this path lives in the un-vendored
third-party package ``gpflow-sampling>=0.2`` (``/root/reference/setup.py:5``); the reference only
calls it (``gpflow_pilco/models/svgp.py:124-130``: ``PathwiseSVGP.__call__ = predict_f_samples``;
``loops/pilco.py:263-298``: ``generate_paths(num_samples, num_bases, sample_axis=0)`` then
``solve_forward`` with the plain ``Euler`` solver, ``dynamics/solvers.py:50-65``; tensor branch of
``forward_sde``, ``dynamics/forward_sde.py:23-31``) and no reference test touches it.  Restated here
is the published algorithm of that package (Wilson et al. 2020, "Efficiently sampling functions
from Gaussian process posteriors"; decoupled sampler with a random-Fourier prior and an
inducing-point update):

  f_s(x) = sum_k w_{s,k} phi_k(x) + sum_m v_{s,m} k(x, z_m) + mean,      per latent GP
  phi_k(x) = sqrt(2 sigma^2 / K) cos(omega_k . x + b_k),  omega_k ~ N(0, Lambda^-1),  b_k ~ U[0, 2 pi)
  w_s ~ N(0, I),   u_s ~ q(u)  (whitened: u = Luu (q_mu + q_sqrt eps)),
  v_s = (Kuu + jitter I)^-1 (u_s - Phi(Z) w_s)

The pins available here are statistical: the sample mean / variance of f_s(x) over s must equal
the SVGP predictive mean / variance (``tests/test_pathwise.py``).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.linalg import cho_solve, cholesky

from oracle import mm_oracle as mo


@dataclass
class Paths:
  omega: np.ndarray    # [L, K, d]  spectral frequencies / lengthscales
  phase: np.ndarray    # [L, K]
  w: np.ndarray        # [S, L, K]  prior weights
  v: np.ndarray        # [S, L, M]  update weights


def features(omega, phase, variance, x):
  """phi(x) for one latent: x [n, d] -> [n, K]."""
  K = omega.shape[0]
  return np.sqrt(2.0 * variance / K) * np.cos(x @ omega.T + phase[None, :])


def draw_paths(rng, model: mo.SVGPParams, num_samples: int, num_bases: int) -> Paths:
  L, M, d = model.Z.shape
  S, K = num_samples, num_bases
  omega = rng.standard_normal((L, K, d)) / model.lengthscales[:, None, :]
  phase = rng.uniform(0.0, 2.0 * np.pi, size=(L, K))
  w = rng.standard_normal((S, L, K))
  v = np.empty((S, L, M))
  for a in range(L):
    Kuu = mo.se_kernel(model.Z[a], None, model.lengthscales[a], model.variance[a]) \
        + model.kuu_jitter * np.eye(M)
    Lu = cholesky(Kuu, lower=True)
    eps = rng.standard_normal((S, M))
    q_sqrt = np.tril(model.q_sqrt[a])
    u = model.q_mu[:, a][None, :] + eps @ q_sqrt.T                   # samples of the (whitened) q(u)
    if model.whiten:
      u = u @ Lu.T
    Phi_Z = features(omega[a], phase[a], model.variance[a], model.Z[a])   # [M, K]
    resid = u - w[:, a, :] @ Phi_Z.T                                  # [S, M]
    v[:, a, :] = cho_solve((Lu, True), resid.T).T
  return Paths(omega=omega, phase=phase, w=w, v=v)


def eval_paths(paths: Paths, model: mo.SVGPParams, x: np.ndarray) -> np.ndarray:
  """f_s(x_s): x [S, d] (one input per sample path) -> [S, L]."""
  S, L, K = paths.w.shape
  out = np.empty((S, L))
  for a in range(L):
    phi = features(paths.omega[a], paths.phase[a], model.variance[a], x)          # [S, K]
    kxz = mo.se_kernel(x, model.Z[a], model.lengthscales[a], model.variance[a])   # [S, M]
    out[:, a] = np.sum(paths.w[:, a, :] * phi, -1) + np.sum(paths.v[:, a, :] * kxz, -1)
  if model.mean_c is not None:
    out = out + np.asarray(model.mean_c)[None]
  return out


def rollout(paths: Paths, model: mo.SVGPParams, x0: np.ndarray, num_steps: int, dt: float = 1.0,
            keep: bool = False):
  """Euler.step (solvers.py:50-65, no diffusion) folded over the horizon, drift-only, d == L."""
  x = x0.copy()
  traj = []
  for _ in range(num_steps):
    x = x + dt * eval_paths(paths, model, x)
    if keep:
      traj.append(x.copy())
  return (x, np.stack(traj)) if keep else x


def tensor_cost(x, target, precis):
  """GaussianObjective on samples (components.py:39-41)."""
  err = x - target
  return -np.exp(-0.5 * np.einsum('...i,ij,...j->...', err, precis, err))


# ---- the pathwise POLICY rollout (PathwisePILCO._policy_loss_closure) ------------------------------------------------
def encode(x, active_dims):
  """TrigonometricEncoder on tensors (gpflow_pilco/components.py:44-75): [sin a, cos a, x_inactive (ascending)]."""
  nx = x.shape[-1]
  active = list(active_dims)
  inactive = [i for i in range(nx) if i not in set(active)]
  a = x[..., active]
  return np.concatenate([np.sin(a), np.cos(a), x[..., inactive]], axis=-1)


def policy_mean(policy: mo.SVGPParams, e):
  """KernelRegressor.__call__ = predict_f(x)[0] (gpflow_pilco/models/core.py:60-62) of a one-latent SVGP: e [n, ne] -> [n]."""
  M = policy.Z.shape[1]
  Kuu = mo.se_kernel(policy.Z[0], None, policy.lengthscales[0], policy.variance[0]) + policy.kuu_jitter * np.eye(M)
  Lu = cholesky(Kuu, lower=True)
  u = policy.q_mu[:, 0]
  if policy.whiten:
    u = Lu @ u
  beta = cho_solve((Lu, True), u)
  f = mo.se_kernel(e, policy.Z[0], policy.lengthscales[0], policy.variance[0]) @ beta
  if policy.mean_c is not None:
    f = f + np.asarray(policy.mean_c).reshape(-1)[0]
  return f


def policy_rollout_costs(paths: Paths, drift: mo.SVGPParams, policy: mo.SVGPParams, head_scale, head_shift, active_dims,
                         target, precis, x0, num_steps, dt=1.0, keep=False):
  """loops/pilco.py:263-298 for one batch of sample paths: per step e = encoder(x); u = scale (Phi(policy(e)) + shift)
  (InverseLinkWrapper with Chain[Scale, Shift, NormalCDF]: applied NormalCDF -> Shift -> Scale); x <- x + dt f_s([e, u])
  (forward_sde.py:23-31, solvers.py:50-65); cost[h] = objective(encoder(x)) (pilco.py:272-275).  -> cost [H, S]."""
  from scipy.special import ndtr
  x = np.array(x0, dtype=np.float64, copy=True)
  costs, states = [], [x.copy()]
  for _ in range(num_steps):
    e = encode(x, active_dims)
    u = head_scale * (ndtr(policy_mean(policy, e)) + head_shift)
    x = x + dt * eval_paths(paths, drift, np.concatenate([e, u[:, None]], axis=-1))
    costs.append(tensor_cost(encode(x, active_dims), target, precis))
    states.append(x.copy())
  return (np.stack(costs), np.stack(states)) if keep else np.stack(costs)
