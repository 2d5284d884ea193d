"""Rollout drivers: ``forward_sde``, ``MomentMatchingEuler`` and ``DynamicalSystem``.

Mirrors ``gpflow_pilco/dynamics/forward_sde.py:17-137``, ``dynamics/solvers.py:49-135`` and
``dynamics/dynamical_system.py:17-66`` on torch tensors.  The per-step GP propagation goes
through ``moment_matching`` (HIP kernels); ``closed_rollout`` is the fused fast path for the
drift-only case (one C-ABI call enqueues all H steps).
"""
from __future__ import annotations

from typing import Any, Callable, List, Optional, Sequence, Tuple

import torch

from . import ops
from .linalg import index_tensor
from .moment_matching import GaussianMatch, GaussianMoments, moment_matching

__all__ = ("forward_sde", "Euler", "MomentMatchingEuler", "DynamicalSystem", "closed_rollout")


def forward_sde(x, drift, noise=None, policy=None, encoder=None):
  """``forward_sde`` dispatcher (forward_sde.py:17): composition encoder -> policy -> drift."""
  if isinstance(x, torch.Tensor):                                   # forward_sde.py:23-31
    e = x if encoder is None else encoder(x)
    eu = e if policy is None else torch.cat([e, policy(e)], dim=-1)
    return drift(eu), None if noise is None else noise(e)
  if not isinstance(x, GaussianMoments):
    raise NotImplementedError(type(x))
  if policy is None and encoder is None:                            # :34-46
    match_drift = moment_matching(x, drift)
    match_noise = None if noise is None else moment_matching(x, noise)
    return match_drift, match_noise
  if encoder is None:                                               # :49-68
    match_policy = moment_matching(x, policy)
    match_drift = moment_matching(match_policy.joint(), drift)
    if match_drift.cross[1]:
      preinv = match_policy.cross[1]
      cross = (match_policy.cross_covariance(preinv=preinv)
               @ match_drift.cross_covariance(preinv=True), preinv)
    else:
      cross = match_drift.cross_covariance()[..., :x.mean().shape[-1], :], False
    chain = GaussianMatch(x=x, y=match_drift.y, cross=cross)
    match_noise = None if noise is None else moment_matching(x, noise)
    return chain, match_noise
  if policy is None:                                                # :71-92
    match_encoder = moment_matching(x, encoder)
    match_drift = moment_matching(match_encoder.y, drift)
    preinv = match_encoder.cross[1]
    Sxe = match_encoder.cross_covariance(preinv=preinv)
    cross = Sxe @ match_drift.cross_covariance(preinv=True), preinv
    chain = GaussianMatch(x=x, y=match_drift.y, cross=cross)
    if noise is None:
      return chain, None
    match_noise = moment_matching(match_encoder.y, noise)
    crossz = Sxe @ match_noise.cross_covariance(preinv=True), preinv
    return chain, GaussianMatch(x=x, y=match_noise.y, cross=crossz)
  # encoder and policy both present                                 # :95-137
  match_encoder = moment_matching(x, encoder)
  match_policy = moment_matching(match_encoder.y, policy)
  match_drift = moment_matching(match_policy.joint(), drift)
  ndims_x = x.mean().shape[-1]
  ndims_u = match_policy.y[0].shape[-1]
  ndims_b = ndims_x - len(encoder.active_dims)
  active, inactive = encoder.get_partition_indices(ndims_x)
  if match_encoder.cross[1]:
    Sax = x.covariance(dense=True).index_select(-2, index_tensor(active, x.mean().device))
    Sae = Sax @ match_encoder.cross_covariance(preinv=True)
  else:
    Sxe = match_encoder.cross_covariance(dense=True)
    Sae = Sxe.index_select(-2, index_tensor(active, Sxe.device))
  Sau = Sae @ match_policy.cross_covariance(preinv=True)
  _, perm = zip(*sorted(zip(tuple(active) + tuple(inactive), range(ndims_x))))
  Sad = torch.cat([Sae, Sau], dim=-1)
  Sdd = match_drift.x.covariance()
  nd = Sdd.shape[-2]
  Sbd = Sdd[..., nd - ndims_b - ndims_u: nd - ndims_u, :]
  Sxd = torch.cat([Sad, Sbd], dim=-2).index_select(-2, index_tensor(perm, Sad.device))
  Sxf = Sxd @ match_drift.cross_covariance(preinv=True)
  chain = GaussianMatch(x=x, y=match_drift.y, cross=(Sxf, False))
  if noise is None:
    return chain, None
  preinv = match_encoder.cross[1]
  match_noise = moment_matching(match_encoder.y, noise)
  Sxz = match_encoder.cross_covariance(preinv=preinv) @ match_noise.cross_covariance(preinv=True)
  return chain, GaussianMatch(x=x, y=match_noise.y, cross=(Sxz, preinv))


class Euler:
  """``Euler`` (solvers.py:49-105): fold of ``step`` over the solution times with callbacks."""

  @classmethod
  def step(cls, func: Callable, t, dt, x: torch.Tensor) -> torch.Tensor:
    dx_dt, sqrt_cov = func(t, x)
    _x = x + dt * dx_dt
    if sqrt_cov is None:
      return _x
    rvs = torch.randn_like(_x)
    return _x + ((dt ** 0.5) * sqrt_cov @ rvs.unsqueeze(-1)).squeeze(-1)

  @classmethod
  def __call__(cls, func: Callable, initial_time: float, initial_state: Any,
               solution_times: Sequence[float],
               callbacks_and_initializers: Optional[List[Tuple[Callable, Any]]] = None,
               iterator: str = "scan"):
    """``iterator``: "scan" keeps every state (tf.scan), "foldl" only the last (tf.foldl)."""
    times = [float(t) for t in solution_times]
    steps = [times[0] - float(initial_time)] + [t1 - t0 for t0, t1 in zip(times[:-1], times[1:])]
    state = initial_state
    if callbacks_and_initializers is None:
      callbacks, cb_args = (), []
    else:
      callbacks, inits = zip(*callbacks_and_initializers)
      cb_args = list(inits)
    history = []
    for t, dt in zip(times, steps):
      state = cls.step(func=func, t=t, dt=dt, x=state)
      cb_args = [cb(t, state, arg) for cb, arg in zip(callbacks, cb_args)]
      if iterator == "scan":
        history.append((state,) + tuple(cb_args) if callbacks else state)
    if iterator == "scan":
      return history
    return (state,) + tuple(cb_args) if callbacks else state

  def __init__(self):
    pass


class MomentMatchingEuler(Euler):
  """``MomentMatchingEuler.step`` (solvers.py:108-135)."""

  @classmethod
  def step(cls, func: Callable, t, dt, x: Tuple[torch.Tensor, torch.Tensor]):
    x = GaussianMoments(moments=x, centered=True)
    match_drift, match_noise = func(t, x)
    mx, Sxx = x.mean(), x.covariance()
    mf = match_drift.y.mean()
    Sff = match_drift.y.covariance(dense=True)
    cross, is_preinv = match_drift.cross
    needs_grad = torch.is_grad_enabled() and any(
        isinstance(t, torch.Tensor) and t.requires_grad for t in (mx, Sxx, mf, Sff, cross))
    if (is_preinv and mx.is_cuda and mf.shape == mx.shape and Sff.shape == Sxx.shape
        and isinstance(cross, torch.Tensor) and mx.ndim == 2 and not needs_grad):
      _mx, _Sxx = ops.euler_update(mx, Sxx, mf, Sff, cross, dt)       # HIP k_euler
    else:
      Sxf = match_drift.cross_covariance()
      _mx = mx + dt * mf
      _Sxx = Sxx + dt * (Sxf + Sxf.transpose(-1, -2)) + (dt ** 2) * Sff
    if match_noise is not None:
      # the reference reads match_drift here (solvers.py:130-133); diffusion is asserted
      # None by the loops (loops/pilco.py:41-42), so this branch is dead code there too
      Sxz = match_noise.cross_covariance()
      Szz = match_noise.y.covariance(dense=True)
      _Sxx = _Sxx + (dt ** 0.5) * (Sxz + Sxz.transpose(-1, -2)) + dt * Szz
    return _mx, _Sxx


class DynamicalSystem:
  """``DynamicalSystem`` (dynamical_system.py:17-66)."""

  def __init__(self, drift, diffusion=None, policy=None, encoder=None, solver=None):
    self.drift = drift
    self.diffusion = diffusion
    self.policy = policy
    self.encoder = encoder
    self.solver = Euler() if solver is None else solver

  def forward(self, t, x):
    return forward_sde(x, self.drift, self.diffusion, self.policy, self.encoder)

  def solve_forward(self, initial_time, initial_state, solution_times, **kwargs):
    return self.solver(func=self.forward, initial_time=initial_time,
                       initial_state=initial_state, solution_times=solution_times, **kwargs)

  def solve_forward_closure(self, initial_time, state_initializer, solution_times, **kwargs):
    kwargs.pop("compile", None)

    def closure(state_initializer=state_initializer):
      return self.solve_forward(initial_time=initial_time, initial_state=state_initializer(),
                                solution_times=solution_times, **kwargs)
    return closure


def closed_rollout(model, mu: torch.Tensor, Sigma: torch.Tensor, num_steps: int, dt: float = 1.0,
                   model_uncertainty: bool = True, keep_trajectory: bool = False):
  """Fused drift-only rollout (state dim == d == L): every step's kernels are enqueued by one
  ``mm_rollout_closed`` call; same result as ``MomentMatchingEuler`` folded over
  ``forward_sde(x, drift, None, None, None)``."""
  pm = model.packed(dtype=mu.dtype, with_C=bool(model_uncertainty), device=mu.device)
  return ops.rollout_closed(pm, mu, Sigma, num_steps, dt=dt, model_uncertainty=model_uncertainty,
                            keep_trajectory=keep_trajectory)
