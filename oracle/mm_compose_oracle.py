"""fp64 CPU restatement of the rollout COMPOSITION around the GP moment match
(TEST INFRASTRUCTURE ONLY).

Follows, line by line:
  gpflow_pilco/moment_matching/maths.py:41-176        identity/add/sub/mul/matvec, sin, cos, sincos
  gpflow_pilco/moment_matching/components.py:19-57    Encoder on a subset of dims
  gpflow_pilco/moment_matching/bijectors.py:21-69     Chain / Shift / Scale / NormalCDF (1-D: Owen's T)
  gpflow_pilco/moment_matching/gaussian.py:27-83      GaussianMatch algebra, chain rule
  gpflow_pilco/moment_matching/models.py:27-41        InverseLinkWrapper, KernelRegressor
  gpflow_pilco/dynamics/forward_sde.py:95-137         encoder -> policy -> drift composition
  gpflow_pilco/components.py:26-37                    GaussianObjective expected cost
  gpflow_pilco/loops/pilco.py:192-220                 the policy-loss rollout harness

Third-party pieces: ``tensorflow_probability`` ``owens_t`` (bijectors.py:15,58) is taken from
``scipy.special.owens_t`` (same function); the n-D NormalCDF branch needs the Genz BVN of
``utils/bvn.py`` and is out of scope (cartpole's action is 1-D).

A "match" here is a dict: x=(m, S), y=(m1, m2, centered), cross=(array | ("diag", v), is_preinv).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import numpy as np
from scipy.special import erfc, owens_t

from oracle import mm_oracle as mo


# ---- carrier helpers --------------------------------------------------------
def covariance(mom):
  m1, m2, centered = mom
  return m2 if centered else m2 - m1[..., :, None] * m1[..., None, :]


def _dense(cross, nrows=None):
  if isinstance(cross, tuple) and cross[0] == "diag":
    v = cross[1]
    return v[..., :, None] * np.eye(v.shape[-1])
  return cross


def cross_covariance(match, preinv=False):
  """gaussian.py:33-51."""
  c, is_preinv = match["cross"]
  c = _dense(c)
  Sxx = covariance(match["x"])
  if not preinv and is_preinv:
    return Sxx @ c
  if preinv and not is_preinv:
    return np.linalg.solve(Sxx, c)
  return c


def joint(match):
  """gaussian.py:53-63."""
  mx = match["x"][0]; Sxx = covariance(match["x"])
  my = match["y"][0]; Syy = covariance(match["y"])
  Sxy = cross_covariance(match, preinv=False)
  m = np.concatenate([mx, my], -1)
  S = np.concatenate([np.concatenate([Sxx, Sxy], -1),
                      np.concatenate([np.swapaxes(Sxy, -1, -2), Syy], -1)], -2)
  return (m, S, True)


# ---- maths.py ------------------------------------------------------------------
def mm_add(x, c):
  """maths.py:48-52 (Shift)."""
  y = (x[0] + c, covariance(x), True)
  return dict(x=x, y=y, cross=(("diag", np.ones_like(x[0])), True))


def mm_mul(x, c):
  """maths.py:62-79 (Scale): second moment scales by c^2, centredness is inherited."""
  y = (c * x[0], (c ** 2) * x[1], x[2])
  return dict(x=x, y=y, cross=(("diag", np.full_like(x[0], c)), True))


def mm_sincos(x):
  """maths.py:143-176: y = [sin x, cos x], uncentred second moment, pre-inverted cross."""
  x1 = x[0]; Sxx = covariance(x)
  vx = np.diagonal(Sxx, axis1=-2, axis2=-1)
  vx_add = vx[..., :, None] + vx[..., None, :]
  S_add = Sxx + np.swapaxes(Sxx, -1, -2)
  A = np.exp(-0.5 * (vx_add + S_add))
  Bm = np.exp(-0.5 * (vx_add - S_add))
  A_cos = A * np.cos(x1[..., :, None] + x1[..., None, :])
  B_cos = Bm * np.cos(x1[..., :, None] - x1[..., None, :])
  evx = np.exp(-0.5 * vx)
  cx, sx = np.cos(x1), np.sin(x1)
  c1 = evx * cx; c2 = 0.5 * (B_cos + A_cos)
  s1 = evx * sx; s2 = 0.5 * (B_cos - A_cos)
  sc_outer = sx[..., :, None] * cx[..., None, :]
  sc = 0.5 * (sc_outer * (Bm + A) - np.swapaxes(sc_outer, -1, -2) * (Bm - A))
  y1 = np.concatenate([s1, c1], -1)
  y2 = np.concatenate([np.concatenate([s2, sc], -1),
                       np.concatenate([np.swapaxes(sc, -1, -2), c2], -1)], -2)
  n = x1.shape[-1]
  eye = np.eye(n)
  cross = np.concatenate([c1[..., :, None] * eye, -s1[..., :, None] * eye], -1)
  return dict(x=x, y=(y1, y2, False), cross=(cross, True))


# ---- components.py -------------------------------------------------------------
def partition_indices(ndims, active_dims):
  """components.py:58-66."""
  idx = tuple(range(ndims))
  active = tuple(idx[d] for d in active_dims)
  return active, tuple(sorted(set(idx) - set(active)))


def mm_encoder(x, active_dims, transform=mm_sincos, append_inactive=True):
  """moment_matching/components.py:19-57."""
  x1 = x[0]; Sxx = covariance(x)
  active, inactive = partition_indices(x1.shape[-1], active_dims)
  a1 = x1[..., list(active)]
  Sxa = Sxx[..., :, list(active)]
  Saa = Sxa[..., list(active), :]
  part = transform((a1, Saa, True))
  iSaa_Say = cross_covariance(part, preinv=True)
  Sxy = Sxa @ iSaa_Say
  ymom = part["y"]
  if append_inactive:
    b1 = x1[..., list(inactive)]
    y1 = np.concatenate([ymom[0], b1], -1)
    Sxb = Sxx[..., :, list(inactive)]
    Sbb = Sxb[..., list(inactive), :]
    Sby = Sxy[..., list(inactive), :]
    Syy = covariance(ymom)
    Syy = np.concatenate([np.concatenate([Syy, np.swapaxes(Sby, -1, -2)], -1),
                          np.concatenate([Sby, Sbb], -1)], -2)
    ymom = (y1, Syy, True)
    Sxy = np.concatenate([Sxy, Sxb], -1)
  return dict(x=x, y=ymom, cross=(Sxy, False))


# ---- bijectors.py ----------------------------------------------------------------
def ndtr(x):
  return 0.5 * erfc(-x / np.sqrt(2.0))                       # utils/bvn.py:38-42


def mm_ndtr(x):
  """bijectors.py:39-69, 1-D branch (Owen's T)."""
  x1 = x[0]; Sxx = covariance(x)
  if x1.shape[-1] != 1:
    raise NotImplementedError("n-D NormalCDF needs the Genz BVN (utils/bvn.py): out of scope")
  vx = np.diagonal(Sxx, axis1=-2, axis2=-1)
  vw = vx + 1.0
  isq_vw = 1.0 / np.sqrt(vw)
  z = isq_vw * x1
  y1 = ndtr(z)
  y2 = (y1 - 2.0 * owens_t(z, 1.0 / np.sqrt(1.0 + 2.0 * vx)))[..., None]   # [...,1,1], uncentred
  vxy = isq_vw * vx * (2.0 * np.pi) ** -0.5 * np.exp(-0.5 * z ** 2)
  return dict(x=x, y=(y1, y2, False), cross=(("diag", vxy / vx), True))


def mm_chain(x, ops: Sequence[Callable]):
  """gaussian.py:66-83; ``ops`` is in Chain order, applied right to left."""
  state = x
  cross = None; preinv = None
  for i, op in enumerate(reversed(list(ops))):
    match = op(state)
    state = match["y"]
    if i:
      cross = _dense(cross) @ cross_covariance(match, preinv=True)
    else:
      cross, preinv = match["cross"]
  return dict(x=x, y=state, cross=(cross, preinv))


# ---- models: policy = InverseLinkWrapper(KernelRegressor(SVGP), Chain[Scale, Shift, NormalCDF]) ---
def mm_svgp(x, model: mo.SVGPParams, model_uncertainty=True):
  f1, Sff, pre = mo.mm_gauss_svgp_mo(x[0], covariance(x), model, True, model_uncertainty, 0.0)
  return dict(x=x, y=(f1, Sff, True), cross=(pre, True))


def mm_policy(x, model: mo.SVGPParams, scale: float, shift: float):
  """models.py:27-41 + swingup_loops.py:85-91: u = scale * (Phi(f(e)) + shift), mean-only regressor."""
  ops = [lambda s: mm_mul(s, scale), lambda s: mm_add(s, shift), mm_ndtr,
         lambda s: mm_svgp(s, model, model_uncertainty=False)]
  return mm_chain(x, ops)


# ---- forward_sde.py:95-137 -------------------------------------------------------
def forward_sde_full(x, drift: mo.SVGPParams, policy_fn: Callable, active_dims):
  """x -> encoder -> policy -> drift with the cross-covariance bookkeeping. Returns the chained
  match (cross = Cov(x, f), not pre-inverted)."""
  match_encoder = mm_encoder(x, active_dims)
  match_policy = policy_fn(match_encoder["y"])
  match_drift = mm_svgp(joint(match_policy), drift, model_uncertainty=True)
  ndims_x = x[0].shape[-1]
  ndims_u = match_policy["y"][0].shape[-1]
  active, inactive = partition_indices(ndims_x, active_dims)
  ndims_b = ndims_x - len(active_dims)
  if match_encoder["cross"][1]:
    Sax = covariance(x)[..., list(active), :]
    Sae = Sax @ cross_covariance(match_encoder, preinv=True)
  else:
    Sae = cross_covariance(match_encoder)[..., list(active), :]
  Sau = Sae @ cross_covariance(match_policy, preinv=True)
  order = sorted(zip(active + inactive, range(ndims_x)))
  perm = [p for _, p in order]
  Sad = np.concatenate([Sae, Sau], -1)
  Sdd = covariance(match_drift["x"])
  nd = Sdd.shape[-2]
  Sbd = Sdd[..., nd - ndims_b - ndims_u: nd - ndims_u, :]
  Sxd = np.concatenate([Sad, Sbd], -2)[..., perm, :]
  Sxf = Sxd @ cross_covariance(match_drift, preinv=True)
  return dict(x=x, y=match_drift["y"], cross=(Sxf, False))


# ---- components.py:26-37 ---------------------------------------------------------
def expected_gaussian_cost(mean, cov, target, precis):
  d = mean.shape[-1]
  IpSW = np.eye(d) + cov @ precis
  iSpW = precis @ np.linalg.inv(IpSW)
  err = mean - target
  dist2 = np.einsum('...i,...ij,...j->...', err, iSpW, err)
  return -np.linalg.det(IpSW) ** -0.5 * np.exp(-0.5 * dist2)


# ---- loops/pilco.py:192-220 + solvers.py:67-135 ------------------------------------
def policy_rollout_loss(mu0, S0, drift, policy_fn, active_dims, target, precis, num_steps, dt=1.0,
                        keep=False):
  """Accumulated expected cost of the moment-matched rollout (one scalar per batch element)."""
  mu, S = mu0.copy(), S0.copy()
  loss = np.zeros(mu.shape[:-1])
  traj = []
  for _ in range(num_steps):
    x = (mu, S, True)
    m = forward_sde_full(x, drift, policy_fn, active_dims)
    Sxf = cross_covariance(m)
    mu, S = mo.euler_moment_update(mu, S, m["y"][0], covariance(m["y"]), Sxf, dt)
    enc = mm_encoder((mu, S, True), active_dims)["y"]
    loss = loss + expected_gaussian_cost(enc[0], covariance(enc), target, precis)
    if keep:
      traj.append((mu.copy(), S.copy()))
  return (loss, traj) if keep else loss
