"""Pathwise (decoupled-sampling) GP paths and sample rollouts -- SURVEY.md row f-3.

Mirrors the surface the reference uses from ``gpflow_sampling`` (un-vendored third party):
``PathwiseSVGP.generate_paths(num_samples, num_bases, sample_axis=0)``,
``set_temporary_paths`` and ``predict_f_samples`` / ``__call__``
(``gpflow_pilco/models/svgp.py:124-130``, ``loops/pilco.py:263-303``).  Path *generation*
(sampling weights, one Cholesky solve per latent) is torch plumbing done once per rollout
closure; path *evaluation* -- the per-step hot loop -- runs in ``mm_pathwise_eval`` /
``mm_pathwise_rollout`` (``csrc/mm_pathwise.hip``).  Parity with gpflow_sampling is unpinned
(see ``oracle/pathwise_oracle.py``).
"""
from __future__ import annotations

import contextlib
import math
from dataclasses import dataclass
from typing import Optional

import torch

from .linalg import cholesky

from . import _lib
from .models import DEFAULT_FLOAT, DEFAULT_JITTER, SVGP, Constant, Zero, _stack_kernel_params, unpack_multioutput
from .ops import _dtype_code, _ptr, _require_device, _stream, check


def _pad_last(t: torch.Tensor, mult: int) -> torch.Tensor:
  n = t.shape[-1]
  pad = (-n) % mult
  return t if pad == 0 else torch.nn.functional.pad(t, (0, pad))


@dataclass
class Paths:
  """S sample paths of L latent GPs (device tensors, element type ``dtype``)."""
  omega: torch.Tensor      # [L, d, Kp]   omega^T / 2 pi  (revolutions, k-major)
  phase: torch.Tensor      # [L, Kp]      b / 2 pi
  zs: torch.Tensor         # [L, d, Mp]   (Z * x_scale)^T, x_scale = sqrt(log2 e) / lengthscales
  hz: torch.Tensor         # [L, Mp]      |zs|^2 / 2
  wb: torch.Tensor         # [G, L, NB, 4, BT] blocked weight stream (prior blocks, then update blocks)
  num_samples: int
  lengthscales: torch.Tensor   # [L, d] f64: x_scale = sqrt(log2 e) / lengthscales
  prior_scale: torch.Tensor    # [L] f64  sqrt(2 var / K)
  variance: torch.Tensor       # [L] f64
  mean_c: Optional[torch.Tensor]  # [L] f64 or None

  @property
  def dtype(self):
    return self.wb.dtype

  def _dims(self):
    L, d, Kp = self.omega.shape
    return self.num_samples, L, self.zs.shape[-1], Kp, d

  def __call__(self, x: torch.Tensor) -> torch.Tensor:
    """f_s(x_s): x [S, d] -> [S, L]."""
    _require_device(x, self.wb)
    S, L, Mp, Kp, d = self._dims()
    if x.shape != (S, d) or x.dtype != self.dtype:
      raise ValueError(f"expected x [{S},{d}] of {self.dtype}, got {tuple(x.shape)} {x.dtype}")
    x = x.contiguous()
    out = torch.empty(S, L, dtype=self.dtype, device=x.device)
    rc = _lib.lib().mm_pathwise_eval(S, L, Mp, Kp, d, _dtype_code(self.dtype), x.data_ptr(), self.omega.data_ptr(),
                                     self.phase.data_ptr(), self.zs.data_ptr(), self.hz.data_ptr(),
                                     self.lengthscales.data_ptr(), self.prior_scale.data_ptr(),
                                     self.variance.data_ptr(), _ptr(self.mean_c), self.wb.data_ptr(),
                                     out.data_ptr(), _stream(x.device))
    check(rc, "mm_pathwise_eval")
    return out

  def rollout(self, x0: torch.Tensor, num_steps: int, dt: float = 1.0, keep_trajectory: bool = False):
    """Drift-only Euler rollout of all S paths (d == L): x <- x + dt f(x), H steps in one ABI call."""
    _require_device(x0, self.wb)
    S, L, Mp, Kp, d = self._dims()
    x = x0.contiguous().clone()
    tmp = torch.empty_like(x)
    traj = torch.empty(num_steps, S, d, dtype=self.dtype, device=x.device) if keep_trajectory else None
    rc = _lib.lib().mm_pathwise_rollout(S, L, Mp, Kp, d, _dtype_code(self.dtype), int(num_steps), float(dt),
                                        x.data_ptr(), tmp.data_ptr(), self.omega.data_ptr(), self.phase.data_ptr(),
                                        self.zs.data_ptr(), self.hz.data_ptr(), self.lengthscales.data_ptr(),
                                        self.prior_scale.data_ptr(), self.variance.data_ptr(), _ptr(self.mean_c),
                                        self.wb.data_ptr(), _ptr(traj), _stream(x.device))
    check(rc, "mm_pathwise_rollout")
    return (x, traj) if keep_trajectory else x


def paths_from_arrays(omega, phase, w, v, Z, lengthscales, variance, mean_c=None, dtype=torch.float32,
                      device="cuda") -> Paths:
  """Build ``Paths`` from explicit arrays (omega [L,K,d], phase [L,K], w [S,L,K], v [S,L,M], Z [L,M,d])."""
  t64 = lambda a: torch.as_tensor(a, dtype=DEFAULT_FLOAT, device=device)
  mult = 128 if dtype == torch.float64 else 256          # BT: terms per block of the weight stream
  xscale = math.sqrt(math.log2(math.e)) / t64(lengthscales)                    # [L, d]
  zs = t64(Z) * xscale[:, None, :]
  hz = 0.5 * (zs * zs).sum(-1)
  K = omega.shape[1]
  tt = lambda a: a.to(dtype).contiguous()
  padk = lambda a: tt(_pad_last(t64(a), mult))
  two_pi = 2.0 * math.pi
  omega_p = tt(_pad_last(t64(omega).transpose(1, 2) / two_pi, mult))           # [L, d, Kp]
  zs_p = tt(_pad_last(zs.transpose(1, 2), mult))                                # [L, d, Mp]
  var = t64(variance)
  # blocked weight stream [G, L, NB, 4, BT]
  wp, vp = _pad_last(t64(w), mult), _pad_last(t64(v), mult)
  S, L = wp.shape[:2]
  G = (S + 3) // 4
  allw = torch.cat([wp, vp], dim=-1)                                              # [S, L, Kp + Mp]
  if G * 4 != S:
    allw = torch.cat([allw, torch.zeros(G * 4 - S, L, allw.shape[-1], dtype=DEFAULT_FLOAT, device=device)], 0)
  NB = allw.shape[-1] // mult
  wb = allw.reshape(G, 4, L, NB, mult).permute(0, 2, 3, 1, 4).to(dtype).contiguous()
  return Paths(omega=omega_p, phase=padk(t64(phase) / two_pi), zs=zs_p, hz=padk(hz), wb=wb, num_samples=S,
               lengthscales=xscale.contiguous(), prior_scale=torch.sqrt(2.0 * var / K).contiguous(),
               variance=var.contiguous(), mean_c=None if mean_c is None else t64(mean_c).contiguous())


def generate_paths(model: SVGP, num_samples: int, num_bases: int = 1024, dtype=torch.float32,
                   device="cuda", generator: Optional[torch.Generator] = None) -> Paths:
  """Draw S decoupled sample paths of an SVGP: random-Fourier prior + inducing-point update
  (gpflow_sampling's decoupled sampler; ``loops/pilco.py:281-284``)."""
  kernels, Zs = unpack_multioutput(model.kernel, model.inducing_variable, model.num_latent_gps)
  Z, ls, var = _stack_kernel_params(kernels, Zs, device)
  L, M, d = Z.shape
  S, K = num_samples, num_bases
  rn = lambda *shape: torch.randn(*shape, dtype=DEFAULT_FLOAT, device=device, generator=generator)
  omega = rn(L, K, d) / ls[:, None, :]
  phase = 2.0 * math.pi * torch.rand(L, K, dtype=DEFAULT_FLOAT, device=device, generator=generator)
  w = rn(S, L, K)
  A = Z / ls[:, None, :]
  d2 = (A * A).sum(-1)[:, :, None] + (A * A).sum(-1)[:, None, :] - 2.0 * A @ A.transpose(1, 2)
  Kuu = var[:, None, None] * torch.exp(-0.5 * d2.clamp_min(0.0)) + DEFAULT_JITTER * torch.eye(M, dtype=DEFAULT_FLOAT, device=device)
  Luu = cholesky(Kuu)
  q_mu = model.q_mu.to(device=device, dtype=DEFAULT_FLOAT).T                       # [L, M]
  q_sqrt = torch.tril(model.q_sqrt.to(device=device, dtype=DEFAULT_FLOAT))          # [L, M, M]
  eps = rn(S, L, M)
  u = q_mu[None] + torch.einsum('slm,lnm->sln', eps, q_sqrt)                       # samples of q(u)
  if model.whiten:
    u = torch.einsum('lnm,slm->sln', Luu, u)
  Phi_Z = torch.sqrt(2.0 * var / K)[:, None, None] * torch.cos(Z @ omega.transpose(1, 2) + phase[:, None, :])  # [L,M,K]
  resid = u - torch.einsum('lmk,slk->slm', Phi_Z, w)
  v = torch.cholesky_solve(resid.permute(1, 2, 0), Luu).permute(2, 0, 1)             # [S, L, M]
  mean_c = None
  if isinstance(model.mean_function, Constant):
    mean_c = model.mean_function.c.to(device=device, dtype=DEFAULT_FLOAT).expand(L).contiguous()
  elif not isinstance(model.mean_function, Zero):
    raise NotImplementedError
  return paths_from_arrays(omega, phase, w, v, Z, ls, var, mean_c, dtype=dtype, device=device)


class PathwiseSVGP(SVGP):
  """``gpflow_pilco.models.PathwiseSVGP`` (models/svgp.py:124-130): an SVGP whose ``__call__``
  evaluates the currently attached sample paths."""

  _paths: Optional[Paths] = None

  def generate_paths(self, num_samples: int, num_bases: int = 1024, sample_axis: int = 0,
                     dtype=torch.float32, device="cuda", generator=None) -> Paths:
    assert sample_axis == 0
    return generate_paths(self, num_samples, num_bases, dtype=dtype, device=device, generator=generator)

  @contextlib.contextmanager
  def set_temporary_paths(self, paths: Paths):
    prev, self._paths = self._paths, paths
    try:
      yield
    finally:
      self._paths = prev

  def predict_f_samples(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
    if self._paths is None:
      raise RuntimeError("no sample paths attached: use generate_paths / set_temporary_paths")
    return self._paths(x)

  def __call__(self, x, **kwargs):
    return self.predict_f_samples(x, **kwargs)
