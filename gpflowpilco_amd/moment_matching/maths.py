"""Moment matching of elementary maps (``gpflow_pilco/moment_matching/maths.py:24-176``).

Plain callables are dispatched through ``register_type`` proxies, as in the reference
(``tf.math.add`` -> ``torch.add`` etc.)."""
from __future__ import annotations

import torch

from ..components import sincos
from .core import LinearOperatorDiag, dispatcher, register_type
from .gaussian import GaussianMatch, GaussianMoments

NumericalTypes = (int, float, complex, torch.Tensor)

_type_identity = register_type(torch.clone, name="torch.identity")
_type_add = register_type(torch.add)
_type_sub = register_type(torch.sub)
_type_mul = register_type(torch.mul)
_type_matvec = register_type(torch.mv, name="torch.matvec")
_type_cos = register_type(torch.cos)
_type_sin = register_type(torch.sin)
_type_sincos = register_type(sincos)


def _outer(op, a, b):
  return op(a.unsqueeze(-1), b.unsqueeze(-2))


@dispatcher.register(GaussianMoments, _type_identity)
def _mm_gauss_identity(x: GaussianMoments, _, c=None):
  return GaussianMatch(x=x, y=x, cross=(LinearOperatorDiag.identity_like(x.mean()), True))


@dispatcher.register(GaussianMoments, _type_add, NumericalTypes)
def _mm_gauss_add(x: GaussianMoments, _, c):
  """maths.py:48-52."""
  y = GaussianMoments(moments=(x.mean() + c, x.covariance()), centered=True)
  return GaussianMatch(x=x, y=y, cross=(LinearOperatorDiag.identity_like(x.mean()), True))


@dispatcher.register(GaussianMoments, _type_sub, NumericalTypes)
def _mm_gauss_sub(x: GaussianMoments, _, c):
  """maths.py:55-59."""
  y = GaussianMoments(moments=(x.mean() - c, x.covariance()), centered=True)
  return GaussianMatch(x=x, y=y, cross=(LinearOperatorDiag.identity_like(x.mean()), True))


@dispatcher.register(GaussianMoments, _type_mul, NumericalTypes)
def _mm_gauss_mul(x: GaussianMoments, _, c):
  """maths.py:62-79."""
  if isinstance(c, torch.Tensor):              # Python scalars stay scalars (no host->device upload per call)
    c = c.to(dtype=x.dtype, device=x.mean().device)
  y2 = LinearOperatorDiag((c ** 2) * x[1].diag) if isinstance(x[1], LinearOperatorDiag) else (c ** 2) * x[1]
  y = GaussianMoments(moments=(c * x[0], y2), centered=x.centered)
  return GaussianMatch(x=x, y=y, cross=(LinearOperatorDiag.identity_like(x.mean(), c), True))


@dispatcher.register(GaussianMoments, _type_matvec, torch.Tensor)
def _mm_gauss_matvec(x: GaussianMoments, _, a: torch.Tensor, adjoint_a: bool = False):
  """maths.py:82-96."""
  A = a.transpose(-1, -2) if adjoint_a else a
  y1 = (A @ x[0].unsqueeze(-1)).squeeze(-1)
  y2 = A @ x[1] @ A.transpose(-1, -2)
  y = GaussianMoments(moments=(y1, y2), centered=x.centered)
  return GaussianMatch(x=x, y=y, cross=(A.transpose(-1, -2), True))


def _trig_terms(x: GaussianMoments):
  x1 = x.mean()
  Sxx = x.covariance(dense=True)
  vx = torch.diagonal(Sxx, dim1=-2, dim2=-1)
  vx_add = _outer(torch.add, vx, vx)
  S_add = Sxx + Sxx.transpose(-1, -2)
  A = torch.exp(-0.5 * (vx_add + S_add))
  B = torch.exp(-0.5 * (vx_add - S_add))
  A_cos = A * torch.cos(_outer(torch.add, x1, x1))
  B_cos = B * torch.cos(_outer(torch.sub, x1, x1))
  return x1, A, B, A_cos, B_cos, torch.exp(-0.5 * vx)


@dispatcher.register(GaussianMoments, _type_cos)
def _mm_gauss_cos(x: GaussianMoments, _):
  """maths.py:99-118."""
  x1, A, B, A_cos, B_cos, evx = _trig_terms(x)
  y = GaussianMoments(moments=(evx * torch.cos(x1), 0.5 * (B_cos + A_cos)), centered=False)
  return GaussianMatch(x=x, y=y, cross=(LinearOperatorDiag(-torch.sin(x1) * evx), True))


@dispatcher.register(GaussianMoments, _type_sin)
def _mm_gauss_sin(x: GaussianMoments, _):
  """maths.py:121-140."""
  x1, A, B, A_cos, B_cos, evx = _trig_terms(x)
  y = GaussianMoments(moments=(evx * torch.sin(x1), 0.5 * (B_cos - A_cos)), centered=False)
  return GaussianMatch(x=x, y=y, cross=(LinearOperatorDiag(torch.cos(x1) * evx), True))


@dispatcher.register(GaussianMoments, _type_sincos)
def _mm_gauss_sincos(x: GaussianMoments, _):
  """maths.py:143-176."""
  x1, A, B, A_cos, B_cos, evx = _trig_terms(x)
  cos_x1, sin_x1 = torch.cos(x1), torch.sin(x1)
  c1 = evx * cos_x1; c2 = 0.5 * (B_cos + A_cos)
  s1 = evx * sin_x1; s2 = 0.5 * (B_cos - A_cos)
  sc_outer = _outer(torch.mul, sin_x1, cos_x1)
  sc = 0.5 * (sc_outer * (B + A) - sc_outer.transpose(-1, -2) * (B - A))
  y1 = torch.cat([s1, c1], dim=-1)
  y2 = torch.cat([torch.cat([s2, sc], dim=-1), torch.cat([sc.transpose(-1, -2), c2], dim=-1)], dim=-2)
  cross = torch.cat([torch.diag_embed(c1), torch.diag_embed(-s1)], dim=-1)
  y = GaussianMoments(moments=(y1, y2), centered=False)
  return GaussianMatch(x=x, y=y, cross=(cross, True))
