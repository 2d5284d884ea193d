"""Moment matching through the policy head's bijectors
(``gpflow_pilco/moment_matching/bijectors.py:21-69``)."""
from __future__ import annotations

import math

import torch

from .. import bijectors as tfb
from ..special import ndtr, owens_t
from .core import Chain, LinearOperatorDiag, Moments, dispatcher, moment_matching
from .gaussian import GaussianMatch, GaussianMoments


@dispatcher.register(Moments, tfb.Chain)
def _mm_chain(x: Moments, bijector: tfb.Chain, /, **kwargs):
  return moment_matching(x, Chain(*bijector.bijectors), **kwargs)


@dispatcher.register(Moments, tfb.Shift)
def _mm_shift(x: Moments, bijector: tfb.Shift, /, **kwargs):
  return moment_matching(x, torch.add, bijector.shift, **kwargs)


@dispatcher.register(Moments, tfb.Scale)
def _mm_scale(x: Moments, bijector: tfb.Scale, /, **kwargs):
  return moment_matching(x, torch.mul, bijector.scale, **kwargs)


@dispatcher.register(GaussianMoments, tfb.NormalCDF)
def _mm_gauss_ndtr(x: GaussianMoments, _):
  """bijectors.py:39-69.  E[Phi(x_i) Phi(x_j)] = P(w_i <= 0, w_j <= 0), w = z - x.

  Only the 1-D branch (Owen's T, :57-58) is built; the n-D one needs the Genz BVN of
  ``utils/bvn.py`` (out of scope).  The reference returns the 1-D
  second moment with shape [N, 1], which is only consistent for N == 1 (it always uses one
  input distribution); here it is [N, 1, 1] so that batches work."""
  x1 = x.mean()
  Sxx = x.covariance(dense=True)
  if x.ndim != 1:
    raise NotImplementedError("NormalCDF moment matching of an n-D input needs the bivariate normal CDF")
  vx = torch.diagonal(Sxx, dim1=-2, dim2=-1)
  isq_vw = torch.rsqrt(vx + 1.0)
  z = isq_vw * x1
  y1 = ndtr(z)
  y2 = (y1 - 2.0 * owens_t(z, torch.rsqrt(1.0 + 2.0 * vx))).unsqueeze(-1)
  vxy = isq_vw * vx * ((2.0 * math.pi) ** -0.5) * torch.exp(-0.5 * z * z)
  y = GaussianMoments(moments=(y1, y2), centered=False)
  return GaussianMatch(x=x, y=y, cross=(LinearOperatorDiag(vxy / vx), True))
