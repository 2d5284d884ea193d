"""Moment matching through an ``Encoder`` on a subset of dims
(``gpflow_pilco/moment_matching/components.py:19-57``)."""
from __future__ import annotations

import torch

from ..components import Encoder
from .core import dispatcher, moment_matching
from .gaussian import GaussianMatch, GaussianMoments


@dispatcher.register(GaussianMoments, Encoder)
def _mm_gauss_encoder(x: GaussianMoments, encoder: Encoder, append_inactive: bool = True) -> GaussianMatch:
  x1 = x.mean()
  active, inactive = encoder.get_partition_indices(ndims=x1.shape[-1])
  from ..linalg import index_tensor
  a, b = index_tensor(active, x1.device), index_tensor(inactive, x1.device)
  a1 = x1.index_select(-1, a)
  Sxx = x.covariance(dense=True)
  Sxa = Sxx.index_select(-1, a)
  Saa = Sxa.index_select(-2, a)
  match_part = moment_matching(GaussianMoments(moments=(a1, Saa), centered=True), encoder.transform)
  moments_y = match_part.y
  Sxy = Sxa @ match_part.cross_covariance(preinv=True)
  if append_inactive:
    y1 = torch.cat([moments_y.mean(), x1.index_select(-1, b)], dim=-1)
    Sxb = Sxx.index_select(-1, b)
    Sbb = Sxb.index_select(-2, b)
    Sby = Sxy.index_select(-2, b)
    Syy = moments_y.covariance(dense=True)
    Syy = torch.cat([torch.cat([Syy, Sby.transpose(-1, -2)], dim=-1),
                     torch.cat([Sby, Sbb], dim=-1)], dim=-2)
    moments_y = GaussianMoments(moments=(y1, Syy), centered=True)
    Sxy = torch.cat([Sxy, Sxb], dim=-1)
  return GaussianMatch(x=x, y=moments_y, cross=(Sxy, False))
