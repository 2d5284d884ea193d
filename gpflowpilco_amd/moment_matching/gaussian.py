"""``GaussianMoments`` / ``GaussianMatch`` and the chain rule
(``gpflow_pilco/moment_matching/gaussian.py:23-83``), on torch tensors."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple, Union

import torch

from .core import (Chain, LinearOperatorDiag, MomentMatch, Moments, dispatcher,
                   moment_matching)

__all__ = ("GaussianMoments", "GaussianMatch")


class GaussianMoments(Moments):
  pass  # here for multiple dispatching


@dataclass
class GaussianMatch(MomentMatch):
  x: GaussianMoments
  y: GaussianMoments
  cross: Tuple[Union[torch.Tensor, LinearOperatorDiag], bool]

  def cross_covariance(self, dense: Optional[bool] = None, preinv: bool = False):
    """gaussian.py:33-51: convert between Cov(x,y) and Cov(x,x)^-1 Cov(x,y)."""
    Sxy, is_preinv = self.cross
    if not preinv and is_preinv:
      Sxx = self.x.covariance()
      Sxy = Sxx.matmul(Sxy if isinstance(Sxy, torch.Tensor) else Sxy.to_dense()) \
          if isinstance(Sxx, LinearOperatorDiag) else Sxx @ (Sxy if isinstance(Sxy, torch.Tensor) else Sxy.to_dense())
    elif preinv and not is_preinv:
      Sxx = self.x.covariance()
      rhs = Sxy.to_dense() if isinstance(Sxy, LinearOperatorDiag) else Sxy
      if isinstance(Sxx, LinearOperatorDiag):
        Sxy = Sxx.solve(rhs)
      else:
        Sxy = torch.cholesky_solve(rhs, torch.linalg.cholesky(Sxx))
    if dense and isinstance(Sxy, LinearOperatorDiag):
      Sxy = Sxy.to_dense()
    return Sxy

  def joint(self) -> GaussianMoments:
    """gaussian.py:53-63: Gaussian approximation of the joint of x and y."""
    m = torch.cat([self.x.mean(), self.y.mean()], dim=-1)
    Sxx = self.x.covariance(dense=True)
    Sxy = self.cross_covariance(dense=True, preinv=False)
    Syy = self.y.covariance(dense=True)
    S = torch.cat([torch.cat([Sxx, Sxy], dim=-1),
                   torch.cat([Sxy.transpose(-1, -2), Syy], dim=-1)], dim=-2)
    return GaussianMoments(moments=(m, S), centered=True)


@dispatcher.register(GaussianMoments, Chain)
def _mm_gauss_chain(x: GaussianMoments, chain: Chain):
  """gaussian.py:66-83: linearised propagation through a sequence of transformations."""
  state = x
  preinv = None
  cross_covariance = None
  for i, op in enumerate(reversed(chain)):
    match = moment_matching(state, op)
    state = match.y
    if i:
      cross_covariance = cross_covariance @ match.cross_covariance(preinv=True)
    else:
      cross_covariance, preinv = match.cross
  return GaussianMatch(x=x, y=state, cross=(cross_covariance, preinv))
