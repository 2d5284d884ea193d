"""``gpflow_pilco/components.py:21-75`` on torch: GaussianObjective, Encoder, TrigonometricEncoder."""
from __future__ import annotations

from typing import Callable, Tuple

import torch

from .cost import GaussianObjective  # noqa: F401  (components.py:21-41)

__all__ = ("GaussianObjective", "Encoder", "TrigonometricEncoder", "sincos")


def sincos(x: torch.Tensor, axis: int = -1) -> torch.Tensor:
  """moment_matching/maths.py:24-25."""
  return torch.cat([torch.sin(x), torch.cos(x)], dim=axis)


class Encoder:
  """components.py:44-70."""

  def __init__(self, transform: Callable, active_dims: Tuple[int, ...]):
    self._transform = transform
    self.active_dims = tuple(active_dims)

  def __call__(self, x: torch.Tensor, append_inactive: bool = True) -> torch.Tensor:
    active, inactive = self.get_partition_indices(ndims=x.shape[-1])
    ret = self.transform(x[..., list(active)])
    if append_inactive and len(inactive):
      ret = torch.cat([ret, x[..., list(inactive)]], dim=-1)
    return ret

  def get_partition_indices(self, ndims: int):
    idx = tuple(range(ndims))
    active = tuple(idx[d] for d in self.active_dims)
    assert len(active) == len(set(active))
    return active, tuple(sorted(set(idx) - set(active)))

  @property
  def transform(self):
    return self._transform


class TrigonometricEncoder(Encoder):
  """components.py:73-75."""

  def __init__(self, active_dims: Tuple[int, ...]):
    super().__init__(transform=sincos, active_dims=active_dims)
