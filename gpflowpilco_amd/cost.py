"""Closed-form expected saturating cost (``GaussianObjective``, gpflow_pilco/components.py:21-41).

The per-step cost statistic that the multi-GPU rollout all-gathers.
On the GPU it runs as one HIP kernel (``mm_expected_cost``); the torch expression below is
the host-side definition used for CPU tensors (policy evaluation off the hot path).
"""
from __future__ import annotations

import torch


def expected_gaussian_cost(mean: torch.Tensor, cov: torch.Tensor, target: torch.Tensor,
                           precis: torch.Tensor) -> torch.Tensor:
  """E_{x~N(mean,cov)}[-exp(-0.5 (x-x*)^T W (x-x*))]  (components.py:29-37) -> mean.shape[:-1]."""
  needs_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (mean, cov, target, precis))
  if mean.is_cuda and not needs_grad:     # the HIP kernel has no backward: differentiable calls use torch ops
    from . import ops
    return ops.expected_cost(mean, cov, target, precis)
  d = mean.shape[-1]
  eye = torch.eye(d, dtype=mean.dtype, device=mean.device)
  IpSW = eye + cov @ precis
  iSpW = precis @ torch.linalg.inv_ex(IpSW).inverse
  err = mean - target
  dist2 = (err * (iSpW @ err.unsqueeze(-1)).squeeze(-1)).sum(-1)
  return -torch.rsqrt(torch.linalg.det(IpSW)) * torch.exp(-0.5 * dist2)


class GaussianObjective:
  """``GaussianObjective`` (components.py:21-41)."""

  def __init__(self, target: torch.Tensor, precis: torch.Tensor):
    self.target = target
    self.precis = precis

  def __call__(self, x, t=None):
    from .moment_matching.core import Moments          # lazy: moment_matching imports components
    if isinstance(x, Moments):
      return expected_gaussian_cost(x.mean(), x.covariance(dense=True), self.target.to(x.mean()),
                                    self.precis.to(x.mean()))
    err = x - self.target
    dist2 = (err * (self.precis @ err.unsqueeze(-1)).squeeze(-1)).sum(-1)
    return -torch.exp(-0.5 * dist2)
