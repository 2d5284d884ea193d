"""Special functions used by the NormalCDF moment match (moment_matching/bijectors.py:39-69):
``ndtr`` (utils/bvn.py:38-42) and Owen's T (tensorflow_probability ``owens_t``, third party)."""
from __future__ import annotations

import math

import numpy as np
import torch

_GL_X, _GL_W = np.polynomial.legendre.leggauss(48)


_GL_CACHE = {}


def ndtr(x: torch.Tensor) -> torch.Tensor:
  return 0.5 * torch.erfc(-x / math.sqrt(2.0))


def owens_t(h: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
  """T(h, a) = 1/(2 pi) int_0^a exp(-h^2 (1 + x^2) / 2) / (1 + x^2) dx for 0 <= a <= 1
  (the only range bijectors.py:58 needs: a = rsqrt(1 + 2 v)), by 48-point Gauss-Legendre."""
  key = (h.dtype, str(h.device))
  if key not in _GL_CACHE:                                     # uploaded once per (dtype, device)
    _GL_CACHE[key] = (torch.as_tensor(_GL_X, dtype=h.dtype, device=h.device),
                      torch.as_tensor(_GL_W, dtype=h.dtype, device=h.device))
  xs, ws = _GL_CACHE[key]
  t = 0.5 * a.unsqueeze(-1) * (xs + 1.0)                       # nodes on [0, a]
  f = torch.exp(-0.5 * h.unsqueeze(-1) ** 2 * (1.0 + t * t)) / (1.0 + t * t)
  return (0.5 * a) * (f * ws).sum(-1) / (2.0 * math.pi)
