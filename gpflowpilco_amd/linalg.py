"""Small dense linear-algebra helpers of the model set-up (not on the per-step path)."""
from __future__ import annotations

import torch

__all__ = ("cholesky",)


def cholesky(A: torch.Tensor) -> torch.Tensor:
  """Lower Cholesky factor; raises like ``torch.linalg.cholesky`` if A is not positive definite.

  Kuu + jitter I has cond ~ 1e9 at the reference's default jitter, and the device factorisation
  (rocSOLVER potrf) has been seen to report a non-positive pivot on such a matrix when two processes
  time-slice one GPU while the same matrix factorises elsewhere.  A failed device factorisation is
  therefore retried once on the host (LAPACK, same dtype) before the error is raised.
  """
  L, info = torch.linalg.cholesky_ex(A)
  if not bool((info != 0).any()):
    return L
  if A.is_cuda:
    Lh, info_h = torch.linalg.cholesky_ex(A.cpu())
    if not bool((info_h != 0).any()):
      return Lh.to(A.device)
  return torch.linalg.cholesky(A)          # raises torch.linalg.LinAlgError with the usual message
