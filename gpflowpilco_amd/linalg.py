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
  if A.is_cuda and torch.cuda.is_current_stream_capturing():
    return L                                 # inside a HIP-graph capture no host check is possible
  if not bool((info != 0).any()):
    return L
  if A.is_cuda:
    Lh, info_h = torch.linalg.cholesky_ex(A.cpu())
    if not bool((info_h != 0).any()):
      return Lh.to(A.device)
  return torch.linalg.cholesky(A)          # raises torch.linalg.LinAlgError with the usual message


_INDEX_CACHE = {}


def index_tensor(indices, device) -> "torch.Tensor":
  """Device index tensor for a (short) Python index list, cached: ``x[..., [1, 3]]`` builds and uploads
  an index tensor on every call (a host->device copy and a stream synchronisation per use)."""
  key = (tuple(int(i) for i in indices), str(device))
  t = _INDEX_CACHE.get(key)
  if t is None:
    t = torch.tensor(key[0], dtype=torch.long, device=device)
    _INDEX_CACHE[key] = t
  return t
