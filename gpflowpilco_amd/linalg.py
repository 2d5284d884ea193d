"""Small dense linear-algebra helpers of the model set-up (not on the per-step path)."""
from __future__ import annotations

import torch

__all__ = ("cholesky", "CholeskyError")


class CholeskyError(torch.linalg.LinAlgError):
  """Raised when a model set-up factorisation reports a non-positive pivot; carries the evidence."""


def _describe(A: torch.Tensor, info: torch.Tensor) -> str:
  """Facts about a failed factorisation, gathered with plain reductions (no second factorisation)."""
  A2 = A.reshape(-1, A.shape[-2], A.shape[-1])
  inf = info.reshape(-1)
  bad = int(torch.nonzero(inf)[0]) if bool((inf != 0).any()) else 0
  Ab = A2[bad]
  dg = torch.diagonal(Ab)
  asym = float((Ab - Ab.T).abs().max())
  stream = torch.cuda.current_stream(A.device).cuda_stream if A.is_cuda else None
  return (f"batch item {bad} of {A2.shape[0]}, n={Ab.shape[-1]}, info={int(inf[bad])} (leading minor that failed), "
          f"finite={bool(torch.isfinite(Ab).all())}, max|A-A^T|={asym:.3e}, diag in [{float(dg.min()):.6e}, {float(dg.max()):.6e}], "
          f"device={A.device}, stream={stream}")


_CAPTURE_STATUS = {}


def capture_status(device) -> torch.Tensor:
  """Device int32[1]: max |info| of every factorisation enqueued INSIDE a HIP-graph capture on ``device``
  (0 = all fine).  Updated by the graph itself on every replay; read it with ``check_capture_status``."""
  key = str(torch.device(device))
  t = _CAPTURE_STATUS.get(key)
  if t is None:
    t = torch.zeros(1, dtype=torch.int32, device=device)
    _CAPTURE_STATUS[key] = t
  return t


def check_capture_status(device):
  """Synchronising check of the in-graph factorisations (what the eager path raises immediately)."""
  t = _CAPTURE_STATUS.get(str(torch.device(device)))
  if t is not None and int(t.item()) != 0:
    v = int(t.item())
    t.zero_()
    raise CholeskyError(f"a Cholesky factorisation inside a replayed HIP graph failed (info={v}): its factor was "
                        "replaced by NaN, so every value computed from it in that replay is NaN")


def cholesky(A: torch.Tensor) -> torch.Tensor:
  """Lower Cholesky factor of the model set-up matrices (Kuu + jitter, K + s2 I); never on the per-step path.

  Eager: a device factorisation that reports ``info != 0`` RAISES ``CholeskyError`` with the failing item, the
  symmetry defect and the diagonal range in the message and, when ``GPFLOWPILCO_DUMP_DIR`` is set, the matrix
  saved there.  There is no host retry and no CPU fallback.

  Inside a HIP-graph capture (the trainable policy of ``loops.GraphedPolicyLoss`` is re-factorised from its
  parameters on every replay) the host cannot look at ``info``; the check moves onto the device instead of
  being skipped: a failed item's factor is overwritten with NaN -- it cannot pass as a plausible-looking
  garbage factor -- and ``info`` is max-ed into ``capture_status(device)``, which ``GraphedPolicyLoss`` checks
  after each replay (``check_capture_status``).
  """
  L, info = torch.linalg.cholesky_ex(A)
  if A.is_cuda and torch.cuda.is_current_stream_capturing():
    bad = info != 0
    L = torch.where(bad.reshape(bad.shape + (1, 1)), torch.full_like(L, float("nan")), L)
    st = capture_status(A.device)
    st.copy_(torch.maximum(st, info.abs().max().reshape(1).to(torch.int32)))
    return L
  if bool((info != 0).any()):
    msg = _describe(A, info)
    import os
    dump = os.environ.get("GPFLOWPILCO_DUMP_DIR")
    if dump:
      os.makedirs(dump, exist_ok=True)
      path = os.path.join(dump, f"cholesky_fail_pid{os.getpid()}.pt")
      torch.save({"A": A.detach().cpu(), "info": info.cpu()}, path)
      msg += f"; matrix saved to {path}"
    raise CholeskyError(f"Cholesky factorisation failed: the matrix is not positive definite ({msg})")
  return L


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
