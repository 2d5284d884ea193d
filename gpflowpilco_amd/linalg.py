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
    raise CholeskyError(f"a Cholesky factorisation inside a replayed HIP graph failed (info={v}; n + 1 = the factor did not "
                        "reproduce its matrix on the probe vectors): its factor was replaced by NaN, so every value computed "
                        "from it in that replay is NaN")


# rocSOLVER's BLOCKED potrf (what torch.linalg.cholesky_ex reaches for n beyond a few hundred) is not safe when a
# second process uses the same GPU: tools/potrf_probe.py (profiles/r02_potrf_probe.txt) factorises one fixed
# K + 1e-6 I (n = 2000; host LAPACK and a lone GPU process factorise it to 1e-11) from two processes on one device
# and gets a failed minor at a random position in ~10 % of the calls -- on the default stream, on a side stream and
# with device synchronisations around the call alike -- and once a factor that is silently wrong by 5e-4.  The same
# factorisation composed of the UNBLOCKED kernel on <= 256-wide diagonal blocks + triangular solves + GEMMs never
# failed there.  That composition is what the set-up uses; its result is verified (below) instead of trusted.
_BLOCK = 256


def _blocked_cholesky_ex(A: torch.Tensor, nb: int = _BLOCK):
  """Right-looking blocked Cholesky out of small ``cholesky_ex`` calls, ``solve_triangular`` and matmuls (any
  leading batch dims).  -> (L, info) with info = 1-based position of the first failed minor (0 = none)."""
  n = A.shape[-1]
  L = A.clone()
  info = torch.zeros(A.shape[:-2], dtype=torch.int32, device=A.device)
  for j in range(0, n, nb):
    e = min(j + nb, n)
    Ljj, inf = torch.linalg.cholesky_ex(L[..., j:e, j:e])
    info = torch.where((info == 0) & (inf != 0), inf + j, info)
    L[..., j:e, j:e] = Ljj
    if e < n:
      P = torch.linalg.solve_triangular(Ljj, L[..., e:, j:e].transpose(-1, -2), upper=False).transpose(-1, -2)
      L[..., e:, j:e] = P
      L[..., e:, e:] -= P @ P.transpose(-1, -2)
  return torch.tril(L), info


def _residual(A: torch.Tensor, L: torch.Tensor) -> float:
  """max |L (L^T v) - A v| / (n |A|_max |v|_max) on two fixed probe vectors: O(n^2), catches a silently wrong factor."""
  n = A.shape[-1]
  g = torch.Generator(device="cpu").manual_seed(1234)
  v = torch.randn(n, 2, generator=g, dtype=A.dtype).to(A.device)
  r = L @ (L.transpose(-1, -2) @ v) - A @ v
  return float(r.abs().max() / (n * A.abs().max() * v.abs().max()))


def cholesky(A: torch.Tensor) -> torch.Tensor:
  """Lower Cholesky factor of the model set-up matrices (Kuu + jitter, K + s2 I); never on the per-step path.

  Eager: a factorisation that reports ``info != 0`` -- or whose factor does not reproduce A on two probe vectors --
  RAISES ``CholeskyError`` with the failing item, the symmetry defect and the diagonal range in the message and,
  when ``GPFLOWPILCO_DUMP_DIR`` is set, the matrix saved there.  There is no retry and no CPU fallback.

  Inside a HIP-graph capture (the trainable policy of ``loops.GraphedPolicyLoss`` is re-factorised from its
  parameters on every replay) the host cannot look at ``info``; the check moves onto the device instead of
  being skipped: a failed item's factor is overwritten with NaN -- it cannot pass as a plausible-looking
  garbage factor -- and ``info`` is max-ed into ``capture_status(device)``, which ``GraphedPolicyLoss`` checks
  after each replay (``check_capture_status``).
  """
  big = A.is_cuda and A.shape[-1] > _BLOCK
  L, info = _blocked_cholesky_ex(A) if big else torch.linalg.cholesky_ex(A)
  if A.is_cuda and torch.cuda.is_current_stream_capturing():
    bad = info != 0
    code = info.abs().max().reshape(1).to(torch.int32)
    if A.shape[-1] > 64:
      # the probe-residual guard of the eager path, on the device: a factor that is silently wrong with info = 0 (seen
      # once, tools/potrf_probe.py) is poisoned too, and recorded as code n + 1
      n = A.shape[-1]
      t = torch.arange(1, n + 1, dtype=A.dtype, device=A.device)
      v = torch.stack([torch.cos(0.7 * t), torch.sin(1.3 * t) + 0.5], dim=-1)
      r = L @ (L.transpose(-1, -2) @ v) - A @ v
      res = r.abs().amax(dim=(-1, -2)) / (n * A.abs().amax(dim=(-1, -2)) * v.abs().max())
      wrong = ~(res < 1e-9)                                  # also true for NaN
      bad = bad | wrong
      code = torch.maximum(code, (wrong.any().to(torch.int32) * (n + 1)).reshape(1))
    L = torch.where(bad.reshape(bad.shape + (1, 1)), torch.full_like(L, float("nan")), L)
    st = capture_status(A.device)
    st.copy_(torch.maximum(st, code))
    return L
  failed = bool((info != 0).any())
  msg = None
  if failed:
    msg = _describe(A, info)
  elif A.is_cuda and A.shape[-1] > 64:
    res = _residual(A, L)
    if not res < 1e-9:                                       # also catches NaN
      failed = True
      msg = f"info = 0 but the factor does not reproduce the matrix: probe residual {res:.3e} (n={A.shape[-1]}, device={A.device})"
  if failed:
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
