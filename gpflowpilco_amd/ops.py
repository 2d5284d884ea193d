"""torch-tensor front end of the C ABI (device pointers + current HIP stream).

PyTorch is plumbing here: it owns the device buffers and the stream; every
floating-point operation of the moment match happens in the HIP kernels of
``csrc/``.  Nothing in this module runs on the CPU -- tensors must live on a
ROCm device.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import (MM_F32, MM_F64, MM_FORCE_GENERIC, MM_FULL_OUTPUT_COV, MM_WORKSPACE_CURRENT, MM_SUMS_CURRENT,
                   MM_MODEL_UNCERTAINTY, check, lib)

_DTYPES = {torch.float32: MM_F32, torch.float64: MM_F64}


def _dtype_code(dtype: torch.dtype) -> int:
  try:
    return _DTYPES[dtype]
  except KeyError:
    raise TypeError(f"moment matching supports float32/float64 states, got {dtype}") from None


def _ptr(t: Optional[torch.Tensor]):
  return None if t is None else t.data_ptr()


def _require_device(*tensors: torch.Tensor):
  for t in tensors:
    if t is not None and not t.is_cuda:
      raise RuntimeError("gpflowpilco_amd kernels run on the GPU only (no CPU fallback): "
                         f"got a tensor on {t.device}")


def _stream(device) -> int:
  return torch.cuda.current_stream(device).cuda_stream


def make_flags(full_output_cov: bool = True, model_uncertainty: bool = True,
               force_generic: bool = False) -> int:
  return ((MM_FULL_OUTPUT_COV if full_output_cov else 0)
          | (MM_MODEL_UNCERTAINTY if model_uncertainty else 0)
          | (MM_FORCE_GENERIC if force_generic else 0))


@dataclass
class PackedModel:
  """Device-resident packed model (output of ``mm_pack_model``)."""
  L: int
  M: int
  d: int
  dtype: torch.dtype
  with_C: bool
  buf: torch.Tensor                      # uint8 [packed_bytes]
  _workspaces: Dict[Tuple[int, int], torch.Tensor] = field(default_factory=dict, repr=False)
  _ws_gen: Dict[Tuple[int, int], int] = field(default_factory=dict, repr=False)
  _status: Optional[torch.Tensor] = field(default=None, repr=False)
  _keep: Optional[tuple] = field(default=None, repr=False)
  _perm: Optional[torch.Tensor] = field(default=None, repr=False)

  @property
  def device(self):
    return self.buf.device

  @property
  def nbytes(self) -> int:
    return self.buf.numel()

  def workspace(self, B: int, flags: int, peek: bool = False) -> torch.Tensor:
    """The per-(B, flags) scratch buffer of the kernels.  Every request counts as a use (``workspace_generation``):
    whoever asks may overwrite it; ``peek`` looks without counting."""
    key = (B, flags & (MM_FULL_OUTPUT_COV | MM_MODEL_UNCERTAINTY))
    if not peek:
      self._ws_gen[key] = self._ws_gen.get(key, 0) + 1
    ws = self._workspaces.get(key)
    if ws is None:
      n = lib().mm_workspace_bytes(B, self.L, self.M, self.d, _dtype_code(self.dtype), flags)
      if n == 0:
        raise ValueError("mm_workspace_bytes rejected the shape")
      ws = torch.empty(n, dtype=torch.uint8, device=self.device)
      self._workspaces[key] = ws
    return ws

  def workspace_generation(self, B: int, flags: int) -> int:
    """Counter of the requests for this workspace: unchanged since a forward <=> that forward's q stage is still on it."""
    return self._ws_gen.get((B, flags & (MM_FULL_OUTPUT_COV | MM_MODEL_UNCERTAINTY)), 0)

  def perm(self) -> torch.Tensor:
    """[L, M] int64: the caller's index of the inducing point at packed position m (``mm_pack_perm``: packs of M > 256 points
    are sorted per latent by |(z - mean z) / lengthscale|).  ``q_forward``'s q is in the caller's order; the per-point sums of
    ``mm_backward_sums`` are in packed order."""
    if self._perm is None:
      p32 = torch.empty(self.L, self.M, dtype=torch.int32, device=self.device)
      check(lib().mm_pack_perm(self.buf.data_ptr(), self.nbytes, self.L, self.M, self.d, _dtype_code(self.dtype),
                               p32.data_ptr(), _stream(self.device)), "mm_pack_perm")
      self._perm = p32.long()
    return self._perm

  def status(self) -> torch.Tensor:
    if self._status is None:
      self._status = torch.zeros(4, dtype=torch.int32, device=self.device)
    return self._status

  def routed(self) -> Tuple[int, int]:
    """(forward, backward) counts of (batch element, off-diagonal pair) items a float32 pack re-reduced in float64 since the
    status word was last zeroed (csrc/mm_route.hip: items whose estimated f32 rounding error exceeded MM_ROUTE_TOL of the
    covariance block's scale).  Synchronises."""
    st = self.status().tolist()
    return st[2], st[3]

  def check_status(self, B: int):
    """Synchronising check of the non-PD flag (the reference raises at this point)."""
    st = self.status().tolist()
    if st[0] != 0 and st[1] == -1:
      self._status.zero_()
      raise RuntimeError(
          f"moment_match_backward was told the workspace still holds the forward's q stage (MM_WORKSPACE_CURRENT), but batch "
          f"element {B - st[0]} of it belongs to another state: something wrote the workspace in between without going "
          "through PackedModel.workspace() (a graph replay, another stream, an external ABI caller)")
    if st[0] != 0 and st[1] == -2:
      self._status.zero_()
      raise RuntimeError(
          f"moment_match_backward was given kept sums (MM_SUMS_CURRENT) that were swept for another state: batch element "
          f"{B - st[0]} of `sums` does not belong to this (mu, Sigma)")
    if st[0] != 0:
      self._status.zero_()
      raise FloatingPointError(
          f"Cholesky of (Sigma + V) failed for batch element {B - st[0]} (item {st[1]}): "
          "the input covariance is not positive definite")


def pack_model(Z: torch.Tensor, lengthscales: torch.Tensor, variance: torch.Tensor,
               beta: torch.Tensor, C: Optional[torch.Tensor] = None,
               mean_c: Optional[torch.Tensor] = None,
               dtype: torch.dtype = torch.float32, sync: bool = True) -> PackedModel:
  """Z [L,M,d], lengthscales [L,d], variance [L], beta [L,M], C [L,M,M]|None, mean_c [L]|None
  (all float64 on the GPU) -> PackedModel whose reduce operands are stored as ``dtype``.
  ``sync=False`` (graph capture): no stream synchronisation; the f64 inputs are kept alive on the returned object."""
  _require_device(Z, lengthscales, variance, beta, C, mean_c)
  L, M, d = Z.shape
  f64 = lambda t: None if t is None else t.to(torch.float64).contiguous()
  Z, lengthscales, variance, beta, C, mean_c = map(f64, (Z, lengthscales, variance, beta, C, mean_c))
  assert lengthscales.shape == (L, d) and variance.shape == (L,) and beta.shape == (L, M)
  assert C is None or C.shape == (L, M, M)
  assert mean_c is None or mean_c.shape == (L,)
  code = _dtype_code(dtype)
  n = lib().mm_packed_model_bytes(L, M, d, code, int(C is not None))
  if n == 0:
    raise ValueError(f"unsupported model shape L={L} M={M} d={d} (d <= {_lib.MM_DMAX})")
  buf = torch.empty(n, dtype=torch.uint8, device=Z.device)
  rc = lib().mm_pack_model(buf.data_ptr(), n, L, M, d, code, _ptr(Z), _ptr(lengthscales),
                           _ptr(variance), _ptr(beta), _ptr(C), _ptr(mean_c), _stream(Z.device))
  check(rc, "mm_pack_model")
  # the f64 inputs must outlive the asynchronous pack kernels
  pm = PackedModel(L=L, M=M, d=d, dtype=dtype, with_C=C is not None, buf=buf)
  if sync:
    torch.cuda.current_stream(Z.device).synchronize()
  else:
    pm._keep = (Z, lengthscales, variance, beta, C, mean_c)
  return pm


def _prep_state(pm: PackedModel, mu: torch.Tensor, Sigma: torch.Tensor):
  _require_device(mu, Sigma)
  if mu.dtype != pm.dtype or Sigma.dtype != pm.dtype:
    raise TypeError(f"state dtype {mu.dtype}/{Sigma.dtype} does not match the packed model ({pm.dtype})")
  B, d = mu.shape
  if d != pm.d or Sigma.shape != (B, d, d):
    raise ValueError(f"expected mu [B,{pm.d}] and Sigma [B,{pm.d},{pm.d}], got {tuple(mu.shape)}, {tuple(Sigma.shape)}")
  return B, mu.contiguous(), Sigma.contiguous()


def moment_match(pm: PackedModel, mu: torch.Tensor, Sigma: torch.Tensor,
                 full_output_cov: bool = True, model_uncertainty: bool = True,
                 jitter: float = 0.0, force_generic: bool = False, extra_flags: int = 0):
  """(mu [B,d], Sigma [B,d,d]) -> f1 [B,L], Sff [B,L,L] | [B,L], Sigma^-1 Cov(x,f) [B,d,L].
  ``extra_flags``: test / measurement bits of the C ABI (``MM_FORCE_ROUTE``, ``MM_NO_ROUTE``, ``MM_FORCE_WORST_TIER``)."""
  B, mu, Sigma = _prep_state(pm, mu, Sigma)
  flags = make_flags(full_output_cov, model_uncertainty, force_generic) | int(extra_flags)
  f1 = torch.empty(B, pm.L, dtype=pm.dtype, device=pm.device)
  Sff = torch.empty((B, pm.L, pm.L) if full_output_cov else (B, pm.L), dtype=pm.dtype, device=pm.device)
  cross = torch.empty(B, pm.d, pm.L, dtype=pm.dtype, device=pm.device)
  if B == 0:                       # an empty batch gives empty outputs (as the reference's tensor ops do); no launch
    if model_uncertainty and not pm.with_C:
      check(-5, "mm_moment_match")                    # MM_E_NO_C, as the C ABI reports it
    return f1, Sff, cross
  ws = pm.workspace(B, flags)
  rc = lib().mm_moment_match(pm.buf.data_ptr(), pm.nbytes, pm.L, pm.M, pm.d, _dtype_code(pm.dtype), B,
                             mu.data_ptr(), Sigma.data_ptr(), flags, float(jitter),
                             f1.data_ptr(), Sff.data_ptr(), cross.data_ptr(),
                             ws.data_ptr(), ws.numel(), pm.status().data_ptr(), _stream(pm.device))
  check(rc, "mm_moment_match")
  return f1, Sff, cross


def q_forward(pm: PackedModel, mu: torch.Tensor, Sigma: torch.Tensor, flags: int, want_q: bool = False):
  """Stage 1: <K_xZ> terms.  Returns f1, cross_pre, q ([B,L,M] or None)."""
  B, mu, Sigma = _prep_state(pm, mu, Sigma)
  ws = pm.workspace(B, flags)
  f1 = torch.empty(B, pm.L, dtype=pm.dtype, device=pm.device)
  cross = torch.empty(B, pm.d, pm.L, dtype=pm.dtype, device=pm.device)
  q = torch.empty(B, pm.L, pm.M, dtype=pm.dtype, device=pm.device) if want_q else None
  rc = lib().mm_q_forward(pm.buf.data_ptr(), pm.nbytes, pm.L, pm.M, pm.d, _dtype_code(pm.dtype), B,
                          mu.data_ptr(), Sigma.data_ptr(), flags, f1.data_ptr(), cross.data_ptr(), _ptr(q),
                          ws.data_ptr(), ws.numel(), pm.status().data_ptr(), _stream(pm.device))
  check(rc, "mm_q_forward")
  return f1, cross, q


def Q_reduce_forward(pm: PackedModel, B: int, flags: int, jitter: float = 0.0):
  """Stage 2: fused <K_Zx K_xZ'> reduce -> Sff.  Must follow ``q_forward`` with the same B/flags."""
  ws = pm.workspace(B, flags)
  full = bool(flags & MM_FULL_OUTPUT_COV)
  Sff = torch.empty((B, pm.L, pm.L) if full else (B, pm.L), dtype=pm.dtype, device=pm.device)
  rc = lib().mm_Q_reduce_forward(pm.buf.data_ptr(), pm.nbytes, pm.L, pm.M, pm.d, _dtype_code(pm.dtype), B,
                                 flags, float(jitter), Sff.data_ptr(), ws.data_ptr(), ws.numel(),
                                 _stream(pm.device))
  check(rc, "mm_Q_reduce_forward")
  return Sff


def offdiag_stats(pm: PackedModel, B: int, flags: int):
  """(collapsed, total, wholly inside) (b, off-diagonal pair) items of the last ``q_forward`` / ``moment_match`` with
  this B and flags (f32 models with d <= 8; zeros otherwise): items whose degree-3..6 remainder polynomial comes from
  weight moments (Cauchy-Schwarz bound on |b| <= 1/2), all items, and collapsed items whose bound puts every |b| <= 1/4 (the
  tile kernel does nothing for them).  Synchronises."""
  ws = pm.workspace(B, flags, peek=True)
  out = torch.zeros(6, dtype=torch.int32, device=pm.device)
  rc = lib().mm_offdiag_stats(pm.buf.data_ptr(), pm.nbytes, pm.L, pm.M, pm.d, _dtype_code(pm.dtype), B, flags,
                              ws.data_ptr(), ws.numel(), out.data_ptr(), _stream(pm.device))
  check(rc, "mm_offdiag_stats")
  c, n, inside, _, _, _ = out.tolist()
  return c, n, inside


def offdiag_row_groups(pm: PackedModel, B: int, flags: int):
  """(partly collapsed items, collapsed 64-row groups, all row groups) of the last ``q_forward`` / ``moment_match``: the collapse is
  decided per group of 64 rows of an item (csrc/mm_mono.h); an item counted by ``offdiag_stats`` as collapsed has all its groups
  collapsed, a partly collapsed one some.  Synchronises."""
  ws = pm.workspace(B, flags, peek=True)
  out = torch.zeros(6, dtype=torch.int32, device=pm.device)
  rc = lib().mm_offdiag_stats(pm.buf.data_ptr(), pm.nbytes, pm.L, pm.M, pm.d, _dtype_code(pm.dtype), B, flags,
                              ws.data_ptr(), ws.numel(), out.data_ptr(), _stream(pm.device))
  check(rc, "mm_offdiag_stats")
  o = out.tolist()
  Mp = (pm.M + _lib.MM_M_ALIGN - 1) // _lib.MM_M_ALIGN * _lib.MM_M_ALIGN
  return o[4], o[5], o[1] * (Mp // 64)


def offdiag_routed(pm: PackedModel, B: int, flags: int) -> int:
  """Items of the last ``moment_match`` / ``Q_reduce_forward`` with this B and flags that were re-reduced in float64
  (``mm_offdiag_stats`` out[3]; float32 models, else 0).  Synchronises."""
  if pm.dtype != torch.float32 or pm.L < 2 or not (flags & MM_FULL_OUTPUT_COV):
    return 0
  ws = pm.workspace(B, flags, peek=True)
  out = torch.zeros(6, dtype=torch.int32, device=pm.device)
  rc = lib().mm_offdiag_stats(pm.buf.data_ptr(), pm.nbytes, pm.L, pm.M, pm.d, _dtype_code(pm.dtype), B, flags,
                              ws.data_ptr(), ws.numel(), out.data_ptr(), _stream(pm.device))
  check(rc, "mm_offdiag_stats")
  return int(out[3])


def route_estimates(pm: PackedModel, B: int, flags: int) -> torch.Tensor:
  """[B, P - L, 2] float64: per (batch element, off-diagonal pair) the last sweep's estimate of its own f32 rounding error and the
  scale it is compared with (``mm_route_estimates``; float32 packs)."""
  Po = pm.L * (pm.L - 1) // 2 if (flags & MM_FULL_OUTPUT_COV) else 0
  out = torch.zeros(B, Po, 2, dtype=torch.float64, device=pm.device)
  if Po:
    ws = pm.workspace(B, flags, peek=True)
    rc = lib().mm_route_estimates(pm.buf.data_ptr(), pm.nbytes, pm.L, pm.M, pm.d, _dtype_code(pm.dtype), B, flags,
                                  ws.data_ptr(), ws.numel(), out.data_ptr(), _stream(pm.device))
    check(rc, "mm_route_estimates")
  return out


def euler_update(mu, Sigma, f1, Sff, cross_pre, dt: float = 1.0):
  """MomentMatchingEuler.step on the GPU (d == L)."""
  _require_device(mu, Sigma, f1, Sff, cross_pre)
  B, d = mu.shape
  if f1.shape != (B, d) or Sff.shape != (B, d, d) or cross_pre.shape != (B, d, d):
    raise ValueError("euler_update needs d == L and a full output covariance")
  args = [t.contiguous() for t in (mu, Sigma, f1, Sff, cross_pre)]
  mu_o, S_o = torch.empty_like(args[0]), torch.empty_like(args[1])
  rc = lib().mm_euler_update(B, d, _dtype_code(mu.dtype), float(dt), *[t.data_ptr() for t in args],
                             mu_o.data_ptr(), S_o.data_ptr(), _stream(mu.device))
  check(rc, "mm_euler_update")
  return mu_o, S_o


def rollout_closed(pm: PackedModel, mu: torch.Tensor, Sigma: torch.Tensor, num_steps: int,
                   dt: float = 1.0, model_uncertainty: bool = True, jitter: float = 0.0,
                   keep_trajectory: bool = False, force_generic: bool = False):
  """Drift-only moment-matched rollout, H steps enqueued back-to-back (state dim == d == L).

  Returns (mu_H, Sigma_H) or (mu_H, Sigma_H, traj_mu [H,B,d], traj_Sigma [H,B,d,d]).
  The inputs are not modified.
  """
  B, mu, Sigma = _prep_state(pm, mu, Sigma)
  mu, Sigma = mu.clone(), Sigma.clone()
  flags = make_flags(True, model_uncertainty, force_generic)
  ws = pm.workspace(B, flags)
  tmu = tS = None
  if keep_trajectory:
    tmu = torch.empty(num_steps, B, pm.d, dtype=pm.dtype, device=pm.device)
    tS = torch.empty(num_steps, B, pm.d, pm.d, dtype=pm.dtype, device=pm.device)
  rc = lib().mm_rollout_closed(pm.buf.data_ptr(), pm.nbytes, pm.L, pm.M, pm.d, _dtype_code(pm.dtype), B,
                               int(num_steps), float(dt), flags, float(jitter),
                               mu.data_ptr(), Sigma.data_ptr(), _ptr(tmu), _ptr(tS),
                               ws.data_ptr(), ws.numel(), pm.status().data_ptr(), _stream(pm.device))
  check(rc, "mm_rollout_closed")
  return (mu, Sigma, tmu, tS) if keep_trajectory else (mu, Sigma)


class GraphedRollout:
  """``rollout_closed`` captured once into a HIP graph and replayed (hipGraph through
  ``torch.cuda.CUDAGraph``): small configurations (C1: B = 1, M ~ 100) are launch-latency bound --
  7 kernels per step, each a few microseconds -- and the graph removes the per-launch host cost.

  The shapes (B, H), flags and the packed model are frozen at construction; ``__call__`` copies
  (mu, Sigma) into static buffers, replays, and returns the static output buffers (valid until the
  next call; clone to keep).  The device-side non-PD flag is still written: ``pm.check_status(B)``.
  """

  def __init__(self, pm: PackedModel, B: int, num_steps: int, dt: float = 1.0, model_uncertainty: bool = True,
               jitter: float = 0.0, keep_trajectory: bool = False):
    if pm.L != pm.d:
      raise ValueError("closed rollout needs state dim == d == L")
    self.pm, self.B, self.H = pm, int(B), int(num_steps)
    kw = dict(dtype=pm.dtype, device=pm.device)
    self.mu_in, self.S_in = torch.zeros(B, pm.d, **kw), torch.eye(pm.d, **kw).expand(B, pm.d, pm.d).contiguous()
    self.mu, self.S = torch.empty(B, pm.d, **kw), torch.empty(B, pm.d, pm.d, **kw)
    self.tmu = torch.empty(num_steps, B, pm.d, **kw) if keep_trajectory else None
    self.tS = torch.empty(num_steps, B, pm.d, pm.d, **kw) if keep_trajectory else None
    flags = make_flags(True, model_uncertainty, False)
    self._flags = flags
    ws = pm.workspace(B, flags)
    status = pm.status()

    def enqueue():
      self.mu.copy_(self.mu_in); self.S.copy_(self.S_in)
      rc = lib().mm_rollout_closed(pm.buf.data_ptr(), pm.nbytes, pm.L, pm.M, pm.d, _dtype_code(pm.dtype), self.B,
                                   self.H, float(dt), flags, float(jitter),
                                   self.mu.data_ptr(), self.S.data_ptr(), _ptr(self.tmu), _ptr(self.tS),
                                   ws.data_ptr(), ws.numel(), status.data_ptr(), _stream(pm.device))
      check(rc, "mm_rollout_closed")

    side = torch.cuda.Stream(device=pm.device)          # warm-up off the capture (module load, lazy init)
    side.wait_stream(torch.cuda.current_stream(pm.device))
    with torch.cuda.stream(side):
      enqueue()
    torch.cuda.current_stream(pm.device).wait_stream(side)
    torch.cuda.synchronize(pm.device)
    status.zero_()
    self.graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(self.graph):
      enqueue()

  def __call__(self, mu: torch.Tensor, Sigma: torch.Tensor):
    if tuple(mu.shape) != (self.B, self.pm.d) or tuple(Sigma.shape) != (self.B, self.pm.d, self.pm.d):
      raise ValueError(f"graph was captured for B={self.B}, d={self.pm.d}")
    _require_device(mu, Sigma)
    self.mu_in.copy_(mu); self.S_in.copy_(Sigma)
    self.pm.workspace(self.B, self._flags)                 # the replay overwrites it: counts as a use (workspace_generation)
    self.graph.replay()
    return (self.mu, self.S, self.tmu, self.tS) if self.tmu is not None else (self.mu, self.S)


class ComposedRollout:
  """The whole moment-matched policy rollout of the cartpole-shaped system on the device (``mm_rollout_composed``):
  TrigonometricEncoder -> policy (SVGP mean, Chain[Scale, Shift, NormalCDF]) -> drift SVGP with the
  ``forward_sde`` cross-covariance bookkeeping -> Euler moment update -> per-step expected cost.

  ``drift`` / ``policy``: PackedModel (drift with C; policy one latent).  ``__call__(mx, Sxx, H)`` returns
  ``(mx_H, Sxx_H, cost [B, H])`` (and the trajectory if asked); the inputs are not modified.
  """

  def __init__(self, drift: PackedModel, policy: PackedModel, nx: int, active_dims, head_scale: float, head_shift: float,
               target: torch.Tensor, precis: torch.Tensor):
    self.drift, self.policy = drift, policy
    self.nx, self.active = int(nx), tuple(int(i) for i in active_dims)
    self.na = len(self.active)
    self.ne = self.nx + self.na
    self.nd = self.ne + 1
    if drift.dtype != policy.dtype:
      raise TypeError("drift and policy must be packed with the same dtype")
    if drift.L != self.nx or drift.d != self.nd or policy.L != 1 or policy.d != self.ne:
      raise ValueError(f"shapes do not compose: drift L={drift.L} d={drift.d} (want {self.nx}, {self.nd}), "
                       f"policy L={policy.L} d={policy.d} (want 1, {self.ne})")
    if not drift.with_C:
      raise ValueError("the drift is evaluated with model uncertainty: pack it with C")
    self.scale, self.shift = float(head_scale), float(head_shift)
    self.target = target.to(dtype=drift.dtype, device=drift.device).contiguous()
    self.precis = precis.to(dtype=drift.dtype, device=drift.device).contiguous()
    self._act = (_lib.C.c_int32 * self.na)(*self.active)
    self._wsc = {}

  def _compose_ws(self, B):
    ws = self._wsc.get(B)
    if ws is None:
      n = lib().mm_compose_workspace_bytes(B, self.nx, self.na, _dtype_code(self.drift.dtype))
      if n == 0:
        raise ValueError("mm_compose_workspace_bytes rejected the shape")
      ws = torch.empty(n, dtype=torch.uint8, device=self.drift.device)
      self._wsc[B] = ws
    return ws

  def _policy_pack(self, policy: Optional[PackedModel]) -> PackedModel:
    """``policy`` (another pack of the same shape: the current parameters of a trainable policy) or ``self.policy``; a pack
    of another (L, M, d, dtype) would be read through the wrong layout -- the C side only sees a byte count."""
    pol = self.policy if policy is None else policy
    if (pol.L, pol.M, pol.d, pol.dtype) != (self.policy.L, self.policy.M, self.policy.d, self.policy.dtype):
      raise ValueError("the policy pack does not have the shape this rollout was built for")
    return pol

  def _check_state(self, mx: torch.Tensor, Sxx: torch.Tensor) -> int:
    _require_device(mx, Sxx)
    dt_ = self.drift.dtype
    if mx.dtype != dt_ or Sxx.dtype != dt_:
      raise TypeError(f"state dtype {mx.dtype} / {Sxx.dtype} does not match the packed models ({dt_})")
    B = mx.shape[0]
    if mx.shape != (B, self.nx) or Sxx.shape != (B, self.nx, self.nx):
      raise ValueError(f"expected mx [B,{self.nx}], Sxx [B,{self.nx},{self.nx}]")
    return B

  def __call__(self, mx: torch.Tensor, Sxx: torch.Tensor, num_steps: int, dt: float = 1.0, keep_trajectory: bool = False,
               policy: Optional[PackedModel] = None):
    """``policy``: another pack of the same shape to evaluate instead of ``self.policy`` (the current parameters of a
    trainable policy, packed by the caller)."""
    pol = self._policy_pack(policy)
    B = self._check_state(mx, Sxx)
    dt_ = self.drift.dtype
    mx, Sxx = mx.contiguous().clone(), Sxx.contiguous().clone()
    H = int(num_steps)
    cost = torch.empty(H, B, dtype=dt_, device=mx.device)
    tmu = torch.empty(H, B, self.nx, dtype=dt_, device=mx.device) if keep_trajectory else None
    tS = torch.empty(H, B, self.nx, self.nx, dtype=dt_, device=mx.device) if keep_trajectory else None
    wd = self.drift.workspace(B, MM_FULL_OUTPUT_COV | MM_MODEL_UNCERTAINTY)
    wp = pol.workspace(B, MM_FULL_OUTPUT_COV)
    wc = self._compose_ws(B)
    rc = lib().mm_rollout_composed(self.drift.buf.data_ptr(), self.drift.nbytes, self.drift.L, self.drift.M, self.drift.d,
                                   pol.buf.data_ptr(), pol.nbytes, pol.M, pol.d,
                                   _dtype_code(dt_), B, H, float(dt), self.nx, self.na, self._act,
                                   self.scale, self.shift, self.target.data_ptr(), self.precis.data_ptr(),
                                   mx.data_ptr(), Sxx.data_ptr(), cost.data_ptr(), _ptr(tmu), _ptr(tS),
                                   wd.data_ptr(), wd.numel(), wp.data_ptr(), wp.numel(), wc.data_ptr(), wc.numel(),
                                   self.drift.status().data_ptr(), _stream(mx.device))
    check(rc, "mm_rollout_composed")
    out = (mx, Sxx, cost.T.contiguous())
    return out + (tmu, tS) if keep_trajectory else out


  # ---- differentiable form: tape + reverse sweep (csrc/mm_compose_bwd.hip) -------------------------------------------
  BACKWARD_MAX_POLICY_M = 256

  def supports_backward(self) -> bool:
    """The native reverse sweep exists for f64 rollouts whose policy has M <= 256 centres on ne <= 8 encoded dims (one
    workgroup per batch element sweeps the policy's M x M block from LDS: 120 KB at M = 256, ne = 8)."""
    return self.drift.dtype == torch.float64 and self.policy.M <= self.BACKWARD_MAX_POLICY_M and self.ne <= 8

  def taped(self, mx: torch.Tensor, Sxx: torch.Tensor, num_steps: int, dt: float = 1.0, policy: Optional[PackedModel] = None):
    """``mm_rollout_composed_taped``: -> (mx_H, Sxx_H, cost [H, B], tape).  ``policy``: another pack of the same shape
    (the current parameters of a trainable policy)."""
    pol = self._policy_pack(policy)
    dt_ = self.drift.dtype
    B, H = self._check_state(mx, Sxx), int(num_steps)
    mx, Sxx = mx.contiguous().clone(), Sxx.contiguous().clone()
    cost = torch.empty(H, B, dtype=dt_, device=mx.device)
    n = lib().mm_compose_tape_bytes(B, H, self.nx, self.na, self.drift.M, _dtype_code(dt_))
    tape = torch.empty(n, dtype=torch.uint8, device=mx.device)
    wd = self.drift.workspace(B, MM_FULL_OUTPUT_COV | MM_MODEL_UNCERTAINTY)
    wp = pol.workspace(B, MM_FULL_OUTPUT_COV)
    rc = lib().mm_rollout_composed_taped(self.drift.buf.data_ptr(), self.drift.nbytes, self.drift.L, self.drift.M, self.drift.d,
                                         pol.buf.data_ptr(), pol.nbytes, pol.M, pol.d, _dtype_code(dt_), B, H, float(dt),
                                         self.nx, self.na, self._act, self.scale, self.shift, self.target.data_ptr(),
                                         self.precis.data_ptr(), mx.data_ptr(), Sxx.data_ptr(), cost.data_ptr(),
                                         wd.data_ptr(), wd.numel(), wp.data_ptr(), wp.numel(), tape.data_ptr(), tape.numel(),
                                         self.drift.status().data_ptr(), _stream(mx.device))
    check(rc, "mm_rollout_composed_taped")
    return mx, Sxx, cost, tape

  def backward(self, tape: torch.Tensor, g_cost: torch.Tensor, B: int, num_steps: int, dt: float = 1.0,
               policy: Optional[PackedModel] = None, want_state_grad: bool = True):
    """``mm_rollout_composed_backward``: g_cost [H, B] -> (g_policy [B, M d + M + d + 2], g_mx0 [B,nx] | None,
    g_Sxx0 [B,nx,nx] | None): the gradient w.r.t. the packed policy (Z, beta, lengthscales^2, variance, mean) per batch
    element and w.r.t. the initial state.  ``policy`` must be the pack the tape was recorded with (``taped(policy=...)``)."""
    pol = self._policy_pack(policy)
    dev = tape.device
    H = int(num_steps)
    f64 = torch.float64
    g_cost = g_cost.to(f64).contiguous()
    if g_cost.shape != (H, B):
      raise ValueError(f"g_cost must be [H={H}, B={B}]")
    npar = pol.M * pol.d + pol.M + pol.d + 2
    g_pol = torch.empty(B, npar, dtype=f64, device=dev)
    g_m = torch.empty(B, self.nx, dtype=f64, device=dev) if want_state_grad else None
    g_S = torch.empty(B, self.nx, self.nx, dtype=f64, device=dev) if want_state_grad else None
    wd = self.drift.workspace(B, MM_FULL_OUTPUT_COV | MM_MODEL_UNCERTAINTY)
    key = ("bwd", B)
    wb = self._wsc.get(key)
    if wb is None:
      n = lib().mm_compose_backward_workspace_bytes(B, self.nx, self.na, self.drift.M)
      if n == 0:
        raise ValueError("mm_compose_backward_workspace_bytes rejected the shape")
      wb = torch.empty(n, dtype=torch.uint8, device=dev)
      self._wsc[key] = wb
    rc = lib().mm_rollout_composed_backward(self.drift.buf.data_ptr(), self.drift.nbytes, self.drift.L, self.drift.M, self.drift.d,
                                            pol.buf.data_ptr(), pol.nbytes, pol.M, pol.d, _dtype_code(self.drift.dtype), B, H,
                                            float(dt), self.nx, self.na, self._act, self.scale, self.shift,
                                            self.target.data_ptr(), self.precis.data_ptr(), tape.data_ptr(), tape.numel(),
                                            g_cost.data_ptr(), g_pol.data_ptr(), _ptr(g_m), _ptr(g_S),
                                            wd.data_ptr(), wd.numel(), wb.data_ptr(), wb.numel(),
                                            self.drift.status().data_ptr(), _stream(dev))
    check(rc, "mm_rollout_composed_backward")
    return g_pol, g_m, g_S


class GraphedComposedRollout:
  """``ComposedRollout`` captured once into a HIP graph (``torch.cuda.CUDAGraph``) and replayed: at cartpole sizes
  (B = 1) a composed step is ~16 launches of a few microseconds each, and the eager path is paced by the host
  enqueueing them.  Shapes (B, H) are frozen at construction; ``__call__`` copies (mx, Sxx) into static buffers,
  replays, and returns static outputs ``(mx_H, Sxx_H, cost [B, H])`` (valid until the next call)."""

  def __init__(self, roll: ComposedRollout, B: int, num_steps: int, dt: float = 1.0):
    self.roll, self.B, self.H = roll, int(B), int(num_steps)
    kw = dict(dtype=roll.drift.dtype, device=roll.drift.device)
    self.mx_in = torch.zeros(B, roll.nx, **kw)
    self.S_in = torch.eye(roll.nx, **kw).expand(B, roll.nx, roll.nx).contiguous() * 1e-2
    dev = roll.drift.device
    side = torch.cuda.Stream(device=dev)                  # warm-up off the capture (module load, workspaces)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
      roll(self.mx_in, self.S_in, self.H, dt=dt)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    roll.drift.status().zero_()
    self.graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(self.graph):
      self.out = roll(self.mx_in, self.S_in, self.H, dt=dt)

  def __call__(self, mx: torch.Tensor, Sxx: torch.Tensor):
    if tuple(mx.shape) != (self.B, self.roll.nx):
      raise ValueError(f"graph was captured for B={self.B}, nx={self.roll.nx}")
    self.mx_in.copy_(mx); self.S_in.copy_(Sxx)
    self.roll.drift.workspace(self.B, MM_FULL_OUTPUT_COV | MM_MODEL_UNCERTAINTY)      # (see GraphedRollout.__call__)
    self.graph.replay()
    return self.out


def expected_cost(mean: torch.Tensor, cov: torch.Tensor, target: torch.Tensor, precis: torch.Tensor):
  """GaussianObjective expected cost on the GPU: mean [...,d], cov [...,d,d] -> [...]."""
  _require_device(mean, cov, target, precis)
  d = mean.shape[-1]
  lead = mean.shape[:-1]
  m2 = mean.reshape(-1, d).contiguous()
  c2 = cov.reshape(-1, d, d).contiguous().to(mean.dtype)
  out = torch.empty(m2.shape[0], dtype=mean.dtype, device=mean.device)
  rc = lib().mm_expected_cost(m2.shape[0], d, _dtype_code(mean.dtype), m2.data_ptr(), c2.data_ptr(),
                              target.to(mean).contiguous().data_ptr(), precis.to(mean).contiguous().data_ptr(),
                              out.data_ptr(), _stream(mean.device))
  check(rc, "mm_expected_cost")
  return out.reshape(lead)


def backward_supported(pm: PackedModel) -> bool:
  """Whether ``moment_match_backward`` runs on this pack itself (else: on a float64 pack of the same model)."""
  return pm.dtype == torch.float64 or bool(lib().mm_bwd_f32_supported(pm.d))


def backward_workspace_bytes(pm: PackedModel, B: int, flags: int) -> int:
  return lib().mm_moment_match_backward_bytes_dtype(B, pm.L, pm.M, pm.d, _dtype_code(pm.dtype), flags)


def moment_match_with_sums(pm: PackedModel, mu: torch.Tensor, Sigma: torch.Tensor, full_output_cov: bool = True,
                           model_uncertainty: bool = True, jitter: float = 0.0):
  """``mm_moment_match_with_sums``: the outputs of ``moment_match`` computed from the q stage and the BACKWARD's M x M sweeps
  (whose sums do not depend on the incoming gradient and contain the forward's), which stay on the returned buffer: the
  matching ``moment_match_backward(..., sums=buffer)`` is then the chain rule alone -- value and gradient for one pair of
  sweeps instead of two (C3 shape: 38 instead of 49 ms).  Returns (f1, Sff, cross_pre, sums, generation)."""
  if not backward_supported(pm):
    raise NotImplementedError("float32 packs with d > 8 differentiate through a float64 pack of the model")
  B, mu, Sigma = _prep_state(pm, mu, Sigma)
  flags = make_flags(full_output_cov, model_uncertainty)
  f1 = torch.empty(B, pm.L, dtype=pm.dtype, device=pm.device)
  Sff = torch.zeros((B, pm.L, pm.L) if full_output_cov else (B, pm.L), dtype=pm.dtype, device=pm.device)
  cross = torch.empty(B, pm.d, pm.L, dtype=pm.dtype, device=pm.device)
  if B == 0:
    return f1, Sff, cross, None, 0
  ws = pm.workspace(B, flags)
  wb = torch.empty(backward_workspace_bytes(pm, B, flags), dtype=torch.uint8, device=pm.device)   # owned by the caller's tape
  rc = lib().mm_moment_match_with_sums(pm.buf.data_ptr(), pm.nbytes, pm.L, pm.M, pm.d, _dtype_code(pm.dtype), B, mu.data_ptr(),
                                       Sigma.data_ptr(), flags, float(jitter), f1.data_ptr(), Sff.data_ptr(), cross.data_ptr(),
                                       ws.data_ptr(), ws.numel(), wb.data_ptr(), wb.numel(), pm.status().data_ptr(),
                                       _stream(pm.device))
  check(rc, "mm_moment_match_with_sums")
  return f1, Sff, cross, wb, pm.workspace_generation(B, flags)


def moment_match_backward(pm: PackedModel, mu: torch.Tensor, Sigma: torch.Tensor, g_f1: torch.Tensor, g_Sff: torch.Tensor,
                          g_cross: torch.Tensor, full_output_cov: bool = True, model_uncertainty: bool = True,
                          forward_generation: Optional[int] = None, stages: int = 0, sums: Optional[torch.Tensor] = None):
  """``mm_moment_match_backward``: the vector-Jacobian product of one moment match of a frozen pack,
  (g_f1 [B,L], g_Sff [B,L,L] | [B,L], g_cross [B,d,L]) -> (g_mu [B,d], g_Sigma [B,d,d] symmetric), gradients in float64.
  float64 packs: f64 sweeps for every pair; float32 packs with d <= 8 (``backward_supported``): f64 for the diagonal
  pairs, moment + bf16-MFMA aggregates for the off-diagonal pairs (csrc/mm_bwd_f32.hip).
  ``stages`` (``MM_STAGE_*``, measurement only): run just those parts of the backward on what earlier calls left behind.
  ``sums``: the buffer ``moment_match_with_sums`` returned for exactly this (mu, Sigma, flags): chain rule alone."""
  if not backward_supported(pm):
    raise NotImplementedError("float32 packs with d > 8 differentiate through a float64 pack of the model")
  B, mu, Sigma = _prep_state(pm, mu, Sigma)
  flags = make_flags(full_output_cov, model_uncertainty)
  f64 = torch.float64
  g_f1, g_Sff, g_cross = (t.to(f64).contiguous() for t in (g_f1, g_Sff, g_cross))
  # forward_generation: pm.workspace_generation(B, flags) right after the forward of THIS match (same mu, Sigma, flags): if
  # nobody has asked for the workspace since, its q stage is still there and is not run again (MM_WORKSPACE_CURRENT)
  if B == 0:                        # an empty batch has an empty gradient (as the forward returns empty outputs); no launch
    return (torch.empty(0, pm.d, dtype=f64, device=pm.device), torch.empty(0, pm.d, pm.d, dtype=f64, device=pm.device))
  current = (forward_generation is not None and forward_generation > 0
             and forward_generation == pm.workspace_generation(B, flags))
  ws = pm.workspace(B, flags, peek=current)
  key = ("bwd", B, flags)
  wb = sums if sums is not None else pm._workspaces.get(key)
  if wb is None:
    wb = torch.empty(backward_workspace_bytes(pm, B, flags), dtype=torch.uint8, device=pm.device)
    pm._workspaces[key] = wb
  if sums is not None:
    if sums.numel() < backward_workspace_bytes(pm, B, flags) or sums.device != pm.device:
      raise ValueError("`sums` is not the buffer moment_match_with_sums returned for this pack, batch and flags")
    stages |= MM_SUMS_CURRENT
  g_mu = torch.empty(B, pm.d, dtype=f64, device=pm.device)
  g_S = torch.empty(B, pm.d, pm.d, dtype=f64, device=pm.device)
  rc = lib().mm_moment_match_backward(pm.buf.data_ptr(), pm.nbytes, pm.L, pm.M, pm.d, _dtype_code(pm.dtype), B, mu.data_ptr(), Sigma.data_ptr(),
                                      flags | (MM_WORKSPACE_CURRENT if current else 0) | stages, g_f1.data_ptr(), g_Sff.data_ptr(),
                                      g_cross.data_ptr(), g_mu.data_ptr(),
                                      g_S.data_ptr(), 0, ws.data_ptr(), ws.numel(), wb.data_ptr(), wb.numel(),
                                      pm.status().data_ptr(), _stream(pm.device))
  check(rc, "mm_moment_match_backward")
  return g_mu, g_S
