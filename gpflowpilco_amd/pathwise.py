"""Pathwise (decoupled-sampling) GP paths and sample rollouts -- SURVEY.md row f-3.

Mirrors the surface the reference uses from ``gpflow_sampling`` (un-vendored third party):
``PathwiseSVGP.generate_paths(num_samples, num_bases, sample_axis=0)``,
``set_temporary_paths`` and ``predict_f_samples`` / ``__call__``
(``gpflow_pilco/models/svgp.py:124-130``, ``loops/pilco.py:263-303``).  Path *generation*
(sampling weights, one Cholesky solve per latent) is torch plumbing done once per rollout
closure; path *evaluation* -- the per-step hot loop -- runs in ``mm_pathwise_eval`` /
``mm_pathwise_rollout`` (``csrc/mm_pathwise.hip``).  Parity with gpflow_sampling is unpinned
(see ``oracle/pathwise_oracle.py``).
"""
from __future__ import annotations

import contextlib
import math
from dataclasses import dataclass
from typing import Optional

import torch

from .linalg import cholesky

from . import _lib
from .models import DEFAULT_FLOAT, DEFAULT_JITTER, SVGP, Constant, Zero, _stack_kernel_params, unpack_multioutput
from .ops import _dtype_code, _ptr, _require_device, _stream, check


BOUND_ULPS = 8.0      # Paths.eval_with_bound: rounding bound = BOUND_ULPS x unit roundoff x the sum of the absolute terms


def _pad_last(t: torch.Tensor, mult: int) -> torch.Tensor:
  n = t.shape[-1]
  pad = (-n) % mult
  return t if pad == 0 else torch.nn.functional.pad(t, (0, pad))


@dataclass
class Paths:
  """S sample paths of L latent GPs (device tensors, element type ``dtype``)."""
  omega: torch.Tensor      # [L, d, Kp]   omega^T / 2 pi  (revolutions, k-major)
  phase: torch.Tensor      # [L, Kp]      b / 2 pi
  zs: torch.Tensor         # [L, d, Mp]   (Z * x_scale)^T, x_scale = sqrt(log2 e) / lengthscales
  hz: torch.Tensor         # [L, Mp]      |zs|^2 / 2
  wb: torch.Tensor         # [G, L, NB, 4, BT] blocked weight stream (prior blocks, then update blocks)
  num_samples: int
  lengthscales: torch.Tensor   # [L, d] f64: x_scale = sqrt(log2 e) / lengthscales
  prior_scale: torch.Tensor    # [L] f64  sqrt(2 var / K)
  variance: torch.Tensor       # [L] f64
  mean_c: Optional[torch.Tensor]  # [L] f64 or None

  @property
  def dtype(self):
    return self.wb.dtype

  def _dims(self):
    L, d, Kp = self.omega.shape
    return self.num_samples, L, self.zs.shape[-1], Kp, d

  def __call__(self, x: torch.Tensor) -> torch.Tensor:
    """f_s(x_s): x [S, d] -> [S, L].  Differentiable in x where the Jacobian pass exists (d <= 8): the torch composition of a
    sample rollout (``loops.pathwise_policy_loss_closure``'s fallback) then carries gradients through the paths."""
    if torch.is_grad_enabled() and x.requires_grad:
      return _PathsEval.apply(x, self)
    _require_device(x, self.wb)
    S, L, Mp, Kp, d = self._dims()
    if x.shape != (S, d) or x.dtype != self.dtype:
      raise ValueError(f"expected x [{S},{d}] of {self.dtype}, got {tuple(x.shape)} {x.dtype}")
    x = x.contiguous()
    out = torch.empty(S, L, dtype=self.dtype, device=x.device)
    rc = _lib.lib().mm_pathwise_eval(S, L, Mp, Kp, d, _dtype_code(self.dtype), x.data_ptr(), self.omega.data_ptr(),
                                     self.phase.data_ptr(), self.zs.data_ptr(), self.hz.data_ptr(),
                                     self.lengthscales.data_ptr(), self.prior_scale.data_ptr(),
                                     self.variance.data_ptr(), _ptr(self.mean_c), self.wb.data_ptr(),
                                     out.data_ptr(), _stream(x.device))
    check(rc, "mm_pathwise_eval")
    return out

  def eval_jac(self, x: torch.Tensor):
    """f_s(x_s) and its Jacobian: x [S, d] -> (f [S, L], d f / d x [S, L, d]) from ONE pass over the weight stream (d <= 8)."""
    _require_device(x, self.wb)
    S, L, Mp, Kp, d = self._dims()
    if x.shape != (S, d) or x.dtype != self.dtype:
      raise ValueError(f"expected x [{S},{d}] of {self.dtype}, got {tuple(x.shape)} {x.dtype}")
    x = x.contiguous()
    out = torch.empty(S, L, dtype=self.dtype, device=x.device)
    jac = torch.empty(S, L, d, dtype=self.dtype, device=x.device)
    rc = _lib.lib().mm_pathwise_eval_jac(S, L, Mp, Kp, d, _dtype_code(self.dtype), x.data_ptr(), self.omega.data_ptr(),
                                         self.phase.data_ptr(), self.zs.data_ptr(), self.hz.data_ptr(),
                                         self.lengthscales.data_ptr(), self.prior_scale.data_ptr(),
                                         self.variance.data_ptr(), _ptr(self.mean_c), self.wb.data_ptr(),
                                         out.data_ptr(), jac.data_ptr(), _stream(x.device))
    check(rc, "mm_pathwise_eval_jac")
    return out, jac

  def eval_with_bound(self, x: torch.Tensor):
    """f_s(x_s) and a bound on what this dtype's rounding did to it: x [S, d] -> (f [S, L], err [S, L]) from ONE pass over the
    weight stream (``mm_pathwise_eval_bound``).  err = BOUND_ULPS u (scale sum_k |w cos| + var sum_m |v k|): the update weights
    v = Kuu^-1 (u - Phi w) cancel 1e5 .. 1e7-fold in sum_m v_m k(x, z_m) at M = 2000, so a float32 sample's value can have lost
    its digits (C5 shard: ~2e-2 of max |f|) -- this is where the caller sees it; float64 paths are the accurate mode."""
    _require_device(x, self.wb)
    S, L, Mp, Kp, d = self._dims()
    if x.shape != (S, d) or x.dtype != self.dtype:
      raise ValueError(f"expected x [{S},{d}] of {self.dtype}, got {tuple(x.shape)} {x.dtype}")
    x = x.contiguous()
    out = torch.empty(S, L, dtype=self.dtype, device=x.device)
    ab = torch.empty(S, L, dtype=self.dtype, device=x.device)
    rc = _lib.lib().mm_pathwise_eval_bound(S, L, Mp, Kp, d, _dtype_code(self.dtype), x.data_ptr(), self.omega.data_ptr(),
                                           self.phase.data_ptr(), self.zs.data_ptr(), self.hz.data_ptr(),
                                           self.lengthscales.data_ptr(), self.prior_scale.data_ptr(),
                                           self.variance.data_ptr(), _ptr(self.mean_c), self.wb.data_ptr(),
                                           out.data_ptr(), ab.data_ptr(), _stream(x.device))
    check(rc, "mm_pathwise_eval_bound")
    # unit roundoff u = finfo.eps / 2; v and the basis value each carry ~u, the exponent's own rounding (|arg| up to ~20 at the
    # C5 shape) a few u more: BOUND_ULPS u covers the measured worst case with a factor ~2 in hand (tests/test_pathwise.py)
    return out, ab * (BOUND_ULPS * 0.5 * torch.finfo(self.dtype).eps)

  def flagged(self, x: torch.Tensor, tol: float):
    """(f, mask [S, L], count): the (sample, latent) values whose rounding bound exceeds ``tol`` x max |f| of the batch."""
    f, err = self.eval_with_bound(x)
    mask = err > tol * f.abs().amax()
    return f, mask, int(mask.sum().item())

  def rollout(self, x0: torch.Tensor, num_steps: int, dt: float = 1.0, keep_trajectory: bool = False):
    """Drift-only Euler rollout of all S paths (d == L): x <- x + dt f(x), H steps in one ABI call."""
    _require_device(x0, self.wb)
    S, L, Mp, Kp, d = self._dims()
    x = x0.contiguous().clone()
    tmp = torch.empty_like(x)
    traj = torch.empty(num_steps, S, d, dtype=self.dtype, device=x.device) if keep_trajectory else None
    rc = _lib.lib().mm_pathwise_rollout(S, L, Mp, Kp, d, _dtype_code(self.dtype), int(num_steps), float(dt),
                                        x.data_ptr(), tmp.data_ptr(), self.omega.data_ptr(), self.phase.data_ptr(),
                                        self.zs.data_ptr(), self.hz.data_ptr(), self.lengthscales.data_ptr(),
                                        self.prior_scale.data_ptr(), self.variance.data_ptr(), _ptr(self.mean_c),
                                        self.wb.data_ptr(), _ptr(traj), _stream(x.device))
    check(rc, "mm_pathwise_rollout")
    return (x, traj) if keep_trajectory else x


class _PathsEval(torch.autograd.Function):
  """f = paths(x) with d f / d x from the same weight-stream pass (``mm_pathwise_eval_jac``)."""

  @staticmethod
  def forward(ctx, x, paths):
    f, jac = paths.eval_jac(x.detach())
    ctx.save_for_backward(jac)
    return f

  @staticmethod
  def backward(ctx, g):
    (jac,) = ctx.saved_tensors
    return torch.einsum('sl,sld->sd', g, jac), None


def paths_from_arrays(omega, phase, w, v, Z, lengthscales, variance, mean_c=None, dtype=torch.float32,
                      device="cuda") -> Paths:
  """Build ``Paths`` from explicit arrays (omega [L,K,d], phase [L,K], w [S,L,K], v [S,L,M], Z [L,M,d])."""
  t64 = lambda a: torch.as_tensor(a, dtype=DEFAULT_FLOAT, device=device)
  mult = 128 if dtype == torch.float64 else 256          # BT: terms per block of the weight stream
  xscale = math.sqrt(math.log2(math.e)) / t64(lengthscales)                    # [L, d]
  zs = t64(Z) * xscale[:, None, :]
  hz = 0.5 * (zs * zs).sum(-1)
  K = omega.shape[1]
  tt = lambda a: a.to(dtype).contiguous()
  padk = lambda a: tt(_pad_last(t64(a), mult))
  two_pi = 2.0 * math.pi
  omega_p = tt(_pad_last(t64(omega).transpose(1, 2) / two_pi, mult))           # [L, d, Kp]
  zs_p = tt(_pad_last(zs.transpose(1, 2), mult))                                # [L, d, Mp]
  var = t64(variance)
  # blocked weight stream [G, L, NB, 4, BT]
  wp, vp = _pad_last(t64(w), mult), _pad_last(t64(v), mult)
  S, L = wp.shape[:2]
  G = (S + 3) // 4
  allw = torch.cat([wp, vp], dim=-1)                                              # [S, L, Kp + Mp]
  if G * 4 != S:
    allw = torch.cat([allw, torch.zeros(G * 4 - S, L, allw.shape[-1], dtype=DEFAULT_FLOAT, device=device)], 0)
  NB = allw.shape[-1] // mult
  wb = allw.reshape(G, 4, L, NB, mult).permute(0, 2, 3, 1, 4).to(dtype).contiguous()
  return Paths(omega=omega_p, phase=padk(t64(phase) / two_pi), zs=zs_p, hz=padk(hz), wb=wb, num_samples=S,
               lengthscales=xscale.contiguous(), prior_scale=torch.sqrt(2.0 * var / K).contiguous(),
               variance=var.contiguous(), mean_c=None if mean_c is None else t64(mean_c).contiguous())


def generate_paths(model: SVGP, num_samples: int, num_bases: int = 1024, dtype=torch.float32,
                   device="cuda", generator: Optional[torch.Generator] = None) -> Paths:
  """Draw S decoupled sample paths of an SVGP: random-Fourier prior + inducing-point update
  (gpflow_sampling's decoupled sampler; ``loops/pilco.py:281-284``)."""
  kernels, Zs = unpack_multioutput(model.kernel, model.inducing_variable, model.num_latent_gps)
  Z, ls, var = _stack_kernel_params(kernels, Zs, device)
  L, M, d = Z.shape
  S, K = num_samples, num_bases
  rn = lambda *shape: torch.randn(*shape, dtype=DEFAULT_FLOAT, device=device, generator=generator)
  omega = rn(L, K, d) / ls[:, None, :]
  phase = 2.0 * math.pi * torch.rand(L, K, dtype=DEFAULT_FLOAT, device=device, generator=generator)
  w = rn(S, L, K)
  A = Z / ls[:, None, :]
  d2 = (A * A).sum(-1)[:, :, None] + (A * A).sum(-1)[:, None, :] - 2.0 * A @ A.transpose(1, 2)
  Kuu = var[:, None, None] * torch.exp(-0.5 * d2.clamp_min(0.0)) + DEFAULT_JITTER * torch.eye(M, dtype=DEFAULT_FLOAT, device=device)
  Luu = cholesky(Kuu)
  q_mu = model.q_mu.to(device=device, dtype=DEFAULT_FLOAT).T                       # [L, M]
  q_sqrt = torch.tril(model.q_sqrt.to(device=device, dtype=DEFAULT_FLOAT))          # [L, M, M]
  eps = rn(S, L, M)
  u = q_mu[None] + torch.einsum('slm,lnm->sln', eps, q_sqrt)                       # samples of q(u)
  if model.whiten:
    u = torch.einsum('lnm,slm->sln', Luu, u)
  Phi_Z = torch.sqrt(2.0 * var / K)[:, None, None] * torch.cos(Z @ omega.transpose(1, 2) + phase[:, None, :])  # [L,M,K]
  resid = u - torch.einsum('lmk,slk->slm', Phi_Z, w)
  v = torch.cholesky_solve(resid.permute(1, 2, 0), Luu).permute(2, 0, 1)             # [S, L, M]
  mean_c = None
  if isinstance(model.mean_function, Constant):
    mean_c = model.mean_function.c.to(device=device, dtype=DEFAULT_FLOAT).expand(L).contiguous()
  elif not isinstance(model.mean_function, Zero):
    raise NotImplementedError
  return paths_from_arrays(omega, phase, w, v, Z, ls, var, mean_c, dtype=dtype, device=device)


class PathwiseSVGP(SVGP):
  """``gpflow_pilco.models.PathwiseSVGP`` (models/svgp.py:124-130): an SVGP whose ``__call__``
  evaluates the currently attached sample paths."""

  _paths: Optional[Paths] = None

  def generate_paths(self, num_samples: int, num_bases: int = 1024, sample_axis: int = 0,
                     dtype=torch.float32, device="cuda", generator=None) -> Paths:
    assert sample_axis == 0
    return generate_paths(self, num_samples, num_bases, dtype=dtype, device=device, generator=generator)

  @contextlib.contextmanager
  def set_temporary_paths(self, paths: Paths):
    prev, self._paths = self._paths, paths
    try:
      yield
    finally:
      self._paths = prev

  def predict_f_samples(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
    if self._paths is None:
      raise RuntimeError("no sample paths attached: use generate_paths / set_temporary_paths")
    return self._paths(x)

  def __call__(self, x, **kwargs):
    return self.predict_f_samples(x, **kwargs)


class PolicyRollout:
  """The pathwise policy rollout on the device (``mm_pathwise_policy_rollout``, csrc/mm_pathwise_policy.hip): per sample path
  encoder -> policy mean through Chain[Scale, Shift, NormalCDF] -> drift sample -> Euler -> cost -- the body of
  ``PathwisePILCO._policy_loss_closure`` (gpflow_pilco/loops/pilco.py:263-298).

  ``paths``: the drift's sample paths (nx latents on nd = nx + na + 1 inputs, nd <= 8); ``policy``: a one-latent
  ``ops.PackedModel`` on ne = nx + na inputs (any dtype: only its float64 blocks are read).  ``__call__(x0, H)`` ->
  ``(cost [H, S], tape)``; ``backward(tape, g_cost)`` -> ``(g_policy [M ne + M + ne + 2], g_x0 [S, nx] | None)``."""

  def __init__(self, paths: Paths, policy, nx: int, active_dims, head_scale: float, head_shift: float,
               target: torch.Tensor, precis: torch.Tensor):
    S, L, Mp, Kp, d = paths._dims()
    self.paths, self.policy = paths, policy
    self.nx, self.active = int(nx), tuple(int(i) for i in active_dims)
    self.na = len(self.active)
    self.ne, self.nd = self.nx + self.na, self.nx + self.na + 1
    if L != self.nx or d != self.nd or policy.L != 1 or policy.d != self.ne:
      raise ValueError(f"shapes do not compose: paths L={L} d={d} (want {self.nx}, {self.nd}), policy L={policy.L} d={policy.d} "
                       f"(want 1, {self.ne})")
    if self.nd > 8 or policy.M > 256:
      raise ValueError("the pathwise policy rollout takes drift inputs of dimension <= 8 and policies of <= 256 centres")
    self.scale, self.shift = float(head_scale), float(head_shift)
    self.target = target.to(dtype=paths.dtype, device=paths.wb.device).contiguous()
    self.precis = precis.to(dtype=paths.dtype, device=paths.wb.device).contiguous()
    self._act = (_lib.C.c_int32 * self.na)(*self.active)

  def _policy(self, policy):
    pol = self.policy if policy is None else policy
    if (pol.L, pol.M, pol.d) != (self.policy.L, self.policy.M, self.policy.d):
      raise ValueError("the policy pack does not have the shape this rollout was built for")
    return pol

  def __call__(self, x0: torch.Tensor, num_steps: int, dt: float = 1.0, with_jacobians: bool = False, policy=None):
    pol = self._policy(policy)
    P = self.paths
    S, L, Mp, Kp, d = P._dims()
    _require_device(x0, P.wb)
    if x0.shape != (S, self.nx) or x0.dtype != P.dtype:
      raise ValueError(f"expected x0 [{S},{self.nx}] of {P.dtype}, got {tuple(x0.shape)} {x0.dtype}")
    H, code = int(num_steps), _dtype_code(P.dtype)
    n = _lib.lib().mm_pathwise_tape_bytes(S, H, self.nx, self.na, code, int(with_jacobians))
    if n == 0:
      raise ValueError("mm_pathwise_tape_bytes rejected the shape")
    tape = torch.empty(n, dtype=torch.uint8, device=x0.device)
    cost = torch.empty(H, S, dtype=P.dtype, device=x0.device)
    x0 = x0.contiguous()
    rc = _lib.lib().mm_pathwise_policy_rollout(S, Mp, Kp, code, H, float(dt), self.nx, self.na, self._act,
                                               P.omega.data_ptr(), P.phase.data_ptr(), P.zs.data_ptr(), P.hz.data_ptr(),
                                               P.lengthscales.data_ptr(), P.prior_scale.data_ptr(), P.variance.data_ptr(),
                                               _ptr(P.mean_c), P.wb.data_ptr(), pol.buf.data_ptr(), pol.nbytes, pol.M,
                                               self.scale, self.shift, self.target.data_ptr(), self.precis.data_ptr(),
                                               x0.data_ptr(), cost.data_ptr(), tape.data_ptr(), tape.numel(),
                                               int(with_jacobians), _stream(x0.device))
    check(rc, "mm_pathwise_policy_rollout")
    return cost, tape

  def states(self, tape: torch.Tensor, num_steps: int) -> torch.Tensor:
    """x_0 .. x_H [H + 1, S, nx] (a view of the tape)."""
    S = self.paths.num_samples
    es = 8 if self.paths.dtype == torch.float64 else 4
    n = (num_steps + 1) * S * self.nx
    return tape[:n * es].view(self.paths.dtype).view(num_steps + 1, S, self.nx)

  def backward(self, tape: torch.Tensor, g_cost: torch.Tensor, num_steps: int, dt: float = 1.0, policy=None,
               want_state_grad: bool = False):
    pol = self._policy(policy)
    S, H, code = self.paths.num_samples, int(num_steps), _dtype_code(self.paths.dtype)
    dev = tape.device
    g_cost = g_cost.to(torch.float64).contiguous()
    if g_cost.shape != (H, S):
      raise ValueError(f"g_cost must be [H={H}, S={S}]")
    npar = pol.M * self.ne + pol.M + self.ne + 2
    g_pol = torch.empty(npar, dtype=torch.float64, device=dev)
    g_x0 = torch.empty(S, self.nx, dtype=torch.float64, device=dev) if want_state_grad else None
    ns = _lib.lib().mm_pathwise_backward_scratch_bytes(S, pol.M, self.ne)
    scratch = torch.empty(ns, dtype=torch.uint8, device=dev)
    rc = _lib.lib().mm_pathwise_policy_rollout_backward(S, code, H, float(dt), self.nx, self.na, self._act, pol.buf.data_ptr(),
                                                        pol.nbytes, pol.M, self.scale, self.shift, self.target.data_ptr(),
                                                        self.precis.data_ptr(), tape.data_ptr(), tape.numel(),
                                                        g_cost.data_ptr(), g_pol.data_ptr(), _ptr(g_x0), scratch.data_ptr(),
                                                        scratch.numel(), _stream(dev))
    check(rc, "mm_pathwise_policy_rollout_backward")
    return g_pol, g_x0


class PolicyRolloutFunction(torch.autograd.Function):
  """The pathwise policy loss as ONE differentiable op: forward = ``mm_pathwise_policy_rollout`` with the Jacobian tape,
  backward = ``mm_pathwise_policy_rollout_backward``.  The policy enters in packed coordinates (Z [1,M,ne], lengthscales
  [1,ne], variance [1], beta [1,M], mean_c [1]) computed from its parameters by differentiable torch ops, as in
  ``autodiff.ComposedRolloutFunction``.  Output: cost [S, H]."""

  @staticmethod
  def forward(ctx, x0, Z, ls, var, beta, mean_c, roll, num_steps, dt):
    from . import ops
    f64 = torch.float64
    det = lambda t: t.detach().to(f64)
    pol = ops.pack_model(det(Z), det(ls), det(var), det(beta), None, det(mean_c), dtype=f64, sync=False)
    cost, tape = roll(x0.detach(), num_steps, dt=dt, with_jacobians=True, policy=pol)
    ctx.roll, ctx.pol, ctx.tape, ctx.H, ctx.dt = roll, pol, tape, int(num_steps), float(dt)
    ctx.save_for_backward(ls)
    ctx.need_state = x0.requires_grad
    ctx.shapes = (Z.shape, ls.shape, var.shape, beta.shape, mean_c.shape)
    ctx.x_dtype = x0.dtype
    return cost.T.contiguous()

  @staticmethod
  def backward(ctx, g_cost):
    (ls,) = ctx.saved_tensors
    g, g_x0 = ctx.roll.backward(ctx.tape, g_cost.T.contiguous(), ctx.H, dt=ctx.dt, policy=ctx.pol, want_state_grad=ctx.need_state)
    M, d = ctx.pol.M, ctx.pol.d
    zs, lss, vs, bs, ms = ctx.shapes
    gZ = g[:M * d].reshape(zs)
    gbeta = g[M * d:M * d + M].reshape(bs)
    gls = (2.0 * ls.detach().reshape(-1) * g[M * d + M:M * d + M + d]).reshape(lss)        # d/d ls = 2 ls d/d ls^2
    gvar = g[M * d + M + d].reshape(vs)
    gmean = g[M * d + M + d + 1].reshape(ms)
    return (None if g_x0 is None else g_x0.to(ctx.x_dtype)), gZ, gls, gvar, gbeta, gmean, None, None, None
