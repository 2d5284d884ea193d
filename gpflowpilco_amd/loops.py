"""The rollout harness of ``MomentMatchingPILCO`` (``gpflow_pilco/loops/pilco.py:176-227``):
``policy_loss_closure`` builds the function whose value the policy optimiser minimises -- the
expected cost accumulated along a moment-matched rollout.  The RL orchestration around it
(data collection, checkpoints, optimisers) is out of scope."""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import warnings

import numpy as np
import torch

from .dynamics import DynamicalSystem, Euler, MomentMatchingEuler
from .moment_matching import GaussianMoments, moment_matching


def get_state_initializer(mean: torch.Tensor, covariance: torch.Tensor) -> Callable:
  """pilco.py:222-227: the initial state as moments with a leading batch axis."""
  mx = mean if mean.ndim > 1 else mean[None]
  Sxx = covariance if covariance.ndim > 2 else covariance[None]
  return lambda: (mx, Sxx)


def _native_parts(system: DynamicalSystem, objective: Callable, why: Optional[list], moment_solver: bool):
  """The pieces the native rollouts are written for -- TrigonometricEncoder, policy = InverseLinkWrapper(KernelRegressor(SVGP with
  one latent), Chain[Scale, Shift, NormalCDF]) with scalar scale and shift, SVGP drift, no diffusion, GaussianObjective (the
  cartpole wiring of ``examples/cartpole_swingup/swingup_loops.py:41-91``) -- or None, with the reason appended to ``why``."""
  from . import bijectors as tfb
  from .components import TrigonometricEncoder
  from .cost import GaussianObjective
  from .models import SVGP, InverseLinkWrapper, KernelRegressor, LinearCoregionalization
  enc, pol, drift = system.encoder, system.policy, system.drift

  def no(reason):
    if why is not None:
      why.append(reason)
    return None
  if system.diffusion is not None:
    return no("a diffusion term")
  if moment_solver and not isinstance(system.solver, MomentMatchingEuler):
    return no("a solver other than MomentMatchingEuler")
  if not isinstance(enc, TrigonometricEncoder):
    return no(f"encoder {type(enc).__name__} (the native rollout implements TrigonometricEncoder)")
  if not isinstance(objective, GaussianObjective):
    return no(f"objective {type(objective).__name__} (the native rollout implements GaussianObjective)")
  if not isinstance(pol, InverseLinkWrapper) or not isinstance(pol.model, KernelRegressor):
    return no("the policy is not InverseLinkWrapper(KernelRegressor(SVGP))")
  pm_, head = pol.model.model, pol.invlink
  if not isinstance(pm_, SVGP) or not isinstance(drift, SVGP) or pm_.num_latent_gps != 1:
    return no("the policy is not a one-latent SVGP or the drift is not an SVGP")
  if isinstance(pm_.kernel, LinearCoregionalization) or isinstance(drift.kernel, LinearCoregionalization):
    return no("a LinearCoregionalization kernel (its mixing stays on the host)")
  if any(k.active_dims is not None for k in pm_.latent_kernels + drift.latent_kernels):
    return no("a kernel with active_dims")
  bj = head.bijectors if isinstance(head, tfb.Chain) else None
  if not (bj and len(bj) == 3 and isinstance(bj[0], tfb.Scale) and isinstance(bj[1], tfb.Shift) and isinstance(bj[2], tfb.NormalCDF)):
    return no("the policy head is not Chain[Scale, Shift, NormalCDF]")

  def head_constants():
    sc, sh = bj[0].scale, bj[1].shift
    return (float(sc.detach()) if isinstance(sc, torch.Tensor) else float(sc),
            float(sh.detach()) if isinstance(sh, torch.Tensor) else float(sh))
  try:
    head_constants()
  except (TypeError, ValueError, RuntimeError):
    return no("the policy head's scale / shift are not scalars (n-D action: Genz BVN, out of scope)")
  return enc, pm_, drift, bj, head_constants


def native_policy_loss(system: DynamicalSystem, objective: Callable, num_steps: int, dt: float = 1.0,
                       why: Optional[list] = None):
  """``f(mx, Sxx) -> loss [B]`` running the whole rollout in ``mm_rollout_composed`` (csrc/mm_compose.hip), or None
  when the system is not the shape that entry point implements: TrigonometricEncoder, policy =
  InverseLinkWrapper(KernelRegressor(SVGP with one latent), Chain[Scale, Shift, NormalCDF]) with scalar scale and
  shift, SVGP drift, no diffusion, MomentMatchingEuler, GaussianObjective -- the cartpole wiring of
  ``examples/cartpole_swingup/swingup_loops.py:41-91``.  ``f.with_grad(mx, Sxx)`` is the same loss as a differentiable op
  (native reverse sweep, csrc/mm_compose_bwd.hip) where ``f.supports_grad(mx)``.  ``why``: a list that receives the reason
  when None is returned."""
  from . import ops
  parts = _native_parts(system, objective, why, moment_solver=True)
  if parts is None:
    return None
  enc, pm_, drift, bj, head_constants = parts
  cache = {}

  def current_roll(mx: torch.Tensor, fresh_policy: bool = True):
    # (fresh_policy=False: the caller brings its own pack of the policy's current parameters -- the differentiable
    # path -- and the rollout object only has to have the right shapes: no re-pack of the policy here)
    # Everything the rollout reads is looked up on EVERY call: ``packed()`` re-packs a model whose parameters were
    # updated in place (optimiser step, refit between episodes: _PackCache keys on the tensors' versions), and the
    # head / objective constants are read from their owners.  Only the compose workspace is kept across calls.
    key = (mx.dtype, str(mx.device))
    ent = cache.get(key)
    roll = None if ent is None else ent[0]
    pd = drift.packed(mx.dtype, True, mx.device)
    pp = roll.policy if (roll is not None and not fresh_policy) else pm_.packed(mx.dtype, False, mx.device)
    scale, shift = head_constants()
    if (roll is None or roll.drift is not pd or roll.policy is not pp or roll.scale != scale or roll.shift != shift
        or ent[1] is not objective.target or ent[2] is not objective.precis
        or ent[3] != (objective.target._version, objective.precis._version)):
      new = ops.ComposedRollout(pd, pp, nx=mx.shape[-1], active_dims=enc.active_dims, head_scale=scale, head_shift=shift,
                                target=objective.target, precis=objective.precis)
      if roll is not None:
        new._wsc = roll._wsc                               # same shapes: the workspace carries over
      roll = new
      cache[key] = (roll, objective.target, objective.precis, (objective.target._version, objective.precis._version))
    return roll

  def run(mx: torch.Tensor, Sxx: torch.Tensor):
    _, _, cost = current_roll(mx)(mx, Sxx, num_steps, dt=dt)
    return cost.sum(1)

  def run_with_grad(mx: torch.Tensor, Sxx: torch.Tensor):
    """The same loss as a differentiable function of the policy's parameters (and of the initial state): forward =
    the taped native rollout, backward = the native reverse sweep (autodiff.ComposedRolloutFunction); the policy enters
    in packed coordinates computed from its parameters by differentiable torch ops (a 30 x 30 precompute).  The tape and
    the reverse sweep are float64; a float32 state is cast up on the way in and the loss back down (autograd carries both)."""
    from .autodiff import ComposedRolloutFunction
    out_dtype = mx.dtype
    if mx.dtype != torch.float64:
      mx, Sxx = mx.double(), Sxx.double()
    roll = current_roll(mx, fresh_policy=False)
    Zp, lsp, varp, betap, _, mcp = pm_.precompute(mx.device)
    if mcp is None:
      mcp = torch.zeros(1, dtype=Zp.dtype, device=mx.device)
    cost = ComposedRolloutFunction.apply(mx, Sxx, Zp, lsp, varp, betap, mcp, roll, num_steps, dt)
    return cost.sum(1).to(out_dtype)

  def run_from_parameters(mx: torch.Tensor, Sxx: torch.Tensor):
    """Forward only, the policy packed from its parameters on the stream (no snapshot from outside): what a HIP-graph
    capture of the loss of a trainable policy must record, so that replays follow the optimiser's in-place updates."""
    with torch.no_grad():
      roll = current_roll(mx, fresh_policy=False)
      Zp, lsp, varp, betap, _, mcp = pm_.precompute(mx.device)
      pol = ops.pack_model(Zp, lsp, varp, betap, None, mcp, dtype=mx.dtype, sync=False)
      _, _, cost = roll(mx, Sxx, num_steps, dt=dt, policy=pol)
    return cost.sum(1)
  run.from_parameters = run_from_parameters

  def grad_obstacle(mx: torch.Tensor) -> Optional[str]:
    """None when ``with_grad`` covers everything that asks for a gradient here, else the reason it does not."""
    if mx.dtype not in (torch.float32, torch.float64):
      return f"state dtype {mx.dtype}"
    # the native reverse sweep returns gradients for the policy SVGP's parameters and the initial state only: the head's
    # Scale / Shift and the objective's target / precision enter as constants (float() / raw pointers)
    outside = {"the policy head's Scale.scale": bj[0].scale, "the policy head's Shift.shift": bj[1].shift,
               "objective.target": objective.target, "objective.precis": objective.precis}
    for name, t in outside.items():
      if isinstance(t, torch.Tensor) and t.requires_grad:
        return f"{name} requires a gradient (the native reverse sweep covers the policy SVGP's parameters and the initial state)"
    if any(t.requires_grad for t in drift._parameters()):
      return "the drift is being trained (the native reverse sweep takes a frozen drift)"
    mx64 = mx if mx.dtype == torch.float64 else torch.empty(mx.shape, dtype=torch.float64, device=mx.device)
    roll = current_roll(mx64, fresh_policy=False)
    if not roll.supports_backward():
      return (f"the policy has M = {roll.policy.M} centres on {roll.ne} encoded dims (the native reverse sweep takes "
              f"M <= {ops.ComposedRollout.BACKWARD_MAX_POLICY_M}, encoded dim <= 8)")
    return None

  def supports_grad(mx: torch.Tensor) -> bool:
    return grad_obstacle(mx) is None
  run.with_grad = run_with_grad
  run.supports_grad = supports_grad
  run.grad_obstacle = grad_obstacle
  return run


def policy_loss_closure(system: DynamicalSystem, objective: Callable, state_initializer: Callable,
                        num_steps: int, initial_time: float = 0.0,
                        solution_times: Optional[Sequence[float]] = None, native: Optional[bool] = None,
                        **kwargs) -> Callable:
  """pilco.py:176-220.  Returns ``closure() -> loss [B]``; ``system.solver`` should be a
  ``MomentMatchingEuler`` (pilco.py:141-144).

  ``native``: None (default) runs the rollout natively whenever the system has the shape ``mm_rollout_composed``
  implements (``native_policy_loss``) and the state is on the GPU: forward only (one ``mm_rollout_composed`` call) when
  nothing requires a gradient, and as ONE differentiable op -- taped forward + native reverse sweep
  (``autodiff.ComposedRolloutFunction``) -- when the policy's parameters or the initial state do (float64, frozen
  drift); False always takes the torch composition (``forward_sde`` over ``moment_matching``); True insists on the
  native path."""
  uniform = solution_times is None
  if solution_times is None:
    solution_times = np.arange(1, 1 + num_steps, dtype=np.float64)     # pilco.py:186
  encoder = system.encoder
  fast = None
  shape_reason = None
  if native is not False:
    if not uniform or float(initial_time) != 0.0:
      shape_reason = "non-uniform solution times (the native rollout takes unit steps from t = 0)"
    elif kwargs:
      shape_reason = f"solver options {sorted(kwargs)}"
    else:
      why_not = []
      fast = native_policy_loss(system, objective, num_steps, dt=1.0, why=why_not)
      if fast is None:
        shape_reason = why_not[0] if why_not else "the system is not the shape mm_rollout_composed implements"
  if native is True and fast is None:
    raise ValueError("native=True: the system is not the shape mm_rollout_composed implements")

  def _accumulate_loss(t, state, loss):                                # pilco.py:199-205
    x = GaussianMoments(moments=state, centered=True)
    if encoder is not None:
      x = moment_matching(x, encoder).y
    return loss + objective(x=x, t=t)

  warned = []

  def _fallback(reason):
    """The torch composition is about to run on GPU tensors (60 x slower than the native path at cartpole sizes): say so, once."""
    if native is not False and not warned:
      warned.append(reason)
      warnings.warn(f"policy_loss_closure: falling back to the torch composition of the rollout ({reason})", RuntimeWarning,
                    stacklevel=3)

  def _use_native(mx, Sxx):
    if fast is None and mx.is_cuda:
      _fallback(shape_reason or "the system is not the shape mm_rollout_composed implements")
    if fast is None or not mx.is_cuda or mx.ndim != 2:
      return False
    models = [system.drift, getattr(getattr(system.policy, "model", None), "model", None)]
    trainable = any(t.requires_grad for m in models if m is not None for t in m._parameters())
    if torch.is_grad_enabled() and (trainable or mx.requires_grad or Sxx.requires_grad):
      return False                       # someone differentiates: the torch composition carries the autograd graph
    if trainable and torch.cuda.is_current_stream_capturing():
      # a captured graph must evaluate a trainable model FROM its parameters, not from a packed snapshot that goes
      # stale at the next optimiser step: only the policy can be re-packed inside the graph (run.from_parameters)
      return "from_parameters" if not any(t.requires_grad for t in system.drift._parameters()) else False
    return True

  def _use_native_grad(mx, Sxx):
    """Someone differentiates, and what is differentiated is what the native reverse sweep covers: the policy's
    parameters and / or the initial state, with a frozen drift (the tape is float64; a float32 state is cast up)."""
    if native is False or not mx.is_cuda or not torch.is_grad_enabled():
      return False
    if fast is None:
      _fallback(shape_reason or "the system is not the shape mm_rollout_composed implements")
      return False
    if mx.ndim != 2:
      _fallback(f"state of rank {mx.ndim} (the native path takes mx [B, nx])")
      return False
    why = fast.grad_obstacle(mx)
    if why is not None:
      _fallback(why)
      return False
    return True

  def _closure():                                                      # pilco.py:207-217
    mx, Sxx = state_initializer()
    use = _use_native(mx, Sxx)
    if use == "from_parameters":
      return fast.from_parameters(mx, Sxx)
    if use:
      return fast(mx, Sxx)
    if _use_native_grad(mx, Sxx):
      return fast.with_grad(mx, Sxx)
    loss = torch.zeros(mx.shape[:-1], dtype=mx.dtype, device=mx.device)
    _, loss = system.solve_forward(iterator="foldl", initial_time=initial_time,
                                   initial_state=(mx, Sxx), solution_times=solution_times,
                                   callbacks_and_initializers=((_accumulate_loss, loss),), **kwargs)
    return loss

  return _closure


class GraphedPolicyLoss:
  """The policy loss closure -- and its gradient w.r.t. the policy parameters -- captured once into HIP
  graphs (``torch.cuda.CUDAGraph``) and replayed.

  The reference traces the closure once under ``tf.function`` (pilco.py:219-220); here the equivalent
  is a graph capture: at cartpole sizes one composed rollout step is several hundred small kernels
  and the eager path is bound by the host launching them (measured: 10.7 ms/step forward+backward
  eager, 2.4 ms/step replayed; 1.5 -> 0.77 ms/step forward only).

  Everything the closure reads must live in fixed tensors: the state initializer has to return the SAME
  tensors on every call (``get_state_initializer`` does; write a new initial state into them with
  ``copy_``), parameters are updated in place (optimisers do), shapes are frozen.  Replays read the
  current values of the TRAINABLE models (they are evaluated from their parameters inside the graph);
  frozen models (the drift) enter through their packed snapshot -- rebuild the object after refitting them.
  ``loss()`` / ``loss_and_grad()`` return static tensors that the next replay
  overwrites; gradients are also left in ``p.grad`` (overwritten, not accumulated).  No host-side
  checks run inside a replay: a non-PD state shows up as nan in the loss and in the packed models'
  status words (``PackedModel.check_status``); a failed Kuu factorisation of the trainable policy poisons its
  factor with NaN inside the graph and is recorded in ``linalg.capture_status`` -- ``check()`` (synchronising)
  raises for either.
  """

  def __init__(self, closure: Callable, parameters: Sequence[torch.Tensor], warmup: int = 2):
    self.closure = closure
    self.parameters = [p for p in parameters if p.requires_grad]
    if not self.parameters:
      raise ValueError("GraphedPolicyLoss needs at least one parameter with requires_grad=True")
    dev = self.parameters[0].device
    side = torch.cuda.Stream(device=dev)                 # warm-up off the capturing stream (lazy init, caches)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
      for _ in range(max(1, warmup)):
        with torch.no_grad():
          closure()
        for p in self.parameters:
          p.grad = None
        closure().sum().backward()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    self._fwd = torch.cuda.CUDAGraph()
    with torch.no_grad():
      with torch.cuda.graph(self._fwd):
        self._loss_fwd = closure()
    for p in self.parameters:
      p.grad = None
    self._bwd = torch.cuda.CUDAGraph()
    with torch.cuda.graph(self._bwd):
      self._loss_bwd = closure()
      self._loss_bwd.sum().backward()
    self._grads = [p.grad for p in self.parameters]
    self._loss_bwd = self._loss_bwd.detach()
    self._device = dev

  def check(self):
    """Synchronising check of what the replays could not raise: in-graph factorisation failures."""
    from .linalg import check_capture_status
    check_capture_status(self._device)

  def loss(self) -> torch.Tensor:
    """Forward only: the per-batch-element loss [B]."""
    self._fwd.replay()
    return self._loss_fwd

  def loss_and_grad(self):
    """-> (loss [B], [d sum(loss) / d p for p in parameters]); the gradients are also in ``p.grad``."""
    self._bwd.replay()
    for p, g in zip(self.parameters, self._grads):
      p.grad = g
    return self._loss_bwd, self._grads


def pathwise_policy_loss_closure(system: DynamicalSystem, objective: Callable, state_initializer: Callable, num_steps: int,
                                 dt: float = 1.0, num_bases: int = 1024, paths=None, native: Optional[bool] = None,
                                 generator: Optional[torch.Generator] = None) -> Callable:
  """``PathwisePILCO._policy_loss_closure`` (gpflow_pilco/loops/pilco.py:263-298).  Returns ``closure() -> loss [S]``: the cost
  accumulated along one sample rollout per initial state -- per step encoder -> policy -> drift sample path -> Euler -> objective
  of the encoded state (tensor branch of ``forward_sde``, dynamics/forward_sde.py:23-31; ``Euler.step``, solvers.py:50-65).  The
  caller takes the mean and differentiates it w.r.t. the policy (examples/cartpole_swingup/train_utils.py:108-135).

  ``system.drift``: a ``pathwise.PathwiseSVGP``; ``state_initializer() -> x0 [S, nx]`` (pilco.py:300-303: ``p.sample([batch_size])``);
  ``paths``: None = new sample paths on every call (pilco.py:281-284), else a ``pathwise.Paths`` to reuse.  Unit-spaced solution
  times ``dt, 2 dt, ...`` (pilco.py:257: ``arange(1, 1 + num_steps)``).

  On the GPU, for the cartpole wiring (``_native_parts``) with drift inputs of dimension <= 8, the whole closure is the native
  rollout (csrc/mm_pathwise_policy.hip): forward only when nothing requires a gradient, else ONE differentiable op
  (``pathwise.PolicyRolloutFunction``: the stream pass also emits the paths' Jacobians, the reverse sweep is one kernel).
  Otherwise -- ``native=False``, another wiring, a gradient the native sweep does not cover -- the torch composition through
  ``DynamicalSystem.solve_forward`` runs, saying so once (the paths stay on the device and are differentiable in x)."""
  from . import ops
  from .pathwise import PathwiseSVGP, PolicyRollout, PolicyRolloutFunction
  drift = system.drift
  if not isinstance(drift, PathwiseSVGP):
    raise TypeError("pathwise_policy_loss_closure needs a PathwiseSVGP drift (gpflow_pilco/loops/pilco.py:230-236)")
  H = int(num_steps)
  why_not: list = []
  parts = None if native is False else _native_parts(system, objective, why_not, moment_solver=False)
  if native is True and parts is None:
    raise ValueError(f"native=True: {why_not[0] if why_not else 'the system is not the shape the native rollout implements'}")
  warned = []

  def _fallback(reason):
    if native is not False and not warned:
      warned.append(reason)
      warnings.warn(f"pathwise_policy_loss_closure: falling back to the torch composition of the rollout ({reason})",
                    RuntimeWarning, stacklevel=3)

  def _torch_loss(x0, pth):
    times = dt * np.arange(1, 1 + H, dtype=np.float64)
    enc = system.encoder

    def _accumulate_loss(t, state, loss):                              # pilco.py:272-275
      return loss + objective(x=state if enc is None else enc(state), t=t)
    loss0 = torch.zeros(x0.shape[:-1], dtype=x0.dtype, device=x0.device)
    solver = Euler()
    with drift.set_temporary_paths(pth):
      _, loss = solver(func=system.forward, initial_time=0.0, initial_state=x0, solution_times=times,
                       callbacks_and_initializers=((_accumulate_loss, loss0),), iterator="foldl")
    return loss

  def _closure():
    x0 = state_initializer()
    pth = paths if paths is not None else drift.generate_paths(x0.shape[0], num_bases, dtype=x0.dtype, device=x0.device,
                                                                generator=generator)
    if parts is None or not x0.is_cuda or x0.ndim != 2:
      if x0.is_cuda and native is not False:
        _fallback(why_not[0] if why_not else f"state of rank {x0.ndim}")
      return _torch_loss(x0, pth)
    enc, pm_, _, bj, head_constants = parts
    nx, na = x0.shape[-1], len(enc.active_dims)
    if nx + na + 1 > 8 or pm_.inducing_variable.inducing_variables[0].Z.shape[0] > 256:
      _fallback("drift inputs of dimension > 8 or a policy of more than 256 centres")
      return _torch_loss(x0, pth)
    outside = {"the policy head's Scale.scale": bj[0].scale, "the policy head's Shift.shift": bj[1].shift,
               "objective.target": objective.target, "objective.precis": objective.precis}
    grad = torch.is_grad_enabled()
    if grad:
      for name, t in outside.items():
        if isinstance(t, torch.Tensor) and t.requires_grad:
          _fallback(f"{name} requires a gradient (the native reverse sweep covers the policy SVGP's parameters and the initial states)")
          return _torch_loss(x0, pth)
    scale, shift = head_constants()
    pol_pack = pm_.packed(torch.float64, False, x0.device)
    roll = PolicyRollout(pth, pol_pack, nx=nx, active_dims=enc.active_dims, head_scale=scale, head_shift=shift,
                         target=objective.target, precis=objective.precis)
    needs = grad and (x0.requires_grad or any(t.requires_grad for t in pm_._parameters()))
    if not needs:
      with torch.no_grad():
        cost, _ = roll(x0, H, dt=dt)
      return cost.sum(0)
    Zp, lsp, varp, betap, _, mcp = pm_.precompute(x0.device)
    if mcp is None:
      mcp = torch.zeros(1, dtype=Zp.dtype, device=x0.device)
    return PolicyRolloutFunction.apply(x0, Zp, lsp, varp, betap, mcp, roll, H, dt).sum(1).to(x0.dtype)

  return _closure
