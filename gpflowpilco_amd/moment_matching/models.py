"""Moment-matching handlers for GP models: the drop-in boundary of the HIP kernels.

Registered on ``dispatcher`` with the reference's signature
``(x: GaussianMoments, model, /, full_output_cov=True, model_uncertainty=True, jitter=0.0)
-> GaussianMatch`` (``gpflow_pilco/moment_matching/models.py:44-50,114-133,200-204``) and
return contract ``GaussianMatch(x, y=GaussianMoments((f1, Sff), centered=True),
cross=(iSxx_Sxf, True))`` (``:110-111,196-197,298-299``).

The arithmetic of ``_mm_gauss_svgp_mo`` / ``_so`` / ``_mm_gauss_gpr`` runs in
``libgpflowpilco_mm.so``; only slicing, the LinearCoregionalization mixing
(``:279-286``) and wrapper plumbing stay here.
"""
from __future__ import annotations

from functools import partial

import torch

from .. import ops
from ..models import (GPR, SVGP, Constant, InverseLinkWrapper, KernelRegressor,
                      LinearCoregionalization, Zero)
from .core import Chain, LinearOperatorDiag, dispatcher
from .gaussian import GaussianMatch, GaussianMoments


@dispatcher.register(GaussianMoments, InverseLinkWrapper)
def _mm_gauss_invlink(x: GaussianMoments, wrapper: InverseLinkWrapper, /, **kw):
  """models.py:27-31."""
  base = partial(wrapper.model, **kw)
  chain = Chain(wrapper.invlink, base)
  return dispatcher(x, chain)


@dispatcher.register(GaussianMoments, KernelRegressor)
def _mm_gauss_kr(x: GaussianMoments, regressor: KernelRegressor, /, **kwargs):
  """models.py:34-41."""
  uncertainty = kwargs.pop("model_uncertainty", False)
  assert not uncertainty, ValueError("Kernel regressors have no uncertainty.")
  return dispatcher(x, regressor.model, model_uncertainty=False, **kwargs)


def _sliced_state(x: GaussianMoments, kernels):
  """The state on the latent kernels' common input dimensions (models.py:264-270 slices per kernel; latents that act
  on different subsets are embedded into the union of them, gpflowpilco_amd/models.py:_stack_kernel_params).
  Returns (mu, Sxx, union | None)."""
  from ..models import kernel_input_dims
  mu, Sxx = x.mean(), x.covariance(dense=True)
  union, differ = kernel_input_dims(kernels, mu.shape[-1])
  if not differ:
    return kernels[0].slice(mu).contiguous(), kernels[0].slice_cov(Sxx).contiguous(), None
  idx = list(union)
  return mu[..., idx].contiguous(), Sxx[..., idx, :][..., :, idx].contiguous(), union


def _run_kernels(x: GaussianMoments, model, full_output_cov, model_uncertainty, jitter,
                 latent_full_cov=None):
  kernels = model.latent_kernels
  mu, Sxx, union = _sliced_state(x, kernels)
  if not mu.is_cuda:
    raise RuntimeError("moment matching of GP models runs on the GPU only (no CPU fallback)")
  lead = mu.shape[:-1]
  d = mu.shape[-1]
  mu2, S2 = mu.reshape(-1, d), Sxx.reshape(-1, d, d)
  full = full_output_cov if latent_full_cov is None else latent_full_cov
  grad_on = torch.is_grad_enabled()
  trainable = any(t.requires_grad for t in model._parameters())
  # inside a HIP-graph capture a trainable model must be evaluated FROM its parameters: the packed
  # device model is a snapshot taken outside the graph and would go stale at the next optimiser step
  capturing = mu.is_cuda and torch.cuda.is_current_stream_capturing()
  if trainable and (grad_on or capturing):
    # a model whose parameters are being trained (the policy): fully differentiable torch evaluation
    from ..autodiff import moment_match_torch
    Z, ls, var, beta, C, mean_c = model.precompute(mu.device)
    f1, Sff, cross = moment_match_torch(mu2.to(Z.dtype), S2.to(Z.dtype), Z, ls, var, beta,
                                        C if model_uncertainty else None, mean_c, full, bool(model_uncertainty))
    f1, Sff, cross = f1.to(mu.dtype), Sff.to(mu.dtype), cross.to(mu.dtype)
  elif grad_on and (mu2.requires_grad or S2.requires_grad):
    # frozen model, differentiable inputs (the drift during a policy update): HIP forward + backward
    from ..autodiff import moment_match_differentiable
    f1, Sff, cross = moment_match_differentiable(model, mu2, S2, full, bool(model_uncertainty))
  else:
    pm = model.packed(dtype=mu.dtype, with_C=bool(model_uncertainty), device=mu.device)
    f1, Sff, cross = ops.moment_match(pm, mu2, S2, full_output_cov=full,
                                      model_uncertainty=model_uncertainty, jitter=0.0)
  if union is not None:
    # the reference stacks every latent's cross term in that latent's OWN sliced coordinates (models.py:264-277:
    # x1, dX and Sxx are sliced per kernel and stacked, which needs equally many active dims per latent); the kernels
    # return it on the union of the dims (zero, to ~1e-12, where a latent does not act): gather column a at a's dims
    pos = {u: i for i, u in enumerate(union)}
    acts = [union if k.active_dims is None else k.active_dims for k in kernels]
    if len(set(len(a) for a in acts)) != 1:
      raise NotImplementedError("latent kernels with different NUMBERS of active dims cannot be stacked (models.py:264-270)")
    gather = torch.tensor([[pos[u] for u in a] for a in acts], device=cross.device)        # [L, D]
    cross = torch.stack([cross[:, gather[a], a] for a in range(len(acts))], dim=-1)        # [B, D, L]
  f1 = f1.reshape(lead + f1.shape[1:])
  Sff = Sff.reshape(lead + Sff.shape[1:])
  cross = cross.reshape(lead + cross.shape[1:])
  return f1, Sff, cross


def _finish(x, f1, Sff, cross, full_output_cov, jitter):
  if full_output_cov:                                            # models.py:293-296
    if jitter:
      Sff = Sff + jitter * torch.eye(Sff.shape[-1], dtype=Sff.dtype, device=Sff.device)
  else:
    Sff = LinearOperatorDiag(Sff + jitter)
  y = GaussianMoments(moments=(f1, Sff), centered=True)
  return GaussianMatch(x=x, y=y, cross=(cross, True))


@dispatcher.register(GaussianMoments, GPR)
def _mm_gauss_gpr(x: GaussianMoments, model: GPR, /, full_output_cov: bool = True,
                  model_uncertainty: bool = True, jitter: float = 0.0):
  """models.py:44-111."""
  f1, Sff, cross = _run_kernels(x, model, full_output_cov, model_uncertainty, jitter)
  return _finish(x, f1, Sff, cross, full_output_cov, jitter)


@dispatcher.register(GaussianMoments, SVGP)
def _mm_gauss_svgp(x: GaussianMoments, model: SVGP, /, full_output_cov: bool = True,
                   model_uncertainty: bool = True, jitter: float = 0.0):
  """models.py:114-126 (dispatch), :129-197 (single output), :200-299 (multi output)."""
  kernel = model.kernel
  if isinstance(kernel, LinearCoregionalization):
    # the latent covariance stays full (models.py:244,258), then f = W g  (:279-286)
    f1, Sff, cross = _run_kernels(x, model, full_output_cov, model_uncertainty, jitter,
                                  latent_full_cov=True)
    W = kernel.W.to(f1)
    f1 = f1 @ W.T
    cross = cross @ W.T
    if full_output_cov:
      Sff = W @ Sff @ W.T
    else:
      Sff = (W * (W @ Sff.transpose(-1, -2))).sum(-1)
    if isinstance(model.mean_function, Constant):                # :288-289
      f1 = f1 + model.mean_function.c.to(f1)
    elif not isinstance(model.mean_function, Zero):
      raise NotImplementedError
    return _finish(x, f1, Sff, cross, full_output_cov, jitter)
  f1, Sff, cross = _run_kernels(x, model, full_output_cov, model_uncertainty, jitter)
  return _finish(x, f1, Sff, cross, full_output_cov, jitter)
