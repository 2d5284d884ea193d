"""Parameter containers for the moment-matching handlers.

Minimal torch equivalents of the gpflow / gpflow_pilco objects the hot path
reads (fields only; hyper-parameter fitting is out of scope):

* kernels: ``SquaredExponential`` and the multi-output ``SeparateIndependent`` /
  ``SharedIndependent`` / ``LinearCoregionalization`` (gpflow.kernels), read at
  ``gpflow_pilco/moment_matching/models.py:205-212,279-286,331-354``;
* inducing variables (gpflow.inducing_variables), read at ``utils/kernel_expectation.py:41-69``;
* mean functions ``Zero`` / ``Constant`` (``gpflow_pilco/models/mean_functions.py:24-38``);
* ``SVGP`` / ``GPR`` (``gpflow_pilco/models/svgp.py:32-45``, ``gpr.py:26-37``) and the wrappers
  ``KernelRegressor`` / ``InverseLinkWrapper`` (``gpflow_pilco/models/core.py:30-71``).

``model.precompute()`` does, once per (frozen) model, what the reference redoes every
call at ``moment_matching/models.py:216-235``: Kuu + jitter, its Cholesky, and from it
``beta = Kuu^-1 u`` and ``C = Kuu^-1 S Kuu^-1 - Kuu^-1`` (float64, torch/rocSOLVER).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch

from .linalg import cholesky

DEFAULT_JITTER = 1e-6  # gpflow.config.default_jitter()
DEFAULT_FLOAT = torch.float64  # gpflow.config.default_float()


def _as_param(x, device=None) -> torch.Tensor:
  t = x if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=DEFAULT_FLOAT)
  t = t.to(DEFAULT_FLOAT)
  return t.to(device) if device is not None else t


# ---------------------------------------------------------------------------
# kernels
# ---------------------------------------------------------------------------
class Kernel:
  pass


class SquaredExponential(Kernel):
  """gpflow.kernels.SquaredExponential: variance * exp(-0.5 |(x - x')/lengthscales|^2)."""

  def __init__(self, variance=1.0, lengthscales=1.0, active_dims: Optional[Sequence[int]] = None):
    self.variance = _as_param(variance)
    self.lengthscales = _as_param(lengthscales)
    self.active_dims = None if active_dims is None else tuple(int(i) for i in active_dims)

  @property
  def ard(self) -> bool:
    return self.lengthscales.ndim > 0

  def slice(self, X: torch.Tensor) -> torch.Tensor:
    if self.active_dims is None:
      return X
    return X[..., list(self.active_dims)]

  def slice_cov(self, cov: torch.Tensor) -> torch.Tensor:
    if self.active_dims is None:
      return cov
    idx = list(self.active_dims)
    return cov[..., idx, :][..., :, idx]

  def lengthscales_vector(self, ndims: int) -> torch.Tensor:
    ls = self.lengthscales
    return ls if ls.ndim > 0 else ls.expand(ndims)

  def K(self, X: torch.Tensor, X2: Optional[torch.Tensor] = None) -> torch.Tensor:
    X = self.slice(X)
    ls = self.lengthscales_vector(X.shape[-1]).to(X)
    A = X / ls
    B = A if X2 is None else self.slice(X2) / ls
    d2 = (A * A).sum(-1)[:, None] + (B * B).sum(-1)[None, :] - 2.0 * A @ B.T
    return self.variance.to(X) * torch.exp(-0.5 * d2.clamp_min(0.0))


class MultioutputKernel(Kernel):
  kernels: List[SquaredExponential]

  @property
  def num_latent_gps(self) -> int:
    return len(self.kernels)


class SeparateIndependent(MultioutputKernel):
  def __init__(self, kernels: Sequence[SquaredExponential]):
    self.kernels = list(kernels)


class SharedIndependent(MultioutputKernel):
  def __init__(self, kernel: SquaredExponential, output_dim: int):
    self.kernel = kernel
    self.kernels = [kernel] * output_dim


class LinearCoregionalization(MultioutputKernel):
  def __init__(self, kernels: Sequence[SquaredExponential], W):
    self.kernels = list(kernels)
    self.W = _as_param(W)  # [P, L]


# ---------------------------------------------------------------------------
# inducing variables, mean functions, likelihood
# ---------------------------------------------------------------------------
class InducingPoints:
  def __init__(self, Z):
    self.Z = _as_param(Z)


class SeparateIndependentInducingVariables:
  def __init__(self, inducing_variable_list: Sequence[InducingPoints]):
    self.inducing_variables = list(inducing_variable_list)


class SharedIndependentInducingVariables:
  def __init__(self, inducing_variable: InducingPoints):
    self.inducing_variable = inducing_variable
    self.inducing_variables = [inducing_variable]


class Zero:
  def __call__(self, X):
    return torch.zeros_like(X[..., :1])


class Constant:
  def __init__(self, c):
    self.c = _as_param(c).reshape(-1)

  def __call__(self, X):
    return self.c.to(X).expand(X.shape[:-1] + self.c.shape)


class GaussianLikelihood:
  def __init__(self, variance=1.0):
    self.variance = _as_param(variance)


def unpack_multioutput(kernel, inducing_variable, num_latent: Optional[int] = None):
  """``unpack_multioutput`` (utils/kernel_expectation.py:41-69) -> (kernels, [Z_a])."""
  if isinstance(kernel, MultioutputKernel):
    kernels = list(kernel.kernels)
  else:
    kernels = [kernel] * (num_latent or 1)
  L = len(kernels)
  if isinstance(inducing_variable, SeparateIndependentInducingVariables):
    ivs = list(inducing_variable.inducing_variables)
    assert len(ivs) == L
  elif isinstance(inducing_variable, SharedIndependentInducingVariables):
    ivs = L * list(inducing_variable.inducing_variables)
  elif isinstance(inducing_variable, InducingPoints):
    ivs = L * [inducing_variable]
  else:
    raise NotImplementedError(type(inducing_variable))
  return kernels, [iv.Z for iv in ivs]


# ---------------------------------------------------------------------------
# models
# ---------------------------------------------------------------------------
class _PackCache:
  """Caches the precompute and the packed device model per (dtype, with_C); keyed on the
  parameter tensors' versions so an in-place update (training step) invalidates it."""

  def __init__(self):
    self._key = None
    self._pre = None
    self._packed = {}

  def get(self, model, dtype, with_C, device):
    key = tuple((id(t), t._version, t.device) for t in model._parameters()) + (str(device),)
    if key != self._key:
      self._key, self._pre, self._packed = key, None, {}
    if self._pre is None:
      self._pre = model.precompute(device)
    slot = (dtype, bool(with_C))
    if slot not in self._packed:
      from . import ops
      Z, ls, var, beta, C, mean_c = self._pre
      capturing = Z.is_cuda and torch.cuda.is_current_stream_capturing()
      det = lambda t: None if t is None else t.detach()
      self._packed[slot] = ops.pack_model(det(Z), det(ls), det(var), det(beta), det(C) if with_C else None, det(mean_c),
                                          dtype=dtype, sync=not capturing)
    return self._packed[slot]


# lengthscale given to an input dimension a latent kernel does NOT act on (per-latent ``active_dims``): the SE kernel
# with lengthscale l -> infinity on a dimension is the kernel that ignores it; 1e6 perturbs every expectation by
# O(Sigma_kk / 1e12) relative
INACTIVE_LENGTHSCALE = 1.0e6


def kernel_input_dims(kernels, ndims: Optional[int] = None):
  """Sorted union of the latent kernels' ``active_dims`` (None if every kernel acts on all inputs), and whether
  the kernels differ in what they act on."""
  if all(k.active_dims is None for k in kernels):
    return None, False
  if any(k.active_dims is None for k in kernels):
    if ndims is None:
      raise ValueError("mixing sliced and unsliced latent kernels needs the input dimension")
    union = tuple(range(ndims))
  else:
    union = tuple(sorted(set(i for k in kernels for i in k.active_dims)))
  differ = any((k.active_dims if k.active_dims is not None else union) != kernels[0].active_dims for k in kernels) \
      or kernels[0].active_dims is None
  return union, differ


def _stack_kernel_params(kernels, Zs, device):
  """-> Z [L,M,d], lengthscales [L,d], variance [L] on the kernels' common input dimensions.

  Kernels with the same ``active_dims`` (the reference's assumption, utils/kernel_expectation.py:98-100): sliced
  as the reference slices.  Kernels acting on DIFFERENT subsets (moment_matching/models.py:264-270 slices per
  kernel): every latent is embedded into the sorted union of the active dimensions with the lengthscale
  ``INACTIVE_LENGTHSCALE`` (and inducing coordinate 0) on the dimensions it ignores -- the kernels, their
  expectations under any (dense or diagonal) Gaussian and all pair terms are then exact to ~1e-12, including the
  product shortcut of disjoint kernels under a diagonal Gaussian (utils/kernel_expectation.py:85-89), which is the
  special case G = 0 of the general pair term."""
  union, differ = kernel_input_dims(kernels, Zs[0].shape[-1])
  if not differ:
    d = Zs[0].shape[-1] if kernels[0].active_dims is None else len(kernels[0].active_dims)
    Z = torch.stack([k.slice(z.to(device=device, dtype=DEFAULT_FLOAT)) for k, z in zip(kernels, Zs)])
    ls = torch.stack([k.lengthscales_vector(d).to(device=device, dtype=DEFAULT_FLOAT) for k in kernels])
  else:
    d = len(union)
    pos = {u: i for i, u in enumerate(union)}
    Zl, lsl = [], []
    for k, z in zip(kernels, Zs):
      act = union if k.active_dims is None else k.active_dims
      zf = torch.zeros(z.shape[0], d, dtype=DEFAULT_FLOAT, device=device)
      lf = torch.full((d,), INACTIVE_LENGTHSCALE, dtype=DEFAULT_FLOAT, device=device)
      idx = torch.tensor([pos[u] for u in act], device=device)
      zf[:, idx] = k.slice(z.to(device=device, dtype=DEFAULT_FLOAT))
      lf[idx] = k.lengthscales_vector(len(act)).to(device=device, dtype=DEFAULT_FLOAT)
      Zl.append(zf); lsl.append(lf)
    Z, ls = torch.stack(Zl), torch.stack(lsl)
  var = torch.stack([k.variance.to(device=device, dtype=DEFAULT_FLOAT).reshape(()) for k in kernels])
  return Z, ls, var


class SVGP:
  """``gpflow_pilco.models.SVGP`` fields: kernel, inducing_variable, q_mu [M,L], q_sqrt [L,M,M],
  whiten, mean_function, likelihood, num_latent_gps."""

  def __init__(self, kernel, inducing_variable, q_mu, q_sqrt, whiten: bool = True,
               mean_function=None, likelihood=None, num_latent_gps: Optional[int] = None,
               prior: Optional[Callable] = None):
    self.kernel = kernel
    self.inducing_variable = (InducingPoints(inducing_variable)
                              if isinstance(inducing_variable, torch.Tensor) else inducing_variable)
    self.q_mu = _as_param(q_mu)
    self.q_sqrt = _as_param(q_sqrt)
    self.whiten = bool(whiten)
    self.mean_function = Zero() if mean_function is None else mean_function
    self.likelihood = likelihood
    self.num_latent_gps = num_latent_gps or self.q_mu.shape[-1]
    self.prior = prior
    self._cache = _PackCache()

  def predict_mean(self, x: torch.Tensor) -> torch.Tensor:
    """Posterior mean at deterministic inputs x [..., D]: K(x, Z) Kuu^-1 u (+ mean function), i.e. the
    mean half of gpflow's ``predict_f``; what ``KernelRegressor.__call__`` returns (models/core.py:60-62).
    Plain torch (policy evaluation on real states; not on the moment-matching hot path)."""
    Z, ls, var, beta, _, mean_c = self.precompute(x.device)
    kernels = self.latent_kernels
    union, differ = kernel_input_dims(kernels, x.shape[-1])
    xs = (x[..., list(union)] if differ else kernels[0].slice(x)).to(DEFAULT_FLOAT)
    lead = xs.shape[:-1]
    xs2 = xs.reshape(-1, xs.shape[-1])
    A = xs2[None] / ls[:, None, :]                              # [L, n, d]
    Bz = Z / ls[:, None, :]                                     # [L, M, d]
    d2 = (A * A).sum(-1)[:, :, None] + (Bz * Bz).sum(-1)[:, None, :] - 2.0 * A @ Bz.transpose(1, 2)
    Kxz = var[:, None, None] * torch.exp(-0.5 * d2.clamp_min(0.0))
    g = (Kxz @ beta.unsqueeze(-1)).squeeze(-1).T                # [n, L]
    if isinstance(self.kernel, LinearCoregionalization):
      g = g @ self.kernel.W.to(g).T
      if isinstance(self.mean_function, Constant):
        g = g + self.mean_function.c.to(g)
    elif mean_c is not None:
      g = g + mean_c
    return g.reshape(lead + g.shape[-1:]).to(x.dtype)

  def __call__(self, x, **kwargs):
    if isinstance(x, torch.Tensor):
      return self.predict_mean(x)
    raise NotImplementedError("SVGP.__call__ takes a tensor of inputs")

  # -- helpers used by the handlers ---------------------------------------
  def _parameters(self):
    kernels, Zs = unpack_multioutput(self.kernel, self.inducing_variable, self.num_latent_gps)
    out = [self.q_mu, self.q_sqrt] + list(Zs)
    for k in kernels:
      out += [k.variance, k.lengthscales]
    if isinstance(self.mean_function, Constant):
      out.append(self.mean_function.c)
    return out

  @property
  def latent_kernels(self):
    return unpack_multioutput(self.kernel, self.inducing_variable, self.num_latent_gps)[0]

  def precompute(self, device):
    """-> Z [L,M,d], ls [L,d], var [L], beta [L,M], C [L,M,M], mean_c [L]|None (float64, device)."""
    kernels, Zs = unpack_multioutput(self.kernel, self.inducing_variable, self.num_latent_gps)
    Z, ls, var = _stack_kernel_params(kernels, Zs, device)
    L, M, d = Z.shape
    A = Z / ls[:, None, :]
    d2 = (A * A).sum(-1)[:, :, None] + (A * A).sum(-1)[:, None, :] - 2.0 * A @ A.transpose(1, 2)
    Kuu = var[:, None, None] * torch.exp(-0.5 * d2.clamp_min(0.0))
    Kuu = Kuu + DEFAULT_JITTER * torch.eye(M, dtype=DEFAULT_FLOAT, device=device)  # models.py:216
    Luu = cholesky(Kuu)                                                # :217
    v = self.q_mu.to(device=device, dtype=DEFAULT_FLOAT).T.unsqueeze(-1)            # [L,M,1]  :228
    S = torch.tril(self.q_sqrt.to(device=device, dtype=DEFAULT_FLOAT))              # :229
    if not self.whiten:                                                             # :230-232
      v = torch.linalg.solve_triangular(Luu, v, upper=False)
      S = torch.linalg.solve_triangular(Luu, S, upper=False)
    LuuT = Luu.transpose(1, 2)
    beta = torch.linalg.solve_triangular(LuuT, v, upper=True).squeeze(-1)           # :235
    Acov = S @ S.transpose(1, 2) - torch.eye(M, dtype=DEFAULT_FLOAT, device=device)
    X = torch.linalg.solve_triangular(LuuT, Acov, upper=True)                       # L^-T (A - I)
    C = torch.linalg.solve_triangular(LuuT, X.transpose(1, 2), upper=True)          # L^-T (A - I) L^-1
    C = 0.5 * (C + C.transpose(1, 2))
    mean_c = None
    if isinstance(self.mean_function, Constant):
      if isinstance(self.kernel, LinearCoregionalization):
        mean_c = None            # added after the W mixing, on the host
      else:
        mean_c = self.mean_function.c.to(device=device, dtype=DEFAULT_FLOAT).expand(L).contiguous()
    elif not isinstance(self.mean_function, Zero):
      raise NotImplementedError                                                     # :291
    return Z, ls, var, beta, C, mean_c

  def packed(self, dtype, with_C: bool, device):
    return self._cache.get(self, dtype, with_C, device)


class GPR:
  """``gpflow_pilco.models.GPR`` fields: kernel, data=(X, Y), likelihood.variance, mean_function."""

  def __init__(self, data: Tuple, kernel: SquaredExponential, mean_function=None,
               noise_variance=1.0, prior: Optional[Callable] = None):
    X, Y = data
    self.data = (_as_param(X), _as_param(Y))
    self.kernel = kernel
    self.mean_function = Zero() if mean_function is None else mean_function
    self.likelihood = GaussianLikelihood(noise_variance)
    self.prior = prior
    self._cache = _PackCache()

  def __call__(self, x, **kwargs):
    raise NotImplementedError("GPR.predict_f is outside the accelerated path")

  def _parameters(self):
    out = [self.data[0], self.data[1], self.kernel.variance, self.kernel.lengthscales,
           self.likelihood.variance]
    if isinstance(self.mean_function, Constant):
      out.append(self.mean_function.c)
    return out

  @property
  def latent_kernels(self):
    return [self.kernel]

  def precompute(self, device):
    X, Y = (t.to(device=device, dtype=DEFAULT_FLOAT) for t in self.data)
    if Y.shape[-1] != 1:
      raise NotImplementedError("GPR moment matching is single-output (models.py:44-111)")
    c = None
    if isinstance(self.mean_function, Constant):                                    # models.py:53-54
      c = self.mean_function.c.to(device=device, dtype=DEFAULT_FLOAT).reshape(1)
      Y = Y - c
    elif not isinstance(self.mean_function, Zero):
      raise NotImplementedError
    Z, ls, var = _stack_kernel_params([self.kernel], [X], device)
    N = X.shape[0]
    Kyy = self.kernel.K(X) + self.likelihood.variance.to(X) * torch.eye(N, dtype=DEFAULT_FLOAT, device=device)
    Lyy = cholesky(Kyy)                                                # :66-68
    beta = torch.cholesky_solve(Y, Lyy).T.contiguous()                              # [1,N]   :75
    C = -torch.cholesky_inverse(Lyy).unsqueeze(0)                                   # -(K + s2 I)^-1  (:86-88)
    return Z, ls, var, beta, C, c

  def packed(self, dtype, with_C: bool, device):
    return self._cache.get(self, dtype, with_C, device)


class GPModelWrapper:
  """``gpflow_pilco.models.core.GPModelWrapper``: attribute access falls through to the model."""

  def __init__(self, model, **attrs):
    self.__dict__["_model"] = model
    self.__dict__.update(attrs)

  def __getattr__(self, name):
    return getattr(self.__dict__["_model"], name)

  @property
  def model(self):
    return self.__dict__["_model"]


class KernelRegressor(GPModelWrapper):
  """Mean-only view of a model (no predictive uncertainty); models/core.py:60-62."""

  def __call__(self, x, **kwargs):
    return self.model.predict_mean(x)


class InverseLinkWrapper(GPModelWrapper):
  """model followed by an inverse link (bijector chain); models/core.py:65-71."""

  def __init__(self, model, invlink):
    super().__init__(model=model, invlink=invlink)

  def __call__(self, *args, **kwargs):
    return self.invlink(self.model(*args, **kwargs))
