"""Shared builders: the same numpy parameters go to the CPU oracle and to the GPU path."""
import numpy as np
import torch

from gpflowpilco_amd import models as gp
from gpflowpilco_amd.synthetic import SyntheticSVGP, generate_covariance, make_inputs, make_svgp
from oracle import mm_oracle as mo

F64 = torch.float64


def oracle_params(s: SyntheticSVGP) -> mo.SVGPParams:
  L, M, d = s.shape
  return mo.SVGPParams(Z=np.broadcast_to(s.Z, (L, M, d)).copy(), lengthscales=s.lengthscales,
                       variance=s.variance, q_mu=s.q_mu, q_sqrt=s.q_sqrt, whiten=s.whiten,
                       mean_c=s.mean_c)


def random_svgp_params(seed, L, M, d, whiten, ls_bounds=(0.3, 3.0), mean=True, W_rows=None,
                       separate_Z=True):
  """Reference-test style random SVGP (tests/test_moment_matching.py:199-236)."""
  rng = np.random.default_rng(seed)
  Z = rng.uniform(size=(L, M, d)) if separate_Z else np.broadcast_to(rng.uniform(size=(M, d)), (L, M, d)).copy()
  ls = np.exp(rng.uniform(np.log(ls_bounds[0]), np.log(ls_bounds[1]), size=(L, d)))
  q_mu = 0.89 * rng.standard_normal((M, L))
  q_cov = generate_covariance(rng, M, (L,), 0.89)
  W = None
  P = L
  if W_rows is not None:
    W = rng.uniform(size=(W_rows, L))
    W = W / np.linalg.norm(W, axis=-1, keepdims=True)
    P = W_rows
  mean_c = 1 + rng.standard_normal(P) if mean else None
  return mo.SVGPParams(Z=Z, lengthscales=ls, variance=np.full(L, 0.89 ** 2), q_mu=q_mu,
                       q_sqrt=np.linalg.cholesky(q_cov), whiten=whiten, mean_c=mean_c, W=W)


def gp_model_from_oracle(p: mo.SVGPParams, device) -> gp.SVGP:
  L = p.Z.shape[0]
  t = lambda a: torch.tensor(np.asarray(a), dtype=F64, device=device)
  kernels = [gp.SquaredExponential(variance=t(p.variance[a]), lengthscales=t(p.lengthscales[a])) for a in range(L)]
  iv = gp.SeparateIndependentInducingVariables([gp.InducingPoints(t(p.Z[a])) for a in range(L)])
  kernel = gp.LinearCoregionalization(kernels, t(p.W)) if p.W is not None else gp.SeparateIndependent(kernels)
  mean = gp.Zero() if p.mean_c is None else gp.Constant(t(p.mean_c))
  return gp.SVGP(kernel=kernel, inducing_variable=iv, q_mu=t(p.q_mu), q_sqrt=t(p.q_sqrt),
                 whiten=p.whiten, mean_function=mean, num_latent_gps=L)


def to_dev(a, device, dtype):
  return torch.tensor(np.asarray(a), dtype=dtype, device=device)


def contract_err(Sff, Sffo):
  """The f32 pack's accuracy contract (include/gpflowpilco_mm.h): per batch element the error of the
  off-diagonal block relative to that block's own scale (its largest |entry|: what MM_ROUTE_TOL is stated against) and of the
  diagonal relative to the largest variance.  Returns (worst off-diagonal, worst diagonal) over the batch."""
  got = Sff.detach().double().cpu().numpy() if isinstance(Sff, torch.Tensor) else np.asarray(Sff)
  L = got.shape[-1]
  eye = np.eye(L, dtype=bool)
  err = np.abs(got - Sffo)
  off = dia = 0.0
  for b in range(got.shape[0]):
    if L > 1:
      off = max(off, float(err[b][~eye].max() / max(np.abs(Sffo[b][~eye]).max(), 1e-300)))
    dia = max(dia, float(err[b][eye].max() / max(np.abs(Sffo[b][eye]).max(), 1e-300)))
  return off, dia


def f32_state(mu, Sigma):
  """The state as an f32 pack receives it (the oracle is evaluated there: the comparison point of an f32 tolerance)."""
  return np.asarray(mu, dtype=np.float32).astype(np.float64), np.asarray(Sigma, dtype=np.float32).astype(np.float64)


def scale_err(got, want):
  """max |got - want| relative to max |want| (per tensor)."""
  got = got.detach().double().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
  return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-300))
