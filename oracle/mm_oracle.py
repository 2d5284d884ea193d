"""fp64 CPU restatement of GPflowPILCO's moment-matched GP propagation.

TEST INFRASTRUCTURE ONLY -- the product path (``gpflowpilco_amd``) never imports
this module.  It is the checker for the HIP kernels and the timed "reference
CPU path" of ``bench.py``.

Parity status: the reference is pure TensorFlow/GPflow and neither is installed
here, so it cannot be executed to capture golden vectors, and its own tests hold
none (they are 10^6-sample Monte-Carlo checks at 1e-2 absolute,
``/root/reference/tests/utils.py:43-44,66-67``).  This oracle is therefore
pinned the way the reference pins itself: ``oracle/pin_oracle.py`` re-runs the
reference's three Monte-Carlo test designs
(``tests/test_kernel_expectation.py:51-93``, ``tests/test_moment_matching.py:88-264``)
against this file, plus Gauss-Hermite / Sigma->0 / GPR==SVGP identities.  Digits
beyond the Monte-Carlo tolerance are unpinned by the reference ("parity
unpinned" for whiten=True, SeparateIndependent, model_uncertainty=False and the
Euler update, which no reference test touches -- SURVEY.md section 8c).

Every function follows the reference line by line (materialises eKuffu
[B,L,M,L,M], does the two triangular solves) and cites it.  Third-party
arithmetic (gpflow>=2.2.1, un-vendored: ``/root/reference/setup.py:4``) is
restated from GPflow's published SE expectations.

Shapes: B input distributions, d input dims, L latent GPs, M inducing points.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
from scipy.linalg import solve_triangular

DEFAULT_JITTER = 1e-6  # gpflow.config.default_jitter(); used at models.py:145,216


# ---------------------------------------------------------------------------
# parameter containers (fields of gpflow.models.SVGP / GPR that the path reads)
# ---------------------------------------------------------------------------
@dataclass
class SVGPParams:
  """Fields read by ``_mm_gauss_svgp_mo`` / ``_so`` (moment_matching/models.py:129-299)."""
  Z: np.ndarray            # [L, M, d] inducing inputs per latent GP
  lengthscales: np.ndarray  # [L, d]
  variance: np.ndarray     # [L]
  q_mu: np.ndarray         # [M, L]
  q_sqrt: np.ndarray       # [L, M, M] lower triangular
  whiten: bool = True
  mean_c: Optional[np.ndarray] = None  # Constant mean [P]; None = Zero
  W: Optional[np.ndarray] = None       # LinearCoregionalization mixing [P, L]
  kuu_jitter: float = DEFAULT_JITTER
  shared_kernel: bool = False          # SharedIndependent: one kernel object for all latents


@dataclass
class GPRParams:
  """Fields read by ``_mm_gauss_gpr`` (moment_matching/models.py:44-111)."""
  X: np.ndarray            # [N, d]
  Y: np.ndarray            # [N, 1]
  lengthscales: np.ndarray  # [d]
  variance: float
  noise_variance: float
  mean_c: Optional[float] = None


# ---------------------------------------------------------------------------
# kernels and kernel expectations
# ---------------------------------------------------------------------------
def se_kernel(X, X2, lengthscales, variance):
  """gpflow.kernels.SquaredExponential.K (third-party; published formula)."""
  A = X / lengthscales
  B = A if X2 is None else X2 / lengthscales
  d2 = (np.sum(A * A, -1)[:, None] + np.sum(B * B, -1)[None, :]
        - 2.0 * A @ B.T)
  return variance * np.exp(-0.5 * np.maximum(d2, 0.0))


def _batched_cholesky(S):
  return np.linalg.cholesky(S)


def _tri_solve_batched(L, rhs):
  """Solve L[n] X[n] = rhs[n] for lower-triangular L: [N,d,d], rhs [N,d,K]."""
  out = np.empty_like(rhs)
  for n in range(L.shape[0]):
    out[n] = solve_triangular(L[n], rhs[n], lower=True, check_finite=False)
  return out


def eKff_se(mu, variance):
  """<k(x,x)> for SE = kernel.variance (gpflow.expectations, K_diag branch)."""
  return np.full(mu.shape[:1], float(variance))


def eKfu_se(mu, Sigma, Z, lengthscales, variance):
  """<k(x, Z)>_{N(mu,Sigma)} for SE-ARD -> [B, M].

  Restates gpflow.expectations (Gaussian, SquaredExponential, InducingPoints):
  chol(Lambda + Sigma), triangular solve of (Z - mu), exp(-0.5 maha) scaled by
  prod(ls)/sqrt|Lambda+Sigma|.  Call sites: moment_matching/models.py:62,141,212.
  """
  B, d = mu.shape
  chol = _batched_cholesky(np.diag(lengthscales ** 2)[None] + Sigma)  # [B,d,d]
  diffs = Z.T[None] - mu[:, :, None]                                  # [B,d,M]
  sol = _tri_solve_batched(chol, diffs)
  maha = np.sum(sol * sol, axis=1)                                    # [B,M]
  sqrt_det_L = np.prod(lengthscales)
  sqrt_det_LS = np.exp(np.sum(np.log(np.diagonal(chol, axis1=1, axis2=2)), axis=1))
  return variance * (sqrt_det_L / sqrt_det_LS)[:, None] * np.exp(-0.5 * maha)


def eKuffu_se_pair(mu, Sigma, ls1, var1, Z1, ls2, var2, Z2,
                   is_same_kern: bool, is_same_feat: bool):
  """<k1(Z1,x) k2(x,Z2)> -> [B, M1, M2]; utils/kernel_expectation.py:72-187."""
  N, D = mu.shape
  V1 = ls1 ** 2                                   # :109
  z1 = Z1
  iV1_z1 = (1.0 / V1) * z1                        # :111
  V2 = V1 if is_same_kern else ls2 ** 2           # :114
  z2 = z1 if is_same_feat else Z2
  iV2_z2 = iV1_z1 if (is_same_kern and is_same_feat) else (1.0 / V2) * z2

  V = 0.5 * V1 if is_same_kern else (V1 * V2) / (V1 + V2)   # :119

  S = Sigma + np.diag(V)[None]                    # :125
  L = _batched_cholesky(S)
  half_logdet_L = np.sum(np.log(np.diagonal(L, axis1=1, axis2=2)), axis=1)
  sqrt_det_iL = np.exp(-half_logdet_L)
  sqrt_det_L = np.sqrt(np.prod(V))
  determinant = sqrt_det_L * sqrt_det_iL          # :130

  iL_mu = _tri_solve_batched(L, mu[:, :, None])   # [N,D,1]   :134
  V_iV1_z1 = np.broadcast_to((V * iV1_z1).T[None], (N, D, z1.shape[0]))
  iL_z1 = _tri_solve_batched(L, np.ascontiguousarray(V_iV1_z1))   # [N,D,M1]  :139

  z1_iS_z1 = np.sum(iL_z1 ** 2, axis=1)           # [N,M1]
  z1_iS_mu = np.squeeze(np.matmul(np.swapaxes(iL_z1, 1, 2), iL_mu), 2)
  if is_same_kern and is_same_feat:
    iL_z2, z2_iS_z2, z2_iS_mu = iL_z1, z1_iS_z1, z1_iS_mu
  else:
    V_iV2_z2 = np.broadcast_to((V * iV2_z2).T[None], (N, D, z2.shape[0]))
    iL_z2 = _tri_solve_batched(L, np.ascontiguousarray(V_iV2_z2))
    z2_iS_z2 = np.sum(iL_z2 ** 2, axis=1)
    z2_iS_mu = np.squeeze(np.matmul(np.swapaxes(iL_z2, 1, 2), iL_mu), 2)

  z1_iS_z2 = np.matmul(np.swapaxes(iL_z1, 1, 2), iL_z2)   # [N,M1,M2]  :157
  mu_iS_mu = np.sum(iL_mu ** 2, axis=1)[:, :, None]       # [N,1,1]

  exp_mahalanobis = np.exp(-0.5 * (mu_iS_mu + 2.0 * z1_iS_z2
                                   + (z1_iS_z1 - 2.0 * z1_iS_mu)[:, :, None]
                                   + (z2_iS_z2 - 2.0 * z2_iS_mu)[:, None, :]))  # :161-164

  if is_same_kern:                                # :167-174
    ampl2 = var1 ** 2
    sq_iV = 1.0 / np.sqrt(V)
    a = sq_iV * z1
    b = a if is_same_feat else sq_iV * z2
    d2 = (np.sum(a * a, -1)[:, None] + np.sum(b * b, -1)[None, :] - 2.0 * a @ b.T)
    matrix_term = ampl2 * np.exp(-0.125 * np.maximum(d2, 0.0))
  else:                                           # :175-185
    z1_iV1_z1 = np.sum(z1 * iV1_z1, axis=-1)
    z2_iV2_z2 = np.sum(z2 * iV2_z2, axis=-1)
    z1_iV1pV2_z1 = np.sum(iV1_z1 * V * iV1_z1, axis=-1)
    z2_iV1pV2_z2 = np.sum(iV2_z2 * V * iV2_z2, axis=-1)
    z1_iV1pV2_z2 = iV1_z1 @ (V * iV2_z2).T
    matrix_term = var1 * var2 * np.exp(0.5 * (
        2.0 * z1_iV1pV2_z2
        + (z1_iV1pV2_z1 - z1_iV1_z1)[:, None]
        + (z2_iV1pV2_z2 - z2_iV2_z2)[None, :]))

  return determinant[:, None, None] * matrix_term[None] * exp_mahalanobis  # :187


def eKuffu_se_pair_separate_dims(mu, var_diag, dims1, ls1, var1, Z1, dims2, ls2, var2, Z2):
  """The shortcut of utils/kernel_expectation.py:85-89: two kernels on DISJOINT input dims under a DiagonalGaussian
  (mean mu [B,D], variances var_diag [B,D]) need no joint expectation --
  <k1(Z1,x) k2(x,Z2)> = <k1(x,Z1)> (x) <k2(x,Z2)>  ->  [B, M1, M2].
  Z1 [M1, len(dims1)], Z2 [M2, len(dims2)] are already sliced to their kernels' active dims."""
  assert not set(dims1) & set(dims2), "kern1.on_separate_dims(kern2)"
  d1, d2 = list(dims1), list(dims2)
  S1 = var_diag[:, d1, None] * np.eye(len(d1))[None]
  S2 = var_diag[:, d2, None] * np.eye(len(d2))[None]
  e1 = eKfu_se(mu[:, d1], S1, Z1, ls1, var1)                 # :86
  e2 = eKfu_se(mu[:, d2], S2, Z2, ls2, var2)                 # :87
  return e1[:, :, None] * e2[:, None, :]                     # :88


def eKff_list(mu, variances):
  """_eKff fan-out -> [B, L]; utils/kernel_expectation.py:190-197."""
  return np.stack([eKff_se(mu, v) for v in variances], axis=-1)


def eKfu_list(mu, Sigma, Z, lengthscales, variances):
  """_eKfu fan-out -> [B, M, L]; utils/kernel_expectation.py:200-214."""
  return np.stack([eKfu_se(mu, Sigma, Z[a], lengthscales[a], variances[a])
                   for a in range(Z.shape[0])], axis=-1)


def eKuffu_list(mu, Sigma, Z, lengthscales, variances, shared_kernel=False,
                out=None):
  """_eKuffu fan-out -> [B, L, M, L, M]; utils/kernel_expectation.py:217-247.

  The reference computes the hash-ordered half and fills the rest with the
  adjoint (:238-244); hash order is arbitrary, here the half a <= a' is used.
  """
  L, M, _ = Z.shape
  B = mu.shape[0]
  if out is None:
    out = np.empty((B, L, M, L, M), dtype=np.float64)
  for a in range(L):
    for b in range(a, L):
      same = (a == b)
      same_kern = same or shared_kernel
      blk = eKuffu_se_pair(mu, Sigma,
                           lengthscales[a], variances[a], Z[a],
                           lengthscales[b], variances[b], Z[b],
                           is_same_kern=same_kern, is_same_feat=same)
      out[:, a, :, b, :] = blk
      if not same:
        out[:, b, :, a, :] = np.swapaxes(blk, 1, 2)
  return out


# ---------------------------------------------------------------------------
# moment matching handlers
# ---------------------------------------------------------------------------
def _cross_term(mu, Sigma, Z, lengthscales, eKfu, iK_u):
  """Sigma^{-1} Cov(x, f): models.py:263-277 (and :91-98, :176-186).

  Z [L,M,d], lengthscales [L,d], eKfu [B,M,L], iK_u [L,M] -> [B,d,L].
  """
  B, d = mu.shape
  L = Z.shape[0]
  out = np.empty((B, d, L))
  for a in range(L):
    dX = Z[a][None] - mu[:, None, :]                        # [B,M,d]
    V_sqrt = _batched_cholesky(Sigma + np.diag(lengthscales[a] ** 2)[None])
    y = _tri_solve_batched(V_sqrt, np.swapaxes(dX, 1, 2))   # [B,d,M]
    iV_dXt = np.empty_like(y)
    for n in range(B):                                      # cholesky_solve
      iV_dXt[n] = solve_triangular(V_sqrt[n].T, y[n], lower=False, check_finite=False)
    out[:, :, a] = np.sum(iK_u[a][None, None, :] * eKfu[:, None, :, a] * iV_dXt, axis=-1)
  return out


def mm_gauss_svgp_mo(mu, Sigma, model: SVGPParams, full_output_cov=True,
                     model_uncertainty=True, jitter=0.0):
  """``_mm_gauss_svgp_mo``: moment_matching/models.py:200-299.

  Returns (f1 [B,P], Sff [B,P,P] or [B,P] diag, iSxx_Sxf [B,d,P]); the cross
  term is pre-multiplied by Sigma^{-1} (``cross=(iSxx_Sxf, True)``, :298-299).
  """
  Z, ls, var = model.Z, model.lengthscales, model.variance
  L, M, d = Z.shape
  B = mu.shape[0]
  is_lcm = model.W is not None

  eKff = eKff_list(mu, var)                                   # [B,L]   :210
  eKfu = eKfu_list(mu, Sigma, Z, ls, var)                     # [B,M,L] :211
  eKuffu = eKuffu_list(mu, Sigma, Z, ls, var, model.shared_kernel)  # :212

  # Kuu + jitter, Cholesky (:216-217)
  Luu = np.empty((L, M, M))
  for a in range(L):
    Kuu = se_kernel(Z[a], None, ls[a], var[a]) + model.kuu_jitter * np.eye(M)
    Luu[a] = np.linalg.cholesky(Kuu)

  # W[b,a,i,a',j] = (L_a^{-1} Q_{aa'} L_{a'}^{-T})[i,j]   (:219-226)
  Wm = np.empty_like(eKuffu)
  for a in range(L):          # first solve along M1 with L_a
    rhs = np.moveaxis(eKuffu[:, a], 1, 0).reshape(M, -1)      # [M1, B*L2*M2]
    sol = solve_triangular(Luu[a], rhs, lower=True, check_finite=False)
    Wm[:, a] = np.moveaxis(sol.reshape(M, B, L, M), 0, 1)
  for a2 in range(L):         # then along M2 with L_{a'}
    rhs = np.moveaxis(Wm[:, :, :, a2, :], 3, 0).reshape(M, -1)  # [M2, B*L1*M1]
    sol = solve_triangular(Luu[a2], rhs, lower=True, check_finite=False)
    Wm[:, :, :, a2, :] = np.moveaxis(sol.reshape(M, B, L, M), 0, 3)

  iLuu_qmu = model.q_mu.T.copy()                              # [L,M]   :228
  iLuu_qsqrt = np.tril(model.q_sqrt)                          # [L,M,M] :229
  if not model.whiten:                                        # :230-232
    iLuu_qmu = np.stack([solve_triangular(Luu[a], iLuu_qmu[a], lower=True)
                         for a in range(L)])
    iLuu_qsqrt = np.stack([solve_triangular(Luu[a], iLuu_qsqrt[a], lower=True)
                           for a in range(L)])

  iKuu_qmu = np.stack([solve_triangular(Luu[a].T, iLuu_qmu[a], lower=False)
                       for a in range(L)])                    # [L,M]   :235
  f1 = np.einsum('bma,am->ba', eKfu, iKuu_qmu)                # :236

  if model_uncertainty or not full_output_cov:                # :239-241
    blkdiag = np.stack([Wm[:, a, :, a, :] for a in range(L)], axis=1)  # [B,L,M,M]

  if full_output_cov or is_lcm:                               # :244-248
    f2 = np.einsum('ai,baicj,cj->bac', iLuu_qmu, Wm, iLuu_qmu, optimize=True)
    Sff = f2 - f1[:, :, None] * f1[:, None, :]
  else:                                                       # :249-252
    Sff = np.einsum('ai,baij,aj->ba', iLuu_qmu, blkdiag, iLuu_qmu, optimize=True) - f1 ** 2

  if model_uncertainty:                                       # :254-261
    Li_qcov_Lit = iLuu_qsqrt @ np.swapaxes(iLuu_qsqrt, 1, 2)  # [L,M,M]
    trace = np.trace(blkdiag, axis1=2, axis2=3)               # [B,L]
    matmul = np.einsum('baij,aij->ba', blkdiag, Li_qcov_Lit)
    e_cov = eKff + matmul - trace
    if full_output_cov or is_lcm:
      idx = np.arange(L)
      Sff[:, idx, idx] += e_cov
    else:
      Sff = Sff + e_cov

  iSxx_Sxf = _cross_term(mu, Sigma, Z, ls, eKfu, iKuu_qmu)   # [B,d,L]  :263-277

  if is_lcm:                                                  # :279-286
    Wmix = model.W
    f1 = f1 @ Wmix.T
    iSxx_Sxf = iSxx_Sxf @ Wmix.T
    if full_output_cov:
      Sff = Wmix[None] @ Sff @ Wmix.T[None]
    else:
      Sff = np.sum(Wmix[None] * (Wmix[None] @ np.swapaxes(Sff, 1, 2)), axis=-1)

  if model.mean_c is not None:                                # :288-291
    f1 = f1 + np.asarray(model.mean_c)[None]

  if full_output_cov:                                         # :293-296
    idx = np.arange(Sff.shape[-1])
    Sff = Sff.copy()
    Sff[:, idx, idx] += jitter
  else:
    Sff = Sff + jitter
  return f1, Sff, iSxx_Sxf


def mm_gauss_svgp_so(mu, Sigma, model: SVGPParams, full_output_cov=True,
                     model_uncertainty=True, jitter=0.0):
  """``_mm_gauss_svgp_so``: moment_matching/models.py:129-197 (L == 1)."""
  assert model.Z.shape[0] == 1
  Z, ls, var = model.Z[0], model.lengthscales[0], float(model.variance[0])
  M = Z.shape[0]
  eKff = eKff_se(mu, var)
  eKfu = eKfu_se(mu, Sigma, Z, ls, var)                       # [B,M]
  eKuffu = eKuffu_se_pair(mu, Sigma, ls, var, Z, ls, var, Z, True, True)  # [B,M,M]

  Luu = np.linalg.cholesky(se_kernel(Z, None, ls, var) + model.kuu_jitter * np.eye(M))
  B = mu.shape[0]
  W = np.empty_like(eKuffu)
  for n in range(B):                                          # :147-148
    t = solve_triangular(Luu, eKuffu[n], lower=True, check_finite=False)
    W[n] = solve_triangular(Luu, t.T, lower=True, check_finite=False)

  iLuu_qmu = model.q_mu[:, :1].copy()                         # [M,1]
  iLuu_qsqrt = np.tril(model.q_sqrt[0])
  if not model.whiten:
    iLuu_qmu = solve_triangular(Luu, iLuu_qmu, lower=True)
    iLuu_qsqrt = solve_triangular(Luu, iLuu_qsqrt, lower=True)
  iKuu_qmu = solve_triangular(Luu.T, iLuu_qmu, lower=False)   # [M,1]
  f1 = eKfu @ iKuu_qmu                                        # [B,1]

  f2 = np.einsum('i,bij,j->b', iLuu_qmu[:, 0], W, iLuu_qmu[:, 0])
  if full_output_cov:
    Sff = (f2[:, None] - f1 ** 2)[:, :, None]                 # [B,1,1]
  else:
    Sff = f2[:, None] - f1 ** 2                               # [B,1]

  if model_uncertainty:                                       # :169-174
    Li_qcov_LiT = iLuu_qsqrt @ iLuu_qsqrt.T
    e_cov = eKff - np.trace(W, axis1=1, axis2=2) + np.sum(W * Li_qcov_LiT[None], axis=(1, 2))
    Sff = Sff + (e_cov[:, None, None] if full_output_cov else e_cov[:, None])

  iSxx_Sxf = _cross_term(mu, Sigma, Z[None], ls[None], eKfu[:, :, None], iKuu_qmu.T)

  if model.mean_c is not None:
    f1 = f1 + np.asarray(model.mean_c).reshape(1, -1)
  Sff = Sff + jitter            # [B,1,1] or [B,1]: set_diag on a 1x1 == add
  return f1, Sff, iSxx_Sxf


def mm_gauss_gpr(mu, Sigma, model: GPRParams, full_output_cov=True,
                 model_uncertainty=True, jitter=0.0):
  """``_mm_gauss_gpr``: moment_matching/models.py:44-111."""
  X, Y = model.X, model.Y
  ls, var = np.asarray(model.lengthscales, dtype=np.float64), float(model.variance)
  if model.mean_c is not None:
    Y = Y - model.mean_c                                      # :53-54
  N = X.shape[0]
  B = mu.shape[0]
  eKff = eKff_se(mu, var)
  eKfu = eKfu_se(mu, Sigma, X, ls, var)
  eKuffu = eKuffu_se_pair(mu, Sigma, ls, var, X, ls, var, X, True, True)

  Kyy = se_kernel(X, None, ls, var) + model.noise_variance * np.eye(N)   # :66-67
  Lyy = np.linalg.cholesky(Kyy)
  iLyy_y = solve_triangular(Lyy, Y, lower=True)               # [N,1]
  W = np.empty_like(eKuffu)
  for n in range(B):                                          # :71-72
    t = solve_triangular(Lyy, eKuffu[n], lower=True, check_finite=False)
    W[n] = solve_triangular(Lyy, t.T, lower=True, check_finite=False)
  iKyy_y = solve_triangular(Lyy.T, iLyy_y, lower=False)       # :75
  f1 = eKfu @ iKyy_y                                          # [B,1]

  f2 = np.einsum('i,bij,j->b', iLyy_y[:, 0], W, iLyy_y[:, 0])
  if full_output_cov:
    Sff = (f2[:, None] - f1 ** 2)[:, :, None]
  else:
    Sff = f2[:, None] - f1 ** 2
  if model_uncertainty:                                       # :86-88
    e_cov = eKff - np.trace(W, axis1=1, axis2=2)
    Sff = Sff + (e_cov[:, None, None] if full_output_cov else e_cov[:, None])

  iSxx_Sxf = _cross_term(mu, Sigma, X[None], ls[None], eKfu[:, :, None], iKyy_y.T)
  if model.mean_c is not None:
    f1 = f1 + model.mean_c
  Sff = Sff + jitter
  return f1, Sff, iSxx_Sxf


# ---------------------------------------------------------------------------
# GaussianMatch algebra and the Euler moment update
# ---------------------------------------------------------------------------
def cross_covariance(Sigma, cross, is_preinv: bool, preinv: bool = False):
  """``GaussianMatch.cross_covariance``: moment_matching/gaussian.py:33-51."""
  if not preinv and is_preinv:
    return Sigma @ cross
  if preinv and not is_preinv:
    return np.linalg.solve(Sigma, cross)
  return cross


def joint(mu, Sigma, f1, Sff, Sxy):
  """``GaussianMatch.joint``: moment_matching/gaussian.py:53-63 (Sxy = Cov(x,y))."""
  m = np.concatenate([mu, f1], axis=-1)
  top = np.concatenate([Sigma, Sxy], axis=-1)
  bot = np.concatenate([np.swapaxes(Sxy, -1, -2), Sff], axis=-1)
  return m, np.concatenate([top, bot], axis=-2)


def euler_moment_update(mu, Sigma, f1, Sff, Sxf, dt=1.0):
  """``MomentMatchingEuler.step``: dynamics/solvers.py:110-135 (diffusion None)."""
  mu_new = mu + dt * f1
  Sigma_new = Sigma + dt * (Sxf + np.swapaxes(Sxf, -1, -2)) + (dt ** 2) * Sff
  return mu_new, Sigma_new


def rollout_closed(mu0, Sigma0, model: SVGPParams, num_steps: int, dt=1.0,
                   model_uncertainty=True, keep=False):
  """Drift-only moment-matched rollout (state dim == d == L).

  ``Euler.__call__`` fold (dynamics/solvers.py:67-105) with
  ``forward_sde(x, drift, None, None, None)`` (dynamics/forward_sde.py:34-46).
  """
  mu, Sigma = mu0.copy(), Sigma0.copy()
  traj = []
  for _ in range(num_steps):
    f1, Sff, cross = mm_gauss_svgp_mo(mu, Sigma, model, True, model_uncertainty, 0.0)
    Sxf = cross_covariance(Sigma, cross, is_preinv=True)
    mu, Sigma = euler_moment_update(mu, Sigma, f1, Sff, Sxf, dt)
    if keep:
      traj.append((mu.copy(), Sigma.copy()))
  return (mu, Sigma, traj) if keep else (mu, Sigma)
