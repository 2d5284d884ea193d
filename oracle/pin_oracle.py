"""Pin ``oracle/mm_oracle.py`` the way the reference pins itself (TEST INFRASTRUCTURE).

Re-runs the reference's Monte-Carlo test designs in numpy (TensorFlow/GPflow are
not installed, so the reference tests themselves cannot run here):

* ``tests/test_kernel_expectation.py:51-93``  -- <K_aX>, <K_bX>, <K_aX K_Xb> for two
  different SE-ARD kernels, d=2, 32 inducing points, input std 0.1.
* ``tests/test_moment_matching.py:88-136``    -- GPR,
  ``:140-194`` single-output SVGP (whiten=False),
  ``:199-264`` 2-latent / 3-output LinearCoregionalization SVGP (whiten=False);
  d=4, 16 conditioning points, 2 input distributions, input std 0.01, Constant mean.
* the 1e-12 diag-vs-full consistency checks (``:127-136,185-194,255-264``).

Acceptance is the reference's: |a-b| <= 10/sqrt(num_samples) absolute
(``tests/utils.py:43-44,66-67``).  Run ``python -m oracle.pin_oracle`` for the
full 10^6-sample version; ``tests/test_oracle_pin.py`` runs a smaller one.
"""
from __future__ import annotations

import sys
from math import log

import numpy as np
from scipy.linalg import cho_solve, cholesky, solve_triangular

from oracle import mm_oracle as mo


# ---- helpers restating tests/utils.py -------------------------------------
def generate_covariance(rng, ndims, sample_shape=(), scale=None):
  """tests/utils.py:99-121 (random eigen prior, rescaled to marginal std)."""
  shape = tuple(sample_shape)
  eigen_vals = -np.log(rng.uniform(size=shape + (1, ndims)))
  A = rng.standard_normal(shape + (ndims, ndims))
  orthog = np.linalg.svd(A, full_matrices=True)[0]
  sqrt_cov = np.sqrt(eigen_vals) * orthog
  cov = sqrt_cov @ np.swapaxes(sqrt_cov, -1, -2)
  if scale is not None:
    istd = 1.0 / np.sqrt(np.diagonal(cov, axis1=-2, axis2=-1))
    cov = (scale ** 2) * cov * istd[..., None] * istd[..., None, :]
  return cov


def draw_samples_mvn(rng, mu, cov, num_samples):
  """tests/utils.py:70-81 -> [S, B, d]."""
  sqrt = np.linalg.cholesky(cov)
  rvs = rng.standard_normal((num_samples,) + mu.shape)
  return mu + np.einsum('...ij,s...j->s...i', sqrt, rvs)


def mc_tol(num_samples):
  return 10.0 * num_samples ** -0.5          # tests/utils.py:43-44


# ---- third-party predictive equations (gpflow conditionals, published) ----
def svgp_predict_f(X, model: mo.SVGPParams):
  """gpflow SVGP.predict_f(full_cov=False, full_output_cov=True) -> mean [n,P], cov [n,P,P]."""
  L, M, d = model.Z.shape
  n = X.shape[0]
  g_mu = np.empty((n, L))
  g_var = np.empty((n, L))
  for a in range(L):
    Kuu = mo.se_kernel(model.Z[a], None, model.lengthscales[a], model.variance[a]) \
        + model.kuu_jitter * np.eye(M)
    Lu = cholesky(Kuu, lower=True)
    Kuf = mo.se_kernel(model.Z[a], X, model.lengthscales[a], model.variance[a])  # [M,n]
    A = solve_triangular(Lu, Kuf, lower=True)                 # L^-1 Kuf
    q_mu = model.q_mu[:, a]
    q_sqrt = np.tril(model.q_sqrt[a])
    if not model.whiten:
      A2 = solve_triangular(Lu.T, A, lower=False)             # Kuu^-1 Kuf
      g_mu[:, a] = A2.T @ q_mu
      g_var[:, a] = model.variance[a] - np.sum(A * A, 0) + np.sum((q_sqrt.T @ A2) ** 2, 0)
    else:
      g_mu[:, a] = A.T @ q_mu
      g_var[:, a] = model.variance[a] - np.sum(A * A, 0) + np.sum((q_sqrt.T @ A) ** 2, 0)
  if model.W is not None:
    mean = g_mu @ model.W.T
    cov = np.einsum('pl,nl,ql->npq', model.W, g_var, model.W)
  else:
    mean = g_mu
    cov = np.einsum('nl,lk->nlk', g_var, np.eye(L))
  if model.mean_c is not None:
    mean = mean + np.asarray(model.mean_c)[None]
  return mean, cov


def gpr_predict_f(X, model: mo.GPRParams):
  N = model.X.shape[0]
  Kyy = mo.se_kernel(model.X, None, model.lengthscales, model.variance) \
      + model.noise_variance * np.eye(N)
  Ly = cholesky(Kyy, lower=True)
  c = 0.0 if model.mean_c is None else model.mean_c
  Kxf = mo.se_kernel(model.X, X, model.lengthscales, model.variance)
  A = solve_triangular(Ly, Kxf, lower=True)
  mean = A.T @ solve_triangular(Ly, model.Y - c, lower=True) + c
  var = model.variance - np.sum(A * A, 0)
  return mean, var[:, None, None]


def monte_carlo_estimator(rng, predict, mx, Sxx, num_samples, chunk=100000):
  """tests/test_moment_matching.py:57-84 (streamed in chunks to bound memory)."""
  B, d = mx.shape
  mean0, cov0 = predict(mx)
  P = mean0.shape[-1]
  s_f = np.zeros((B, P)); s_ff = np.zeros((B, P, P)); s_var = np.zeros((B, P, P))
  s_xf = np.zeros((B, d, P))
  done = 0
  while done < num_samples:
    n = min(chunk, num_samples - done)
    X = draw_samples_mvn(rng, mx, Sxx, n)                     # [n,B,d]
    F_mu, F_cov = predict(X.reshape(-1, d))
    F_mu = F_mu.reshape(n, B, P); F_cov = F_cov.reshape(n, B, P, P)
    s_f += F_mu.sum(0)
    s_ff += np.einsum('sni,snj->nij', F_mu, F_mu)
    s_var += F_cov.sum(0)
    s_xf += np.einsum('sni,snj->nij', X, F_mu)
    done += n
  mf = s_f / num_samples
  Sff = s_ff / num_samples - mf[:, :, None] * mf[:, None, :] + s_var / num_samples
  Sxf = s_xf / num_samples - mx[:, :, None] * mf[:, None, :]
  return mf, Sff, Sxf


# ---- the reference's test designs ------------------------------------------
def check_kernel_expectation(seed, num_samples=int(1e6), ndims_x=2, num_inducing=32,
                             scale_x=0.10, scale_f=0.89, ls_bounds=(0.1, 10.0)):
  """tests/test_kernel_expectation.py:51-93."""
  rng = np.random.default_rng(seed)
  mx = rng.standard_normal(ndims_x)
  Sxx = generate_covariance(rng, ndims_x, scale=scale_x)

  def kernel_and_inducing():
    ls = np.exp(rng.uniform(log(ls_bounds[0]), log(ls_bounds[1]), size=ndims_x))
    Z1 = draw_samples_mvn(rng, mx, 0.1 * Sxx, num_inducing // 2)
    Z2 = rng.uniform(size=(num_inducing - len(Z1), ndims_x))
    return ls, np.concatenate([Z1, Z2], 0)

  lsA, A = kernel_and_inducing()
  lsB, Bz = kernel_and_inducing()
  var = scale_f ** 2
  eA = mo.eKfu_se(mx[None], Sxx[None], A, lsA, var)[0]
  eB = mo.eKfu_se(mx[None], Sxx[None], Bz, lsB, var)[0]
  eAB = mo.eKuffu_se_pair(mx[None], Sxx[None], lsA, var, A, lsB, var, Bz, False, False)[0]
  eAA = mo.eKuffu_se_pair(mx[None], Sxx[None], lsA, var, A, lsA, var, A, True, True)[0]
  eAA2 = mo.eKuffu_se_pair(mx[None], Sxx[None], lsA, var, A, lsA, var, A, False, False)[0]

  X = draw_samples_mvn(rng, mx, Sxx, num_samples)
  KA = mo.se_kernel(A, X, lsA, var); KB = mo.se_kernel(Bz, X, lsB, var)
  errs = {
      'eKfu_A': np.abs(eA - KA.mean(1)).max(),
      'eKfu_B': np.abs(eB - KB.mean(1)).max(),
      'eKuffu_AB': np.abs(eAB - KA @ KB.T / num_samples).max(),
      'eKuffu_AA': np.abs(eAA - KA @ KA.T / num_samples).max(),
      # same-kernel shortcut (:167-174) vs general branch (:175-185): exact identity
      'branch_identity': np.abs(eAA - eAA2).max() / 1e3,
  }
  return errs, mc_tol(num_samples)


def _mm_config(rng, ndims_x=4, ls_bounds=(0.01, 10.0)):
  return np.exp(rng.uniform(log(ls_bounds[0]), log(ls_bounds[1]), size=ndims_x))


def check_gpr(seed, num_samples=int(1e6), ndims_x=4, num_cond=16, num_eval=2,
              scale_x=0.01, scale_f=0.89):
  """tests/test_moment_matching.py:88-136."""
  rng = np.random.default_rng(seed)
  ls = _mm_config(rng, ndims_x)
  model = mo.GPRParams(X=rng.uniform(size=(num_cond, ndims_x)),
                       Y=scale_f * rng.standard_normal((num_cond, 1)),
                       lengthscales=ls, variance=scale_f ** 2, noise_variance=1e-5,
                       mean_c=float(1 + rng.standard_normal()))
  mx = rng.uniform(size=(num_eval, ndims_x))
  Sxx = generate_covariance(rng, ndims_x, (num_eval,), scale_x)
  _mf, _Sff, _Sxf = monte_carlo_estimator(rng, lambda X: gpr_predict_f(X, model), mx, Sxx, num_samples)
  f1, Sff, pre = mo.mm_gauss_gpr(mx, Sxx, model)
  f1d, Sffd, pred = mo.mm_gauss_gpr(mx, Sxx, model, full_output_cov=False)
  errs = {'mean': np.abs(f1 - _mf).max(), 'cov': np.abs(Sff - _Sff).max(),
          'cross': np.abs(Sxx @ pre - _Sxf).max()}
  exact = {'diag_mean': np.abs(f1d - f1).max(),
           'diag_cov': np.abs(Sffd - np.diagonal(Sff, axis1=1, axis2=2)).max(),
           'diag_cross': np.abs(pred - pre).max()}
  return errs, exact, mc_tol(num_samples)


def make_svgp_test_model(rng, ndims_x=4, num_cond=16, ndims_f=1, ndims_y=None,
                         scale_f=0.89, whiten=False):
  Z = rng.uniform(size=(ndims_f, num_cond, ndims_x))
  ls = np.stack([_mm_config(rng, ndims_x) for _ in range(ndims_f)])
  q_mu = scale_f * rng.standard_normal((num_cond, ndims_f))
  q_cov = generate_covariance(rng, num_cond, (ndims_f,), scale_f)
  W = None
  P = ndims_f
  if ndims_y is not None:
    W = rng.uniform(size=(ndims_y, ndims_f))
    W = W / np.linalg.norm(W, axis=-1, keepdims=True)
    P = ndims_y
  return mo.SVGPParams(Z=Z, lengthscales=ls, variance=np.full(ndims_f, scale_f ** 2),
                       q_mu=q_mu, q_sqrt=np.linalg.cholesky(q_cov), whiten=whiten,
                       mean_c=1 + rng.standard_normal(P), W=W)


def check_svgp(seed, num_samples=int(1e6), ndims_x=4, num_eval=2, scale_x=0.01,
               multi_output=False, whiten=False):
  """tests/test_moment_matching.py:140-194 (so) and :199-264 (mo, LCM 2->3)."""
  rng = np.random.default_rng(seed)
  if multi_output:
    model = make_svgp_test_model(rng, ndims_x, ndims_f=2, ndims_y=3, whiten=whiten)
    handler = mo.mm_gauss_svgp_mo
  else:
    model = make_svgp_test_model(rng, ndims_x, ndims_f=1, whiten=whiten)
    handler = mo.mm_gauss_svgp_so
  mx = rng.uniform(size=(num_eval, ndims_x))
  Sxx = generate_covariance(rng, ndims_x, (num_eval,), scale_x)
  _mf, _Sff, _Sxf = monte_carlo_estimator(rng, lambda X: svgp_predict_f(X, model), mx, Sxx, num_samples)
  f1, Sff, pre = handler(mx, Sxx, model)
  f1d, Sffd, pred = handler(mx, Sxx, model, full_output_cov=False)
  errs = {'mean': np.abs(f1 - _mf).max(), 'cov': np.abs(Sff - _Sff).max(),
          'cross': np.abs(Sxx @ pre - _Sxf).max()}
  exact = {'diag_mean': np.abs(f1d - f1).max(),
           'diag_cov': np.abs(Sffd - np.diagonal(Sff, axis1=1, axis2=2)).max(),
           'diag_cross': np.abs(pred - pre).max()}
  return errs, exact, mc_tol(num_samples)


def main(argv=None):
  n = int(float(argv[1])) if argv and len(argv) > 1 else int(1e6)
  ok = True
  for seed in (11, 12, 13):
    errs, tol = check_kernel_expectation(seed, n)
    print(f'[kernel_expectation seed={seed}] tol={tol:.2e} ' +
          ' '.join(f'{k}={v:.2e}' for k, v in errs.items()))
    ok &= all(v <= tol for v in errs.values())
    for name, fn, kw in (('gpr', check_gpr, {}),
                         ('svgp_so', check_svgp, {}),
                         ('svgp_mo_lcm', check_svgp, {'multi_output': True}),
                         ('svgp_so_whiten', check_svgp, {'whiten': True}),
                         ('svgp_mo_lcm_whiten', check_svgp, {'multi_output': True, 'whiten': True})):
      errs, exact, tol = fn(seed, n, **kw)
      print(f'[{name} seed={seed}] tol={tol:.2e} ' +
            ' '.join(f'{k}={v:.2e}' for k, v in {**errs, **exact}.items()))
      ok &= all(v <= tol for v in errs.values()) and all(v <= 1e-12 for v in exact.values())
  print('PINNED' if ok else 'FAILED')
  return 0 if ok else 1


if __name__ == '__main__':
  sys.exit(main(sys.argv))
