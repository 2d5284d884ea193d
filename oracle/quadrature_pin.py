"""Pin ``oracle/mm_oracle.py`` to ~1e-10 with tensor Gauss-Hermite quadrature (TEST INFRASTRUCTURE).

The reference's own tests accept 1e-2 absolute (10^6-sample Monte Carlo,
``/root/reference/tests/utils.py:43-44,66-67``) and never touch ``whiten=True``,
``SeparateIndependent`` kernels, ``model_uncertainty=False`` or the Euler moment update
(SURVEY.md section 8c).  The reference cannot be imported here (TensorFlow / GPflow absent), so this
module pins those cases by the DEFINITION the reference's Monte-Carlo estimator samples
(``tests/test_moment_matching.py:57-84``): for x ~ N(mu, Sigma) and f | x ~ predict_f(x),

    E f        = E[F_mu(x)]
    Cov f      = Cov[F_mu(x)] + E[F_cov(x)]        (without E[F_cov] when model_uncertainty=False)
    Cov(x, f)  = E[(x - mu) F_mu(x)^T]

evaluated with an n^d-node Gauss-Hermite rule in d = 2, 3 -- and in d = 4 on the reference's OWN three test designs at their own
sizes (``reference_design`` / ``check_reference_design``: 24^4 nodes); the kernel expectations themselves on the reference's own
kernel-expectation design (``check_kernel_expectation_design``: d = 2, 160^2 nodes, 1e-14).  The integrands are sums of products of
squared-exponential kernels in x (entire functions), so the rule converges geometrically; the
``predict_f`` used is ``pin_oracle.svgp_predict_f`` / ``gpr_predict_f`` (gpflow's published
conditional), which shares no kernel-expectation code with the moment-matching oracle.

The Euler update is pinned the same way: the first two moments of x' = x + dt f(x)
(``dynamics/solvers.py:110-135``) by quadrature against ``mm_oracle.euler_moment_update``.
"""
from __future__ import annotations

import itertools

import numpy as np

from oracle import mm_oracle as mo
from oracle import pin_oracle as po


def gauss_hermite_nodes(mu, Sigma, n):
  """Nodes x_k [K, d] and weights w_k [K] (sum 1) of the n^d tensor rule for N(mu, Sigma)."""
  d = mu.shape[0]
  t, w = np.polynomial.hermite.hermgauss(n)                   # int e^{-t^2} g(t) dt
  Lc = np.linalg.cholesky(Sigma)
  grids = np.array(list(itertools.product(range(n), repeat=d)))          # [K, d]
  T = t[grids]                                                            # [K, d]
  W = np.prod(w[grids], axis=1) / np.pi ** (d / 2.0)
  X = mu[None, :] + np.sqrt(2.0) * T @ Lc.T
  return X, W


def quadrature_moments(predict, mu, Sigma, n, model_uncertainty=True):
  """(E f [P], Cov f [P,P], Cov(x,f) [d,P]) of one input distribution by the n^d rule."""
  X, W = gauss_hermite_nodes(mu, Sigma, n)
  F_mu, F_cov = predict(X)                                                # [K,P], [K,P,P]
  mf = W @ F_mu
  Fc = F_mu - mf[None]
  Sff = np.einsum('k,ki,kj->ij', W, Fc, Fc)
  if model_uncertainty:
    Sff = Sff + np.einsum('k,kij->ij', W, F_cov)
  Sxf = np.einsum('k,ki,kj->ij', W, X - mu[None], F_mu)
  return mf, Sff, Sxf


def quadrature_euler(predict, mu, Sigma, n, dt, model_uncertainty=True):
  """First two moments of x' = x + dt f(x), f | x ~ N(F_mu(x), F_cov(x)), by the n^d rule."""
  X, W = gauss_hermite_nodes(mu, Sigma, n)
  F_mu, F_cov = predict(X)
  Y = X + dt * F_mu
  m = W @ Y
  Yc = Y - m[None]
  S = np.einsum('k,ki,kj->ij', W, Yc, Yc)
  if model_uncertainty:
    S = S + dt * dt * np.einsum('k,kij->ij', W, F_cov)
  return m, S


def make_case(seed, d, L, M=12, whiten=True, lcm_outputs=None, mean=True, ls_bounds=(0.5, 2.0), scale_x=0.25,
              B=2):
  """A small SVGP with L latents of DIFFERENT lengthscales (SeparateIndependent unless lcm_outputs)."""
  rng = np.random.default_rng(seed)
  Z = rng.uniform(size=(L, M, d))
  ls = np.exp(rng.uniform(np.log(ls_bounds[0]), np.log(ls_bounds[1]), size=(L, d)))
  q_mu = 0.89 * rng.standard_normal((M, L))
  q_cov = po.generate_covariance(rng, M, (L,), 0.5)
  W = None
  P = L
  if lcm_outputs is not None:
    W = rng.uniform(size=(lcm_outputs, L))
    W = W / np.linalg.norm(W, axis=-1, keepdims=True)
    P = lcm_outputs
  model = mo.SVGPParams(Z=Z, lengthscales=ls, variance=0.89 ** 2 * (1.0 + 0.3 * rng.uniform(size=L)),
                        q_mu=q_mu, q_sqrt=np.linalg.cholesky(q_cov), whiten=whiten,
                        mean_c=(1 + rng.standard_normal(P)) if mean else None, W=W,
                        # whiten=False multiplies (q_mu, q_sqrt) by Kuu^-1: at the default jitter 1e-6 BOTH sides of the
                        # comparison lose cond(Kuu) * eps ~ 1e-4 of the (O(50)) covariance; the jitter is a model field
                        # of both, so a well-conditioned Kuu pins the same formulas to the quadrature's accuracy
                        kuu_jitter=mo.DEFAULT_JITTER if whiten else 1e-2)
  mu = rng.uniform(0.2, 0.8, size=(B, d))
  Sigma = po.generate_covariance(rng, d, (B,), scale_x)
  return model, mu, Sigma


def check_svgp(seed, d, L, n, whiten=True, model_uncertainty=True, lcm_outputs=None, single_output=False):
  """max abs error of (f1, Sff incl. off-diagonal pairs, Cov(x,f)) against the quadrature."""
  model, mu, Sigma = make_case(seed, d, 1 if single_output else L, whiten=whiten, lcm_outputs=lcm_outputs)
  handler = mo.mm_gauss_svgp_so if single_output else mo.mm_gauss_svgp_mo
  f1, Sff, pre = handler(mu, Sigma, model, True, model_uncertainty, 0.0)
  Sxf = mo.cross_covariance(Sigma, pre, is_preinv=True)
  errs = {'mean': 0.0, 'cov': 0.0, 'cov_offdiag': 0.0, 'cross': 0.0}
  for b in range(mu.shape[0]):
    qf, qS, qX = quadrature_moments(lambda X: po.svgp_predict_f(X, model), mu[b], Sigma[b], n, model_uncertainty)
    errs['mean'] = max(errs['mean'], np.abs(f1[b] - qf).max())
    errs['cov'] = max(errs['cov'], np.abs(Sff[b] - qS).max())
    off = ~np.eye(qS.shape[0], dtype=bool)
    if off.any():
      errs['cov_offdiag'] = max(errs['cov_offdiag'], np.abs((Sff[b] - qS)[off]).max())
    errs['cross'] = max(errs['cross'], np.abs(Sxf[b] - qX).max())
  scale = {'mean': np.abs(f1).max(), 'cov': np.abs(Sff).max(), 'cross': np.abs(Sxf).max()}
  return errs, scale


def check_gpr(seed, d, n, N=14, model_uncertainty=True):
  rng = np.random.default_rng(seed)
  model = mo.GPRParams(X=rng.uniform(size=(N, d)), Y=0.89 * rng.standard_normal((N, 1)),
                       lengthscales=np.exp(rng.uniform(np.log(0.5), np.log(2.0), size=d)),
                       variance=0.89 ** 2, noise_variance=1e-3, mean_c=float(1 + rng.standard_normal()))
  mu = rng.uniform(0.2, 0.8, size=(2, d))
  Sigma = po.generate_covariance(rng, d, (2,), 0.25)
  f1, Sff, pre = mo.mm_gauss_gpr(mu, Sigma, model, True, model_uncertainty, 0.0)
  Sxf = mo.cross_covariance(Sigma, pre, is_preinv=True)
  errs = {'mean': 0.0, 'cov': 0.0, 'cross': 0.0}

  def predict(X):
    m, c = po.gpr_predict_f(X, model)
    return m, c

  for b in range(2):
    qf, qS, qX = quadrature_moments(predict, mu[b], Sigma[b], n, model_uncertainty)
    errs['mean'] = max(errs['mean'], np.abs(f1[b] - qf).max())
    errs['cov'] = max(errs['cov'], np.abs(Sff[b] - qS).max())
    errs['cross'] = max(errs['cross'], np.abs(Sxf[b] - qX).max())
  return errs


def check_euler(seed, d, n, dt=0.7, model_uncertainty=True):
  """One MomentMatchingEuler step (state dim == d == L) against the moments of x + dt f(x)."""
  model, mu, Sigma = make_case(seed, d, d, whiten=True, mean=True)
  f1, Sff, pre = mo.mm_gauss_svgp_mo(mu, Sigma, model, True, model_uncertainty, 0.0)
  Sxf = mo.cross_covariance(Sigma, pre, is_preinv=True)
  m1, S1 = mo.euler_moment_update(mu, Sigma, f1, Sff, Sxf, dt)
  errs = {'mean': 0.0, 'cov': 0.0}
  for b in range(mu.shape[0]):
    qm, qS = quadrature_euler(lambda X: po.svgp_predict_f(X, model), mu[b], Sigma[b], n, dt, model_uncertainty)
    errs['mean'] = max(errs['mean'], np.abs(m1[b] - qm).max())
    errs['cov'] = max(errs['cov'], np.abs(S1[b] - qS).max())
  return errs


def quadrature_moments_chunked(predict, mu, Sigma, n, model_uncertainty=True, chunk=1 << 17):
  """``quadrature_moments`` for large rules (d = 4: n^4 nodes): the nodes are visited in chunks, raw moments accumulated."""
  X, W = gauss_hermite_nodes(mu, Sigma, n)
  s0 = None
  for i in range(0, X.shape[0], chunk):
    Xc, Wc = X[i:i + chunk], W[i:i + chunk]
    F_mu, F_cov = predict(Xc)
    parts = (Wc @ F_mu, np.einsum('k,ki,kj->ij', Wc, F_mu, F_mu), np.einsum('k,kij->ij', Wc, F_cov),
             np.einsum('k,ki,kj->ij', Wc, Xc - mu[None], F_mu))
    s0 = parts if s0 is None else tuple(a + b for a, b in zip(s0, parts))
  mf, m2, ec, Sxf = s0
  Sff = m2 - np.outer(mf, mf)
  if model_uncertainty:
    Sff = Sff + ec
  return mf, Sff, Sxf


def reference_design(kind, seed):
  """One draw of the reference's OWN test designs at its own sizes (tests/test_moment_matching.py:25-54: d = 4, 16 conditioning
  points, 2 input distributions of std 0.01, lengthscales log-uniform in [0.01, 10], signal std 0.89, Constant mean; Kuu jitter
  1e-6 and noise 1e-5 as there): "gpr" (:88-136), "svgp_so" (:140-194, whiten=False), "svgp_mo_lcm" (:199-264,
  LinearCoregionalization 2 latents -> 3 outputs, whiten=False).  -> (model, mx [2,4], Sxx [2,4,4], handler, predict_f)."""
  rng = np.random.default_rng(seed)
  if kind == "gpr":
    model = mo.GPRParams(X=rng.uniform(size=(16, 4)), Y=0.89 * rng.standard_normal((16, 1)), lengthscales=po._mm_config(rng, 4),
                         variance=0.89 ** 2, noise_variance=1e-5, mean_c=float(1 + rng.standard_normal()))
    handler, predict = mo.mm_gauss_gpr, (lambda X: po.gpr_predict_f(X, model))
  elif kind == "svgp_so":
    model = po.make_svgp_test_model(rng, 4, ndims_f=1, whiten=False)
    handler, predict = mo.mm_gauss_svgp_so, (lambda X: po.svgp_predict_f(X, model))
  elif kind == "svgp_mo_lcm":
    model = po.make_svgp_test_model(rng, 4, ndims_f=2, ndims_y=3, whiten=False)
    handler, predict = mo.mm_gauss_svgp_mo, (lambda X: po.svgp_predict_f(X, model))
  else:
    raise ValueError(kind)
  mx = rng.uniform(size=(2, 4))
  Sxx = po.generate_covariance(rng, 4, (2,), 0.01)
  return model, mx, Sxx, handler, predict


REFERENCE_DESIGNS = (("gpr", 501), ("gpr", 502), ("svgp_so", 501), ("svgp_so", 502), ("svgp_mo_lcm", 501), ("svgp_mo_lcm", 502))


def check_reference_design(kind, seed, n=24):
  """``reference_design`` against the n^4-node Gauss-Hermite rule of the DEFINITION the reference's Monte-Carlo estimator samples
  (tests/test_moment_matching.py:57-84).  The reference accepts 1e-2 there; this is the digit-level pin of the oracle on the
  reference's own designs.  -> (max abs errors of mean / full covariance / cross-covariance, their scales).  n = 16 and 24 agree
  to the last digit shown (what is left, <= 1e-10, is cond(K) * eps of the conditionals on BOTH sides)."""
  model, mx, Sxx, handler, predict = reference_design(kind, seed)
  f1, Sff, pre = handler(mx, Sxx, model)
  Sxf = mo.cross_covariance(Sxx, pre, is_preinv=True)
  errs = {'mean': 0.0, 'cov': 0.0, 'cross': 0.0}
  for b in range(2):
    qf, qS, qX = quadrature_moments_chunked(predict, mx[b], Sxx[b], n)
    errs['mean'] = max(errs['mean'], np.abs(f1[b] - qf).max())
    errs['cov'] = max(errs['cov'], np.abs(Sff[b] - qS).max())
    errs['cross'] = max(errs['cross'], np.abs(Sxf[b] - qX).max())
  scale = {'mean': np.abs(f1).max(), 'cov': np.abs(Sff).max(), 'cross': np.abs(Sxf).max()}
  return errs, scale


def kernel_expectation_design(seed):
  """One draw of the reference's kernel-expectation test design (tests/test_kernel_expectation.py:51-93) at its own sizes
  -> (mx [2], Sxx [2,2], lsA [2], A [32,2], lsB [2], B [32,2], variance)."""
  rng = np.random.default_rng(seed)
  mx = rng.standard_normal(2)
  Sxx = po.generate_covariance(rng, 2, scale=0.10)

  def kernel_and_inducing():
    ls = np.exp(rng.uniform(np.log(0.1), np.log(10.0), size=2))
    Z1 = po.draw_samples_mvn(rng, mx, 0.1 * Sxx, 16)
    Z2 = rng.uniform(size=(16, 2))
    return ls, np.concatenate([Z1, Z2], 0)
  lsA, A = kernel_and_inducing()
  lsB, Bz = kernel_and_inducing()
  return mx, Sxx, lsA, A, lsB, Bz, 0.89 ** 2


def check_kernel_expectation_design(seed, n=160):
  """The reference's kernel-expectation test (tests/test_kernel_expectation.py:51-93) at its own sizes -- d = 2, two SE kernels
  with lengthscales log-uniform in [0.1, 10], 32 inducing points each (half drawn near the mode of p(x), half uniform), input
  std 0.1, signal std 0.89 -- with its 1e6-sample Monte-Carlo estimate (accepted at 1e-2) replaced by the n^2-node Gauss-Hermite
  rule of the SAME definitions:  <k(x, A)>,  <k(x, B)>,  <k2(A, x) k3(x, B)>  and the same-kernel  <k2(A, x) k2(x, A)>  (both
  branches of utils/kernel_expectation.py:167-185).  -> max abs errors of the oracle's closed forms."""
  mx, Sxx, lsA, A, lsB, Bz, var = kernel_expectation_design(seed)
  eA = mo.eKfu_se(mx[None], Sxx[None], A, lsA, var)[0]
  eB = mo.eKfu_se(mx[None], Sxx[None], Bz, lsB, var)[0]
  eAB = mo.eKuffu_se_pair(mx[None], Sxx[None], lsA, var, A, lsB, var, Bz, False, False)[0]
  eAA = mo.eKuffu_se_pair(mx[None], Sxx[None], lsA, var, A, lsA, var, A, True, True)[0]
  eAA2 = mo.eKuffu_se_pair(mx[None], Sxx[None], lsA, var, A, lsA, var, A, False, False)[0]
  X, W = gauss_hermite_nodes(mx, Sxx, n)
  KA = mo.se_kernel(A, X, lsA, var); KB = mo.se_kernel(Bz, X, lsB, var)           # [32, K]
  errs = {'eKfu_A': np.abs(eA - KA @ W).max(), 'eKfu_B': np.abs(eB - KB @ W).max(),
          'eKuffu_AB': np.abs(eAB - (KA * W[None]) @ KB.T).max(),
          'eKuffu_AA_same_kernel_branch': np.abs(eAA - (KA * W[None]) @ KA.T).max(),
          'eKuffu_AA_general_branch': np.abs(eAA2 - (KA * W[None]) @ KA.T).max()}
  scale = {'eKfu': max(np.abs(eA).max(), np.abs(eB).max()), 'eKuffu': max(np.abs(eAB).max(), np.abs(eAA).max())}
  return errs, scale


KERNEL_EXPECTATION_SEEDS = (601, 602, 603)


def main():
  ok = True
  for seed in KERNEL_EXPECTATION_SEEDS:
    errs, scale = check_kernel_expectation_design(seed)
    print(f'[reference kernel-expectation design seed={seed} d=2 n=160] ' + ' '.join(f'{k}={v:.2e}' for k, v in errs.items()) +
          '  | scale ' + ' '.join(f'{k}={v:.2e}' for k, v in scale.items()))
    ok &= all(v <= 1e-12 for v in errs.values())
  for d, n in ((2, 60), (3, 40)):
    for kw in (dict(whiten=True), dict(whiten=True, model_uncertainty=False), dict(whiten=False),
               dict(whiten=False, lcm_outputs=4), dict(whiten=True, single_output=True)):
      errs, scale = check_svgp(7 + d, d, 3, n, **kw)
      print(f'[svgp d={d} n={n} {kw}] ' + ' '.join(f'{k}={v:.2e}' for k, v in errs.items()) +
            '  | scale ' + ' '.join(f'{k}={v:.2e}' for k, v in scale.items()))
      ok &= all(v <= 1e-9 for v in errs.values())
    e = check_gpr(17 + d, d, n)
    print(f'[gpr d={d} n={n}] ' + ' '.join(f'{k}={v:.2e}' for k, v in e.items()))
    ok &= all(v <= 1e-9 for v in e.values())
    for mu_flag in (True, False):
      e = check_euler(27 + d, d, n, model_uncertainty=mu_flag)
      print(f'[euler d={d} n={n} model_uncertainty={mu_flag}] ' + ' '.join(f'{k}={v:.2e}' for k, v in e.items()))
      ok &= all(v <= 1e-9 for v in e.values())
  for kind, seed in REFERENCE_DESIGNS:
    errs, scale = check_reference_design(kind, seed, 24)
    print(f'[reference design {kind} seed={seed} d=4 n=24] ' + ' '.join(f'{k}={v:.2e}' for k, v in errs.items()) +
          '  | scale ' + ' '.join(f'{k}={v:.2e}' for k, v in scale.items()))
    ok &= all(v <= 1e-8 for v in errs.values())
  print('PINNED (<= 1e-9; the reference\'s own d = 4 designs <= 1e-8)' if ok else 'FAILED')
  return 0 if ok else 1


if __name__ == '__main__':
  raise SystemExit(main())
