"""Numpy prototype of the degree-4 moment collapse of the f32 off-diagonal reduce (design check, fp64).

  S = sum_ij what_i what'_j P(b_ij),   b_ij = (zc_i - dmu)^T G zc'_j,   P(x) = sum_{n<=4} a_n x^n

evaluated (a) densely, (b) from raw moments of the two weight vectors against the MODEL-CONSTANT monomial tables
(all monomials of total degree <= 4 of the centred inducing inputs, colex order inside a degree), with the
per-(b, pair) d^n contraction the HIP kernel k_spoly performs:
  T_n = sym. tensor of the column side's degree-n moments;  T'_n = G^{(x)n} T_n;
  M_n[k1..kn] = sum_{S subset [n]} (-1)^{n-|S|} N_|S|[k_S] prod_{t not in S} dmu[k_t]   (row side, shifted);
  S_n = <M_n, T'_n>.
Prints the agreement and the Cauchy-Schwarz bound statistics that decide the collapse per (b, pair).
"""
import itertools
import sys
from math import comb

import numpy as np

sys.path.insert(0, '/root/repo')
from gpflowpilco_amd.synthetic import make_inputs, make_svgp   # noqa: E402
from oracle import mm_oracle as mo   # noqa: E402

C0, C1 = 1.666936278e-01, 4.167173430e-02        # tier-1 minimax coefficients of r(x)/x^3 (mm_mfma.hip MMRem<1>)


def colex_rank(t):
  """rank of a sorted tuple k1 <= ... <= kn (multiset of size n) inside its degree block."""
  return sum(comb(k + i, i + 1) for i, k in enumerate(t))


def degree_offset(n, d):
  return sum(comb(d + m - 1, m) for m in range(n))


def monomial_table(zc, maxdeg=4):
  M, d = zc.shape
  ncol = degree_offset(maxdeg + 1, d)
  tab = np.zeros((M, ncol))
  for n in range(maxdeg + 1):
    for t in itertools.combinations_with_replacement(range(d), n):
      tab[:, degree_offset(n, d) + colex_rank(t)] = np.prod(zc[:, list(t)], axis=1) if n else 1.0
  return tab


def expand_sym(packed, n, d):
  """packed degree-n block -> full d^n symmetric tensor."""
  T = np.empty((d,) * n)
  for idx in itertools.product(range(d), repeat=n):
    T[idx] = packed[degree_offset(n, d) + colex_rank(tuple(sorted(idx)))]
  return T


def poly_sum_from_moments(nhat, qhat, G, dmu, a):
  d = G.shape[0]
  total = 0.0
  for n in range(len(a)):
    if a[n] == 0.0:
      continue
    if n == 0:
      total += a[0] * nhat[0] * qhat[0]
      continue
    Tn = expand_sym(qhat, n, d)
    for ax in range(n):                                   # T' = G applied to every index
      Tn = np.moveaxis(np.tensordot(G, Tn, axes=([1], [ax])), 0, ax)
    Mn = np.zeros((d,) * n)
    for idx in itertools.product(range(d), repeat=n):
      v = 0.0
      for r in range(n + 1):
        for S in itertools.combinations(range(n), r):
          rest = [t for t in range(n) if t not in S]
          coef = (-1.0) ** (n - r) * np.prod([dmu[idx[t]] for t in rest]) if rest else 1.0
          v += coef * nhat[degree_offset(r, d) + colex_rank(tuple(sorted(idx[t] for t in S)))]
      Mn[idx] = v
    total += a[n] * float(np.sum(Mn * Tn))
  return total


def main():
  L, M, d = 8, 600, int(sys.argv[1]) if len(sys.argv) > 1 else 5
  syn = make_svgp(L, M, d, seed=1002, ls_bounds=(0.7, 3.0))
  mu, Sig = make_inputs(2, d, seed=2000, scale=0.1, lo=0.3, hi=0.7)
  po = mo.SVGPParams(Z=np.broadcast_to(syn.Z, (L, M, d)).copy(), lengthscales=syn.lengthscales, variance=syn.variance,
                     q_mu=syn.q_mu, q_sqrt=syn.q_sqrt, whiten=True)
  from scipy.linalg import solve_triangular
  beta = np.empty((L, M))
  for a in range(L):
    Kuu = mo.se_kernel(po.Z[a], None, po.lengthscales[a], po.variance[a]) + po.kuu_jitter * np.eye(M)
    beta[a] = solve_triangular(np.linalg.cholesky(Kuu).T, po.q_mu[:, a], lower=False)
  ls, var, Z = po.lengthscales, po.variance, po.Z
  b = 0
  S = Sig[b]
  w, P = [], []
  for a in range(L):
    Lam = np.diag(ls[a] ** 2); Pa = np.linalg.inv(S + Lam)
    ln = np.log(var[a]) + np.sum(np.log(ls[a])) - 0.5 * np.linalg.slogdet(S + Lam)[1]
    zeta = Z[a] - mu[b]
    w.append(beta[a] * np.exp(ln - 0.5 * np.einsum('id,de,ie->i', zeta, Pa, zeta))); P.append(Pa)
  tabs = [monomial_table(Z[a] - Z[a].mean(0)) for a in range(L)]
  for (a, a2) in [(0, 1), (2, 5), (3, 7)]:
    La, Lb = ls[a] ** 2, ls[a2] ** 2
    V = La * Lb / (La + Lb); Sv = S + np.diag(V)
    T = np.diag(V) @ np.linalg.solve(Sv, S); T = 0.5 * (T + T.T)
    G = T / La[:, None] / Lb[None, :]
    Dr = (S @ P[a]) / La[:, None]; Dr = 0.5 * (Dr + Dr.T) - T / La[:, None] / La[None, :]
    Dc = (S @ P[a2]) / Lb[:, None]; Dc = 0.5 * (Dc + Dc.T) - T / Lb[:, None] / Lb[None, :]
    const = (-0.5 * np.linalg.slogdet(Sv)[1] + 0.5 * np.sum(np.log(V)) - 0.5 * np.sum(np.log(La)) - 0.5 * np.sum(np.log(Lb))
             + 0.5 * np.linalg.slogdet(S + np.diag(La))[1] + 0.5 * np.linalg.slogdet(S + np.diag(Lb))[1])
    zbar_a, zbar_b = Z[a].mean(0), Z[a2].mean(0)
    zr = Z[a] - mu[b]; zc = Z[a2] - zbar_b
    A = zr @ G
    rho = -0.5 * np.einsum('id,de,ie->i', zr, Dr, zr) + const + A @ (zbar_b - mu[b])
    zc_mu = Z[a2] - mu[b]
    gam = -0.5 * np.einsum('id,de,ie->i', zc_mu, Dc, zc_mu)
    bij = A @ zc.T
    wh, wh2 = w[a] * np.exp(rho), w[a2] * np.exp(gam)
    coef = (1.0, 1.0, 0.5, C0, C1)
    dense = wh @ (sum(c * bij ** n for n, c in enumerate(coef))) @ wh2
    dense34 = wh @ (C0 * bij ** 3 + C1 * bij ** 4) @ wh2
    nhat, qhat = wh @ tabs[a], wh2 @ tabs[a2]
    dmu = mu[b] - zbar_a
    mom = poly_sum_from_moments(nhat, qhat, G, dmu, coef)
    mom34 = poly_sum_from_moments(nhat, qhat, G, dmu, (0, 0, 0, C0, C1))
    full = wh @ np.expm1(bij) @ wh2 + wh.sum() * wh2.sum()
    cs = np.sqrt((A * A).sum(1).max() * (zc * zc).sum(1).max())
    print(f"pair ({a},{a2}): dense {dense:+.12e} moments {mom:+.12e} diff {abs(dense - mom):.2e} | deg 3+4 part {dense34:+.3e} "
          f"(diff {abs(dense34 - mom34):.1e}) | exact sum {full:+.12e}, poly error {abs(full - dense):.2e} | max|b| {np.abs(bij).max():.4f} "
          f"CS bound {cs:.4f} | sum|w||w'| {np.abs(wh).sum() * np.abs(wh2).sum():.2e}")


if __name__ == '__main__':
  main()
