"""Algorithm-matched fp64 CPU restatement of the GPU path (TEST INFRASTRUCTURE ONLY).

Same outputs as ``mm_oracle.mm_gauss_svgp_mo`` (moment_matching/models.py:200-299) but
through the reformulation the HIP kernels use (DESIGN.md "Centred fused reduce"):

  beta_a = Kuu_a^-1 u_a,   C_a = Kuu_a^-1 S_a Kuu_a^-1 - Kuu_a^-1      (once per model)
  f1_a   = sum_i w_i,  w_i = beta_i q_i
  Sff_aa' = sum_ij w_i expm1(delta_ij) w'_j + [a == a'] (var_a + sum_ij C_ij q_i exp(delta_ij) q_j)
  delta_ij = log Q_ij - log q_i - log q'_j

It is O(M^2) per kernel pair (no materialised [B,L,M,L,M] tensor, no O(M^3) solves), so
``bench.py`` can also time it as the "algorithm-matched" CPU baseline, and the tests use
it to show that the reformulation equals the literal reference algorithm.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import solve_triangular

from oracle import mm_oracle as mo


def precompute(model: mo.SVGPParams):
  """-> beta [L,M], C [L,M,M]   (models.py:216-235 hoisted out of the step)."""
  L, M, _ = model.Z.shape
  beta = np.empty((L, M)); C = np.empty((L, M, M))
  for a in range(L):
    Kuu = mo.se_kernel(model.Z[a], None, model.lengthscales[a], model.variance[a]) \
        + model.kuu_jitter * np.eye(M)
    Lu = np.linalg.cholesky(Kuu)
    v = model.q_mu[:, a].copy()
    S = np.tril(model.q_sqrt[a])
    if not model.whiten:
      v = solve_triangular(Lu, v, lower=True)
      S = solve_triangular(Lu, S, lower=True)
    beta[a] = solve_triangular(Lu.T, v, lower=False)
    A = S @ S.T - np.eye(M)
    X = solve_triangular(Lu.T, A, lower=False)
    Ca = solve_triangular(Lu.T, X.T, lower=False)
    C[a] = 0.5 * (Ca + Ca.T)
  return beta, C


def moment_match(mu, Sigma, model: mo.SVGPParams, beta, C, model_uncertainty=True):
  """-> f1 [B,L], Sff [B,L,L], Sigma^-1 Cov(x,f) [B,d,L] (latent space, before any W mixing)."""
  Z, ls, var = model.Z, model.lengthscales, model.variance
  L, M, d = Z.shape
  B = mu.shape[0]
  f1 = np.zeros((B, L)); Sff = np.zeros((B, L, L)); cross = np.zeros((B, d, L))
  for b in range(B):
    S = Sigma[b]
    lq, w, P, lognorm = [], [], [], []
    for a in range(L):
      Lam = np.diag(ls[a] ** 2)
      Pa = np.linalg.inv(S + Lam)
      ln = np.log(var[a]) + np.sum(np.log(ls[a])) - 0.5 * np.linalg.slogdet(S + Lam)[1]
      zeta = Z[a] - mu[b]
      lqa = ln - 0.5 * np.einsum('id,de,ie->i', zeta, Pa, zeta)
      qa = np.exp(lqa)
      wa = beta[a] * qa
      f1[b, a] = wa.sum()
      cross[b, :, a] = Pa @ (wa @ zeta)
      lq.append(lqa); w.append(wa); P.append(Pa); lognorm.append(ln)
    for a in range(L):
      for a2 in range(a, L):
        La, Lb = ls[a] ** 2, ls[a2] ** 2
        V = La * Lb / (La + Lb)
        Sv = S + np.diag(V)
        T = np.diag(V) @ np.linalg.solve(Sv, S)
        T = 0.5 * (T + T.T)
        G = T / La[:, None] / Lb[None, :]
        Dr = (S @ P[a]) / La[:, None]; Dr = 0.5 * (Dr + Dr.T) - T / La[:, None] / La[None, :]
        Dc = (S @ P[a2]) / Lb[:, None]; Dc = 0.5 * (Dc + Dc.T) - T / Lb[:, None] / Lb[None, :]
        const = (-0.5 * np.linalg.slogdet(Sv)[1] + 0.5 * np.sum(np.log(V))
                 - 0.5 * np.sum(np.log(La)) - 0.5 * np.sum(np.log(Lb))
                 + 0.5 * np.linalg.slogdet(S + np.diag(La))[1]
                 + 0.5 * np.linalg.slogdet(S + np.diag(Lb))[1])
        zr = Z[a] - mu[b]; zc = Z[a2] - mu[b]
        rho = -0.5 * np.einsum('id,de,ie->i', zr, Dr, zr)
        gam = -0.5 * np.einsum('id,de,ie->i', zc, Dc, zc)
        delta = rho[:, None] + gam[None, :] + const + zr @ G @ zc.T
        # Q_ij = q_i q_j e^{delta_ij} in the LOG domain: with lengthscales far below |z - mu| (the reference's own test designs
        # draw them down to 0.01) q underflows to 0 where e^{delta} overflows, although their product is bounded by var_a var_a'
        lQ = lq[a][:, None] + lq[a2][None, :] + delta
        Qn = np.exp(lQ)                                            # <k_a(z_i, x) k_a'(x, z_j)>
        qq = np.exp(lq[a][:, None] + lq[a2][None, :])              # q_i q'_j
        centred = np.where(np.abs(delta) < 1.0, qq * np.expm1(np.clip(delta, -1.0, 1.0)), Qn - qq)     # q q' (e^delta - 1)
        val = beta[a] @ centred @ beta[a2]
        if a == a2 and model_uncertainty:
          val += var[a] + np.sum(C[a] * Qn)
        Sff[b, a, a2] = Sff[b, a2, a] = val
  if model.mean_c is not None and model.W is None:
    f1 = f1 + np.asarray(model.mean_c)[None]
  return f1, Sff, cross
