"""Distribution of the per-wave-tile range of delta_ij (and of a_ij = delta_ij - gamma_j) at the C3
bench configuration: which polynomial tier (|x| <= 0.25 / 0.5 / 1) the f32 kernel's tiles fall into."""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from gpflowpilco_amd.synthetic import make_svgp, make_inputs

L, M, d = 8, 2000, 8
syn = make_svgp(L, M, d, seed=1002, ls_bounds=(0.7, 3.0))
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
mu, Sig = make_inputs(4, d, seed=2000, scale=scale, lo=0.3, hi=0.7)
Z = syn.Z; ls = syn.lengthscales


def delta(b, a, a2):
  La = ls[a] ** 2; Lb = ls[a2] ** 2; V = La * Lb / (La + Lb)
  S = Sig[b] + np.diag(V); Si = np.linalg.inv(S)
  T = np.diag(V) @ Si @ Sig[b]; T = 0.5 * (T + T.T)
  G = np.diag(1 / La) @ T @ np.diag(1 / Lb)
  Pa = np.linalg.inv(Sig[b] + np.diag(La)); Pb = np.linalg.inv(Sig[b] + np.diag(Lb))
  Dr = np.diag(1 / La) - np.diag(1 / La) @ T @ np.diag(1 / La) - Pa
  Dc = np.diag(1 / Lb) - np.diag(1 / Lb) @ T @ np.diag(1 / Lb) - Pb
  logk = -0.5 * (np.linalg.slogdet(S)[1] - np.sum(np.log(V)))
  lna = 0.5 * np.sum(np.log(La)) - 0.5 * np.linalg.slogdet(Sig[b] + np.diag(La))[1]
  lnb = 0.5 * np.sum(np.log(Lb)) - 0.5 * np.linalg.slogdet(Sig[b] + np.diag(Lb))[1]
  zeta = Z - mu[b]
  rho = -0.5 * np.einsum('id,de,ie->i', zeta, Dr, zeta)
  gam = -0.5 * np.einsum('id,de,ie->i', zeta, Dc, zeta)
  D = rho[:, None] + gam[None, :] + (logk - lna - lnb) + zeta @ G @ zeta.T
  return D, gam


for (a, a2) in [(0, 1), (2, 5), (3, 7), (1, 6)]:
  D, gam = delta(0, a, a2)
  A = D - gam[None, :]
  out = []
  for name, X in (("delta", D), ("a", A)):
    Xp = np.abs(X[:1984, :1984]).reshape(31, 64, 62, 32).max(axis=(1, 3))
    out.append(f"{name}: max {np.abs(X).max():.3f} tiers<=.25/.5/1/>1: {(Xp<=.25).mean():.2f}/{((Xp>.25)&(Xp<=.5)).mean():.2f}/{((Xp>.5)&(Xp<=1)).mean():.2f}/{(Xp>1).mean():.2f}")
  print((a, a2), ' | '.join(out), ' |gamma| max', np.abs(gam).max())
