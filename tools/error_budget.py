"""Error budget of the f32 off-diagonal reduce at the C3 bench configuration (numpy, fp64 truth).

For a few (b, pair) it forms S = sum_ij what_i expm1(b_ij) what'_j and perturbs one ingredient at a
time with the rounding model of a candidate kernel design; printed is |dS| relative to max|Sff|.
"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from gpflowpilco_amd.synthetic import make_svgp, make_inputs
from oracle import mm_oracle as mo, mm_fused_ref as fr

L, M, d = 8, 2000, 8
syn = make_svgp(L, M, d, seed=1002, ls_bounds=(0.7, 3.0))
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
mu, Sig = make_inputs(2, d, seed=2000, scale=scale, lo=0.3, hi=0.7)
po = mo.SVGPParams(Z=np.broadcast_to(syn.Z, (L, M, d)).copy(), lengthscales=syn.lengthscales, variance=syn.variance,
                   q_mu=syn.q_mu, q_sqrt=syn.q_sqrt, whiten=True)
beta, _ = fr.precompute(po) if False else (None, None)
# beta only (skip the O(M^3) C): Kuu^-1 u
beta = np.empty((L, M))
for a in range(L):
  Kuu = mo.se_kernel(po.Z[a], None, po.lengthscales[a], po.variance[a]) + po.kuu_jitter * np.eye(M)
  Lu = np.linalg.cholesky(Kuu)
  from scipy.linalg import solve_triangular
  beta[a] = solve_triangular(Lu.T, po.q_mu[:, a], lower=False)
ls, var, Z = po.lengthscales, po.variance, po.Z
rng = np.random.default_rng(0)
b = 0
S = Sig[b]
w, P = [], []
for a in range(L):
  Lam = np.diag(ls[a] ** 2); Pa = np.linalg.inv(S + Lam)
  ln = np.log(var[a]) + np.sum(np.log(ls[a])) - 0.5 * np.linalg.slogdet(S + Lam)[1]
  zeta = Z[a] - mu[b]
  w.append(beta[a] * np.exp(ln - 0.5 * np.einsum('id,de,ie->i', zeta, Pa, zeta))); P.append(Pa)
f1 = np.array([x.sum() for x in w])
print("f1", np.round(f1, 3), " |w| max", [f"{np.abs(x).max():.2e}" for x in w][:3])
Sff_scale = None
rows = []
for (a, a2) in [(0, 1), (2, 5), (3, 7), (1, 6)]:
  La, Lb = ls[a] ** 2, ls[a2] ** 2
  V = La * Lb / (La + Lb); Sv = S + np.diag(V)
  T = np.diag(V) @ np.linalg.solve(Sv, S); T = 0.5 * (T + T.T)
  G = T / La[:, None] / Lb[None, :]
  Dr = (S @ P[a]) / La[:, None]; Dr = 0.5 * (Dr + Dr.T) - T / La[:, None] / La[None, :]
  Dc = (S @ P[a2]) / Lb[:, None]; Dc = 0.5 * (Dc + Dc.T) - T / Lb[:, None] / Lb[None, :]
  const = (-0.5 * np.linalg.slogdet(Sv)[1] + 0.5 * np.sum(np.log(V)) - 0.5 * np.sum(np.log(La)) - 0.5 * np.sum(np.log(Lb))
           + 0.5 * np.linalg.slogdet(S + np.diag(La))[1] + 0.5 * np.linalg.slogdet(S + np.diag(Lb))[1])
  zr = Z[a] - mu[b]; zbar = Z[a2].mean(0); zc = Z[a2] - zbar
  A = zr @ G                                    # [M, d]
  rho = -0.5 * np.einsum('id,de,ie->i', zr, Dr, zr) + const + A @ (zbar - mu[b])
  zc_mu = Z[a2] - mu[b]
  gam = -0.5 * np.einsum('id,de,ie->i', zc_mu, Dc, zc_mu)
  bij = A @ zc.T
  wh, wh2 = w[a] * np.exp(rho), w[a2] * np.exp(gam)
  E = np.expm1(bij)
  core = wh @ E @ wh2
  Strue = core + wh.sum() * wh2.sum() - w[a].sum() * w[a2].sum()
  terms = np.abs(wh)[:, None] * np.abs(wh2)[None, :]
  u = 2.0 ** -24
  res = {}
  res["S"] = Strue
  res["core"] = core
  res["max|b|"] = np.abs(bij).max()
  # (1) polynomial: relative error 2.3e-7 on E
  res["poly 2.3e-7 rel"] = abs(wh @ (E * 2.3e-7 * rng.uniform(-1, 1, E.shape)) @ wh2)
  # (2) v_exp_f32 then -1: absolute error ~1 ulp(1) = 1.2e-7 * U(-.5,.5) on E
  res["v_exp abs 6e-8"] = abs(wh @ (2 * u * rng.uniform(-1, 1, E.shape)) @ wh2)
  # (3) b from f16 2-way split: abs error 2^-22 |A|inf |z|inf sqrt(d)
  db = 2.0 ** -22 * np.abs(A).max() * np.abs(zc).max() * np.sqrt(d) * rng.uniform(-1, 1, E.shape)
  res["b f16x2 split"] = abs(wh @ (np.expm1(bij + db) - E) @ wh2)
  db = 2.0 ** -24 * np.sqrt(np.abs(A) ** 2 @ (np.abs(zc) ** 2).T) * rng.uniform(-1, 1, E.shape)
  res["b bf16x3 / f32"] = abs(wh @ (np.expm1(bij + db) - E) @ wh2)
  # (4) weights rounded to f32
  res["what f32"] = abs(wh.astype(np.float32).astype(np.float64) @ E @ wh2.astype(np.float32).astype(np.float64) - core)
  # (5) f32 accumulation of 32-term lane partials: ~ u * sqrt(32) * |partial|
  res["O(M) correction f64?"] = abs(wh.sum() * wh2.sum()), abs(w[a].sum() * w[a2].sum())
  rows.append(((a, a2), res))
smax = max(abs(r["S"]) for _, r in rows)
for pr, r in rows:
  print(pr, f"S={r['S']:.4e} core={r['core']:.4e} max|b|={r['max|b|']:.3f}")
  for k, v in r.items():
    if k in ("S", "core", "max|b|"): continue
    if isinstance(v, tuple): print(f"    {k}: {v[0]:.3e} {v[1]:.3e}")
    else: print(f"    {k:18s}: |dS| = {v:.2e}   / max|S offdiag| = {v / smax:.1e}")
