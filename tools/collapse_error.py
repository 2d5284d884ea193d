"""Truncation error of the (not built) moment collapse of the off-diagonal sum, DESIGN.md section 4.

For sampled (step, batch element, pair) of the C3 bench rollout: S = sum_ij what_i what'_j exp(b_ij) with
b_ij = A_i . zc_j, against its series truncated after degree N (evaluated densely in f64: the collapse
computes the same numbers from symmetric-tensor moments).  Prints the actual error, the rigorous bound
sum|what| sum|what'| beta^(N+1)/(N+1)! with beta = max||A_i|| max||zc_j||, and the size of S."""
import os, sys, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflowpilco_amd import ops
from gpflowpilco_amd.synthetic import make_inputs, make_svgp

L, M, d, H, B = 8, 2000, 8, 40, 256
dev = torch.device("cuda", 0)
syn = make_svgp(L, M, d, seed=1002, device=str(dev), ls_bounds=(0.7, 3.0))
model = syn.to_model(dev)
pm = model.packed(torch.float32, True, dev)
mu0, S0 = make_inputs(B, d, seed=2000, scale=0.1, lo=0.3, hi=0.7)
mu0 = torch.tensor(mu0, dtype=torch.float32, device=dev); S0 = torch.tensor(S0, dtype=torch.float32, device=dev)
out = ops.rollout_closed(pm, mu0, S0, H, keep_trajectory=True)
tmu, tS = out[-2], out[-1]
f64 = torch.float64
Z, ls, var, beta, _, _ = model.precompute(dev)            # [L,M,d], [L,d], [L], [L,M] float64
rows = []
for h in (0, 5, 10, 20, 39):
  for b in (0, 17, 101, 255):
    mu = (mu0 if h == 0 else tmu[h - 1])[b].double(); S = (S0 if h == 0 else tS[h - 1])[b].double()
    # q_a, w_a = beta_a q_a per latent
    w = []
    for a in range(L):
      La = ls[a] ** 2
      zeta = Z[a] - mu
      Sa = S + torch.diag(La)
      sol = torch.linalg.solve(Sa, zeta.T).T
      q = var[a] * torch.sqrt(torch.prod(La) / torch.linalg.det(Sa)) * torch.exp(-0.5 * (zeta * sol).sum(1))
      w.append(q * beta[a])
    for (a, a2) in ((0, 1), (2, 5), (3, 7), (1, 6), (4, 6), (0, 7)):
      La, Lb = ls[a] ** 2, ls[a2] ** 2
      V = La * Lb / (La + Lb)
      T = torch.diag(V) @ torch.linalg.solve(S + torch.diag(V), S); T = 0.5 * (T + T.T)
      G = T / La[:, None] / Lb[None, :]
      za, zb = Z[a] - mu, Z[a2] - mu
      # delta = const + rho_i + gamma_j + za G zb^T; fold rho, gamma (quadratic forms) into the weights
      Ea = torch.diag(1 / La) @ S @ torch.linalg.inv(S + torch.diag(La)); Eb = torch.diag(1 / Lb) @ S @ torch.linalg.inv(S + torch.diag(Lb))
      Dr = Ea - torch.diag(1 / La) @ T @ torch.diag(1 / La); Dc = Eb - torch.diag(1 / Lb) @ T @ torch.diag(1 / Lb)
      rho = -0.5 * ((za @ Dr) * za).sum(1); gam = -0.5 * ((zb @ Dc) * zb).sum(1)
      zc = Z[a2] - Z[a2].mean(0)
      A = za @ G                                   # A_i = G^T zeta_i  (row vector form)
      shift = A @ (mu - Z[a2].mean(0))             # A_i . (mu - zbar_a')
      wr = w[a] * torch.exp(rho - shift); wc = w[a2] * torch.exp(gam)
      bij = A @ zc.T
      exact = float(wr @ torch.exp(bij) @ wc)
      scale = float(wr.abs().sum() * wc.abs().sum())
      cs = float(A.norm(dim=1).max() * zc.norm(dim=1).max())
      term = torch.ones_like(bij); part = term.clone(); errs = {}
      for n in range(1, 7):
        term = term * bij / n; part = part + term
        if n >= 3: errs[n] = abs(float(wr @ part @ wc) - exact)
      rows.append((h, b, a, a2, float(bij.abs().max()), cs, scale, exact, errs))
print("step b pair   max|b|  CSbound  sum|w|sum|w'|   S_exact      err N=3    N=4      N=5      N=6   | rigorous bound N=4, N=5")
for (h, b, a, a2, mb, cs, sc, ex, e) in rows:
  bd = [sc * cs ** (n + 1) / math.factorial(n + 1) * math.exp(cs) for n in (4, 5)]
  print(f"{h:3d} {b:3d} ({a},{a2}) {mb:7.4f} {cs:7.4f} {sc:10.3e} {ex:12.4e}  " + " ".join(f"{e[n]:8.1e}" for n in (3, 4, 5, 6)) + f" | {bd[0]:8.1e} {bd[1]:8.1e}")
act = np.array([[r[8][n] for n in (3, 4, 5, 6)] for r in rows])
print("max actual error  N=3..6:", " ".join(f"{v:.1e}" for v in act.max(0)))
