"""Which polynomial tier do the f32 kernel's 64x32 wave tiles fall into along the C3 bench rollout?
Runs the closed rollout on the GPU, then re-derives b_ij = A_i . zc_j in torch f64 for sampled
(step, batch element, pair) and histograms the per-tile max |b|."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflowpilco_amd import ops
from gpflowpilco_amd.synthetic import make_inputs, make_svgp

L, M, d, H, B = 8, 2000, 8, 40, 256
dev = torch.device("cuda", 0)
syn = make_svgp(L, M, d, seed=1002, device=str(dev), ls_bounds=(0.7, 3.0))
pm = syn.to_model(dev).packed(torch.float32, True, dev)
mu0, S0 = make_inputs(B, d, seed=2000, scale=0.1, lo=0.3, hi=0.7)
mu0 = torch.tensor(mu0, dtype=torch.float32, device=dev); S0 = torch.tensor(S0, dtype=torch.float32, device=dev)
out = ops.rollout_closed(pm, mu0, S0, H, keep_trajectory=True)
tmu, tS = out[-2], out[-1]
Z = torch.tensor(syn.Z, dtype=torch.float64, device=dev); ls = torch.tensor(syn.lengthscales, dtype=torch.float64, device=dev)
if Z.ndim == 2: Z = Z.expand(L, M, d)
edges = [1 / 512, 1 / 256, 1 / 128, 0.015625, 1 / 32, 0.0625, 0.25, 0.5, 1.0]
for h in (0, 5, 10, 20, 39):
  hist = np.zeros(len(edges) + 1); n = 0; trS = []
  for b in (0, 17, 101, 255):
    mu = (mu0 if h == 0 else tmu[h - 1])[b].double(); S = (S0 if h == 0 else tS[h - 1])[b].double()
    trS.append(float(torch.diagonal(S).mean()))
    for (a, a2) in ((0, 1), (2, 5), (3, 7), (1, 6), (4, 6)):
      La, Lb = ls[a] ** 2, ls[a2] ** 2
      V = La * Lb / (La + Lb)
      T = torch.diag(V) @ torch.linalg.solve(S + torch.diag(V), S); T = 0.5 * (T + T.T)
      G = T / La[:, None] / Lb[None, :]
      A = (Z[a] - mu) @ G
      zc = Z[a2] - Z[a2].mean(0)
      bij = (A @ zc.T).abs()[:1984, :1984].reshape(31, 64, 62, 32).amax(dim=(1, 3)).flatten().cpu().numpy()
      hist += np.histogram(bij, bins=[0] + edges + [1e30])[0]; n += bij.size
  names = ["<=1/512", "<=1/256", "<=1/128", "<=1/64", "<=1/32", "<=1/16", "<=1/4", "<=1/2", "<=1", ">1"]
  print(f"step {h:2d} mean diag(S) {np.mean(trS):.4f}  tile max|b|: " + " ".join(f"{nm} {v / n:.2f}" for nm, v in zip(names, hist)))
