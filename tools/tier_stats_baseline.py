"""Per-tile and per-entry |b| histogram of the f32 off-diagonal reduce on BASELINE.md's recipe at C3 (torch f64 on the GPU; every 16th
batch element, all 28 pairs).  Result (round 3): 64 x 32 wave tiles by max|b|: <= 1/20 9 %, (1/20, 1/4] 76 %, (1/4, 1/2] 14 %, above 1 %;
ENTRIES by |b|: <= 1/20 82 %, (1/20, 1/4] 18 %.  The tier of a tile is its maximum over 2048 roughly Gaussian b_ij (~3.6 sigma), so
the degree-3 tier is paid on tiles whose typical entry is far inside the first tier."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflowpilco_amd.synthetic import make_inputs, make_svgp
L, M, d, B = 8, 2000, 8, 256
dev = torch.device("cuda", 0)
syn = make_svgp(L, M, d, seed=1002, device=str(dev), ls_bounds=(0.3, 3.0), stable=False)
mu0, S0 = make_inputs(B, d, seed=3002, scale=0.1, lo=0.0, hi=1.0)
Z = torch.tensor(syn.Z, dtype=torch.float64, device=dev).expand(L, M, d); ls = torch.tensor(syn.lengthscales, dtype=torch.float64, device=dev)
edges = [0.05, 0.25, 0.5, 1.0, 2.0, 4.0]
hist = np.zeros(len(edges) + 1); n = 0
ent = np.zeros(len(edges) + 1); ne = 0
import itertools
pairs = list(itertools.combinations(range(L), 2))
for b in range(0, 256, 16):
  mu = torch.tensor(mu0[b], dtype=torch.float64, device=dev); S = torch.tensor(S0[b], dtype=torch.float64, device=dev)
  for (a, a2) in pairs:
    La, Lb = ls[a] ** 2, ls[a2] ** 2
    V = La * Lb / (La + Lb)
    T = torch.diag(V) @ torch.linalg.solve(S + torch.diag(V), S); T = 0.5 * (T + T.T)
    G = T / La[:, None] / Lb[None, :]
    A = (Z[a] - mu) @ G
    zc = Z[a2] - Z[a2].mean(0)
    bb = (A @ zc.T).abs()[:1984, :1984]
    t = bb.reshape(31, 64, 62, 32).amax(dim=(1, 3)).flatten().cpu().numpy()
    hist += np.histogram(t, bins=[0] + edges + [1e30])[0]; n += t.size
    e = bb.flatten().cpu().numpy()
    ent += np.histogram(e, bins=[0] + edges + [1e30])[0]; ne += e.size
names = ["<=1/20", "<=1/4", "<=1/2", "<=1", "<=2", "<=4", ">4"]
print("tile max|b| :", " ".join(f"{nm} {v / n:.3f}" for nm, v in zip(names, hist)))
print("entry |b|   :", " ".join(f"{nm} {v / ne:.3f}" for nm, v in zip(names, ent)))
