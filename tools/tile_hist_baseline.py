"""Cumulative histogram of max|b| over the 64 x 32 wave tiles of the f32 off-diagonal reduce, BASELINE.md recipe at C3 (numpy fp64 on the
host: every 32nd batch element of the bench's own first-step draw, all 28 pairs) -- the statistic that decides how deep a moment
collapse has to be before the screening can skip most tiles.  Also the Cauchy-Schwarz bound per item."""
import itertools
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflowpilco_amd.synthetic import make_inputs, make_svgp

L, M, d, B, H = 8, 2000, 8, 256, 40
syn = make_svgp(L, M, d, seed=1002, ls_bounds=(0.3, 3.0), stable=False)
mu_all, S_all = make_inputs(B * H, d, seed=2000 + 1002, scale=0.1, lo=0.0, hi=1.0)      # bench.py's draw (independent steps)
Z = np.broadcast_to(syn.Z, (L, M, d)); ls = syn.lengthscales
edges = [0.05, 0.075, 0.10, 0.125, 0.15, 0.175, 0.20, 0.25, 0.3, 0.4, 0.5, 1.0]
tile_max, cs = [], []
pairs = list(itertools.combinations(range(L), 2))
step = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for b in range(0, B, step):
  mu, S = mu_all[b], S_all[b]
  for (a, a2) in pairs:
    La, Lb = ls[a] ** 2, ls[a2] ** 2
    V = La * Lb / (La + Lb)
    T = np.diag(V) @ np.linalg.solve(S + np.diag(V), S); T = 0.5 * (T + T.T)
    G = T / La[:, None] / Lb[None, :]
    A = (Z[a] - mu) @ G
    zc = Z[a2] - Z[a2].mean(0)
    bb = np.abs(A @ zc.T)[:1984, :1984]
    tile_max.append(bb.reshape(31, 64, 62, 32).max(axis=(1, 3)).ravel())
    cs.append(np.sqrt((A * A).sum(1).max() * (zc * zc).sum(1).max()))
t = np.concatenate(tile_max); cs = np.array(cs)
print("items", cs.size, "tiles", t.size)
print("tile max|b| cumulative:", " ".join(f"<={e:g}: {np.mean(t <= e):.3f}" for e in edges))
print("item CS bound cumulative:", " ".join(f"<={e:g}: {np.mean(cs <= e):.3f}" for e in edges + [2.0, 4.0]))
# per item: fraction of its tiles under a threshold, against its CS bound
per = np.array([[np.mean(x <= e) for e in (0.05, 0.15, 0.25)] for x in tile_max])
for lo, hi in ((0, 0.15), (0.15, 0.3), (0.3, 0.5), (0.5, 1.0), (1.0, 1e9)):
  m = (cs > lo) & (cs <= hi)
  if m.any():
    print(f"items with CS bound in ({lo}, {hi}]: {m.mean():.3f} of items; mean tile fraction <=0.05 / 0.15 / 0.25: "
          + " / ".join(f"{v:.3f}" for v in per[m].mean(0)))
