"""numpy studies behind the pack's norm order, the diagonal recentring, the row-group collapse and the Cauchy-Schwarz prefix skip
(DESIGN.md 2.2, 4.5), on the bench's own BASELINE-recipe draws at C3 (every 32nd / 64th batch element of the first step):

  1. f64 diagonal sweep: mean degree of the per-tile e^b polynomial (FMAs per entry) as the operands were, with the columns
     recentred, with the points in norm order, with both;
  2. f32 off-diagonal sweep: the dense items' wave tiles under 1/4 with and without the order;
  3. row-group collapse: the area of a dense item that stays collapsed when the predicate is taken per 64-row group (rows only,
     columns only, both), and the fraction of all row groups collapsed against the bound;
  4. Cauchy-Schwarz prefix skip: the screened wave tiles a collapsed group can skip from its own rows' norm and the column
     tiles' norms alone, against what the screening product finds.

  python tools/pack_order_study.py"""
import itertools, os, sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflowpilco_amd.synthetic import make_inputs, make_svgp
L, M, d, B, H = 8, 2000, 8, 256, 40
syn = make_svgp(L, M, d, seed=1002, ls_bounds=(0.3, 3.0), stable=False)
mu_all, S_all = make_inputs(B * H, d, seed=2000 + 1002, scale=0.1, lo=0.0, hi=1.0)
Z = np.broadcast_to(syn.Z, (L, M, d)); ls = syn.lengthscales
def deg(t):   # POLYC FMAs per entry by tile max
  return np.select([t < 1/64, t < 1/32, t < 1/16, t < 1/8, t < 1/4, t < 3/4], [5, 6, 7, 8, 9, 12], 15)
res = {k: [] for k in ("cur", "sort", "sort_rc", "rc")}
hist = {k: [] for k in res}
for b in range(0, B, 64):
  mu, S = mu_all[b], S_all[b]
  for a in range(L):
    La = ls[a] ** 2
    V = La / 2
    T = np.diag(V) @ np.linalg.solve(S + np.diag(V), S); T = 0.5 * (T + T.T)
    G = T / La[:, None] / La[None, :]
    zb = Z[a].mean(0)
    nrm = np.sqrt((((Z[a] - zb) / ls[a]) ** 2).sum(1))
    order = np.argsort(nrm)
    for key in res:
      Zp = Z[a][order] if key.startswith("sort") else Z[a]
      zc = Zp - (zb if key.endswith("rc") else mu)
      bb = np.abs(zc @ G @ zc.T)[:1984, :1984]
      t = bb.reshape(62, 32, 62, 32).max(axis=(1, 3))
      iu = np.triu_indices(62)
      res[key].append(deg(t[iu]).mean()); hist[key].append(t[iu])
for k in res:
  t = np.concatenate(hist[k])
  print(k, "mean poly FMAs/entry %.2f" % np.mean(res[k]), "tile max cum:", " ".join(f"<{e:g}:{np.mean(t < e):.2f}" for e in (1/64, 1/32, 1/16, 1/8, 1/4, 3/4)))
# actual current diag form: rows zeta, cols mu-centred
print("---- diag: actual current (rows recentred, cols mu-centred) vs full rc vs sorted")
res2 = {k: [] for k in ("half", "half_sort")}
for b in range(0, B, 64):
  mu, S = mu_all[b], S_all[b]
  for a in range(L):
    La = ls[a] ** 2; V = La / 2
    T = np.diag(V) @ np.linalg.solve(S + np.diag(V), S); T = 0.5 * (T + T.T)
    G = T / La[:, None] / La[None, :]
    zb = Z[a].mean(0)
    order = np.argsort(np.sqrt((((Z[a] - zb) / ls[a]) ** 2).sum(1)))
    for key in res2:
      Zp = Z[a][order] if key.endswith("sort") else Z[a]
      bb = np.abs((Zp - zb) @ G @ (Zp - mu).T)[:1984, :1984]
      t = bb.reshape(62, 32, 62, 32).max(axis=(1, 3))
      res2[key].append(deg(t[np.triu_indices(62)]).mean())
for k in res2: print(k, "mean poly FMAs/entry %.2f" % np.mean(res2[k]))
print("---- off-diagonal (f32 sweep): 64x32 wave tiles")
pairs = list(itertools.combinations(range(L), 2))
out = {k: {"inside": [], "dense_tiles": 0, "tiles": 0, "skip": 0, "scr": 0} for k in ("cur", "sort")}
for b in range(0, B, 64):
  mu, S = mu_all[b], S_all[b]
  for (a, a2) in pairs:
    La, Lb = ls[a] ** 2, ls[a2] ** 2
    V = La * Lb / (La + Lb)
    T = np.diag(V) @ np.linalg.solve(S + np.diag(V), S); T = 0.5 * (T + T.T)
    G = T / La[:, None] / Lb[None, :]
    for key in out:
      Za, Zb = Z[a], Z[a2]
      if key == "sort":
        Za = Za[np.argsort((((Za - Za.mean(0)) / ls[a]) ** 2).sum(1))]; Zb = Zb[np.argsort((((Zb - Zb.mean(0)) / ls[a2]) ** 2).sum(1))]
      A = (Za - Za.mean(0)) @ G; zc = Zb - Zb.mean(0)
      X = np.sqrt((A * A).sum(1).max() * (zc * zc).sum(1).max())
      o = out[key]
      if X * X <= 0.998 / 16: o["inside"].append(1); continue
      o["inside"].append(0)
      bb = np.abs(A @ zc.T)[:1984, :1984]
      t = bb.reshape(31, 64, 62, 32).max(axis=(1, 3)).ravel()
      if X <= 0.5:
        o["scr"] += t.size; o["skip"] += int((t <= 0.25).sum())
      else:
        o["dense_tiles"] += t.size
        o.setdefault("dense_hist", []).append(t)
for k, o in out.items():
  dh = np.concatenate(o["dense_hist"]) if "dense_hist" in o else np.zeros(1)
  print(k, "inside %.3f" % np.mean(o["inside"]), "screened tiles", o["scr"], "skipped %.3f" % (o["skip"] / max(o["scr"], 1)), "dense tiles", o["dense_tiles"],
        "dense tile max cum:", " ".join(f"<{e:g}:{np.mean(dh < e):.2f}" for e in (1/16, 1/8, 1/4, 1/2, 1.0, 2.0)))

# ---- 3. row-group collapse -----------------------------------------------------------------------------------------------
Zs = []
for a in range(L):
  zb = Z[a].mean(0); o = np.argsort((((Z[a] - zb) / ls[a]) ** 2).sum(1)); Zs.append(Z[a][o] - zb)
pairs = list(itertools.combinations(range(L), 2))
fr, sk, nd = [], [], 0
for b in range(0, B, 32):
  mu, S = mu_all[b], S_all[b]
  for (a, a2) in pairs:
    La, Lb = ls[a] ** 2, ls[a2] ** 2
    V = La * Lb / (La + Lb)
    T = np.diag(V) @ np.linalg.solve(S + np.diag(V), S); T = 0.5 * (T + T.T)
    G = T / La[:, None] / Lb[None, :]
    A = Zs[a] @ G; zc = Zs[a2]
    an = (A * A).sum(1); zn = (zc * zc).sum(1)
    if an.max() * zn.max() <= 0.25: continue
    nd += 1
    ag = an[:1984].reshape(31, 64).max(1)           # per 64-row group
    zt = np.maximum.accumulate(zn[:1984].reshape(62, 32).max(1))   # prefix max per 32-col tile
    best = (0, 0, None)
    for jc in range(1, 63):
      rows = ag * zt[jc - 1] <= 0.25
      area = rows.sum() * jc
      if area > best[0]: best = (area, jc, rows)
    area, jc, rows = best
    fr.append(area / (31 * 62))
    if rows is not None and rows.any():
      bb = np.abs(A[:1984] @ zc[:1984].T).reshape(31, 64, 62, 32).max(axis=(1, 3))
      sk.append((bb[rows][:, :jc] <= 0.25).mean())
print("dense items", nd, "inner-area fraction mean %.3f  (min %.3f, max %.3f)" % (np.mean(fr), np.min(fr), np.max(fr)), " inner tiles skippable %.3f" % np.mean(sk))
fc, frw = [], []
for b in range(0, B, 32):
  mu, S = mu_all[b], S_all[b]
  for (a, a2) in pairs:
    La, Lb = ls[a] ** 2, ls[a2] ** 2
    V = La * Lb / (La + Lb)
    T = np.diag(V) @ np.linalg.solve(S + np.diag(V), S); T = 0.5 * (T + T.T)
    G = T / La[:, None] / Lb[None, :]
    A = Zs[a] @ G; zc = Zs[a2]
    an = (A * A).sum(1); zn = (zc * zc).sum(1)
    if an.max() * zn.max() <= 0.25: continue
    zt = np.maximum.accumulate(zn[:1984].reshape(62, 32).max(1))
    fc.append((an.max() * zt <= 0.25).sum() / 62)
    ag = an[:1984].reshape(31, 64).max(1)
    frw.append((ag * zn.max() <= 0.25).mean())
print("cols-only inner fraction %.3f, rows-only %.3f" % (np.mean(fc), np.mean(frw)))
cnt = {0.25: 0, 0.5: 0, 1.0: 0, 2.0: 0}; tot = 0
for b in range(0, B, 16):
  mu, S = mu_all[b], S_all[b]
  for (a, a2) in pairs:
    La, Lb = ls[a] ** 2, ls[a2] ** 2
    V = La * Lb / (La + Lb)
    T = np.diag(V) @ np.linalg.solve(S + np.diag(V), S); T = 0.5 * (T + T.T)
    G = T / La[:, None] / Lb[None, :]
    A = Zs[a] @ G; zc = Zs[a2]
    an = (A * A).sum(1); zn = (zc * zc).sum(1)
    ag = an[:1984].reshape(31, 64).max(1)
    tot += 31
    for k in cnt: cnt[k] += int((ag * zn.max() <= k).sum())
print("row groups collapsed at bound^2 <=", {k: round(v / tot, 4) for k, v in cnt.items()})

# ---- 4. Cauchy-Schwarz prefix skip ------------------------------------------------------------------------------------------
pairs = list(itertools.combinations(range(L), 2))
tot = cs = act = 0; n_items = 0
for b in range(0, B, 32):
  mu, S = mu_all[b], S_all[b]
  for (a, a2) in pairs:
    La, Lb = ls[a] ** 2, ls[a2] ** 2
    V = La * Lb / (La + Lb)
    T = np.diag(V) @ np.linalg.solve(S + np.diag(V), S); T = 0.5 * (T + T.T)
    G = T / La[:, None] / Lb[None, :]
    A = Zs[a] @ G; zc = Zs[a2]
    an = (A * A).sum(1); zn = (zc * zc).sum(1)
    if an.max() * zn.max() <= 0.998 / 16: continue     # wholly inside: no sweep
    n_items += 1
    ag = an[:1984].reshape(31, 64).max(1); zt = zn[:1984].reshape(62, 32).max(1)
    inner = ag * zn.max() <= 0.25
    bnd = ag[:, None] * zt[None, :]
    skip_cs = (bnd <= 0.998 / 16) & inner[:, None]
    bb = np.abs(A[:1984] @ zc[:1984].T).reshape(31, 64, 62, 32).max(axis=(1, 3))
    skip_act = (bb <= 0.25) & inner[:, None]
    tot += inner.sum() * 62; cs += skip_cs.sum(); act += skip_act.sum()
print("items swept", n_items, "collapsed-group wave tiles", tot, "skippable by block CS bound %.3f" % (cs / tot), "by the screening product %.3f" % (act / tot))
