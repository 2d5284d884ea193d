"""CPU study (numpy, fp64 + emulated f32 operands) of WHERE the f32 off-diagonal reduce loses its digits, and what
re-reducing the wave tiles with max|b| above a threshold in f64 buys.

Per (b, pair) it forms the tile kernel's operands exactly as k_prep / k_pairvec_reg do (A_i, what_i, what'_j, zc'_j: f64),
rounds them to f32 the way the f32 pack stores them, evaluates the remainder sum  sum_ij what_i what'_j r(b_ij)  both ways
and splits the error by 64 x 32 wave tile according to the tile's max|b|.

  python tools/route_study.py draw13 | baseline [B] | wide
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rem(x):
  return np.expm1(x) - x - 0.5 * x * x


def pair_operands(Z, ls, var, beta, mu, Sigma, a, a2):
  """f64 operands of the off-diagonal pair (a, a2) for ONE batch element (mm_kernels.hip: k_prep, k_qvec, MODE 1 of k_pairvec_reg)."""
  d = Z.shape[-1]
  la, lb = ls[a] ** 2, ls[a2] ** 2

  def latent(x, lam, s2, bet):
    Pm = np.linalg.inv(Sigma + np.diag(lam))
    zeta = Z[x] - mu
    ld = np.linalg.slogdet(Sigma + np.diag(lam))[1]
    lognorm = np.log(s2) + 0.5 * np.log(lam).sum() - 0.5 * ld
    q = np.exp(lognorm - 0.5 * np.einsum("mi,ij,mj->m", zeta, Pm, zeta))
    E = np.diag(1.0 / lam) @ Sigma @ Pm
    E = 0.5 * (E + E.T)
    return zeta, bet * q, np.einsum("mi,ij,mj->m", zeta, E, zeta), ld

  za, wa, r1a, lda = latent(a, la, var[a], beta[a])
  zb, wb, r1b, ldb = latent(a2, lb, var[a2], beta[a2])
  V = la * lb / (la + lb)
  Sv = Sigma + np.diag(V)
  T = np.diag(V) @ np.linalg.solve(Sv, Sigma)
  T = 0.5 * (T + T.T)
  cst = -0.5 * np.linalg.slogdet(Sv)[1] - 0.5 * np.log(la + lb).sum() + 0.5 * lda + 0.5 * ldb
  sr, sc = za / la, zb / lb
  u = sr @ T
  A = u / lb
  tA = (sr * u).sum(-1)
  tg = np.einsum("mi,ij,mj->m", sc, T, sc)
  zbar = Z[a2].mean(0)
  corrA = A @ (mu - zbar)
  whr = wa * np.exp(-0.5 * (r1a - tA) + cst - corrA)
  whc = wb * np.exp(-0.5 * (r1b - tg))
  return A, whr, whc, Z[a2] - zbar, wa.sum(), wb.sum()


def study_item(A, whr, whc, zc, thresholds):
  f32 = np.float32
  bex = A @ zc.T
  full = np.outer(whr, whc) * np.exp(bex)
  Rex = np.outer(whr, whc) * rem(bex)
  # what the f32 pack sees: A, what, what' and zc rounded to f32; b as an f32-accurate product; r(b) and the weighting in f32
  A32, zc32 = A.astype(f32).astype(np.float64), zc.astype(f32).astype(np.float64)
  b32 = (A32 @ zc32.T).astype(f32).astype(np.float64)
  r32 = rem(b32).astype(f32).astype(np.float64)
  R32 = (whr.astype(f32).astype(np.float64)[:, None] * r32).astype(f32).astype(np.float64) * whc.astype(f32).astype(np.float64)[None, :]
  M1, M2 = bex.shape
  n1, n2 = (M1 + 63) // 64, (M2 + 31) // 32
  pad = lambda X: np.pad(X, ((0, n1 * 64 - M1), (0, n2 * 32 - M2)))
  tiles = lambda X: pad(X).reshape(n1, 64, n2, 32)
  tmax = np.abs(tiles(b32)).max(axis=(1, 3))
  terr = (tiles(R32) - tiles(Rex)).sum(axis=(1, 3))
  out = {"S": full.sum(), "abs": np.abs(full).sum(), "rem": Rex.sum(), "bmax": float(tmax.max()), "ntile": tmax.size}
  eps = 2.0 ** -24
  rho = lambda x: np.abs(rem(x)) + x * (np.expm1(x) - x)
  bcs = float(np.sqrt((A * A).sum(-1).max() * (zc * zc).sum(-1).max()))
  ra = np.pad(np.abs(whr), (0, n1 * 64 - M1)).reshape(n1, 64).sum(-1)
  ca = np.pad(np.abs(whc), (0, n2 * 32 - M2)).reshape(n2, 32).sum(-1)
  out["E1"] = eps * np.abs(whr).sum() * np.abs(whc).sum() * rho(bcs)
  out["E1t"] = eps * np.abs(whr).sum() * np.abs(whc).sum() * rho(out["bmax"])
  out["E2"] = eps * float((np.outer(ra, ca) * rho(tmax)).sum())
  out["Eq"] = eps * np.linalg.norm(whr) * np.linalg.norm(whc) * rho(out["bmax"])
  out["Eabs"] = eps * float(np.abs(Rex).sum())
  # the in-kernel estimator: per (32-row block, column) the block's max|b|, independent-rounding model
  bb = np.abs(pad(b32)).reshape(n1 * 2, 32, n2 * 32).max(axis=1)                       # [row blocks of 32][columns]
  rsq = np.pad(whr ** 2, (0, n1 * 64 - M1)).reshape(n1 * 2, 32).sum(-1)
  csq = np.pad(whc ** 2, (0, n2 * 32 - M2))
  out["E2q"] = eps * float(np.sqrt((rsq[:, None] * csq[None, :] * rho(bb) ** 2).sum()))
  out["Eind"] = eps * float(np.sqrt((Rex ** 2).sum()))
  out["bcs"] = bcs
  for t in thresholds:
    keep = tmax <= t
    out[t] = (float(terr[keep].sum()), int((~keep).sum()))
  return out


def run(name, Z, ls, var, beta, mus, Sigmas, thresholds=(np.inf, 4.0, 2.0, 1.0, 0.5, 0.25), max_items=None):
  L = Z.shape[0]
  pairs = [(a, a2) for a in range(L) for a2 in range(a + 1, L)]
  rows = []
  for b in range(len(mus)):
    for (a, a2) in pairs:
      A, whr, whc, zc, s1, s2 = pair_operands(Z, ls, var, beta, mus[b], Sigmas[b], a, a2)
      o = study_item(A, whr, whc, zc, thresholds)
      o["Sff"] = o["S"] - s1 * s2
      o["S2"] = o["S"] - o["rem"] - s1 * s2          # what the q stage knows: orders 0..2 from the moments
      o["id"] = (b, a, a2)
      rows.append(o)
      if max_items and len(rows) >= max_items:
        break
    if max_items and len(rows) >= max_items:
      break
  scale = max(abs(r["Sff"]) for r in rows)
  print(f"== {name}: {len(rows)} (b, pair) items, off-diagonal |Sff| scale {scale:.3e}, max|b| {max(r['bmax'] for r in rows):.2f}, "
        f"median item max|b| {np.median([r['bmax'] for r in rows]):.3f}, cancellation sum|.|/|S| up to {max(r['abs'] / max(abs(r['Sff']), 1e-300) for r in rows):.1e}")
  ntile = sum(r["ntile"] for r in rows)
  # the routing rule: est (E2q) > tau * max over the batch element's pairs of |S2|
  for tau in (1e-3, 3e-4, 1e-4, 3e-5):
    nroute, worst_rel, worst_all = 0, 0.0, 0.0
    for b in sorted({r["id"][0] for r in rows}):
      rb = [r for r in rows if r["id"][0] == b]
      sc2 = max(abs(r["S2"]) for r in rb)
      own = max(abs(r["Sff"]) for r in rb)
      for r in rb:
        worst_all = max(worst_all, abs(r[np.inf][0]) / own)
        if r["E2q"] > tau * sc2: nroute += 1
        else: worst_rel = max(worst_rel, abs(r[np.inf][0]) / own)
    print(f"   rule tau = {tau:.0e}: {nroute} of {len(rows)} items routed; worst f32 item error / own off-diagonal scale of its batch element: "
          f"kept {worst_rel:.2e} (all {worst_all:.2e})")
  print("   item: actual err | E1 (CS bound) | E1t (true max) | E2 (per tile) | Eq (2-norms) | E2q (in-kernel) | Eind | eps*sum|w w' r| | bcs  bmax  |Sff|")
  order = sorted(rows, key=lambda r: -abs(r[np.inf][0]))
  for r in order[:6] + order[len(order) // 2:len(order) // 2 + 2]:
    print(f"   {r['id']}: {abs(r[np.inf][0]):.2e} | {r['E1']:.2e} | {r['E1t']:.2e} | {r['E2']:.2e} | {r['Eq']:.2e} | {r['E2q']:.2e} | {r['Eind']:.2e} | {r['Eabs']:.2e} | {r['bcs']:.2f} {r['bmax']:.2f} {abs(r['Sff']):.2e}")
  for t in thresholds:
    worst = max(abs(r[t][0]) for r in rows)
    routed = sum(r[t][1] for r in rows)
    items = sum(1 for r in rows if r[t][1] > 0)
    print(f"   tiles with max|b| > {t:5}: {routed:8d} of {ntile} ({100.0 * routed / ntile:6.2f} %), items touched {items:5d}; "
          f"worst item error of the rest {worst:.3e} = {worst / scale:.2e} of the off-diagonal scale")
  return rows


def main():
  import torch
  from gpflowpilco_amd.synthetic import make_inputs, make_svgp
  from tests.helpers import gp_model_from_oracle, random_svgp_params
  which = sys.argv[1] if len(sys.argv) > 1 else "draw13"
  if which == "draw13":
    from tests.test_gpu_backward_f32 import BWD_DRAWS
    for idx in (13,) if len(sys.argv) < 3 else [int(x) for x in sys.argv[2:]]:
      c = BWD_DRAWS[idx]
      lo = 0.2 if c["d"] <= 2 else 0.5 * max(1.0, np.sqrt(c["d"] / 4.0))
      p = random_svgp_params(seed=c["seed"], L=c["L"], M=c["M"], d=c["d"], whiten=True, ls_bounds=(lo, 3.0 * lo), mean=True)
      model = gp_model_from_oracle(p, "cpu")
      Z, ls, var, beta, _, _ = model.precompute("cpu")
      rng = np.random.default_rng(c["seed"] + 1)
      mu = rng.uniform(0.25, 0.75, size=(c["B"], c["d"]))
      S = make_inputs(c["B"], c["d"], seed=c["seed"] + 2, scale=c["scale"] * (0.3 if c["d"] <= 2 else 1.0))[1]
      if c["L"] < 2:
        continue
      run(f"draw {idx} {c}", Z.numpy(), ls.numpy(), var.numpy(), beta.numpy(), mu, S)
  elif which in ("baseline", "pilco", "wide"):
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    L, M, d = 8, 2000, 8
    if which == "baseline":
      syn = make_svgp(L, M, d, seed=1002, stable=False)
      mu, S = make_inputs(B, d, seed=2002, scale=0.1)
    elif which == "pilco":
      syn = make_svgp(L, M, d, seed=1002, stable=True, ls_bounds=(0.7, 3.0))
      mu, S = make_inputs(B, d, seed=2002, scale=0.1, lo=0.3, hi=0.7)
    else:
      syn = make_svgp(L, M, d, seed=1002, stable=True, ls_bounds=(0.7, 3.0))
      mu, S = make_inputs(B, d, seed=2002, scale=0.25, lo=0.3, hi=0.7)
    model = syn.to_model("cpu")
    Z, ls, var, beta, _, _ = model.precompute("cpu")
    run(which, Z.numpy(), ls.numpy(), var.numpy(), beta.numpy(), mu, S)


if __name__ == "__main__":
  main()
