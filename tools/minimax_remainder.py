"""Near-minimax polynomials R(x) ~ (expm1(x) - x - x^2/2) / x^3 by range tier, for the f32 off-diagonal
kernel's remainder form (linear and quadratic terms are taken exactly from f64 moments).
Requirement: |x^3 R(x) - r(x)| <= ~5e-8 |x|  (the f32 rounding level of a full expm1), evaluated with
fused f32 Horner steps."""
import numpy as np
from numpy.polynomial import chebyshev as C


def target(x):
  x = np.asarray(x, dtype=np.float64)
  small = np.abs(x) < 1e-3
  xs = np.where(small, 1.0, x)
  full = (np.expm1(xs) - xs - 0.5 * xs * xs) / xs ** 3
  series = 1 / 6 + x / 24 + x * x / 120 + x ** 3 / 720
  return np.where(small, series, full)


def fit(Rg, deg, iters=60):
  n = 4001
  x = np.cos(np.pi * (np.arange(n) + 0.5) / n) * Rg
  f = target(x)
  w = np.ones(n)
  best = None
  for _ in range(iters):
    V = np.vander(x / Rg, deg + 1, increasing=True)
    c, *_ = np.linalg.lstsq(V * w[:, None], f * w, rcond=None)
    err = np.abs(V @ c - f) * np.abs(x) ** 2          # absolute error of x^3 R relative to |x|
    if best is None or err.max() < best[0]:
      best = (err.max(), c.copy())
    w = w * (1 + 3 * err / err.max()); w /= w.mean()
  return best[1] / Rg ** np.arange(deg + 1), best[0]


def check_f32(coef, Rg):
  c32 = coef.astype(np.float32)
  x = np.linspace(-Rg, Rg, 200001).astype(np.float32)
  p = np.full_like(x, c32[-1])
  for k in range(len(c32) - 2, -1, -1):
    p = (p.astype(np.float64) * x.astype(np.float64) + c32[k]).astype(np.float32)      # fused: one rounding
  t = (x.astype(np.float64) ** 2).astype(np.float32)
  v = (t.astype(np.float64) * p.astype(np.float64)).astype(np.float32)
  r = (v.astype(np.float64) * x.astype(np.float64))                                      # (t * R) * x
  xd = x.astype(np.float64)
  ref = np.expm1(xd) - xd - 0.5 * xd * xd
  m = np.abs(xd) > 0
  return (np.abs(r - ref)[m] / np.abs(xd)[m]).max()


for Rg, deg in [(1 / 16, 0), (1 / 16, 1), (1 / 16, 2), (1 / 4, 2), (1 / 4, 3), (1 / 2, 3), (1 / 2, 4), (1.0, 5), (1.0, 6)]:
  coef, e = fit(Rg, deg)
  print(f"R={Rg:<7.4f} deg={deg}  fit err/|x| = {e:.2e}   f32-evaluated err/|x| = {check_f32(coef, Rg):.2e}")
  print("    ", ", ".join(f"{v:.9e}f" for v in coef.astype(np.float32)))
