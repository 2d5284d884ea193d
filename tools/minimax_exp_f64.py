"""Near-minimax polynomials of e^x on [-h, h] for the range tiers of the f64 reduce kernel's f32-model (LOWP) path
(csrc/mm_f64.hip):  e^x ~ 1 + x (c1 + c2 x + ... + cn x^(n-1)), minimising the ABSOLUTE error of e^x.  One degree
below the Taylor polynomial of the same accuracy (the error of the degree-n minimax is ~ h^(n+1) / (2^n (n+1)!)).
Fitted in 80-bit long double (the target errors are a few 1e-16), reported as f64 literals with the error of the
f64-rounded polynomial evaluated in long double.

  python tools/minimax_exp_f64.py
"""
import numpy as np

LD = np.longdouble


def fit(h, n, iters=80, npts=6001):
  k = np.arange(npts, dtype=LD)
  x = np.cos(LD(np.pi) * (k + LD(0.5)) / npts) * LD(h)
  xs = np.where(np.abs(x) > 1e-6, x, LD(1))
  q = np.where(np.abs(x) > 1e-6, np.expm1(xs) / xs, 1 + x / 2 + x * x / 6 + x ** 3 / 24)      # (e^x - 1) / x
  w = np.ones(npts, dtype=LD)
  best = None
  for _ in range(iters):
    V = np.vander((x / LD(h)).astype(LD), n, increasing=True).astype(LD)
    A = (V * (w * np.abs(x))[:, None]).astype(np.float64)          # weight |x|: absolute error of e^x
    b = (q * w * np.abs(x)).astype(np.float64)
    c0 = np.linalg.lstsq(A, b, rcond=None)[0].astype(LD)
    # one step of iterative refinement in long double (the f64 solve leaves ~1e-16 relative)
    r = (q - V @ c0) * w * np.abs(x)
    dc = np.linalg.lstsq(A, r.astype(np.float64), rcond=None)[0].astype(LD)
    c = c0 + dc
    err = np.abs(V @ c - q) * np.abs(x)
    if best is None or err.max() < best[0]:
      best = (err.max(), c.copy())
    w = w * (1 + 3 * err / err.max()); w /= w.mean()
  c = best[1] / LD(h) ** np.arange(n)
  return c


def check(c64, h):
  x = np.linspace(-h, h, 400001).astype(LD)
  p = np.full_like(x, LD(c64[-1]))
  for k in range(len(c64) - 2, -1, -1):
    p = p * x + LD(c64[k])
  return float(np.abs(1 + p * x - np.exp(x)).max())


if __name__ == "__main__":
  import math
  for h, n, taylor in ((1 / 64, 5, 6), (1 / 32, 6, 7), (1 / 16, 7, 8), (1 / 8, 8, 9), (1 / 4, 9, 10), (3 / 4, 12, 15)):
    c = fit(h, n)
    c64 = c.astype(np.float64)
    print(f"// |x| <= {h:g}: degree {n}, max |p - e^x| = {check(c64, h):.1e}   (Taylor degree {taylor}: {h ** (taylor + 1) / math.factorial(taylor + 1):.1e})")
    print("  {" + ", ".join(f"{v:.17e}" for v in c64) + "},")
