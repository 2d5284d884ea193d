"""numpy prototype of k_spoly56's partially-symmetric contraction (csrc/mm_moments6.hip):

    < N_n , G^{(x) n} Q_n > = sum_ij w_i w'_j (zc_i^T G zc'_j)^n          n = 5, 6

from PACKED symmetric moments (one entry per sorted index tuple, colex rank: mm_mono.h), G applied one index at a time on
tensors that are symmetric in the k transformed and in the n - k untransformed indices separately:

    T_{k+1}[I + {i}][J'] = sum_j G[i][j] T_k[I][J' + {j}],   i >= max(I)   (appending keeps I sorted: rank += C(i + k, k + 1))

with the model-independent index tables the kernel reads (ins: rank of J' + {j}; last: max element of I; mult: multinomials).
Checks the tables and the step against the brute-force double sum.
"""
import itertools
from math import comb, factorial

import numpy as np


def sym(k, d):
  return comb(d + k - 1, k)


def rank(t):                     # colex rank of a sorted tuple (mm_mono_rank)
  return sum(comb(v + i, i + 1) for i, v in enumerate(t))


def tuples(k, d):                # sorted tuples of length k in rank order
  ts = sorted(itertools.combinations_with_replacement(range(d), k), key=rank)
  assert [rank(t) for t in ts] == list(range(len(ts)))
  return ts


def build_tables(d, nmax=6):
  ins = {}                       # ins[m][J][j] = rank in sym(m + 1) of J + {j}
  last = {}                      # last[k][I] = max element of I (0 for k = 0)
  for m in range(nmax):
    ts = tuples(m, d)
    ins[m] = np.array([[rank(tuple(sorted(t + (j,)))) for j in range(d)] for t in ts], dtype=np.int64).reshape(len(ts), d)
    last[m] = np.array([t[-1] if m else 0 for t in ts], dtype=np.int64)
  mult = {}
  for n in (5, 6):
    mult[n] = np.array([factorial(n) / np.prod([factorial(c) for c in np.bincount(t, minlength=d)]) for t in tuples(n, d)])
  return ins, last, mult


def contract(Nn, Qn, G, n, d, ins, last, mult):
  T = Qn.copy().reshape(1, -1)                                  # [sym(0)][sym(n)]
  for k in range(n):
    nI, nJ = sym(k, d), sym(n - k - 1, d)
    out = np.full((sym(k + 1, d), nJ), np.nan)
    for I in range(nI):
      for J in range(nJ):
        v = T[I, ins[n - k - 1][J]]                             # the d entries T_k[I][J + {j}]
        for i in range(last[k][I], d):
          out[I + comb(i + k, k + 1), J] = G[i] @ v
    assert not np.isnan(out).any()                              # every sorted (k + 1)-tuple is produced exactly once
    T = out
  return float(np.sum(mult[n] * Nn * T[:, 0]))


if __name__ == "__main__":
  rng = np.random.default_rng(0)
  for d in (8, 5, 3, 1):
    ins, last, mult = build_tables(d)
    M = 40
    zc, zc2 = rng.standard_normal((M, d)), rng.standard_normal((M, d))
    w, w2 = rng.standard_normal(M), rng.standard_normal(M)
    G = rng.standard_normal((d, d)) * 0.3
    bij = zc @ G @ zc2.T
    for n in (5, 6):
      ts = tuples(n, d)
      mono = lambda z: np.stack([np.prod(z[:, list(t)], axis=1) for t in ts], axis=1)
      Nn, Qn = w @ mono(zc), w2 @ mono(zc2)
      got = contract(Nn, Qn, G, n, d, ins, last, mult)
      ref = float(w @ bij ** n @ w2)
      print(f"d={d} n={n}: {got:+.12e} vs {ref:+.12e}  rel {abs(got - ref) / abs(ref):.1e}")
      assert abs(got - ref) <= 1e-10 * max(abs(ref), 1.0) * 100
  print("ok")


def contract_mitm(Nn, Qn, G, n, d, ins, last):
  """meet in the middle (what k_spoly56 runs): KX = 3 indices of Q_n go to the row side with G, the other n - 3 indices of N_n to
  the column side with G^T, then one dot over sym(3) x sym(n - 3) entries with both multinomials:
      X[I][J] (I: 3 transformed, J: n - 3 raw column-side),   Y[J][I] (J: n - 3 transformed, I: 3 raw row-side)"""
  def steps(T0, Gm, ksteps):
    T = T0.copy().reshape(1, -1)
    for k in range(ksteps):
      nI, nJ = sym(k, d), sym(n - k - 1, d)
      out = np.full((sym(k + 1, d), nJ), np.nan)
      for I in range(nI):
        for J in range(nJ):
          v = T[I, ins[n - k - 1][J]]
          for i in range(last[k][I], d):
            out[I + comb(i + k, k + 1), J] = Gm[i] @ v
      T = out
    return T
  X = steps(Qn, G, 3)                      # [sym3][sym(n-3)]
  Y = steps(Nn, G.T, n - 3)                # [sym(n-3)][sym3]
  mult = lambda k: np.array([factorial(k) / np.prod([factorial(c) for c in np.bincount(t, minlength=d)]) for t in tuples(k, d)])
  return float(np.sum(mult(3)[:, None] * mult(n - 3)[None, :] * X * Y.T))


if __name__ == "__main__":
  rng = np.random.default_rng(1)
  for d in (8, 4, 2, 1):
    ins, last, mult = build_tables(d)
    M = 30
    zc, zc2 = rng.standard_normal((M, d)), rng.standard_normal((M, d))
    w, w2 = rng.standard_normal(M), rng.standard_normal(M)
    G = rng.standard_normal((d, d)) * 0.3
    bij = zc @ G @ zc2.T
    for n in (5, 6):
      ts = tuples(n, d)
      mono = lambda z: np.stack([np.prod(z[:, list(t)], axis=1) for t in ts], axis=1)
      Nn, Qn = w @ mono(zc), w2 @ mono(zc2)
      got = contract_mitm(Nn, Qn, G, n, d, ins, last)
      ref = float(w @ bij ** n @ w2)
      print(f"mitm d={d} n={n}: {got:+.12e} vs {ref:+.12e}")
      assert abs(got - ref) <= 1e-9 * max(abs(ref), 1.0)
  print("mitm ok")
