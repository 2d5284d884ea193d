#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ from the fp64 CPU oracle.

The reference (TensorFlow/GPflow) cannot run here and holds no golden vectors of its own
(its tests are Monte-Carlo), so these are produced by ``oracle/mm_oracle.py`` -- which is pinned
to the reference's own test designs by ``tests/test_oracle_pin.py`` -- with seeded numpy inputs
(SURVEY.md section 8c "Golden vectors the build must create itself").

The fixtures are only (re)written after the oracle passes its digit-level pin: the tensor Gauss-Hermite
quadrature of oracle/quadrature_pin.py (<= 1e-9 on mean, full covariance and cross-covariance for whiten=True,
SeparateIndependent, model_uncertainty=False, GPR and the Euler update).

  python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from gpflowpilco_amd.synthetic import generate_covariance, make_inputs, make_svgp  # noqa: E402
from oracle import mm_oracle as mo  # noqa: E402
from oracle import quadrature_pin as qp  # noqa: E402
from tests.helpers import oracle_params, random_svgp_params  # noqa: E402

if qp.main() != 0:
  raise SystemExit("oracle/quadrature_pin.py failed: the golden fixtures are NOT regenerated from an unpinned oracle")


def svgp_fixture(name, p: mo.SVGPParams, mu, Sigma, rollout_steps=0):
  out = dict(Z=p.Z, lengthscales=p.lengthscales, variance=p.variance, q_mu=p.q_mu, q_sqrt=p.q_sqrt,
             whiten=np.array(p.whiten), mu=mu, Sigma=Sigma)
  if p.mean_c is not None:
    out["mean_c"] = np.asarray(p.mean_c)
  if p.W is not None:
    out["W"] = p.W
  out["eKfu"] = mo.eKfu_list(mu, Sigma, p.Z, p.lengthscales, p.variance)
  for unc in (True, False):
    f1, Sff, cross = mo.mm_gauss_svgp_mo(mu, Sigma, p, model_uncertainty=unc)
    tag = "unc" if unc else "nounc"
    out[f"f1_{tag}"], out[f"Sff_{tag}"], out[f"cross_{tag}"] = f1, Sff, cross
  f1d, Sffd, _ = mo.mm_gauss_svgp_mo(mu, Sigma, p, full_output_cov=False)
  out["Sff_diag"] = Sffd
  if rollout_steps and p.W is None and p.Z.shape[0] == p.Z.shape[2]:
    Sxf = mo.cross_covariance(Sigma, out["cross_unc"], True)
    out["mu_next"], out["Sigma_next"] = mo.euler_moment_update(mu, Sigma, out["f1_unc"], out["Sff_unc"], Sxf)
    muH, SH, traj = mo.rollout_closed(mu, Sigma, p, rollout_steps, keep=True)
    out["traj_mu"] = np.stack([t[0] for t in traj]); out["traj_Sigma"] = np.stack([t[1] for t in traj])
  np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
  print("wrote", name, {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def main():
  rng = np.random.default_rng(20240)
  # reference test design: d=4, 16 points, 2 distributions (tests/test_moment_matching.py:25-54)
  for name, L, W_rows, whiten in (("reftest_so", 1, None, False), ("reftest_mo_lcm", 2, 3, False),
                                  ("reftest_mo_sep_whiten", 3, None, True)):
    p = random_svgp_params(seed=int(rng.integers(1 << 30)), L=L, M=16, d=4, whiten=whiten,
                           ls_bounds=(0.1, 10.0), W_rows=W_rows)
    mu = rng.uniform(size=(2, 4)); Sigma = generate_covariance(rng, 4, (2,), 0.05)
    svgp_fixture(name, p, mu, Sigma)
  # the reference's OWN designs at its own sizes (d = 4, M = 16, B = 2, input std 0.01, lengthscales log-U[0.01, 10]; the draws of
  # oracle/quadrature_pin.py:reference_design, pinned to the 24^4-node quadrature by qp.main() above)
  for kind in ("svgp_so", "svgp_mo_lcm"):
    model, mx, Sxx, _, _ = qp.reference_design(kind, 501)
    svgp_fixture("refdesign_" + kind, model, mx, Sxx)
  gm, mx, Sxx, _, _ = qp.reference_design("gpr", 501)
  out = dict(X=gm.X, Y=gm.Y, lengthscales=gm.lengthscales, variance=np.array(gm.variance), noise_variance=np.array(gm.noise_variance),
             mean_c=np.array(gm.mean_c), mu=mx, Sigma=Sxx)
  for unc in (True, False):
    f1, Sff, cross = mo.mm_gauss_gpr(mx, Sxx, gm, model_uncertainty=unc)
    tag = "unc" if unc else "nounc"
    out[f"f1_{tag}"], out[f"Sff_{tag}"], out[f"cross_{tag}"] = f1, Sff, cross
  np.savez_compressed(os.path.join(HERE, "gpr_refdesign.npz"), **out)
  print("wrote gpr_refdesign")
  # kernel-expectation design: d=2, 32 inducing points, two different kernels
  d = 2
  mu = rng.standard_normal((1, d)); Sigma = generate_covariance(rng, d, (1,), 0.1)
  lsA, lsB = np.exp(rng.uniform(np.log(0.1), np.log(10), (2, d)))
  A, Bz = rng.uniform(size=(32, d)) + mu, rng.uniform(size=(32, d)) + mu
  np.savez_compressed(os.path.join(HERE, "kernel_expectation_d2.npz"), mu=mu, Sigma=Sigma, lsA=lsA, lsB=lsB,
                      A=A, B=Bz, var=np.array(0.89 ** 2),
                      eKfu_A=mo.eKfu_se(mu, Sigma, A, lsA, 0.89 ** 2),
                      eKuffu_AB=mo.eKuffu_se_pair(mu, Sigma, lsA, 0.89 ** 2, A, lsB, 0.89 ** 2, Bz, False, False),
                      eKuffu_AA=mo.eKuffu_se_pair(mu, Sigma, lsA, 0.89 ** 2, A, lsA, 0.89 ** 2, A, True, True))
  print("wrote kernel_expectation_d2")
  # C1-shaped (state dim 6 = input dim, 100 points, B=1) and a cut-down C2, with 3-step rollouts
  for name, L, M, dd, B, seed in (("c1_shaped", 6, 100, 6, 1, 1000), ("c2_cut", 5, 128, 5, 4, 1001)):
    syn = make_svgp(L, M, dd, seed=seed, ls_bounds=(0.7, 3.0))
    mu, Sigma = make_inputs(B, dd, seed=seed + 7, scale=0.1, lo=0.3, hi=0.7)
    svgp_fixture(name, oracle_params(syn), mu, Sigma, rollout_steps=3)


if __name__ == "__main__":
  main()
