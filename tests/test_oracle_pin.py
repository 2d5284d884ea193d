"""Pins the CPU oracle the way the reference pins itself: its three Monte-Carlo test designs
(tests/test_kernel_expectation.py:51-93, tests/test_moment_matching.py:88-264) re-run in numpy
against oracle/mm_oracle.py, with the reference's acceptance |a-b| <= 10/sqrt(n) and the 1e-12
diag-vs-full checks.  2e5 samples here (CI time); ``python -m oracle.pin_oracle`` runs 1e6.

Monte Carlo pins the formulas, not the digits (1e-2 absolute).  The second half pins the digits: tensor
Gauss-Hermite quadrature (oracle/quadrature_pin.py) of the DEFINITION the reference's estimator samples
(tests/test_moment_matching.py:57-84) -- mean, full covariance incl. off-diagonal pairs of latents with
different lengthscales, cross-covariance -- to <= 1e-9 for exactly the cases no reference test touches
(SURVEY.md section 8c): whiten=True, SeparateIndependent, model_uncertainty=False, the Euler moment update."""
import pytest

from oracle import pin_oracle as po
from oracle import quadrature_pin as qp

N = int(2e5)


@pytest.mark.parametrize("seed", [101, 102])
def test_kernel_expectation_design(seed):
  errs, tol = po.check_kernel_expectation(seed, N)
  for name, v in errs.items():
    assert v <= tol, (name, v, tol)
  # same-kernel shortcut vs general branch of kernel_expectation.py:167-185 is an exact identity
  assert errs["branch_identity"] * 1e3 < 1e-12


@pytest.mark.parametrize("seed", [201, 202])
def test_gpr_design(seed):
  errs, exact, tol = po.check_gpr(seed, N)
  assert all(v <= tol for v in errs.values()), errs
  assert all(v <= 1e-12 for v in exact.values()), exact


@pytest.mark.parametrize("whiten", [False, True])
@pytest.mark.parametrize("multi_output", [False, True])
def test_svgp_design(multi_output, whiten):
  errs, exact, tol = po.check_svgp(301 + int(multi_output), N, multi_output=multi_output, whiten=whiten)
  assert all(v <= tol for v in errs.values()), errs
  assert all(v <= 1e-12 for v in exact.values()), exact


def test_larger_input_covariance_still_matches_mc():
  """The reference designs use input std 0.01 (a weak test of the Sigma dependence); std 0.3 here."""
  errs, exact, tol = po.check_svgp(401, N, multi_output=True, whiten=True, scale_x=0.3)
  assert all(v <= tol for v in errs.values()), errs


# ---- digits: Gauss-Hermite quadrature of the full handler outputs (d = 2: 60^2 nodes, d = 3: 40^3 nodes) ------
GH = {2: 60, 3: 40}


@pytest.mark.parametrize("d", [2, 3])
@pytest.mark.parametrize("kw", [dict(whiten=True), dict(whiten=True, model_uncertainty=False), dict(whiten=False),
                                dict(whiten=False, lcm_outputs=4), dict(whiten=True, single_output=True)],
                         ids=["sep_whiten", "sep_whiten_no_model_unc", "sep", "lcm", "single_output_whiten"])
def test_svgp_handlers_match_quadrature(d, kw):
  errs, scale = qp.check_svgp(7 + d, d, 3, GH[d], **kw)
  assert all(v <= 1e-9 for v in errs.values()), (errs, scale)
  if not kw.get("single_output"):
    assert scale["cov"] > 1e-2                                 # the pinned off-diagonal entries are not vacuous


@pytest.mark.parametrize("d", [2, 3])
def test_gpr_handler_matches_quadrature(d):
  errs = qp.check_gpr(17 + d, d, GH[d])
  assert all(v <= 1e-9 for v in errs.values()), errs


@pytest.mark.parametrize("d", [2, 3])
@pytest.mark.parametrize("model_uncertainty", [True, False])
def test_euler_moment_update_matches_quadrature(d, model_uncertainty):
  """MomentMatchingEuler.step (dynamics/solvers.py:110-135): moments of x + dt f(x), dt = 0.7."""
  errs = qp.check_euler(27 + d, d, GH[d], model_uncertainty=model_uncertainty)
  assert all(v <= 1e-9 for v in errs.values()), errs


# ---- digits, on the reference's OWN test designs: d = 4, M = 16, B = 2, input std 0.01, lengthscales log-U[0.01, 10] ------------
@pytest.mark.parametrize("kind,seed", qp.REFERENCE_DESIGNS, ids=[f"{k}-{s}" for k, s in qp.REFERENCE_DESIGNS])
def test_reference_designs_match_quadrature_at_d4(kind, seed):
  """tests/test_moment_matching.py:88-136 (GPR), :140-194 (single-output SVGP, whiten=False), :199-264 (LinearCoregionalization
  2 -> 3, whiten=False, Constant mean) at the reference's own sizes, where it accepts 1e-2 from 1e6 Monte-Carlo samples: the
  24^4-node Gauss-Hermite rule of the same definition pins mean, FULL covariance and cross-covariance of the oracle to 1e-8
  absolute (measured <= 1.1e-10; 1e-6 relative to each quantity's own scale)."""
  errs, scale = qp.check_reference_design(kind, seed, 24)
  assert all(v <= 1e-8 for v in errs.values()), (errs, scale)
  for k in errs:
    assert errs[k] <= 1e-6 * scale[k] + 1e-14, (k, errs[k], scale[k])


# ---- digits, on the reference's own KERNEL-EXPECTATION design: d = 2, 32 + 32 inducing points, input std 0.1, lengthscales log-U[0.1, 10] ----
@pytest.mark.parametrize("seed", qp.KERNEL_EXPECTATION_SEEDS)
def test_reference_kernel_expectation_design_matches_quadrature(seed):
  """tests/test_kernel_expectation.py:51-93 (rows a-4, a-5: <k(x, Z)> and <k2(A, x) k3(x, B)>, same-kernel and general branch of
  utils/kernel_expectation.py:167-185): where the reference compares with 1e6 Monte-Carlo samples at 1e-2, the 160^2-node
  Gauss-Hermite rule of the same expectations agrees with the oracle's closed forms to 1e-12 (measured 1e-15 ... 2e-14)."""
  errs, scale = qp.check_kernel_expectation_design(seed)
  assert all(v <= 1e-12 for v in errs.values()), errs
  assert scale['eKfu'] > 0.1 and scale['eKuffu'] > 0.1                 # the pinned entries are not vacuous
