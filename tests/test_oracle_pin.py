"""Pins the CPU oracle the way the reference pins itself: its three Monte-Carlo test designs
(tests/test_kernel_expectation.py:51-93, tests/test_moment_matching.py:88-264) re-run in numpy
against oracle/mm_oracle.py, with the reference's acceptance |a-b| <= 10/sqrt(n) and the 1e-12
diag-vs-full checks.  2e5 samples here (CI time); ``python -m oracle.pin_oracle`` runs 1e6."""
import pytest

from oracle import pin_oracle as po

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
