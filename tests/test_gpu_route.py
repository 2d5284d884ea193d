"""The f32 pack's accuracy contract (csrc/mm_route.hip): the off-diagonal (b, pair) items whose f32 rounding-error estimate
exceeds MM_ROUTE_TOL of the covariance block's scale are re-reduced in f64 -- forward sum and backward aggregates -- and counted.

The reference computes every <K_Zx K_xZ'> term in float64 (gpflow_pilco/utils/kernel_expectation.py:158-165); the oracle is
``oracle/mm_oracle.py`` (its literal restatement)."""
import numpy as np
import pytest
import torch

from gpflowpilco_amd import _lib, ops
from gpflowpilco_amd.synthetic import make_inputs, make_svgp
from oracle import mm_oracle as mo
from tests.helpers import gp_model_from_oracle, oracle_params, random_svgp_params, to_dev

pytestmark = pytest.mark.gpu
F64 = torch.float64


@pytest.fixture(scope="module")
def device():
  assert torch.cuda.is_available()
  return torch.device("cuda", 0)


def _offdiag(S):
  S = np.array(S, dtype=np.float64, copy=True)
  idx = np.arange(S.shape[-1])
  S[..., idx, idx] = 0.0
  return S


def _wide_case(device, seed=3013, L=5, M=639, d=4, B=4, scale=0.25):
  """Draw 13 of tests/test_gpu_backward_f32.py: state std 0.25 at lengthscales 0.5-1.5, Kuu of 639 points in 4 dimensions
  (beta up to 1e4): the case whose f32 off-diagonal covariances were at 1e-2 of their own scale."""
  p = random_svgp_params(seed=seed, L=L, M=M, d=d, whiten=True, ls_bounds=(0.5, 1.5), mean=True)
  model = gp_model_from_oracle(p, device)
  rng = np.random.default_rng(seed + 1)
  mu = rng.uniform(0.25, 0.75, size=(B, d))
  S = make_inputs(B, d, seed=seed + 2, scale=scale)[1]
  mu32, S32 = to_dev(mu, device, torch.float32), to_dev(S, device, torch.float32)
  return p, model, mu32, S32


def test_wide_state_offdiagonal_block_at_its_own_scale(device):
  p, model, mu32, S32 = _wide_case(device)
  pm = model.packed(torch.float32, True, device)
  pm.status().zero_()
  f1, Sff, cross = ops.moment_match(pm, mu32, S32)
  routed_fwd, _ = pm.routed()
  f1o, Sffo, cro = mo.mm_gauss_svgp_mo(mu32.double().cpu().numpy(), S32.double().cpu().numpy(), p)
  own = np.abs(_offdiag(Sffo)).max()
  err = np.abs(_offdiag(Sff.double().cpu().numpy()) - _offdiag(Sffo)).max()
  assert routed_fwd > 0                                      # the estimate saw it
  assert ops.offdiag_routed(pm, mu32.shape[0], ops.make_flags(True, True)) == routed_fwd
  assert err <= 1e-4 * own, (err, own, routed_fwd)
  # what the f32 sweep alone returns there (rounds 1-3): two to three digits worse
  _, Sff_n, _ = ops.moment_match(pm, mu32, S32, extra_flags=_lib.MM_NO_ROUTE)
  err_n = np.abs(_offdiag(Sff_n.double().cpu().numpy()) - _offdiag(Sffo)).max()
  assert err_n > 10.0 * err, (err_n, err)
  # the diagonal and the first moments never depended on it
  assert np.abs(f1.double().cpu().numpy() - f1o).max() <= 2e-6 * np.abs(f1o).max()
  dg = np.abs(np.diagonal(Sff.double().cpu().numpy(), axis1=1, axis2=2) - np.diagonal(Sffo, axis1=1, axis2=2)).max()
  assert dg <= 2e-5 * np.abs(Sffo).max()


def test_wide_state_value_from_the_backward_sweeps_keeps_the_contract(device):
  """mm_moment_match_with_sums takes the off-diagonal covariances from the backward's remainder aggregates: its sweep carries the
  same error estimate and routes the same kind of items, so the wide-state draw meets the same bound there."""
  p, model, mu32, S32 = _wide_case(device)
  pm = model.packed(torch.float32, True, device)
  pm.status().zero_()
  _, Sff, _, sums, _ = ops.moment_match_with_sums(pm, mu32, S32)
  pm.check_status(mu32.shape[0])
  assert pm.routed()[1] > 0                                  # (counted as backward-sweep routes: that is the sweep that ran)
  _, Sffo, _ = mo.mm_gauss_svgp_mo(mu32.double().cpu().numpy(), S32.double().cpu().numpy(), p)
  own = np.abs(_offdiag(Sffo)).max()
  err = np.abs(_offdiag(Sff.double().cpu().numpy()) - _offdiag(Sffo)).max()
  assert err <= 1e-4 * own, (err, own)
  assert np.abs(np.diagonal(Sff.double().cpu().numpy(), axis1=1, axis2=2) - np.diagonal(Sffo, axis1=1, axis2=2)).max() <= 2e-6 * np.abs(Sffo).max()


@pytest.mark.parametrize("L,M,d,B", [(3, 150, 3, 3), (2, 300, 6, 2), (4, 200, 8, 5), (3, 260, 12, 2), (2, 130, 20, 3)])
def test_forced_route_equals_the_f64_pack(L, M, d, B, device):
  """Every off-diagonal item through k_route_f64 (all four input-dimension instantiations, M not a multiple of the panel):
  the f32 pack then returns the f64 pack's off-diagonal covariances to f32 output rounding."""
  syn = make_svgp(L, M, d, seed=500 + d, ls_bounds=(0.6, 2.5) if d <= 8 else (1.5, 4.0))
  model = syn.to_model(device)
  mu, S = make_inputs(B, d, seed=77 + d, scale=0.15, lo=0.3, hi=0.7)
  mu32, S32 = to_dev(mu, device, torch.float32), to_dev(S, device, torch.float32)
  pm32, pm64 = model.packed(torch.float32, True, device), model.packed(F64, True, device)
  pm32.status().zero_()
  _, Sff, _ = ops.moment_match(pm32, mu32, S32, extra_flags=_lib.MM_FORCE_ROUTE)
  assert pm32.routed()[0] == B * L * (L - 1) // 2
  _, Sff64, _ = ops.moment_match(pm64, mu32.double(), S32.double())
  want = Sff64.cpu().numpy()
  err = np.abs(_offdiag(Sff.double().cpu().numpy()) - _offdiag(want)).max()
  assert err <= 3e-7 * np.abs(_offdiag(want)).max() + 1e-7 * np.abs(want).max() * 1e-2, (err, np.abs(_offdiag(want)).max())


def test_nothing_is_routed_on_the_baseline_recipe_and_the_result_is_the_sweeps(device):
  """BASELINE.md's recipe at a cut-down C3 shape: the estimate stays a factor >= 5 under the threshold (no item routed), so the
  routed build returns bit for bit what the f32 sweep alone returns."""
  L, M, d, B = 6, 700, 8, 16
  syn = make_svgp(L, M, d, seed=1002, stable=False)
  model = syn.to_model(device)
  mu, S = make_inputs(B, d, seed=2002, scale=0.1)
  mu32, S32 = to_dev(mu, device, torch.float32), to_dev(S, device, torch.float32)
  pm = model.packed(torch.float32, True, device)
  pm.status().zero_()
  out = ops.moment_match(pm, mu32, S32)
  assert pm.routed() == (0, 0)
  out_n = ops.moment_match(pm, mu32, S32, extra_flags=_lib.MM_NO_ROUTE)
  for a, b in zip(out, out_n):
    assert torch.equal(a, b)
  # and it is right: off-diagonal block against the oracle at its own scale
  p = oracle_params(syn)
  _, Sffo, _ = mo.mm_gauss_svgp_mo(mu32[:2].double().cpu().numpy(), S32[:2].double().cpu().numpy(), p)
  got = out[1][:2].double().cpu().numpy()
  assert np.abs(_offdiag(got) - _offdiag(Sffo)).max() <= 2e-5 * np.abs(_offdiag(Sffo)).max()


def test_backward_routes_the_same_wide_items_and_matches_the_f64_pack(device):
  p, model, mu32, S32 = _wide_case(device)
  B, L, d = mu32.shape[0], 5, 4
  pm32, pm64 = model.packed(torch.float32, True, device), model.packed(F64, True, device)
  rng = np.random.default_rng(11)
  g1, g2, g3 = (to_dev(rng.standard_normal(s), device, F64) for s in ((B, L), (B, L, L), (B, d, L)))
  pm32.status().zero_()
  a = ops.moment_match_backward(pm32, mu32, S32, g1, g2, g3)
  assert pm32.routed()[1] > 0
  b = ops.moment_match_backward(pm64, mu32.double(), S32.double(), g1, g2, g3)
  for x, y in zip(a, b):
    sc = float(y.abs().amax())
    assert float((x - y).abs().amax()) < 2e-4 * sc, (float((x - y).abs().amax()), sc)
  # every item through the f64 aggregates: the f32 pack's gradient is then the f64 pack's
  a2 = ops.moment_match_backward(pm32, mu32, S32, g1, g2, g3, stages=_lib.MM_FORCE_ROUTE)
  for x, y in zip(a2, b):
    sc = float(y.abs().amax())
    assert float((x - y).abs().amax()) < 1e-7 * sc, (float((x - y).abs().amax()), sc)


@pytest.mark.parametrize("L,M,d,B", [(3, 150, 3, 3), (2, 300, 6, 2), (4, 200, 8, 5), (3, 333, 7, 2)])
def test_forced_route_backward_equals_the_f64_pack(L, M, d, B, device):
  syn = make_svgp(L, M, d, seed=600 + d, ls_bounds=(0.6, 2.5))
  model = syn.to_model(device)
  mu, S = make_inputs(B, d, seed=88 + d, scale=0.12, lo=0.3, hi=0.7)
  mu32, S32 = to_dev(mu, device, torch.float32), to_dev(S, device, torch.float32)
  pm32, pm64 = model.packed(torch.float32, True, device), model.packed(F64, True, device)
  rng = np.random.default_rng(12)
  g1, g2, g3 = (to_dev(rng.standard_normal(s), device, F64) for s in ((B, L), (B, L, L), (B, d, L)))
  a = ops.moment_match_backward(pm32, mu32, S32, g1, g2, g3, stages=_lib.MM_FORCE_ROUTE)
  b = ops.moment_match_backward(pm64, mu32.double(), S32.double(), g1, g2, g3)
  for x, y in zip(a, b):
    sc = float(y.abs().amax())
    assert float((x - y).abs().amax()) < 1e-7 * sc, (float((x - y).abs().amax()), sc)
