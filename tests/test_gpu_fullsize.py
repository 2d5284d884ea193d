"""Parity at BASELINE.json's full sizes.

The literal oracle needs [B,L,M,L,M] memory and O(L^2 M^3) work (C3: 2 GB and ~10 s per batch
element), so at full size the checker is the algorithm-matched fp64 restatement
``oracle/mm_fused_ref.py`` (O(M^2); shown equal to the literal oracle in
tests/test_oracle_identities.py) on two batch elements, plus size-independent properties:
batch-composition (shard) invariance, symmetry / positive definiteness, MFMA vs portable kernel.
"""
import numpy as np
import pytest
import torch

from gpflowpilco_amd import ops
from gpflowpilco_amd.synthetic import make_inputs, make_svgp
from oracle import mm_fused_ref as fr
from tests.helpers import contract_err, f32_state, oracle_params, scale_err, to_dev

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: L, M, d, dtype, seed, tolerances (f1/cross, Sff)
    # f64: Kuu + 1e-6 I has cond ~1e9 here and |C| ~ 1e6, so two fp64 routes to C (numpy vs
    # rocSOLVER Cholesky) already differ by ~1e-8 absolute in the variances (5e-6 relative)
    "C2_full": (4, 1000, 5, torch.float64, 1001, 1e-8, 5e-5),
    "C3_full": (8, 2000, 8, torch.float32, 1002, 2e-6, 5e-5),
    "C4_shape_L4": (4, 4000, 16, torch.float32, 1003, 2e-6, 5e-5),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_against_fused_restatement(name, device):
  L, M, d, dtype, seed, tol1, tol2 = CONFIGS[name]
  syn = make_svgp(L, M, d, seed=seed, device=str(device), ls_bounds=(0.7, 3.0))
  po = oracle_params(syn)
  B = 16
  mu, Sigma = make_inputs(B, d, seed=2000, scale=0.1, lo=0.3, hi=0.7)
  beta, C = fr.precompute(po)
  f1o, Sffo, cro = fr.moment_match(mu[:2], Sigma[:2], po, beta, C)
  pm = syn.to_model(device).packed(dtype, True, device)
  mu_t, S_t = to_dev(mu, device, dtype), to_dev(Sigma, device, dtype)
  f1, Sff, cross = ops.moment_match(pm, mu_t, S_t)
  pm.check_status(B)
  assert scale_err(f1[:2], f1o) < tol1 and scale_err(cross[:2], cro) < tol1
  assert scale_err(Sff[:2], Sffo) < tol2
  # the diagonal (variance) entries are reduced in f64 in both modes
  dg = torch.diagonal(Sff[:2], dim1=-2, dim2=-1).double().cpu().numpy()
  assert np.abs(dg - np.diagonal(Sffo, axis1=-2, axis2=-1)).max() / np.abs(Sffo).max() < 1e-5
  # shard invariance: a batch element's result does not depend on what else is in the batch
  f1s, Sffs, crs = ops.moment_match(pm, mu_t[5:9].contiguous(), S_t[5:9].contiguous())
  assert torch.equal(f1s, f1[5:9]) and torch.equal(Sffs, Sff[5:9]) and torch.equal(crs, cross[5:9])
  # symmetric, positive definite output covariance
  assert torch.equal(Sff, Sff.transpose(1, 2))
  assert torch.linalg.eigvalsh(Sff.double()).min() > 0
  # MFMA kernels vs the portable VALU kernel on the same operands
  _, Sffg, _ = ops.moment_match(pm, mu_t[:2].contiguous(), S_t[:2].contiguous(), force_generic=True)
  assert scale_err(Sffg, Sff[:2].double().cpu().numpy()) < tol2


def test_c2_as_stated_batch_64(device):
  """BASELINE configs[1] exactly as stated -- N = 1000, d = 5 -> D = 4, B = 64, fp64 (what `bench.py --config c2` times: the
  step on H independent draws, d != D has no closed rollout): three elements (first, middle, last of the batch) against the
  literal fp64 oracle's algebraic twin (oracle/mm_fused_ref.py), shard invariance of the last element (the tail of every
  grid), symmetry, positive definiteness, and the deterministic limit's exact zeros on the off-diagonal pairs."""
  L, M, d, B = 4, 1000, 5, 64
  syn = make_svgp(L, M, d, seed=1001, device=str(device), ls_bounds=(0.7, 3.0))
  po = oracle_params(syn)
  mu, Sigma = make_inputs(B, d, seed=2000 + 1001, scale=0.1, lo=0.3, hi=0.7)
  beta, C = fr.precompute(po)
  sel = [0, 31, 63]
  f1o, Sffo, cro = fr.moment_match(mu[sel], Sigma[sel], po, beta, C)
  pm = syn.to_model(device).packed(torch.float64, True, device)
  mu_t, S_t = to_dev(mu, device, torch.float64), to_dev(Sigma, device, torch.float64)
  f1, Sff, cross = ops.moment_match(pm, mu_t, S_t)
  pm.check_status(B)
  assert scale_err(f1[sel], f1o) < 1e-8 and scale_err(cross[sel], cro) < 1e-8
  assert scale_err(Sff[sel], Sffo) < 5e-5                  # (two fp64 routes to C: see CONFIGS above)
  off, dia = contract_err(Sff[sel], Sffo)
  assert off < 1e-9, off                                   # the off-diagonal pairs carry no C: fp64 to rounding
  f1s, Sffs, crs = ops.moment_match(pm, mu_t[63:].contiguous(), S_t[63:].contiguous())
  assert torch.equal(f1s, f1[63:]) and torch.equal(Sffs, Sff[63:]) and torch.equal(crs, cross[63:])
  assert torch.equal(Sff, Sff.transpose(1, 2))
  assert torch.linalg.eigvalsh(Sff).min() > 0


def test_c3_rollout_stays_in_regime_and_matches_f64_mode(device):
  """10 closed-rollout steps at C3 size: f32 mode tracks the f64 mode of the same kernels."""
  L = d = 8
  syn = make_svgp(L, 2000, d, seed=1002, device=str(device), ls_bounds=(0.7, 3.0))
  model = syn.to_model(device)
  mu, Sigma = make_inputs(8, d, seed=2000, scale=0.1, lo=0.3, hi=0.7)
  out = {}
  for dtype in (torch.float64, torch.float32):
    pm = model.packed(dtype, True, device)
    m, S = ops.rollout_closed(pm, to_dev(mu, device, dtype), to_dev(Sigma, device, dtype), 10)
    pm.check_status(8)
    out[dtype] = (m.double(), S.double())
  # this draw's state widens to std 0.19 (|b| up to 0.3) around step 6: the regime where the f32 mode's error is set by
  # the SYSTEMATIC error of the first-tier approximant the moment collapse uses (mm_common.h MM_TIER1_DIV): measured
  # 2.9e-6 with the tier at |b| <= 1/16 (round 2, tolerance then widened to 1e-5), 1.2e-6 at 1/20 (now), 1.5e-7 at 1/32
  assert (out[torch.float32][0] - out[torch.float64][0]).abs().max() < 2e-6
  assert (out[torch.float32][1] - out[torch.float64][1]).abs().max() < 2e-6
  std = torch.diagonal(out[torch.float64][1], dim1=-2, dim2=-1).sqrt()
  assert 0.02 < std.mean() < 0.5            # the state stays inside the data's support


def test_c4_shard_shape_L32_d16(device):
  """BASELINE configs[3] per-GPU shard: N=4000, d=16, D=32 (496 off-diagonal pairs), B=32, fp32, BASELINE.md's own
  recipe (lengthscales log-U[0.3,3], GP-prior targets, mu ~ U[0,1]^d, Sigma std 0.1).  Checker: the fused fp64
  restatement on a sub-model of 3 latents (latents are independent: the (a, a') blocks of the full result ARE the
  sub-model's result) for two batch elements; plus shard invariance, symmetry, positive definiteness."""
  L, M, d, B = 32, 4000, 16, 32
  syn = make_svgp(L, M, d, seed=1003, device=str(device), ls_bounds=(0.3, 3.0), stable=False)
  mu, Sigma = make_inputs(B, d, seed=3003, scale=0.1, lo=0.0, hi=1.0)
  pm = syn.to_model(device).packed(torch.float32, True, device)
  mu_t, S_t = to_dev(mu, device, torch.float32), to_dev(Sigma, device, torch.float32)
  f1, Sff, cross = ops.moment_match(pm, mu_t, S_t)
  pm.check_status(B)
  assert torch.isfinite(f1).all() and torch.isfinite(Sff).all() and torch.isfinite(cross).all()
  sel = [0, 13, 31]
  from oracle import mm_oracle as mo
  po = mo.SVGPParams(Z=np.broadcast_to(syn.Z, (L, M, d))[sel].copy(), lengthscales=syn.lengthscales[sel],
                     variance=syn.variance[sel], q_mu=syn.q_mu[:, sel], q_sqrt=syn.q_sqrt[sel], whiten=True)
  beta, C = fr.precompute(po)
  f1o, Sffo, cro = fr.moment_match(mu[:2], Sigma[:2], po, beta, C)
  idx = torch.tensor(sel, device=device)
  assert scale_err(f1[:2][:, idx], f1o) < 2e-6 and scale_err(cross[:2][:, :, idx], cro) < 2e-6
  assert scale_err(Sff[:2][:, idx][:, :, idx], Sffo) < 5e-5
  f1s, Sffs, crs = ops.moment_match(pm, mu_t[9:12].contiguous(), S_t[9:12].contiguous())
  assert torch.equal(f1s, f1[9:12]) and torch.equal(Sffs, Sff[9:12]) and torch.equal(crs, cross[9:12])
  assert torch.equal(Sff, Sff.transpose(1, 2))
  assert torch.linalg.eigvalsh(Sff.double()).min() > 0


def test_c3_baseline_recipe_f32_error_bound(device):
  """f32 mode on BASELINE.md's own synthetic recipe (lengthscales log-U[0.3,3], GP-prior targets, mu ~ U[0,1]^d,
  Sigma std 0.1) -- the regime with short lengthscales, larger |b| and the higher range tiers -- against the fused
  fp64 restatement, and against the same kernels in f64 mode on a larger batch."""
  L = d = 8
  M = 2000
  syn = make_svgp(L, M, d, seed=1002, device=str(device), ls_bounds=(0.3, 3.0), stable=False)
  mu, Sigma = make_inputs(32, d, seed=3002, scale=0.1, lo=0.0, hi=1.0)
  model = syn.to_model(device)
  pm32, pm64 = model.packed(torch.float32, True, device), model.packed(torch.float64, True, device)
  g32 = ops.moment_match(pm32, to_dev(mu, device, torch.float32), to_dev(Sigma, device, torch.float32))
  g64 = ops.moment_match(pm64, to_dev(mu, device, torch.float64), to_dev(Sigma, device, torch.float64))
  pm32.check_status(32); pm64.check_status(32)
  po = oracle_params(syn)
  beta, C = fr.precompute(po)
  f1o, Sffo, cro = fr.moment_match(mu[:2], Sigma[:2], po, beta, C)
  assert scale_err(g64[0][:2], f1o) < 1e-8 and scale_err(g64[2][:2], cro) < 1e-8 and scale_err(g64[1][:2], Sffo) < 5e-5
  # f32 mode vs f64 mode, all 32 elements: the stated f32 tolerance of the BASELINE recipe
  for a, b, tol in zip(g32, g64, (2e-6, 5e-5, 2e-6)):
    assert float((a.double() - b).abs().max() / b.abs().max()) < tol


def test_forced_worst_tier_gives_the_same_result(device):
  """MM_FORCE_WORST_TIER (bench.py --recipe worst) only changes which range tier evaluates a tile."""
  from gpflowpilco_amd import _lib
  L = d = 4
  syn = make_svgp(L, 300, d, seed=77, ls_bounds=(0.7, 3.0))
  mu, Sigma = make_inputs(6, d, seed=5, scale=0.1, lo=0.3, hi=0.7)
  model = syn.to_model(device)
  # f64: the two forms of expm1 differ by ~1e-16 per entry, amplified by |C| ~ 1/jitter in the variances (measured 3e-10)
  for dtype, tol in ((torch.float64, 1e-8), (torch.float32, 2e-6)):
    pm = model.packed(dtype, True, device)
    mu_t, S_t = to_dev(mu, device, dtype), to_dev(Sigma, device, dtype)
    _, Sff, _ = ops.moment_match(pm, mu_t, S_t)
    flags = ops.make_flags(True, True) | _lib.MM_FORCE_WORST_TIER
    ops.q_forward(pm, mu_t, S_t, flags)
    Sw = ops.Q_reduce_forward(pm, 6, flags)
    assert float((Sw - Sff).abs().max() / Sff.abs().max()) < tol


def test_wide_sigma_short_lengthscales_f32_is_finite_and_bounded(device):
  """|b| >> 1: input std 1.0 against lengthscales down to 0.3 -- the exp2 branch with factored weights
  what_i = w_i e^{rho'_i}.  Outputs must be finite (no overflow of the factored weights); the f64 mode within 1e-6 of the
  oracle, the f32 mode within its accuracy contract (3e-4 of the off-diagonal block's scale, 2e-5 of the largest variance,
  oracle at the f32-rounded state: what its own estimate does not cover is re-reduced in f64, csrc/mm_route.hip)."""
  from oracle import mm_oracle as mo
  syn = make_svgp(3, 160, 4, seed=21, ls_bounds=(0.3, 1.0), stable=False)
  mu, Sigma = f32_state(*make_inputs(5, 4, seed=8, scale=1.0, lo=0.0, hi=1.0))
  f1o, Sffo, cro = mo.mm_gauss_svgp_mo(mu, Sigma, oracle_params(syn))
  model = syn.to_model(device)
  for dtype in (torch.float64, torch.float32):
    pm = model.packed(dtype, True, device)
    f1, Sff, cross = ops.moment_match(pm, to_dev(mu, device, dtype), to_dev(Sigma, device, dtype))
    pm.check_status(5)
    assert torch.isfinite(f1).all() and torch.isfinite(Sff).all() and torch.isfinite(cross).all()
    if dtype == torch.float64:
      assert scale_err(Sff, Sffo) < 1e-6 and scale_err(f1, f1o) < 1e-6
    else:
      off, dia = contract_err(Sff, Sffo)
      assert off < 3e-4 and dia < 2e-5 and scale_err(f1, f1o) < 2e-6, (off, dia, scale_err(f1, f1o))


@pytest.mark.parametrize("recipe", ["baseline", "pilco"])
def test_c3_at_the_configs_own_batch_256(recipe, device):
  """BASELINE configs[2] at ITS batch, B = 256 (what bench.py times), on both data recipes: two elements -- the first
  and the last of the batch -- against the fused fp64 restatement, shard invariance of a slice from the middle and
  of the last element (the tail of every grid), symmetry, positive definiteness; one closed Euler step on top."""
  L = d = 8
  M, B = 2000, 256
  kw = dict(ls_bounds=(0.3, 3.0), stable=False) if recipe == "baseline" else dict(ls_bounds=(0.7, 3.0), stable=True)
  syn = make_svgp(L, M, d, seed=1002, device=str(device), **kw)
  lo, hi = (0.0, 1.0) if recipe == "baseline" else (0.3, 0.7)
  mu, Sigma = make_inputs(B, d, seed=3002, scale=0.1, lo=lo, hi=hi)
  pm = syn.to_model(device).packed(torch.float32, True, device)
  mu_t, S_t = to_dev(mu, device, torch.float32), to_dev(Sigma, device, torch.float32)
  f1, Sff, cross = ops.moment_match(pm, mu_t, S_t)
  pm.check_status(B)
  po = oracle_params(syn)
  beta, C = fr.precompute(po)
  sel = [0, B - 1]
  f1o, Sffo, cro = fr.moment_match(mu[sel], Sigma[sel], po, beta, C)
  assert scale_err(f1[sel], f1o) < 2e-6 and scale_err(cross[sel], cro) < 2e-6
  assert scale_err(Sff[sel], Sffo) < 5e-5
  for a, b in ((100, 104), (B - 1, B)):
    f1s, Sffs, crs = ops.moment_match(pm, mu_t[a:b].contiguous(), S_t[a:b].contiguous())
    assert torch.equal(f1s, f1[a:b]) and torch.equal(Sffs, Sff[a:b]) and torch.equal(crs, cross[a:b])
  assert torch.equal(Sff, Sff.transpose(1, 2))
  assert torch.linalg.eigvalsh(Sff.double()).min() > 0
  m1, S1 = ops.euler_update(mu_t, S_t, f1, Sff, cross, 1.0)
  m2, S2 = ops.rollout_closed(pm, mu_t, S_t, 1)
  assert torch.equal(m1, m2) and torch.equal(S1, S2)
  Sxf = Sigma[sel] @ cro
  assert scale_err(S1[sel], Sigma[sel] + Sxf + np.swapaxes(Sxf, 1, 2) + Sffo) < 5e-5


@pytest.mark.parametrize("recipe", ["baseline", "pilco"])
def test_c3_backward_on_the_f32_pack_matches_the_f64_pack(recipe, device):
  """Row f-1 at C3's own sizes (N = 2000, d = D = 8, f32 model): the vector-Jacobian product of one match taken on the f32
  pack (csrc/mm_bwd_f32.hip: diagonal pairs f64, off-diagonal pairs as moment + bf16-MFMA aggregates) against the f64 pack
  of the same model on the same f32-rounded state, plus shard invariance of the gradient (size-independent property)."""
  L, M, d, B = 8, 2000, 8, 6
  ls, lo, hi = ((0.3, 3.0), 0.0, 1.0) if recipe == "baseline" else ((0.7, 3.0), 0.3, 0.7)
  syn = make_svgp(L, M, d, seed=1002, device=str(device), ls_bounds=ls, stable=recipe == "pilco")
  model = syn.to_model(device)
  pm32, pm64 = model.packed(torch.float32, True, device), model.packed(torch.float64, True, device)
  mu, S = make_inputs(B, d, seed=3002, scale=0.1, lo=lo, hi=hi)
  mu32, S32 = to_dev(mu, device, torch.float32), to_dev(S, device, torch.float32)
  g = torch.Generator(device="cpu").manual_seed(5)
  g1 = torch.randn(B, L, generator=g, dtype=torch.float64).to(device)
  g2 = torch.randn(B, L, L, generator=g, dtype=torch.float64).to(device)
  g3 = torch.randn(B, d, L, generator=g, dtype=torch.float64).to(device)
  gm32, gS32 = ops.moment_match_backward(pm32, mu32, S32, g1, g2, g3)
  gm64, gS64 = ops.moment_match_backward(pm64, mu32.double(), S32.double(), g1, g2, g3)
  pm32.check_status(B)
  for a_, b_ in ((gm32, gm64), (gS32, gS64)):
    sc = float(b_.abs().amax())
    assert float((a_ - b_).abs().amax()) < 1e-4 * sc, (recipe, float((a_ - b_).abs().amax()), sc)
  # the gradient of element b does not depend on what else is in the batch
  gm_a, gS_a = ops.moment_match_backward(pm32, mu32[2:5].contiguous(), S32[2:5].contiguous(), g1[2:5], g2[2:5], g3[2:5])
  assert float((gm_a - gm32[2:5]).abs().amax()) <= 1e-12 * float(gm32.abs().amax())
  assert float((gS_a - gS32[2:5]).abs().amax()) <= 1e-12 * float(gS32.abs().amax())


def test_c3_backward_on_the_f32_pack_matches_cpu_autograd_of_the_materialised_evaluation(device):
  """Row f-1 at C3's own sizes against the ORACLE side, not against another HIP path: one batch element of the BASELINE recipe,
  float64 torch autograd on the CPU through the materialised [P, M, M] evaluation (autodiff.moment_match_torch: the
  transliteration of models.py:200-299 -- what the reference's GradientTape differentiates, utils/optimizers.py:51-56), against
  the f32 pack's gradient on the same f32-rounded state, for both routes: the two-pass backward (forward's sweeps, then
  mm_moment_match_backward's) and value-and-gradient from one pair of sweeps (mm_moment_match_with_sums + the chain rule).
  bench.py --config c3_grad measures 8e-7 of the gradient's scale on this element; asserted at 1e-5."""
  from gpflowpilco_amd import autodiff
  L, M, d = 8, 2000, 8
  syn = make_svgp(L, M, d, seed=1002, device=str(device), ls_bounds=(0.3, 3.0), stable=False)
  model = syn.to_model(device)
  pm32 = model.packed(torch.float32, True, device)
  mu, S = make_inputs(1, d, seed=3002, scale=0.1, lo=0.0, hi=1.0)
  mu32, S32 = to_dev(mu, device, torch.float32), to_dev(S, device, torch.float32)
  g = torch.Generator(device="cpu").manual_seed(5)
  g1 = torch.randn(1, L, generator=g, dtype=torch.float64)
  g2 = torch.randn(1, L, L, generator=g, dtype=torch.float64)
  g3 = torch.randn(1, d, L, generator=g, dtype=torch.float64)
  # CPU float64 autograd at the f32-rounded state
  Z, ls, var, beta, Cm, mc = (None if t is None else t.detach().cpu() for t in model.precompute(device))
  mu_c = mu32.double().cpu().requires_grad_(True)
  S_c = S32.double().cpu().requires_grad_(True)
  f1c, Sffc, crc = autodiff.moment_match_torch(mu_c, S_c, Z, ls, var, beta, Cm, mc, True, True)
  ((g1 * f1c).sum() + (g2 * Sffc).sum() + (g3 * crc).sum()).backward()
  gmu_c, gS_c = mu_c.grad, 0.5 * (S_c.grad + S_c.grad.transpose(1, 2))
  rel = lambda x, y: float((x.cpu() - y).abs().amax() / y.abs().amax())
  gd = [t.to(device) for t in (g1, g2, g3)]
  # (i) two passes
  f1, Sff, cross = ops.moment_match(pm32, mu32, S32)
  gm, gS = ops.moment_match_backward(pm32, mu32, S32, *gd)
  pm32.check_status(1)
  assert rel(gm, gmu_c) < 1e-5 and rel(gS, gS_c) < 1e-5, (rel(gm, gmu_c), rel(gS, gS_c))
  # the value of that element against the same evaluation (f32 outputs)
  assert scale_err(Sff, Sffc.detach().numpy()) < 2e-5 and scale_err(f1, f1c.detach().numpy()) < 2e-6
  # (ii) value and gradient from one pair of sweeps
  f1s, Sffs, crs, sums, gen = ops.moment_match_with_sums(pm32, mu32, S32)
  gm2, gS2 = ops.moment_match_backward(pm32, mu32, S32, *gd, forward_generation=gen, sums=sums)
  pm32.check_status(1)
  assert rel(gm2, gmu_c) < 1e-5 and rel(gS2, gS_c) < 1e-5, (rel(gm2, gmu_c), rel(gS2, gS_c))
  assert scale_err(Sffs, Sffc.detach().numpy()) < 2e-5
