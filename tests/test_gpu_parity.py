"""GPU parity: HIP path (through the C ABI) vs the fp64 CPU oracle on identical inputs.

Tolerances (max abs error / max abs value of the tensor):
  float64: 1e-9 on f1 / cross, 1e-6 on Sff -- the oracle itself carries cond(Kuu) * eps from
           its O(M^3) triangular solves (Kuu + 1e-6 I has condition numbers ~1e7 here; the two
           algebraically identical routes differ by ~1e-8 on the worst case), measured <= 1.3e-8.
  float32: 2e-6 on f1 / cross, 2e-5 on Sff -- the state and the outputs are f32; the diagonal pairs are
           reduced in f64 and the off-diagonal pairs take their constant + linear + quadratic part
           from f64 weight moments, only the O(b^3) remainder is reduced in f32 (DESIGN.md "fp32 error
           budget"); measured <= 1.4e-6 (it was 2.6e-4 while the whole expm1 was summed in f32).
"""
import numpy as np
import pytest
import torch

from gpflowpilco_amd import ops
from gpflowpilco_amd.moment_matching import GaussianMoments, moment_matching
from gpflowpilco_amd.synthetic import make_inputs, make_svgp
from oracle import mm_oracle as mo
from tests.helpers import (contract_err, f32_state, gp_model_from_oracle, oracle_params, random_svgp_params, scale_err, to_dev)

pytestmark = pytest.mark.gpu

TOL = {torch.float64: dict(f1=1e-9, Sff=1e-6, cross=1e-9),
       torch.float32: dict(f1=2e-6, Sff=2e-5, cross=2e-6)}

CASES = [
    # name,        L, M,   d, B, scale
    ("reftest_so", 1, 16, 4, 2, 0.01),
    ("reftest_mo", 2, 16, 4, 2, 0.01),
    ("c1_like", 4, 100, 6, 1, 0.1),
    ("c2_cut", 4, 128, 5, 4, 0.1),
    ("ragged_M", 3, 203, 7, 3, 0.2),
    ("wide_sigma", 2, 64, 3, 5, 0.5),
    ("d16", 2, 130, 16, 2, 0.1),
    ("m512", 4, 512, 8, 2, 0.1),
    # edges of the dispatch tables: d = 1, the maximum d = 32 (4 blocks of 8 / 8 K-steps), d = 9 and 24
    # (partly filled blocks), fewer points than one tile, a batch that is not a multiple of anything
    ("d1", 2, 40, 1, 3, 0.1),
    ("d9", 3, 64, 9, 2, 0.1, (1.0, 3.0)),
    ("d24", 2, 70, 24, 1, 0.1, (2.0, 4.0)),
    ("d32_max", 2, 96, 32, 2, 0.1, (2.5, 5.0)),
    ("tiny_M", 2, 3, 2, 2, 0.1),
    ("b33", 2, 48, 4, 33, 0.15),
]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_moment_match_synthetic(case, dtype, device):
  _, L, M, d, B, scale = case[:6]
  syn = make_svgp(L, M, d, seed=1000 + L + M, mean_c=True, **({"ls_bounds": case[6]} if len(case) > 6 else {}))
  mu, Sigma = make_inputs(B, d, seed=7, scale=scale)
  f1o, Sffo, cro = mo.mm_gauss_svgp_mo(mu, Sigma, oracle_params(syn))
  model = syn.to_model(device)
  x = GaussianMoments((to_dev(mu, device, dtype), to_dev(Sigma, device, dtype)), centered=True)
  match = moment_matching(x, model)
  tol = TOL[dtype]
  assert match.cross[1] is True
  assert scale_err(match.y.mean(), f1o) < tol["f1"]
  assert scale_err(match.y.covariance(), Sffo) < tol["Sff"]
  assert scale_err(match.cross[0], cro) < tol["cross"]
  # Cov(x, f) = Sigma @ cross_pre (gaussian.py:38-39)
  assert scale_err(match.cross_covariance(), Sigma @ cro) < tol["cross"]


@pytest.mark.parametrize("whiten", [False, True])
@pytest.mark.parametrize("flags", [(True, True), (True, False), (False, True), (False, False)],
                         ids=["full_unc", "full_nounc", "diag_unc", "diag_nounc"])
def test_reference_test_design_svgp(whiten, flags, device):
  """tests/test_moment_matching.py:199-264 design (LCM 2->3, Constant mean), all flag combos."""
  full, unc = flags
  p = random_svgp_params(seed=5, L=2, M=16, d=4, whiten=whiten, W_rows=3)
  rng = np.random.default_rng(3)
  mu = rng.uniform(size=(2, 4))
  from gpflowpilco_amd.synthetic import generate_covariance
  Sigma = generate_covariance(rng, 4, (2,), 0.05)
  f1o, Sffo, cro = mo.mm_gauss_svgp_mo(mu, Sigma, p, full_output_cov=full, model_uncertainty=unc, jitter=1e-3)
  model = gp_model_from_oracle(p, device)
  x = GaussianMoments((to_dev(mu, device, torch.float64), to_dev(Sigma, device, torch.float64)), centered=True)
  m = moment_matching(x, model, full_output_cov=full, model_uncertainty=unc, jitter=1e-3)
  cov = m.y.covariance()
  cov = cov.diag_part() if not full else cov
  assert scale_err(m.y.mean(), f1o) < 1e-9
  assert scale_err(cov, Sffo) < 1e-8
  assert scale_err(m.cross[0], cro) < 1e-9


def test_single_output_and_gpr(device):
  """_mm_gauss_svgp_so (models.py:129-197) and _mm_gauss_gpr (:44-111)."""
  from gpflowpilco_amd import models as gp
  rng = np.random.default_rng(11)
  d, N, B = 4, 16, 2
  p = random_svgp_params(seed=9, L=1, M=N, d=d, whiten=False)
  mu = rng.uniform(size=(B, d))
  from gpflowpilco_amd.synthetic import generate_covariance
  Sigma = generate_covariance(rng, d, (B,), 0.05)
  f1o, Sffo, cro = mo.mm_gauss_svgp_so(mu, Sigma, p)
  x = GaussianMoments((to_dev(mu, device, torch.float64), to_dev(Sigma, device, torch.float64)), centered=True)
  m = moment_matching(x, gp_model_from_oracle(p, device))
  assert scale_err(m.y.mean(), f1o) < 1e-9 and scale_err(m.y.covariance(), Sffo) < 1e-8
  assert scale_err(m.cross[0], cro) < 1e-9

  X = rng.uniform(size=(N, d)); Y = 0.89 * rng.standard_normal((N, 1))
  ls = np.exp(rng.uniform(np.log(0.3), np.log(3), size=d))
  gpr_o = mo.GPRParams(X=X, Y=Y, lengthscales=ls, variance=0.89 ** 2, noise_variance=1e-3, mean_c=1.3)
  f1o, Sffo, cro = mo.mm_gauss_gpr(mu, Sigma, gpr_o)
  t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=device)
  gpr = gp.GPR(data=(t(X), t(Y)), kernel=gp.SquaredExponential(variance=t(0.89 ** 2), lengthscales=t(ls)),
               mean_function=gp.Constant(t([1.3])), noise_variance=t(1e-3))
  m = moment_matching(x, gpr)
  assert scale_err(m.y.mean(), f1o) < 1e-9 and scale_err(m.y.covariance(), Sffo) < 1e-7
  assert scale_err(m.cross[0], cro) < 1e-9
  md = moment_matching(x, gpr, full_output_cov=False)
  assert torch.allclose(md.y.covariance().diag_part(), torch.diagonal(m.y.covariance(), dim1=-2, dim2=-1), rtol=1e-12, atol=0)


def test_stage_api_q_terms(device):
  """mm_q_forward's q equals eKfu (gpflow <k(x,Z)>) and stage-wise == fused."""
  syn = make_svgp(3, 150, 5, seed=21)
  mu, Sigma = make_inputs(4, 5, seed=8, scale=0.15)
  po = oracle_params(syn)
  eKfu = mo.eKfu_list(mu, Sigma, po.Z, po.lengthscales, po.variance)          # [B,M,L]
  model = syn.to_model(device)
  pm = model.packed(torch.float64, True, device)
  flags = ops.make_flags(True, True)
  mu_t, S_t = to_dev(mu, device, torch.float64), to_dev(Sigma, device, torch.float64)
  f1, cross, q = ops.q_forward(pm, mu_t, S_t, flags, want_q=True)
  Sff = ops.Q_reduce_forward(pm, 4, flags)
  assert scale_err(q.transpose(1, 2), eKfu) < 1e-12
  f1b, Sffb, crossb = ops.moment_match(pm, mu_t, S_t)
  assert torch.equal(f1, f1b) and torch.equal(Sff, Sffb) and torch.equal(cross, crossb)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("seed", [601, 603])
def test_reference_kernel_expectation_design_on_the_gpu(seed, dtype, device):
  """Rows a-4 / a-5 on the reference's OWN kernel-expectation design (tests/test_kernel_expectation.py:51-93: d = 2, two kernels,
  32 inducing points each, lengthscales log-U[0.1, 10]; oracle/quadrature_pin.py pins the oracle's closed forms there to 1e-14):
  the q terms <k(x, Z)> of both kernels through the stage API, and the pair expectation <k2(A, x) k3(x, B)> -- which the HIP
  path never materialises -- through Sff = beta^T (<k k^T> - <k><k>^T) beta' with two random weight vectors."""
  from oracle import quadrature_pin as qp
  mx, Sxx, lsA, A, lsB, Bz, var = qp.kernel_expectation_design(seed)
  rng = np.random.default_rng(seed + 1)
  beta = rng.standard_normal((2, 32))
  eA = mo.eKfu_se(mx[None], Sxx[None], A, lsA, var)[0]
  eB = mo.eKfu_se(mx[None], Sxx[None], Bz, lsB, var)[0]
  eAB = mo.eKuffu_se_pair(mx[None], Sxx[None], lsA, var, A, lsB, var, Bz, False, False)[0]
  eAA = mo.eKuffu_se_pair(mx[None], Sxx[None], lsA, var, A, lsA, var, A, True, True)[0]
  eBB = mo.eKuffu_se_pair(mx[None], Sxx[None], lsB, var, Bz, lsB, var, Bz, True, True)[0]
  want = np.array([[beta[0] @ (eAA - np.outer(eA, eA)) @ beta[0], beta[0] @ (eAB - np.outer(eA, eB)) @ beta[1]],
                   [0.0, beta[1] @ (eBB - np.outer(eB, eB)) @ beta[1]]])
  want[1, 0] = want[0, 1]
  t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=device)
  pm = ops.pack_model(t64(np.stack([A, Bz])), t64(np.stack([lsA, lsB])), t64(np.full(2, var)), t64(beta), None, None, dtype=dtype)
  mu_t, S_t = to_dev(mx[None], device, dtype), to_dev(Sxx[None], device, dtype)
  flags = ops.make_flags(True, False)
  f1, cross, q = ops.q_forward(pm, mu_t, S_t, flags, want_q=True)
  Sff = ops.Q_reduce_forward(pm, 1, flags)
  pm.check_status(1)
  # the f32 pack sees the f32-rounded state: compare at it (lengthscales down to 0.1 make <k> sensitive to 1e-8 of x)
  if dtype == torch.float32:
    m32, S32 = mu_t.double().cpu().numpy()[0], S_t.double().cpu().numpy()[0]
    eA = mo.eKfu_se(m32[None], S32[None], A, lsA, var)[0]; eB = mo.eKfu_se(m32[None], S32[None], Bz, lsB, var)[0]
  qtol, stol = (1e-12, 1e-11) if dtype == torch.float64 else (2e-6, 2e-5)
  assert scale_err(q[0, 0], eA) < qtol and scale_err(q[0, 1], eB) < qtol
  if dtype == torch.float64:
    assert scale_err(Sff[0], want) < stol, (Sff[0].cpu().numpy(), want)
  else:
    assert np.abs(Sff[0].double().cpu().numpy() - want).max() < stol * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_euler_and_rollout(dtype, device):
  """MomentMatchingEuler.step and a 5-step closed rollout vs the oracle (solvers.py:67-135)."""
  L = d = 4
  syn = make_svgp(L, 96, d, seed=33)
  mu, Sigma = make_inputs(3, d, seed=5, scale=0.1, lo=0.3, hi=0.7)
  po = oracle_params(syn)
  muH, SH, traj = mo.rollout_closed(mu, Sigma, po, 5, dt=1.0, keep=True)
  model = syn.to_model(device)
  pm = model.packed(dtype, True, device)
  mu_t, S_t = to_dev(mu, device, dtype), to_dev(Sigma, device, dtype)
  m1, S1, tmu, tS = ops.rollout_closed(pm, mu_t, S_t, 5, dt=1.0, keep_trajectory=True)
  tol = 1e-6 if dtype == torch.float64 else 2e-5
  for h in range(5):
    assert scale_err(tmu[h], traj[h][0]) < tol
    assert scale_err(tS[h], traj[h][1]) < tol
  assert torch.equal(m1, tmu[-1]) and torch.equal(S1, tS[-1])
  # one explicit step through the stage API
  f1, Sff, cross = ops.moment_match(pm, mu_t, S_t)
  m2, S2 = ops.euler_update(mu_t, S_t, f1, Sff, cross, dt=1.0)
  assert torch.allclose(m2, tmu[0]) and torch.allclose(S2, tS[0])
  pm.check_status(3)


def test_graphed_rollout_replays_bitwise(device):
  """The HIP-graph capture of the closed rollout replays to the eager result, for new inputs too."""
  L = d = 4
  syn = make_svgp(L, 96, d, seed=34)
  pm = syn.to_model(device).packed(torch.float64, True, device)
  g = ops.GraphedRollout(pm, 3, 6, dt=1.0, keep_trajectory=True)
  for seed in (5, 6):
    mu, Sigma = make_inputs(3, d, seed=seed, scale=0.1, lo=0.3, hi=0.7)
    mu_t, S_t = to_dev(mu, device, torch.float64), to_dev(Sigma, device, torch.float64)
    ref = ops.rollout_closed(pm, mu_t, S_t, 6, dt=1.0, keep_trajectory=True)
    out = g(mu_t, S_t)
    for a, b in zip(out, ref):
      assert torch.equal(a, b)
  pm.check_status(3)
  with pytest.raises(ValueError):
    g(mu_t[:2], S_t[:2])



def test_sigma_to_zero_limit(device):
  """Sigma -> 0: f1 -> GP predictive mean, Sff -> predictive variance."""
  from oracle.pin_oracle import svgp_predict_f
  syn = make_svgp(3, 80, 4, seed=41)
  mu, _ = make_inputs(5, 4, seed=2)
  Sigma = np.broadcast_to(1e-12 * np.eye(4), (5, 4, 4)).copy()
  mean, cov = svgp_predict_f(mu, oracle_params(syn))
  model = syn.to_model(device)
  x = GaussianMoments((to_dev(mu, device, torch.float64), to_dev(Sigma, device, torch.float64)), centered=True)
  m = moment_matching(x, model)
  assert scale_err(m.y.mean(), mean) < 1e-8
  assert np.abs(m.y.covariance().cpu().numpy() - cov).max() < 1e-8


def test_non_pd_sigma_is_flagged(device):
  syn = make_svgp(2, 32, 3, seed=2)
  mu, Sigma = make_inputs(4, 3, seed=1)
  Sigma[2] = -np.eye(3)
  model = syn.to_model(device)
  pm = model.packed(torch.float64, True, device)
  ops.moment_match(pm, to_dev(mu, device, torch.float64), to_dev(Sigma, device, torch.float64))
  with pytest.raises(FloatingPointError, match="batch element 2"):
    pm.check_status(4)


def test_bad_arguments_raise(device):
  syn = make_svgp(2, 32, 3, seed=2)
  model = syn.to_model(device)
  pm_noC = model.packed(torch.float64, False, device)
  mu, Sigma = make_inputs(2, 3, seed=1)
  with pytest.raises(ValueError, match="MM_E_NO_C"):
    ops.moment_match(pm_noC, to_dev(mu, device, torch.float64), to_dev(Sigma, device, torch.float64), model_uncertainty=True)
  with pytest.raises(TypeError):
    ops.moment_match(pm_noC, to_dev(mu, device, torch.float32), to_dev(Sigma, device, torch.float32))
  with pytest.raises(RuntimeError, match="GPU only"):
    ops.moment_match(pm_noC, torch.tensor(mu), torch.tensor(Sigma))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("shape", [(3, 200, 5, 3), (2, 128, 8, 2), (4, 330, 11, 2), (2, 64, 1, 2)],
                         ids=["d5", "d8", "d11", "d1"])
def test_mfma_kernels_match_generic(shape, dtype, device):
  """The MFMA reduce kernels (f64 16x16x4 tiles, f32 32x32x2 panels) against the portable VALU
  kernel (MM_FORCE_GENERIC) and the oracle; exercises every K-step template and the padding."""
  L, M, d, B = shape
  syn = make_svgp(L, M, d, seed=77 + d)
  mu, Sigma = make_inputs(B, d, seed=3, scale=0.2)
  _, Sffo, _ = mo.mm_gauss_svgp_mo(mu, Sigma, oracle_params(syn))
  pm = syn.to_model(device).packed(dtype, True, device)
  mu_t, S_t = to_dev(mu, device, dtype), to_dev(Sigma, device, dtype)
  _, Sff_fast, _ = ops.moment_match(pm, mu_t, S_t)
  _, Sff_gen, _ = ops.moment_match(pm, mu_t, S_t, force_generic=True)
  scale = np.abs(Sffo).max()
  # the two f64 kernels sum the ill-conditioned C term (|C| up to 1/jitter) in different orders
  tol = 1e-6 if dtype == torch.float64 else 1e-5
  assert float((Sff_fast - Sff_gen).abs().max()) / scale < tol
  assert scale_err(Sff_fast, Sffo) < TOL[dtype]["Sff"]
  # diagonal-only and mean-only variants go through the same kernels
  _, Sd, _ = ops.moment_match(pm, mu_t, S_t, full_output_cov=False)
  assert torch.allclose(Sd, torch.diagonal(Sff_fast, dim1=-2, dim2=-1), rtol=1e-12, atol=0)
  _, Sn, _ = ops.moment_match(pm, mu_t, S_t, model_uncertainty=False)
  _, Sno, _ = mo.mm_gauss_svgp_mo(mu, Sigma, oracle_params(syn), model_uncertainty=False)
  assert scale_err(Sn, Sno) < TOL[dtype]["Sff"]


def test_large_delta_slow_path_f32(device):
  """Wide input covariance: |delta| > 1 takes the exp2 branch of the f32 kernel.  Asserted at the accuracy contract of the
  f32 pack (3e-4 of the off-diagonal block's scale, 2e-5 of the largest variance; oracle at the f32-rounded state): items
  whose own rounding estimate exceeds it are re-reduced in f64 (csrc/mm_route.hip)."""
  syn = make_svgp(2, 96, 3, seed=5)
  mu, Sigma = f32_state(*make_inputs(3, 3, seed=4, scale=1.5))
  _, Sffo, _ = mo.mm_gauss_svgp_mo(mu, Sigma, oracle_params(syn))
  pm = syn.to_model(device).packed(torch.float32, True, device)
  _, Sff, _ = ops.moment_match(pm, to_dev(mu, device, torch.float32), to_dev(Sigma, device, torch.float32))
  off, dia = contract_err(Sff, Sffo)
  assert off < 3e-4 and dia < 2e-5, (off, dia)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_expected_cost_kernel(dtype, device):
  """mm_expected_cost vs the closed form of components.py:26-37 evaluated in numpy."""
  rng = np.random.default_rng(0)
  d, N = 5, 37
  from gpflowpilco_amd.synthetic import generate_covariance
  mean = rng.uniform(size=(N, d)); cov = generate_covariance(rng, d, (N,), 0.3)
  target = rng.uniform(size=d); A = rng.standard_normal((d, d)); W = A @ A.T
  IpSW = np.eye(d) + cov @ W
  err = mean - target
  dist2 = np.einsum('ni,nij,nj->n', err, W @ np.linalg.inv(IpSW), err)
  want = -np.linalg.det(IpSW) ** -0.5 * np.exp(-0.5 * dist2)
  got = ops.expected_cost(to_dev(mean, device, dtype), to_dev(cov, device, dtype),
                          to_dev(target, device, dtype), to_dev(W, device, dtype))
  assert scale_err(got, want) < (1e-12 if dtype == torch.float64 else 1e-5)


def test_f64_tier_selection_sees_every_row(device):
  """The f64 reduce picks its expm1 Taylor degree from the tile's max |delta|.  A tight cluster of
  inducing points (|delta| ~ 1e-4) plus four far-away points at rows that are NOT among the first four
  of their 16-row tile (|delta| ~ 0.4 between them): a range test that samples only some accumulator
  entries picks the lowest degree and is off by 1e-9 (this happened: tools/bitcast_repro.hip); the
  portable kernel evaluates expm1 per entry.  (The f32 mode's diagonal pairs run the same code.)"""
  rng = np.random.default_rng(11)
  L, M, d, B = 1, 64, 2, 2
  dtype = torch.float64
  Z = 0.5 + 0.01 * rng.standard_normal((L, M, d))
  beta = rng.standard_normal((L, M))
  for row in (5, 22, 47, 62):
    Z[0, row] = 0.5 + np.array([1.0, -0.8]) + 0.02 * rng.standard_normal(d)
    beta[0, row] = 3.0
  ls = np.full((L, d), 0.8); var = np.ones(L)
  Cm = rng.standard_normal((L, M, M)); Cm = 0.05 * (Cm + Cm.transpose(0, 2, 1))
  t64 = lambda a: torch.tensor(a, dtype=torch.float64, device=device)
  pm = ops.pack_model(t64(Z), t64(ls), t64(var), t64(beta), t64(Cm), None, dtype=dtype)
  mu = to_dev(0.5 + 0.01 * rng.standard_normal((B, d)), device, dtype)
  Sigma = to_dev(np.broadcast_to(0.16 * np.eye(d), (B, d, d)).copy(), device, dtype)
  _, Sff_fast, _ = ops.moment_match(pm, mu, Sigma)
  _, Sff_gen, _ = ops.moment_match(pm, mu, Sigma, force_generic=True)
  rel = float((Sff_fast - Sff_gen).abs().max() / Sff_gen.abs().max())
  assert rel < 1e-12, rel


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_c_abi_from_a_compiled_host(dtype, device, tmp_path):
  """examples/abi_host.cpp: a C++ program on include/gpflowpilco_mm.h and the HIP runtime alone (no torch,
  no Python) packs a model and runs one moment match.  Its outputs must equal the Python binding's bit for
  bit (same kernels, same launch parameters) and agree with the oracle."""
  import os, shutil, struct, subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
  if not os.path.exists(hipcc):
    pytest.skip("hipcc not available")
  exe = str(tmp_path / "abi_host")
  libdir = os.path.join(root, "gpflowpilco_amd")
  subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-I" + os.path.join(root, "include"),
                  os.path.join(root, "examples", "abi_host.cpp"), "-L" + libdir, "-lgpflowpilco_mm",
                  "-Wl,-rpath," + libdir, "-o", exe], check=True, capture_output=True, timeout=300)
  L, M, d, B = 3, 150, 5, 4
  syn = make_svgp(L, M, d, seed=91)
  model = syn.to_model(device)
  Z, ls, var, beta, C, mean_c = model.precompute(device)
  mu, Sigma = make_inputs(B, d, seed=4, scale=0.2)
  code = 1 if dtype == torch.float64 else 0
  flags = ops.make_flags(True, True, False)
  mc = np.zeros(L) if mean_c is None else mean_c.cpu().numpy()
  with open(tmp_path / "in.bin", "wb") as f:
    f.write(struct.pack("6i", L, M, d, B, code, flags))
    for a in (Z.cpu().numpy(), ls.cpu().numpy(), var.cpu().numpy(), beta.cpu().numpy(), C.cpu().numpy(), mc, mu, Sigma):
      f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
  r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=120)
  assert r.returncode == 0, r.stderr
  raw = open(tmp_path / "out.bin", "rb").read()
  n1, n2, n3 = B * L, B * L * L, B * d * L
  vals = np.frombuffer(raw[: 8 * (n1 + n2 + n3)], dtype=np.float64)
  status = struct.unpack("i", raw[8 * (n1 + n2 + n3):])[0]
  assert status == 0
  f1c, Sffc, crc = vals[:n1].reshape(B, L), vals[n1:n1 + n2].reshape(B, L, L), vals[n1 + n2:].reshape(B, d, L)
  pm = ops.pack_model(Z, ls, var, beta, C, None if mean_c is None else mean_c, dtype=dtype)
  f1, Sff, cr = ops.moment_match(pm, to_dev(mu, device, dtype), to_dev(Sigma, device, dtype))
  for got, want in ((f1c, f1), (Sffc, Sff), (crc, cr)):
    assert np.array_equal(got, want.double().cpu().numpy())
  _, Sffo, _ = mo.mm_gauss_svgp_mo(mu, Sigma, oracle_params(syn))
  assert scale_err(torch.tensor(Sffc), Sffo) < TOL[dtype]["Sff"]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_deterministic_input_is_the_limit_of_small_covariance(dtype, device):
  """Sigma = 0 exactly (a point input): the centred reduce has no 0/0 -- Sff is the predictive covariance at mu
  (diagonal for independent latents), finite, and continuous with Sigma = 1e-10 I, where the oracle is defined."""
  L, M, d, B = 3, 200, 4, 3
  syn = make_svgp(L, M, d, seed=5)
  mu, _ = make_inputs(B, d, seed=1, scale=0.1)
  pm = syn.to_model(device).packed(dtype, True, device)
  S0 = np.zeros((B, d, d)); S1 = np.broadcast_to(1e-10 * np.eye(d), (B, d, d)).copy()
  f0, Sff0, cr0 = ops.moment_match(pm, to_dev(mu, device, dtype), to_dev(S0, device, dtype))
  pm.check_status(B)
  assert bool(torch.isfinite(Sff0).all() and torch.isfinite(cr0).all())
  off = Sff0 - torch.diag_embed(torch.diagonal(Sff0, dim1=-2, dim2=-1))
  assert float(off.abs().max()) < 1e-12 and float(torch.diagonal(Sff0, dim1=-2, dim2=-1).min()) > 0
  f1o, Sffo, cro = mo.mm_gauss_svgp_mo(mu, S1, oracle_params(syn))
  # (the two inputs differ by O(1e-10) themselves)
  assert scale_err(f0, f1o) < max(TOL[dtype]["f1"], 1e-7) and scale_err(Sff0, Sffo) < TOL[dtype]["Sff"]
  assert scale_err(cr0, cro) < max(TOL[dtype]["cross"], 1e-7)


def test_empty_batch_returns_empty_outputs(device):
  """B = 0: shapes follow the reference's tensor ops (empty outputs), no kernel launch, argument checks still apply."""
  syn = make_svgp(2, 40, 3, seed=2)
  model = syn.to_model(device)
  pm = model.packed(torch.float64, True, device)
  mu = torch.empty(0, 3, dtype=torch.float64, device=device); S = torch.empty(0, 3, 3, dtype=torch.float64, device=device)
  f1, Sff, cr = ops.moment_match(pm, mu, S)
  assert f1.shape == (0, 2) and Sff.shape == (0, 2, 2) and cr.shape == (0, 3, 2)
  _, Sd, _ = ops.moment_match(pm, mu, S, full_output_cov=False)
  assert Sd.shape == (0, 2)
  with pytest.raises(ValueError, match="MM_E_NO_C"):
    ops.moment_match(model.packed(torch.float64, False, device), mu, S, model_uncertainty=True)
  with pytest.raises(ValueError):
    ops.moment_match(pm, torch.empty(0, 4, dtype=torch.float64, device=device), torch.empty(0, 4, 4, dtype=torch.float64, device=device))
  # ... and so does the backward: an empty gradient, for either pack type
  z = lambda *s_: torch.empty(*s_, dtype=torch.float64, device=device)
  gmu, gS = ops.moment_match_backward(pm, mu, S, z(0, 2), z(0, 2, 2), z(0, 3, 2))
  assert gmu.shape == (0, 3) and gS.shape == (0, 3, 3)
  gmu, gS = ops.moment_match_backward(model.packed(torch.float32, True, device), mu.float(), S.float(), z(0, 2), z(0, 2, 2), z(0, 3, 2))
  assert gmu.shape == (0, 3) and gS.shape == (0, 3, 3)


def test_latents_with_different_active_dims_match_quadrature(device):
  """Per-latent ``active_dims`` (moment_matching/models.py:264-270 slices per kernel): latents acting on different,
  pairwise OVERLAPPING subsets of a 3-D input, dense input covariance (disjoint subsets under a diagonal covariance:
  test_disjoint_active_dims_under_a_diagonal_covariance below).  Checker: the tensor Gauss-Hermite
  quadrature of the definition (oracle/quadrature_pin.py) with a predict_f that slices per latent, so nothing of the
  embedding the product path uses enters the check.  The cross term comes back per latent in that latent's own
  sliced coordinates, as the reference stacks it."""
  from gpflowpilco_amd import models as gp
  from gpflowpilco_amd.moment_matching import GaussianMoments, moment_matching
  from oracle import mm_oracle as mo
  from oracle import pin_oracle as po
  from oracle import quadrature_pin as qp
  rng = np.random.default_rng(31)
  D, M = 3, 10
  acts = [(0, 1), (1, 2), (0, 2)]
  L = len(acts)
  Zs = [rng.uniform(size=(M, D)) for _ in range(L)]
  lss = [np.exp(rng.uniform(np.log(0.5), np.log(2.0), size=2)) for _ in range(L)]
  var = 0.89 ** 2 * (1 + 0.3 * rng.uniform(size=L))
  q_mu = 0.89 * rng.standard_normal((M, L))
  q_sqrt = np.linalg.cholesky(po.generate_covariance(rng, M, (L,), 0.5))
  mu = rng.uniform(0.2, 0.8, size=(2, D)); Sigma = po.generate_covariance(rng, D, (2,), 0.25)
  t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=device)
  kernels = [gp.SquaredExponential(variance=t(var[a]), lengthscales=t(lss[a]), active_dims=acts[a]) for a in range(L)]
  iv = gp.SeparateIndependentInducingVariables([gp.InducingPoints(t(Zs[a])) for a in range(L)])
  model = gp.SVGP(kernel=gp.SeparateIndependent(kernels), inducing_variable=iv, q_mu=t(q_mu), q_sqrt=t(q_sqrt),
                  whiten=True, num_latent_gps=L)
  x = GaussianMoments((t(mu), t(Sigma)), centered=True)
  m = moment_matching(x, model)
  f1, Sff, cross = (v.cpu().numpy() for v in (m.y.mean(), m.y.covariance(), m.cross[0]))
  assert cross.shape == (2, 2, L) and m.cross[1] is True

  def predict(X):                                  # the definition: every latent sees its own slice of x
    mean = np.empty((X.shape[0], L)); covd = np.empty((X.shape[0], L))
    for a in range(L):
      pa = mo.SVGPParams(Z=Zs[a][None][:, :, list(acts[a])], lengthscales=lss[a][None], variance=var[a:a + 1],
                         q_mu=q_mu[:, a:a + 1], q_sqrt=q_sqrt[a:a + 1], whiten=True)
      ma, ca = po.svgp_predict_f(X[:, list(acts[a])], pa)
      mean[:, a], covd[:, a] = ma[:, 0], ca[:, 0, 0]
    return mean, covd[:, :, None] * np.eye(L)[None]

  for b in range(2):
    qf, qS, qX = qp.quadrature_moments(predict, mu[b], Sigma[b], 40)
    assert np.abs(f1[b] - qf).max() < 1e-9 and np.abs(Sff[b] - qS).max() < 1e-9
    for a in range(L):                             # Sigma_act^-1 Cov(x_act, f_a) in latent a's coordinates
      idx = list(acts[a])
      pre = np.linalg.solve(Sigma[b][np.ix_(idx, idx)], qX[idx, a])
      assert np.abs(cross[b, :, a] - pre).max() < 1e-8


def test_offdiag_regimes_inside_collapsed_dense(device):
  """The three regimes of the f32 off-diagonal reduce (csrc/mm_mfma.hip, mm_moments.hip, mm_moments6.hip), each against the
  oracle at the accuracy contract of the f32 pack (3e-4 of the off-diagonal block's scale, 2e-5 of the largest variance):
  narrow states (the Cauchy-Schwarz bound puts every |b| <= 1/4: no tile work, all of the remainder in the moments up to
  degree 6), wider ones (collapsed, tiles screened) and -- with short lengthscales, where G = Lam^-1 T Lam'^-1 is large --
  items that are not collapsed (every tile reduced; |b| up to 4: the exp2 branch).  ops.offdiag_stats reports the regime."""
  L, M, d, B = 3, 300, 4, 4
  n = B * (L * (L - 1) // 2)
  flags = ops.make_flags(True, True, False)
  seen = []
  # (Cauchy-Schwarz bounds of these draws, tools-style numpy: 0 .. 0.22 | 0.23 .. 0.43, 0.39 .. 0.68 | 2.6 .. 4.8)
  for ls_bounds, scales in (((0.7, 2.0), (0.01, 0.12, 0.6)), ((0.4, 0.9), (0.2, 0.3)), ((0.2, 0.45), (0.3, 0.8))):
    syn = make_svgp(L, M, d, seed=4242, ls_bounds=ls_bounds)
    pm = syn.to_model(device).packed(torch.float32, True, device)
    for scale in scales:
      mu, Sigma = f32_state(*make_inputs(B, d, seed=11, scale=scale, lo=0.3, hi=0.7))
      _, Sffo, _ = mo.mm_gauss_svgp_mo(mu, Sigma, oracle_params(syn))
      _, Sff, _ = ops.moment_match(pm, to_dev(mu, device, torch.float32), to_dev(Sigma, device, torch.float32))
      collapsed, total, inside = ops.offdiag_stats(pm, B, flags)
      partly, groups, all_groups = ops.offdiag_row_groups(pm, B, flags)
      assert total == n and inside <= collapsed <= total and collapsed + partly <= total
      # (Mp = 384: six 64-row groups, the last one padding alone -- flagged whatever the item is, and read by nobody)
      assert all_groups == n * 6 and collapsed * 6 + partly <= groups <= collapsed * 6 + partly * 5 + (n - collapsed - partly)
      seen.append((collapsed, inside, partly))
      off, dia = contract_err(Sff, Sffo)
      assert off < 3e-4 and dia < 2e-5, (ls_bounds, scale, collapsed, inside, partly, off, dia)
  assert seen[0][:2] == (n, n), seen                              # narrow: every item wholly inside
  assert any(c == n and i < n for c, i, _ in seen), seen          # collapsed, tiles screened
  assert any(0 < c < n for c, i, _ in seen), seen                 # items collapsed in every row group and others in one call
  assert any(c == 0 for c, i, _ in seen), seen                    # no item collapsed in every row group
  # the collapse is decided per 64-row group (csrc/mm_mono.h): with the pack in norm order an item beyond the bound keeps the
  # groups of its small rows collapsed
  assert any(pt > 0 for _, _, pt in seen), seen


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_diagonal_operator_input_covariance(dtype, device):
  """Row a-2 (`_moments_to_gpflow_distrib`, moment_matching/models.py:302-311): an input whose covariance is a
  diagonal LINEAR OPERATOR (what the diagonal-output handlers and the NormalCDF head return) enters a GP handler;
  the reference re-expands it with tf.linalg.diag (kernel_expectation.py:100-103).  Same outputs as the dense
  diagonal matrix, and equal to the oracle."""
  from gpflowpilco_amd.moment_matching import LinearOperatorDiag
  L, M, d, B = 3, 70, 4, 3
  syn = make_svgp(L, M, d, seed=909)
  rng = np.random.default_rng(5)
  mu = rng.uniform(0.2, 0.8, size=(B, d))
  var = rng.uniform(0.01, 0.06, size=(B, d))
  Sigma = np.stack([np.diag(v) for v in var])
  f1o, Sffo, cro = mo.mm_gauss_svgp_mo(mu, Sigma, oracle_params(syn))
  model = syn.to_model(device)
  x_op = GaussianMoments((to_dev(mu, device, dtype), LinearOperatorDiag(to_dev(var, device, dtype))), centered=True)
  x_dn = GaussianMoments((to_dev(mu, device, dtype), to_dev(Sigma, device, dtype)), centered=True)
  m_op, m_dn = moment_matching(x_op, model), moment_matching(x_dn, model)
  tol = TOL[dtype]
  assert torch.equal(m_op.y.mean(), m_dn.y.mean()) and torch.equal(m_op.y.covariance(), m_dn.y.covariance())
  assert torch.equal(m_op.cross[0], m_dn.cross[0])
  assert scale_err(m_op.y.mean(), f1o) < tol["f1"] and scale_err(m_op.y.covariance(), Sffo) < tol["Sff"]
  assert scale_err(m_op.cross[0], cro) < tol["cross"]
  assert scale_err(m_op.cross_covariance(), Sigma @ cro) < tol["cross"]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_shared_independent_kernel_and_inducing_variables(dtype, device):
  """``SharedIndependent`` kernel + ``SharedIndependentInducingVariables`` (utils/kernel_expectation.py:48-69): one
  kernel OBJECT and one inducing set for all latents, so the reference takes the ``is_same_kern`` branch for a != a'
  too (V = Lambda / 2 and the squared-distance matrix term, :119,168-174).  Against the oracle with
  ``shared_kernel=True`` (which takes that branch literally)."""
  from gpflowpilco_amd import models as gp
  from oracle import pin_oracle as po
  rng = np.random.default_rng(77)
  L, M, d, B = 3, 48, 4, 3
  Z = rng.uniform(size=(M, d))
  ls = np.exp(rng.uniform(np.log(0.5), np.log(2.0), size=d))
  var = 0.89 ** 2
  q_mu = 0.5 * rng.standard_normal((M, L))
  q_sqrt = np.linalg.cholesky(po.generate_covariance(rng, M, (L,), 0.4))
  mu, Sigma = make_inputs(B, d, seed=78, scale=0.1, lo=0.3, hi=0.7)
  pr = mo.SVGPParams(Z=np.broadcast_to(Z, (L, M, d)).copy(), lengthscales=np.broadcast_to(ls, (L, d)).copy(),
                     variance=np.full(L, var), q_mu=q_mu, q_sqrt=q_sqrt, whiten=True, shared_kernel=True)
  f1o, Sffo, cro = mo.mm_gauss_svgp_mo(mu, Sigma, pr)
  t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=device)
  kern = gp.SharedIndependent(gp.SquaredExponential(variance=t(var), lengthscales=t(ls)), output_dim=L)
  iv = gp.SharedIndependentInducingVariables(gp.InducingPoints(t(Z)))
  model = gp.SVGP(kernel=kern, inducing_variable=iv, q_mu=t(q_mu), q_sqrt=t(q_sqrt), whiten=True, num_latent_gps=L)
  ks, Zs = gp.unpack_multioutput(model.kernel, model.inducing_variable, L)
  assert len(ks) == L and all(k is ks[0] for k in ks) and all(z is Zs[0] for z in Zs)
  m = moment_matching(GaussianMoments((to_dev(mu, device, dtype), to_dev(Sigma, device, dtype)), centered=True), model)
  tol = TOL[dtype]
  assert scale_err(m.y.mean(), f1o) < tol["f1"] and scale_err(m.y.covariance(), Sffo) < tol["Sff"]
  assert scale_err(m.cross[0], cro) < tol["cross"]
  md = moment_matching(GaussianMoments((to_dev(mu, device, dtype), to_dev(Sigma, device, dtype)), centered=True), model,
                       full_output_cov=False)
  assert scale_err(md.y.covariance(dense=True), Sffo * np.eye(L)[None]) < tol["Sff"]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_disjoint_active_dims_under_a_diagonal_covariance(dtype, device):
  """The reference's product shortcut (utils/kernel_expectation.py:85-89): two kernels on DISJOINT input dims under a
  ``DiagonalGaussian`` need no joint expectation, <k1(Z1,x) k2(x,Z2)> = <k1(x,Z1)> (x) <k2(x,Z2)>
  (``mo.eKuffu_se_pair_separate_dims``).  Here the latents are embedded into the union of their dims with a 1e6
  lengthscale on the inactive ones and go through the general pair kernels with a ``LinearOperatorDiag`` covariance:
  f1, the diagonal of Sff and the cross term must equal the per-latent oracle on that latent's own slice, and the
  off-diagonal entry must equal what the shortcut gives, beta_a^T (q_a q_b^T) beta_b - f1_a f1_b."""
  from gpflowpilco_amd import models as gp
  from gpflowpilco_amd.moment_matching import LinearOperatorDiag
  from oracle import pin_oracle as po
  rng = np.random.default_rng(91)
  D, M, B = 4, 24, 3
  acts = [(0, 2), (1, 3)]
  L = len(acts)
  Zs = [rng.uniform(size=(M, D)) for _ in range(L)]
  lss = [np.exp(rng.uniform(np.log(0.5), np.log(2.0), size=2)) for _ in range(L)]
  var = 0.89 ** 2 * (1 + 0.3 * rng.uniform(size=L))
  q_mu = 0.6 * rng.standard_normal((M, L))
  q_sqrt = np.linalg.cholesky(po.generate_covariance(rng, M, (L,), 0.4))
  mu = rng.uniform(0.2, 0.8, size=(B, D)); vdiag = rng.uniform(0.01, 0.06, size=(B, D))
  t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=device)
  kernels = [gp.SquaredExponential(variance=t(var[a]), lengthscales=t(lss[a]), active_dims=acts[a]) for a in range(L)]
  iv = gp.SeparateIndependentInducingVariables([gp.InducingPoints(t(Zs[a])) for a in range(L)])
  model = gp.SVGP(kernel=gp.SeparateIndependent(kernels), inducing_variable=iv, q_mu=t(q_mu), q_sqrt=t(q_sqrt),
                  whiten=True, num_latent_gps=L)
  x = GaussianMoments((to_dev(mu, device, dtype), LinearOperatorDiag(to_dev(vdiag, device, dtype))), centered=True)
  m = moment_matching(x, model)
  f1, Sff, cross = m.y.mean(), m.y.covariance(), m.cross[0]
  tol = TOL[dtype]
  per = []
  for a in range(L):
    idx = list(acts[a])
    pa = mo.SVGPParams(Z=Zs[a][None][:, :, idx], lengthscales=lss[a][None], variance=var[a:a + 1],
                       q_mu=q_mu[:, a:a + 1], q_sqrt=q_sqrt[a:a + 1], whiten=True)
    Sa = np.stack([np.diag(v[idx]) for v in vdiag])
    f1a, Sa_ff, cra = mo.mm_gauss_svgp_mo(mu[:, idx], Sa, pa)
    per.append((pa, f1a))
    assert scale_err(f1[:, a], f1a[:, 0]) < tol["f1"] and scale_err(Sff[:, a, a], Sa_ff[:, 0, 0]) < tol["Sff"]
    assert scale_err(cross[:, :, a], cra[:, :, 0]) < tol["cross"]       # in latent a's own sliced coordinates
  # off-diagonal pair through the shortcut: Q_01 = q_0 (x) q_1, so beta_0^T Q_01 beta_1 = f1_0 f1_1 and Sff_01 = 0
  Q01 = mo.eKuffu_se_pair_separate_dims(mu, vdiag, acts[0], lss[0], var[0], Zs[0][:, list(acts[0])],
                                        acts[1], lss[1], var[1], Zs[1][:, list(acts[1])])
  betas = []
  for a in range(L):
    Kuu = mo.se_kernel(Zs[a][:, list(acts[a])], None, lss[a], var[a]) + mo.DEFAULT_JITTER * np.eye(M)
    Lu = np.linalg.cholesky(Kuu)
    betas.append(np.linalg.solve(Lu.T, q_mu[:, a]))                    # whiten=True: beta = L^-T q_mu
  want01 = np.einsum('i,bij,j->b', betas[0], Q01, betas[1]) - per[0][1][:, 0] * per[1][1][:, 0]
  scale = float(np.abs(Sff.double().cpu().numpy()).max())
  got01 = Sff[:, 0, 1].double().cpu().numpy()
  assert np.abs(want01).max() < 1e-8 * max(scale, 1.0)    # the shortcut: independent outputs (beta carries cond(Kuu) eps)
  assert np.abs(got01 - want01).max() < tol["Sff"] * scale and torch.equal(Sff[:, 0, 1], Sff[:, 1, 0])


def test_pack_order_is_internal(device):
  """Packs of M > 256 points are sorted per latent by |(z - mean z) / lengthscale| (csrc/mm_kernels.hip: k_pack_key / k_pack_rank;
  include/gpflowpilco_mm.h: mm_pack_perm); smaller packs keep the caller's order.  perm is a permutation in key order, q comes
  back in the CALLER's order, and the outputs do not depend on the order the caller listed the points in."""
  L, M, d, B = 3, 300, 5, 3
  syn = make_svgp(L, M, d, seed=33, ls_bounds=(0.6, 2.0))
  po = oracle_params(syn)
  mu, Sigma = make_inputs(B, d, seed=4, scale=0.15)
  model = syn.to_model(device)
  flags = ops.make_flags(True, True)
  for dtype in (torch.float64, torch.float32):
    pm = model.packed(dtype, True, device)
    perm = pm.perm().cpu().numpy()
    Z = np.broadcast_to(np.asarray(po.Z), (L, M, d))
    for a in range(L):
      assert sorted(perm[a].tolist()) == list(range(M))
      key = ((((Z[a] - Z[a].mean(0)) / np.asarray(po.lengthscales)[a]) ** 2).sum(1))[perm[a]]
      assert np.all(np.diff(key) >= -1e-12 * key.max())
    mu_t, S_t = to_dev(mu, device, dtype), to_dev(Sigma, device, dtype)
    _, _, q = ops.q_forward(pm, mu_t, S_t, flags, want_q=True)
    eKfu = mo.eKfu_list(mu_t.double().cpu().numpy(), S_t.double().cpu().numpy(), po.Z, po.lengthscales, po.variance)   # [B,M,L]
    assert scale_err(q.double().transpose(1, 2), eKfu) < (1e-12 if dtype == torch.float64 else 2e-6)
  small = make_svgp(2, 200, 4, seed=5).to_model(device).packed(torch.float64, True, device)
  assert torch.equal(small.perm(), torch.arange(200, device=device)[None].expand(2, 200))
  # the same model with its points listed in another order
  pre = model._cache._pre
  Zt, ls, var, beta, C, mean_c = pre
  sh = torch.randperm(M, generator=torch.Generator().manual_seed(1)).to(device)
  pm1 = ops.pack_model(Zt, ls, var, beta, C, mean_c, dtype=torch.float64)
  pm2 = ops.pack_model(Zt[:, sh], ls, var, beta[:, sh], C[:, sh][:, :, sh], mean_c, dtype=torch.float64)
  mu_t, S_t = to_dev(mu, device, torch.float64), to_dev(Sigma, device, torch.float64)
  o1 = ops.moment_match(pm1, mu_t, S_t)
  o2 = ops.moment_match(pm2, mu_t, S_t)
  for x, y in zip(o1, o2):
    assert float((x - y).abs().max()) <= 1e-9 * float(x.abs().max())
