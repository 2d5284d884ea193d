"""Backward of the moment match w.r.t. (mu, Sigma) (row f-1): HIP M^2 sums + torch surrogate vs
central finite differences of the fp64 CPU oracle on a random linear functional of the outputs."""
import numpy as np
import pytest
import torch

from gpflowpilco_amd import _lib, ops
from gpflowpilco_amd.autodiff import moment_match_differentiable
from gpflowpilco_amd.synthetic import generate_covariance, make_inputs
from oracle import mm_oracle as mo
from tests.helpers import gp_model_from_oracle, random_svgp_params, to_dev

pytestmark = pytest.mark.gpu


def _loss_np(mu, S, p, A1, A2, A3, unc):
  f1, Sff, cr = mo.mm_gauss_svgp_mo(mu, S, p, True, unc, 0.0)
  return (A1 * f1).sum() + (A2 * Sff).sum() + (A3 * cr).sum()


@pytest.mark.parametrize("unc", [True, False], ids=["unc", "nounc"])
@pytest.mark.parametrize("shape", [(2, 12, 3), (3, 20, 4), (1, 16, 2)], ids=["L2", "L3", "L1"])
def test_gradients_match_finite_differences(shape, unc, device):
  L, M, d = shape
  B = 2
  p = random_svgp_params(seed=10 + L, L=L, M=M, d=d, whiten=True, ls_bounds=(0.5, 2.0), mean=False)
  rng = np.random.default_rng(1)
  mu = rng.uniform(size=(B, d)); S = generate_covariance(rng, d, (B,), 0.3)
  A1 = rng.standard_normal((B, L)); A2 = rng.standard_normal((B, L, L)); A3 = rng.standard_normal((B, d, L))
  model = gp_model_from_oracle(p, device)
  mu_t = to_dev(mu, device, torch.float64).requires_grad_(True)
  S_t = to_dev(S, device, torch.float64).requires_grad_(True)
  f1, Sff, cr = moment_match_differentiable(model, mu_t, S_t, True, unc)
  loss = (to_dev(A1, device, torch.float64) * f1).sum() + (to_dev(A2, device, torch.float64) * Sff).sum() \
      + (to_dev(A3, device, torch.float64) * cr).sum()
  assert abs(float(loss) - _loss_np(mu, S, p, A1, A2, A3, unc)) < 1e-8 * max(1.0, abs(float(loss)))
  loss.backward()
  gmu, gS = mu_t.grad.cpu().numpy(), S_t.grad.cpu().numpy()
  eps = 1e-4   # the oracle carries ~1e-10 of cond(Kuu) noise: larger steps keep noise/eps small
  for b in range(B):
    for k in range(d):
      mp, mm = mu.copy(), mu.copy(); mp[b, k] += eps; mm[b, k] -= eps
      fd = (_loss_np(mp, S, p, A1, A2, A3, unc) - _loss_np(mm, S, p, A1, A2, A3, unc)) / (2 * eps)
      assert abs(fd - gmu[b, k]) < 3e-5 * max(1.0, abs(fd)), (b, k, fd, gmu[b, k])
    for i in range(d):
      for j in range(i, d):
        D = np.zeros((d, d)); D[i, j] = D[j, i] = 1.0
        Sp, Sm = S.copy(), S.copy(); Sp[b] += eps * D; Sm[b] -= eps * D
        fd = (_loss_np(mu, Sp, p, A1, A2, A3, unc) - _loss_np(mu, Sm, p, A1, A2, A3, unc)) / (2 * eps)
        an = gS[b, i, j] + gS[b, j, i] if i != j else gS[b, i, i]
        assert abs(fd - an) < 3e-5 * max(1.0, abs(fd)), (b, i, j, fd, an)


@pytest.mark.parametrize("shape", [(3, 300, 5, 2), (2, 130, 16, 1), (4, 70, 3, 3), (1, 200, 8, 2)],
                         ids=["L3d5", "L2d16", "L4d3", "L1d8"])
@pytest.mark.parametrize("unc", [True, False], ids=["unc", "nounc"])
def test_mfma_backward_sums_match_portable_kernel(shape, unc, device):
  """k_bwd_mfma (column-owner MFMA sweep) against k_bwd_sums (plain VALU) on the same workspace."""
  from gpflowpilco_amd import _lib, ops
  from gpflowpilco_amd.synthetic import make_inputs, make_svgp
  L, M, d, B = shape
  syn = make_svgp(L, M, d, seed=40 + L, device=str(device), ls_bounds=(0.7, 3.0))
  pm = syn.to_model(device).packed(torch.float64, True, device)
  mu, S = make_inputs(B, d, seed=7, scale=0.2, lo=0.3, hi=0.7)
  mu_t, S_t = to_dev(mu, device, torch.float64), to_dev(S, device, torch.float64)
  flags = ops.make_flags(True, unc)
  ops.q_forward(pm, mu_t, S_t, flags)
  ws = pm.workspace(B, flags)
  n = _lib.lib().mm_backward_bytes(B, L, M, d, flags)
  outs = []
  for fl in (flags, flags | _lib.MM_FORCE_GENERIC):
    out = torch.zeros(n // 8, dtype=torch.float64, device=device)
    rc = _lib.lib().mm_backward_sums(pm.buf.data_ptr(), pm.nbytes, L, M, d, _lib.MM_F64, B, mu_t.data_ptr(), fl,
                                     ws.data_ptr(), ws.numel(), out.data_ptr(), n, ops._stream(device))
    _lib.check(rc, "mm_backward_sums")
    outs.append(out)
  Mp = (M + _lib.MM_M_ALIGN - 1) // _lib.MM_M_ALIGN * _lib.MM_M_ALIGN
  P = L * (L + 1) // 2
  ncol = B * P * (3 + d) * Mp
  for lo, hi, shp in ((0, ncol, (B, P, 3 + d, Mp)), (ncol, n // 8, (B, P - L, 2, Mp))):
    if hi > lo:
      a_, b_ = outs[0][lo:hi].view(shp)[..., :M], outs[1][lo:hi].view(shp)[..., :M]
      scale = float(b_.abs().amax())
      assert float((a_ - b_).abs().amax()) < 1e-10 * max(scale, 1.0), (float((a_ - b_).abs().amax()), scale)


def test_mfma_backward_tier_selection_sees_every_row(device):
  """Same construction as test_gpu_parity.py::test_f64_tier_selection_sees_every_row for k_bwd_mfma:
  far-away inducing points at rows 5, 22, 47, 62 (large |delta| among them) inside a tight cluster."""
  from gpflowpilco_amd import _lib, ops
  rng = np.random.default_rng(12)
  L, M, d, B = 2, 64, 2, 2
  Z = 0.5 + 0.01 * rng.standard_normal((L, M, d))
  beta = rng.standard_normal((L, M))
  for row in (5, 22, 47, 62):
    Z[:, row] = 0.5 + np.array([1.0, -0.8]) + 0.02 * rng.standard_normal((L, d))
    beta[:, row] = 3.0
  Cm = rng.standard_normal((L, M, M)); Cm = 0.05 * (Cm + Cm.transpose(0, 2, 1))
  t64 = lambda a: torch.tensor(a, dtype=torch.float64, device=device)
  pm = ops.pack_model(t64(Z), t64(np.full((L, d), 0.8)), t64(np.ones(L)), t64(beta), t64(Cm), None, dtype=torch.float64)
  mu_t = to_dev(0.5 + 0.01 * rng.standard_normal((B, d)), device, torch.float64)
  S_t = to_dev(np.broadcast_to(0.16 * np.eye(d), (B, d, d)).copy(), device, torch.float64)
  flags = ops.make_flags(True, True)
  ops.q_forward(pm, mu_t, S_t, flags)
  ws = pm.workspace(B, flags)
  n = _lib.lib().mm_backward_bytes(B, L, M, d, flags)
  outs = []
  for fl in (flags, flags | _lib.MM_FORCE_GENERIC):
    out = torch.zeros(n // 8, dtype=torch.float64, device=device)
    rc = _lib.lib().mm_backward_sums(pm.buf.data_ptr(), pm.nbytes, L, M, d, _lib.MM_F64, B, mu_t.data_ptr(), fl,
                                     ws.data_ptr(), ws.numel(), out.data_ptr(), n, ops._stream(device))
    _lib.check(rc, "mm_backward_sums")
    outs.append(out)
  scale = float(outs[1].abs().amax())
  assert float((outs[0] - outs[1]).abs().amax()) < 1e-12 * scale, (float((outs[0] - outs[1]).abs().amax()), scale)


@pytest.mark.parametrize("full", [True, False], ids=["full", "diagcov"])
@pytest.mark.parametrize("unc", [True, False], ids=["unc", "nounc"])
def test_moment_form_backward_equals_reference_surrogate(full, unc, device):
  """The moment form of the surrogate (autograd over d x d algebra only) against the first version
  (autograd over [B,P,M,d] tensors), same HIP sums."""
  from gpflowpilco_amd import autodiff as ad
  from gpflowpilco_amd.synthetic import make_inputs, make_svgp
  L, M, d, B = 3, 90, 4, 3
  syn = make_svgp(L, M, d, seed=50, device=str(device), ls_bounds=(0.7, 3.0), mean_c=True)
  model = syn.to_model(device)
  pm = model.packed(torch.float64, True, device)
  pre = model._cache._pre
  mu, S = make_inputs(B, d, seed=8, scale=0.2, lo=0.3, hi=0.7)
  mu_t, S_t = to_dev(mu, device, torch.float64), to_dev(S, device, torch.float64)
  g = torch.Generator(device="cpu").manual_seed(1)
  g1 = torch.randn(B, L, generator=g, dtype=torch.float64).to(device)
  g2 = (torch.randn(B, L, L, generator=g, dtype=torch.float64) if full
        else torch.randn(B, L, generator=g, dtype=torch.float64)).to(device)
  g3 = torch.randn(B, d, L, generator=g, dtype=torch.float64).to(device)
  new = ad.moment_match_backward(pm, pre, mu_t, S_t, full, unc, g1, g2, g3)
  ref = ad.moment_match_backward_reference(pm, pre, mu_t, S_t, full, unc, g1, g2, g3)
  for a_, b_ in zip(new, ref):
    assert float((a_ - b_).abs().max()) < 1e-9 * max(1.0, float(b_.abs().max()))


def test_f32_model_gradients_track_f64(device):
  """An f32 model's backward (d <= 8: taken on the f32 pack itself, csrc/mm_bwd_f32.hip) agrees with the f64 model's."""
  L, M, d, B = 3, 40, 4, 3
  p = random_svgp_params(seed=21, L=L, M=M, d=d, whiten=True, ls_bounds=(0.5, 2.0), mean=False)
  rng = np.random.default_rng(2)
  mu = rng.uniform(size=(B, d)); S = generate_covariance(rng, d, (B,), 0.3)
  A2 = to_dev(rng.standard_normal((B, L, L)), device, torch.float64)
  model = gp_model_from_oracle(p, device)
  grads = {}
  for dtype in (torch.float64, torch.float32):
    mu_t = to_dev(mu, device, dtype).requires_grad_(True)
    S_t = to_dev(S, device, dtype).requires_grad_(True)
    f1, Sff, cr = moment_match_differentiable(model, mu_t, S_t, True, True)
    assert f1.dtype == dtype and Sff.dtype == dtype
    (f1.sum() + (A2.to(dtype) * Sff).sum() + cr.sum()).backward()
    assert mu_t.grad.dtype == dtype and S_t.grad.dtype == dtype
    grads[dtype] = (mu_t.grad.double(), S_t.grad.double())
  for g64, g32 in zip(grads[torch.float64], grads[torch.float32]):
    assert (g64 - g32).abs().max() < 1e-4 * max(1.0, float(g64.abs().max()))


def test_rollout_gradient_through_two_steps(device):
  """d loss / d (mu0, Sigma0) through two Euler steps of the differentiable match (torch glue)."""
  L = d = 3
  p = random_svgp_params(seed=3, L=L, M=18, d=d, whiten=True, ls_bounds=(0.6, 2.0), mean=False)
  p.q_mu = 0.2 * p.q_mu
  rng = np.random.default_rng(5)
  mu = rng.uniform(0.3, 0.7, size=(1, d)); S = generate_covariance(rng, d, (1,), 0.1)
  model = gp_model_from_oracle(p, device)

  def rollout_np(mu0, S0):
    m, Sg = mo.rollout_closed(mu0, S0, p, 2)
    return m.sum() + np.trace(Sg[0])

  mu_t = to_dev(mu, device, torch.float64).requires_grad_(True)
  S_t = to_dev(S, device, torch.float64).requires_grad_(True)
  m, Sg = mu_t, S_t
  for _ in range(2):
    f1, Sff, cr = moment_match_differentiable(model, m, Sg)
    Sxf = Sg @ cr
    m, Sg = m + f1, Sg + Sxf + Sxf.transpose(1, 2) + Sff
  loss = m.sum() + torch.diagonal(Sg[0]).sum()
  assert abs(float(loss) - rollout_np(mu, S)) < 1e-9
  loss.backward()
  eps = 1e-4   # the oracle carries ~1e-10 of cond(Kuu) noise: larger steps keep noise/eps small
  for k in range(d):
    mp, mm = mu.copy(), mu.copy(); mp[0, k] += eps; mm[0, k] -= eps
    fd = (rollout_np(mp, S) - rollout_np(mm, S)) / (2 * eps)
    assert abs(fd - mu_t.grad[0, k].item()) < 3e-5 * max(1.0, abs(fd))


def test_torch_path_equals_hip_path_and_policy_gradient(device):
  """(i) the parameter-differentiable torch evaluation equals the HIP kernels; (ii) d(policy loss)/d(q_mu)
  of the composed encoder -> policy -> drift rollout (the quantity update_policy needs,
  examples/cartpole_swingup/train_utils.py:91-105) matches finite differences of the oracle rollout."""
  from gpflowpilco_amd import bijectors as tfb, dynamics, models as gp, ops
  from gpflowpilco_amd.autodiff import moment_match_torch
  from gpflowpilco_amd.components import GaussianObjective, TrigonometricEncoder
  from gpflowpilco_amd.loops import get_state_initializer, policy_loss_closure
  from oracle import mm_compose_oracle as co
  from tests.test_compose import _cartpole_like
  drift_o, pol_o, mu, S, target, precis = _cartpole_like()
  F64 = torch.float64
  # (i)
  drift = gp_model_from_oracle(drift_o, device)
  jm = np.concatenate([mu, np.zeros((2, 2))], -1); jS = np.stack([np.eye(6) * 0.02] * 2)
  pm = drift.packed(F64, True, device)
  a = ops.moment_match(pm, to_dev(jm, device, F64), to_dev(jS, device, F64))
  Z, ls, var, beta, C, mc = drift.precompute(device)
  b = moment_match_torch(to_dev(jm, device, F64), to_dev(jS, device, F64), Z, ls, var, beta, C, mc)
  for x, y in zip(a, b):
    assert (x - y).abs().max() < 1e-9 * max(1.0, float(y.abs().max()))
  # (ii)
  scale, shift, active, H = 2.0, -0.5, (1,), 3
  def loss_np(qmu):
    pol = mo.SVGPParams(Z=pol_o.Z, lengthscales=pol_o.lengthscales, variance=pol_o.variance, q_mu=qmu,
                        q_sqrt=pol_o.q_sqrt, whiten=True)
    return co.policy_rollout_loss(mu, S, drift_o, lambda s: co.mm_policy(s, pol, scale, shift), active,
                                  target, precis, H).sum()
  pol_model = gp_model_from_oracle(pol_o, device)
  pol_model.q_mu = pol_model.q_mu.clone().requires_grad_(True)
  policy = gp.InverseLinkWrapper(gp.KernelRegressor(pol_model),
                                 invlink=tfb.Chain([tfb.Scale(scale), tfb.Shift(shift), tfb.NormalCDF()]))
  encoder = TrigonometricEncoder(active_dims=active)
  objective = GaussianObjective(target=to_dev(target, device, F64), precis=to_dev(precis, device, F64))
  system = dynamics.DynamicalSystem(drift=drift, policy=policy, encoder=encoder, solver=dynamics.MomentMatchingEuler())
  closure = policy_loss_closure(system, objective, get_state_initializer(to_dev(mu, device, F64), to_dev(S, device, F64)), H)
  loss = closure().sum()
  assert abs(float(loss.detach()) - loss_np(pol_o.q_mu)) < 1e-7
  loss.backward()
  g = pol_model.q_mu.grad.cpu().numpy()
  eps = 1e-5
  for idx in (0, 7, 19):
    qp, qm = pol_o.q_mu.copy(), pol_o.q_mu.copy(); qp[idx, 0] += eps; qm[idx, 0] -= eps
    fd = (loss_np(qp) - loss_np(qm)) / (2 * eps)
    assert abs(fd - g[idx, 0]) < 1e-5 * max(1.0, abs(fd)), (idx, fd, g[idx, 0])


def test_graphed_policy_loss_replays_eager_values_and_gradients(device):
  """HIP-graph capture of the composed policy loss (forward, and forward+backward) against eager."""
  from gpflowpilco_amd import bijectors as tfb, dynamics, models as gp
  from gpflowpilco_amd.components import GaussianObjective, TrigonometricEncoder
  from gpflowpilco_amd.loops import GraphedPolicyLoss, get_state_initializer, policy_loss_closure
  from tests.test_compose import _cartpole_like
  drift_o, pol_o, mu, S, target, precis = _cartpole_like()
  F64 = torch.float64
  drift = gp_model_from_oracle(drift_o, device)
  pol_model = gp_model_from_oracle(pol_o, device)
  pol_model.q_mu = pol_model.q_mu.clone().requires_grad_(True)
  policy = gp.InverseLinkWrapper(gp.KernelRegressor(pol_model),
                                 invlink=tfb.Chain([tfb.Scale(2.0), tfb.Shift(-0.5), tfb.NormalCDF()]))
  system = dynamics.DynamicalSystem(drift=drift, policy=policy, encoder=TrigonometricEncoder(active_dims=(1,)),
                                    solver=dynamics.MomentMatchingEuler())
  objective = GaussianObjective(target=to_dev(target, device, F64), precis=to_dev(precis, device, F64))
  m0, S0 = to_dev(mu, device, F64), to_dev(S, device, F64)
  closure = policy_loss_closure(system, objective, get_state_initializer(m0, S0), 3)

  def eager():
    pol_model.q_mu.grad = None
    loss = closure()
    loss.sum().backward()
    return loss.detach().clone(), pol_model.q_mu.grad.clone()

  graphed = GraphedPolicyLoss(closure, [pol_model.q_mu])
  for trial in range(2):
    le, ge = eager()
    lg, (gg,) = graphed.loss_and_grad()
    assert torch.allclose(lg, le, rtol=1e-12, atol=1e-14) and torch.allclose(gg, ge, rtol=1e-10, atol=1e-13)
    assert torch.allclose(graphed.loss(), le, rtol=1e-12, atol=1e-14)
    assert pol_model.q_mu.grad is gg
    with torch.no_grad():                      # an in-place parameter update and a new initial state
      pol_model.q_mu.mul_(0.9)
      m0.add_(0.01)


@pytest.mark.parametrize("case", [(3, 90, 4, 3, True, True), (2, 130, 6, 2, True, False), (3, 40, 3, 4, False, True),
                                  (1, 200, 8, 2, True, True), (4, 100, 6, 1, True, True), (2, 70, 16, 2, True, True),
                                  (2, 50, 31, 1, True, True), (3, 600, 5, 2, True, True), (2, 520, 12, 1, True, False)],
                         ids=["L3d4", "L2d6nounc", "L3diagcov", "L1d8", "c1drift", "d16", "d31", "M600chunks", "M520d12chunks"])
def test_native_match_backward_equals_the_torch_chain_rule(case, device):
  """mm_moment_match_backward (M x M sweeps + k_gp_bwd_items + k_gp_bwd_sum: everything on the device) against
  autodiff.moment_match_backward (the same sums, chain rule in torch) -- which the tests above pin to finite differences
  of the oracle."""
  from gpflowpilco_amd import autodiff as ad, ops
  from gpflowpilco_amd.synthetic import make_inputs, make_svgp
  L, M, d, B, full, unc = case
  syn = make_svgp(L, M, d, seed=60 + L + d, device=str(device), ls_bounds=(0.7, 3.0) if d <= 16 else (2.5, 5.0), mean_c=True)
  model = syn.to_model(device)
  pm = model.packed(torch.float64, True, device)
  pre = model._cache._pre
  mu, S = make_inputs(B, d, seed=9, scale=0.2, lo=0.3, hi=0.7)
  mu_t, S_t = to_dev(mu, device, torch.float64), to_dev(S, device, torch.float64)
  g = torch.Generator(device="cpu").manual_seed(2)
  g1 = torch.randn(B, L, generator=g, dtype=torch.float64).to(device)
  g2 = (torch.randn(B, L, L, generator=g, dtype=torch.float64) if full else torch.randn(B, L, generator=g, dtype=torch.float64)).to(device)
  g3 = torch.randn(B, d, L, generator=g, dtype=torch.float64).to(device)
  want = ad.moment_match_backward(pm, pre, mu_t, S_t, full, unc, g1, g2, g3)
  got = ops.moment_match_backward(pm, mu_t, S_t, g1, g2, g3, full, unc)
  pm.check_status(B)
  for a_, b_ in zip(got, want):
    assert float((a_ - b_).abs().max()) < 1e-10 * max(1.0, float(b_.abs().max()))


def test_native_policy_gradient_at_H30_all_parameters_and_initial_state(device):
  """Rows f-1 x f-2: the gradient of the cartpole-shaped policy loss (drift M = 100, policy M = 30, H = 30) w.r.t. EVERY
  policy parameter (q_mu, inducing inputs, lengthscales, variance) and the initial state through the native reverse sweep
  (mm_rollout_composed_backward) against (i) autograd through the torch composition of the same kernels and (ii) central
  finite differences of the CPU oracle rollout on sampled coordinates."""
  from gpflowpilco_amd import bijectors as tfb, dynamics, models as gp
  from gpflowpilco_amd.components import GaussianObjective, TrigonometricEncoder
  from gpflowpilco_amd.loops import get_state_initializer, policy_loss_closure
  from gpflowpilco_amd.synthetic import make_svgp
  from oracle import mm_compose_oracle as co
  from tests.helpers import oracle_params
  F64 = torch.float64
  drift_syn = make_svgp(4, 100, 6, seed=10, ls_bounds=(0.8, 3.0))
  drift_o = oracle_params(drift_syn)
  drift_o.Z = drift_o.Z * np.array([1, 1, 1, 1, 1, 4.0]) - np.array([0, 0, 0, 0, 0, 2.0])
  pol_o = random_svgp_params(seed=11, L=1, M=30, d=5, whiten=True, ls_bounds=(0.7, 2.0), mean=False)
  pol_o.q_mu = 0.3 * pol_o.q_mu
  rng = np.random.default_rng(12)
  mu = np.array([[0.4, 0.2, 0.5, 0.3], [0.6, -0.1, 0.4, 0.5]])
  S = generate_covariance(rng, 4, (2,), 0.05)
  target = np.array([0.0, 1.0, 0, 0, 0])
  precis = 16 * np.array([[0.25, 0, -0.5, 0, 0], [0, 0.25, 0, 0, 0], [-0.5, 0, 1, 0, 0], [0] * 5, [0] * 5], dtype=float)
  scale, shift, active, H = 2.0, -0.5, (1,), 30
  drift = gp_model_from_oracle(drift_o, device)
  pol_model = gp_model_from_oracle(pol_o, device)
  kern = pol_model.kernel.kernels[0]
  params = {"q_mu": pol_model.q_mu, "Z": pol_model.inducing_variable.inducing_variables[0].Z,
            "lengthscales": kern.lengthscales, "variance": kern.variance}
  for t in params.values():
    t.requires_grad_(True)
  policy = gp.InverseLinkWrapper(gp.KernelRegressor(pol_model),
                                 invlink=tfb.Chain([tfb.Scale(scale), tfb.Shift(shift), tfb.NormalCDF()]))
  system = dynamics.DynamicalSystem(drift=drift, policy=policy, encoder=TrigonometricEncoder(active_dims=active),
                                    solver=dynamics.MomentMatchingEuler())
  objective = GaussianObjective(target=to_dev(target, device, F64), precis=to_dev(precis, device, F64))
  m0 = to_dev(mu, device, F64).requires_grad_(True); S0 = to_dev(S, device, F64).requires_grad_(True)
  init = get_state_initializer(m0, S0)
  wts = to_dev(np.array([1.0, 0.6]), device, F64)

  def grads(native):
    for t in list(params.values()) + [m0, S0]:
      t.grad = None
    loss = policy_loss_closure(system, objective, init, H, native=native)()
    (loss * wts).sum().backward()
    out = {k: t.grad.detach().clone() for k, t in params.items()}
    out["m0"], out["S0"] = m0.grad.detach().clone(), 0.5 * (S0.grad + S0.grad.transpose(1, 2)).detach()
    return loss.detach().cpu().numpy(), out
  loss_n, gn = grads(None)              # native: taped rollout + reverse sweep
  loss_t, gt = grads(False)             # torch composition
  drift.packed(F64, True, device).check_status(2)
  assert np.abs(loss_n - loss_t).max() < 1e-9
  for k in gn:
    err = float((gn[k] - gt[k]).abs().max()) / max(1e-12, float(gt[k].abs().max()))
    assert err < 1e-7, (k, err)

  def loss_np(q_mu=None, Z=None, ls=None, var=None, mu0=None):
    pol = mo.SVGPParams(Z=pol_o.Z if Z is None else Z, lengthscales=pol_o.lengthscales if ls is None else ls,
                        variance=pol_o.variance if var is None else var, q_mu=pol_o.q_mu if q_mu is None else q_mu,
                        q_sqrt=pol_o.q_sqrt, whiten=True)
    l = co.policy_rollout_loss(mu if mu0 is None else mu0, S, drift_o, lambda s: co.mm_policy(s, pol, scale, shift), active,
                               target, precis, H)
    return float((l * np.array([1.0, 0.6])).sum())
  assert abs(loss_np() - float((loss_n * np.array([1.0, 0.6])).sum())) < 1e-7
  eps = 1e-5
  def fd(**kw_pm):
    (name, (arr, idx)), = kw_pm.items()
    ap, am = arr.copy(), arr.copy(); ap[idx] += eps; am[idx] -= eps
    return (loss_np(**{name: ap}) - loss_np(**{name: am})) / (2 * eps)
  checks = [("q_mu", fd(q_mu=(pol_o.q_mu, (3, 0))), gn["q_mu"][3, 0]), ("q_mu", fd(q_mu=(pol_o.q_mu, (21, 0))), gn["q_mu"][21, 0]),
            ("Z", fd(Z=(pol_o.Z, (0, 5, 2))), gn["Z"][5, 2]), ("Z", fd(Z=(pol_o.Z, (0, 17, 0))), gn["Z"][17, 0]),
            ("ls", fd(ls=(pol_o.lengthscales, (0, 1))), gn["lengthscales"][1]),
            ("var", fd(var=(pol_o.variance, (0,))), gn["variance"].reshape(-1)[0]),
            ("m0", fd(mu0=(mu, (0, 1))), gn["m0"][0, 1]), ("m0", fd(mu0=(mu, (1, 3))), gn["m0"][1, 3])]
  for name, want, got in checks:
    assert abs(want - float(got)) < 2e-5 * max(1.0, abs(want)), (name, want, float(got))

  # the data-parallel form of the same evaluation (distributed.distributed_loss_and_grad; one process here, two ranks with
  # gloo in tests/test_distributed.py): mean loss over the batch and its gradient through the native differentiable op
  from gpflowpilco_amd import distributed as D
  from gpflowpilco_amd.loops import native_policy_loss
  f = native_policy_loss(system, objective, H)
  plist = list(params.values())
  loss_d, g_d = D.distributed_loss_and_grad(f.with_grad, plist, m0.detach(), S0.detach())
  for t in plist:
    t.grad = None
  lm = policy_loss_closure(system, objective, get_state_initializer(m0.detach(), S0.detach()), H, native=None)().mean()
  lm.backward()
  assert abs(float(loss_d) - float(lm)) < 1e-12 * max(1.0, abs(float(lm)))
  for t, g in zip(plist, g_d):
    assert float((t.grad - g).abs().max()) <= 1e-12 * max(1e-12, float(t.grad.abs().max()))


def test_policy_update_with_native_gradients_lowers_the_loss(device):
  """The real caller (examples/cartpole_swingup/train_utils.py:91-105): Adam on the rollout loss.  Forward and backward are
  replayed from one HIP graph per step (native taped rollout + reverse sweep inside); 40 steps must lower the loss, and the
  graphed trajectory must equal the eager one."""
  import importlib.util, os
  spec = importlib.util.spec_from_file_location("policy_update", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                               "examples", "policy_update.py"))
  pu = importlib.util.module_from_spec(spec); spec.loader.exec_module(pu)
  from gpflowpilco_amd.loops import GraphedPolicyLoss
  runs = {}
  for mode in ("graph", "eager"):
    closure, params, drift = pu.build(device, H=30, B=2, seed=1000)
    opt = torch.optim.Adam(params, lr=1e-2)
    graphed = GraphedPolicyLoss(closure, params) if mode == "graph" else None
    losses = []
    for _ in range(40):
      if graphed is None:
        opt.zero_grad(set_to_none=True)
        loss = closure().sum(); loss.backward()
      else:
        loss = graphed.loss_and_grad()[0].sum()
      torch.nn.utils.clip_grad_norm_(params, 1.0)
      opt.step()
      losses.append(float(loss))
    if graphed is not None:
      graphed.check()
    drift.packed(torch.float64, True, device).check_status(2)
    runs[mode] = np.array(losses)
  assert runs["graph"][-1] < runs["graph"][0] - 1e-3, runs["graph"][[0, -1]]
  assert np.abs(runs["graph"] - runs["eager"]).max() < 1e-9 * max(1.0, np.abs(runs["eager"]).max())


def _cartpole_like(device, M_pol, seed=21):
  """(system, objective, params, m0, S0) of the cartpole wiring with a trainable policy of M_pol centres."""
  from gpflowpilco_amd import bijectors as tfb, dynamics, models as gp
  from gpflowpilco_amd.components import GaussianObjective, TrigonometricEncoder
  from gpflowpilco_amd.synthetic import make_svgp
  from tests.helpers import oracle_params
  F64 = torch.float64
  drift_o = oracle_params(make_svgp(4, 60, 6, seed=seed, ls_bounds=(0.8, 3.0)))
  drift_o.Z = drift_o.Z * np.array([1, 1, 1, 1, 1, 4.0]) - np.array([0, 0, 0, 0, 0, 2.0])
  pol_o = random_svgp_params(seed=seed + 1, L=1, M=M_pol, d=5, whiten=True, ls_bounds=(0.9, 2.0), mean=False)
  pol_o.q_mu = 0.05 * pol_o.q_mu
  drift, pol_model = gp_model_from_oracle(drift_o, device), gp_model_from_oracle(pol_o, device)
  kern = pol_model.kernel.kernels[0]
  params = {"q_mu": pol_model.q_mu, "Z": pol_model.inducing_variable.inducing_variables[0].Z, "lengthscales": kern.lengthscales}
  for t in params.values():
    t.requires_grad_(True)
  scale_t = torch.tensor(2.0, dtype=F64, device=device)
  policy = gp.InverseLinkWrapper(gp.KernelRegressor(pol_model),
                                 invlink=tfb.Chain([tfb.Scale(scale_t), tfb.Shift(-0.5), tfb.NormalCDF()]))
  system = dynamics.DynamicalSystem(drift=drift, policy=policy, encoder=TrigonometricEncoder(active_dims=(1,)),
                                    solver=dynamics.MomentMatchingEuler())
  target = to_dev(np.array([0.0, 1.0, 0, 0, 0]), device, F64)
  objective = GaussianObjective(target=target, precis=4.0 * torch.eye(5, dtype=F64, device=device))
  rng = np.random.default_rng(seed + 2)
  m0 = to_dev(np.array([[0.4, 0.2, 0.5, 0.3], [0.6, -0.1, 0.4, 0.5]]), device, F64)
  S0 = to_dev(generate_covariance(rng, 4, (2,), 0.05), device, F64)
  return system, objective, params, m0, S0, scale_t


def _closure_grads(system, objective, params, m0, S0, H, native, extra=()):
  from gpflowpilco_amd.loops import get_state_initializer, policy_loss_closure
  leaves = list(params.values()) + list(extra)
  for t in leaves:
    t.grad = None
  loss = policy_loss_closure(system, objective, get_state_initializer(m0, S0), H, native=native)()
  loss.sum().backward()
  return loss.detach(), [t.grad.detach().clone() for t in leaves]


def test_policy_with_200_centres_takes_the_native_reverse_sweep(device):
  """The native gradient's limits that cost nothing are lifted (policy M up to 256; loops.policy_loss_closure): a policy with
  M = 200 differentiates through mm_rollout_composed_backward -- no fallback warning -- and agrees with the torch composition."""
  import warnings
  system, objective, params, m0, S0, _ = _cartpole_like(device, 200)
  with warnings.catch_warnings():
    warnings.simplefilter("error", RuntimeWarning)           # a fallback would raise here
    loss_n, gn = _closure_grads(system, objective, params, m0, S0, 4, None)
  loss_t, gt = _closure_grads(system, objective, params, m0, S0, 4, False)
  assert float((loss_n - loss_t).abs().max()) < 1e-9
  for a_, b_ in zip(gn, gt):
    assert float((a_ - b_).abs().max()) < 1e-7 * max(1e-12, float(b_.abs().max()))


def test_float32_state_takes_the_native_reverse_sweep(device):
  """A float32 initial state: cast up into the float64 tape, the loss cast back (no fallback)."""
  import warnings
  system, objective, params, m0, S0, _ = _cartpole_like(device, 30)
  loss64, g64 = _closure_grads(system, objective, params, m0, S0, 5, None)
  with warnings.catch_warnings():
    warnings.simplefilter("error", RuntimeWarning)
    loss32, g32 = _closure_grads(system, objective, params, m0.float(), S0.float(), 5, None)
  assert loss32.dtype == torch.float32
  assert float((loss32.double() - loss64).abs().max()) < 1e-5
  for a_, b_ in zip(g32, g64):
    assert float((a_ - b_).abs().max()) < 1e-4 * max(1e-12, float(b_.abs().max()))


def test_a_gradient_outside_the_native_sweep_warns_and_is_carried_by_the_torch_path(device):
  """ADVICE (round 3): the native op returns gradients for the policy SVGP and the initial state only; if the head's scale or
  the objective's target requires a gradient the closure must take the torch composition -- saying so once -- and carry it."""
  system, objective, params, m0, S0, scale_t = _cartpole_like(device, 30)
  objective.target.requires_grad_(True)
  with pytest.warns(RuntimeWarning, match="objective.target requires a gradient"):
    loss_a, ga = _closure_grads(system, objective, params, m0, S0, 3, None, extra=(objective.target,))
  assert ga[-1] is not None and float(ga[-1].abs().max()) > 0.0
  loss_t, gt = _closure_grads(system, objective, params, m0, S0, 3, False, extra=(objective.target,))
  for a_, b_ in zip(ga, gt):
    assert float((a_ - b_).abs().max()) < 1e-12 + 1e-10 * float(b_.abs().max())
  objective.target.requires_grad_(False)
  scale_t.requires_grad_(True)
  with pytest.warns(RuntimeWarning, match="Scale.scale requires a gradient"):
    _, gs = _closure_grads(system, objective, params, m0, S0, 3, None, extra=(scale_t,))
  assert gs[-1] is not None and float(gs[-1].abs()) > 0.0


def test_a_stale_workspace_under_MM_WORKSPACE_CURRENT_is_reported(device):
  """ADVICE (round 3): the reuse of the forward's q stage is decided by a host-side counter; the device now checks the promise
  (the q stage stamps the mean it read) and flags a workspace that belongs to another state instead of differentiating it."""
  from gpflowpilco_amd.synthetic import make_svgp
  model = make_svgp(3, 96, 4, seed=41).to_model(device)
  pm = model.packed(torch.float64, True, device)
  mu, S = make_inputs(3, 4, seed=42, scale=0.1, lo=0.3, hi=0.7)
  mu, S = to_dev(mu, device, torch.float64), to_dev(S, device, torch.float64)
  g1, g2, g3 = (torch.ones(s, dtype=torch.float64, device=device) for s in ((3, 3), (3, 3, 3), (3, 4, 3)))
  fl = ops.make_flags(True, True)
  ops.moment_match(pm, mu, S)
  gen = pm.workspace_generation(3, fl)
  good = ops.moment_match_backward(pm, mu, S, g1, g2, g3, forward_generation=gen)
  pm.check_status(3)
  # someone overwrites the workspace behind the counter's back (another state's q stage through the raw C ABI)
  ws = pm.workspace(3, fl, peek=True)
  f1 = torch.empty(3, 3, dtype=torch.float64, device=device); cr = torch.empty(3, 4, 3, dtype=torch.float64, device=device)
  mu2 = (mu + 0.05).contiguous()
  rc = _lib.lib().mm_q_forward(pm.buf.data_ptr(), pm.nbytes, 3, 96, 4, _lib.MM_F64, 3, mu2.data_ptr(), S.data_ptr(), fl,
                               f1.data_ptr(), cr.data_ptr(), None, ws.data_ptr(), ws.numel(), pm.status().data_ptr(),
                               ops._stream(device))
  assert rc == 0
  ops.moment_match_backward(pm, mu, S, g1, g2, g3, forward_generation=gen)
  with pytest.raises(RuntimeError, match="belongs to another state"):
    pm.check_status(3)
  again = ops.moment_match_backward(pm, mu, S, g1, g2, g3)          # without the promise: the q stage is re-run
  pm.check_status(3)
  for a_, b_ in zip(again, good):
    assert torch.equal(a_, b_)
