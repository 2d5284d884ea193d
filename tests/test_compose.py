"""Rollout composition: encoder, policy head, forward_sde, cost.

* The numpy oracle (oracle/mm_compose_oracle.py) is pinned with the reference's own designs
  (tests/test_components.py:39-104: expected cost and trig encoder vs Monte Carlo) plus MC
  checks of the NormalCDF head, which no reference test covers ("parity unpinned").
* The torch host mirror must agree with the oracle to rounding on CPU tensors.
* On the GPU the full encoder -> policy -> drift -> Euler -> cost rollout (the cartpole loop's
  per-step path, C1-shaped) must agree with the oracle rollout.
"""
import numpy as np
import pytest
import torch

from gpflowpilco_amd import bijectors as tfb
from gpflowpilco_amd import dynamics, models as gp
from gpflowpilco_amd.components import GaussianObjective, TrigonometricEncoder, sincos
from gpflowpilco_amd.moment_matching import GaussianMoments, moment_matching
from gpflowpilco_amd.special import ndtr, owens_t
from gpflowpilco_amd.synthetic import generate_covariance, make_svgp
from oracle import mm_compose_oracle as co
from oracle import mm_oracle as mo
from oracle.pin_oracle import draw_samples_mvn, mc_tol
from tests.helpers import gp_model_from_oracle, oracle_params, random_svgp_params, scale_err, to_dev

F64 = torch.float64
NS = int(4e5)


def test_oracle_expected_cost_vs_monte_carlo():
  """tests/test_components.py:39-66."""
  rng = np.random.default_rng(0)
  d = 4
  mx = rng.standard_normal(d); Sxx = generate_covariance(rng, d, scale=0.1)
  mt = mx + 0.1 * rng.standard_normal(d); iStt = np.linalg.inv(generate_covariance(rng, d, scale=0.1))
  X = draw_samples_mvn(rng, mx, Sxx, NS)
  err = X - mt
  mc = np.mean(-np.exp(-0.5 * np.einsum('ni,ij,nj->n', err, iStt, err)))
  assert abs(co.expected_gaussian_cost(mx, Sxx, mt, iStt) - mc) <= mc_tol(NS)


def test_oracle_trig_encoder_vs_monte_carlo():
  """tests/test_components.py:70-104."""
  rng = np.random.default_rng(1)
  d = 4; active = (2, 3)
  mx = rng.standard_normal(d); Sxx = generate_covariance(rng, d, scale=0.3)
  m = co.mm_encoder((mx[None], Sxx[None], True), active)
  X = draw_samples_mvn(rng, mx, Sxx, NS)
  E = np.concatenate([np.sin(X[:, active]), np.cos(X[:, active]), X[:, (0, 1)]], -1)
  tol = mc_tol(NS)
  assert np.abs(m["y"][0][0] - E.mean(0)).max() <= tol
  assert np.abs(co.covariance(m["y"])[0] - np.cov(E.T)).max() <= tol
  Xc = X - X.mean(0)
  assert np.abs(co.cross_covariance(m)[0] - Xc.T @ (E - E.mean(0)) / NS).max() <= tol


def test_oracle_policy_head_vs_monte_carlo():
  """Chain[Scale, Shift, NormalCDF] on a 1-D Gaussian (bijectors.py:21-69; no reference test)."""
  rng = np.random.default_rng(2)
  m0 = np.array([[0.3]]); v0 = np.array([[[0.6]]])
  ops = [lambda s: co.mm_mul(s, 20.0 - 1e-5), lambda s: co.mm_add(s, -0.5), co.mm_ndtr]
  mt = co.mm_chain((m0, v0, True), ops)
  z = m0[0, 0] + np.sqrt(v0[0, 0, 0]) * rng.standard_normal(NS)
  u = (20.0 - 1e-5) * (co.ndtr(z) - 0.5)
  tol = mc_tol(NS) * 20
  assert abs(mt["y"][0][0, 0] - u.mean()) <= tol
  assert abs(co.covariance(mt["y"])[0, 0, 0] - u.var()) <= tol * 20
  assert abs(co.cross_covariance(mt)[0, 0, 0] - np.mean((z - z.mean()) * (u - u.mean()))) <= tol


def test_special_functions():
  from scipy.special import owens_t as ot
  h = torch.linspace(-4, 4, 41, dtype=F64); a = torch.linspace(0.0, 1.0, 41, dtype=F64)
  H, A = torch.meshgrid(h, a, indexing="ij")
  assert np.abs(owens_t(H, A).numpy() - ot(H.numpy(), A.numpy())).max() < 1e-14
  assert np.abs(ndtr(h).numpy() - co.ndtr(h.numpy())).max() < 1e-15


def _mom(mu, S, device="cpu", dtype=F64):
  return GaussianMoments((to_dev(mu, device, dtype), to_dev(S, device, dtype)), centered=True)


def test_host_elementary_maps_match_oracle():
  rng = np.random.default_rng(3)
  mu = rng.standard_normal((3, 2)); S = generate_covariance(rng, 2, (3,), 0.4)
  x = _mom(mu, S)
  o = co.mm_sincos((mu, S, True))
  h = moment_matching(x, sincos)
  assert np.allclose(h.y.mean().numpy(), o["y"][0]) and np.allclose(h.y[1].numpy(), o["y"][1])
  assert np.allclose(h.y.covariance().numpy(), co.covariance(o["y"]))
  assert np.allclose(h.cross_covariance().numpy(), co.cross_covariance(o))
  # sin / cos alone are the corresponding blocks of sincos
  hs, hc = moment_matching(x, torch.sin), moment_matching(x, torch.cos)
  assert np.allclose(hs.y.mean().numpy(), o["y"][0][:, :2]) and np.allclose(hc.y[1].numpy(), o["y"][1][:, 2:, 2:])
  assert np.allclose(hs.cross_covariance().numpy(), co.cross_covariance(o)[:, :, :2])
  # add / mul / sub / matvec
  assert np.allclose(moment_matching(x, torch.add, 0.7).y.mean().numpy(), mu + 0.7)
  assert np.allclose(moment_matching(x, torch.sub, 0.7).y.mean().numpy(), mu - 0.7)
  mm = moment_matching(x, torch.mul, 3.0)
  assert np.allclose(mm.y.covariance().numpy(), 9 * S) and np.allclose(mm.cross_covariance().numpy(), 3 * S)
  Am = torch.tensor(rng.standard_normal((3, 2)), dtype=F64)
  mv = moment_matching(x, torch.mv, Am)
  assert np.allclose(mv.y.covariance().numpy(), Am.numpy() @ S @ Am.numpy().T)
  assert np.allclose(mv.cross_covariance().numpy(), S @ Am.numpy().T)


def test_host_encoder_and_bijector_chain_match_oracle():
  rng = np.random.default_rng(4)
  mu = rng.standard_normal((2, 4)); S = generate_covariance(rng, 4, (2,), 0.3)
  enc = TrigonometricEncoder(active_dims=(1,))
  h = moment_matching(_mom(mu, S), enc)
  o = co.mm_encoder((mu, S, True), (1,))
  assert np.allclose(h.y.mean().numpy(), o["y"][0]) and np.allclose(h.y.covariance().numpy(), co.covariance(o["y"]))
  assert np.allclose(h.cross_covariance().numpy(), co.cross_covariance(o)) and h.cross[1] is False
  assert torch.allclose(enc(torch.tensor(mu)), torch.tensor(
      np.concatenate([np.sin(mu[:, 1:2]), np.cos(mu[:, 1:2]), mu[:, [0, 2, 3]]], -1)))
  m1 = rng.standard_normal((3, 1)); v1 = rng.uniform(0.1, 1.0, (3, 1, 1))
  chain = tfb.Chain([tfb.Scale(20.0 - 1e-5), tfb.Shift(-0.5), tfb.NormalCDF()])
  hb = moment_matching(_mom(m1, v1), chain)
  ob = co.mm_chain((m1, v1, True), [lambda s: co.mm_mul(s, 20.0 - 1e-5), lambda s: co.mm_add(s, -0.5), co.mm_ndtr])
  assert np.allclose(hb.y.mean().numpy(), ob["y"][0], atol=1e-13)
  assert np.allclose(hb.y.covariance().numpy(), co.covariance(ob["y"]), atol=1e-12)
  assert np.allclose(hb.cross_covariance().numpy(), co.cross_covariance(ob), atol=1e-13)
  assert torch.allclose(chain(torch.tensor(m1)), 19.99999 * (ndtr(torch.tensor(m1)) - 0.5))
  with pytest.raises(NotImplementedError):
    moment_matching(_mom(mu, S), tfb.NormalCDF())


def _cartpole_like(seed=0):
  """x (4) -> e (5) -> u (1) -> d (6) -> dx (4): swingup_loops.py:41-53, loops/pilco.py:40-108."""
  drift_syn = make_svgp(4, 60, 6, seed=seed, ls_bounds=(0.8, 3.0))
  drift_o = oracle_params(drift_syn)
  drift_o.Z = drift_o.Z * np.array([1, 1, 1, 1, 1, 4.0]) - np.array([0, 0, 0, 0, 0, 2.0])   # action axis in [-2, 2]
  pol_o = random_svgp_params(seed=seed + 1, L=1, M=30, d=5, whiten=True, ls_bounds=(0.7, 2.0), mean=False)
  pol_o.q_mu = 0.3 * pol_o.q_mu
  rng = np.random.default_rng(seed + 2)
  mu = np.array([[0.4, 0.2, 0.5, 0.3], [0.6, -0.1, 0.4, 0.5]])
  S = generate_covariance(rng, 4, (2,), 0.05)
  target = np.array([0.0, 1.0, 0.0, 0.0, 0.0]) * 0 + np.array([np.sin(0.0), np.cos(0.0), 0, 0, 0])
  precis = 16 * np.array([[0.25, 0, -0.5, 0, 0], [0, 0.25, 0, 0, 0], [-0.5, 0, 1, 0, 0], [0] * 5, [0] * 5], dtype=float)
  return drift_o, pol_o, mu, S, target, precis


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_gpu_policy_rollout_matches_oracle(dtype, device):
  drift_o, pol_o, mu, S, target, precis = _cartpole_like()
  scale, shift, active, H = 2.0, -0.5, (1,), 4
  policy_fn = lambda s: co.mm_policy(s, pol_o, scale, shift)
  loss_o, traj_o = co.policy_rollout_loss(mu, S, drift_o, policy_fn, active, target, precis, H, keep=True)
  # one composed step first
  mo_ = co.forward_sde_full((mu, S, True), drift_o, policy_fn, active)

  drift = gp_model_from_oracle(drift_o, device)
  policy = gp.InverseLinkWrapper(gp.KernelRegressor(gp_model_from_oracle(pol_o, device)),
                                 invlink=tfb.Chain([tfb.Scale(scale), tfb.Shift(shift), tfb.NormalCDF()]))
  encoder = TrigonometricEncoder(active_dims=active)
  objective = GaussianObjective(target=to_dev(target, device, dtype), precis=to_dev(precis, device, dtype))
  system = dynamics.DynamicalSystem(drift=drift, policy=policy, encoder=encoder,
                                    solver=dynamics.MomentMatchingEuler())
  x = _mom(mu, S, device, dtype)
  tol = 1e-7 if dtype == torch.float64 else 5e-4
  md, _ = system.forward(0.0, x)
  assert md.cross[1] is False
  errs = [scale_err(md.y.mean(), mo_["y"][0]), scale_err(md.y.covariance(), co.covariance(mo_["y"])),
          scale_err(md.cross[0], mo_["cross"][0])]
  assert max(errs) < tol, errs

  def accumulate(t, state, loss):                                    # loops/pilco.py:199-205
    e = moment_matching(GaussianMoments(state, centered=True), encoder).y
    return loss + objective(x=e, t=t)
  out = system.solve_forward(initial_time=0.0, initial_state=(x.mean(), x.covariance()),
                             solution_times=np.arange(1.0, H + 1.0), iterator="foldl",
                             callbacks_and_initializers=[(accumulate, torch.zeros(2, dtype=dtype, device=device))])
  (m_H, S_H), loss = out[0], out[1]
  errs = [scale_err(m_H, traj_o[-1][0]), scale_err(S_H, traj_o[-1][1]), scale_err(loss, loss_o)]
  assert max(errs) < tol, errs
  # the same through the MomentMatchingPILCO harness (loops/pilco.py:176-220)
  from gpflowpilco_amd.loops import get_state_initializer, policy_loss_closure
  closure = policy_loss_closure(system, objective, get_state_initializer(x.mean(), x.covariance()), H, native=False)
  assert torch.allclose(closure(), loss)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_native_composed_rollout_matches_oracle_and_torch_path(dtype, device):
  """mm_rollout_composed (csrc/mm_compose.hip): the WHOLE policy rollout on the device -- encoder sincos moments,
  policy GP + NormalCDF head (Owen's T), joint, drift GP, forward_sde's cross-covariance bookkeeping, Euler update,
  per-step expected cost -- against the numpy oracle rollout and against the torch composition of the same kernels.
  Config-1 sizes: drift M = 100, H = 30 (f64: 1e-7; f32: 2e-4 of the per-quantity scale)."""
  from gpflowpilco_amd import ops
  from gpflowpilco_amd.loops import get_state_initializer, native_policy_loss, policy_loss_closure
  drift_syn = make_svgp(4, 100, 6, seed=10, ls_bounds=(0.8, 3.0))
  drift_o = oracle_params(drift_syn)
  drift_o.Z = drift_o.Z * np.array([1, 1, 1, 1, 1, 4.0]) - np.array([0, 0, 0, 0, 0, 2.0])
  pol_o = random_svgp_params(seed=11, L=1, M=30, d=5, whiten=True, ls_bounds=(0.7, 2.0), mean=False)
  pol_o.q_mu = 0.3 * pol_o.q_mu
  rng = np.random.default_rng(12)
  mu = np.array([[0.4, 0.2, 0.5, 0.3], [0.6, -0.1, 0.4, 0.5], [0.5, 0.0, 0.45, 0.4]])
  S = generate_covariance(rng, 4, (3,), 0.05)
  target = np.array([0.0, 1.0, 0, 0, 0])
  precis = 16 * np.array([[0.25, 0, -0.5, 0, 0], [0, 0.25, 0, 0, 0], [-0.5, 0, 1, 0, 0], [0] * 5, [0] * 5], dtype=float)
  scale, shift, active, H = 2.0, -0.5, (1,), 30
  policy_fn = lambda s: co.mm_policy(s, pol_o, scale, shift)
  loss_o, traj_o = co.policy_rollout_loss(mu, S, drift_o, policy_fn, active, target, precis, H, keep=True)

  drift = gp_model_from_oracle(drift_o, device)
  pol_model = gp_model_from_oracle(pol_o, device)
  roll = ops.ComposedRollout(drift.packed(dtype, True, device), pol_model.packed(dtype, False, device), nx=4,
                             active_dims=active, head_scale=scale, head_shift=shift,
                             target=to_dev(target, device, dtype), precis=to_dev(precis, device, dtype))
  m_H, S_H, cost, tmu, tS = roll(to_dev(mu, device, dtype), to_dev(S, device, dtype), H, keep_trajectory=True)
  roll.drift.check_status(3)
  tol = 1e-7 if dtype == torch.float64 else 2e-4
  for h in (0, 1, H // 2, H - 1):
    assert scale_err(tmu[h], traj_o[h][0]) < tol and scale_err(tS[h], traj_o[h][1]) < tol, h
  assert scale_err(cost.sum(1), loss_o) < tol
  assert torch.equal(m_H, tmu[-1]) and torch.equal(S_H, tS[-1])
  # the torch composition of the same GP kernels (dynamics.forward_sde + moment_matching/{maths,components,bijectors}.py)
  policy = gp.InverseLinkWrapper(gp.KernelRegressor(pol_model),
                                 invlink=tfb.Chain([tfb.Scale(scale), tfb.Shift(shift), tfb.NormalCDF()]))
  encoder = TrigonometricEncoder(active_dims=active)
  objective = GaussianObjective(target=to_dev(target, device, dtype), precis=to_dev(precis, device, dtype))
  system = dynamics.DynamicalSystem(drift=drift, policy=policy, encoder=encoder, solver=dynamics.MomentMatchingEuler())
  init = get_state_initializer(to_dev(mu, device, dtype), to_dev(S, device, dtype))
  loss_t = policy_loss_closure(system, objective, init, H, native=False)()
  loss_n = policy_loss_closure(system, objective, init, H)()            # picks the native rollout by itself
  assert scale_err(loss_n, loss_t.double().cpu().numpy()) < (1e-9 if dtype == torch.float64 else 2e-4)
  assert torch.allclose(loss_n, cost.sum(1))
  assert native_policy_loss(system, objective, H) is not None


@pytest.mark.gpu
def test_native_closure_follows_in_place_parameter_updates(device):
  """The native rollout reads the models through their packed snapshots: an in-place update of a policy parameter
  (an optimiser step), of a drift parameter (a refit between episodes), of the head constants or of the objective
  must show up in the next call -- no stale ComposedRollout (loops.native_policy_loss)."""
  from gpflowpilco_amd.loops import get_state_initializer, policy_loss_closure
  dtype = F64
  drift_o = oracle_params(make_svgp(4, 48, 6, seed=20, ls_bounds=(0.8, 3.0)))
  pol_o = random_svgp_params(seed=21, L=1, M=16, d=5, whiten=True, ls_bounds=(0.7, 2.0), mean=False)
  pol_o.q_mu = 0.3 * pol_o.q_mu
  rng = np.random.default_rng(22)
  mu = np.array([[0.4, 0.2, 0.5, 0.3], [0.6, -0.1, 0.4, 0.5]])
  S = generate_covariance(rng, 4, (2,), 0.05)
  target = np.array([0.0, 1.0, 0, 0, 0]); precis = 4.0 * np.eye(5)
  drift = gp_model_from_oracle(drift_o, device)
  pol_model = gp_model_from_oracle(pol_o, device)
  head = tfb.Chain([tfb.Scale(2.0), tfb.Shift(-0.5), tfb.NormalCDF()])
  policy = gp.InverseLinkWrapper(gp.KernelRegressor(pol_model), invlink=head)
  objective = GaussianObjective(target=to_dev(target, device, dtype), precis=to_dev(precis, device, dtype))
  system = dynamics.DynamicalSystem(drift=drift, policy=policy, encoder=TrigonometricEncoder(active_dims=(1,)),
                                    solver=dynamics.MomentMatchingEuler())
  init = get_state_initializer(to_dev(mu, device, dtype), to_dev(S, device, dtype))
  H = 5
  native = policy_loss_closure(system, objective, init, H, native=True)
  torch_path = policy_loss_closure(system, objective, init, H, native=False)
  pol_model.q_mu.requires_grad_(True)                       # a trainable policy evaluated under no_grad (metrics, line search)

  def both():
    with torch.no_grad():
      return native(), torch_path()
  n0, t0 = both()
  assert scale_err(n0, t0.cpu().numpy()) < 1e-9
  with torch.no_grad():
    pol_model.q_mu.mul_(1.5)                                # optimiser step on the policy
  n1, t1 = both()
  assert scale_err(n1, t1.cpu().numpy()) < 1e-9 and (n1 - n0).abs().max() > 1e-6
  with torch.no_grad():
    drift.q_mu.mul_(0.9)                                    # drift refit in place between episodes
  n2, t2 = both()
  assert scale_err(n2, t2.cpu().numpy()) < 1e-9 and (n2 - n1).abs().max() > 1e-6
  head.bijectors[0].scale = 1.5                             # head constant
  objective.target.add_(0.1)                                # objective, in place
  n3, t3 = both()
  assert scale_err(n3, t3.cpu().numpy()) < 1e-9 and (n3 - n2).abs().max() > 1e-6
