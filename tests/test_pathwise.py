"""Pathwise rollout.  The arithmetic of this path is in the un-vendored
gpflow-sampling package, so parity is UNPINNED; what is checked:
  * statistical pin of the oracle construction: sample mean / variance of the paths at fixed
    inputs equal the SVGP predictive mean / variance (CPU);
  * the HIP kernel reproduces the oracle on identical path tensors (GPU, f64 / f32);
  * torch path generation has the right statistics; rollouts match the oracle rollout.
"""
import numpy as np
import pytest
import torch

from gpflowpilco_amd.synthetic import make_svgp
from oracle import pathwise_oracle as pw
from oracle.pin_oracle import svgp_predict_f
from tests.helpers import gp_model_from_oracle, oracle_params, random_svgp_params, scale_err


def test_oracle_paths_match_predictive_moments():
  p = random_svgp_params(seed=2, L=2, M=24, d=3, whiten=True, ls_bounds=(0.5, 2.0))
  rng = np.random.default_rng(0)
  S, K = 6000, 2048
  paths = pw.draw_paths(rng, p, S, K)
  xs = rng.uniform(size=(3, 3))
  mean, cov = svgp_predict_f(xs, p)
  for i in range(3):
    f = pw.eval_paths(paths, p, np.broadcast_to(xs[i], (S, 3)))
    se = np.sqrt(np.diagonal(cov[i]) / S)
    assert np.all(np.abs(f.mean(0) - mean[i]) < 6 * se + 2e-2)           # RFF bias allowance
    assert np.all(np.abs(f.var(0) - np.diagonal(cov[i])) < 0.1 * np.diagonal(cov[i]) + 2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("shape", [(3, 50, 3, 37, 130), (8, 203, 8, 9, 1024), (2, 64, 16, 5, 66),
                                   (2, 1100, 16, 6, 1500)],      # operands beyond 144 KB of LDS: streaming kernel
                         ids=["small", "d8", "d16", "d16_no_lds"])
def test_gpu_eval_and_rollout_match_oracle(shape, dtype, device):
  from gpflowpilco_amd.pathwise import paths_from_arrays
  L, M, d, S, K = shape
  syn = make_svgp(L, M, d, seed=5 + L, ls_bounds=(0.7, 3.0), mean_c=True)
  po = oracle_params(syn)
  rng = np.random.default_rng(1)
  paths = pw.draw_paths(rng, po, S, K)
  x = rng.uniform(0.2, 0.8, size=(S, d))
  fo = pw.eval_paths(paths, po, x)
  gp_paths = paths_from_arrays(paths.omega, paths.phase, paths.w, paths.v, po.Z, po.lengthscales, po.variance,
                               po.mean_c, dtype=dtype, device=device)
  xt = torch.tensor(x, dtype=dtype, device=device)
  # f32: v = Kuu^-1 (u - Phi w) has entries of 1e2..1e4 that cancel in sum_m v_m k(x, z_m); storing v
  # and evaluating exp in f32 leaves ~3e-4 relative error (measured), f64 is exact to 1e-12
  tol = 1e-11 if dtype == torch.float64 else 2e-3
  assert scale_err(gp_paths(xt), fo) < tol
  fb, err = gp_paths.eval_with_bound(xt)                        # the rounding bound of this dtype holds for every value
  assert np.all(np.abs(fb.double().cpu().numpy() - fo) <= err.double().cpu().numpy() + 1e-300)
  if L == d:
    xo, traj = pw.rollout(paths, po, x, 4, dt=0.5, keep=True)
    xg, tg = gp_paths.rollout(xt, 4, dt=0.5, keep_trajectory=True)
    rtol = 1e-10 if dtype == torch.float64 else 5e-3
    assert scale_err(tg, traj) < rtol and torch.equal(xg, tg[-1])
    x3 = gp_paths.rollout(xt, 3, dt=0.5)                      # odd step count: result copied back
    assert scale_err(x3, traj[2]) < rtol


@pytest.mark.gpu
def test_generated_paths_statistics_and_model_surface(device):
  from gpflowpilco_amd.pathwise import PathwiseSVGP
  p = random_svgp_params(seed=4, L=2, M=20, d=2, whiten=True, ls_bounds=(0.5, 2.0))
  base = gp_model_from_oracle(p, device)
  model = PathwiseSVGP(kernel=base.kernel, inducing_variable=base.inducing_variable, q_mu=base.q_mu,
                       q_sqrt=base.q_sqrt, whiten=True, mean_function=base.mean_function, num_latent_gps=2)
  S = 8192
  g = torch.Generator(device=device).manual_seed(0)
  paths = model.generate_paths(num_samples=S, num_bases=2048, dtype=torch.float64, device=device, generator=g)
  x0 = np.array([0.3, 0.6])
  mean, cov = svgp_predict_f(x0[None], p)
  with model.set_temporary_paths(paths):
    f = model(torch.tensor(np.broadcast_to(x0, (S, 2)).copy(), device=device))
  fm, fv = f.mean(0).cpu().numpy(), f.var(0).cpu().numpy()
  se = np.sqrt(np.diagonal(cov[0]) / S)
  assert np.all(np.abs(fm - mean[0]) < 6 * se + 2e-2)
  assert np.all(np.abs(fv - np.diagonal(cov[0])) < 0.1 * np.diagonal(cov[0]) + 2e-2)
  with pytest.raises(RuntimeError, match="no sample paths"):
    model(torch.zeros(S, 2, dtype=torch.float64, device=device))


@pytest.mark.gpu
def test_c5_shard_shape_against_oracle_on_a_slice_of_samples(device):
  """BASELINE configs[4] per-GPU shard: S = 8192 (65536 / 8) sample paths, N = 2000, K = 1024 bases, d = D = 8,
  fp32 -- the LDS-resident streaming kernel at the shape the bench runs.  The HIP evaluation and a 3-step rollout
  of ALL paths run on the GPU; the numpy oracle re-evaluates a slice of 24 samples spread over the stream
  (first / middle / last sample groups) on identical path tensors."""
  from gpflowpilco_amd.pathwise import paths_from_arrays
  L, M, d, S, K = 8, 2000, 8, 8192, 1024
  syn = make_svgp(L, M, d, seed=1004, device=str(device), ls_bounds=(0.7, 3.0))
  po = oracle_params(syn)
  rng = np.random.default_rng(11)
  paths = pw.draw_paths(rng, po, S, K)
  x = rng.uniform(0.3, 0.7, size=(S, d))
  gp_paths = paths_from_arrays(paths.omega, paths.phase, paths.w, paths.v, po.Z, po.lengthscales, po.variance,
                               po.mean_c, dtype=torch.float32, device=device)
  xt = torch.tensor(x, dtype=torch.float32, device=device)
  fg = gp_paths(xt)
  assert torch.isfinite(fg).all()
  sel = np.r_[0:8, 4093:4101, 8184:8192]
  sub = pw.Paths(omega=paths.omega, phase=paths.phase, w=paths.w[sel], v=paths.v[sel])
  fo = pw.eval_paths(sub, po, x[sel])
  # f32 at this shape: v = Kuu^-1 (u - Phi w) has entries up to ~1e5 at M = 2000 (Kuu + 1e-6 I, cond ~1e9) that
  # cancel in sum_m v_m k(x, z_m); storing v in f32 leaves 1.8e-2 of max|f| (measured) -- the price of the config's
  # fp32 weight stream.  The same kernel in f64 on the same tensors is exact to 1e-9 (below).
  assert scale_err(fg[torch.tensor(sel, device=device)], fo) < 4e-2
  # ... and the caller is TOLD: the same pass with the sum of the absolute terms (mm_pathwise_eval_bound).  (i) the bound holds on
  # the checked slice; (ii) the (sample, latent) values it does not flag at 1e-3 of max|f| are within 1e-3; (iii) the count of
  # flagged values is reported (bench.py --config c5 prints it beside `parity`) -- at this shape nearly every value is flagged:
  # that is the honest statement about an f32 weight stream at M = 2000
  fb, err = gp_paths.eval_with_bound(xt)
  seld = torch.tensor(sel, device=device)
  adiff = np.abs(fb[seld].double().cpu().numpy() - fo)
  eb = err[seld].double().cpu().numpy()
  assert np.all(adiff <= eb), float((adiff / eb).max())
  assert float((adiff / eb).max()) > 0.02                       # ... and it is not vacuous: within 50 x of the worst error
  fmax = float(fb.abs().amax())
  _, mask, nflag = gp_paths.flagged(xt, 1e-3)
  ok = ~mask[seld].cpu().numpy()
  assert np.all(adiff[ok] <= 1e-3 * fmax)
  assert 0 <= nflag <= S * L
  print(f"C5 shard f32: {nflag} of {S * L} (sample, latent) values flagged at 1e-3 of max|f|; worst error / bound {float((adiff / eb).max()):.3f}")
  xo, traj = pw.rollout(sub, po, x[sel], 3, dt=0.5, keep=True)
  xg, tg = gp_paths.rollout(xt, 3, dt=0.5, keep_trajectory=True)
  assert scale_err(tg[:, torch.tensor(sel, device=device)], traj) < 4e-2
  # f64 mode at the same (M, K): the first 256 paths
  sub64 = pw.Paths(omega=paths.omega, phase=paths.phase, w=paths.w[:256], v=paths.v[:256])
  gp64 = paths_from_arrays(sub64.omega, sub64.phase, sub64.w, sub64.v, po.Z, po.lengthscales, po.variance, po.mean_c,
                           dtype=torch.float64, device=device)
  f64 = gp64(torch.tensor(x[:256], dtype=torch.float64, device=device))
  assert scale_err(f64[:24], pw.eval_paths(pw.Paths(omega=paths.omega, phase=paths.phase, w=paths.w[:24], v=paths.v[:24]),
                                           po, x[:24])) < 1e-9


# ---- the pathwise POLICY rollout and its gradient (row f-3 completed) ------------------------------------------------
def _policy_case(device, dtype, S=37, K=130, M=50, Mp=12, seed=3):
  """Cartpole wiring at small sizes: nx = 4 (one angle) -> ne = 5 -> nd = 6; drift paths of 4 latents; policy of Mp centres."""
  from gpflowpilco_amd.pathwise import PolicyRollout, paths_from_arrays
  rng = np.random.default_rng(seed)
  drift = oracle_params(make_svgp(4, M, 6, seed=seed + 1, ls_bounds=(0.8, 3.0)))
  drift.Z = drift.Z * np.array([1, 1, 1, 1, 1, 4.0]) - np.array([0, 0, 0, 0, 0, 2.0])
  pol = random_svgp_params(seed=seed + 2, L=1, M=Mp, d=5, whiten=True, ls_bounds=(0.8, 2.0), mean=True)
  pol.q_mu = 0.3 * pol.q_mu
  paths = pw.draw_paths(rng, drift, S, K)
  paths.w *= 0.3; paths.v *= 0.3                                  # (keeps 6-step sample rollouts inside the data's support)
  x0 = rng.uniform(0.2, 0.8, size=(S, 4))
  target = np.array([0.0, 1.0, 0.2, 0.0, 0.1])
  A = rng.standard_normal((5, 5))
  precis = 0.5 * (A @ A.T) / 5 + 0.5 * np.eye(5)
  scale, shift, active = 2.0, -0.5, (1,)
  gp_paths = paths_from_arrays(paths.omega, paths.phase, paths.w, paths.v, drift.Z, drift.lengthscales, drift.variance,
                               drift.mean_c, dtype=dtype, device=device)
  pol_model = gp_model_from_oracle(pol, device)
  roll = PolicyRollout(gp_paths, pol_model.packed(torch.float64, False, device), nx=4, active_dims=active, head_scale=scale,
                       head_shift=shift, target=torch.tensor(target), precis=torch.tensor(precis))
  return dict(paths=paths, drift=drift, pol=pol, x0=x0, target=target, precis=precis, scale=scale, shift=shift, active=active,
              gp_paths=gp_paths, pol_model=pol_model, roll=roll)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_gpu_jacobian_of_the_paths_matches_finite_differences_of_the_oracle(dtype, device):
  c = _policy_case(device, dtype)
  rng = np.random.default_rng(9)
  x = rng.uniform(0.2, 0.8, size=(37, 6))
  xt = torch.tensor(x, dtype=dtype, device=device)
  f, J = c["gp_paths"].eval_jac(xt)
  fo = pw.eval_paths(c["paths"], c["drift"], x)
  assert scale_err(f, fo) < (1e-11 if dtype == torch.float64 else 2e-3)
  assert torch.equal(f, c["gp_paths"](xt))                        # the Jacobian pass returns the plain pass's values
  Jo = np.empty((37, 4, 6))
  h = 1e-6
  for k in range(6):
    dx = np.zeros(6); dx[k] = h
    Jo[:, :, k] = (pw.eval_paths(c["paths"], c["drift"], x + dx) - pw.eval_paths(c["paths"], c["drift"], x - dx)) / (2 * h)
  assert scale_err(J, Jo) < (1e-7 if dtype == torch.float64 else 3e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_gpu_policy_rollout_costs_match_the_oracle(dtype, device):
  c = _policy_case(device, dtype)
  H = 6
  co, so = pw.policy_rollout_costs(c["paths"], c["drift"], c["pol"], c["scale"], c["shift"], c["active"], c["target"],
                                   c["precis"], c["x0"], H, dt=0.5, keep=True)
  x0 = torch.tensor(c["x0"], dtype=dtype, device=device)
  for jac in (False, True):
    cost, tape = c["roll"](x0, H, dt=0.5, with_jacobians=jac)
    tol = 1e-10 if dtype == torch.float64 else 5e-3
    assert scale_err(cost, co) < tol
    assert scale_err(c["roll"].states(tape, H), so) < tol


@pytest.mark.gpu
def test_gpu_policy_gradient_of_the_mean_sample_loss(device):
  """The gradient the reference takes with a tape (train_utils.py:108-135: mean over samples of the summed costs) w.r.t. every
  policy parameter and the initial states: native reverse sweep vs (i) central differences of the numpy oracle on sampled
  coordinates (1e-6) and (ii) torch autograd of a torch mirror of the same composition."""
  from gpflowpilco_amd.pathwise import PolicyRolloutFunction
  F64 = torch.float64
  c = _policy_case(device, F64)
  H, dt, S = 5, 0.5, 37
  pm = c["pol_model"]
  kern = pm.kernel.kernels[0]
  params = {"q_mu": pm.q_mu, "Z": pm.inducing_variable.inducing_variables[0].Z, "lengthscales": kern.lengthscales,
            "variance": kern.variance}
  for t in params.values():
    t.requires_grad_(True)
  x0 = torch.tensor(c["x0"], dtype=F64, device=device, requires_grad=True)

  def native_loss():
    Zp, lsp, varp, betap, _, mcp = pm.precompute(device)
    cost = PolicyRolloutFunction.apply(x0, Zp, lsp, varp, betap, mcp, c["roll"], H, dt)       # [S, H]
    return cost.sum(1).mean()
  loss = native_loss()
  loss.backward()
  g_native = {k: t.grad.detach().clone() for k, t in params.items()}
  g_native["x0"] = x0.grad.detach().clone()
  co = pw.policy_rollout_costs(c["paths"], c["drift"], c["pol"], c["scale"], c["shift"], c["active"], c["target"], c["precis"],
                               c["x0"], H, dt=dt)
  assert abs(float(loss) - co.sum(0).mean()) < 1e-10

  # (ii) torch mirror: the same composition in differentiable torch ops on the path arrays
  P = c["paths"]; dr = c["drift"]
  tt = lambda a: torch.tensor(np.asarray(a), dtype=F64, device=device)
  om, ph, w, v = tt(P.omega), tt(P.phase), tt(P.w), tt(P.v)
  Zd, lsd, vard = tt(dr.Z), tt(dr.lengthscales), tt(dr.variance)
  target, precis = tt(c["target"]), tt(c["precis"])

  def mirror_loss():
    Zp, lsp, varp, betap, _, mcp = pm.precompute(device)
    x = x0
    tot = 0.0
    enc = lambda y: torch.cat([torch.sin(y[:, 1:2]), torch.cos(y[:, 1:2]), y[:, [0, 2, 3]]], dim=-1)
    for _ in range(H):
      e = enc(x)
      r2 = (((e[:, None, :] - Zp[0][None]) / lsp[0]) ** 2).sum(-1)
      fp = (varp[0] * torch.exp(-0.5 * r2)) @ betap[0] + mcp[0]
      u = c["scale"] * (0.5 * torch.erfc(-fp / np.sqrt(2.0)) + c["shift"])
      dd = torch.cat([e, u[:, None]], dim=-1)
      f = []
      for a in range(4):
        phi = torch.sqrt(2.0 * vard[a] / om.shape[1]) * torch.cos(dd @ om[a].T + ph[a][None])
        kk = vard[a] * torch.exp(-0.5 * (((dd[:, None, :] - Zd[a][None]) / lsd[a]) ** 2).sum(-1))
        f.append((w[:, a] * phi).sum(-1) + (v[:, a] * kk).sum(-1) + (0.0 if dr.mean_c is None else float(dr.mean_c[a])))
      x = x + dt * torch.stack(f, dim=-1)
      err = enc(x) - target
      tot = tot - torch.exp(-0.5 * ((err @ precis) * err).sum(-1))
    return tot.mean()
  for t in list(params.values()) + [x0]:
    t.grad = None
  lm = mirror_loss()
  lm.backward()
  assert abs(float(lm) - float(loss)) < 1e-10
  for k, t in list(params.items()) + [("x0", x0)]:
    ref = t.grad.detach()
    err = float((g_native[k] - ref).abs().max()) / max(1e-14, float(ref.abs().max()))
    assert err < 1e-8, (k, err)

  # (i) central differences of the numpy oracle along random directions of every parameter group
  import copy
  rng = np.random.default_rng(5)

  def oracle_loss(pol, x_init):
    return pw.policy_rollout_costs(c["paths"], c["drift"], pol, c["scale"], c["shift"], c["active"], c["target"], c["precis"],
                                   x_init, H, dt=dt).sum(0).mean()
  h = 1e-6
  for name, field in (("q_mu", "q_mu"), ("Z", "Z"), ("lengthscales", "lengthscales"), ("variance", "variance")):
    base = np.asarray(getattr(c["pol"], field), dtype=np.float64)
    dirn = rng.standard_normal(base.shape)
    lp, lm_ = [], []
    for sgn, out in ((1.0, lp), (-1.0, lm_)):
      pol2 = copy.deepcopy(c["pol"])
      setattr(pol2, field, base + sgn * h * dirn)
      out.append(oracle_loss(pol2, c["x0"]))
    fd = (lp[0] - lm_[0]) / (2 * h)
    an = float((g_native[name].cpu().numpy().reshape(base.shape) * dirn).sum())
    assert abs(fd - an) < 1e-6 * max(1.0, abs(fd)), (name, fd, an)
  dirx = rng.standard_normal(c["x0"].shape)
  fd = (oracle_loss(c["pol"], c["x0"] + h * dirx) - oracle_loss(c["pol"], c["x0"] - h * dirx)) / (2 * h)
  an = float((g_native["x0"].cpu().numpy() * dirx).sum())
  assert abs(fd - an) < 1e-6 * max(1.0, abs(fd)), ("x0", fd, an)


@pytest.mark.gpu
def test_pathwise_policy_loss_closure_native_equals_the_torch_composition(device):
  """loops.pathwise_policy_loss_closure (PathwisePILCO._policy_loss_closure, loops/pilco.py:263-298): the native closure and
  the torch composition through DynamicalSystem / Euler on the SAME sample paths give the same per-sample loss and the same
  gradient of its mean w.r.t. the policy parameters and the initial states; unsupported gradients warn and take the torch path."""
  import warnings
  from gpflowpilco_amd import bijectors as tfb, dynamics, models as gp
  from gpflowpilco_amd.components import GaussianObjective, TrigonometricEncoder
  from gpflowpilco_amd.loops import pathwise_policy_loss_closure
  from gpflowpilco_amd.pathwise import PathwiseSVGP
  F64 = torch.float64
  S, H = 41, 4
  base = make_svgp(4, 48, 6, seed=31, ls_bounds=(0.9, 3.0)).to_model(device)
  drift = PathwiseSVGP(kernel=base.kernel, inducing_variable=base.inducing_variable, q_mu=base.q_mu, q_sqrt=base.q_sqrt, whiten=True,
                       num_latent_gps=4)
  pol_o = random_svgp_params(seed=32, L=1, M=14, d=5, whiten=True, ls_bounds=(0.9, 2.0), mean=False)
  pol_o.q_mu = 0.2 * pol_o.q_mu
  pol = gp_model_from_oracle(pol_o, device)
  kern = pol.kernel.kernels[0]
  params = [pol.q_mu, pol.inducing_variable.inducing_variables[0].Z, kern.lengthscales, kern.variance]
  for t in params:
    t.requires_grad_(True)
  policy = gp.InverseLinkWrapper(gp.KernelRegressor(pol), invlink=tfb.Chain([tfb.Scale(2.0), tfb.Shift(-0.5), tfb.NormalCDF()]))
  system = dynamics.DynamicalSystem(drift=drift, policy=policy, encoder=TrigonometricEncoder(active_dims=(1,)), solver=dynamics.Euler())
  target = torch.tensor([0.0, 1.0, 0.2, 0.0, 0.1], dtype=F64, device=device)
  objective = GaussianObjective(target=target, precis=2.0 * torch.eye(5, dtype=F64, device=device))
  g = torch.Generator(device=device).manual_seed(3)
  x0 = (0.2 + 0.6 * torch.rand(S, 4, dtype=F64, device=device, generator=g)).requires_grad_(True)
  paths = drift.generate_paths(S, 256, dtype=F64, device=device, generator=g)

  def run(native):
    for t in params + [x0]:
      t.grad = None
    loss = pathwise_policy_loss_closure(system, objective, lambda: x0, H, dt=0.5, paths=paths, native=native)()
    loss.mean().backward()
    return loss.detach(), [t.grad.detach().clone() for t in params + [x0]]
  with warnings.catch_warnings():
    warnings.simplefilter("error", RuntimeWarning)
    ln, gn = run(None)
  lt, gt = run(False)
  assert ln.shape == (S,) and float((ln - lt).abs().max()) < 1e-10
  for a_, b_ in zip(gn, gt):
    assert float((a_ - b_).abs().max()) < 1e-8 * max(1e-12, float(b_.abs().max()))
  # forward only (nothing requires a gradient): the same numbers without the Jacobian tape
  with torch.no_grad():
    l0 = pathwise_policy_loss_closure(system, objective, lambda: x0.detach(), H, dt=0.5, paths=paths)()
  assert float((l0 - ln).abs().max()) < 1e-12
  # fresh paths on every call (pilco.py:281-284) with float32 states: runs, finite, and differs between calls
  x32 = x0.detach().float()
  cl = pathwise_policy_loss_closure(system, objective, lambda: x32, H, dt=0.5, num_bases=256)
  with torch.no_grad():
    a1, a2 = cl(), cl()
  assert a1.dtype == torch.float32 and torch.isfinite(a1).all() and not torch.equal(a1, a2)
  # a gradient the native sweep does not cover: warned, carried by the torch composition
  target.requires_grad_(True)
  with pytest.warns(RuntimeWarning, match="objective.target requires a gradient"):
    loss = pathwise_policy_loss_closure(system, objective, lambda: x0, H, dt=0.5, paths=paths)()
  loss.mean().backward()
  assert target.grad is not None and float(target.grad.abs().max()) > 0.0
