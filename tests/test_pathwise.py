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
