"""Host-side mirror of the reference interface on CPU tensors: dispatcher, carrier types,
GaussianMatch algebra, chain rule, model precompute, solvers.  No GPU compute is called."""
from functools import partial

import numpy as np
import pytest
import torch

from gpflowpilco_amd import dynamics, models as gp
from gpflowpilco_amd.moment_matching import (Chain, GaussianMatch, GaussianMoments, LinearOperatorDiag,
                                             Moments, dispatcher, moment_matching, register_type)
from gpflowpilco_amd.moment_matching.core import Dispatcher
from oracle import mm_fused_ref as fr
from oracle import mm_oracle as mo
from tests.helpers import gp_model_from_oracle, random_svgp_params

F64 = torch.float64


class Linear:
  """y = A x + c: exact moment match, used to exercise the composition rules."""
  def __init__(self, A, c):
    self.A, self.c = A, c


@dispatcher.register(GaussianMoments, Linear)
def _mm_linear(x, op, /, preinv=False):
  m, S = x.mean(), x.covariance(dense=True)
  y = GaussianMoments(((op.A @ m.unsqueeze(-1)).squeeze(-1) + op.c, op.A @ S @ op.A.T), centered=True)
  cross = op.A.T.expand(m.shape[:-1] + op.A.T.shape) if preinv else S @ op.A.T
  return GaussianMatch(x=x, y=y, cross=(cross.contiguous(), preinv))


def _state(B=3, d=3, seed=0):
  g = torch.Generator().manual_seed(seed)
  m = torch.randn(B, d, generator=g, dtype=F64)
  A = torch.randn(B, d, d, generator=g, dtype=F64)
  return GaussianMoments((m, A @ A.transpose(1, 2) + 0.1 * torch.eye(d, dtype=F64)), centered=True)


def test_dispatcher_resolution_and_errors():
  disp = Dispatcher("t")

  class A: pass
  class B(A): pass

  @disp.register(A, object)
  def _a(x, y): return "A"

  @disp.register(B, (int, float))
  def _b(x, y): return "B"

  assert disp(A(), 1) == "A" and disp(B(), 1) == "B" and disp(B(), "s") == "A" and disp(B(), 2.0) == "B"
  with pytest.raises(NotImplementedError, match="Could not find signature"):
    disp(1, 2)


def test_moments_and_linear_operator_diag():
  m = torch.tensor([[1.0, 2.0]], dtype=F64)
  m2 = torch.tensor([[[2.0, 2.5], [2.5, 5.0]]], dtype=F64)
  unc = Moments((m, m2), centered=False)
  assert torch.allclose(unc.covariance(), m2 - m.unsqueeze(-1) * m.unsqueeze(-2))
  assert Moments((m, m2), centered=True).covariance() is m2 and unc.ndim == 2 and unc.dtype == F64
  D = LinearOperatorDiag(torch.tensor([[2.0, 4.0]], dtype=F64))
  rhs = torch.ones(1, 2, 3, dtype=F64)
  assert torch.allclose(D.to_dense() @ rhs, D @ rhs) and torch.allclose(D.solve(D @ rhs), rhs)
  assert torch.equal(Moments((m, D), centered=True).covariance(dense=True), D.to_dense())


def test_gaussian_match_cross_covariance_and_joint():
  x = _state()
  A = torch.randn(2, 3, dtype=F64); c = torch.randn(2, dtype=F64)
  for preinv in (False, True):
    mt = moment_matching(x, Linear(A, c), preinv=preinv)
    Sxy = x.covariance() @ A.T
    assert torch.allclose(mt.cross_covariance(), Sxy)
    assert torch.allclose(mt.cross_covariance(preinv=True), A.T.expand(3, 3, 2))
    j = mt.joint()
    assert j.mean().shape == (3, 5) and torch.allclose(j.covariance()[:, :3, 3:], Sxy)
    assert torch.allclose(j.covariance(), j.covariance().transpose(1, 2))
    Sx = x.covariance().numpy()
    want = mo.cross_covariance(Sx, mt.cross[0].numpy(), is_preinv=preinv, preinv=not preinv)
    assert np.allclose(mt.cross_covariance(preinv=not preinv).numpy(), want)


def test_chain_rule_and_partial_and_register_type():
  x = _state(seed=1)
  A1 = torch.randn(3, 3, dtype=F64); A2 = torch.randn(2, 3, dtype=F64)
  mt = moment_matching(x, Chain(Linear(A2, torch.zeros(2, dtype=F64)), Linear(A1, torch.ones(3, dtype=F64))))
  full = A2 @ A1
  assert torch.allclose(mt.y.covariance(), full @ x.covariance() @ full.T)
  assert torch.allclose(mt.cross_covariance(), x.covariance() @ full.T)
  assert Chain(lambda v: v + 1, lambda v: 2 * v)(3) == 7            # applied right to left

  def double(v):
    return 2 * v
  T = register_type(double)

  @dispatcher.register(GaussianMoments, T)
  def _mm_double(x, _):
    return GaussianMatch(x=x, y=GaussianMoments((2 * x.mean(), 4 * x.covariance()), True),
                         cross=(2 * x.covariance(), False))
  assert torch.allclose(moment_matching(x, double).y.mean(), 2 * x.mean())
  with pytest.raises(ValueError):
    register_type(double)
  p = partial(Linear, A1)       # a partial whose func is a type -> dispatches on (Moments, partial)
  assert isinstance(p, partial)


@pytest.mark.parametrize("whiten", [True, False])
def test_model_precompute_matches_reference_algebra(whiten):
  """models.SVGP.precompute (torch, what gets packed for the kernels) == beta, C of the restatement."""
  p = random_svgp_params(seed=3, L=2, M=20, d=3, whiten=whiten)
  model = gp_model_from_oracle(p, "cpu")
  Z, ls, var, beta, C, mean_c = model.precompute(torch.device("cpu"))
  b_ref, C_ref = fr.precompute(p)
  assert np.abs(beta.numpy() - b_ref).max() / np.abs(b_ref).max() < 1e-9
  assert np.abs(C.numpy() - C_ref).max() / np.abs(C_ref).max() < 1e-9
  assert np.allclose(Z.numpy(), p.Z) and np.allclose(ls.numpy(), p.lengthscales) and np.allclose(mean_c.numpy(), p.mean_c)


def test_gpr_precompute_and_kernel_slicing():
  rng = np.random.default_rng(0)
  X = rng.uniform(size=(10, 4)); Y = rng.standard_normal((10, 1))
  k = gp.SquaredExponential(variance=0.7, lengthscales=[0.5, 0.8], active_dims=(1, 3))
  gpr = gp.GPR(data=(X, Y), kernel=k, mean_function=gp.Constant([0.2]), noise_variance=0.1)
  Z, ls, var, beta, C, c = gpr.precompute(torch.device("cpu"))
  assert Z.shape == (1, 10, 2) and np.allclose(Z[0].numpy(), X[:, [1, 3]])
  Ky = mo.se_kernel(X[:, [1, 3]], None, np.array([0.5, 0.8]), 0.7) + 0.1 * np.eye(10)
  assert np.allclose(beta[0].numpy(), np.linalg.solve(Ky, Y - 0.2)[:, 0])
  assert np.allclose(C[0].numpy(), -np.linalg.inv(Ky))
  S = torch.eye(4, dtype=F64)[None]
  assert k.slice_cov(S).shape == (1, 2, 2) and k.slice(torch.zeros(1, 4)).shape == (1, 2)


def test_gp_handlers_refuse_cpu_tensors_and_bad_wrappers():
  p = random_svgp_params(seed=3, L=2, M=8, d=3, whiten=True)
  model = gp_model_from_oracle(p, "cpu")
  with pytest.raises(RuntimeError, match="GPU only"):
    moment_matching(_state(), model)
  with pytest.raises(AssertionError):
    moment_matching(_state(), gp.KernelRegressor(model), model_uncertainty=True)
  model.mean_function = object()
  with pytest.raises(NotImplementedError):
    model.precompute(torch.device("cpu"))


def test_moment_matching_euler_python_path():
  """MomentMatchingEuler.step + Euler.__call__ fold on CPU tensors with a linear drift."""
  x0 = _state(B=2, d=3, seed=4)
  A = -0.3 * torch.eye(3, dtype=F64); c = torch.tensor([0.1, 0.0, -0.1], dtype=F64)
  system = dynamics.DynamicalSystem(drift=Linear(A, c), solver=dynamics.MomentMatchingEuler())
  losses = []

  def cb(t, state, acc):
    losses.append(float(t))
    return acc + state[0].sum(-1)
  out = system.solve_forward(initial_time=0.0, initial_state=(x0.mean(), x0.covariance()),
                             solution_times=np.arange(1.0, 4.0), iterator="foldl",
                             callbacks_and_initializers=[(cb, torch.zeros(2, dtype=F64))])
  (m, S), acc = out[0], out[1]
  mu, Sig = x0.mean().numpy(), x0.covariance().numpy()
  tot = np.zeros(2)
  for _ in range(3):
    f1 = mu @ A.numpy().T + c.numpy()
    Sxf = Sig @ A.numpy().T
    mu, Sig = mo.euler_moment_update(mu, Sig, f1, A.numpy() @ Sig @ A.numpy().T, Sxf, 1.0)
    tot += mu.sum(-1)
  assert np.allclose(m.numpy(), mu) and np.allclose(S.numpy(), Sig) and np.allclose(acc.numpy(), tot)
  assert losses == [1.0, 2.0, 3.0]
  hist = system.solve_forward(0.0, (x0.mean(), x0.covariance()), [0.5, 1.0])
  assert len(hist) == 2


def test_cholesky_matches_torch_and_raises_with_evidence_on_non_pd(tmp_path, monkeypatch):
  import torch
  from gpflowpilco_amd.linalg import CholeskyError, cholesky
  g = torch.Generator().manual_seed(0)
  A = torch.randn(3, 6, 6, generator=g, dtype=torch.float64)
  A = A @ A.transpose(1, 2) + 1e-3 * torch.eye(6, dtype=torch.float64)
  assert torch.equal(cholesky(A), torch.linalg.cholesky(A))
  bad = A.clone()
  bad[1] = -bad[1]
  monkeypatch.setenv("GPFLOWPILCO_DUMP_DIR", str(tmp_path))
  with pytest.raises(torch.linalg.LinAlgError) as ei:           # no retry, no fallback: the failure is reported
    cholesky(bad)
  assert isinstance(ei.value, CholeskyError)
  msg = str(ei.value)
  assert "batch item 1 of 3" in msg and "info=1" in msg and "max|A-A^T|" in msg
  dumps = list(tmp_path.glob("cholesky_fail_pid*.pt"))
  assert len(dumps) == 1 and torch.equal(torch.load(dumps[0])["A"], bad)


def test_synthetic_posterior_is_exact_and_pd_by_construction():
  """make_svgp's (q_mu, q_sqrt) is the exact whitened posterior of u ~ N(0, Kuu + jitter) given the targets,
  formed without an eigendecomposition: q_sqrt is lower triangular and reproduces (I + L^T L / s2)^-1."""
  import numpy as np
  from gpflowpilco_amd.synthetic import make_svgp
  syn = make_svgp(2, 40, 3, seed=9)
  for a in range(2):
    A = syn.Z / syn.lengthscales[a]
    d2 = ((A[:, None, :] - A[None, :, :]) ** 2).sum(-1)
    K = syn.variance[a] * np.exp(-0.5 * d2) + 1e-6 * np.eye(40)
    Lk = np.linalg.cholesky(K)
    Am = np.eye(40) + Lk.T @ Lk / syn.noise[a]
    qs = syn.q_sqrt[a]
    assert np.allclose(qs, np.tril(qs))
    assert np.abs(qs @ qs.T @ Am - np.eye(40)).max() < 1e-8
    # unwhitened: cov(u | y) = K - K (K + s2 I)^-1 K
    S = Lk @ qs @ qs.T @ Lk.T
    ref = K - K @ np.linalg.solve(K + syn.noise[a] * np.eye(40), K)
    assert np.abs(S - ref).max() < 1e-8


def test_blocked_cholesky_equals_lapack_and_reports_the_failing_minor():
  """linalg._blocked_cholesky_ex (the composition the GPU set-up uses instead of rocSOLVER's blocked potrf, which
  tools/potrf_probe.py shows to be unsafe beside a second process on the device): same factor as LAPACK, batched,
  and info = position of the first failed leading minor."""
  import torch
  from gpflowpilco_amd.linalg import _blocked_cholesky_ex, _residual
  g = torch.Generator().manual_seed(3)
  X = torch.randn(2, 150, 40, generator=g, dtype=torch.float64)
  A = X @ X.transpose(1, 2) + 1e-3 * torch.eye(150, dtype=torch.float64)
  L, info = _blocked_cholesky_ex(A, nb=32)
  assert int(info.abs().max()) == 0 and torch.allclose(L, torch.linalg.cholesky(A), rtol=1e-9, atol=1e-10)
  assert _residual(A[0], L[0]) < 1e-12 and not _residual(A[0], L[0] * 1.001) < 1e-9
  B = A.clone()
  B[1, 70, 70] = -1.0                                         # minor 71 of item 1 is the first non-positive one
  _, info = _blocked_cholesky_ex(B, nb=32)
  ref = torch.linalg.cholesky_ex(B)[1]
  assert info.tolist() == ref.tolist() == [0, 71]


def test_workspace_generation_counts_every_request():
  """ops.PackedModel.workspace_generation: the match backward reuses the forward's q stage only while nobody has asked for
  that workspace since (no GPU needed: the size query and the bookkeeping are host code)."""
  from gpflowpilco_amd import ops
  pm = ops.PackedModel(L=2, M=16, d=3, dtype=torch.float64, with_C=True, buf=torch.zeros(8, dtype=torch.uint8))
  fl = ops.make_flags(True, True)
  assert pm.workspace_generation(4, fl) == 0
  ws = pm.workspace(4, fl)
  assert pm.workspace_generation(4, fl) == 1 and ws.numel() > 0
  assert pm.workspace(4, fl, peek=True) is ws and pm.workspace_generation(4, fl) == 1          # looking does not count
  assert pm.workspace(4, fl | 4) is ws and pm.workspace_generation(4, fl) == 2                  # FORCE_GENERIC: same workspace
  pm.workspace(5, fl)
  assert pm.workspace_generation(4, fl) == 2 and pm.workspace_generation(5, fl) == 1            # per batch size
  assert pm.workspace_generation(4, ops.make_flags(False, True)) == 0                           # per pair layout


def test_partially_symmetric_contraction_of_the_degree_5_6_moments():
  """The algorithm of csrc/mm_moments6.hip:k_spoly56 restated in numpy (tools/spoly56_proto.py): <N_n, G^{(x)n} Q_n> from PACKED
  symmetric moments -- G applied one index at a time with the colex rank arithmetic and the insertion tables the kernel reads,
  sequentially and meeting in the middle (three indices from the column side, n - 3 from the row side, both multinomials) --
  against the double sum  sum_ij w_i w'_j (zc_i^T G zc'_j)^n.  (The kernel itself is checked end to end on the GPU; this pins the
  index tables and the identity it relies on.)"""
  import importlib.util, os
  import numpy as np
  spec = importlib.util.spec_from_file_location("spoly56_proto", os.path.join(os.path.dirname(__file__), "..", "tools", "spoly56_proto.py"))
  sp = importlib.util.module_from_spec(spec); spec.loader.exec_module(sp)
  rng = np.random.default_rng(7)
  for d in (8, 3):
    ins, last, mult = sp.build_tables(d)
    M = 25
    zc, zc2 = rng.standard_normal((M, d)), rng.standard_normal((M, d))
    w, w2 = rng.standard_normal(M), rng.standard_normal(M)
    G = rng.standard_normal((d, d)) * 0.3
    bij = zc @ G @ zc2.T
    for n in (5, 6):
      ts = sp.tuples(n, d)
      mono = lambda z: np.stack([np.prod(z[:, list(t)], axis=1) for t in ts], axis=1)
      Nn, Qn = w @ mono(zc), w2 @ mono(zc2)
      ref = float(w @ bij ** n @ w2)
      assert abs(sp.contract(Nn, Qn, G, n, d, ins, last, mult) - ref) <= 1e-9 * max(abs(ref), 1.0)
      assert abs(sp.contract_mitm(Nn, Qn, G, n, d, ins, last) - ref) <= 1e-9 * max(abs(ref), 1.0)
