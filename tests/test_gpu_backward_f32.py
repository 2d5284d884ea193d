"""Backward of one moment match on an f32 pack (row f-1; csrc/mm_bwd_f32.hip): the off-diagonal pairs' aggregates
(f64 moments for 1 + b + b^2/2, bf16-MFMA tile sweep for the remainder) against their definition in torch float64, and the
whole vector-Jacobian product against the f64 pack of the same model on the same (f32-rounded) state."""
import numpy as np
import pytest
import torch

from gpflowpilco_amd import _lib, autodiff, ops
from gpflowpilco_amd.synthetic import make_inputs, make_svgp
from tests.helpers import to_dev

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _aggregates_reference(Z, ls, var, beta, mu, S):
  """sum_ij Omega_ij (1 | zeta_i | zeta_i zeta_i^T | zeta'_j | zeta'_j zeta'_j^T | zeta_i zeta'_j^T) per off-diagonal pair,
  from the definitions (materialised [B, Po, M, M] blocks, float64)."""
  L, M, d = Z.shape
  ia, ib = autodiff.pair_indices(L, True, Z.device)
  Pa, lognorm, G, Dr, Dc, const = autodiff.small_algebra(S, ls * ls, var, ia, ib)
  zeta = Z[None] - mu[:, None, None, :]
  q = torch.exp(lognorm[..., None] - 0.5 * torch.einsum('blmi,blij,blmj->blm', zeta, Pa, zeta))
  w = beta[None] * q
  io, jo = ia[L:], ib[L:]
  zr, zc = zeta[:, io], zeta[:, jo]
  delta = (const[:, L:, None, None] - 0.5 * torch.einsum('bpmi,bpij,bpmj->bpm', zr, Dr[:, L:], zr)[..., :, None]
           - 0.5 * torch.einsum('bpmi,bpij,bpmj->bpm', zc, Dc[:, L:], zc)[..., None, :]
           + torch.einsum('bpmi,bpij,bpnj->bpmn', zr, G[:, L:], zc))
  Om = w[:, io][..., :, None] * w[:, jo][..., None, :] * torch.exp(delta)
  B, Po = Om.shape[:2]
  return torch.cat([Om.sum((2, 3))[..., None], torch.einsum('bpij,bpik->bpk', Om, zr),
                    torch.einsum('bpij,bpik,bpil->bpkl', Om, zr, zr).reshape(B, Po, -1), torch.einsum('bpij,bpjk->bpk', Om, zc),
                    torch.einsum('bpij,bpjk,bpjl->bpkl', Om, zc, zc).reshape(B, Po, -1),
                    torch.einsum('bpij,bpik,bpjl->bpkl', Om, zr, zc).reshape(B, Po, -1)], -1)


def _pair_aggregates(pm, mu, S, flags):
  B = mu.shape[0]
  Po = pm.L * (pm.L - 1) // 2
  nT = 1 + 2 * pm.d + 3 * pm.d * pm.d
  ws = pm.workspace(B, flags)
  n = _lib.lib().mm_backward_pair_aggregates_bytes(B, pm.L, pm.M, pm.d, flags)
  scratch = torch.empty(n, dtype=torch.uint8, device=mu.device)
  out = torch.zeros(B, Po, nT, dtype=F64, device=mu.device)
  rc = _lib.lib().mm_backward_pair_aggregates(pm.buf.data_ptr(), pm.nbytes, pm.L, pm.M, pm.d, _lib.MM_F32, B, mu.data_ptr(),
                                               S.data_ptr(), flags, ws.data_ptr(), ws.numel(), scratch.data_ptr(), n,
                                               out.data_ptr(), out.numel() * 8, pm.status().data_ptr(), ops._stream(mu.device))
  _lib.check(rc, "mm_backward_pair_aggregates")
  return out


@pytest.mark.parametrize("shape,scale", [((3, 200, 8, 3), 0.1), ((3, 300, 5, 2), 0.2), ((2, 520, 8, 2), 0.05), ((4, 130, 3, 2), 0.3)],
                         ids=["L3d8", "L3d5", "L2M520", "L4d3wide"])
def test_pair_aggregates_match_their_definition(shape, scale, device):
  L, M, d, B = shape
  syn = make_svgp(L, M, d, seed=50 + L + d, device=str(device), ls_bounds=(0.5, 2.5))
  model = syn.to_model(device)
  pm = model.packed(torch.float32, True, device)
  mu, S = make_inputs(B, d, seed=3, scale=scale, lo=0.2, hi=0.8)
  mu32, S32 = to_dev(mu, device, torch.float32), to_dev(S, device, torch.float32)
  got = _pair_aggregates(pm, mu32, S32, ops.make_flags(True, True))
  Z, ls, var, beta, _, _ = model.precompute(device)          # what mm_pack_model received (float64)
  want = _aggregates_reference(Z, ls, var, beta, mu32.to(F64), S32.to(F64))
  # blocks: N0 | r1 | R2 | k1 | K2 | XC -- each against its own scale (the sums cancel: |Omega| >> |sum Omega|)
  o = np.cumsum([0, 1, d, d * d, d, d * d, d * d])
  for k in range(6):
    g, w_ = got[..., o[k]:o[k + 1]], want[..., o[k]:o[k + 1]]
    sc = float(w_.abs().amax())
    assert float((g - w_).abs().amax()) < 2e-5 * sc, (k, float((g - w_).abs().amax()), sc)


@pytest.mark.parametrize("shape,full,unc", [((4, 300, 8, 3), True, True), ((3, 200, 5, 2), True, False), ((2, 140, 8, 2), False, True),
                                            ((1, 260, 6, 2), True, True), ((5, 530, 7, 2), True, True)],
                         ids=["L4d8", "L3d5nounc", "L2diagcov", "L1", "L5M530"])
def test_f32_pack_backward_matches_f64_pack(shape, full, unc, device):
  L, M, d, B = shape
  syn = make_svgp(L, M, d, seed=60 + L + d, device=str(device), ls_bounds=(0.5, 2.5))
  model = syn.to_model(device)
  pm32, pm64 = model.packed(torch.float32, unc, device), model.packed(F64, unc, device)
  assert ops.backward_supported(pm32)
  mu, S = make_inputs(B, d, seed=4, scale=0.15, lo=0.2, hi=0.8)
  mu32, S32 = to_dev(mu, device, torch.float32), to_dev(S, device, torch.float32)
  rng = np.random.default_rng(5)
  g_f1 = to_dev(rng.standard_normal((B, L)), device, F64)
  g_Sff = to_dev(rng.standard_normal((B, L, L) if full else (B, L)), device, F64)
  g_cross = to_dev(rng.standard_normal((B, d, L)), device, F64)
  gmu32, gS32 = ops.moment_match_backward(pm32, mu32, S32, g_f1, g_Sff, g_cross, full, unc)
  gmu64, gS64 = ops.moment_match_backward(pm64, mu32.to(F64), S32.to(F64), g_f1, g_Sff, g_cross, full, unc)
  assert int(pm32.status()[0]) == 0
  for a_, b_ in ((gmu32, gmu64), (gS32, gS64)):
    sc = float(b_.abs().amax())
    assert float((a_ - b_).abs().amax()) < 1e-4 * sc, (float((a_ - b_).abs().amax()), sc)


def test_differentiable_match_of_an_f32_model_uses_its_own_pack(device):
  """autodiff.moment_match_differentiable on float32 inputs: the gradient comes from the f32 pack (no f64 pack is built)
  and agrees with the float64 evaluation of the same functional."""
  L, M, d, B = 3, 260, 8, 2
  syn = make_svgp(L, M, d, seed=71, device=str(device), ls_bounds=(0.6, 2.5))
  model = syn.to_model(device)
  mu, S = make_inputs(B, d, seed=8, scale=0.1, lo=0.3, hi=0.7)
  rng = np.random.default_rng(9)
  A1, A2, A3 = (to_dev(rng.standard_normal(s), device, F64) for s in ((B, L), (B, L, L), (B, d, L)))
  grads = []
  for dt in (torch.float32, F64):
    mu_t = to_dev(mu, device, torch.float32).to(dt).requires_grad_(True)
    S_t = to_dev(S, device, torch.float32).to(dt).requires_grad_(True)
    f1, Sff, cr = autodiff.moment_match_differentiable(model, mu_t, S_t, True, True)
    ((A1.to(dt) * f1).sum() + (A2.to(dt) * Sff).sum() + (A3.to(dt) * cr).sum()).backward()
    grads.append((mu_t.grad.to(F64), S_t.grad.to(F64)))
  for a_, b_ in zip(grads[0], grads[1]):
    sc = float(b_.abs().amax())
    assert float((a_ - b_).abs().amax()) < 2e-4 * sc, (float((a_ - b_).abs().amax()), sc)


def test_backward_reuses_the_forward_q_stage_only_while_it_is_current(device):
  """MomentMatchFunction saves the workspace generation of its forward; the backward skips the q stage
  (MM_WORKSPACE_CURRENT) only if nothing asked for that workspace in between -- another match on the same pack with the same
  batch size in between must not change the gradient."""
  L, M, d, B = 3, 260, 6, 2
  syn = make_svgp(L, M, d, seed=81, device=str(device), ls_bounds=(0.6, 2.5))
  model = syn.to_model(device)
  mu, S = make_inputs(B, d, seed=8, scale=0.1, lo=0.3, hi=0.7)
  mu2, S2 = make_inputs(B, d, seed=9, scale=0.2, lo=0.1, hi=0.9)
  grads = []
  for disturb in (False, True):
    for dt in (torch.float32, F64):
      mu_t = to_dev(mu, device, dt).requires_grad_(True); S_t = to_dev(S, device, dt).requires_grad_(True)
      f1, Sff, cr = autodiff.moment_match_differentiable(model, mu_t, S_t, True, True)
      pm = model.packed(dt, True, device)
      gen = pm.workspace_generation(B, ops.make_flags(True, True))
      if disturb:
        ops.moment_match(pm, to_dev(mu2, device, dt), to_dev(S2, device, dt))
        assert pm.workspace_generation(B, ops.make_flags(True, True)) == gen + 1
      (f1.sum() + Sff.sum() + cr.sum()).backward()
      grads.append((mu_t.grad.to(F64).clone(), S_t.grad.to(F64).clone()))
  for k in (0, 1):            # f32, f64: identical with and without the call in between
    for a_, b_ in zip(grads[k], grads[k + 2]):
      assert float((a_ - b_).abs().amax()) <= 1e-12 * float(b_.abs().amax())


def test_f32_pack_gradient_against_central_differences_of_the_oracle(device):
  """The f32 pack's vector-Jacobian product checked directly against the fp64 CPU oracle (the literal restatement of
  models.py:200-299): directional derivatives of a random linear functional of (f1, Sff, cross) by central differences."""
  from oracle import mm_oracle as mo
  from tests.helpers import gp_model_from_oracle, random_svgp_params
  L, M, d, B = 3, 150, 8, 2
  p = random_svgp_params(seed=91, L=L, M=M, d=d, whiten=True, ls_bounds=(0.8, 2.5), mean=False)
  rng = np.random.default_rng(3)
  mu = rng.uniform(0.3, 0.7, size=(B, d)).astype(np.float32).astype(np.float64)
  S = make_inputs(B, d, seed=4, scale=0.15)[1].astype(np.float32).astype(np.float64)
  A1, A2, A3 = rng.standard_normal((B, L)), rng.standard_normal((B, L, L)), rng.standard_normal((B, d, L))
  A2 = 0.5 * (A2 + A2.transpose(0, 2, 1))

  def loss(mu_, S_):
    f1, Sff, cr = mo.mm_gauss_svgp_mo(mu_, S_, p, True, True, 0.0)
    return (A1 * f1).sum() + (A2 * Sff).sum() + (A3 * cr).sum()
  model = gp_model_from_oracle(p, device)
  pm = model.packed(torch.float32, True, device)
  gmu, gS = ops.moment_match_backward(pm, to_dev(mu, device, torch.float32), to_dev(S, device, torch.float32),
                                      to_dev(A1, device, F64), to_dev(A2, device, F64), to_dev(A3, device, F64))
  gmu, gS = gmu.cpu().numpy(), gS.cpu().numpy()
  eps = 1e-4
  scale = max(np.abs(gmu).max(), np.abs(gS).max())
  for trial in range(4):
    dm = rng.standard_normal((B, d)); dS = rng.standard_normal((B, d, d)); dS = 0.5 * (dS + dS.transpose(0, 2, 1))
    fd = (loss(mu + eps * dm, S + eps * dS) - loss(mu - eps * dm, S - eps * dS)) / (2 * eps)
    an = (gmu * dm).sum() + (gS * dS).sum()
    assert abs(fd - an) < 2e-4 * scale * np.sqrt(dm.size + dS.size), (trial, fd, an, scale)


def test_f32_pack_backward_beyond_the_lds_weight_cache(device):
  """M > 16384: k_bwd_rem_f32 reads the pair's column weights from global memory instead of its LDS copy."""
  L, M, d, B = 2, 16500, 4, 1
  rng = np.random.default_rng(7)
  t = lambda a: torch.as_tensor(a, dtype=F64, device=device)
  Z = rng.uniform(size=(L, M, d)); ls = np.exp(rng.uniform(np.log(0.6), np.log(1.5), size=(L, d)))
  beta = rng.standard_normal((L, M)) / np.sqrt(M)
  pm32 = ops.pack_model(t(Z), t(ls), t(np.ones(L)), t(beta), None, None, dtype=torch.float32)
  pm64 = ops.pack_model(t(Z), t(ls), t(np.ones(L)), t(beta), None, None, dtype=F64)
  mu, S = make_inputs(B, d, seed=4, scale=0.1, lo=0.3, hi=0.7)
  mu32, S32 = to_dev(mu, device, torch.float32), to_dev(S, device, torch.float32)
  g1, g2, g3 = t(rng.standard_normal((B, L))), t(rng.standard_normal((B, L, L))), t(rng.standard_normal((B, d, L)))
  a = ops.moment_match_backward(pm32, mu32, S32, g1, g2, g3, True, False)
  b = ops.moment_match_backward(pm64, mu32.double(), S32.double(), g1, g2, g3, True, False)
  for x, y in zip(a, b):
    sc = float(y.abs().amax())
    assert float((x - y).abs().amax()) < 1e-4 * sc, (float((x - y).abs().amax()), sc)


def _bwd_draws(n):
  rng = np.random.default_rng(20261004)
  out = []
  for i in range(n):
    d = int(rng.choice([1, 2, 3, 4, 6, 7, 8]))
    L = int(rng.integers(1, 6))
    M = int(rng.integers(5, 700)) if d >= 3 else int(rng.integers(5, 60))      # (few points in 1-2 dimensions: Kuu conditioning)
    out.append(dict(seed=3000 + i, L=L, M=M, d=d, B=int(rng.integers(1, 6)), full=bool(rng.integers(0, 2)),
                    unc=bool(rng.integers(0, 2)), scale=float(rng.choice([0.03, 0.1, 0.25]))))
  return out


BWD_DRAWS = _bwd_draws(16)


@pytest.mark.parametrize("c", BWD_DRAWS, ids=[f"{i}-L{c['L']}M{c['M']}d{c['d']}B{c['B']}" for i, c in enumerate(BWD_DRAWS)])
def test_random_shapes_f32_pack_backward_matches_f64_pack(c, device):
  """Seeded sweep over shapes and flags (M and B not multiples of anything, d on both sides of the 32-slot monomial block,
  one latent, diagonal output covariance): the f32 pack's vector-Jacobian product against the f64 pack's."""
  from tests.helpers import gp_model_from_oracle, random_svgp_params
  lo = 0.2 if c["d"] <= 2 else 0.5 * max(1.0, np.sqrt(c["d"] / 4.0))
  p = random_svgp_params(seed=c["seed"], L=c["L"], M=c["M"], d=c["d"], whiten=True, ls_bounds=(lo, 3.0 * lo), mean=True)
  model = gp_model_from_oracle(p, device)
  pm32, pm64 = model.packed(torch.float32, c["unc"], device), model.packed(F64, c["unc"], device)
  rng = np.random.default_rng(c["seed"] + 1)
  B, L, d = c["B"], c["L"], c["d"]
  mu = rng.uniform(0.25, 0.75, size=(B, d))
  S = make_inputs(B, d, seed=c["seed"] + 2, scale=c["scale"] * (0.3 if d <= 2 else 1.0))[1]
  mu32, S32 = to_dev(mu, device, torch.float32), to_dev(S, device, torch.float32)
  g1 = to_dev(rng.standard_normal((B, L)), device, F64)
  g2 = to_dev(rng.standard_normal((B, L, L) if c["full"] else (B, L)), device, F64)
  g3 = to_dev(rng.standard_normal((B, d, L)), device, F64)
  a = ops.moment_match_backward(pm32, mu32, S32, g1, g2, g3, c["full"], c["unc"])
  b = ops.moment_match_backward(pm64, mu32.double(), S32.double(), g1, g2, g3, c["full"], c["unc"])
  assert int(pm32.status()[0]) == 0
  # one tolerance for every draw: where the f32 / bf16 sweep's own error estimate leaves the contract (draws 13 and 15: state std
  # 0.25 at lengthscales 0.5-1.4 with beta ~ 1e4 -- rounds 1-3 accepted 5e-2 there) those items' aggregates are re-reduced in
  # f64 (csrc/mm_route.hip)
  tol = 2e-4
  for x, y in zip(a, b):
    sc = float(y.abs().amax())
    assert float((x - y).abs().amax()) < tol * sc, (c, float((x - y).abs().amax()), sc)
