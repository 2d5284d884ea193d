"""The hand-derived adjoints of csrc/mm_adjoint.h (rows f-1 x f-2), checked on the CPU.

The header is written for an execution context; ``tests/hostcheck/mm_adjoint_host.hip`` compiles the SAME arithmetic for one
host thread (test infrastructure, built by ``__graft_entry__.build()``; the product library has no host path).  Each
adjoint is compared with torch autograd of the torch mirror of the forward it differentiates
(gpflowpilco_amd/moment_matching/{maths,components,bijectors}.py, cost.py, autodiff.moment_match_torch -- the
transliterations of forward_sde.py:95-137, components.py:19-57, maths.py:143-176, bijectors.py:39-69, models.py:200-299).
The device kernels that call the same functions are checked on the GPU in tests/test_gpu_backward.py.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest
import torch

from gpflowpilco_amd import autodiff
from gpflowpilco_amd.components import TrigonometricEncoder
from gpflowpilco_amd.cost import expected_gaussian_cost
from gpflowpilco_amd.moment_matching import GaussianMoments, moment_matching
from gpflowpilco_amd.special import ndtr, owens_t
from gpflowpilco_amd.synthetic import generate_covariance

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F64 = torch.float64
P = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def hc():
  so = os.path.join(ROOT, "tests", "hostcheck", "libmm_adjoint_host.so")
  src = os.path.join(ROOT, "tests", "hostcheck", "mm_adjoint_host.hip")
  hdr = os.path.join(ROOT, "gpflowpilco_amd", "csrc", "mm_adjoint.h")
  if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
    subprocess.run(["bash", os.path.join(ROOT, "tests", "hostcheck", "build.sh")], check=True)
  return C.CDLL(so)


def _p(a):
  return a.ctypes.data_as(P)


def _ip(a):
  return a.ctypes.data_as(C.POINTER(C.c_int32))


def _c(a):
  return np.ascontiguousarray(a, dtype=np.float64)


def _sym(A):
  return 0.5 * (A + np.swapaxes(A, -1, -2))


def _t(a, grad=False):
  return torch.tensor(np.asarray(a), dtype=F64, requires_grad=grad)


def _rel(got, want):
  return float(np.abs(np.asarray(got) - np.asarray(want)).max() / max(np.abs(np.asarray(want)).max(), 1e-300))


@pytest.mark.parametrize("nx,active", [(4, (1,)), (5, (3, 0)), (3, (0, 1, 2))])
def test_encode_adjoint(hc, nx, active):
  rng = np.random.default_rng(nx)
  na = len(active); ne = nx + na
  m = rng.standard_normal(nx); S = generate_covariance(rng, nx, (), 0.4)
  gme = rng.standard_normal(ne); gSee = rng.standard_normal((ne, ne)); gSxe = rng.standard_normal((nx, ne))
  mt, St = _t(m, True), _t(S, True)
  match = moment_matching(GaussianMoments((mt[None], (0.5 * (St + St.T))[None]), centered=True), TrigonometricEncoder(active))
  val = ((match.y.mean()[0] * _t(gme)).sum() + (match.y.covariance()[0] * _t(gSee)).sum()
         + (match.cross_covariance(dense=True)[0] * _t(gSxe)).sum())
  gm_w, gS_w = torch.autograd.grad(val, (mt, St))
  gm = np.zeros(nx); gS = np.zeros((nx, nx))
  act = np.array(active, dtype=np.int32)
  hc.hc_encode_bwd(nx, na, _ip(act), _p(_c(m)), _p(_c(S)), _p(_c(gme)), _p(_c(gSee)), _p(_c(gSxe)), _p(gm), _p(gS))
  assert _rel(gm, gm_w.numpy()) < 1e-12 and _rel(_sym(gS), _sym(gS_w.numpy())) < 1e-12


def test_cost_adjoint(hc):
  rng = np.random.default_rng(3)
  for n, sing in ((5, True), (4, False)):
    mean = rng.standard_normal(n); cov = generate_covariance(rng, n, (), 0.5); target = rng.standard_normal(n)
    W = generate_covariance(rng, n, (), 1.5)
    if sing:
      W[-2:] = 0.0; W[:, -2:] = 0.0                       # the cartpole objective ignores two encoded dims
    mt, ct = _t(mean, True), _t(cov, True)
    cost = expected_gaussian_cost(mt, 0.5 * (ct + ct.T), _t(target), _t(W))
    gm_w, gc_w = torch.autograd.grad(1.7 * cost, (mt, ct))
    gm = np.zeros(n); gc = np.zeros((n, n))
    hc.hc_cost_bwd.restype = C.c_double
    got = hc.hc_cost_bwd(n, _p(_c(mean)), _p(_c(cov)), _p(_c(target)), _p(_c(W)), C.c_double(1.7), _p(gm), _p(gc))
    assert abs(got - float(cost)) < 1e-13
    assert _rel(gm, gm_w.numpy()) < 1e-11 and _rel(_sym(gc), _sym(gc_w.numpy())) < 1e-11


def _step_forward(D, dt, m, S, Sxe, cp, Sdd, df1, dSff, dcross):
  """mmc_step_body (csrc/mm_compose.hip) in torch."""
  nx, na, ne, nd, active, inactive = D
  n2 = 2 * na
  rows = []
  for r in range(nx):
    if r in active:
      rows.append(torch.cat([Sxe[r], (Sxe[r] * cp).sum()[None]]))
    else:
      rows.append(Sdd[n2 + inactive.index(r)])
  Sxd = torch.stack(rows)
  Sxf = Sxd @ dcross
  return m + dt * df1, S + dt * (Sxf + Sxf.T) + dt * dt * dSff


def test_step_adjoint(hc):
  rng = np.random.default_rng(5)
  nx, active = 5, (3, 1)
  na = len(active); ne = nx + na; nd = ne + 1
  inactive = [r for r in range(nx) if r not in active]
  dt = 0.7
  vals = dict(Sxe=rng.standard_normal((nx, ne)), cp=rng.standard_normal(ne), Sdd=rng.standard_normal((nd, nd)),
              df1=rng.standard_normal(nx), dSff=rng.standard_normal((nx, nx)), dcross=rng.standard_normal((nd, nx)))
  ts = {k: _t(v, True) for k, v in vals.items()}
  m, S = _t(rng.standard_normal(nx), True), _t(rng.standard_normal((nx, nx)), True)
  m1, S1 = _step_forward((nx, na, ne, nd, active, inactive), dt, m, S, **ts)
  gm1, gS1 = rng.standard_normal(nx), rng.standard_normal((nx, nx))
  names = ["Sxe", "cp", "Sdd", "df1", "dSff", "dcross"]
  want = torch.autograd.grad((m1 * _t(gm1)).sum() + (S1 * _t(gS1)).sum(), [ts[k] for k in names] + [m, S])
  out = {k: np.zeros_like(vals[k]) for k in names}
  act = np.array(active, dtype=np.int32)
  hc.hc_step_bwd(nx, na, _ip(act), C.c_double(dt), _p(_c(vals["Sxe"])), _p(_c(vals["cp"])), _p(_c(vals["Sdd"])),
                 _p(_c(vals["dcross"])), _p(_c(gm1)), _p(_c(gS1)), _p(out["Sxe"]), _p(out["cp"]), _p(out["Sdd"]),
                 _p(out["df1"]), _p(out["dSff"]), _p(out["dcross"]))
  for k, w in zip(names, want):
    assert _rel(out[k], w.numpy()) < 1e-13, k
  assert _rel(gm1, want[-2].numpy()) < 1e-15 and _rel(gS1, want[-1].numpy()) < 1e-15     # the direct terms


def _head_forward(scale, shift, pf1, pSff, pcross, See):
  """k_compose_policy (csrc/mm_compose.hip) in torch."""
  vx = torch.clamp(pSff, min=0.0)
  isq = torch.rsqrt(vx + 1.0); z = isq * pf1
  y1 = ndtr(z)
  y2 = y1 - 2.0 * owens_t(z, torch.rsqrt(1.0 + 2.0 * vx))
  head_pre = isq * (2.0 * math.pi) ** -0.5 * torch.exp(-0.5 * z * z) * scale
  cp = pcross * head_pre
  Seu = See @ cp
  md_last = scale * (y1 + shift)
  Suu = scale * scale * (y2 - y1 * y1)
  return cp, Seu, md_last, Suu


def test_head_adjoint(hc):
  rng = np.random.default_rng(7)
  ne = 5; nd = ne + 1
  for pf1v, pSffv in ((0.3, 0.6), (-1.2, 0.05), (0.0, 2.0)):
    pf1, pSff = _t(pf1v, True), _t(pSffv, True)
    pcross = _t(rng.standard_normal(ne), True); See = _t(generate_covariance(rng, ne, (), 0.3), True)
    me = _t(rng.standard_normal(ne), True)
    scale, shift = 2.0, -0.5
    cp, Seu, mu_u, Suu = _head_forward(scale, shift, pf1, pSff, pcross, See)
    md = torch.cat([me, mu_u[None]])
    Sdd = torch.cat([torch.cat([See, Seu[:, None]], 1), torch.cat([Seu[None], Suu[None, None]], 1)], 0)
    gmd, gSdd, gcp = rng.standard_normal(nd), rng.standard_normal((nd, nd)), rng.standard_normal(ne)
    want = torch.autograd.grad((md * _t(gmd)).sum() + (Sdd * _t(gSdd)).sum() + (cp * _t(gcp)).sum(), (me, See, pcross, pf1, pSff))
    gme = np.zeros(ne); gSee = np.zeros((ne, ne)); gpc = np.zeros(ne); gp2 = np.zeros(2)
    hc.hc_head_bwd(ne, C.c_double(scale), C.c_double(shift), C.c_double(pf1v), C.c_double(pSffv), _p(_c(pcross.detach().numpy())),
                   _p(_c(See.detach().numpy())), _p(_c(gmd)), _p(_c(gSdd)), _p(_c(gcp)), _p(gme), _p(gSee), _p(gpc), _p(gp2))
    assert _rel(gme, want[0].numpy()) < 1e-13 and _rel(gSee, want[1].numpy()) < 1e-12
    assert _rel(gpc, want[2].numpy()) < 1e-12
    assert abs(gp2[0] - float(want[3])) < 1e-9 * max(1.0, abs(float(want[3])))       # Owen's T: 48-point quadrature vs closed form
    assert abs(gp2[1] - float(want[4])) < 1e-9 * max(1.0, abs(float(want[4])))


@pytest.mark.parametrize("M,d", [(30, 5), (7, 2), (40, 8)])
def test_policy_match_adjoint_inputs_and_parameters(hc, M, d):
  rng = np.random.default_rng(M + d)
  Z = rng.uniform(size=(M, d)); ls = np.exp(rng.uniform(np.log(0.6), np.log(2.0), size=d)); var = 0.8
  beta = rng.standard_normal(M); meanc = 0.3
  mu = rng.uniform(0.2, 0.8, size=d); Sigma = generate_covariance(rng, d, (), 0.15)
  gf1, gSff, gcross = 0.7, -1.3, rng.standard_normal(d)
  Zt, lst, vart, bt, mct = _t(Z[None], True), _t(ls[None], True), _t([var], True), _t(beta[None], True), _t([meanc], True)
  mut, St = _t(mu[None], True), _t(Sigma, True)
  f1, Sff, cross = autodiff.moment_match_torch(mut, (0.5 * (St + St.T))[None], Zt, lst, vart, bt, None, mct, True, False)
  val = gf1 * f1.sum() + gSff * Sff.sum() + (cross[0, :, 0] * _t(gcross)).sum()
  want = torch.autograd.grad(val, (mut, St, Zt, bt, lst, vart, mct))
  gmu = np.zeros(d); gS = np.zeros((d, d)); gpar = np.zeros(M * d + M + d + 2)
  rc = hc.hc_policy_small_bwd(M, d, _p(_c(Z)), _p(_c(beta)), _p(_c(ls * ls)), C.c_double(var), _p(_c(mu)), _p(_c(Sigma)),
                              C.c_double(gf1), C.c_double(gSff), _p(_c(gcross)), _p(gmu), _p(gS), _p(gpar))
  assert rc == 0
  tol = 2e-10
  assert _rel(gmu, want[0].numpy()[0]) < tol and _rel(gS, _sym(want[1].numpy())) < tol
  assert _rel(gpar[:M * d].reshape(M, d), want[2].numpy()[0]) < tol
  assert _rel(gpar[M * d:M * d + M], want[3].numpy()[0]) < tol
  assert _rel(2.0 * ls * gpar[M * d + M:M * d + M + d], want[4].numpy()[0]) < tol        # d/d ls = 2 ls d/d ls^2
  assert abs(gpar[-2] - float(want[5])) < tol * max(1.0, abs(float(want[5])))
  assert abs(gpar[-1] - float(want[6])) < tol * max(1.0, abs(float(want[6])))


@pytest.mark.parametrize("L,M,d,full,unc", [(3, 20, 4, True, True), (2, 33, 6, True, False), (3, 12, 3, False, True), (1, 9, 2, True, True)])
def test_gp_match_adjoint_from_the_M_sized_sums(hc, L, M, d, full, unc):
  """mma_gp_item_bwd consumes the sums mm_backward_sums forms on the GPU; here they are formed from their definitions
  (csrc/mm_backward.hip header) in torch, and the result is compared with autograd of the materialised evaluation."""
  rng = np.random.default_rng(10 * L + d)
  Z = rng.uniform(size=(L, M, d)); ls = np.exp(rng.uniform(np.log(0.6), np.log(2.0), size=(L, d))); var = 0.7 + 0.3 * rng.uniform(size=L)
  beta = rng.standard_normal((L, M)); Cm = _sym(0.2 * rng.standard_normal((L, M, M)))
  mu = rng.uniform(0.2, 0.8, size=(1, d)); Sigma = generate_covariance(rng, d, (1,), 0.15)
  Pn = L * (L + 1) // 2 if full else L
  g_f1 = rng.standard_normal((1, L)); g_Sff = rng.standard_normal((1, L, L) if full else (1, L)); g_cross = rng.standard_normal((1, d, L))
  Zt, lst, vart, bt, Ct = _t(Z), _t(ls), _t(var), _t(beta), _t(Cm)
  mut, St = _t(mu, True), _t(Sigma, True)
  f1, Sff, cross = autodiff.moment_match_torch(mut, 0.5 * (St + St.transpose(1, 2)), Zt, lst, vart, bt, Ct if unc else None, None, full, unc)
  val = (f1 * _t(g_f1)).sum() + (Sff * _t(g_Sff)).sum() + (cross * _t(g_cross)).sum()
  gmu_w, gS_w = torch.autograd.grad(val, (mut, St))
  # ---- the sums, from their definitions -------------------------------------------------------------------------------
  with torch.no_grad():
    ia, ib = autodiff.pair_indices(L, full)
    Pa, lognorm, G, Dr, Dc, const = autodiff.small_algebra(St, lst * lst, vart, ia, ib)
    zeta = Zt[None] - mut[:, None, None, :]
    q = torch.exp(lognorm[..., None] - 0.5 * torch.einsum('blmi,blij,blmj->blm', zeta, Pa, zeta))
    w = bt[None] * q
    zr, zc = zeta[:, ia], zeta[:, ib]
    delta = (const[..., None, None] - 0.5 * torch.einsum('bpmi,bpij,bpmj->bpm', zr, Dr, zr)[..., :, None]
             - 0.5 * torch.einsum('bpmi,bpij,bpmj->bpm', zc, Dc, zc)[..., None, :] + torch.einsum('bpmi,bpij,bpnj->bpmn', zr, G, zc))
    E = torch.expm1(delta); e = E + 1.0
    Om = w[:, ia][..., :, None] * w[:, ib][..., None, :] * e
    Cq = torch.zeros_like(e)
    if unc:
      Cq[:, :L] = Ct[None] * q[..., :, None] * e[:, :L]              # C_ij q_i e_ij
      Om[:, :L] = Om[:, :L] + Cq[:, :L] * q[..., None, :]
    Mp = M + 3                                                       # any padding >= M
    col = np.zeros((Pn, 3 + d, Mp)); row = np.zeros((max(Pn - L, 1), 2, Mp))
    col[:, 0, :M] = Om.sum(2)[0].numpy(); col[:, 1, :M] = (w[:, ia][..., :, None] * E).sum(2)[0].numpy(); col[:, 2, :M] = Cq.sum(2)[0].numpy()
    col[:, 3:, :M] = torch.einsum('bpij,bpid->bpdj', Om, zr)[0].numpy()
    if Pn > L:
      row[:, 0, :M] = Om.sum(3)[0, L:].numpy(); row[:, 1, :M] = (E * w[:, ib][..., None, :]).sum(3)[0, L:].numpy()
    latmat = np.zeros((L, 2 * d * d + 2)); latmat[:, :d * d] = Pa[0].reshape(L, d * d).numpy()
    wp = np.zeros((L, Mp)); qp = np.zeros((L, Mp)); wp[:, :M] = w[0].numpy(); qp[:, :M] = q[0].numpy()
  gmu = np.zeros(d); gS = np.zeros((d, d))
  rc = hc.hc_gp_bwd(L, M, Mp, d, int(full), int(unc), _p(_c(Z)), _p(_c(ls * ls)), _p(_c(mu[0])), _p(_c(Sigma[0])), _p(_c(latmat)),
                    _p(_c(wp)), _p(_c(qp)), _p(_c(col)), _p(_c(row)), _p(_c(g_f1[0])), _p(_c(g_Sff[0])), _p(_c(g_cross[0])),
                    _p(gmu), _p(gS))
  assert rc == 0
  assert _rel(gmu, gmu_w.numpy()[0]) < 1e-10 and _rel(gS, _sym(gS_w.numpy()[0])) < 1e-10
  # the same through partial moment sums over chunks of 7 centres (k_gp_bwd_moments on the device)
  gmu2 = np.zeros(d); gS2 = np.zeros((d, d))
  rc = hc.hc_gp_bwd_chunked(L, M, Mp, d, int(full), int(unc), _p(_c(Z)), _p(_c(ls * ls)), _p(_c(mu[0])), _p(_c(Sigma[0])), _p(_c(latmat)),
                            _p(_c(wp)), _p(_c(qp)), _p(_c(col)), _p(_c(row)), _p(_c(g_f1[0])), _p(_c(g_Sff[0])), _p(_c(g_cross[0])),
                            7, _p(gmu2), _p(gS2))
  assert rc == 0 and _rel(gmu2, gmu) < 1e-12 and _rel(gS2, gS) < 1e-12


@pytest.mark.parametrize("L,M,d,unc", [(3, 20, 4, True), (2, 33, 6, False), (4, 15, 3, True)])
def test_gp_match_adjoint_from_pair_aggregates(hc, L, M, d, unc):
  """The f32-model backward hands the off-diagonal pairs to mma_gp_item_bwd as AGGREGATES
  sum_ij Omega_ij (1 | zeta_i | zeta_i zeta_i^T | zeta'_j | zeta'_j zeta'_j^T | zeta_i zeta'_j^T) instead of M-sized sums
  (csrc/mm_bwd_f32.hip): formed here from their definition, result against autograd of the materialised evaluation."""
  rng = np.random.default_rng(100 * L + d)
  Z = rng.uniform(size=(L, M, d)); ls = np.exp(rng.uniform(np.log(0.6), np.log(2.0), size=(L, d))); var = 0.7 + 0.3 * rng.uniform(size=L)
  beta = rng.standard_normal((L, M)); Cm = _sym(0.2 * rng.standard_normal((L, M, M)))
  mu = rng.uniform(0.2, 0.8, size=(1, d)); Sigma = generate_covariance(rng, d, (1,), 0.15)
  Pn = L * (L + 1) // 2
  g_f1 = rng.standard_normal((1, L)); g_Sff = rng.standard_normal((1, L, L)); g_cross = rng.standard_normal((1, d, L))
  Zt, lst, vart, bt, Ct = _t(Z), _t(ls), _t(var), _t(beta), _t(Cm)
  mut, St = _t(mu, True), _t(Sigma, True)
  f1, Sff, cross = autodiff.moment_match_torch(mut, 0.5 * (St + St.transpose(1, 2)), Zt, lst, vart, bt, Ct if unc else None, None, True, unc)
  val = (f1 * _t(g_f1)).sum() + (Sff * _t(g_Sff)).sum() + (cross * _t(g_cross)).sum()
  gmu_w, gS_w = torch.autograd.grad(val, (mut, St))
  with torch.no_grad():
    ia, ib = autodiff.pair_indices(L, True)
    Pa, lognorm, G, Dr, Dc, const = autodiff.small_algebra(St, lst * lst, vart, ia, ib)
    zeta = Zt[None] - mut[:, None, None, :]
    q = torch.exp(lognorm[..., None] - 0.5 * torch.einsum('blmi,blij,blmj->blm', zeta, Pa, zeta))
    w = bt[None] * q
    zr, zc = zeta[:, ia], zeta[:, ib]
    delta = (const[..., None, None] - 0.5 * torch.einsum('bpmi,bpij,bpmj->bpm', zr, Dr, zr)[..., :, None]
             - 0.5 * torch.einsum('bpmi,bpij,bpmj->bpm', zc, Dc, zc)[..., None, :] + torch.einsum('bpmi,bpij,bpnj->bpmn', zr, G, zc))
    E = torch.expm1(delta); e = E + 1.0
    Om = w[:, ia][..., :, None] * w[:, ib][..., None, :] * e
    Cq = torch.zeros_like(e)
    if unc:
      Cq[:, :L] = Ct[None] * q[..., :, None] * e[:, :L]
      Om[:, :L] = Om[:, :L] + Cq[:, :L] * q[..., None, :]
    Mp = M + 5
    col = np.zeros((L, 3 + d, Mp))                                   # the diagonal pairs only
    col[:, 0, :M] = Om.sum(2)[0, :L].numpy(); col[:, 1, :M] = (w[:, ia][..., :, None] * E).sum(2)[0, :L].numpy()
    col[:, 2, :M] = Cq.sum(2)[0, :L].numpy(); col[:, 3:, :M] = torch.einsum('bpij,bpid->bpdj', Om, zr)[0, :L].numpy()
    Oo, zro, zco = Om[0, L:], zr[0, L:], zc[0, L:]
    agg = torch.cat([Oo.sum((1, 2))[:, None], torch.einsum('pij,pik->pk', Oo, zro),
                     torch.einsum('pij,pik,pil->pkl', Oo, zro, zro).reshape(Pn - L, -1), torch.einsum('pij,pjk->pk', Oo, zco),
                     torch.einsum('pij,pjk,pjl->pkl', Oo, zco, zco).reshape(Pn - L, -1),
                     torch.einsum('pij,pik,pjl->pkl', Oo, zro, zco).reshape(Pn - L, -1)], 1).numpy()
    assert agg.shape[1] == 1 + 2 * d + 3 * d * d
    latmat = np.zeros((L, 2 * d * d + 2)); latmat[:, :d * d] = Pa[0].reshape(L, d * d).numpy()
    wp = np.zeros((L, Mp)); qp = np.zeros((L, Mp)); wp[:, :M] = w[0].numpy(); qp[:, :M] = q[0].numpy()
    f1raw = w[0].sum(1).numpy()
  for chunk in (0, 6):                                               # 6: partial moment sums over chunks of centres
    gmu = np.zeros(d); gS = np.zeros((d, d))
    rc = hc.hc_gp_bwd_agg(L, M, Mp, d, 1, int(unc), _p(_c(Z)), _p(_c(ls * ls)), _p(_c(mu[0])), _p(_c(Sigma[0])), _p(_c(latmat)),
                          _p(_c(wp)), _p(_c(qp)), _p(_c(col)), _p(_c(agg)), _p(_c(f1raw)), _p(_c(g_f1[0])), _p(_c(g_Sff[0])),
                          _p(_c(g_cross[0])), chunk, _p(gmu), _p(gS))
    assert rc == 0
    assert _rel(gmu, gmu_w.numpy()[0]) < 1e-10 and _rel(gS, _sym(gS_w.numpy()[0])) < 1e-10


def _packed_moments(hc, wt, zc, deg=4):
  """sum_m wt[m] zc[m]^alpha for every monomial of degree <= deg, graded colex (csrc/mm_mono.h)."""
  import itertools
  d = zc.shape[1]
  out = np.zeros(hc.hc_mono_off(deg + 1, d))
  for n in range(deg + 1):
    for ks in itertools.combinations_with_replacement(range(d), n):
      kk = np.array(ks, dtype=np.int32)
      r = hc.hc_mono_rank(kk.ctypes.data_as(C.POINTER(C.c_int)), n) if n else 0
      out[hc.hc_mono_off(n, d) + r] = (wt * np.prod(zc[:, list(ks)], axis=1)).sum() if n else wt.sum()
  return out


@pytest.mark.parametrize("d,M", [(3, 17), (8, 12), (5, 9)])
def test_pair_polynomial_part_and_recentring_equal_brute_force(hc, d, M):
  """mma_pair_poly: sum_ij what_i what'_j (1 + b + b^2/2) zeta_i^alpha zc'_j^beta, |alpha| + |beta| <= 2, from the packed
  degree-4 moments of the two weight vectors (b = zeta_i^T G zc'_j, zeta = zc - dmu); mma_pair_convert: zc' -> zeta'."""
  rng = np.random.default_rng(d * 31 + M)
  zc = 0.6 * rng.standard_normal((M, d)); zc2 = 0.6 * rng.standard_normal((M + 3, d))
  wr = rng.standard_normal(M); wc = rng.standard_normal(M + 3)
  G = 0.4 * rng.standard_normal((d, d)); dmu = 0.3 * rng.standard_normal(d); dmu2 = 0.3 * rng.standard_normal(d)
  mR = _packed_moments(hc, wr, zc); mC = _packed_moments(hc, wc, zc2)
  nT = 1 + 2 * d + 3 * d * d
  T = np.zeros(nT)
  hc.hc_pair_poly(d, _p(_c(G)), _p(_c(dmu)), _p(_c(mR)), _p(_c(mC)), _p(T))
  zeta = zc - dmu
  b = zeta @ G @ zc2.T
  Om = wr[:, None] * wc[None, :] * (1.0 + b + 0.5 * b * b)

  def brute(colside):
    return np.concatenate([[Om.sum()], np.einsum('ij,ik->k', Om, zeta), np.einsum('ij,ik,il->kl', Om, zeta, zeta).ravel(),
                           np.einsum('ij,jk->k', Om, colside), np.einsum('ij,jk,jl->kl', Om, colside, colside).ravel(),
                           np.einsum('ij,ik,jl->kl', Om, zeta, colside).ravel()])
  assert _rel(T, brute(zc2)) < 1e-12
  hc.hc_pair_convert(d, _p(_c(dmu2)), _p(T))
  assert _rel(T, brute(zc2 - dmu2)) < 1e-12


@pytest.mark.parametrize("d,M", [(3, 17), (8, 12)])
def test_pair_polynomial_part_with_rows_recentred_at_the_centroid(hc, d, M):
  """Round 5: the bilinear form of an off-diagonal item runs on rows centred at the latent's centroid, b_ij = zc_i^T G zc'_j (the
  shift's column factor e^{-dmu^T G zc'_j} sits in the column weight), while the aggregates keep the row monomials of
  zeta_i = zc_i - dmu: mma_pair_poly with a zero shift, then mma_pair_convert_rows, equals the brute-force sums."""
  rng = np.random.default_rng(d * 17 + M)
  zc = 0.6 * rng.standard_normal((M, d)); zc2 = 0.6 * rng.standard_normal((M + 3, d))
  wr = rng.standard_normal(M); wc = rng.standard_normal(M + 3)
  G = 0.4 * rng.standard_normal((d, d)); dmu = 0.3 * rng.standard_normal(d); dmu2 = 0.3 * rng.standard_normal(d)
  mR = _packed_moments(hc, wr, zc); mC = _packed_moments(hc, wc, zc2)
  T = np.zeros(1 + 2 * d + 3 * d * d)
  hc.hc_pair_poly(d, _p(_c(G)), _p(_c(np.zeros(d))), _p(_c(mR)), _p(_c(mC)), _p(T))
  hc.hc_pair_convert_rows(d, _p(_c(dmu)), _p(T))
  zeta = zc - dmu
  b = zc @ G @ zc2.T                                       # rows centred at the centroid inside the exponent
  Om = wr[:, None] * wc[None, :] * (1.0 + b + 0.5 * b * b)

  def brute(colside):
    return np.concatenate([[Om.sum()], np.einsum('ij,ik->k', Om, zeta), np.einsum('ij,ik,il->kl', Om, zeta, zeta).ravel(),
                           np.einsum('ij,jk->k', Om, colside), np.einsum('ij,jk,jl->kl', Om, colside, colside).ravel(),
                           np.einsum('ij,ik,jl->kl', Om, zeta, colside).ravel()])
  assert _rel(T, brute(zc2)) < 1e-12
  hc.hc_pair_convert(d, _p(_c(dmu2)), _p(T))
  assert _rel(T, brute(zc2 - dmu2)) < 1e-12
