"""Backward of the GP moment match w.r.t. the input moments.

The reference differentiates the whole rollout with ``tf.GradientTape``
(``gpflow_pilco/utils/optimizers.py:52-56``).  During a policy update the drift model is
frozen, so what the rollout needs from the big GP is d(f1, Sff, cross)/d(mu, Sigma).

Split of work:
  * M x M part -- HIP (``csrc/mm_backward.hip``, ``mm_backward_sums``): with
    Omega_ij = (w_i w'_j + [a=a'] C_ij q_i q_j) exp(delta_ij) and E = expm1(delta) the kernel
    returns the M-sized sums Ksum_j, csum_j, cC_j, Usum_j (columns) and Rsum_i, rsum_i (rows).
  * everything else -- torch autograd on a *surrogate*: a scalar whose gradient w.r.t.
    (mu, Sigma) equals the true one because the M^2-derived coefficients enter it as detached
    constants:  dSff_p = sum_i r_i dw_i + sum_j c_j dw'_j + sum_ij Omega_ij d(delta_ij)  (+ C term)
    and  d(delta_ij) = d const + d rho_i + d gamma_j + d(zeta_i^T G zeta'_j).

The sums are taken in f64 (an f32 model's backward runs on the f64 pack of the same model); the
sweep is a plain VALU kernel (correctness first).
``MomentMatchFunction`` makes ``ops.moment_match`` differentiable w.r.t. (mu, Sigma).
"""
from __future__ import annotations

import os
import weakref
from typing import Optional,  Tuple

import torch

from . import _lib, ops
from ._lib import MM_FULL_OUTPUT_COV, MM_MODEL_UNCERTAINTY, check, lib


_PAIR_CACHE = {}


def pair_indices(L: int, full: bool = True, device="cpu") -> Tuple[torch.Tensor, torch.Tensor]:
  """Kernel pair order: the L diagonal pairs first, then a < a' row by row (cached per device)."""
  key = (int(L), bool(full), str(device))
  if key in _PAIR_CACHE:
    return _PAIR_CACHE[key]
  ia, ib = list(range(L)), list(range(L))
  if full:
    for a in range(L):
      for b in range(a + 1, L):
        ia.append(a); ib.append(b)
  _PAIR_CACHE[key] = (torch.tensor(ia, device=device), torch.tensor(ib, device=device))
  return _PAIR_CACHE[key]


def _sym(A):
  return 0.5 * (A + A.transpose(-1, -2))


def small_algebra(Sigma: torch.Tensor, ls2: torch.Tensor, var: torch.Tensor, ia, ib):
  """The per-(b, latent) and per-(b, pair) d x d quantities of ``k_prep`` as differentiable torch ops.
  (``inv_ex`` / ``solve_ex``: the checked variants synchronise the stream once per call; a non-PD state
  is reported by the kernels' status word, here it propagates as inf / nan.)"""
  B, d, _ = Sigma.shape
  eye = torch.eye(d, dtype=Sigma.dtype, device=Sigma.device)
  SL = Sigma[:, None] + ls2[None, :, :, None] * eye                       # [B,L,d,d]
  Pa = torch.linalg.inv_ex(SL).inverse
  ld = torch.linalg.slogdet(SL)[1]                                         # [B,L]
  lognorm = torch.log(var)[None] + 0.5 * torch.log(ls2).sum(-1)[None] - 0.5 * ld
  La, Lb = ls2[ia], ls2[ib]                                                # [P,d]
  V = La * Lb / (La + Lb)
  Sv = Sigma[:, None] + V[None, :, :, None] * eye                         # [B,P,d,d]
  T = _sym(V[None, :, :, None] * torch.linalg.solve_ex(Sv, Sigma[:, None].expand_as(Sv)).result)
  G = T / (La[None, :, :, None] * Lb[None, :, None, :])
  SP_a = Sigma[:, None] @ Pa[:, ia]
  SP_b = Sigma[:, None] @ Pa[:, ib]
  Dr = _sym(SP_a / La[None, :, :, None]) - T / (La[None, :, :, None] * La[None, :, None, :])
  Dc = _sym(SP_b / Lb[None, :, :, None]) - T / (Lb[None, :, :, None] * Lb[None, :, None, :])
  const = (-0.5 * torch.linalg.slogdet(Sv)[1] + 0.5 * torch.log(V).sum(-1)[None]
           - 0.5 * torch.log(La).sum(-1)[None] - 0.5 * torch.log(Lb).sum(-1)[None]
           + 0.5 * ld[:, ia] + 0.5 * ld[:, ib])
  return Pa, lognorm, G, Dr, Dc, const


def _packed_order(pm: ops.PackedModel, pre):
  """``pre`` with the inducing points in the pack's own order (``PackedModel.perm``): the per-point sums of
  ``mm_backward_sums`` are in that order, and everything below multiplies them with Z and beta point by point."""
  Z, ls, var, beta, C, mean_c = pre
  perm = pm.perm()
  return (Z.gather(1, perm[..., None].expand(-1, -1, Z.shape[-1])), ls, var, beta.gather(1, perm), C, mean_c)


def _backward_sums(pm: ops.PackedModel, mu: torch.Tensor, L: int, M: int, d: int, B: int, flags: int,
                   full_output_cov: bool):
  """``mm_backward_sums`` -> Ksum, csum, cC [B,P,M], Usum [B,P,M,d], Rsum, rsum [B,P,M]."""
  dev = mu.device
  ws = pm.workspace(B, flags)
  nbytes = lib().mm_backward_bytes(B, L, M, d, flags)
  out = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
  rc = lib().mm_backward_sums(pm.buf.data_ptr(), pm.nbytes, L, M, d, _lib.MM_F64, B, mu.contiguous().data_ptr(),
                              flags, ws.data_ptr(), ws.numel(), out.data_ptr(), nbytes, ops._stream(dev))
  check(rc, "mm_backward_sums")
  Mp = (M + _lib.MM_M_ALIGN - 1) // _lib.MM_M_ALIGN * _lib.MM_M_ALIGN
  P = L * (L + 1) // 2 if full_output_cov else L
  Po = P - L
  ncol = B * P * (3 + d) * Mp
  col = out[:ncol].view(B, P, 3 + d, Mp)[..., :M]
  Ksum, csum, cC, Usum = col[:, :, 0], col[:, :, 1], col[:, :, 2], col[:, :, 3:].transpose(2, 3)   # U [B,P,M,d]
  if Po:
    row = out[ncol:].view(B, Po, 2, Mp)[..., :M]
    Rsum = torch.cat([Ksum[:, :L], row[:, :, 0]], dim=1)        # diagonal pairs are symmetric
    rsum = torch.cat([csum[:, :L], row[:, :, 1]], dim=1)
  else:
    Rsum, rsum = Ksum, csum
  return Ksum, csum, cC, Usum, Rsum, rsum


def moment_match_backward(pm: ops.PackedModel, pre, mu: torch.Tensor, Sigma: torch.Tensor,
                          full_output_cov: bool, model_uncertainty: bool,
                          g_f1: torch.Tensor, g_Sff: torch.Tensor, g_cross: torch.Tensor):
  """-> (dL/dmu [B,d], dL/dSigma [B,d,d] symmetrised).  Re-runs the q stage for (mu, Sigma) (the backward
  kernel reads its operands from the workspace; the M^2 forward reduce is not needed again).

  Moment form: every M-sized factor enters through a coefficient vector c_m (d w_m = w_m d log q_m,
  and the pair terms are linear in rho, gamma, G), so the sums over m collapse to the raw moments
  sum_m c_m (1, z_m, z_m z_m^T) -- one GEMM over M -- and what is left is the chain rule through the
  d x d algebra of ``k_prep``, written out in closed form on [B,P,d,d] tensors (no autograd graph;
  ``moment_match_backward_reference`` is the autograd version it is tested against)."""
  if pm.dtype != torch.float64:
    raise NotImplementedError("the backward sums are taken on a float64 pack (see MomentMatchFunction)")
  Z, ls, var, beta, _, mean_c = _packed_order(pm, pre)
  L, M, d = Z.shape
  B = mu.shape[0]
  dev = mu.device
  flags = ops.make_flags(full_output_cov, model_uncertainty)
  with torch.no_grad():
    _, _, q = ops.q_forward(pm, mu, Sigma, flags, want_q=True)              # [B,L,M]; fills the workspace the sums read
    q = q.gather(2, pm.perm()[None].expand(B, L, M))                        # (q_forward's q is in the caller's order)
  Ksum, csum, cC, Usum, Rsum, rsum = _backward_sums(pm, mu, L, M, d, B, flags, full_output_cov)
  ia, ib = pair_indices(L, full_output_cov, dev)
  if full_output_cov:
    g_pair = torch.cat([torch.diagonal(g_Sff, dim1=-2, dim2=-1),
                        g_Sff[:, ia[L:], ib[L:]] + g_Sff[:, ib[L:], ia[L:]]], dim=1)       # [B,P]
  else:
    g_pair = g_Sff
  with torch.no_grad():
    # ---- detached M-sized pieces -------------------------------------------------------------
    w = beta[None] * q
    Za = torch.cat([torch.ones(L, M, 1, dtype=Z.dtype, device=dev), Z,
                    (Z[..., :, None] * Z[..., None, :]).reshape(L, M, d * d)], dim=-1)      # [L,M,1+d+d^2]
    mom = lambda coef, idx: torch.einsum('bpm,pmk->bpk', coef, Za[idx])      # one GEMM over M per pair side
    eye_idx = torch.arange(L, device=dev)
    Ssym0 = _sym(Sigma)
    SL0 = Ssym0[:, None] + (ls * ls)[None, :, :, None] * torch.eye(d, dtype=Sigma.dtype, device=dev)
    Pa0 = torch.linalg.inv_ex(SL0).inverse                                             # [B,L,d,d]
    pv = torch.einsum('blij,bil->blj', Pa0, g_cross)                        # Pa v, v = g_cross[b,:,a]   [B,L,d]
    e_m = (torch.einsum('lmd,bld->blm', Z, pv) - torch.einsum('bd,bld->bl', mu, pv)[..., None]) * w
    gp = g_pair[:, :, None]
    c_m = g_f1[:, :, None] * w + e_m
    pair_w = torch.zeros(B, L, M, dtype=w.dtype, device=dev)
    pair_w.index_add_(1, ia, gp * rsum)
    pair_w.index_add_(1, ib, gp * csum)
    c_m = c_m + pair_w * w
    if model_uncertainty:
      c_m = c_m + 2.0 * g_pair[:, :L, None] * cC[:, :L] * q
    cmom = mom(c_m, eye_idx)                                                # [B,L,1+d+d^2]
    wmom = mom(w, eye_idx)
    w0, w1 = wmom[..., 0], wmom[..., 1:1 + d]
    s_det = w1 - w0[..., None] * mu[:, None, :]                             # sum_m w_m zeta_m          [B,L,d]
    Rmom = mom(Rsum, ia)                                                    # zeta^a   side  [B,P,...]
    Kmom = mom(Ksum, ib)                                                    # zeta^a'  side
    K0, K1 = Kmom[..., 0], Kmom[..., 1:1 + d]
    u = Usum.sum(2)                                                         # sum_j U_j                 [B,P,d]
    X = torch.einsum('bpmd,pme->bpde', Usum, Z[ib])                         # sum_j U_j z'_j^T          [B,P,d,d]
    m1c = K1 - K0[..., None] * mu[:, None, :]                               # sum_j K_j zeta'_j (detached)
    # ---- closed-form chain rule through the d x d algebra (no autograd graph) ---------------------
    # per latent:  F_a = c0 lognorm - 1/2 <Pa, C2(mu)> + v^T Pa s - w0 (Pa v)^T mu,   Pa = (Sigma + Lam_a)^-1
    #   dPa = -Pa dSigma Pa,  d logdet(Sigma + Lam_a) = tr(Pa dSigma),  C2(mu) = sum_m c_m zeta_m zeta_m^T
    # per pair:    F_p = g [K0 const - 1/2 <Dr, R2(mu)> - 1/2 <Dc, K2(mu)> + <G, X - u mu^T> - mu^T G m1c]
    #   T = V - V Sv^-1 V,  Sv = Sigma + V:  dT = W dSigma W^T, W = V Sv^-1;   G = Lam_a^-1 T Lam_b^-1,
    #   Dr = Lam_a^-1 - Pa - Lam_a^-1 T Lam_a^-1,  Dc likewise with b,
    #   const = -1/2 logdet Sv + 1/2 logdet(Sigma + Lam_a) + 1/2 logdet(Sigma + Lam_b) + ...
    def second_moment(m, muv):                            # sum_m c_m (z - mu)(z - mu)^T from raw moments
      m0, m1, m2 = m[..., 0], m[..., 1:1 + d], m[..., 1 + d:].reshape(m.shape[:-1] + (d, d))
      o = m1[..., :, None] * muv[..., None, :]
      return m2 - o - o.transpose(-1, -2) + m0[..., None, None] * (muv[..., :, None] * muv[..., None, :])
    muL = mu[:, None, :]                                                    # [B,1,d]
    v = g_cross.transpose(1, 2)                                             # [B,L,d]
    c0, c1 = cmom[..., 0], cmom[..., 1:1 + d]
    C2 = second_moment(cmom, muL)
    vs = v[..., :, None] * s_det[..., None, :]
    Abar = -0.5 * C2 + 0.5 * (vs + vs.transpose(-1, -2))
    gS = -(Pa0 @ Abar @ Pa0) - 0.5 * c0[..., None, None] * Pa0              # [B,L,d,d]
    gmu = (torch.einsum('blij,blj->bli', Pa0, c1 - c0[..., None] * muL) - w0[..., None] * pv)
    gS, gmu = gS.sum(1), gmu.sum(1)
    ls2 = ls * ls
    La, Lb = ls2[ia], ls2[ib]                                               # [P,d]
    V = La * Lb / (La + Lb)
    eye = torch.eye(d, dtype=Sigma.dtype, device=dev)
    Svi = torch.linalg.inv_ex(Ssym0[:, None] + V[None, :, :, None] * eye).inverse      # [B,P,d,d]
    Wm = V[None, :, :, None] * Svi                                          # V Sv^-1
    T = V[None, :, :, None] * eye - Wm * V[None, :, None, :]
    iab = 1.0 / (La[:, :, None] * Lb[:, None, :])
    iaa = 1.0 / (La[:, :, None] * La[:, None, :])
    ibb = 1.0 / (Lb[:, :, None] * Lb[:, None, :])
    G = T * iab[None]
    Pa_p, Pb_p = Pa0[:, ia], Pa0[:, ib]
    Dr = (1.0 / La)[None, :, :, None] * eye - Pa_p - T * iaa[None]
    Dc = (1.0 / Lb)[None, :, :, None] * eye - Pb_p - T * ibb[None]
    R0, R1 = Rmom[..., 0], Rmom[..., 1:1 + d]
    R2, K2 = second_moment(Rmom, muL), second_moment(Kmom, muL)
    XG = (X - u[..., :, None] * muL[..., None, :]) * iab[None]
    Tbar = 0.5 * (XG + XG.transpose(-1, -2)) + 0.5 * R2 * iaa[None] + 0.5 * K2 * ibb[None]
    gS_p = (Wm.transpose(-1, -2) @ Tbar @ Wm - 0.5 * (Pa_p @ R2 @ Pa_p) - 0.5 * (Pb_p @ K2 @ Pb_p)
            + K0[..., None, None] * (0.5 * (Pa_p + Pb_p) - 0.5 * Svi))
    gmu_p = (torch.einsum('bpij,bpj->bpi', Dr, R1 - R0[..., None] * muL)
             + torch.einsum('bpij,bpj->bpi', Dc, K1 - K0[..., None] * muL)
             - torch.einsum('bpji,bpj->bpi', G, u) - torch.einsum('bpij,bpj->bpi', G, m1c))
    gS = gS + (g_pair[..., None, None] * gS_p).sum(1)
    gmu = gmu + (g_pair[..., None] * gmu_p).sum(1)
  return gmu, _sym(gS)


def moment_match_backward_reference(pm: ops.PackedModel, pre, mu: torch.Tensor, Sigma: torch.Tensor,
                          full_output_cov: bool, model_uncertainty: bool,
                          g_f1: torch.Tensor, g_Sff: torch.Tensor, g_cross: torch.Tensor):
  """First version of ``moment_match_backward`` (kept as the cross-check of the moment form): the
  surrogate is evaluated on [B,P,M,d]-sized tensors, so autograd runs over M-sized graphs."""
  ops.q_forward(pm, mu, Sigma, ops.make_flags(full_output_cov, model_uncertainty))
  if pm.dtype != torch.float64:
    raise NotImplementedError("the backward pass is built for float64 models only (first version)")
  Z, ls, var, beta, _, mean_c = _packed_order(pm, pre)
  L, M, d = Z.shape
  B = mu.shape[0]
  dev = mu.device
  flags = ops.make_flags(full_output_cov, model_uncertainty)
  ws = pm.workspace(B, flags)
  nbytes = lib().mm_backward_bytes(B, L, M, d, flags)
  out = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
  rc = lib().mm_backward_sums(pm.buf.data_ptr(), pm.nbytes, L, M, d, _lib.MM_F64, B, mu.contiguous().data_ptr(),
                              flags, ws.data_ptr(), ws.numel(), out.data_ptr(), nbytes, ops._stream(dev))
  check(rc, "mm_backward_sums")
  Mp = (M + _lib.MM_M_ALIGN - 1) // _lib.MM_M_ALIGN * _lib.MM_M_ALIGN
  P = L * (L + 1) // 2 if full_output_cov else L
  Po = P - L
  ncol = B * P * (3 + d) * Mp
  col = out[:ncol].view(B, P, 3 + d, Mp)[..., :M]
  Ksum, csum, cC, Usum = col[:, :, 0], col[:, :, 1], col[:, :, 2], col[:, :, 3:].transpose(2, 3)   # U [B,P,M,d]
  if Po:
    row = out[ncol:].view(B, Po, 2, Mp)[..., :M]
    Rsum = torch.cat([Ksum[:, :L], row[:, :, 0]], dim=1)        # diagonal pairs are symmetric
    rsum = torch.cat([csum[:, :L], row[:, :, 1]], dim=1)
  else:
    Rsum, rsum = Ksum, csum

  ia, ib = pair_indices(L, full_output_cov, dev)
  if full_output_cov:
    g_pair = torch.cat([torch.diagonal(g_Sff, dim1=-2, dim2=-1),
                        g_Sff[:, ia[L:], ib[L:]] + g_Sff[:, ib[L:], ia[L:]]], dim=1)       # [B,P]
  else:
    g_pair = g_Sff
  with torch.enable_grad():
    mu_ = mu.detach().clone().requires_grad_(True)
    S_ = Sigma.detach().clone().requires_grad_(True)
    Ssym = _sym(S_)
    Pa, lognorm, G, Dr, Dc, const = small_algebra(Ssym, ls * ls, var, ia, ib)
    zeta = Z[None] - mu_[:, None, None, :]                                                # [B,L,M,d]
    maha = torch.einsum('blmi,blij,blmj->blm', zeta, Pa, zeta)
    q = torch.exp(lognorm[..., None] - 0.5 * maha)
    w = beta[None] * q
    f1 = w.sum(-1) + (0.0 if mean_c is None else mean_c[None])
    cross = torch.einsum('blde,ble->bdl', Pa, torch.einsum('blm,blmd->bld', w, zeta))
    total = (g_f1 * f1).sum() + (g_cross * cross).sum()
    zr, zc = zeta[:, ia], zeta[:, ib]                                                     # [B,P,M,d]
    rho = -0.5 * torch.einsum('bpmi,bpij,bpmj->bpm', zr, Dr, zr)
    gam = -0.5 * torch.einsum('bpmi,bpij,bpmj->bpm', zc, Dc, zc)
    m1c = torch.einsum('bpm,bpmd->bpd', Ksum, zc.detach())
    bil = (torch.einsum('bpmd,bpde,bpme->bp', Usum, G, zc)
           - torch.einsum('bd,bpd->bp', mu_, torch.einsum('bpde,bpe->bpd', G.detach(), m1c)))
    sur = (Ksum.sum(-1) * const + (Rsum * rho).sum(-1) + (Ksum * gam).sum(-1) + bil
           + (rsum * w[:, ia]).sum(-1) + (csum * w[:, ib]).sum(-1))
    if model_uncertainty:
      sur_diag = 2.0 * (cC[:, :L] * q).sum(-1)
      sur = torch.cat([sur[:, :L] + sur_diag, sur[:, L:]], dim=1)
    total = total + (g_pair * sur).sum()
    gmu, gS = torch.autograd.grad(total, (mu_, S_))
  return gmu, _sym(gS)


# Byte budget of the FUSED differentiable match (value from the backward's sweeps, ``mm_moment_match_with_sums``): every such match
# keeps its sums (``ops.backward_workspace_bytes``: 0.5 GB at C3 with B = 256) on its autograd node until its backward has run, so a
# torch-composed H-step rollout holds H of them.  Above the limit a match takes the two-pass form instead (forward's sweeps now,
# the backward's sweeps in the backward: nothing kept) -- the native tape caps the same data the same way (MM_TAPE_WS_LIMIT).
FUSED_SUMS_LIMIT = int(os.environ.get("GPFLOWPILCO_FUSED_SUMS_LIMIT", str(2 << 30)))
_fused_live = [0]                    # bytes of sums buffers alive on autograd nodes


def fused_sums_live_bytes() -> int:
  return _fused_live[0]


def _release_fused(nbytes: int) -> None:
  _fused_live[0] -= nbytes


class MomentMatchFunction(torch.autograd.Function):
  """``ops.moment_match`` as a differentiable function of (mu, Sigma) for a frozen packed model.

  ``pm`` runs the forward in the inputs' dtype; ``pm_bwd`` is the pack the backward runs on: ``pm`` itself for a
  float64 model and for a float32 model with d <= 8 (``ops.backward_supported``: diagonal pairs in f64, off-diagonal
  pairs as moment + bf16-MFMA aggregates), else a float64 pack of the same model (the f32 state cast up).

  Accuracy contract of the float32 pack (csrc/mm_route.hip): the diagonal pairs, the first moments,
  the cross term and the polynomial part 1 + b + b^2/2 of every off-diagonal pair are float64 in both directions; the
  off-diagonal remainder runs in f32 (forward) / bf16-split MFMA (backward) and carries its own rounding-error estimate --
  a (batch element, pair) whose estimate exceeds 3e-4 of the batch element's off-diagonal covariance scale is re-reduced
  in float64, forward sum and backward aggregates alike, and counted (``PackedModel.routed()``).  Measured: gradients
  within 2e-4 of the float64 pack's on every draw of tests/test_gpu_backward_f32.py (1e-6 .. 3e-5 typical).
  ``moment_match_differentiable(..., backward_dtype=torch.float64)`` keeps the float64-pack backward reachable."""

  @staticmethod
  def forward(ctx, mu, Sigma, pm, pm_bwd, pre, full_output_cov, model_uncertainty, fused=True):
    ctx.sums = None
    ctx.pm_bwd, ctx.pre, ctx.flags = pm_bwd, pre, (full_output_cov, model_uncertainty)
    need = 0
    if fused and pm_bwd is pm and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and mu.shape[0] > 0:
      need = ops.backward_workspace_bytes(pm, mu.shape[0], ops.make_flags(full_output_cov, model_uncertainty))
      if _fused_live[0] + need > FUSED_SUMS_LIMIT:
        need = 0                       # over the budget: two passes, nothing kept (see FUSED_SUMS_LIMIT)
    if need:
      # value AND sums in one pass (mm_moment_match_with_sums): the backward's M x M sweeps do not depend on the incoming
      # gradient and contain the forward's sums, so a call that will be differentiated runs THEM instead of the forward's
      # sweeps and its backward is the chain rule alone.  The sums (C3 shape, B = 256: 0.5 GB) live on this node until its
      # backward has run
      f1, Sff, cross, ctx.sums, ctx.generation = ops.moment_match_with_sums(pm, mu, Sigma, full_output_cov=full_output_cov,
                                                                            model_uncertainty=model_uncertainty)
      if ctx.sums is not None:
        _fused_live[0] += need
        weakref.finalize(ctx.sums, _release_fused, need)      # freed with the buffer: after the backward, or with a dropped graph
      ctx.save_for_backward(mu, Sigma)
      return f1, Sff, cross
    f1, Sff, cross = ops.moment_match(pm, mu, Sigma, full_output_cov=full_output_cov,
                                      model_uncertainty=model_uncertainty)
    ctx.save_for_backward(mu, Sigma)
    # the backward may reuse the q stage this forward left on the workspace, if it runs on the same pack and nothing else
    # touches that workspace in between (ops.PackedModel.workspace_generation)
    ctx.generation = (pm.workspace_generation(mu.shape[0], ops.make_flags(full_output_cov, model_uncertainty))
                      if pm_bwd is pm else None)
    return f1, Sff, cross

  @staticmethod
  def backward(ctx, g_f1, g_Sff, g_cross):
    mu, Sigma = ctx.saved_tensors
    full, unc = ctx.flags
    pmb = ctx.pm_bwd
    mub, Sb = mu.to(pmb.dtype), Sigma.to(pmb.dtype)
    # native: M x M sweeps, M-sized moments and the d x d chain rule all on the device (mm_moment_match_backward);
    # moment_match_backward / _reference above are the torch forms it is tested against
    gmu, gS = ops.moment_match_backward(pmb, mub.contiguous(), Sb.contiguous(), g_f1, g_Sff, g_cross, full, unc,
                                        forward_generation=ctx.generation, sums=ctx.sums)
    ctx.sums = None
    return gmu.to(mu.dtype), gS.to(Sigma.dtype), None, None, None, None, None, None


def moment_match_differentiable(model, mu: torch.Tensor, Sigma: torch.Tensor, full_output_cov: bool = True,
                                model_uncertainty: bool = True, backward_dtype: Optional[torch.dtype] = None,
                                fused: bool = True):
  """(mu, Sigma) -> (f1, Sff, cross_pre) with gradients flowing back to (mu, Sigma).

  ``backward_dtype``: None = the forward's own pack where it has a backward (float64 models; float32 models with d <= 8,
  under the accuracy contract stated on ``MomentMatchFunction``), else a float64 pack of the model; ``torch.float64`` =
  always differentiate through the float64 pack (every pair swept in f64: 2.3 x the time at C3 shape).
  ``fused`` (default): when (mu, Sigma) require grad and the backward runs on the forward's own pack, the forward is
  ``mm_moment_match_with_sums`` -- the backward's sweeps, run once, give the value too -- and the backward is the chain rule
  alone; ``False`` keeps the two-pass form (forward's sweeps, then the backward's)."""
  if backward_dtype not in (None, torch.float64, mu.dtype):
    raise ValueError(f"backward_dtype must be None, torch.float64 or the state's dtype, got {backward_dtype}")
  pm = model.packed(dtype=mu.dtype, with_C=bool(model_uncertainty), device=mu.device)
  own = ops.backward_supported(pm) and backward_dtype in (None, mu.dtype)
  if backward_dtype == mu.dtype and not own:
    raise NotImplementedError("this pack has no backward of its own (float32 with d > 8): use backward_dtype=torch.float64")
  pm_bwd = pm if own else model.packed(dtype=torch.float64, with_C=bool(model_uncertainty), device=mu.device)
  pre = model._cache._pre
  return MomentMatchFunction.apply(mu, Sigma, pm, pm_bwd, pre, full_output_cov, model_uncertainty, bool(fused))


def moment_match_torch(mu, Sigma, Z, ls, var, beta, C=None, mean_c=None, full_output_cov: bool = True,
                       model_uncertainty: bool = True):
  """Fully differentiable torch evaluation of the same centred formulas, materialising the
  [B, P, M, M] blocks: for SMALL models whose PARAMETERS are being trained (the policy: M = 30,
  one output, ``loops/pilco.py:78-108``), where gradients w.r.t. (Z, lengthscales, q_mu) are needed
  and the M^2 work is negligible.  Large frozen models go through the HIP kernels instead."""
  L, M, d = Z.shape
  dev = mu.device
  ia, ib = pair_indices(L, full_output_cov, dev)
  Ssym = _sym(Sigma)
  Pa, lognorm, G, Dr, Dc, const = small_algebra(Ssym, ls * ls, var, ia, ib)
  zeta = Z[None] - mu[:, None, None, :]
  maha = torch.einsum('blmi,blij,blmj->blm', zeta, Pa, zeta)
  q = torch.exp(lognorm[..., None] - 0.5 * maha)
  w = beta[None] * q
  f1 = w.sum(-1) + (0.0 if mean_c is None else mean_c[None])
  cross = torch.einsum('blde,ble->bdl', Pa, torch.einsum('blm,blmd->bld', w, zeta))
  zr, zc = zeta[:, ia], zeta[:, ib]
  rho = -0.5 * torch.einsum('bpmi,bpij,bpmj->bpm', zr, Dr, zr)
  gam = -0.5 * torch.einsum('bpmi,bpij,bpmj->bpm', zc, Dc, zc)
  delta = (const[..., None, None] + rho[..., :, None] + gam[..., None, :]
           + torch.einsum('bpmi,bpij,bpnj->bpmn', zr, G, zc))
  E = torch.expm1(delta)
  Sp = torch.einsum('bpm,bpmn,bpn->bp', w[:, ia], E, w[:, ib])
  diag = Sp[:, :L]
  if model_uncertainty:
    if C is None:
      raise ValueError("model_uncertainty needs C")
    diag = diag + var[None] + torch.einsum('lmn,blm,blmn,bln->bl', C, q, 1.0 + E[:, :L], q)
  if not full_output_cov:
    return f1, diag, cross
  Sff = torch.diag_embed(diag)
  if L > 1:
    off = Sp[:, L:]
    Sff = Sff.clone()
    Sff[:, ia[L:], ib[L:]] = off
    Sff[:, ib[L:], ia[L:]] = off
  return f1, Sff, cross


class ComposedRolloutFunction(torch.autograd.Function):
  """The whole moment-matched policy rollout as ONE differentiable op: forward = ``mm_rollout_composed_taped``,
  backward = ``mm_rollout_composed_backward`` (csrc/mm_compose_bwd.hip) -- the native counterpart of differentiating
  ``policy_loss_closure`` with a gradient tape (gpflow_pilco/utils/optimizers.py:51-56, loops/pilco.py:192-220).

  Inputs: the initial state (mx [B,nx], Sxx [B,nx,nx]) and the policy in PACKED coordinates -- Z [1,M,d],
  lengthscales [1,d], variance [1], beta = Kuu^-1 u [1,M], mean_c [1] -- which the caller computes from the trainable
  parameters with ordinary differentiable torch ops (``SVGP.precompute``: a 30 x 30 Cholesky), so autograd carries the
  gradient the last step to (q_mu, Z, lengthscales, variance).  The drift is frozen.  Output: cost [B, H]."""

  @staticmethod
  def forward(ctx, mx, Sxx, Z, ls, var, beta, mean_c, roll, num_steps, dt):
    f64 = torch.float64
    det = lambda t: t.detach().to(f64)
    pol = ops.pack_model(det(Z), det(ls), det(var), det(beta), None, det(mean_c), dtype=f64, sync=False)
    m_H, S_H, cost, tape = roll.taped(mx.detach(), Sxx.detach(), num_steps, dt=dt, policy=pol)
    ctx.roll, ctx.pol, ctx.tape, ctx.H, ctx.dt, ctx.B = roll, pol, tape, int(num_steps), float(dt), mx.shape[0]
    ctx.save_for_backward(ls)
    ctx.need_state = mx.requires_grad or Sxx.requires_grad
    ctx.shapes = (Z.shape, ls.shape, var.shape, beta.shape, mean_c.shape)
    return cost.T.contiguous()

  @staticmethod
  def backward(ctx, g_cost):
    (ls,) = ctx.saved_tensors
    g_pol, g_m, g_S = ctx.roll.backward(ctx.tape, g_cost.T.contiguous(), ctx.B, ctx.H, dt=ctx.dt, policy=ctx.pol,
                                        want_state_grad=ctx.need_state)
    M, d = ctx.pol.M, ctx.pol.d
    g = g_pol.sum(0)
    zs, lss, vs, bs, ms = ctx.shapes
    gZ = g[:M * d].reshape(zs)
    gbeta = g[M * d:M * d + M].reshape(bs)
    gls = (2.0 * ls.detach().reshape(-1) * g[M * d + M:M * d + M + d]).reshape(lss)        # d/d ls = 2 ls d/d ls^2
    gvar = g[M * d + M + d].reshape(vs)
    gmean = g[M * d + M + d + 1].reshape(ms)
    return g_m, g_S, gZ, gls, gvar, gbeta, gmean, None, None, None
