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
from tests.helpers import oracle_params, scale_err, to_dev

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
  assert (out[torch.float32][0] - out[torch.float64][0]).abs().max() < 2e-6
  assert (out[torch.float32][1] - out[torch.float64][1]).abs().max() < 2e-6
  std = torch.diagonal(out[torch.float64][1], dim1=-2, dim2=-1).sqrt()
  assert 0.02 < std.mean() < 0.5            # the state stays inside the data's support
