"""Backward of the moment match w.r.t. (mu, Sigma) (row f-1): HIP M^2 sums + torch surrogate vs
central finite differences of the fp64 CPU oracle on a random linear functional of the outputs."""
import numpy as np
import pytest
import torch

from gpflowpilco_amd.autodiff import moment_match_differentiable
from gpflowpilco_amd.synthetic import generate_covariance
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
