"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle).

CPU: the oracle and the algorithm-matched restatement reproduce the stored vectors.
GPU: the HIP path, through the C ABI, reproduces them (f64 and f32)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import mm_fused_ref as fr
from oracle import mm_oracle as mo
from tests.helpers import gp_model_from_oracle, scale_err, to_dev

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SVGP_FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "*.npz"))
                       if "kernel_expectation" not in p and not os.path.basename(p).startswith("gpr_"))


def load(name):
  z = np.load(os.path.join(HERE, name + ".npz"))
  p = mo.SVGPParams(Z=z["Z"], lengthscales=z["lengthscales"], variance=z["variance"], q_mu=z["q_mu"],
                    q_sqrt=z["q_sqrt"], whiten=bool(z["whiten"]),
                    mean_c=z["mean_c"] if "mean_c" in z else None, W=z["W"] if "W" in z else None)
  return z, p


def test_fixture_inventory():
  assert {"reftest_so", "reftest_mo_lcm", "reftest_mo_sep_whiten", "c1_shaped", "c2_cut"} <= set(SVGP_FIXTURES)


@pytest.mark.parametrize("name", SVGP_FIXTURES)
def test_oracle_reproduces_fixture(name):
  z, p = load(name)
  f1, Sff, cross = mo.mm_gauss_svgp_mo(z["mu"], z["Sigma"], p)
  assert np.abs(f1 - z["f1_unc"]).max() < 1e-12 * max(1, np.abs(f1).max())
  assert scale_err(Sff, z["Sff_unc"]) < 1e-10 and scale_err(cross, z["cross_unc"]) < 1e-10
  assert scale_err(mo.eKfu_list(z["mu"], z["Sigma"], p.Z, p.lengthscales, p.variance), z["eKfu"]) < 1e-12
  if p.W is None:
    beta, C = fr.precompute(p)
    g1, Gff, gc = fr.moment_match(z["mu"], z["Sigma"], p, beta, C)
    assert scale_err(g1, z["f1_unc"]) < 1e-9 and scale_err(Gff, z["Sff_unc"]) < 1e-6
    assert scale_err(gc, z["cross_unc"]) < 1e-9


def load_gpr(name="gpr_refdesign"):
  z = np.load(os.path.join(HERE, name + ".npz"))
  return z, mo.GPRParams(X=z["X"], Y=z["Y"], lengthscales=z["lengthscales"], variance=float(z["variance"]),
                         noise_variance=float(z["noise_variance"]), mean_c=float(z["mean_c"]))


def test_oracle_reproduces_the_gpr_reference_design():
  """tests/test_moment_matching.py:88-136 at the reference's own sizes (pinned to quadrature: tests/test_oracle_pin.py)."""
  z, p = load_gpr()
  for unc, tag in ((True, "unc"), (False, "nounc")):
    f1, Sff, cross = mo.mm_gauss_gpr(z["mu"], z["Sigma"], p, model_uncertainty=unc)
    assert scale_err(f1, z[f"f1_{tag}"]) < 1e-12 and scale_err(Sff, z[f"Sff_{tag}"]) < 1e-10
    assert scale_err(cross, z[f"cross_{tag}"]) < 1e-10


def test_kernel_expectation_fixture():
  z = np.load(os.path.join(HERE, "kernel_expectation_d2.npz"))
  var = float(z["var"])
  assert scale_err(mo.eKfu_se(z["mu"], z["Sigma"], z["A"], z["lsA"], var), z["eKfu_A"]) < 1e-13
  got = mo.eKuffu_se_pair(z["mu"], z["Sigma"], z["lsA"], var, z["A"], z["lsB"], var, z["B"], False, False)
  assert scale_err(got, z["eKuffu_AB"]) < 1e-13
  # Q = diag(q) exp(delta) diag(q'): the centred form used on the GPU, from the stored q and Q
  qA = z["eKfu_A"][0]
  delta = np.log(z["eKuffu_AA"][0]) - np.log(qA)[:, None] - np.log(qA)[None, :]
  assert np.all(np.isfinite(delta)) and np.abs(delta - delta.T).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("name", SVGP_FIXTURES)
def test_gpu_reproduces_fixture(name, dtype, device):
  from gpflowpilco_amd import ops
  from gpflowpilco_amd.moment_matching import GaussianMoments, moment_matching
  z, p = load(name)
  model = gp_model_from_oracle(p, device)
  x = GaussianMoments((to_dev(z["mu"], device, dtype), to_dev(z["Sigma"], device, dtype)), centered=True)
  tol = dict(f1=1e-9, Sff=1e-6) if dtype == torch.float64 else dict(f1=2e-6, Sff=2e-5)
  for unc, tag in ((True, "unc"), (False, "nounc")):
    m = moment_matching(x, model, model_uncertainty=unc)
    assert scale_err(m.y.mean(), z[f"f1_{tag}"]) < tol["f1"]
    assert scale_err(m.y.covariance(), z[f"Sff_{tag}"]) < tol["Sff"]
    assert scale_err(m.cross[0], z[f"cross_{tag}"]) < tol["f1"]
  md = moment_matching(x, model, full_output_cov=False)
  assert scale_err(md.y.covariance().diag_part(), z["Sff_diag"]) < tol["Sff"]
  if "traj_mu" in z:
    pm = model.packed(dtype, True, device)
    _, _, tmu, tS = ops.rollout_closed(pm, x.mean(), x.covariance(), z["traj_mu"].shape[0], keep_trajectory=True)
    assert scale_err(tmu, z["traj_mu"]) < tol["Sff"] and scale_err(tS, z["traj_Sigma"]) < tol["Sff"]
    assert scale_err(tmu[0], z["mu_next"]) < tol["Sff"] and scale_err(tS[0], z["Sigma_next"]) < tol["Sff"]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_gpu_reproduces_the_gpr_reference_design(dtype, device):
  from gpflowpilco_amd import models as gp
  from gpflowpilco_amd.moment_matching import GaussianMoments, moment_matching
  z, p = load_gpr()
  t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=device)
  model = gp.GPR(data=(t(p.X), t(p.Y)), kernel=gp.SquaredExponential(variance=t(p.variance), lengthscales=t(p.lengthscales)),
                 mean_function=gp.Constant(t([p.mean_c])), noise_variance=t(p.noise_variance))
  x = GaussianMoments((to_dev(z["mu"], device, dtype), to_dev(z["Sigma"], device, dtype)), centered=True)
  tol = dict(f1=1e-9, Sff=1e-6) if dtype == torch.float64 else dict(f1=2e-6, Sff=2e-5)
  for unc, tag in ((True, "unc"), (False, "nounc")):
    m = moment_matching(x, model, model_uncertainty=unc)
    want = (z[f"f1_{tag}"], z[f"Sff_{tag}"], z[f"cross_{tag}"])
    if dtype == torch.float32:
      # lengthscales down to 0.013 against a state rounded to f32: d cross / d mu ~ 1 / ls^2 turns the INPUT's rounding (6e-8)
      # into 6e-6 of the output -- the kernels are compared with the oracle at the state they were given
      want = mo.mm_gauss_gpr(x.mean().double().cpu().numpy(), x.covariance().double().cpu().numpy(), p, model_uncertainty=unc)
    assert scale_err(m.y.mean(), want[0]) < tol["f1"]
    assert scale_err(m.y.covariance(), want[1]) < tol["Sff"]
    assert scale_err(m.cross[0], want[2]) < tol["f1"]
