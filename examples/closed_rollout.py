"""Minimal end-to-end use of the drop-in API on one MI355X (needs the built library and a GPU).

  python examples/closed_rollout.py

1. a synthetic SVGP dynamics model (8 independent SE-ARD latents, 500 inducing points),
2. one moment match through the reference's dispatcher API (``moment_matching(x, model)``),
3. a 20-step moment-matched Euler rollout: through the Python fold (``DynamicalSystem.solve_forward``,
   gpflow_pilco/dynamics/solvers.py:67-135) and through the fused C-ABI call (``closed_rollout``),
4. the per-step expected cost of the trajectory (gpflow_pilco/components.py:26-37).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflowpilco_amd import dynamics                                        # noqa: E402
from gpflowpilco_amd.components import GaussianObjective                    # noqa: E402
from gpflowpilco_amd.moment_matching import GaussianMoments, moment_matching  # noqa: E402
from gpflowpilco_amd.synthetic import make_inputs, make_svgp                # noqa: E402


def main():
  if not torch.cuda.is_available():
    raise SystemExit("this example needs a GPU (the package has no CPU fallback)")
  dev, dtype = torch.device("cuda", 0), torch.float32
  L = d = 8
  model = make_svgp(L, 500, d, seed=0, device=str(dev), ls_bounds=(0.7, 3.0)).to_model(dev)
  mu, Sigma = make_inputs(16, d, seed=1, scale=0.1, lo=0.3, hi=0.7)
  mu = torch.tensor(mu, dtype=dtype, device=dev)
  Sigma = torch.tensor(Sigma, dtype=dtype, device=dev)

  match = moment_matching(GaussianMoments(moments=(mu, Sigma), centered=True), model)
  print("one match: mean", tuple(match.y.mean().shape), "cov", tuple(match.y.covariance().shape),
        "pre-inverted cross", tuple(match.cross[0].shape), match.cross[1])

  system = dynamics.DynamicalSystem(drift=model, solver=dynamics.MomentMatchingEuler())
  times = np.arange(1.0, 21.0)
  m_fold, S_fold = system.solve_forward(initial_time=0.0, initial_state=(mu, Sigma), solution_times=times,
                                        iterator="foldl")
  m_fused, S_fused, traj_m, traj_S = dynamics.closed_rollout(model, mu, Sigma, num_steps=20, keep_trajectory=True)
  print("fold vs fused rollout: max |d mean|", float((m_fold - m_fused).abs().max()),
        " max |d cov|", float((S_fold - S_fused).abs().max()))

  objective = GaussianObjective(target=torch.full((d,), 0.5, dtype=dtype, device=dev),
                                precis=4.0 * torch.eye(d, dtype=dtype, device=dev))
  cost = objective(x=GaussianMoments(moments=(traj_m, traj_S), centered=True), t=None)
  print("expected cost per step, batch element 0:", np.round(cost[:, 0].cpu().numpy(), 4))


if __name__ == "__main__":
  main()
