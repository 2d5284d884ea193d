"""`update_policy` on the cartpole wiring with the native gradient (needs the built library and a GPU).

  python examples/policy_update.py [--steps 200] [--eager]

What the reference does in examples/cartpole_swingup/swingup_loops.py:76-103 + train_utils.py:91-105: minimise the
moment-matched rollout loss of `MomentMatchingPILCO.policy_loss_closure` (loops/pilco.py:176-220) over the policy's
parameters with Adam (clipnorm 1.0, learning rate 1e-2).  Here the closure's value AND gradient come from the HIP library:
`loops.policy_loss_closure` picks the taped native rollout + reverse sweep (mm_rollout_composed_taped / _backward,
csrc/mm_compose_bwd.hip) because the policy is trainable, the drift frozen and the state float64; `GraphedPolicyLoss`
replays forward + backward from one HIP graph per Adam step.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflowpilco_amd import bijectors as tfb, dynamics, models as gp                      # noqa: E402
from gpflowpilco_amd.components import GaussianObjective, TrigonometricEncoder          # noqa: E402
from gpflowpilco_amd.loops import GraphedPolicyLoss, get_state_initializer, policy_loss_closure   # noqa: E402
from gpflowpilco_amd.synthetic import make_cartpole_like, make_inputs                   # noqa: E402


def build(dev, H=30, B=1, seed=1000):
  F64 = torch.float64
  drift_s, pol_s = make_cartpole_like(100, 30, seed, device=str(dev))
  drift, pol = drift_s.to_model(dev), pol_s.to_model(dev)
  kern = pol.latent_kernels[0]
  params = [pol.q_mu, pol.inducing_variable.inducing_variable.Z if hasattr(pol.inducing_variable, "inducing_variable")
            else pol.inducing_variable.Z, kern.lengthscales, kern.variance]
  for p in params:
    p.requires_grad_(True)
  policy = gp.InverseLinkWrapper(gp.KernelRegressor(pol), invlink=tfb.Chain([tfb.Scale(2.0), tfb.Shift(-0.5), tfb.NormalCDF()]))
  system = dynamics.DynamicalSystem(drift=drift, policy=policy, encoder=TrigonometricEncoder(active_dims=(1,)),
                                    solver=dynamics.MomentMatchingEuler())
  t = lambda a: torch.tensor(np.asarray(a), dtype=F64, device=dev)
  target = np.array([0.0, 1.0, 0.0, 0.0, 0.0])
  precis = 16 * np.array([[0.25, 0, -0.5, 0, 0], [0, 0.25, 0, 0, 0], [-0.5, 0, 1, 0, 0], [0] * 5, [0] * 5], dtype=float)
  objective = GaussianObjective(target=t(target), precis=t(precis))
  rng = np.random.default_rng(seed)
  mu = np.array([0.4, 0.2, 0.5, 0.3])[None] + 0.05 * rng.standard_normal((B, 4))
  _, S = make_inputs(B, 4, seed=3000, scale=0.05)
  closure = policy_loss_closure(system, objective, get_state_initializer(t(mu), t(S)), H)
  return closure, params, drift


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--eager", action="store_true", help="no HIP graph: closure() and backward() per step")
  args = ap.parse_args()
  if not torch.cuda.is_available():
    raise SystemExit("this example needs a GPU (the package has no CPU fallback)")
  dev = torch.device("cuda", 0)
  H = 30
  closure, params, drift = build(dev, H)
  opt = torch.optim.Adam(params, lr=1e-2)
  graphed = None if args.eager else GraphedPolicyLoss(closure, params)
  losses = []
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for it in range(args.steps):
    if graphed is None:
      opt.zero_grad(set_to_none=True)
      loss = closure().sum()
      loss.backward()
    else:
      loss, _ = graphed.loss_and_grad()
      loss = loss.sum()
    torch.nn.utils.clip_grad_norm_(params, 1.0)                     # swingup_loops.py: clipnorm 1.0
    opt.step()
    losses.append(loss.detach())
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
  losses = torch.stack(losses).cpu().numpy()
  if graphed is not None:
    graphed.check()
  drift.packed(torch.float64, True, dev).check_status(1)
  print(f"{args.steps} Adam steps of the H = {H} rollout loss: {losses[0]:.5f} -> {losses[-1]:.5f} "
        f"({1e3 * dt / args.steps:.2f} ms per step = {1e3 * dt / args.steps / H:.4f} ms per rollout step, "
        f"{'eager' if graphed is None else 'HIP graph'})")
  return losses


if __name__ == "__main__":
  main()
