"""The rollout harness of ``MomentMatchingPILCO`` (``gpflow_pilco/loops/pilco.py:176-227``):
``policy_loss_closure`` builds the function whose value the policy optimiser minimises -- the
expected cost accumulated along a moment-matched rollout.  The RL orchestration around it
(data collection, checkpoints, optimisers) is out of scope."""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np
import torch

from .dynamics import DynamicalSystem, MomentMatchingEuler
from .moment_matching import GaussianMoments, moment_matching


def get_state_initializer(mean: torch.Tensor, covariance: torch.Tensor) -> Callable:
  """pilco.py:222-227: the initial state as moments with a leading batch axis."""
  mx = mean if mean.ndim > 1 else mean[None]
  Sxx = covariance if covariance.ndim > 2 else covariance[None]
  return lambda: (mx, Sxx)


def policy_loss_closure(system: DynamicalSystem, objective: Callable, state_initializer: Callable,
                        num_steps: int, initial_time: float = 0.0,
                        solution_times: Optional[Sequence[float]] = None, **kwargs) -> Callable:
  """pilco.py:176-220.  Returns ``closure() -> loss [B]``; ``system.solver`` should be a
  ``MomentMatchingEuler`` (pilco.py:141-144)."""
  if solution_times is None:
    solution_times = np.arange(1, 1 + num_steps, dtype=np.float64)     # pilco.py:186
  encoder = system.encoder

  def _accumulate_loss(t, state, loss):                                # pilco.py:199-205
    x = GaussianMoments(moments=state, centered=True)
    if encoder is not None:
      x = moment_matching(x, encoder).y
    return loss + objective(x=x, t=t)

  def _closure():                                                      # pilco.py:207-217
    mx, Sxx = state_initializer()
    loss = torch.zeros(mx.shape[:-1], dtype=mx.dtype, device=mx.device)
    _, loss = system.solve_forward(iterator="foldl", initial_time=initial_time,
                                   initial_state=(mx, Sxx), solution_times=solution_times,
                                   callbacks_and_initializers=((_accumulate_loss, loss),), **kwargs)
    return loss

  return _closure
