"""Data parallelism over the batch of input distributions.

Every tensor of the hot path has B as an untouched leading axis
(``utils/kernel_expectation.py:105,138-141``; ``moment_matching/models.py:236,245-248``), so
the rollout shards cleanly: one process per GPU, a contiguous slice of B each, the model
replicated, NO communication inside the H-step loop, and ONE all-gather of the per-step cost
matrix ``[B_local, H] -> [B, H]`` after the rollout (``torch.distributed``: RCCL on GPUs,
gloo in the CPU tests).  The policy UPDATE over a sharded batch of initial states adds the one exchange a gradient step
has: ONE all-reduce of (the policy-parameter gradient | loss sum) per evaluation -- a few hundred doubles, latency-bound
(``distributed_loss_and_grad``).  The reference has no distributed code; this is new design.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(B: int, rank: int, world: int) -> Tuple[int, int]:
  """Contiguous, balanced slice [lo, hi) of the batch axis (first B % world ranks get one more)."""
  base, rem = divmod(B, world)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(mu: torch.Tensor, Sigma: torch.Tensor, rank: Optional[int] = None,
                world: Optional[int] = None):
  rank = dist.get_rank() if rank is None else rank
  world = dist.get_world_size() if world is None else world
  lo, hi = shard_range(mu.shape[0], rank, world)
  return mu[lo:hi].contiguous(), Sigma[lo:hi].contiguous()


def gather_costs(cost_local: torch.Tensor, B_total: int, group=None) -> torch.Tensor:
  """All-gather ``[B_local, H]`` cost matrices of (possibly uneven) shards into ``[B_total, H]``.

  One collective per rollout; ragged shards are padded to the largest one (the payload is
  B*H scalars, e.g. 256 x 50 x 4 B = 51 KB: latency-bound, xGMI bandwidth is irrelevant)."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return cost_local
  world = dist.get_world_size(group)
  sizes = [shard_range(B_total, r, world) for r in range(world)]
  nmax = max(hi - lo for lo, hi in sizes)
  pad = torch.zeros((nmax,) + tuple(cost_local.shape[1:]), dtype=cost_local.dtype, device=cost_local.device)
  pad[:cost_local.shape[0]] = cost_local
  out = [torch.empty_like(pad) for _ in range(world)]
  dist.all_gather(out, pad, group=group)
  return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


def distributed_rollout_costs(rollout_fn: Callable, mu: torch.Tensor, Sigma: torch.Tensor,
                              group=None) -> torch.Tensor:
  """Shard (mu, Sigma) over the ranks, run ``rollout_fn(mu_local, Sigma_local) -> cost [B_local, H]``
  with no communication, then all-gather the costs.  Every rank returns the full ``[B, H]``."""
  B = mu.shape[0]
  if dist.is_initialized() and dist.get_world_size(group) > 1:
    mu_l, S_l = shard_batch(mu, Sigma, dist.get_rank(group), dist.get_world_size(group))
  else:
    mu_l, S_l = mu, Sigma
  return gather_costs(rollout_fn(mu_l, S_l), B, group)


def distributed_loss_and_grad(loss_fn: Callable, params, mu: torch.Tensor, Sigma: torch.Tensor, group=None):
  """Mean policy loss over ALL B initial states and its gradient w.r.t. ``params`` (tensors that require grad), data
  parallel: every rank evaluates ``loss_fn(mu_local, Sigma_local) -> per-element loss [B_local]`` on its contiguous shard
  (the native closures of ``loops``: taped rollout + reverse sweep, no communication inside), then ONE all-reduce sums
  the bucket (flattened gradients | local loss sum) over the ranks -- the reference's single-device
  ``tape.gradient(loss, variables)`` (``utils/optimizers.py:51-56``) for a batch that does not fit one GPU.
  Returns (loss, [grad per param]); identical on every rank.  Ragged and empty shards are handled (an empty shard
  contributes zeros)."""
  params = list(params)
  B = mu.shape[0]
  multi = dist.is_initialized() and dist.get_world_size(group) > 1
  if multi:
    mu_l, S_l = shard_batch(mu, Sigma, dist.get_rank(group), dist.get_world_size(group))
  else:
    mu_l, S_l = mu, Sigma
  if mu_l.shape[0] > 0:
    local = loss_fn(mu_l, S_l).sum()
    grads = torch.autograd.grad(local, params, allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, params)]
    local = local.detach()
  else:
    grads = [torch.zeros_like(p) for p in params]
    local = torch.zeros((), dtype=params[0].dtype if params else mu.dtype, device=mu.device)
  dtype = torch.float64
  bucket = torch.cat([g.reshape(-1).to(dtype) for g in grads] + [local.reshape(1).to(dtype)])
  if multi:
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
  bucket = bucket / B
  out, off = [], 0
  for p in params:
    n = p.numel()
    out.append(bucket[off:off + n].reshape(p.shape).to(p.dtype))
    off += n
  return bucket[off], out
