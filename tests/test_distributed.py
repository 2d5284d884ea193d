"""world_size-2 gloo test of the B-sharded rollout: shard -> independent work -> one all-gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpflowpilco_amd import distributed as D


def test_shard_range_partitions():
  for B in (1, 2, 7, 256, 257):
    for world in (1, 2, 3, 8):
      spans = [D.shard_range(B, r, world) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == B
      assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
      sizes = [hi - lo for lo, hi in spans]
      assert max(sizes) - min(sizes) <= 1


def _fake_rollout(mu, Sigma, H=5):
  """CPU stand-in for the per-element rollout cost: depends only on its own batch element."""
  steps = torch.arange(1, H + 1, dtype=mu.dtype)
  tr = torch.diagonal(Sigma, dim1=-2, dim2=-1).sum(-1)
  return -torch.exp(-0.5 * (mu.pow(2).sum(-1, keepdim=True) + tr[:, None] * steps[None, :]))


def _worker(rank, world, port, B, result_dir):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    g = torch.Generator().manual_seed(0)
    mu = torch.randn(B, 3, generator=g, dtype=torch.float64)
    A = torch.randn(B, 3, 3, generator=g, dtype=torch.float64)
    Sigma = A @ A.transpose(1, 2)
    full = D.distributed_rollout_costs(_fake_rollout, mu, Sigma)
    want = _fake_rollout(mu, Sigma)
    assert full.shape == want.shape
    assert torch.equal(full, want), "gathered costs differ from the single-process result"
    lo, hi = D.shard_range(B, rank, world)
    mu_l, S_l = D.shard_batch(mu, Sigma)
    assert mu_l.shape[0] == hi - lo and torch.equal(mu_l, mu[lo:hi])
    torch.save(full, os.path.join(result_dir, f"r{rank}.pt"))
  finally:
    dist.destroy_process_group()


def _free_port():
  s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
  return p


@pytest.mark.parametrize("B", [8, 7])     # even and ragged shards
def test_two_rank_gloo_gather(B, tmp_path):
  mp.spawn(_worker, args=(2, _free_port(), B, str(tmp_path)), nprocs=2, join=True)
  a = torch.load(tmp_path / "r0.pt"); b = torch.load(tmp_path / "r1.pt")
  assert torch.equal(a, b) and a.shape == (B, 5)


def _toy_policy_loss(theta, w):
  """Differentiable stand-in for the per-element rollout loss: depends on the shared parameters and on its own element."""
  def fn(mu, Sigma):
    tr = torch.diagonal(Sigma, dim1=-2, dim2=-1).sum(-1)
    return torch.tanh(mu @ theta).pow(2) + w.exp() * tr + (theta * theta).sum() * mu[:, 0]
  return fn


def _grad_worker(rank, world, port, B, result_dir):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    g = torch.Generator().manual_seed(1)
    mu = torch.randn(B, 3, generator=g, dtype=torch.float64)
    A = torch.randn(B, 3, 3, generator=g, dtype=torch.float64)
    Sigma = A @ A.transpose(1, 2)
    theta = torch.randn(3, generator=g, dtype=torch.float64).requires_grad_(True)
    w = torch.tensor(0.3, dtype=torch.float64, requires_grad=True)
    unused = torch.zeros(2, dtype=torch.float64, requires_grad=True)
    loss, grads = D.distributed_loss_and_grad(_toy_policy_loss(theta, w), [theta, w, unused], mu, Sigma)
    want = _toy_policy_loss(theta, w)(mu, Sigma).mean()
    gw = torch.autograd.grad(want, [theta, w])
    assert abs(float(loss) - float(want)) <= 1e-14 * max(1.0, abs(float(want)))
    assert torch.allclose(grads[0], gw[0], rtol=1e-13, atol=1e-15) and torch.allclose(grads[1], gw[1], rtol=1e-13, atol=1e-15)
    assert torch.equal(grads[2], torch.zeros(2, dtype=torch.float64))
    torch.save((loss, grads), os.path.join(result_dir, f"g{rank}.pt"))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7, 1])     # even, ragged, and a rank with an empty shard
def test_two_rank_gloo_policy_gradient_all_reduce(B, tmp_path):
  """The sharded policy update: local loss sums and gradients, ONE all-reduce, every rank gets the full-batch mean loss and
  gradient (bitwise the same on both)."""
  mp.spawn(_grad_worker, args=(2, _free_port(), B, str(tmp_path)), nprocs=2, join=True)
  (l0, g0), (l1, g1) = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
  assert torch.equal(l0, l1) and all(torch.equal(a, b) for a, b in zip(g0, g1))


def test_single_process_loss_and_grad_needs_no_group():
  g = torch.Generator().manual_seed(2)
  mu = torch.randn(5, 3, generator=g, dtype=torch.float64)
  Sigma = torch.eye(3, dtype=torch.float64).expand(5, 3, 3).contiguous()
  theta = torch.randn(3, generator=g, dtype=torch.float64).requires_grad_(True)
  w = torch.tensor(-0.2, dtype=torch.float64, requires_grad=True)
  loss, grads = D.distributed_loss_and_grad(_toy_policy_loss(theta, w), [theta, w], mu, Sigma)
  want = _toy_policy_loss(theta, w)(mu, Sigma).mean()
  gw = torch.autograd.grad(want, [theta, w])
  assert torch.allclose(loss, want) and torch.allclose(grads[0], gw[0]) and torch.allclose(grads[1], gw[1])
