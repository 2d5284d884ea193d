"""Value and gradient from ONE pair of M x M sweeps (``mm_moment_match_with_sums`` + ``MM_SUMS_CURRENT``):
the backward's sweeps do not depend on the incoming gradient and contain the forward's sums.  The value must equal the plain
forward's (and the oracle's), the gradient the two-pass backward's, for both pack types."""
import numpy as np
import pytest
import torch

from gpflowpilco_amd import autodiff, ops
from gpflowpilco_amd.synthetic import make_inputs, make_svgp
from oracle import mm_oracle as mo
from tests.helpers import oracle_params, to_dev

pytestmark = pytest.mark.gpu
F64 = torch.float64

SHAPES = [((3, 200, 8, 3), True, True), ((1, 150, 4, 2), True, True), ((4, 130, 3, 5), False, True),
          ((2, 520, 8, 2), True, False), ((3, 77, 5, 1), True, True)]
IDS = ["L3d8", "L1", "diag_only_cov", "no_model_unc", "M77B1"]


def _setup(shape, dtype, device, seed=11):
  L, M, d, B = shape
  syn = make_svgp(L, M, d, seed=seed + L + d, device=str(device), ls_bounds=(0.5, 2.5))
  model = syn.to_model(device)
  mu, S = make_inputs(B, d, seed=seed, scale=0.1, lo=0.2, hi=0.8)
  mu, S = to_dev(mu, device, dtype), to_dev(S, device, dtype)
  return syn, model, mu, S


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("shape,full,unc", SHAPES, ids=IDS)
def test_value_from_the_backward_sweeps_equals_the_forward(shape, full, unc, dtype, device):
  syn, model, mu, S = _setup(shape, dtype, device)
  pm = model.packed(dtype, unc, device)
  f1, Sff, cross = ops.moment_match(pm, mu, S, full, unc)
  g1, Sgg, gcross, sums, gen = ops.moment_match_with_sums(pm, mu, S, full, unc)
  pm.check_status(mu.shape[0])
  assert torch.equal(f1, g1) and torch.equal(cross, gcross)                    # the same q stage
  scale = float(Sff.abs().max())
  tol = 3e-9 if dtype == torch.float64 else 2e-6      # f64: two summation orders of the C-weighted sums (|C| ~ 1e6: measured 2e-10 .. 1.1e-9)
  assert float((Sff - Sgg).abs().max()) <= tol * scale, (float((Sff - Sgg).abs().max()), scale)
  # and the oracle (literal restatement of the reference), at the state the kernels saw
  o1, oS, _ = mo.mm_gauss_svgp_mo(mu.double().cpu().numpy(), S.double().cpu().numpy(), oracle_params(syn),
                                  full_output_cov=full, model_uncertainty=unc)
  otol = 1e-6 if dtype == torch.float64 else 2e-5
  assert np.abs(Sgg.double().cpu().numpy() - oS).max() <= otol * max(scale, np.abs(oS).max())


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("shape,full,unc", SHAPES, ids=IDS)
def test_chain_rule_on_kept_sums_equals_the_two_pass_backward(shape, full, unc, dtype, device):
  syn, model, mu, S = _setup(shape, dtype, device, seed=23)
  pm = model.packed(dtype, unc, device)
  B, L, d = mu.shape[0], pm.L, pm.d
  gen = torch.Generator(device="cpu").manual_seed(5)
  g_f1 = torch.randn(B, L, generator=gen, dtype=F64).to(device)
  g_Sff = torch.randn((B, L, L) if full else (B, L), generator=gen, dtype=F64).to(device)
  g_cross = torch.randn(B, d, L, generator=gen, dtype=F64).to(device)
  ref_mu, ref_S = ops.moment_match_backward(pm, mu, S, g_f1, g_Sff, g_cross, full, unc)
  _, _, _, sums, generation = ops.moment_match_with_sums(pm, mu, S, full, unc)
  gmu, gS = ops.moment_match_backward(pm, mu, S, g_f1, g_Sff, g_cross, full, unc, forward_generation=generation, sums=sums)
  pm.check_status(B)
  assert torch.equal(gmu, ref_mu) and torch.equal(gS, ref_S)                   # the same sweeps, the same chain rule
  # another match on the pack's workspace in between: the q stage is re-run, the kept sums are still this state's
  _, _, _, sums, generation = ops.moment_match_with_sums(pm, mu, S, full, unc)
  ops.moment_match(pm, mu * 0.5, S * 2.0, full, unc)
  gmu2, gS2 = ops.moment_match_backward(pm, mu, S, g_f1, g_Sff, g_cross, full, unc, forward_generation=generation, sums=sums)
  pm.check_status(B)
  assert torch.equal(gmu2, ref_mu) and torch.equal(gS2, ref_S)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_autograd_takes_the_fused_path_and_agrees_with_two_passes(dtype, device, monkeypatch):
  syn, model, mu, S = _setup((3, 200, 8, 3), dtype, device, seed=31)
  calls = {"fused": 0, "plain": 0}
  real_fused, real_plain = ops.moment_match_with_sums, ops.moment_match
  monkeypatch.setattr(ops, "moment_match_with_sums", lambda *a, **k: (calls.__setitem__("fused", calls["fused"] + 1), real_fused(*a, **k))[1])
  monkeypatch.setattr(ops, "moment_match", lambda *a, **k: (calls.__setitem__("plain", calls["plain"] + 1), real_plain(*a, **k))[1])

  def loss_and_grad(fused):
    m_, S_ = mu.clone().requires_grad_(True), S.clone().requires_grad_(True)
    f1, Sff, cross = autodiff.moment_match_differentiable(model, m_, S_, True, True, fused=fused)
    loss = (f1.double() ** 2).sum() + (Sff.double() * torch.arange(Sff.numel(), device=device, dtype=F64).reshape(Sff.shape).cos()).sum() \
        + cross.double().sin().sum()
    loss.backward()
    return float(loss), m_.grad.clone(), S_.grad.clone()

  l1, gm1, gS1 = loss_and_grad(True)
  assert calls == {"fused": 1, "plain": 0}
  l0, gm0, gS0 = loss_and_grad(False)
  assert calls == {"fused": 1, "plain": 1}
  tol = 1e-10 if dtype == torch.float64 else 2e-5
  assert abs(l1 - l0) <= tol * abs(l0)
  assert float((gm1 - gm0).abs().max()) <= tol * float(gm0.abs().max())
  assert float((gS1 - gS0).abs().max()) <= tol * float(gS0.abs().max())
  # no gradient asked for: the plain forward
  with torch.no_grad():
    autodiff.moment_match_differentiable(model, mu, S, True, True)
  assert calls == {"fused": 1, "plain": 2}


def test_sums_of_another_state_are_reported(device):
  """MM_SUMS_CURRENT is verified on the device (the sums carry the mean they were swept for): kept sums handed to the backward
  of a DIFFERENT state raise instead of differentiating silently with them."""
  syn, model, mu, S = _setup((3, 200, 8, 3), torch.float32, device, seed=41)
  pm = model.packed(torch.float32, True, device)
  B, L, d = mu.shape[0], pm.L, pm.d
  g_f1 = torch.ones(B, L, dtype=F64, device=device); g_Sff = torch.ones(B, L, L, dtype=F64, device=device)
  g_cross = torch.ones(B, d, L, dtype=F64, device=device)
  _, _, _, sums, generation = ops.moment_match_with_sums(pm, mu, S)
  pm.check_status(B)
  other = mu.clone(); other[1] += 0.25
  ops.moment_match_backward(pm, other, S, g_f1, g_Sff, g_cross, forward_generation=None, sums=sums)
  with pytest.raises(RuntimeError, match="kept sums"):
    pm.check_status(B)
  ops.moment_match_backward(pm, mu, S, g_f1, g_Sff, g_cross, forward_generation=None, sums=sums)    # the right state: accepted
  pm.check_status(B)


def test_value_and_gradient_replay_from_a_hip_graph(device):
  """The f32 pack's aggregate chain runs on a side stream beside the diagonal sweep (fork / join by events, mm_compose_bwd.hip):
  under stream capture the side stream must join the capture -- a replayed graph returns what the eager calls return."""
  syn, model, mu, S = _setup((3, 200, 8, 3), torch.float32, device, seed=51)
  pm = model.packed(torch.float32, True, device)
  B, L, d = mu.shape[0], pm.L, pm.d
  gen = torch.Generator(device="cpu").manual_seed(6)
  g_f1 = torch.randn(B, L, generator=gen, dtype=F64).to(device)
  g_Sff = torch.randn(B, L, L, generator=gen, dtype=F64).to(device)
  g_cross = torch.randn(B, d, L, generator=gen, dtype=F64).to(device)

  def run(m_, S_):
    f1, Sff, cross, sums, generation = ops.moment_match_with_sums(pm, m_, S_)
    gmu, gS = ops.moment_match_backward(pm, m_, S_, g_f1, g_Sff, g_cross, forward_generation=generation, sums=sums)
    return f1, Sff, cross, gmu, gS
  want = [t.clone() for t in run(mu, S)]
  pm.check_status(B)
  ms, Ss = mu.clone(), S.clone()
  side = torch.cuda.Stream(device)
  side.wait_stream(torch.cuda.current_stream(device))
  with torch.cuda.stream(side):
    for _ in range(2):
      run(ms, Ss)
  torch.cuda.current_stream(device).wait_stream(side)
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    outs = run(ms, Ss)
  ms.copy_(mu * 0.9); Ss.copy_(S * 1.1)                    # another state through the same graph ...
  graph.replay()
  ms.copy_(mu); Ss.copy_(S)                                # ... and the first one again
  graph.replay()
  torch.cuda.synchronize()
  pm.check_status(B)
  for got, ref in zip(outs, want):
    assert torch.equal(got, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_forward_with_the_side_stream_replays_from_a_hip_graph(dtype, device):
  """mm_moment_match puts the off-diagonal operands and the moment chain on the side stream beside the diagonal sweep when the
  problem is large enough (P B >= 512) and joins before the off-diagonal sweep: captured, the side stream must join the capture."""
  L, M, d, B = 6, 300, 8, 32                                    # P B = 21 * 32 = 672
  syn = make_svgp(L, M, d, seed=61, device=str(device), ls_bounds=(0.5, 2.5))
  model = syn.to_model(device)
  pm = model.packed(dtype, True, device)
  mu, S = make_inputs(B, d, seed=62, scale=0.1, lo=0.2, hi=0.8)
  mu, S = to_dev(mu, device, dtype), to_dev(S, device, dtype)
  want = [t.clone() for t in ops.moment_match(pm, mu, S)]
  pm.check_status(B)
  ms, Ss = mu.clone(), S.clone()
  side = torch.cuda.Stream(device)
  side.wait_stream(torch.cuda.current_stream(device))
  with torch.cuda.stream(side):
    for _ in range(2):
      ops.moment_match(pm, ms, Ss)
  torch.cuda.current_stream(device).wait_stream(side)
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    outs = ops.moment_match(pm, ms, Ss)
  ms.copy_(mu * 0.9); Ss.copy_(S * 1.2)
  graph.replay()
  ms.copy_(mu); Ss.copy_(S)
  graph.replay()
  torch.cuda.synchronize()
  pm.check_status(B)
  for got, ref in zip(outs, want):
    assert torch.equal(got, ref)
  # and the stage API, which stays on the caller's stream, gives the same numbers
  flags = ops.make_flags(True, True)
  f1, cross, _ = ops.q_forward(pm, mu, S, flags)
  Sff = ops.Q_reduce_forward(pm, B, flags)
  assert torch.equal(f1, want[0]) and torch.equal(Sff, want[1]) and torch.equal(cross, want[2])


def test_two_caller_streams_share_nothing(device):
  """The side stream and its fork / join events belong to the CALLER'S stream (csrc/mm_fork.h; round 4 kept one triple per device
  under a process-wide mutex).  Two packed models with their own workspaces, driven from two streams: (i) interleaved eager calls
  give, bit for bit, what each gives alone; (ii) one stream under HIP-graph capture while the other keeps running eagerly -- the
  capture must neither pick up the other stream's events nor leak its own work -- and the replayed graph returns the eager result."""
  L, M, d = 6, 300, 8
  mods = []
  for seed, B in ((61, 32), (67, 40)):                           # P B = 672 / 840: both calls fork
    syn = make_svgp(L, M, d, seed=seed, device=str(device), ls_bounds=(0.5, 2.5))
    pm = syn.to_model(device).packed(torch.float32, True, device)
    mu, S = make_inputs(B, d, seed=seed + 1, scale=0.1, lo=0.2, hi=0.8)
    mu, S = to_dev(mu, device, torch.float32), to_dev(S, device, torch.float32)
    want = [t.clone() for t in ops.moment_match(pm, mu, S)]
    pm.check_status(B)
    mods.append((pm, mu, S, want, B))
  sa, sb = torch.cuda.Stream(device), torch.cuda.Stream(device)
  cur = torch.cuda.current_stream(device)
  sa.wait_stream(cur); sb.wait_stream(cur)
  # (i) interleaved eager calls from the two streams
  outs = [[], []]
  for _ in range(6):
    for k, st in enumerate((sa, sb)):
      pm, mu, S, _, _ = mods[k]
      with torch.cuda.stream(st):
        outs[k].append(ops.moment_match(pm, mu, S))
  torch.cuda.synchronize()
  for k in range(2):
    mods[k][0].check_status(mods[k][4])
    for o in outs[k]:
      for got, ref in zip(o, mods[k][3]):
        assert torch.equal(got, ref)
  # (ii) stream A under capture, stream B eager in the middle of it
  pmA, muA, SA, wantA, BA = mods[0]
  pmB, muB, SB, wantB, BB = mods[1]
  msA, SsA = muA.clone(), SA.clone()
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph, stream=sa, capture_error_mode="relaxed"):
    o1 = ops.moment_match(pmA, msA, SsA)
    with torch.cuda.stream(sb):                                  # not part of the capture: runs now
      eager_mid = ops.moment_match(pmB, muB, SB)
    o2 = ops.moment_match(pmA, msA, SsA)
  torch.cuda.synchronize()
  for got, ref in zip(eager_mid, wantB):
    assert torch.equal(got, ref)
  msA.copy_(muA * 0.9); SsA.copy_(SA * 1.2)
  graph.replay()
  msA.copy_(muA); SsA.copy_(SA)
  with torch.cuda.stream(sb):                                    # and while the graph replays
    eager_during = ops.moment_match(pmB, muB, SB)
  graph.replay()
  torch.cuda.synchronize()
  pmA.check_status(BA); pmB.check_status(BB)
  for got, ref in zip(o1, wantA):
    assert torch.equal(got, ref)
  for got, ref in zip(o2, wantA):
    assert torch.equal(got, ref)
  for got, ref in zip(eager_during, wantB):
    assert torch.equal(got, ref)


def test_a_forward_reduce_on_a_value_and_gradient_workspace_fails_loudly(device):
  """mm_moment_match_with_sums runs a q stage WITHOUT the forward-only part of the moment chain (csrc/mm_common.h:
  MM_ISTAGE_NO_M56: the value comes from the backward's sweeps) and poisons the degree-5/6 sums on the workspace: a forward reduce
  run on that workspace by mistake (the stage API lets a caller do it) must not return stale numbers silently -- every
  off-diagonal entry of a collapsed pair comes back NaN -- while the value-and-gradient call itself is unaffected."""
  L, M, d, B = 3, 300, 4, 4
  syn = make_svgp(L, M, d, seed=4242, device=str(device), ls_bounds=(0.7, 2.0))
  pm = syn.to_model(device).packed(torch.float32, True, device)
  mu, S = make_inputs(B, d, seed=11, scale=0.12, lo=0.3, hi=0.7)
  mu, S = to_dev(mu, device, torch.float32), to_dev(S, device, torch.float32)
  want = [t.clone() for t in ops.moment_match(pm, mu, S)]
  flags = ops.make_flags(True, True)
  collapsed, total, _ = ops.offdiag_stats(pm, B, flags)
  assert collapsed == total > 0                                   # (narrow state: every off-diagonal item is collapsed)
  f1, Sff, cross, sums, gen = ops.moment_match_with_sums(pm, mu, S)
  assert torch.isfinite(Sff).all()
  assert float((Sff - want[1]).abs().max()) < 2e-5 * float(want[1].abs().max())
  bad = ops.Q_reduce_forward(pm, B, flags)                        # the mistake
  off = ~torch.eye(L, dtype=torch.bool, device=device)
  assert torch.isnan(bad[:, off]).all() and torch.isfinite(torch.diagonal(bad, dim1=-2, dim2=-1)).all()
  # a fresh forward puts the workspace right again
  again = ops.moment_match(pm, mu, S)
  for got, ref in zip(again, want):
    assert torch.equal(got, ref)
