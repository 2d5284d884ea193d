"""Timings of the SURVEY section 8(f) rows on one MI355X, one JSON line each (profiles/r01_next_rows.jsonl):

  f-1  forward and forward+backward of one moment match at C1-, C2- and C3-shaped sizes
  f-2  the cartpole-sized composed policy loss (encoder -> policy -> drift -> Euler -> cost), eager and
       replayed from HIP graphs (loops.GraphedPolicyLoss), forward and forward+backward
  C4   one per-GPU shard of BASELINE configs[3] (B=32, d=16, L=32, N=4000, fp32): stage times of one match

  python tools/bench_next_rows.py [--skip-c4] [--skip-composed]
"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpflowpilco_amd import _lib as F, ops, bijectors as tfb, dynamics, models as gp
from gpflowpilco_amd.autodiff import moment_match_differentiable
from gpflowpilco_amd.components import GaussianObjective, TrigonometricEncoder
from gpflowpilco_amd.loops import GraphedPolicyLoss, get_state_initializer, policy_loss_closure
from gpflowpilco_amd.synthetic import make_inputs, make_svgp

ap = argparse.ArgumentParser()
ap.add_argument("--skip-c4", action="store_true")
ap.add_argument("--skip-composed", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)


def timed(fn, n):
  fn(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / n * 1e3


def emit(**kw):
  print(json.dumps(kw), flush=True)


# ---- f-1: backward of one match ---------------------------------------------------------------
for (name, L, M, d, B, dt) in (("C1-shaped", 4, 100, 6, 1, torch.float64), ("C2-shaped", 4, 1000, 5, 64, torch.float64),
                               ("C3-shaped B=32", 8, 2000, 8, 32, torch.float32)):
  syn = make_svgp(L, M, d, seed=1, device=str(dev), ls_bounds=(0.7, 3.0))
  model = syn.to_model(dev)
  mu, S = make_inputs(B, d, seed=5, scale=0.1, lo=0.3, hi=0.7)

  def run(grad):
    mu_t = torch.tensor(mu, dtype=dt, device=dev).requires_grad_(grad)
    S_t = torch.tensor(S, dtype=dt, device=dev).requires_grad_(grad)
    f1, Sff, cr = moment_match_differentiable(model, mu_t, S_t, True, True)
    if grad: (f1.sum() + Sff.sum() + cr.sum()).backward()
  emit(row="f-1 moment match", shape=name, L=L, M=M, d=d, B=B, dtype=str(dt).split(".")[-1],
       forward_ms=round(timed(lambda: run(False), 5), 3), forward_backward_ms=round(timed(lambda: run(True), 5), 3))
  del model, syn

# ---- f-2: composed cartpole-sized policy loss ---------------------------------------------------
from tests.helpers import gp_model_from_oracle, to_dev
from tests.test_compose import _cartpole_like
F64 = torch.float64
drift_o, pol_o, mu, S, target, precis = _cartpole_like()
H = 30
drift = gp_model_from_oracle(drift_o, dev)
pol_model = gp_model_from_oracle(pol_o, dev)
pol_model.q_mu = pol_model.q_mu.clone().requires_grad_(True)
policy = gp.InverseLinkWrapper(gp.KernelRegressor(pol_model),
                               invlink=tfb.Chain([tfb.Scale(2.0), tfb.Shift(-0.5), tfb.NormalCDF()]))
system = dynamics.DynamicalSystem(drift=drift, policy=policy, encoder=TrigonometricEncoder(active_dims=(1,)),
                                  solver=dynamics.MomentMatchingEuler())
objective = GaussianObjective(target=to_dev(target, dev, F64), precis=to_dev(precis, dev, F64))
init = get_state_initializer(to_dev(mu[:1], dev, F64), to_dev(S[:1] * 0.04, dev, F64))
for label, native in (() if args.skip_composed else (("native (taped rollout + mm_rollout_composed_backward)", None),
                                                    ("torch composition (native=False)", False))):
  closure = policy_loss_closure(system, objective, init, H, native=native)

  def eager_fwd():
    with torch.no_grad(): closure()

  def eager_bwd():
    pol_model.q_mu.grad = None
    closure().sum().backward()

  e_f, e_b = timed(eager_fwd, 3), timed(eager_bwd, 3)
  graphed = GraphedPolicyLoss(closure, [pol_model.q_mu])
  g_f, g_b = timed(graphed.loss, 5), timed(graphed.loss_and_grad, 5)
  emit(row="f-2 composed policy loss", path=label, shape="cartpole-sized (x4 -> e5 -> u1 -> d6 -> dx4), B=1, fp64", H=H,
       eager_forward_ms_per_step=round(e_f / H, 4), eager_forward_backward_ms_per_step=round(e_b / H, 4),
       graph_forward_ms_per_step=round(g_f / H, 4), graph_forward_backward_ms_per_step=round(g_b / H, 4))

# ---- C4 shard ---------------------------------------------------------------------------------
if not args.skip_c4:
  L, M, d, B = 32, 4000, 16, 32
  syn = make_svgp(L, M, d, seed=1003, device=str(dev), ls_bounds=(1.0, 3.0))
  pm = syn.to_model(dev).packed(torch.float32, True, dev)
  mu, S = make_inputs(B, d, seed=2000, scale=0.1, lo=0.3, hi=0.7)
  mu = torch.tensor(mu, dtype=torch.float32, device=dev); S = torch.tensor(S, dtype=torch.float32, device=dev)
  f1, Sff, cr = ops.moment_match(pm, mu, S); torch.cuda.synchronize(); pm.check_status(B)
  base = ops.make_flags(True, True, False)
  st = {}
  for nm, fn in (("q_stage", lambda: ops.q_forward(pm, mu, S, base)),
                 ("diag_f64", lambda: ops.Q_reduce_forward(pm, B, base | F.MM_STAGE_DIAG)),
                 ("offdiag_f32", lambda: ops.Q_reduce_forward(pm, B, base | F.MM_STAGE_OFFDIAG)),
                 ("finalize", lambda: ops.Q_reduce_forward(pm, B, base | F.MM_STAGE_FINALIZE))):
    st[nm + "_ms"] = round(timed(fn, 3), 3)
  emit(row="C4 per-GPU shard, one moment match", L=L, M=M, d=d, B=B, dtype="float32",
       packed_model_GiB=round(pm.nbytes / 2 ** 30, 2), total_ms=round(sum(st.values()), 3), **st,
       Sff_finite=bool(torch.isfinite(Sff).all()), min_eig=float(torch.linalg.eigvalsh(Sff.double()).min()))
