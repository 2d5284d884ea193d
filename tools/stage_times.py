"""Per-stage GPU times of one C3 moment-matching step on frozen inputs (HIP events, torch stream).

  python tools/stage_times.py [--reps 20] [--batch 256] [--scale 0.1]

Frozen inputs: the state is not advanced, so kernel ablation builds (which produce wrong
Sff) can be timed without the rollout diverging.
"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflowpilco_amd import _lib as F, ops
from gpflowpilco_amd.synthetic import make_inputs, make_svgp

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--scale", type=float, default=0.1)
ap.add_argument("--L", type=int, default=8); ap.add_argument("--M", type=int, default=2000); ap.add_argument("--d", type=int, default=8)
ap.add_argument("--f64", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0); dt = torch.float64 if a.f64 else torch.float32
syn = make_svgp(a.L, a.M, a.d, seed=0, device=str(dev), ls_bounds=(0.7, 3.0))
pm = syn.to_model(dev).packed(dt, True, dev)
mu, S = make_inputs(a.batch, a.d, seed=2000, scale=a.scale, lo=0.3, hi=0.7)
mu = torch.tensor(mu, dtype=dt, device=dev); S = torch.tensor(S, dtype=dt, device=dev)
base = ops.make_flags(True, True, False)
stages = [("q_forward", lambda: ops.q_forward(pm, mu, S, base)),
          ("diag", lambda: ops.Q_reduce_forward(pm, a.batch, base | F.MM_STAGE_DIAG)),
          ("offdiag", lambda: ops.Q_reduce_forward(pm, a.batch, base | F.MM_STAGE_OFFDIAG)),
          ("finalize", lambda: ops.Q_reduce_forward(pm, a.batch, base | F.MM_STAGE_FINALIZE))]
for _ in range(3):
  for _, f in stages: f()
torch.cuda.synchronize()
tot = {n: 0.0 for n, _ in stages}
for _ in range(a.reps):
  for n, f in stages:
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    tot[n] += e0.elapsed_time(e1)
print(" ".join(f"{n}={tot[n] / a.reps:.3f}ms" for n in tot), f"sum={sum(tot.values()) / a.reps:.3f}ms")

