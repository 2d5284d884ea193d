"""The q stage of one C3 step on the BASELINE recipe, stage API (no side stream: the kernels run one after another), a few times
over -- run under `rocprofv3 --kernel-trace --stats` to read every kernel's STAND-ALONE duration (in the fused call the moment chain
runs beside the diagonal sweep and the trace shows contended times).  GPFLOWPILCO_MM_LIB selects an ablation build.

  rocprofv3 --kernel-trace --stats -d out -o t --output-format csv -- python3 tools/q_stage_kernels.py [--reps 10] [--recipe baseline]
"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflowpilco_amd import _lib as F, ops
from gpflowpilco_amd.synthetic import make_inputs, make_svgp

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--recipe", default="baseline", choices=["baseline", "pilco"])
ap.add_argument("--reduce", action="store_true", help="also run the two sweeps (stage calls)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
L = d = 8; M = 2000
kw = dict(ls_bounds=(0.3, 3.0), stable=False) if a.recipe == "baseline" else dict(ls_bounds=(0.7, 3.0), stable=True)
lo, hi = (0.0, 1.0) if a.recipe == "baseline" else (0.3, 0.7)
syn = make_svgp(L, M, d, seed=1002, device=str(dev), **kw)
pm = syn.to_model(dev).packed(torch.float32, True, dev)
mu, S = make_inputs(a.batch, d, seed=3002, scale=0.1, lo=lo, hi=hi)
mu = torch.tensor(mu, dtype=torch.float32, device=dev); S = torch.tensor(S, dtype=torch.float32, device=dev)
base = ops.make_flags(True, True, False)
for _ in range(a.reps):
  ops.q_forward(pm, mu, S, base)
  if a.reduce:
    ops.Q_reduce_forward(pm, a.batch, base | F.MM_STAGE_DIAG)
    ops.Q_reduce_forward(pm, a.batch, base | F.MM_STAGE_OFFDIAG)
    ops.Q_reduce_forward(pm, a.batch, base | F.MM_STAGE_FINALIZE)
torch.cuda.synchronize()
print("collapsed, total, inside, routed:", ops.offdiag_stats(pm, a.batch, base))
