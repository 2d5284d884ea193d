"""The pathwise stream pass at the C5 shape, plain and with the Jacobian (mm_pathwise_eval / _eval_jac), a few times over -- the
program `rocprofv3 --pmc ...` / `--kernel-trace --stats` is run on to read the two kernels' durations and instruction mix
(what bounds the taped pass).

  python3 tools/pathwise_jac_run.py [--reps 5] [--S 65536]
"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflowpilco_amd import models as gp
from gpflowpilco_amd.pathwise import PathwiseSVGP
from gpflowpilco_amd.synthetic import make_svgp

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--S", type=int, default=65536)
a = ap.parse_args()
dev = torch.device("cuda", 0)
L = d = 8; M = 2000; K = 1024
syn = make_svgp(L, M, d, seed=1004, device=str(dev), ls_bounds=(0.7, 3.0))
base = syn.to_model(dev)
pm = PathwiseSVGP(kernel=base.kernel, inducing_variable=base.inducing_variable, q_mu=base.q_mu, q_sqrt=base.q_sqrt, whiten=True,
                  num_latent_gps=L)
g = torch.Generator(device=dev).manual_seed(3)
paths = pm.generate_paths(a.S, K, dtype=torch.float32, device=dev, generator=g)
x = torch.rand(a.S, d, device=dev, dtype=torch.float32, generator=g) * 0.4 + 0.3
for _ in range(a.reps):
  f = paths(x)
  f2, jac = paths.eval_jac(x)
  f3, err = paths.eval_with_bound(x)
torch.cuda.synchronize()
print("ok", float(f.abs().max()), float(jac.abs().max()), float(err.max()))
