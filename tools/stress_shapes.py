"""Randomised shape sweep on the GPU: f32 mode vs f64 mode vs the portable kernels of the same library,
over odd sizes (M not a multiple of anything, B = 1.., L = 1.., d = 1..12, or up to argv[3]).  Prints the worst case.

  python tools/stress_shapes.py [seed] [cases] [max_d]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflowpilco_amd import ops
from gpflowpilco_amd.synthetic import make_inputs, make_svgp

dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dmax = int(sys.argv[3]) if len(sys.argv) > 3 else 12
worst = {"f32_vs_f64": (0.0, None), "f64_vs_generic": (0.0, None), "f32_vs_generic": (0.0, None)}
for it in range(n):
  L = int(rng.integers(1, 6)); M = int(rng.integers(1, 400)); d = int(rng.integers(1, dmax + 1)); B = int(rng.integers(1, 24))
  scale = float(rng.choice([0.02, 0.1, 0.3, 0.8]))
  syn = make_svgp(L, M, d, seed=int(rng.integers(1 << 30)), device=str(dev), ls_bounds=(0.5 + 0.1 * d, 1.5 + 0.3 * d))
  model = syn.to_model(dev)
  mu, S = make_inputs(B, d, seed=int(rng.integers(1 << 30)), scale=scale, lo=0.2, hi=0.8)
  out = {}
  for dt in (torch.float64, torch.float32):
    pm = model.packed(dt, True, dev)
    m_t, S_t = torch.tensor(mu, dtype=dt, device=dev), torch.tensor(S, dtype=dt, device=dev)
    out[dt] = [t.double() for t in ops.moment_match(pm, m_t, S_t)]
    out[(dt, "g")] = [t.double() for t in ops.moment_match(pm, m_t, S_t, force_generic=True)]
    pm.check_status(B)
  def err(a, b):
    return max(float((x - y).abs().max() / max(float(y.abs().max()), 1e-30)) for x, y in zip(a, b))
  for key, val in (("f32_vs_f64", err(out[torch.float32], out[torch.float64])),
                   ("f64_vs_generic", err(out[torch.float64], out[(torch.float64, "g")])),
                   ("f32_vs_generic", err(out[torch.float32], out[(torch.float32, "g")]))):
    if not np.isfinite(val) or val > worst[key][0]:
      worst[key] = (val, (L, M, d, B, scale))
for k, v in worst.items():
  print(f"{k:16s} worst {v[0]:.2e} at (L, M, d, B, scale) = {v[1]}")
