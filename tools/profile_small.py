"""Stage profile of the one-launch rollout kernel (csrc/mm_rollout_small.hip) at cartpole sizes: cycles per stage of block 0,
from a -DMMS_PROFILE build of the library (GPFLOWPILCO_MM_LIB points at it).

  OUT=$PWD/scratch/variants/lib_prof.so OBJDIR=$PWD/scratch/variants/obj_prof bash gpflowpilco_amd/csrc/build.sh -DMMS_PROFILE
  GPFLOWPILCO_MM_LIB=$PWD/scratch/variants/lib_prof.so python tools/profile_small.py
"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflowpilco_amd import _lib, ops
from gpflowpilco_amd.synthetic import make_cartpole_like, make_inputs
dev = torch.device("cuda:0")
f64 = torch.float64
drift_s, pol_s = make_cartpole_like(100, 30, 1000, device=str(dev))
drift, pol = drift_s.to_model(dev), pol_s.to_model(dev)
t = lambda a: torch.tensor(np.asarray(a), dtype=f64, device=dev)
target = np.array([0.0, 1.0, 0.0, 0.0, 0.0]); precis = 4.0 * np.eye(5)
roll = ops.ComposedRollout(drift.packed(f64, True, dev), pol.packed(f64, False, dev), nx=4, active_dims=(1,), head_scale=2.0,
                           head_shift=-0.5, target=t(target), precis=t(precis))
mu = np.array([[0.4, 0.2, 0.5, 0.3]]); _, S = make_inputs(1, 4, seed=3000, scale=0.05)
mx, Sx = t(mu), t(S)
H = 30
lib = _lib.lib()
prof = torch.zeros(16, dtype=torch.int64, device=dev)
if hasattr(lib, "mm_rollout_small_set_profile"):
  lib.mm_rollout_small_set_profile.argtypes = [ctypes.c_void_p]
  lib.mm_rollout_small_set_profile(prof.data_ptr())
for eng in ("small", "multi"):
  roll(mx, Sx, H, engine=eng); torch.cuda.synchronize()
  prof.zero_()
  t0 = time.perf_counter()
  for _ in range(20): roll(mx, Sx, H, engine=eng)
  torch.cuda.synchronize()
  print(eng, "us/step", (time.perf_counter() - t0) / 20 / H * 1e6)
  if eng == "small":
    p = prof.cpu().numpy() / (20 * H)
    names = ["gp:setup", "gp:dxd items", "gp:latent-centre", "gp:pair phaseA", "gp:pair sweep", "gp:pair reduce", "head", "step", "encode", "cost"]
    for n, v in zip(names, p): print(f"  {n:18s} {v:10.0f} cycles/step")
    print("  total", p.sum(), "cycles/step (policy + drift gp calls share the gp:* ids)")
