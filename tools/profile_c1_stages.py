"""Stage profile of the two slowest kernels of the cartpole step (k_compose_tail, k_policy_match_small) from a
-DMM_STAGE_PROFILE build:  cycles (clock64, 2.4 GHz) per stage of block 0, averaged over the steps of 20 rollouts.

  OUT=$PWD/scratch/variants/lib_stage.so OBJDIR=$PWD/scratch/variants/obj_stage bash gpflowpilco_amd/csrc/build.sh -DMM_STAGE_PROFILE
  GPFLOWPILCO_MM_LIB=$PWD/scratch/variants/lib_stage.so python tools/profile_c1_stages.py
"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflowpilco_amd import _lib, ops
from gpflowpilco_amd.synthetic import make_cartpole_like, make_inputs
dev = torch.device("cuda:0"); f64 = torch.float64
drift_s, pol_s = make_cartpole_like(100, 30, 1000, device=str(dev))
drift, pol = drift_s.to_model(dev), pol_s.to_model(dev)
t = lambda a: torch.tensor(np.asarray(a), dtype=f64, device=dev)
roll = ops.ComposedRollout(drift.packed(f64, True, dev), pol.packed(f64, False, dev), nx=4, active_dims=(1,), head_scale=2.0,
                           head_shift=-0.5, target=t([0.0, 1.0, 0, 0, 0]), precis=t(4.0 * np.eye(5)))
mx = t([[0.4, 0.2, 0.5, 0.3]]); _, S = make_inputs(1, 4, seed=3000, scale=0.05); Sx = t(S)
H, R = 30, 20
lib = _lib.lib()
prof = torch.zeros(16, dtype=torch.int64, device=dev)
lib.mm_stage_profile_set.argtypes = [ctypes.c_void_p]
lib.mm_stage_profile_set(prof.data_ptr())
roll(mx, Sx, H); torch.cuda.synchronize(); prof.zero_()
for _ in range(R): roll(mx, Sx, H)
torch.cuda.synchronize()
p = prof.cpu().numpy() / (R * H)
names = {0: "tail: bookkeeping + Euler", 1: "tail: encode", 2: "tail: cost", 4: "policy: load Sigma", 5: "policy: 2 SPD inverses",
         6: "policy: E, T, G, logs", 7: "policy: per centre", 8: "policy: M x M sweep", 9: "policy: reductions + outputs"}
for k, n in names.items(): print(f"  {n:30s} {p[k]:9.0f} cycles = {p[k] / 2400:6.2f} us")

# ---- backward: k_policy_head_bwd_small (stage ids of mma_policy_small_bwd) ------------------------------------------------
if hasattr(lib, "mm_stage_profile_set_bwd"):
  from gpflowpilco_amd.autodiff import ComposedRolloutFunction
  prof2 = torch.zeros(16, dtype=torch.int64, device=dev)
  lib.mm_stage_profile_set_bwd.argtypes = [ctypes.c_void_p]
  lib.mm_stage_profile_set_bwd(prof2.data_ptr())
  m_H, S_H, cost, tape = roll.taped(mx, Sx, H)
  g = torch.ones(H, 1, dtype=f64, device=dev)
  roll.backward(tape, g, 1, H); torch.cuda.synchronize(); prof2.zero_()
  for _ in range(R): roll.backward(tape, g, 1, H)
  torch.cuda.synchronize()
  p2 = prof2.cpu().numpy() / (R * H)
  names2 = {8: "head adjoint", 0: "policy bwd: load + setup", 1: "policy bwd: 2 SPD inverses", 2: "policy bwd: E, T, G, sb", 3: "policy bwd: per centre (forward)",
            4: "policy bwd: M x M sweep", 5: "policy bwd: per-centre adjoints", 6: "policy bwd: sums over centres", 7: "policy bwd: d x d adjoint algebra", 9: "policy bwd: parameter gradient"}
  for k, n in names2.items(): print(f"  {n:36s} {p2[k]:9.0f} cycles = {p2[k] / 2400:6.2f} us")
