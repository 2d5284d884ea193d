"""Where the f32 error of the pathwise update term comes from (BASELINE configs[4]).

f_update(x) = sum_m v_m k(x, z_m),  v = (Kuu + 1e-6 I)^-1 (u - Phi w): |v| reaches 1e3 .. 1e5 while the sum is O(1).
This script rounds ONE factor at a time to f32 (numpy, fp64 reference): the weights v, the kernel values k, v as an
f32 + bf16 pair.  Result at M = 2000, d = 8 (seed 0; max |v| = 1.6e3):
    v in f32            4.6e-4      k in f32          4.0e-4
    v as f32 + bf16     1.3e-6      v as f32 + bf16 AND k in f32     4.0e-4
Splitting v alone does not help: the rounding of the KERNEL VALUES (which the f32 kernel computes with v_exp_f32 from an f32
argument) costs as much as the rounding of v, because both are multiplied by the same cancellation factor
sum |v_m k_m| / |sum v_m k_m| ~ cond.  An f32 mode below cond * 6e-8 needs the update term in f64 (arguments, exp and
weights: 1.66x the bytes of the stream and an f64 exp per term) -- which is the library's f64 mode.  The f32 tolerance of
tests/test_pathwise.py therefore stays at the measured conditioning floor (4e-2 at the C5 shard, |v| ~ 1e5).

  python tools/pathwise_f32_floor.py
"""
import numpy as np
import scipy.linalg as sl

rng = np.random.default_rng(0)
M, d, K, S = 2000, 8, 1024, 8
Z = rng.uniform(size=(M, d)); ls = np.exp(rng.uniform(np.log(0.7), np.log(3.0), size=d)); var = 0.89 ** 2
A = Z / ls
d2 = (A * A).sum(-1)[:, None] + (A * A).sum(-1)[None] - 2 * A @ A.T
Kuu = var * np.exp(-0.5 * np.clip(d2, 0, None)) + 1e-6 * np.eye(M)
Luu = np.linalg.cholesky(Kuu)
omega = rng.standard_normal((K, d)) / ls; phase = 2 * np.pi * rng.uniform(size=K); w = rng.standard_normal((S, K))
u = (Luu @ rng.standard_normal((M, S))).T * 0.3
PhiZ = np.sqrt(2 * var / K) * np.cos(Z @ omega.T + phase)
v = sl.cho_solve((Luu, True), (u - w @ PhiZ.T).T).T
x = rng.uniform(0.3, 0.7, size=(S, d))
kx = var * np.exp(-0.5 * (((x[:, None, :] - Z[None]) / ls) ** 2).sum(-1))
exact = (v * kx).sum(-1)
f32 = lambda a: a.astype(np.float32).astype(np.float64)
bf16 = lambda a: (a.astype(np.float32).view(np.uint32) & 0xffff0000).view(np.float32).astype(np.float64)
vh = f32(v); vl = bf16(v - vh)
err = lambda got: float(np.abs(got - exact).max())
print(f"max |v| = {np.abs(v).max():.3g}, |f_update| ~ {np.abs(exact).max():.2f}, cancellation sum|v k| / |sum v k| = {np.abs(v * kx).sum(-1).max() / np.abs(exact).max():.3g}")
print(f"v in f32                       {err((f32(v) * kx).sum(-1)):.2e}")
print(f"k in f32                       {err((v * f32(kx)).sum(-1)):.2e}")
print(f"v as f32 + bf16                {err(((vh + vl) * kx).sum(-1)):.2e}")
print(f"v as f32 + bf16, k in f32      {err(((vh + vl) * f32(kx)).sum(-1)):.2e}")
