#!/usr/bin/env python
"""Benchmark of the moment-matched GP rollout (BASELINE.json metric) on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config c3|c2|c1]

One "step" = one rollout time-step of the whole local batch: q stage + fused Q reduce +
Euler moment update for B input distributions (state resets to (mu0, Sigma0) every H steps,
per-step expected costs are all-gathered after each H-step rollout).  value = B_total * K /
wall time ("rollout step-elements per second", B x H per rollout time).  Weak scaling: every
rank owns B_local = B input distributions; no data-path collective inside a rollout.

The JSON line also carries
  roofline     -- the dominant kernel (f32 MFMA off-diagonal reduce), HIP-event timed in the
                  timed region on the launch stream; algorithmic flops per SURVEY.md section 8d;
  cpu_baseline -- the literal fp64 CPU oracle (reference algorithm: materialised eKuffu +
                  triangular solves) timed on this host on a bounded sample (rank 0, N=1);
  parity       -- max abs error of one GPU step against that oracle on the same inputs.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gpflowpilco_amd import _lib, ops  # noqa: E402
from gpflowpilco_amd.cost import expected_gaussian_cost  # noqa: E402
from gpflowpilco_amd.synthetic import make_inputs, make_svgp  # noqa: E402

CONFIGS = {
    # name: (L, M, d, H, B_local, dtype, seed)   -- BASELINE.json configs[0..2]
    "c1": dict(L=6, M=100, d=6, H=30, B=1, dtype=torch.float64, seed=1000,
               label="C1-shaped closed rollout: N=100 d=6 D=6 H=30 B=1 fp64"),
    "c2": dict(L=5, M=1000, d=5, H=40, B=64, dtype=torch.float64, seed=1001,
               label="C2-shaped closed rollout: N=1000 d=5 D=5 H=40 B=64 fp64"),
    "c3": dict(L=8, M=2000, d=8, H=40, B=256, dtype=torch.float32, seed=1002,
               label="C3: N=2000 d=8 D=8 H=40 B per GPU fp32 closed drift rollout"),
}
# BASELINE.json configs[4] per GPU: S = 65536 / 8 sample paths, N = 2000, K = 1024 bases, H = 50
PATHWISE = dict(L=8, M=2000, d=8, K=1024, H=50, S=8192, dtype=torch.float32, seed=1004,
                label="C5 per-GPU shard: pathwise sample rollout S=8192 (65536/8) N=2000 K=1024 d=D=8 H=50 fp32")
PEAK_TFLOPS = {torch.float32: 157.3, torch.float64: 78.6}   # MI355X_MICROARCH.md dense MFMA peaks


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=80)
  ap.add_argument("--warmup", type=int, default=8)
  ap.add_argument("--config", default="c3", choices=sorted(CONFIGS) + ["c5"])
  ap.add_argument("--batch", type=int, default=None, help="override B per GPU")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--force-generic", action="store_true")
  ap.add_argument("--rehearse-gloo", action="store_true",
                  help="multi-process rehearsal on ONE GPU: gloo backend, every rank on cuda:0, costs gathered via host")
  return ap.parse_args()


def main():
  args = parse()
  rank = int(os.environ.get("RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a GPU (no CPU fallback)")
  if args.rehearse_gloo:
    local_rank = 0
  torch.cuda.set_device(local_rank)
  dev = torch.device("cuda", local_rank)
  dist = None
  if world > 1:
    import torch.distributed as dist
    if args.rehearse_gloo:
      dist.init_process_group("gloo")
    else:
      dist.init_process_group("nccl", device_id=dev)     # RCCL over xGMI

  if args.config == "c5":
    return pathwise_bench(args, rank, world, dev, dist)
  cfg = dict(CONFIGS[args.config])
  if args.batch:
    cfg["B"] = args.batch
  L, M, d, H, B, dtype = cfg["L"], cfg["M"], cfg["d"], cfg["H"], cfg["B"], cfg["dtype"]
  # lengthscales log-uniform in [0.7, 3] (not BASELINE.md's [0.3, 3]): with 0.3 the M points cannot
  # cover the d-dimensional cube, the predictive variance stays at the prior and the closed rollout's
  # covariance random-walks out of the data's support within ~5 steps (DESIGN.md "Synthetic workload")
  syn = make_svgp(L, M, d, seed=cfg["seed"], device=str(dev), ls_bounds=(0.7, 3.0))
  model = syn.to_model(dev)
  pm = model.packed(dtype, True, dev)
  mu0_np, S0_np = make_inputs(B, d, seed=2000 + rank, scale=0.1, lo=0.3, hi=0.7)
  mu0 = torch.tensor(mu0_np, dtype=dtype, device=dev)
  S0 = torch.tensor(S0_np, dtype=dtype, device=dev)
  target = torch.full((d,), 0.5, dtype=dtype, device=dev)
  precis = torch.eye(d, dtype=dtype, device=dev) * 4.0

  base = ops.make_flags(True, True, args.force_generic)
  F = _lib
  traj_mu = torch.empty(H, B, d, dtype=dtype, device=dev)
  traj_S = torch.empty(H, B, d, d, dtype=dtype, device=dev)
  gathered = [torch.empty(B, H, dtype=dtype, device=dev) for _ in range(world)] if world > 1 else None
  ev, evd = [], []
  state = {"mu": mu0.clone(), "S": S0.clone(), "h": 0, "cost": None}

  def one_step(timed):
    if state["h"] == 0:
      state["mu"], state["S"] = mu0.clone(), S0.clone()
    f1, cross, _ = ops.q_forward(pm, state["mu"], state["S"], base)
    if timed:
      d0 = torch.cuda.Event(enable_timing=True); d1 = torch.cuda.Event(enable_timing=True)
      e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
      d0.record()
    ops.Q_reduce_forward(pm, B, base | F.MM_STAGE_DIAG)
    if timed:
      d1.record(); e0.record()
    ops.Q_reduce_forward(pm, B, base | F.MM_STAGE_OFFDIAG)
    if timed:
      e1.record(); ev.append((e0, e1)); evd.append((d0, d1))
    Sff = ops.Q_reduce_forward(pm, B, base | F.MM_STAGE_FINALIZE)
    state["mu"], state["S"] = ops.euler_update(state["mu"], state["S"], f1, Sff, cross, 1.0)
    traj_mu[state["h"]].copy_(state["mu"]); traj_S[state["h"]].copy_(state["S"])
    state["h"] += 1
    if state["h"] == H:
      # per-step cost statistic of the finished rollout, [B_local, H] -> all ranks
      cost = expected_gaussian_cost(traj_mu, traj_S, target, precis).T.contiguous()
      if world > 1 and args.rehearse_gloo:
        host = [torch.empty(B, H, dtype=dtype) for _ in range(world)]
        dist.all_gather(host, cost.cpu())
        state["cost"] = torch.cat(host, 0).to(dev)
      elif world > 1:
        dist.all_gather(gathered, cost)
        state["cost"] = torch.cat(gathered, 0)
      else:
        state["cost"] = cost
      state["h"] = 0

  def fence():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    one_step(False)
  state["h"] = 0
  if world > 1 and not args.rehearse_gloo:
    # the warm-up steps do not reach the end of a rollout: bring up the collective's channels untimed
    dist.all_gather(gathered, torch.zeros(B, H, dtype=dtype, device=dev))
  fence()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    one_step(True)
  fence()
  elapsed = time.perf_counter() - t0
  if world > 1:
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_gloo else dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
  pm.check_status(B)
  if not torch.isfinite(state["S"]).all():
    raise SystemExit("non-finite state in the timed rollout")
  if state["cost"] is not None and tuple(state["cost"].shape) != (B * world, H):
    raise SystemExit(f"gathered cost matrix has shape {tuple(state['cost'].shape)}, expected {(B * world, H)}")

  # ---- roofline of the dominant kernel (the off-diagonal reduce) ----------------------------
  k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else float("nan")
  kd_ms = float(np.mean([a.elapsed_time(b) for a, b in evd])) if evd else float("nan")
  Po = L * (L - 1) // 2
  entries = float(B) * Po * M * M
  flops = entries * (2 * d + 12)                           # E * (2d + 12), E = B*Po*M^2
  achieved = flops / (k_ms * 1e-3) / 1e12 if Po else 0.0
  f32_mfma = dtype == torch.float32 and not args.force_generic
  kname = "k_qred_generic" if args.force_generic else ("k_qred_f32_mfma" if f32_mfma else "k_qred_f64_mfma")
  # HBM bytes per launch of the dominant kernel from the PMC passes kept in
  # profiles/r01_pmc_counters.csv (FETCH_SIZE x 2 per MI355X_MICROARCH.md "HBM" + WRITE_SIZE, KB):
  # 2 * 289494 KB + 448 KB = 0.59 GB against 0.59 GB of streamed operands (rowO + colO).
  # Only valid for the default C3 / B = 256 launch; other shapes report null.
  traffic = 2 * 289494 * 1024 + 448 * 1024 if (args.config == "c3" and B == 256 and f32_mfma) else None
  roofline = {"bound": "mfma", "kernel": kname,
              "achieved": round(achieved, 3), "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
              "frac": round(achieved / PEAK_TFLOPS[dtype], 4), "traffic": traffic,
              "kernel_ms": round(k_ms, 4),
              "flops_per_launch": flops, "entries_per_launch": entries}
  if traffic is not None:
    # from the same PMC passes (profiles/r01_pmc_counters.csv): matrix-pipe-busy cycles
    # (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) + VALU issue cycles (SQ_INSTS_VALU / 1024 x 3.6) over kernel cycles
    # (GRBM_GUI_ACTIVE / 8) = (1.89e6 + 5.2e6) / 8.43e6 -- the two serialise on a gfx950 SIMD
    roofline["pipe_busy_frac"] = 0.84
  if f32_mfma:
    # executed work (DESIGN.md "f32 reduce roofline"): the bilinear form runs as a bf16 split product,
    # 4 v_mfma_f32_32x32x16_bf16 per 64 x 32 wave tile and 8 input dims, 6 for tiles with max|b| > 1/32:
    # the figure below is the 6-MFMA upper bound
    nd8 = (d + 7) // 8
    mfma_flops = entries / 2048.0 * 6 * nd8 * 32768.0
    roofline["executed_mfma_bf16"] = {"achieved": round(mfma_flops / (k_ms * 1e-3) / 1e12, 1), "peak": 2500.0,
                                      "unit": "TFLOP/s", "frac": round(mfma_flops / (k_ms * 1e-3) / 2.5e15, 4),
                                      "bound": "upper (6 MFMAs per wave tile; 4 on tiles with max|b| <= 1/32)"}
    roofline["note"] = ("achieved = algorithmic f32 flops, E*(2d+12); the 2d part executes on the bf16 matrix pipe "
                        "(2-3 MFMAs per 8 dims); the tile kernel reduces the remainder "
                        "expm1(b)-b-b^2/2 with a range-adaptive polynomial (5..9 VALU ops per entry) while the constant, "
                        "linear and quadratic parts come from f64 weight moments at O(M d^2), so the fraction of the "
                        "f32 peak can exceed 1; MFMA and f32 FMA VALU time add on a SIMD (tools/ubench_overlap.hip)")
  # second kernel of the step: the diagonal pairs, always f64 (upper-triangular tiles)
  ed = float(B) * L * M * (M + 1) / 2
  roofline_diag = {"bound": "mfma", "kernel": "k_qred_generic" if args.force_generic else "k_qred_f64_mfma",
                   "achieved": round(ed * (2 * d + 12) / (kd_ms * 1e-3) / 1e12, 3), "peak": PEAK_TFLOPS[torch.float64],
                   "unit": "TFLOP/s", "frac": round(ed * (2 * d + 12) / (kd_ms * 1e-3) / 1e12 / PEAK_TFLOPS[torch.float64], 4),
                   "kernel_ms": round(kd_ms, 4), "entries_per_launch": ed}

  out = {
      "metric": "moment_matched_rollout_step_elements_per_sec",
      "value": round(B * world * args.steps / elapsed, 2),
      "unit": "rollout step-elements/s (B*H per rollout second)",
      "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": round(1e3 * elapsed / args.steps, 4),
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
      "dtype": "f32" if dtype == torch.float32 else "f64", "data": "synthetic",
      "config": {"workload": cfg["label"].replace("B per GPU", f"B={B} per GPU"), "N": M, "d": d, "D": L, "H": H,
                 "B_per_gpu": B, "B_total": B * world, "parallelism": f"dp{world} over B",
                 "diag_pairs": "f64", "offdiag_pairs": "f32" if dtype == torch.float32 else "f64"},
      "roofline": roofline,
      "roofline_diag": roofline_diag,
  }

  # ---- CPU baseline + parity (rank 0, N == 1 only) -----------------------------------------
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    from oracle import mm_oracle as mo
    Bc = 1 if M >= 1000 else min(B, 4)
    po = mo.SVGPParams(Z=np.broadcast_to(syn.Z, (L, M, d)).copy(), lengthscales=syn.lengthscales,
                       variance=syn.variance, q_mu=syn.q_mu, q_sqrt=syn.q_sqrt, whiten=True)
    mu_c, S_c = mu0_np[:Bc].copy(), S0_np[:Bc].copy()
    t0 = time.perf_counter()
    nst = 0
    while True:
      f1o, Sffo, cro = mo.mm_gauss_svgp_mo(mu_c, S_c, po)
      nst += 1
      if nst == 1:
        first = (f1o, Sffo, cro)
      Sxf = mo.cross_covariance(S_c, cro, True)
      mu_c, S_c = mo.euler_moment_update(mu_c, S_c, f1o, Sffo, Sxf, 1.0)
      if time.perf_counter() - t0 > 10.0 or nst >= H:
        break
    tc = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": round(Bc * nst / tc, 4), "unit": out["unit"],
                           "cores": os.cpu_count(), "kind": "port",
                           "sample": f"literal fp64 oracle (materialised eKuffu [B,L,M,L,M] + triangular solves, "
                                     f"numpy/OpenBLAS threads), B={Bc}, {nst} step(s) of the same rollout, {tc:.1f}s"}
    # second CPU figure: the algorithm-matched restatement -- beta / C hoisted out of the
    # step, O(M^2) per kernel pair -- so the ratio to the GPU is not merely the O(M^3) -> O(M^2) change
    from oracle import mm_fused_ref as fr
    beta_c, C_c = fr.precompute(po)                        # not timed: once per model, like mm_pack_model
    t0 = time.perf_counter()
    fm = fr.moment_match(mu0_np[:Bc], S0_np[:Bc], po, beta_c, C_c)
    tm = time.perf_counter() - t0
    out["cpu_baseline_matched"] = {"value": round(Bc / tm, 4), "unit": out["unit"], "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"algorithm-matched fp64 restatement (oracle/mm_fused_ref.py: numpy, O(M^2) per pair, "
                                             f"precompute excluded), B={Bc}, 1 step without the Euler update, {tm:.1f}s",
                                   "max_abs_diff_vs_literal": {"f1": float(np.abs(fm[0] - first[0]).max()),
                                                               "Sff": float(np.abs(fm[1] - first[1]).max())}}
    f1, Sff, cross = ops.moment_match(pm, mu0[:Bc].contiguous(), S0[:Bc].contiguous())
    err = lambda g, w: float(np.abs(g.double().cpu().numpy() - w).max())
    out["parity"] = {"vs": "fp64 CPU oracle, first step, same inputs", "B": Bc,
                     "max_abs_err": {"f1": err(f1, first[0]), "Sff": err(Sff, first[1]), "cross_pre": err(cross, first[2])},
                     "max_abs": {"f1": float(np.abs(first[0]).max()), "Sff": float(np.abs(first[1]).max()),
                                 "cross_pre": float(np.abs(first[2]).max())}}
    # rollout-level agreement (final mu_H, Sigma_H): the same kernels in f64 mode on 8 elements
    if dtype == torch.float32:
      Br = min(B, 8)
      pm64 = model.packed(torch.float64, True, dev)
      m32, S32 = ops.rollout_closed(pm, mu0[:Br].contiguous(), S0[:Br].contiguous(), H)
      m64, S64 = ops.rollout_closed(pm64, mu0[:Br].double().contiguous(), S0[:Br].double().contiguous(), H)
      pm64.check_status(Br)
      out["parity"]["rollout_f32_vs_f64_mode"] = {
          "B": Br, "H": H, "max_abs_diff": {"mu_H": float((m32.double() - m64).abs().max()),
                                            "Sigma_H": float((S32.double() - S64).abs().max())},
          "max_abs": {"mu_H": float(m64.abs().max()), "Sigma_H": float(S64.abs().max())}}
  if rank == 0:
    print(json.dumps(out))
  if world > 1:
    dist.destroy_process_group()


def pathwise_bench(args, rank, world, dev, dist):
  """configs[4]: one step = one Euler step of all local sample paths (HBM-bound weight stream)."""
  from gpflowpilco_amd.pathwise import PathwiseSVGP
  c = dict(PATHWISE)
  if args.batch:
    c["S"] = args.batch
  L, M, d, K, H, S, dtype = c["L"], c["M"], c["d"], c["K"], c["H"], c["S"], c["dtype"]
  syn = make_svgp(L, M, d, seed=c["seed"], device=str(dev), ls_bounds=(0.7, 3.0))
  base = syn.to_model(dev)
  model = PathwiseSVGP(kernel=base.kernel, inducing_variable=base.inducing_variable, q_mu=base.q_mu,
                       q_sqrt=base.q_sqrt, whiten=True, num_latent_gps=L)
  g = torch.Generator(device=dev).manual_seed(c["seed"] + rank)
  paths = model.generate_paths(S, K, dtype=dtype, device=dev, generator=g)
  x0 = 0.3 + 0.4 * torch.rand(S, d, dtype=dtype, device=dev, generator=g)
  target = torch.full((d,), 0.5, dtype=dtype, device=dev)

  def rollout_steps(n):
    x, traj = paths.rollout(x0, n, dt=1.0, keep_trajectory=True)
    err = traj - target
    return -torch.exp(-2.0 * (err * err).sum(-1)).T.contiguous()          # [S, n] per-step sample costs

  def fence():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  warm = rollout_steps(max(1, args.warmup))
  if world > 1:
    dist.all_gather([torch.empty_like(warm) for _ in range(world)], warm)   # untimed channel bring-up
  fence()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  done = 0
  e0.record()
  while done < args.steps:
    n = min(H, args.steps - done)
    cost = rollout_steps(n)
    if n == H and world > 1:
      out = [torch.empty_like(cost) for _ in range(world)]
      dist.all_gather(out, cost)
    done += n
  e1.record()
  fence()
  elapsed = time.perf_counter() - t0
  if world > 1:
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
  k_ms = e0.elapsed_time(e1) / args.steps
  bytes_per_launch = float(S) * L * (K + M) * 4
  achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
  out = {"metric": "pathwise_rollout_sample_steps_per_sec", "value": round(S * world * args.steps / elapsed, 1),
         "unit": "sample step-elements/s (S*H per rollout second)", "n_gpus": world, "steps": args.steps,
         "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
         "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
         "config": {"workload": c["label"], "N": M, "d": d, "D": L, "K": K, "H": H, "S_per_gpu": S,
                    "parallelism": f"dp{world} over S"},
         "roofline": {"bound": "hbm", "kernel": "k_pathwise", "achieved": round(achieved, 1), "peak": 8000.0,
                      "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": None,
                      "kernel_ms": round(k_ms, 4), "bytes_per_launch": bytes_per_launch,
                      "note": "kernel_ms includes the (small) cost kernels between launches"}}
  if rank == 0:
    print(json.dumps(out))
  if world > 1:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
