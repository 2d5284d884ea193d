#!/usr/bin/env python
"""Benchmark of the moment-matched GP rollout (BASELINE.json metric) on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config c1|c2|c3|c4|c5] [--scaling weak|strong]
                  [--recipe pilco|baseline|worst]

One "step" = one pass of the hot path over the local batch: q stage + fused Q reduce (+ Euler moment update
where the state is closed, d == L).  Per-step expected costs are evaluated and all-gathered once per H-step
rollout, and once more for the partial rollout at the end of the timed region when K is not a multiple
of H: EXACTLY K steps are timed, and the cost kernel and the collective are always inside the timed region.
value = B_total * K / wall time ("rollout step-elements per second").

Configurations (BASELINE.json configs; `config.workload` names what ran):
  c1  configs[0]: the cartpole wiring x(4) -> encoder -> policy SVGP(30) + NormalCDF head -> drift SVGP(N=100, d=6 -> D=4)
      -> Euler -> cost, H=30, B=1, fp64: the whole rollout in mm_rollout_composed    (weak)
  c1_closed  the same sizes as a closed drift-only rollout (d=D=6)                   (weak)
  c2  configs[1] shaped: N=1000, d=D=5, H=40, B=64, fp64, closed rollout      (weak)
  c3  configs[2]: N=2000, d=D=8, H=40, B=256 per GPU, fp32                    (weak; THE metric's config)
  c4  configs[3]: N=4000, d=16, D=32, H=50, B=256 sharded over the ranks, fp32; d != D, so the step kernel runs on
      H independent (mu, Sigma) draws per rollout                 (strong)
  c5  configs[4]: pathwise sample rollout, S=65536 sharded over the ranks, N=2000, K=1024, H=50, fp32 (strong)
Recipes (the reduce kernels choose range tiers per tile, so their time depends on the data):
  baseline  BASELINE.md's own recipe (SURVEY 8d "Synthetic inputs"): lengthscales log-uniform [0.3, 3], GP-prior
            targets; a closed dt = 1 rollout leaves the support within a few steps there, so every step takes a
            fresh (mu, Sigma) draw (mu ~ U[0,1]^d, Sigma std 0.1) -- the step kernel on H independent draws, as
            SURVEY 8d allows.  Default for c3 and c4: `value` is measured on it;
  pilco     lengthscales log-uniform [0.7, 3], contracting targets: the state stays inside the data's support for
            the whole closed rollout -- default for c1_closed, c2;
  worst     the pilco data with MM_FORCE_WORST_TIER: every tile takes its most expensive tier.
The default N = 1 run of c3 times all three in one process (each with the same --steps / --warmup) and reports them
under `regimes`; `value`, `ms_per_step`, `segments_ms`, `roofline`, `parity`, `cpu_baseline` describe the baseline one.

Output: rank 0 prints ONE compact strict-JSON line (< 6000 bytes: the contract keys, `config`, `roofline` with numbers only,
`cpu_baseline`, `parity` numbers, `segments_ms`, per regime {value, ms_per_step, kernel_ms, frac}) and writes the full result --
everything below, with its definitions -- to bench_detail.json (and gpurun_out/bench_detail.json when that directory exists).

The full result carries
  (timing: the K timed steps call the product's entry point, ONE mm_moment_match per step -- its q stage overlaps the
  off-diagonal operands and the moment chain with the diagonal sweep on a side stream; `segments_ms` and the roofline's kernel
  time come from a separate pass of the SAME kernels run stage by stage through the stage API, where nothing overlaps, so
  sum(segments_ms) >= ms_per_step by what the overlap hides)
  roofline     -- the kernel with the largest measured share of the step, HIP-event timed in that staged pass on
                  the launch stream: `achieved` = SURVEY 8d ALGORITHMIC flops per launch (E (2d + 12)) / that time,
                  `frac` = achieved / dense peak of the dtype (null with a reason where the algorithmic rate exceeds
                  the peak, i.e. the launch does not execute 8d's per-entry work); `issue_frac` = the kernel's OWN
                  executed instruction mix (hardware counters of the same workload, profiles/r*_pmc_<config>_<recipe>.json,
                  collected by tools/collect_pmc.sh and accepted only if taken on the kernel sources now in the tree)
                  priced with the issue costs of tools/ubench_gap.hip, over the measured time -- a diagnostic, never `frac`;
  roofline_step -- the same 8d flop model over the whole step (cross-check);
  cpu_baseline -- the literal fp64 CPU oracle (reference algorithm: materialised eKuffu + triangular solves) timed on
                  this host on a bounded sample (rank 0, N=1);
  parity       -- max abs error of one GPU step against that oracle on the same inputs (+ per regime: against the
                  algorithm-matched fp64 restatement, and f32 mode vs f64 mode of the same kernels).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs[0]: the cartpole example's own wiring, run by mm_rollout_composed (csrc/mm_compose.hip)
CARTPOLE = dict(M=100, Mpol=30, H=30, B=1, dtype="f64", seed=1000, scaling="weak",
                label="C1 (BASELINE configs[0]): cartpole_swingup wiring x(4) -> trig encoder e(5) -> policy SVGP(M=30) + "
                      "Chain[Scale,Shift,NormalCDF] u(1) -> drift SVGP(N=100, d=6 -> D=4) -> Euler -> cost; H=30, B=1, fp64")
CONFIGS = {
    "c1_closed": dict(L=6, M=100, d=6, H=30, B=1, dtype="f64", seed=1000, scaling="weak", recipe="pilco", closed=True,
               label="C1-shaped closed drift rollout: N=100 d=6 D=6 H=30 B=1 fp64 (no encoder / policy)"),
    # BASELINE configs[1] AS STATED: d = 5 -> D = 4 has no closed rollout (state dim != input dim): the step kernel on H independent
    # draws, as C4 (SURVEY 8d "Synthetic inputs").  c2_closed is the D = 5 closed rollout rounds 1-4 ran under the name c2
    "c2": dict(L=4, M=1000, d=5, H=40, B=64, dtype="f64", seed=1001, scaling="weak", recipe="pilco", closed=False,
               label="C2 (BASELINE configs[1]): N=1000 d=5 D=4 H=40 B=64 fp64, step kernel on H independent draws"),
    "c2_closed": dict(L=5, M=1000, d=5, H=40, B=64, dtype="f64", seed=1001, scaling="weak", recipe="pilco", closed=True,
               label="C2-shaped closed drift rollout: N=1000 d=5 D=5 H=40 B=64 fp64 (state dim = input dim)"),
    # the headline config: `value` is BASELINE.md's own recipe (SURVEY 8d "Synthetic inputs"); the default N = 1 run also
    # times the two other data regimes of the range-tiered reduce kernels and reports them under `regimes`
    "c3": dict(L=8, M=2000, d=8, H=40, B=256, dtype="f32", seed=1002, scaling="weak", recipe="baseline", closed=True,
               also=("pilco", "worst"),
               label="C3 (BASELINE configs[2]): N=2000 d=8 D=8 H=40 fp32 drift moment-match step / closed rollout"),
    "c4": dict(L=32, M=4000, d=16, H=50, B=256, dtype="f32", seed=1003, scaling="strong", recipe="baseline", closed=False,
               label="C4 (BASELINE configs[3]): N=4000 d=16 D=32 H=50 B=256 total fp32, step kernel on H independent draws"),
}
# BASELINE.json configs[4]: S = 65536 sample paths in total, N = 2000, K = 1024 bases, H = 50
PATHWISE = dict(L=8, M=2000, d=8, K=1024, H=50, S=65536, dtype="f32", seed=1004, scaling="strong",
                label="C5 (BASELINE configs[4]): pathwise sample rollout S=65536 total N=2000 K=1024 d=D=8 H=50 fp32")
# SURVEY 8(f) row f-1 at C3 shape: one step = forward + backward (vector-Jacobian product w.r.t. mu, Sigma) of the C3 moment
# match -- what one rollout step costs when the rollout is differentiated (the reference: tf.GradientTape through
# moment_matching/models.py:200-299, utils/optimizers.py:51-56).  Not BASELINE.json's metric: a next-row line.
GRAD = dict(L=8, M=2000, d=8, B=256, dtype="f32", seed=1002, scaling="weak", recipe="baseline",
            label="C3-shaped forward + backward of one moment match (row f-1): N=2000 d=8 D=8 fp32 model, full output covariance, "
                  "model uncertainty; gradient of a fixed linear functional of (f1, Sff, cross) w.r.t. (mu, Sigma)")
RECIPES = {
    "pilco": dict(ls_bounds=(0.7, 3.0), stable=True, independent=False, worst=False,
                  text="lengthscales log-U[0.7,3], targets -0.5(z_a-0.5)+0.25*prior draw (state stays in the data's support)"),
    "baseline": dict(ls_bounds=(0.3, 3.0), stable=False, independent=True, worst=False,
                     text="BASELINE.md recipe: lengthscales log-U[0.3,3], GP-prior targets, fresh mu~U[0,1]^d / Sigma std 0.1 draw per step"),
    "worst": dict(ls_bounds=(0.7, 3.0), stable=True, independent=False, worst=True,
                  text="pilco data with MM_FORCE_WORST_TIER: every tile of both reduce kernels takes its most expensive tier"),
}
# MI355X_MICROARCH.md: dense peaks and the peak clock; per-SIMD issue costs measured by tools/ubench_gap.hip /
# tools/ubench_rates.hip on MI355X (profiles/r02_ubench_gap.txt): cycles per wave-instruction on one SIMD
PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6, "bf16": 2500.0}
PEAK_HBM_GBS = 8000.0
PEAK_CLOCK_HZ = 2.4e9
N_SIMD = 1024
ISSUE = {"mfma_bf16_32x32x16": 32.0, "mfma_f64_16x16x4": 64.0, "valu_f32": 4.0, "valu_f64": 5.0, "valu_other": 4.0,
         "valu_trans": 8.0}


# ---- the ONE line the driver parses ----------------------------------------------------------
# Round 4's line grew to 22 KB (three regimes x four roofline blocks with prose definitions) and the driver's record came
# back `parsed: null`.  Now: the full result goes to bench_detail.json (all definitions: DESIGN.md section 4.4), stdout gets
# one compact strict-JSON line, hard-limited in size (tests/test_bench_helpers.py).
LINE_LIMIT = 6000
DETAIL_FILE = "bench_detail.json"
_ROOF_KEYS = ("bound", "kernel", "kernel_ms", "achieved", "peak", "unit", "frac", "traffic", "issue_frac",
              "issue_frac_at_measured_clock", "mfma_busy_frac", "measured_clock_ghz", "entries_per_launch",
              "algorithmic_flops_per_launch", "bytes_per_launch", "executed_mfma_frac_of_bf16_peak")
_CFG_DROP = ("value_is", "offdiag_items_one_rollout", "policy_gradient", "one_sweep_value_and_gradient", "launch", "row")


def _finite(x):
  """Strict JSON has no NaN / Infinity: non-finite floats become null, everything else passes through."""
  if isinstance(x, float):
    return x if x == x and x not in (float("inf"), float("-inf")) else None
  if isinstance(x, dict):
    return {str(k): _finite(v) for k, v in x.items()}
  if isinstance(x, (list, tuple)):
    return [_finite(v) for v in x]
  if hasattr(x, "item") and not isinstance(x, (str, bytes)):      # numpy / torch scalars
    try:
      return _finite(x.item())
    except Exception:
      return str(x)
  return x


def _sig(x, n=6):
  return float(f"{x:.{n}g}") if isinstance(x, float) else x


def _short(s, n):
  s = str(s)
  return s if len(s) <= n else s[:n - 1] + "~"


_DROP = object()


def _numbers_only(x, depth=0):
  """The numeric leaves of a parity block (strings and prose dropped)."""
  if isinstance(x, dict):
    out = {}
    for k, v in x.items():
      w = _numbers_only(v, depth + 1)
      if w is not _DROP and w != {}:
        out[k] = w
    return out
  if isinstance(x, bool) or x is None:                  # null = a non-finite number (_finite): it stays visible
    return x
  if isinstance(x, (int, float)):
    return _sig(x, 4)
  return _DROP


def compact_roofline(r):
  if not isinstance(r, dict):
    return None
  c = {k: _sig(r[k]) for k in _ROOF_KEYS if k in r}
  if "kernel" in c:
    c["kernel"] = _short(c["kernel"], 72)
  if isinstance(r.get("pipes"), dict):
    c["pipes"] = {k: v.get("frac") for k, v in r["pipes"].items() if isinstance(v, dict)}
  if r.get("pmc"):
    c["pmc"] = _short(r["pmc"], 64)
  return c


def compact_line(out):
  """The compact dict printed on stdout: contract keys, `config` (sizes + recipe), `roofline` (numbers only), `cpu_baseline`,
  `parity` (numbers only), `segments_ms`, and per regime just {value, ms_per_step, kernel_ms, frac}."""
  out = _finite(out)
  c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                           "vs_baseline", "dtype", "data") if k in out}
  c["unit"] = _short(c.get("unit", ""), 60)
  cfg = {}
  for k, v in (out.get("config") or {}).items():
    if k in _CFG_DROP or isinstance(v, (dict, list)):
      continue
    cfg[k] = _short(v, 200 if k == "workload" else 80) if isinstance(v, str) else v
  c["config"] = cfg
  if "segments_ms" in out:
    c["segments_ms"] = out["segments_ms"]
  c["roofline"] = compact_roofline(out.get("roofline"))
  if out.get("roofline_other"):
    ro = compact_roofline(out["roofline_other"])
    c["roofline_other"] = {k: ro[k] for k in ("kernel", "kernel_ms", "achieved", "peak", "frac", "traffic", "issue_frac") if k in ro}
  cb = out.get("cpu_baseline")
  if cb:
    c["cpu_baseline"] = {"value": cb.get("value"), "unit": _short(cb.get("unit", ""), 60), "cores": cb.get("cores"),
                         "kind": cb.get("kind"), "sample": _short(cb.get("sample", ""), 160)}
  cm = out.get("cpu_baseline_matched")
  if cm:
    c["cpu_baseline_matched"] = {"value": cm.get("value"), "cores": cm.get("cores"), "kind": cm.get("kind")}
  if out.get("parity"):
    c["parity"] = _numbers_only(out["parity"])
  if out.get("regimes"):
    c["regimes"] = {n: {"value": r.get("value"), "ms_per_step": r.get("ms_per_step"),
                        "kernel_ms": (r.get("roofline") or {}).get("kernel_ms"), "frac": (r.get("roofline") or {}).get("frac")}
                    for n, r in out["regimes"].items()}
  nr = out.get("next_rows")
  if isinstance(nr, dict):
    c["next_rows"] = {row: {k: _sig(v, 4) for k, v in blk.items() if isinstance(v, (int, float)) and not isinstance(v, bool)}
                      for row, blk in nr.items() if isinstance(blk, dict)}
  c["detail"] = DETAIL_FILE
  return c


def render_line(out):
  """compact_line -> ONE line of strict JSON no longer than LINE_LIMIT; optional blocks are dropped, least important
  first, if a future field makes it too long (the contract keys, config, roofline and cpu_baseline never are)."""
  c = compact_line(out)
  for victim in (None, "next_rows", "roofline_other", "cpu_baseline_matched", "segments_ms", "regimes", "parity"):
    if victim is not None:
      c.pop(victim, None)
    line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    if len(line) <= LINE_LIMIT:
      return line
  raise SystemExit(f"bench line is {len(line)} bytes even without its optional blocks")


def emit(out, detail_path=None):
  """Rank 0: full result -> bench_detail.json (+ gpurun_out/ when it exists, so a gpurun call brings it back); stdout gets the
  compact line, flushed, as the LAST thing the process prints."""
  full = _finite(out)
  paths = [detail_path or os.path.join(ROOT, DETAIL_FILE)]
  scratch = os.path.join(ROOT, "gpurun_out")
  if detail_path is None and os.path.isdir(scratch):
    paths.append(os.path.join(scratch, DETAIL_FILE))
  for p in paths:
    try:
      with open(p, "w") as fh:
        json.dump(full, fh, allow_nan=False, indent=1)
        fh.write("\n")
    except OSError as e:                                   # a read-only tree must not cost the measurement
      print(f"bench: could not write {p}: {e}", file=sys.stderr)
  sys.stdout.flush()
  print(render_line(out), flush=True)


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=80)
  ap.add_argument("--warmup", type=int, default=8)
  ap.add_argument("--config", default="c3", choices=sorted(CONFIGS) + ["c1", "c5", "c3_grad"])
  ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                  help="weak: the config's B (or S) per GPU; strong: the config's B (or S) sharded over the ranks "
                       "(default: weak for c1..c3, strong for c4/c5 -- BASELINE.json shards those)")
  ap.add_argument("--recipe", default=None, choices=sorted(RECIPES),
                  help="the regime `value` is measured on (default: the config's; c3: baseline); given explicitly, only it runs")
  ap.add_argument("--regimes", default=None,
                  help="comma-separated recipes to time after the primary one in the same process and report under `regimes` "
                       "(default at N = 1: the config's list -- c3: pilco,worst; none at N > 1)")
  ap.add_argument("--batch", type=int, default=None, help="override B (or S): per GPU if weak, total if strong")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--force-generic", action="store_true")
  ap.add_argument("--extra-flags", type=int, default=0,
                  help="C-ABI flag bits OR-ed into every moment match (measurement: 512 = MM_NO_ROUTE, 256 = MM_FORCE_ROUTE)")
  ap.add_argument("--pmc-run", action="store_true", help="set by tools/collect_pmc.sh: exact --steps, no rounding to rollouts")
  ap.add_argument("--rehearse-gloo", action="store_true",
                  help="multi-process rehearsal on ONE GPU: gloo backend, every rank on cuda:0, costs gathered via host")
  return ap.parse_args()


def spawn_ranks(args):
  """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD process
  (this process has not touched the GPU and only relays the child's output and exit code)."""
  port = 29500 + (os.getpid() % 2000)
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  return subprocess.run(cmd, env=env).returncode


def src_hash():
  from tools.pmc_summary import src_hash as h
  return h(ROOT)


def load_pmc(tag, fallback=None):
  """The newest profiles/r*_pmc_<tag>.json (then <fallback>) collected on the kernel sources now in the tree, else
  (None, reason)."""
  import glob
  cur = src_hash()
  why = []
  for t in (tag, fallback):
    if t is None:
      continue
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9]*_pmc_{t}.json")), reverse=True)
    if not paths:
      why.append(f"profiles/r*_pmc_{t}.json not collected")
    for path in paths:
      with open(path) as fh:
        pmc = json.load(fh)
      if pmc.get("src_hash") == cur:
        return pmc, os.path.relpath(path, ROOT)
      why.append(f"{os.path.relpath(path, ROOT)} is stale (collected on kernel sources {pmc.get('src_hash')}, tree has {cur})")
  return None, "; ".join(why)


def pmc_scale(pmc, pmc_src, units_now, units_default):
  """Counters are per launch of the profiled run.  Every grid of a step is linear in the per-rank batch (B or S), so a
  run with another batch scales them by units_now / units_profiled -- and says so in the `pmc` field."""
  if pmc is None:
    return 1.0, pmc_src
  import re
  m = re.search(r"--batch\s+(\d+)", pmc.get("bench_args", ""))
  prof = int(m.group(1)) if m else units_default
  if prof == units_now:
    return 1.0, pmc_src
  return units_now / prof, f"{pmc_src} (collected at {prof} batch units per launch, scaled x{units_now / prof:g}: every grid of the step is linear in the batch)"


def pmc_kernel(pmc, prefix):
  """Counters of the kernel whose short name starts with ``prefix`` (the instantiation with the most time)."""
  if pmc is None:
    return None
  best = None
  for name, ent in pmc["kernels"].items():
    if name.startswith(prefix):
      w = ent.get("dur_us_under_pmc", 0.0) * ent.get("dispatches", 1)
      if best is None or w > best[0]:
        best = (w, name, ent)
  return None if best is None else (best[1], best[2])


def executed_ceiling(ent, kind):
  """Executed-work ceiling of one dispatch from its hardware counters, priced with the per-SIMD issue costs and the
  overlap rules MEASURED by tools/ubench_gap.hip (profiles/r02_ubench_gap.txt):

    kind "bf16" (v_mfma_f32_32x32x16_bf16, 32 pipe cycles): f32 FMA-class VALU (4 cycles, packed) serialises with the
         MFMA beyond the first 3 instructions per MFMA; f64 FMAs (5 cycles) co-execute up to 5 per MFMA; every other
         VALU instruction (v_max3, compares, moves, integer: 4 cycles) co-executes up to 24 cycles per MFMA;
    kind "f64"  (v_mfma_f64_16x16x4_f64, 64 pipe cycles): nothing co-executes -- MFMA, f64 VALU (5), the rest (4) add.

  cycles per SIMD = (MFMA pipe cycles + VALU issue cycles that cannot hide) / 1024, at the 2.4 GHz peak clock."""
  c = ent["counters"]
  n_mfma = c.get("SQ_INSTS_MFMA", 0.0)
  n_valu = max(0.0, c.get("SQ_INSTS_VALU", 0.0) - n_mfma)              # SQ_INSTS_VALU counts the MFMAs too
  f32 = c.get("SQ_INSTS_VALU_FMA_F32", 0.0) + c.get("SQ_INSTS_VALU_MUL_F32", 0.0) + c.get("SQ_INSTS_VALU_ADD_F32", 0.0)
  f64 = c.get("SQ_INSTS_VALU_FMA_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + c.get("SQ_INSTS_VALU_ADD_F64", 0.0)
  trans = c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
  if "SQ_INSTS_VALU_FMA_F32" not in c:
    f32, f64, trans = (n_valu, 0.0, 0.0) if kind == "bf16" else (0.0, n_valu, 0.0)
  other = max(0.0, n_valu - f32 - f64 - trans)
  mfma_cyc = c.get("SQ_VALU_MFMA_BUSY_CYCLES", n_mfma * ISSUE["mfma_bf16_32x32x16" if kind == "bf16" else "mfma_f64_16x16x4"])
  if kind == "bf16":
    f32_cyc = max(0.0, f32 - 3.0 * n_mfma) * ISSUE["valu_f32"]
    f64_cyc = max(0.0, f64 - 5.0 * n_mfma) * ISSUE["valu_f64"]
    oth_cyc = max(0.0, (other * ISSUE["valu_other"] + trans * ISSUE["valu_trans"]) - 24.0 * n_mfma)
  else:
    f32_cyc = f32 * 2.0
    f64_cyc = f64 * ISSUE["valu_f64"]
    oth_cyc = other * ISSUE["valu_other"] + trans * ISSUE["valu_trans"]
  valu_cyc = f32_cyc + f64_cyc + oth_cyc
  cyc_per_simd = (mfma_cyc + valu_cyc) / N_SIMD
  return {"ceiling_ms": cyc_per_simd / PEAK_CLOCK_HZ * 1e3,
          "mix": {"mfma": n_mfma, "valu_f32": f32, "valu_f64": f64, "valu_trans": trans, "valu_other": other,
                  "mfma_pipe_cycles_per_simd": mfma_cyc / N_SIMD, "valu_issue_cycles_per_simd_not_hidden": valu_cyc / N_SIMD}}


def one_rank_at_a_time(fn, rank, world, dist, rehearsal):
  """Model set-up.  Production (one process per GPU): every rank simply runs it.  The --rehearse-gloo mode puts all
  ranks on ONE GPU, and on this driver stack large f64 library kernels are not reliable while two processes
  time-slice a device (tools/potrf_probe.py, profiles/r02_potrf_probe.txt: rocSOLVER potrf fails at a random minor
  in ~10 % of the calls and once returned a wrong factor with info = 0; a lone process never does): there the ranks
  take turns, so that no two of them run the set-up's factorisations at the same time.  linalg.cholesky verifies
  every large factor it returns either way."""
  if not (rehearsal and world > 1):
    return fn()
  import torch
  out = None
  for turn in range(world):
    if turn == rank:
      out = fn()
      torch.cuda.synchronize()
    dist.barrier()
  return out


def roofline_block(kernel, k_ms, flops, peak, pmc, pmc_src, pscale, kind, prefix, why_flops):
  """`roofline` of one reduce kernel, SURVEY 8d accounting:
       achieved = ALGORITHMIC flops per launch (E * (2d + 12)) / the kernel's live duration      [TFLOP/s]
       frac     = achieved / the dense peak of the dtype (null where the algorithmic figure exceeds the peak: the
                  kernel then replaced part of the per-entry work -- moment collapse, skipped tiles -- and the
                  8d flop model does not describe it)
       issue_frac = executed-work ceiling / measured time (the kernel's OWN instruction mix from the counters,
                  priced with tools/ubench_gap.hip's issue costs): an issue-efficiency diagnostic, NOT a fraction of
                  the peak and never reported as `frac` or in TFLOP/s."""
  ach = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else None
  r = {"bound": "mfma", "kernel": kernel, "kernel_ms": round(k_ms, 4), "algorithmic_flops_per_launch": flops,
       "flops_model": why_flops, "achieved": None if ach is None else round(ach, 2), "peak": peak, "unit": "TFLOP/s",
       "frac": None, "traffic": None}
  if ach is not None:
    if ach < 0.98 * peak:
      r["frac"] = round(ach / peak, 4)
    else:
      r["frac_null_reason"] = ("algorithmic rate within 2 % of or above the dense peak of the dtype: the launch does not execute SURVEY 8d's "
                               "per-entry flops on that pipe (the 2d bilinear flops of an f32 pack run on the bf16 matrix pipe; cubic + quartic "
                               "remainder from f64 moments, tiles with max|b| <= 1/20 skipped: config.offdiag_items) -- see `pipes` for "
                               "what each pipe executed")
  got = pmc_kernel(pmc, prefix)
  if got is None:
    r["pmc"] = pmc_src
    return r
  name, ent = got
  ce = executed_ceiling(ent, kind)
  ceiling_ms = ce["ceiling_ms"] * pscale
  r.update({"kernel": name, "issue_frac": round(ceiling_ms / k_ms, 4) if k_ms > 0 else None,
            "issue_ceiling_ms": round(ceiling_ms, 4),
            "issue_frac_definition": "executed-work ceiling / measured kernel time; ceiling = (MFMA pipe cycles + VALU issue cycles that "
                                     "cannot hide beside the MFMAs, by instruction class: tools/ubench_gap.hip) / 1024 SIMDs / 2.4 GHz, "
                                     "from the kernel's own hardware counters -- issue efficiency, not a fraction of the flop peak",
            "instruction_mix_per_launch": {k: round(v * pscale, 1) for k, v in ce["mix"].items()},
            "traffic": ent["counters"]["hbm_bytes"] * pscale if "hbm_bytes" in ent["counters"] else None, "pmc": pmc_src})
  if r.get("issue_frac") and r["issue_frac"] > 1.0:
    r["issue_frac_note"] = ("above 1: the additive pricing of tools/ubench_gap.hip (every f32 FMA-class instruction beyond three per MFMA at 4 "
                            "cycles) over-prices this mix -- part of it (v_min / v_max / conversions of the exp2 branch) co-executes; the "
                            "kernel issues back to back")
  c = ent["counters"]
  if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
    r["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD / (c["GRBM_GUI_ACTIVE"] / 8.0), 4)
  # ---- which pipe did what: EXECUTED flops per pipe from the instruction counters over the live
  # kernel time, against that pipe's dense peak; and the clock the chip held under this kernel
  sec = k_ms * 1e-3
  mix = ce["mix"]
  pipes = {}
  if kind == "bf16":
    # one v_mfma_f32_32x32x16_bf16 = 32 x 32 x 16 x 2 flop; the kernel's f32 arithmetic is packed (v_pk_fma / mul / add: two
    # elements per lane and instruction)
    mf = mix["mfma"] * pscale * 32768.0
    vf = (2.0 * c.get("SQ_INSTS_VALU_FMA_F32", 0.0) + c.get("SQ_INSTS_VALU_MUL_F32", 0.0) + c.get("SQ_INSTS_VALU_ADD_F32", 0.0)) * pscale * 64 * 2
    pipes["mfma_bf16"] = {"executed_tflops": round(mf / sec / 1e12, 1), "peak": PEAK_TFLOPS["bf16"], "frac": round(mf / sec / 1e12 / PEAK_TFLOPS["bf16"], 4)}
    pipes["valu_f32"] = {"executed_tflops": round(vf / sec / 1e12, 1), "peak": PEAK_TFLOPS["f32"], "frac": round(vf / sec / 1e12 / PEAK_TFLOPS["f32"], 4),
                         "note": "packed f32: 2 elements per lane and instruction"}
  else:
    mf = mix["mfma"] * pscale * 2048.0                        # v_mfma_f64_16x16x4_f64
    vf = (2.0 * c.get("SQ_INSTS_VALU_FMA_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + c.get("SQ_INSTS_VALU_ADD_F64", 0.0)) * pscale * 64
    pipes["mfma_f64"] = {"executed_tflops": round(mf / sec / 1e12, 1), "peak": PEAK_TFLOPS["f64"], "frac": round(mf / sec / 1e12 / PEAK_TFLOPS["f64"], 4)}
    pipes["valu_f64"] = {"executed_tflops": round(vf / sec / 1e12, 1), "peak": PEAK_TFLOPS["f64"], "frac": round(vf / sec / 1e12 / PEAK_TFLOPS["f64"], 4),
                         "note": "the f64 matrix and vector pipes are ONE datapath on gfx950 (tools/ubench_gap.hip): the two fractions add"}
  r["pipes"] = pipes
  if c.get("GRBM_GUI_ACTIVE") and ent.get("dur_us_under_pmc"):
    clk = (c["GRBM_GUI_ACTIVE"] / 8.0) / (ent["dur_us_under_pmc"] * 1e-6)
    r["measured_clock_ghz"] = round(clk / 1e9, 3)
    r["issue_frac_at_measured_clock"] = round(ceiling_ms * (PEAK_CLOCK_HZ / clk) / k_ms, 4) if k_ms > 0 else None
    r["measured_clock_definition"] = ("GRBM_GUI_ACTIVE / 8 XCDs / the kernel's duration in the counter pass; issue_frac is priced at the 2.4 GHz "
                                      "peak clock, issue_frac_at_measured_clock at the clock the chip held under this kernel")
  return r


def main():
  args = parse()
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    raise SystemExit(spawn_ranks(args))
  import numpy as np
  import torch
  from gpflowpilco_amd import _lib, ops
  from gpflowpilco_amd.cost import expected_gaussian_cost
  from gpflowpilco_amd.distributed import shard_range
  from gpflowpilco_amd.synthetic import make_inputs, make_svgp

  rank = int(os.environ.get("RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus and not args.rehearse_gloo:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                     "(or run `python bench.py --gpus N` alone, which starts the ranks itself)")
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
  if args.config == "c1":
    return composed_bench(args, rank, world, dev, dist)
  if args.config == "c3_grad":
    return grad_bench(args, rank, world, dev, dist)
  cfg = dict(CONFIGS[args.config])
  scaling = args.scaling or cfg["scaling"]
  L, M, d, H = cfg["L"], cfg["M"], cfg["d"], cfg["H"]
  dtype = torch.float32 if cfg["dtype"] == "f32" else torch.float64
  f32_mode = dtype == torch.float32
  Bcfg = args.batch or cfg["B"]
  if scaling == "strong":
    B_total = Bcfg
    lo, hi = shard_range(B_total, rank, world)           # contiguous, balanced slice of the batch axis
    B = hi - lo
  else:
    B, B_total, lo = Bcfg, Bcfg * world, rank * Bcfg
  if B <= 0:
    raise SystemExit(f"rank {rank} got an empty shard of B={B_total}")
  steps = args.steps                                       # EXACTLY --steps timed steps (see one_step / finish_rollout)
  primary = args.recipe or cfg["recipe"]
  # the regimes of one run: the primary recipe gives `value`; the others are timed in the same process after it and
  # reported under `regimes` (N = 1 only unless --regimes says otherwise)
  if args.regimes:
    names = [primary] + [r for r in args.regimes.split(",") if r and r != primary]
  elif args.recipe is None and not args.pmc_run and world == 1:
    names = [primary] + [r for r in cfg.get("also", ()) if r != primary]
  else:
    names = [primary]
  for n in names:
    if n not in RECIPES:
      raise SystemExit(f"unknown recipe {n}")
  F = _lib
  Po = L * (L - 1) // 2
  e_off = float(B) * Po * M * M
  e_diag = float(B) * L * M * (M + 1) / 2
  models = {}

  def model_for(rec):
    key = (rec["ls_bounds"], rec["stable"])
    if key not in models:
      def build_model():
        syn_ = make_svgp(L, M, d, seed=cfg["seed"], device=str(dev), ls_bounds=rec["ls_bounds"], stable=rec["stable"])
        model_ = syn_.to_model(dev)
        return syn_, model_, model_.packed(dtype, True, dev)
      models[key] = one_rank_at_a_time(build_model, rank, world, dist, args.rehearse_gloo)
    return models[key]

  def run_regime(recipe_name, is_primary):
    rec = dict(RECIPES[recipe_name])
    if not cfg["closed"] and not rec["independent"]:
      # d != D: no closed rollout exists; the recipe's model on H independent (mu, Sigma) draws (mu in [0.3, 0.7]^d)
      rec["independent"] = True
      rec["text"] += "; d != D, so every step takes a fresh draw mu~U[0.3,0.7]^d / Sigma std 0.1"
      rec["mu_range"] = (0.3, 0.7)
    syn, model, pm = model_for(rec)
    if rec["independent"]:
      # the step kernel on H independent draws of the whole batch; rank-independent global draw so that
      # a strong-scaling run processes the same B_total inputs at every N
      lo_mu, hi_mu = rec.get("mu_range", (0.0, 1.0))
      mu_np, S_np = make_inputs(B_total * H, d, seed=2000 + cfg["seed"], scale=0.1, lo=lo_mu, hi=hi_mu)
      mu_np = mu_np.reshape(H, B_total, d)[:, lo:lo + B]
      S_np = S_np.reshape(H, B_total, d, d)[:, lo:lo + B]
      draws_mu = torch.tensor(np.ascontiguousarray(mu_np), dtype=dtype, device=dev)
      draws_S = torch.tensor(np.ascontiguousarray(S_np), dtype=dtype, device=dev)
      mu0_np, S0_np = mu_np[0], S_np[0]
    else:
      mu_all, S_all = make_inputs(B_total, d, seed=2000 + cfg["seed"], scale=0.1, lo=0.3, hi=0.7)
      mu0_np, S0_np = mu_all[lo:lo + B], S_all[lo:lo + B]
    mu0 = torch.tensor(mu0_np, dtype=dtype, device=dev)
    S0 = torch.tensor(S0_np, dtype=dtype, device=dev)
    dc = d if cfg["closed"] else L                           # dimension of the per-step cost statistic's argument
    target = torch.full((dc,), 0.5 if cfg["closed"] else 0.0, dtype=dtype, device=dev)
    precis = torch.eye(dc, dtype=dtype, device=dev) * 4.0
    base = ops.make_flags(True, True, args.force_generic) | (F.MM_FORCE_WORST_TIER if rec["worst"] else 0) | args.extra_flags
    traj_mu = torch.empty(H, B, dc, dtype=dtype, device=dev)
    traj_S = torch.empty(H, B, dc, dc, dtype=dtype, device=dev)
    Bmax = -(-B_total // world)
    gathered = [torch.empty(Bmax, H, dtype=dtype, device=dev) for _ in range(world)] if world > 1 else None
    pad_cost = torch.zeros(Bmax, H, dtype=dtype, device=dev)
    ev = []
    state = {"mu": mu0.clone(), "S": S0.clone(), "h": 0, "cost": None, "rollouts": 0, "collectives": 0}
    Ev = lambda: torch.cuda.Event(enable_timing=True)

    def finish_rollout(h_done):
      """Per-step cost statistic of the h_done steps just taken, [B_local, H] -> all ranks: ONE collective
      per rollout.  Called at every rollout boundary and once more at the end of the timed region when --steps is not
      a whole number of rollouts, so the cost kernel and the collective are inside the timed region for every K."""
      cost = torch.zeros(B, H, dtype=dtype, device=dev)
      cost[:, :h_done] = expected_gaussian_cost(traj_mu[:h_done], traj_S[:h_done], target, precis).T
      if world > 1:
        pad_cost[:B].copy_(cost)
        if args.rehearse_gloo:
          host = [torch.empty(Bmax, H, dtype=dtype) for _ in range(world)]
          dist.all_gather(host, pad_cost.cpu())
          full = [t.to(dev) for t in host]
        else:
          dist.all_gather(gathered, pad_cost)
          full = gathered
        state["collectives"] += 1
        if scaling == "strong":
          sizes = [shard_range(B_total, r, world) for r in range(world)]
          state["cost"] = torch.cat([t[:b - a] for t, (a, b) in zip(full, sizes)], 0)
        else:
          state["cost"] = torch.cat([t[:B] for t in full], 0)
      else:
        state["cost"] = cost
      state["h"] = 0
      state["rollouts"] += 1

    extra = base & ~(F.MM_FULL_OUTPUT_COV | F.MM_MODEL_UNCERTAINTY | F.MM_FORCE_GENERIC)

    def one_step(timed, staged=False):
      """One rollout step.  The TIMED steps call the product's entry point, mm_moment_match (one call: its q stage puts the
      off-diagonal operands and the moment chain on a side stream beside the diagonal sweep, and joins); `staged` steps run the
      same kernels through the stage API one stage at a time -- no overlap -- so that HIP events give clean per-stage times
      (`segments_ms`, the roofline's kernel time) and rocprofv3 per-kernel durations."""
      h = state["h"]
      if rec["independent"]:
        state["mu"], state["S"] = draws_mu[h], draws_S[h]
      elif h == 0:
        state["mu"], state["S"] = mu0.clone(), S0.clone()
      e = [Ev() for _ in range(5)] if staged else None
      if staged:
        e[0].record()
        f1, cross, _ = ops.q_forward(pm, state["mu"], state["S"], base)
        e[1].record()
        ops.Q_reduce_forward(pm, B, base | F.MM_STAGE_DIAG)
        e[2].record()
        ops.Q_reduce_forward(pm, B, base | F.MM_STAGE_OFFDIAG)
        e[3].record()
        Sff = ops.Q_reduce_forward(pm, B, base | F.MM_STAGE_FINALIZE)
      else:
        f1, Sff, cross = ops.moment_match(pm, state["mu"], state["S"], True, True, 0.0, args.force_generic, extra)
      if cfg["closed"]:
        state["mu"], state["S"] = ops.euler_update(state["mu"], state["S"], f1, Sff, cross, 1.0)
        traj_mu[h].copy_(state["mu"]); traj_S[h].copy_(state["S"])
      else:
        traj_mu[h].copy_(f1); traj_S[h].copy_(Sff)            # the predicted increment's moments carry the cost statistic
      if staged:
        e[4].record(); ev.append(e)
      state["h"] = h + 1
      if state["h"] == H:
        finish_rollout(H)

    def fence():
      torch.cuda.synchronize()
      if world > 1:
        dist.barrier()
      torch.cuda.synchronize()

    for _ in range(args.warmup):
      one_step(False)
    state["h"] = 0
    state["rollouts"] = 0
    state["collectives"] = 0
    if world > 1 and not args.rehearse_gloo:
      # the warm-up steps need not reach the end of a rollout: bring up the collective's channels untimed
      dist.all_gather(gathered, pad_cost)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
      one_step(True, staged=args.pmc_run)                    # (counter passes: stage by stage, one kernel at a time)
    if state["h"] != 0 and not args.pmc_run:
      finish_rollout(state["h"])                             # the partial last rollout's costs + collective
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
      tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_gloo else dev)
      dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
      elapsed = float(tmax.item())
    pm.check_status(B)
    rollouts_timed, collectives_timed = state["rollouts"], state["collectives"]
    # per-stage times: a separate, untimed pass of staged steps (<= one rollout of them)
    if not args.pmc_run:
      ev.clear()
      state["h"] = 0
      for _ in range(min(steps, H)):
        one_step(False, staged=True)
      torch.cuda.synchronize()
    # regime of the off-diagonal reduce over one (untimed) rollout: its kernels' time depends on it
    collapsed = [0, 0, 0]
    groups = [0, 0, 0]
    routed = 0
    if f32_mode and not args.pmc_run:
      state["h"] = 0
      for _ in range(H):
        one_step(False)
        collapsed = [x + y for x, y in zip(collapsed, ops.offdiag_stats(pm, B, base))]
        groups = [x + y for x, y in zip(groups, ops.offdiag_row_groups(pm, B, base))]
        routed += ops.offdiag_routed(pm, B, base)               # items the accuracy contract re-reduced in f64 (csrc/mm_route.hip)
    if cfg["closed"] and not torch.isfinite(state["S"]).all():
      raise SystemExit(f"non-finite state in the timed rollout (recipe {recipe_name})")
    if not args.pmc_run:
      if state["cost"] is None or tuple(state["cost"].shape) != (B_total, H):
        raise SystemExit(f"gathered cost matrix has shape {None if state['cost'] is None else tuple(state['cost'].shape)}, "
                         f"expected {(B_total, H)}")
      if not torch.isfinite(state["cost"]).all():
        raise SystemExit("non-finite per-step costs")

    # ---- per-segment device time (HIP events on the launch stream; the staged pass above / the counter passes' steps) -------------
    seg = {k: float(np.mean([e[i].elapsed_time(e[i + 1]) for e in ev])) for i, k in enumerate(("q_stage", "diag", "offdiag", "tail"))}
    pmc, pmc_src = load_pmc(f"{args.config}_{recipe_name}", fallback=args.config if (recipe_name == cfg["recipe"] and args.config != "c3") else None)
    pscale, pmc_src = pmc_scale(pmc, pmc_src, B, cfg["B"])
    kdim = (d + 3) // 4 if (d + 3) // 4 <= 4 else (6 if (d + 3) // 4 <= 6 else 8)

    def reduce_roofline(which):
      entries = e_off if which == "offdiag" else e_diag
      f32k = which == "offdiag" and f32_mode and not args.force_generic
      if args.force_generic:
        prefix = "k_qred_generic"
      elif f32k:
        prefix = "k_qred_f32_mfma"
      else:
        prefix = f"k_qred_f64_mfma<{kdim}, " + ("true" if which == "diag" else "false")
      why = (f"SURVEY 8d: E * (2d + 12) with E = B*{'Po*M^2' if which == 'offdiag' else 'L*M(M+1)/2'} = {entries:.4g} entries "
             f"({'off-diagonal' if which == 'offdiag' else 'diagonal'} pairs of one step), d = {d}")
      r = roofline_block(prefix, seg[which], entries * (2 * d + 12), PEAK_TFLOPS["f32" if f32k else "f64"], pmc, pmc_src, pscale,
                         "bf16" if f32k else "f64", prefix, why)
      r["entries_per_launch"] = entries
      return r

    roofs = {"offdiag": reduce_roofline("offdiag") if Po else None, "diag": reduce_roofline("diag")}
    dominant = "offdiag" if (Po and seg["offdiag"] >= seg["diag"]) else "diag"
    # step-level cross-check of the same model: all pairs' 8d flops over the whole step, against the f32/f64 peak
    step_ms = 1e3 * elapsed / steps
    step_flops = (e_off + e_diag) * (2 * d + 12) + e_diag * 2
    step_tf = step_flops / (step_ms * 1e-3) / 1e12
    step_peak = PEAK_TFLOPS["f32" if f32_mode else "f64"]
    roof_step = {"flops_per_step": step_flops, "achieved": round(step_tf, 2), "peak": step_peak, "unit": "TFLOP/s",
                 "frac": round(step_tf / step_peak, 4) if step_tf <= step_peak else None,
                 "note": "SURVEY 8d F_Q per step / ms_per_step (q stage included in the time); diagonal pairs run in f64 in both modes"}
    if step_tf > step_peak:
      roof_step["frac_null_reason"] = "exceeds the peak: the step does not execute 8d's per-entry work in this data regime (see roofline.frac_null_reason)"
    # q stage: HBM-bound operand producers: bytes from the counters / the HIP-event segment is the whole stage
    qroof = {"bound": "hbm", "kernels": "k_prep + k_qvec + k_pairvec (+ k_wmom_gemm [f64 MFMA GEMM] + k_spoly + k_wmom56_gemm [bf16 MFMA GEMM] + k_item_classes + k_spoly56 + k_spoly4)", "segment_ms": round(seg["q_stage"], 4),
             "peak": PEAK_HBM_GBS, "unit": "GB/s", "per_kernel": {}}
    if pmc is not None:
      for pre in ("k_qvec", "k_pairvec", "k_wmom_gemm", "k_spoly<", "k_wmom56_gemm", "k_spoly56", "k_spoly4"):
        got = pmc_kernel(pmc, pre)
        if got and "hbm_bytes" in got[1]["counters"]:
          qroof["per_kernel"][got[0]] = {"hbm_bytes": got[1]["counters"]["hbm_bytes"] * pscale}

    res = {
        "recipe": recipe_name, "recipe_text": rec["text"],
        "value": round(B_total * steps / elapsed, 2), "ms_per_step": round(step_ms, 4), "steps": steps,
        "rollouts_timed": rollouts_timed, "collectives_timed": collectives_timed if world > 1 else 0,
        "segments_ms": {k: round(v, 4) for k, v in seg.items()},
        "offdiag_items_one_rollout": {"collapsed": collapsed[0], "wholly_inside": collapsed[2], "total": collapsed[1],
                                      "routed_to_f64": routed, "partly_collapsed": groups[0],
                                      "collapsed_row_groups": groups[1], "row_groups": groups[2]},
        "roofline": roofs[dominant], "roofline_other": roofs["diag" if dominant == "offdiag" else "offdiag"],
        "roofline_step": roof_step, "roofline_q_stage": qroof,
    }

    # ---- parity of THIS regime (rank 0, N == 1): GPU step vs the CPU restatements on the same element ------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.pmc_run:
      from oracle import mm_oracle as mo
      Bc = 1 if M >= 1000 else min(B, 4)
      err = lambda g, w: float(np.abs(g.double().cpu().numpy() - w).max())

      def gpu_step(pm_, mu_t, S_t, flags):
        f1_, cross_, _ = ops.q_forward(pm_, mu_t, S_t, flags)
        Sff_ = ops.Q_reduce_forward(pm_, mu_t.shape[0], flags)
        pm_.check_status(mu_t.shape[0])
        return f1_, Sff_, cross_
      po_full = mo.SVGPParams(Z=np.broadcast_to(syn.Z, (L, M, d)).copy(), lengthscales=syn.lengthscales, variance=syn.variance,
                              q_mu=syn.q_mu, q_sqrt=syn.q_sqrt, whiten=True)
      par = {}
      g = gpu_step(pm, mu0[:Bc].contiguous(), S0[:Bc].contiguous(), base)
      if 8.0 * Bc * L * M * M * 3 < 4e9:
        # algorithm-matched fp64 restatement (O(M^2) per pair): affordable for every regime
        from oracle import mm_fused_ref as fr
        key = ("fused_pre", rec["ls_bounds"], rec["stable"])
        if key not in models:
          models[key] = fr.precompute(po_full)
        t0c = time.perf_counter()
        fm = fr.moment_match(mu0_np[:Bc], S0_np[:Bc], po_full, *models[key])
        tm = time.perf_counter() - t0c
        par["vs_fused_restatement"] = {"B": Bc, "max_abs_err": {"f1": err(g[0], fm[0]), "Sff": err(g[1], fm[1]), "cross_pre": err(g[2], fm[2])},
                                       "max_abs": {"f1": float(np.abs(fm[0]).max()), "Sff": float(np.abs(fm[1]).max()),
                                                   "cross_pre": float(np.abs(fm[2]).max())},
                                       "vs": "oracle/mm_fused_ref.py (fp64 numpy, same algorithm as the kernels), first step, same inputs"}
        res["_fused"] = (fm, tm, Bc)
      if f32_mode:
        # f32 mode against the same kernels in f64 mode on a larger sample of THIS regime's inputs
        Br = min(B, 8)
        pm64 = model.packed(torch.float64, True, dev)
        if cfg["closed"] and not rec["independent"] and not rec["worst"]:
          m32, S32 = ops.rollout_closed(pm, mu0[:Br].contiguous(), S0[:Br].contiguous(), H)
          m64, S64 = ops.rollout_closed(pm64, mu0[:Br].double().contiguous(), S0[:Br].double().contiguous(), H)
          pm64.check_status(Br)
          par["rollout_f32_vs_f64_mode"] = {
              "B": Br, "H": H, "max_abs_diff": {"mu_H": float((m32.double() - m64).abs().max()),
                                                "Sigma_H": float((S32.double() - S64).abs().max())},
              "max_abs": {"mu_H": float(m64.abs().max()), "Sigma_H": float(S64.abs().max())}}
        else:
          g32 = gpu_step(pm, mu0[:Br].contiguous(), S0[:Br].contiguous(), base)
          g64 = gpu_step(pm64, mu0[:Br].double().contiguous(), S0[:Br].double().contiguous(), ops.make_flags(True, True, False))
          par["step_f32_vs_f64_mode"] = {
              "B": Br, "max_abs_diff": {k: float((a.double() - b).abs().max()) for k, a, b in zip(("f1", "Sff", "cross_pre"), g32, g64)},
              "max_abs": {k: float(b.abs().max()) for k, b in zip(("f1", "Sff", "cross_pre"), g64)}}
      res["parity"] = par
      if is_primary:
        res["_ctx"] = dict(syn=syn, pm=pm, mu0=mu0, S0=S0, mu0_np=mu0_np, S0_np=S0_np, rec=rec, base=base, gpu_first=g, Bc=Bc)
    return res

  results = {}
  for i, n in enumerate(names):
    results[n] = run_regime(n, i == 0)
  pr = results[primary]
  rec_p = RECIPES[primary]
  out = {
      "metric": "moment_matched_rollout_step_elements_per_sec",
      "value": pr["value"],
      "unit": "rollout step-elements/s (B*H per rollout second)",
      "n_gpus": world, "steps": steps, "warmup": args.warmup,
      "ms_per_step": pr["ms_per_step"],
      "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
      "dtype": cfg["dtype"], "data": "synthetic",
      "config": {"workload": f"{cfg['label']}; recipe={primary}: {rec_p['text']}",
                 "N": M, "d": d, "D": L, "H": H, "B_per_gpu": B if scaling == "weak" else None, "B_total": B_total,
                 "B_this_rank": B, "parallelism": f"dp{world} over B ({scaling})", "recipe": primary,
                 "value_is": f"the `{primary}` regime; every other entry of `regimes` was timed in the same process with the same --steps/--warmup",
                 "rollouts_timed": pr["rollouts_timed"], "collectives_timed": pr["collectives_timed"],
                 "steps_requested": args.steps,
                 "diag_pairs": "f64", "offdiag_pairs": cfg["dtype"],
                 "offdiag_items_one_rollout": dict(pr["offdiag_items_one_rollout"],
                                                   meaning="(b, off-diagonal pair, step) items of one rollout; collapsed: the degree-3..6 polynomial "
                                                           "p6 of the remainder from weight moments for EVERY row (Cauchy-Schwarz bound <= 1/2), tiles with max|b| <= 1/4 skipped after a "
                                                           "screening MFMA; partly_collapsed: the same for some 64-row groups of the item (collapsed_row_groups of row_groups over "
                                                           "all items), the other groups reduced densely; wholly_inside: the Cauchy-Schwarz bound alone puts every |b| <= 1/4, "
                                                           "no tile work (csrc/mm_moments.hip, mm_moments6.hip, mm_mfma.hip); routed_to_f64: items whose f32 "
                                                           "rounding-error estimate exceeded MM_ROUTE_TOL = 3e-4 of the covariance block's scale and "
                                                           "were re-reduced in f64 inside the timed off-diagonal segment (csrc/mm_route.hip)")},
      "segments_ms": pr["segments_ms"],
      "roofline": pr["roofline"],
      "roofline_other": pr["roofline_other"],
      "roofline_step": pr["roofline_step"],
      "roofline_q_stage": pr["roofline_q_stage"],
  }
  ctx = pr.pop("_ctx", None)

  # ---- CPU baseline + parity vs the literal oracle (rank 0, N == 1 only, primary regime) --------------------------
  if ctx is not None:
    from oracle import mm_oracle as mo
    syn, mu0_np, S0_np, rec, Bc = ctx["syn"], ctx["mu0_np"], ctx["S0_np"], ctx["rec"], ctx["Bc"]
    small = M <= 256
    # the literal algorithm materialises eKuffu [B,L,M,L,M] (8 B L^2 M^2 bytes) and needs 2 L^2 M^3 flops per element:
    # where that does not fit a bounded sample (C4: 131 GB), the sample is the sub-problem of the first Lc latents
    # (latents are independent: f1, Sff[:Lc,:Lc], cross[:,:Lc] of the full model ARE the sub-model's outputs)
    Lc = L
    while Lc > 1 and (8.0 * Bc * Lc * Lc * M * M > 2.2e9 or 2.0 * Lc * Lc * float(M) ** 3 > 1.2e12):
      Lc -= 1
    po = mo.SVGPParams(Z=np.broadcast_to(syn.Z, (L, M, d))[:Lc].copy(), lengthscales=syn.lengthscales[:Lc],
                       variance=syn.variance[:Lc], q_mu=syn.q_mu[:, :Lc], q_sqrt=syn.q_sqrt[:Lc], whiten=True)
    mu_c, S_c = mu0_np[:Bc].copy(), S0_np[:Bc].copy()
    budget = 20.0
    t0 = time.perf_counter()
    nst = 0
    while True:
      f1o, Sffo, cro = mo.mm_gauss_svgp_mo(mu_c, S_c, po)
      nst += 1
      if nst == 1:
        first = (f1o, Sffo, cro)
      if cfg["closed"] and not rec["independent"] and Lc == L:
        Sxf = mo.cross_covariance(S_c, cro, True)
        mu_c, S_c = mo.euler_moment_update(mu_c, S_c, f1o, Sffo, Sxf, 1.0)
      if time.perf_counter() - t0 > budget or nst >= (H if not small else 4 * H):
        break
    tc = time.perf_counter() - t0
    sub = "" if Lc == L else f" of the sub-problem with the first {Lc} of {L} latents (the full eKuffu tensor would need {8.0 * L * L * M * M / 1e9:.0f} GB)"
    out["cpu_baseline"] = {"value": round(Bc * nst / tc, 4), "unit": out["unit"] + ("" if Lc == L else f" [{Lc}-latent sub-problem]"),
                           "cores": os.cpu_count(), "kind": "port",
                           "sample": f"literal fp64 oracle (materialised eKuffu [B,L,M,L,M] + triangular solves, "
                                     f"numpy/OpenBLAS threads), B={Bc}, {nst} step(s) of the same workload ({primary} recipe){sub}, {tc:.1f}s"}
    fused = pr.pop("_fused", None)
    if fused is not None and M >= 1000 and Lc == L:
      # second CPU figure: the algorithm-matched restatement -- beta / C hoisted out of the
      # step, O(M^2) per kernel pair -- so the ratio to the GPU is not merely the O(M^3) -> O(M^2) change
      fm, tm, Bf = fused
      out["cpu_baseline_matched"] = {"value": round(Bf / tm, 4), "unit": out["unit"], "cores": os.cpu_count(), "kind": "port",
                                     "sample": f"algorithm-matched fp64 restatement (oracle/mm_fused_ref.py: numpy, O(M^2) per pair, "
                                               f"precompute excluded), B={Bf}, 1 step without the Euler update, {tm:.1f}s",
                                     "max_abs_diff_vs_literal": {"f1": float(np.abs(fm[0] - first[0]).max()),
                                                                 "Sff": float(np.abs(fm[1] - first[1]).max())}}
    f1, Sff, cross = ctx["gpu_first"]
    f1, Sff, cross = f1[:, :Lc], Sff[:, :Lc, :Lc], cross[:, :, :Lc]
    err = lambda g, w: float(np.abs(g.double().cpu().numpy() - w).max())
    out["parity"] = dict({"vs": "fp64 CPU oracle (literal reference algorithm), first step, same inputs" + ("" if Lc == L else f" (first {Lc} latents)"), "B": Bc,
                          "recipe": primary,
                          "max_abs_err": {"f1": err(f1, first[0]), "Sff": err(Sff, first[1]), "cross_pre": err(cross, first[2])},
                          "max_abs": {"f1": float(np.abs(first[0]).max()), "Sff": float(np.abs(first[1]).max()),
                                      "cross_pre": float(np.abs(first[2]).max())}}, **pr.get("parity", {}))
  # next row f-1, in the same driver-timed line (the full bench of that row: --config c3_grad): a short forward + backward of
  # the primary regime's first step on the f32 pack
  # (ctx: the primary regime's model, pack and first-step inputs, taken above)
  if (ctx is not None and world == 1 and not args.pmc_run and args.recipe is None and f32_mode and not args.force_generic
      and ops.backward_supported(ctx["pm"])):
    pm_, mu_, S_ = ctx["pm"], ctx["mu0"], ctx["S0"]
    gg = torch.Generator(device="cpu").manual_seed(cfg["seed"])
    g1, g2, g3 = (torch.randn(s_, generator=gg, dtype=torch.float64).to(dev) for s_ in ((B, L), (B, L, L), (B, d, L)))
    fl = ops.make_flags(True, True)

    def fwd_bwd(mode):
      if mode == "one_sweep":        # what a differentiating caller runs (autodiff.MomentMatchFunction): the backward's sweeps give the value too
        _, _, _, sums, g_ = ops.moment_match_with_sums(pm_, mu_, S_)
        ops.moment_match_backward(pm_, mu_, S_, g1, g2, g3, True, True, forward_generation=g_, sums=sums)
        return
      ops.moment_match(pm_, mu_, S_)
      if mode == "two_pass":
        ops.moment_match_backward(pm_, mu_, S_, g1, g2, g3, True, True, forward_generation=pm_.workspace_generation(B, fl))
    t_ms = {}
    for name_, mode_ in (("forward_ms", "forward"), ("forward_backward_ms", "one_sweep"), ("two_pass_forward_backward_ms", "two_pass")):
      fwd_bwd(mode_)
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(5):
        fwd_bwd(mode_)
      e1.record(); torch.cuda.synchronize()
      t_ms[name_] = round(e0.elapsed_time(e1) / 5, 4)
    out["next_rows"] = {"f-1": dict(t_ms, what="moment match forward / value + vector-Jacobian product w.r.t. (mu, Sigma) on the f32 pack, first "
                                         "step's inputs of the primary regime, B elements.  forward_backward_ms: what a differentiating caller "
                                         "runs -- mm_moment_match_with_sums (q stage + the BACKWARD's two M x M sweeps, which contain the forward's "
                                         "sums) + the chain rule; two_pass_forward_backward_ms: mm_moment_match, then mm_moment_match_backward "
                                         "(four sweeps; rounds 2-3).  Stage times and roofline: bench.py --config c3_grad", B=B,
                                    forward_backward_over_forward=round(t_ms["forward_backward_ms"] / t_ms["forward_ms"], 2),
                                    two_pass_over_forward=round(t_ms["two_pass_forward_backward_ms"] / t_ms["forward_ms"], 2))}
  for r in results.values():
    r.pop("_fused", None); r.pop("_ctx", None)
  if len(results) > 1 or args.regimes:
    out["regimes"] = {n: {k: v for k, v in r.items() if k not in ("roofline_q_stage",)} for n, r in results.items()}
  if rank == 0:
    emit(out)
  if world > 1:
    dist.destroy_process_group()


def composed_bench(args, rank, world, dev, dist):
  """configs[0]: one step = one COMPOSED rollout step (encoder, policy match + head, drift match, cross-covariance
  bookkeeping, Euler update, expected cost) of the local batch; a rollout = one mm_rollout_composed call."""
  import numpy as np
  import torch
  from gpflowpilco_amd import ops
  from gpflowpilco_amd.synthetic import make_cartpole_like, make_inputs
  c = dict(CARTPOLE)
  H, dtype = c["H"], torch.float64
  B = args.batch or c["B"]
  steps = max(H, -(-args.steps // H) * H)
  drift_s, pol_s = make_cartpole_like(c["M"], c["Mpol"], c["seed"], device=str(dev))
  drift, pol = drift_s.to_model(dev), pol_s.to_model(dev)
  rng = np.random.default_rng(2000 + c["seed"] + rank)
  mu_np = np.array([0.4, 0.2, 0.5, 0.3])[None] + 0.05 * rng.standard_normal((B, 4))
  _, S_np = make_inputs(B, 4, seed=3000 + rank, scale=0.05)
  target = np.array([0.0, 1.0, 0.0, 0.0, 0.0])
  precis = 16 * np.array([[0.25, 0, -0.5, 0, 0], [0, 0.25, 0, 0, 0], [-0.5, 0, 1, 0, 0], [0] * 5, [0] * 5], dtype=float)
  t = lambda a: torch.tensor(np.asarray(a), dtype=dtype, device=dev)
  scale, shift, active = 2.0, -0.5, (1,)
  roll = ops.ComposedRollout(drift.packed(dtype, True, dev), pol.packed(dtype, False, dev), nx=4, active_dims=active,
                             head_scale=scale, head_shift=shift, target=t(target), precis=t(precis))
  mx, Sxx = t(mu_np), t(S_np)
  # the rollout is captured once into a HIP graph and replayed (the reference traces its closure once under
  # tf.function, loops/pilco.py:219-220): at B = 1 the eager path is paced by the host enqueueing ~16 launches per step
  eager_roll = roll
  graphed = ops.GraphedComposedRollout(roll, B, H)
  roll = lambda m, S, h: graphed(m, S)
  gathered = [torch.empty(B, H, dtype=dtype, device=dev) for _ in range(world)] if world > 1 else None

  def fence():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(max(1, -(-args.warmup // H))):
    out = roll(mx, Sxx, H)
  if world > 1:
    dist.all_gather(gathered, out[2].contiguous())
  fence()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  e0.record()
  for _ in range(steps // H):
    m_H, S_H, cost = roll(mx, Sxx, H)
    if world > 1:
      dist.all_gather(gathered, cost.contiguous())
  e1.record()
  fence()
  elapsed = time.perf_counter() - t0
  if world > 1:
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
  eager_roll.drift.check_status(B)
  if not (torch.isfinite(cost).all() and torch.isfinite(S_H).all()):
    raise SystemExit("non-finite composed rollout")
  dev_ms = e0.elapsed_time(e1) / steps
  # the same rollouts enqueued eagerly (one mm_rollout_composed call each), for the record
  torch.cuda.synchronize()
  te = time.perf_counter()
  for _ in range(steps // H):
    eager_roll(mx, Sxx, H)
  torch.cuda.synchronize()
  eager_ms = 1e3 * (time.perf_counter() - te) / steps
  model_bytes = float(drift.packed(dtype, True, dev).nbytes + pol.packed(dtype, False, dev).nbytes)
  out = {"metric": "moment_matched_rollout_step_elements_per_sec", "value": round(B * world * steps / elapsed, 2),
         "unit": "rollout step-elements/s (B*H per rollout second)", "n_gpus": world, "steps": steps, "warmup": args.warmup,
         "ms_per_step": round(1e3 * elapsed / steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
         "dtype": "f64", "data": "synthetic",
         "config": {"workload": c["label"], "N": c["M"], "N_policy": c["Mpol"], "d": 6, "D": 4, "H": H, "B_per_gpu": B, "B_total": B * world,
                    "parallelism": f"dp{world} over B (weak)", "rollouts_timed": steps // H,
                    "collectives_timed": steps // H if world > 1 else 0, "steps_requested": args.steps,
                    "launch": "HIP graph replay of one mm_rollout_composed call per rollout", "eager_ms_per_step": round(eager_ms, 4)},
         "roofline": {"bound": "hbm", "kernel": "mm_rollout_composed: chain of ~16 dependent launches per step (2 GP moment matches + 4 composition kernels + cost)",
                      "achieved": round(model_bytes / (dev_ms * 1e-3) / 1e9, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                      "frac": round(model_bytes / (dev_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 6), "traffic": None, "kernel_ms": round(dev_ms, 4),
                      "bytes_per_launch": model_bytes,
                      "note": "B = 1: every kernel runs for a few microseconds on a fraction of one XCD; the step is bound by the "
                              "dependency chain of its launches (MI355X_MICROARCH.md 'boundary': ~1.5-1.9 us each), not by a "
                              "bandwidth or issue ceiling.  `achieved` = packed model bytes touched once per step / step time"}}
  if rank == 0 and world == 1:
    # what the only real caller of this rollout does with it (examples/cartpole_swingup/train_utils.py:91-105): the
    # gradient of the loss w.r.t. the policy's parameters -- taped native rollout + reverse sweep
    # (mm_rollout_composed_backward), forward + backward replayed from one HIP graph; untimed by the metric, reported beside it
    from gpflowpilco_amd import bijectors as tfb, dynamics, models as gpm
    from gpflowpilco_amd.components import GaussianObjective, TrigonometricEncoder
    from gpflowpilco_amd.loops import GraphedPolicyLoss, get_state_initializer, policy_loss_closure
    kern = pol.latent_kernels[0]
    prm = [pol.q_mu, pol.inducing_variable.inducing_variable.Z, kern.lengthscales, kern.variance]
    for p_ in prm:
      p_.requires_grad_(True)
    policy = gpm.InverseLinkWrapper(gpm.KernelRegressor(pol), invlink=tfb.Chain([tfb.Scale(scale), tfb.Shift(shift), tfb.NormalCDF()]))
    system = dynamics.DynamicalSystem(drift=drift, policy=policy, encoder=TrigonometricEncoder(active_dims=active),
                                      solver=dynamics.MomentMatchingEuler())
    closure = policy_loss_closure(system, GaussianObjective(target=t(target), precis=t(precis)), get_state_initializer(mx, Sxx), H)
    gl = GraphedPolicyLoss(closure, prm)
    for _ in range(3):
      gl.loss_and_grad()
    torch.cuda.synchronize()
    tg = time.perf_counter()
    nrep = max(1, steps // H)
    for _ in range(nrep):
      gl.loss_and_grad()
    torch.cuda.synchronize()
    out["config"]["policy_gradient"] = {
        "forward_backward_ms_per_step": round(1e3 * (time.perf_counter() - tg) / (nrep * H), 4),
        "what": "loss + d loss / d (q_mu, Z, lengthscales, variance) of the policy, H-step rollout, one HIP graph replay per evaluation "
                "(loops.GraphedPolicyLoss over mm_rollout_composed_taped + mm_rollout_composed_backward)",
        "grad_norm": float(torch.sqrt(sum((g_ * g_).sum() for g_ in gl._grads)))}
    for p_ in prm:
      p_.requires_grad_(False)
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    from oracle import mm_compose_oracle as co
    from oracle import mm_oracle as mo
    po_d = mo.SVGPParams(Z=np.broadcast_to(drift_s.Z, (4, c["M"], 6)).copy(), lengthscales=drift_s.lengthscales, variance=drift_s.variance,
                         q_mu=drift_s.q_mu, q_sqrt=drift_s.q_sqrt, whiten=True)
    po_p = mo.SVGPParams(Z=pol_s.Z[None].copy(), lengthscales=pol_s.lengthscales, variance=pol_s.variance, q_mu=pol_s.q_mu,
                         q_sqrt=pol_s.q_sqrt, whiten=True)
    pol_fn = lambda st: co.mm_policy(st, po_p, scale, shift)
    t0 = time.perf_counter()
    nroll = 0
    while True:
      loss_o, traj_o = co.policy_rollout_loss(mu_np[:1], S_np[:1], po_d, pol_fn, active, target, precis, H, keep=True)
      nroll += 1
      if time.perf_counter() - t0 > 10.0:
        break
    tc = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": round(nroll * H / tc, 3), "unit": out["unit"], "cores": os.cpu_count(), "kind": "port",
                           "sample": f"the whole config on the host: oracle/mm_compose_oracle.py policy_rollout_loss (literal reference "
                                     f"algorithm per GP match, numpy), B=1, {nroll} rollout(s) of H={H} steps, {tc:.1f}s"}
    out["parity"] = {"vs": "fp64 CPU oracle rollout, same inputs, element 0",
                     "max_abs_err": {"mu_H": float(np.abs(m_H[:1].cpu().numpy() - traj_o[-1][0]).max()),
                                     "Sigma_H": float(np.abs(S_H[:1].cpu().numpy() - traj_o[-1][1]).max()),
                                     "loss": float(np.abs(cost[:1].sum(1).cpu().numpy() - loss_o).max())},
                     "max_abs": {"mu_H": float(np.abs(traj_o[-1][0]).max()), "Sigma_H": float(np.abs(traj_o[-1][1]).max()),
                                 "loss": float(np.abs(loss_o).max())}}
  if rank == 0:
    emit(out)
  if world > 1:
    dist.destroy_process_group()


def grad_bench(args, rank, world, dev, dist):
  """Row f-1 at C3 shape: one step = moment match forward + its backward on the f32 pack (csrc/mm_bwd_f32.hip for the
  off-diagonal pairs, the f64 sweep of csrc/mm_backward.hip for the diagonal pairs), B elements per step; the batch shards
  over the ranks with no collective on the data path (the gradients stay with their elements)."""
  import numpy as np
  import torch
  from gpflowpilco_amd import _lib as F, ops
  from gpflowpilco_amd.distributed import shard_range
  from gpflowpilco_amd.synthetic import make_inputs, make_svgp
  c = dict(GRAD)
  scaling = args.scaling or c["scaling"]
  L, M, d, dtype = c["L"], c["M"], c["d"], torch.float32
  rec = RECIPES[args.recipe or c["recipe"]]
  Bcfg = args.batch or c["B"]
  if scaling == "strong":
    B_total = Bcfg
    lo, hi = shard_range(B_total, rank, world)
    B = hi - lo
  else:
    B, B_total, lo = Bcfg, Bcfg * world, rank * Bcfg
  steps, ndraw = args.steps, 8

  def build_model():
    syn_ = make_svgp(L, M, d, seed=c["seed"], device=str(dev), ls_bounds=rec["ls_bounds"], stable=rec["stable"])
    model_ = syn_.to_model(dev)
    return syn_, model_, model_.packed(dtype, True, dev)
  syn, model, pm = one_rank_at_a_time(build_model, rank, world, dist, args.rehearse_gloo)
  if not ops.backward_supported(pm):
    raise SystemExit("the f32 pack's backward needs d <= 8")
  lo_mu, hi_mu = (0.0, 1.0) if rec["independent"] else (0.3, 0.7)
  mu_np, S_np = make_inputs(B_total * ndraw, d, seed=2000 + c["seed"], scale=0.1, lo=lo_mu, hi=hi_mu)
  draws_mu = torch.tensor(np.ascontiguousarray(mu_np.reshape(ndraw, B_total, d)[:, lo:lo + B]), dtype=dtype, device=dev)
  draws_S = torch.tensor(np.ascontiguousarray(S_np.reshape(ndraw, B_total, d, d)[:, lo:lo + B]), dtype=dtype, device=dev)
  gen = torch.Generator(device="cpu").manual_seed(c["seed"])
  g1 = torch.randn(B_total, L, generator=gen, dtype=torch.float64)[lo:lo + B].to(dev)
  g2 = torch.randn(B_total, L, L, generator=gen, dtype=torch.float64)[lo:lo + B].to(dev)
  g3 = torch.randn(B_total, d, L, generator=gen, dtype=torch.float64)[lo:lo + B].to(dev)
  flags = ops.make_flags(True, True)
  Ev = lambda: torch.cuda.Event(enable_timing=True)
  seg = {"forward": 0.0, "bwd_diag_f64": 0.0, "bwd_offdiag_f32": 0.0, "bwd_rest": 0.0}
  last = {}

  def one_step(k, timed):
    mu, S = draws_mu[k % ndraw], draws_S[k % ndraw]
    e = [Ev() for _ in range(5)] if timed else None
    if timed: e[0].record()
    f1, Sff, cr = ops.moment_match(pm, mu, S)
    g = pm.workspace_generation(B, flags)
    if timed: e[1].record()
    # the backward in its three stages (the same kernels as one call without stage flags, which the tests use):
    # the forward's q stage is still on the workspace (forward_generation)
    ops.moment_match_backward(pm, mu, S, g1, g2, g3, True, True, forward_generation=g, stages=F.MM_STAGE_DIAG)
    if timed: e[2].record()
    ops.moment_match_backward(pm, mu, S, g1, g2, g3, True, True, forward_generation=g, stages=F.MM_STAGE_OFFDIAG)
    if timed: e[3].record()
    gmu, gS = ops.moment_match_backward(pm, mu, S, g1, g2, g3, True, True, forward_generation=g, stages=F.MM_STAGE_FINALIZE)
    if timed: e[4].record()
    last["out"] = (f1, Sff, gmu, gS)
    return e

  def fence():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for k in range(max(1, args.warmup)):
    one_step(k, False)
  fence()
  pm.check_status(B)
  t0 = time.perf_counter()
  evs = [one_step(k, True) for k in range(steps)]
  fence()
  elapsed = time.perf_counter() - t0
  if world > 1:
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
  for e in evs:
    for name, i in (("forward", 0), ("bwd_diag_f64", 1), ("bwd_offdiag_f32", 2), ("bwd_rest", 3)):
      seg[name] += e[i].elapsed_time(e[i + 1]) / steps
  f1, Sff, gmu, gS = last["out"]
  if not all(bool(torch.isfinite(t).all()) for t in (f1, Sff, gmu, gS)):
    raise SystemExit("non-finite outputs")
  # one-call backward == staged backward (same kernels, same order)
  mu, S = draws_mu[(steps - 1) % ndraw], draws_S[(steps - 1) % ndraw]
  gmu1, gS1 = ops.moment_match_backward(pm, mu, S, g1, g2, g3, True, True)
  same = float((gmu1 - gmu).abs().amax()) <= 1e-12 * float(gmu1.abs().amax()) and float((gS1 - gS).abs().amax()) <= 1e-12 * float(gS1.abs().amax())
  # the product's default for a differentiating caller: value AND sums from one pair of sweeps, then the chain rule alone
  def one_sweep_step(k, ev=None):
    mu_k, S_k = draws_mu[k % ndraw], draws_S[k % ndraw]
    if ev: ev[0].record()
    f1_, Sff_, cr_, sums_, g_ = ops.moment_match_with_sums(pm, mu_k, S_k)
    if ev: ev[1].record()
    gm_, gS_ = ops.moment_match_backward(pm, mu_k, S_k, g1, g2, g3, True, True, forward_generation=g_, sums=sums_)
    if ev: ev[2].record()
    return Sff_, gm_, gS_
  one_sweep_step(0)
  evs1 = []
  for k in range(steps):
    ev = [Ev() for _ in range(3)]
    Sff1, gm1, gS1s = one_sweep_step(k, ev)
    evs1.append(ev)
  torch.cuda.synchronize()
  os_fwd = sum(e[0].elapsed_time(e[1]) for e in evs1) / steps
  os_bwd = sum(e[1].elapsed_time(e[2]) for e in evs1) / steps
  one_sweep = {"value_and_sums_ms": round(os_fwd, 4), "chain_rule_ms": round(os_bwd, 4), "total_ms": round(os_fwd + os_bwd, 4),
               "over_forward": round((os_fwd + os_bwd) / seg["forward"], 2),
               "Sff_max_diff_over_scale_vs_forward": float((Sff1 - Sff).abs().amax() / Sff.abs().amax()),
               "gradient_equals_two_pass": bool(torch.equal(gm1, gmu) and torch.equal(gS1s, gS)),
               "what": "mm_moment_match_with_sums + mm_moment_match_backward(MM_SUMS_CURRENT): the backward's sweeps do not depend on the "
                       "incoming gradient and contain the forward's sums, so value + gradient need ONE diagonal and ONE off-diagonal "
                       "sweep (autodiff.MomentMatchFunction's default); the timed region above is the two-pass form, stage by stage"}
  Po, Mp = L * (L - 1) // 2, -(-M // 128) * 128
  nmono = 1 + d + d * (d + 1) // 2
  e_off = float(B) * Po * M * M
  e_diag = float(B) * L * M * M                        # the backward sweeps every entry of a diagonal pair (no symmetry)
  flops_off = e_off * (2 * d + 12 + 2 * nmono)
  flops_diag = e_diag * (2 * d + 12 + 2 * (d + 1) + 6)
  pmc, pmc_src = load_pmc("c3_grad_" + (args.recipe or c["recipe"]), fallback="c3_grad")
  pscale, pmc_src = pmc_scale(pmc, pmc_src, B, c["B"])
  why_off = ("E_o (2d + 12 + 2 n_mono): per entry the bilinear form (2d), the remainder of e^b (12, as SURVEY 8d prices the forward) and "
             f"the aggregate product against the {nmono} monomials of degree <= 2 (2 n_mono); E_o = B Po M^2.  Executed: 30 "
             "v_mfma_f32_32x32x16_bf16 per 64 x 32 wave tile (3-way split bilinear product 6, hi/lo split aggregate product 24) = 480 "
             "bf16 flop per entry")
  r_off = roofline_block("k_bwd_rem_f32", seg["bwd_offdiag_f32"], flops_off, PEAK_TFLOPS["bf16"], pmc, pmc_src, pscale, "bf16",
                         "k_bwd_rem_f32", why_off)
  r_off["executed_mfma_tflops"] = round(e_off * 480.0 / (seg["bwd_offdiag_f32"] * 1e-3) / 1e12, 1) if seg["bwd_offdiag_f32"] > 0 else None
  r_off["executed_mfma_frac_of_bf16_peak"] = (round(r_off["executed_mfma_tflops"] / PEAK_TFLOPS["bf16"], 4)
                                              if r_off["executed_mfma_tflops"] else None)
  r_off.pop("frac_null_reason", None)
  r_diag = roofline_block("k_bwd_diag", seg["bwd_diag_f64"], flops_diag, PEAK_TFLOPS["f64"], pmc, pmc_src, pscale, "f64", "k_bwd_diag",
                          "E_d (2d + 12 + 2 (d + 1) + 6): bilinear form, expm1, the column product Omega^T (zc | 1), the C-weighted sums; "
                          "E_d = B L M^2 (every entry: column sums use no symmetry)")
  r_diag.pop("frac_null_reason", None)
  ms = 1e3 * elapsed / steps
  out = {"metric": "moment_match_forward_backward_steps_per_sec", "value": round(B_total * steps / elapsed, 2),
         "unit": "step-elements/s (B per forward+backward step)", "n_gpus": world, "steps": steps, "warmup": args.warmup,
         "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
         "data": "synthetic",
         "config": {"workload": c["label"], "recipe": rec["text"], "N": M, "d": d, "D": L, "B_total": B_total, "B_this_rank": B,
                    "parallelism": f"dp{world} over B ({scaling}); no collective on the data path",
                    "row": "SURVEY 8(f) f-1 -- not BASELINE.json's metric", "staged_equals_one_call": same,
                    "forward_ms": round(seg["forward"], 4), "backward_ms": round(ms - seg["forward"], 4),
                    "backward_over_forward": round((ms - seg["forward"]) / seg["forward"], 2),
                    "forward_backward_over_forward": round(ms / seg["forward"], 2),
                    "one_sweep_value_and_gradient": one_sweep},
         "segments_ms": {k: round(v, 4) for k, v in seg.items()},
         "roofline": r_off, "roofline_other": r_diag}
  if not same:
    raise SystemExit("staged backward differs from the one-call backward")
  if rank == 0 and not args.no_cpu_baseline and world == 1:
    # the reference's way: autograd through the materialised [P, M, M] evaluation (autodiff.moment_match_torch, the torch
    # transliteration of models.py:200-299) on the host cores, float64, ONE batch element of the same workload
    from gpflowpilco_amd import autodiff
    torch.set_num_threads(os.cpu_count() or 1)
    Z, ls, var, beta, Cm, mc = (None if t is None else t.detach().cpu() for t in model.precompute(dev))
    mu_c = draws_mu[0, :1].double().cpu().requires_grad_(True)
    S_c = draws_S[0, :1].double().cpu().requires_grad_(True)
    tc0 = time.perf_counter()
    f1c, Sffc, crc = autodiff.moment_match_torch(mu_c, S_c, Z, ls, var, beta, Cm, mc, True, True)
    ((g1[:1].cpu() * f1c).sum() + (g2[:1].cpu() * Sffc).sum() + (g3[:1].cpu() * crc).sum()).backward()
    tc = time.perf_counter() - tc0
    out["cpu_baseline"] = {"value": round(1.0 / tc, 4), "unit": out["unit"], "cores": os.cpu_count(), "kind": "port",
                           "sample": f"ONE batch element of the same workload (forward + torch autograd backward of the materialised "
                                     f"[36, {M}, {M}] float64 evaluation, {tc:.1f} s on {os.cpu_count()} threads)"}
    # parity of the SAME element: the HIP backward on the f32 pack (B = 1 call) against that float64 autograd gradient, and
    # against the f64 pack of the same model (every pair swept in f64)
    m1, S1 = draws_mu[0, :1].contiguous(), draws_S[0, :1].contiguous()
    pm.status().zero_()
    ga = ops.moment_match_backward(pm, m1, S1, g1[:1], g2[:1], g3[:1], True, True)
    pm64 = model.packed(torch.float64, True, dev)
    gb = ops.moment_match_backward(pm64, m1.double(), S1.double(), g1[:1], g2[:1], g3[:1], True, True)
    gc = (mu_c.grad, 0.5 * (S_c.grad + S_c.grad.transpose(1, 2)))
    rel = lambda x, y: float((x.cpu() - y.cpu()).abs().amax() / y.abs().amax())
    out["parity"] = {"vs": "float64 torch autograd of the materialised evaluation on the CPU (autodiff.moment_match_torch), same element, same "
                           "f32-rounded state; and the HIP backward on an f64 pack of the same model", "B": 1,
                     "max_err_over_scale_vs_cpu_autograd": {"g_mu": rel(ga[0], gc[0]), "g_Sigma": rel(ga[1], gc[1])},
                     "max_err_over_scale_vs_f64_pack": {"g_mu": rel(ga[0], gb[0]), "g_Sigma": rel(ga[1], gb[1])},
                     "f64_pack_vs_cpu_autograd": {"g_mu": rel(gb[0], gc[0]), "g_Sigma": rel(gb[1], gc[1])},
                     "items_routed_to_f64_in_this_backward": pm.routed()[1]}
  if rank == 0:
    emit(out)
  if world > 1:
    dist.destroy_process_group()


def pathwise_bench(args, rank, world, dev, dist):
  """configs[4]: one step = one Euler step of all local sample paths (HBM-bound weight stream)."""
  import torch
  from gpflowpilco_amd.distributed import shard_range
  from gpflowpilco_amd.pathwise import PathwiseSVGP
  from gpflowpilco_amd.synthetic import make_svgp
  c = dict(PATHWISE)
  scaling = args.scaling or c["scaling"]
  Scfg = args.batch or c["S"]
  L, M, d, K, H, dtype = c["L"], c["M"], c["d"], c["K"], c["H"], torch.float32
  if scaling == "strong":
    S_total = Scfg
    lo, hi = shard_range(S_total, rank, world)
    S = hi - lo
  else:
    S, S_total = Scfg, Scfg * world
  steps = max(H, -(-args.steps // H) * H)
  g = torch.Generator(device=dev).manual_seed(c["seed"] + rank)

  def build_paths():
    syn = make_svgp(L, M, d, seed=c["seed"], device=str(dev), ls_bounds=(0.7, 3.0))
    base = syn.to_model(dev)
    model = PathwiseSVGP(kernel=base.kernel, inducing_variable=base.inducing_variable, q_mu=base.q_mu,
                         q_sqrt=base.q_sqrt, whiten=True, num_latent_gps=L)
    return model.generate_paths(S, K, dtype=dtype, device=dev, generator=g)
  paths = one_rank_at_a_time(build_paths, rank, world, dist, args.rehearse_gloo)
  x0 = 0.3 + 0.4 * torch.rand(S, d, dtype=dtype, device=dev, generator=g)
  target = torch.full((d,), 0.5, dtype=dtype, device=dev)
  Smax = -(-S_total // world)

  def rollout_steps(n):
    x, traj = paths.rollout(x0, n, dt=1.0, keep_trajectory=True)
    err = traj - target
    return -torch.exp(-2.0 * (err * err).sum(-1)).T.contiguous()          # [S, n] per-step sample costs

  def fence():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  rollout_steps(max(1, min(args.warmup, H)))
  pad = torch.zeros(Smax, H, dtype=dtype, device=dev)
  outs = [torch.empty_like(pad) for _ in range(world)] if world > 1 else None
  if world > 1:
    dist.all_gather(outs, pad)                                            # untimed channel bring-up
  fence()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  done = 0
  ncoll = 0
  e0.record()
  while done < steps:
    cost = rollout_steps(H)
    if world > 1:
      pad[:S].copy_(cost)
      dist.all_gather(outs, pad)                                          # [S_local, H] -> every rank, once per rollout
      ncoll += 1
    done += H
  e1.record()
  fence()
  elapsed = time.perf_counter() - t0
  if world > 1:
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
  if not torch.isfinite(cost).all():
    raise SystemExit("non-finite sample costs")
  k_ms = e0.elapsed_time(e1) / steps
  bytes_per_launch = float(S) * L * (K + M) * 4
  achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
  pmc, pmc_src = load_pmc("c5")
  pscale, pmc_src = pmc_scale(pmc, pmc_src, S, c["S"])
  got = pmc_kernel(pmc, "k_pathwise")
  out = {"metric": "pathwise_rollout_sample_steps_per_sec", "value": round(S_total * steps / elapsed, 1),
         "unit": "sample step-elements/s (S*H per rollout second)", "n_gpus": world, "steps": steps,
         "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / steps, 4), "higher_is_better": True,
         "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
         "config": {"workload": c["label"], "N": M, "d": d, "D": L, "K": K, "H": H, "S_total": S_total, "S_this_rank": S,
                    "parallelism": f"dp{world} over S ({scaling})", "rollouts_timed": steps // H, "collectives_timed": ncoll,
                    "steps_requested": args.steps},
         "roofline": {"bound": "hbm", "kernel": "k_pathwise", "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS,
                      "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 4),
                      "traffic": got[1]["counters"]["hbm_bytes"] * pscale if got and "hbm_bytes" in got[1]["counters"] else None, "pmc": pmc_src,
                      "kernel_ms": round(k_ms, 4), "bytes_per_launch": bytes_per_launch,
                      "note": "achieved = algorithmic bytes S*L*(K+M)*4 per step / step time (the step time includes the small cost kernels "
                              "between launches)"}}
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    out.update(pathwise_extras(args, dev, c, paths, x0, dtype))
  if rank == 0:
    emit(out)
  if world > 1:
    dist.destroy_process_group()


def pathwise_extras(args, dev, c, paths, x0, dtype):
  """N == 1, rank 0: parity and the CPU baseline of the C5 line (the oracle on a slice of sample paths drawn on the host), the
  f64-mode figure of the same rollout, and row f-3's policy rollout: forward, taped forward (Jacobians emitted in the same
  stream pass) and the one-kernel reverse sweep."""
  import numpy as np
  import torch
  from gpflowpilco_amd import models as gp, ops
  from gpflowpilco_amd.pathwise import PathwiseSVGP, PolicyRollout, paths_from_arrays
  from gpflowpilco_amd.synthetic import make_policy, make_svgp
  from oracle import pathwise_oracle as pw
  from oracle.mm_oracle import SVGPParams
  L, M, d, K, H = c["L"], c["M"], c["d"], c["K"], c["H"]
  res = {}
  ev = lambda: torch.cuda.Event(enable_timing=True)

  def timed(fn, reps=2):
    fn()
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(reps):
      fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
  # ---- parity + CPU baseline: the numpy oracle on Sc sample paths of the same model, same kernels on the same path tensors
  syn = make_svgp(L, M, d, seed=c["seed"], device=str(dev), ls_bounds=(0.7, 3.0))
  po = SVGPParams(Z=np.broadcast_to(syn.Z, (L, M, d)).copy(), lengthscales=syn.lengthscales, variance=syn.variance,
                  q_mu=syn.q_mu, q_sqrt=syn.q_sqrt, whiten=True)
  rng = np.random.default_rng(c["seed"])
  Sc, Hc = 64, 3                  # (an un-stabilised drift-only sample rollout leaves the data's support after a few unit steps)
  pths = pw.draw_paths(rng, po, Sc, K)
  xc = rng.uniform(0.3, 0.7, size=(Sc, d))
  t0 = time.perf_counter()
  xo, trajo = pw.rollout(pths, po, xc, Hc, dt=1.0, keep=True)
  tc = time.perf_counter() - t0
  res["cpu_baseline"] = {"value": round(Sc * Hc / tc, 1), "unit": "sample step-elements/s", "cores": os.cpu_count(), "kind": "port",
                         "sample": f"oracle/pathwise_oracle.py (fp64 numpy: the published decoupled-sampling evaluation; parity unpinned), "
                                   f"{Sc} sample paths x {Hc} Euler steps of the same model (N={M}, K={K}, d=D={d}), {tc:.2f}s"}
  par = {"vs": "fp64 numpy oracle on identical path tensors", "S": Sc, "H": Hc}
  for nm, dt_ in (("f32", torch.float32), ("f64", torch.float64)):
    gpp = paths_from_arrays(pths.omega, pths.phase, pths.w, pths.v, po.Z, po.lengthscales, po.variance, None, dtype=dt_, device=dev)
    xg, tg = gpp.rollout(torch.tensor(xc, dtype=dt_, device=dev), Hc, dt=1.0, keep_trajectory=True)
    f1 = gpp(torch.tensor(xc, dtype=dt_, device=dev)).double().cpu().numpy()
    fo = pw.eval_paths(pths, po, xc)
    par[nm] = {"one_evaluation_max_abs_err": float(np.abs(f1 - fo).max()), "one_evaluation_max_abs": float(np.abs(fo).max()),
               "rollout_max_abs_err": float(np.abs(tg.double().cpu().numpy() - trajo).max()), "rollout_max_abs": float(np.abs(trajo).max())}
    # what the kernel itself says about these values (mm_pathwise_eval_bound: the pass with the sum of the absolute terms)
    fb, errb = gpp.eval_with_bound(torch.tensor(xc, dtype=dt_, device=dev))
    par[nm]["rounding_bound_max"] = float(errb.max())
    par[nm]["worst_error_over_bound"] = float((np.abs(fb.double().cpu().numpy() - fo) / errb.double().cpu().numpy()).max())
  res["parity"] = par
  # the timed f32 paths themselves: how many (sample, latent) values of ONE evaluation the bound flags at 1e-3 of max|f|
  _, _, nflag = paths.flagged(x0, 1e-3)
  par["f32_values_flagged_at_1e-3_of_max_f"] = {"flagged": nflag, "of": int(paths.num_samples * L),
                                                "note": "rounding bound of the f32 weight stream > 1e-3 max|f| (all of them at M = 2000: "
                                                        "the f64 mode is the accurate path)"}
  # ---- f64 mode of the headline rollout (the accurate mode: 2 x the bytes per weight)
  S = paths.num_samples
  S64 = min(S, 16384)
  base = syn.to_model(dev)
  pmodel = PathwiseSVGP(kernel=base.kernel, inducing_variable=base.inducing_variable, q_mu=base.q_mu, q_sqrt=base.q_sqrt, whiten=True,
                        num_latent_gps=L)
  g = torch.Generator(device=dev).manual_seed(c["seed"] + 7)
  p64 = pmodel.generate_paths(S64, K, dtype=torch.float64, device=dev, generator=g)
  x64 = x0[:S64].double().contiguous()
  ms64 = timed(lambda: p64.rollout(x64, 10, dt=1.0)) / 10
  res["f64_mode"] = {"S": S64, "ms_per_step": round(ms64, 4), "GBps": round(S64 * L * (K + M) * 8 / (ms64 * 1e-3) / 1e9, 1),
                     "frac_of_hbm_peak": round(S64 * L * (K + M) * 8 / (ms64 * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
  del p64
  # ---- row f-3: the pathwise POLICY rollout (encoder -> policy -> drift sample -> Euler -> cost) and its gradient.
  # nx = 6 states, one angle: ne = 7, nd = 8 drift inputs (the stream kernel's d of the headline), L = 6 latents
  nx, na, Hp = 6, 1, 10
  nd = nx + na + 1
  synd = make_svgp(nx, M, nd, seed=c["seed"] + 1, device=str(dev), ls_bounds=(0.7, 3.0))
  based = synd.to_model(dev)
  pdrift = PathwiseSVGP(kernel=based.kernel, inducing_variable=based.inducing_variable, q_mu=based.q_mu, q_sqrt=based.q_sqrt,
                        whiten=True, num_latent_gps=nx)
  pp = pdrift.generate_paths(S, K, dtype=dtype, device=dev, generator=g)
  pol = make_policy(30, nx + na, seed=c["seed"] + 2).to_model(dev)
  roll = PolicyRollout(pp, pol.packed(torch.float64, False, dev), nx=nx, active_dims=(1,), head_scale=2.0, head_shift=-0.5,
                       target=torch.full((nx + na,), 0.5), precis=torch.eye(nx + na) * 4.0)
  xp = 0.3 + 0.4 * torch.rand(S, nx, dtype=dtype, device=dev, generator=g)
  gcost = torch.full((Hp, S), 1.0 / S, dtype=torch.float64, device=dev)
  fwd = timed(lambda: roll(xp, Hp)) / Hp
  holder = {}

  def taped():
    holder["t"] = roll(xp, Hp, with_jacobians=True)[1]
  tap = timed(taped) / Hp
  bwd = timed(lambda: roll.backward(holder["t"], gcost, Hp)) / Hp
  bytes_step = float(S) * nx * (K + M) * 4
  res["next_rows"] = {"f-3": {
      "what": "pathwise policy rollout (TrigonometricEncoder -> policy mean M=30 through Chain[Scale,Shift,NormalCDF] -> drift sample -> "
              "Euler -> cost) of S sample paths, per step; taped = the stream pass also emits d f / d (e, u) [S,nx,nd]; backward = ONE "
              "kernel over the tape (no second pass over the weight stream)",
      "S": S, "nx": nx, "nd": nd, "H": Hp, "forward_ms": round(fwd, 4), "taped_forward_ms": round(tap, 4), "backward_ms": round(bwd, 4),
      "forward_backward_over_forward": round((tap + bwd) / fwd, 3),
      "forward_frac_of_hbm_peak": round(bytes_step / (fwd * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
      "taped_forward_frac_of_hbm_peak": round(bytes_step / (tap * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}}
  return res


if __name__ == "__main__":
  main()

