"""Host-side pieces of the measurement chain (no GPU): the counter summary bench.py reads, its staleness check and
the executed-work ceiling computed from it."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
  spec = importlib.util.spec_from_file_location(name, path)
  mod = importlib.util.module_from_spec(spec)
  sys.modules[name] = mod
  spec.loader.exec_module(mod)
  return mod


def test_pmc_summary_groups_by_kernel_and_tags_the_sources(tmp_path):
  ps = _load("pmc_summary_t", os.path.join(ROOT, "tools", "pmc_summary.py"))
  assert ps.short_name("void k_qred_f64_mfma<2, true, true, true>(double const*, int)") == "k_qred_f64_mfma<2, true, true, true>"
  assert ps.short_name("k_wmom_gemm(double const*, double const*)") == "k_wmom_gemm"
  d = tmp_path / "pmc" / "pass1"
  d.mkdir(parents=True)
  hdr = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",'
         '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name",'
         '"Counter_Value","Start_Timestamp","End_Timestamp"\n')
  rows = []
  for disp, (fetch, write) in enumerate(((1000.0, 10.0), (3000.0, 30.0))):
    for cname, val in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
      rows.append(f'{disp},{disp},"Agent 2",1,7,7,1024,3,"void k_pairvec<float, 8>(double const*)",256,512,0,104,0,96,"{cname}",{val},100,2100\n')
  rows.append('9,9,"Agent 2",1,7,7,64,4,"void at::native::fill(float*)",64,0,0,8,0,16,"FETCH_SIZE",5.0,1,2\n')
  (d / "1_counter_collection.csv").write_text(hdr + "".join(rows))
  out = tmp_path / "summary.json"
  sys.argv = ["pmc_summary.py", str(tmp_path / "pmc"), "--tag", "t", "-o", str(out)]
  ps.main()
  s = json.loads(out.read_text())
  assert list(s["kernels"]) == ["k_pairvec<float, 8>"]                  # torch's own kernels are not summarised
  k = s["kernels"]["k_pairvec<float, 8>"]
  assert k["dispatches"] == 2 and k["counters"]["FETCH_SIZE"] == 2000.0 * 1024 and k["counters"]["WRITE_SIZE"] == 20.0 * 1024
  assert k["counters"]["hbm_bytes"] == 2 * 2000.0 * 1024 + 20.0 * 1024    # gfx950: FETCH_SIZE counts half the bytes read
  assert s["src_hash"] == ps.src_hash(ROOT) and len(s["src_hash"]) == 16


def test_bench_ceiling_and_staleness(tmp_path, monkeypatch):
  bench = _load("bench_t", os.path.join(ROOT, "bench.py"))
  ent = {"counters": {"SQ_INSTS_MFMA": 1000.0, "SQ_INSTS_VALU": 1000.0 + 40000.0, "SQ_VALU_MFMA_BUSY_CYCLES": 32000.0,
                      "SQ_INSTS_VALU_FMA_F32": 30000.0, "SQ_INSTS_VALU_MUL_F32": 0.0, "SQ_INSTS_VALU_ADD_F32": 0.0,
                      "SQ_INSTS_VALU_FMA_F64": 0.0, "SQ_INSTS_VALU_MUL_F64": 0.0, "SQ_INSTS_VALU_ADD_F64": 0.0,
                      "SQ_INSTS_VALU_TRANS_F32": 0.0}}
  ce = bench.executed_ceiling(ent, "bf16")
  # 32000 MFMA pipe cycles + (30000 - 3 x 1000) f32 FMA-class x 4 cycles + (10000 other x 4 - 24 x 1000) cycles
  want = (32000.0 + 27000.0 * 4.0 + 16000.0) / bench.N_SIMD / bench.PEAK_CLOCK_HZ * 1e3
  assert abs(ce["ceiling_ms"] - want) < 1e-12 * want
  assert ce["mix"]["valu_other"] == 10000.0
  ent64 = {"counters": {"SQ_INSTS_MFMA": 100.0, "SQ_INSTS_VALU": 100.0 + 1500.0, "SQ_VALU_MFMA_BUSY_CYCLES": 6400.0,
                        "SQ_INSTS_VALU_FMA_F32": 0.0, "SQ_INSTS_VALU_FMA_F64": 1000.0}}
  ce64 = bench.executed_ceiling(ent64, "f64")               # f64 pipes are one datapath: everything adds
  assert abs(ce64["ceiling_ms"] - (6400.0 + 1000.0 * 5.0 + 500.0 * 4.0) / bench.N_SIMD / bench.PEAK_CLOCK_HZ * 1e3) < 1e-18
  # a summary taken on other kernel sources is refused, with the reason
  monkeypatch.setattr(bench, "ROOT", str(tmp_path))
  (tmp_path / "profiles").mkdir()
  (tmp_path / "profiles" / "r02_pmc_c3.json").write_text(json.dumps({"src_hash": "0" * 16, "kernels": {}}))
  monkeypatch.setattr(bench, "src_hash", lambda: "f" * 16)
  pmc, why = bench.load_pmc("c3")
  assert pmc is None and "stale" in why
  (tmp_path / "profiles" / "r02_pmc_c3.json").write_text(json.dumps({"src_hash": "f" * 16, "kernels": {"k_x<1>": {"dur_us_under_pmc": 1.0, "dispatches": 2, "counters": {}}}}))
  pmc, src = bench.load_pmc("c3")
  assert pmc is not None and bench.pmc_kernel(pmc, "k_x")[0] == "k_x<1>" and bench.pmc_kernel(pmc, "k_y") is None
  assert bench.load_pmc("c9")[0] is None


def test_pmc_counters_scale_with_the_batch_of_the_run():
  """bench.pmc_scale: counters collected at another per-rank batch are scaled linearly and the `pmc` field says so."""
  import bench
  pmc = {"bench_args": "--config c4 --batch 32"}
  assert bench.pmc_scale(pmc, "profiles/x.json", 32, 256) == (1.0, "profiles/x.json")
  sc, note = bench.pmc_scale(pmc, "profiles/x.json", 256, 256)
  assert sc == 8.0 and "scaled x8" in note and "collected at 32" in note
  assert bench.pmc_scale({"bench_args": "--config c3"}, "p", 256, 256) == (1.0, "p")
  assert bench.pmc_scale({"bench_args": "--config c3"}, "p", 64, 256)[0] == 0.25
  assert bench.pmc_scale(None, "not collected", 64, 256) == (1.0, "not collected")


def test_bench_refuses_to_run_without_a_gpu():
  """`python bench.py` on a host without a GPU: a one-line refusal and a non-zero exit code -- no CPU fallback, no
  traceback from a half-initialised run."""
  import os
  import subprocess
  import sys
  import torch
  if torch.cuda.is_available():
    import pytest
    pytest.skip("a GPU is present")
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "c1_closed", "--steps", "1"],
                     capture_output=True, text=True, timeout=300)
  assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout) and "Traceback" not in r.stderr


def _canned_result():
  """A result dict the size and shape of round 4's (three regimes, four roofline blocks each, prose definitions), with the
  values a bad run can produce: NaN, +-Infinity, numpy scalars."""
  import numpy as np
  prose = "executed-work ceiling / measured kernel time; ceiling = (MFMA pipe cycles + VALU issue cycles that cannot hide) " * 4
  roof = {"bound": "mfma", "kernel": "k_qred_f32_mfma<1, true>", "kernel_ms": 6.1346, "algorithmic_flops_per_launch": 8.02816e11,
          "flops_model": prose, "achieved": 130.87, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.832, "traffic": 590144000.0,
          "issue_frac": float("nan"), "issue_ceiling_ms": float("inf"), "issue_frac_definition": prose,
          "instruction_mix_per_launch": {k: 1.0e7 for k in ("mfma", "valu_f32", "valu_f64", "valu_trans", "valu_other")},
          "pmc": "profiles/r05_pmc_c3_baseline.json", "mfma_busy_frac": np.float32(0.2),
          "pipes": {"mfma_bf16": {"executed_tflops": 440.5, "peak": 2500.0, "frac": 0.1762},
                    "valu_f32": {"executed_tflops": 52.7, "peak": 157.3, "frac": 0.3354, "note": prose}},
          "measured_clock_ghz": 1.977, "issue_frac_at_measured_clock": 0.69, "measured_clock_definition": prose,
          "entries_per_launch": 2.8672e10}
  par = {"vs": prose, "B": 1, "recipe": "baseline", "max_abs_err": {"f1": 6.6e-8, "Sff": float("-inf"), "cross_pre": np.float64(4.3e-7)},
         "max_abs": {"f1": 1.56, "Sff": 0.28, "cross_pre": 4.28}}
  regime = {"recipe": "baseline", "recipe_text": prose, "value": 23920.24, "ms_per_step": 10.7, "steps": 20,
            "segments_ms": {"q_stage": 1.08, "diag": 3.62, "offdiag": 6.13, "tail": 0.02}, "roofline": roof, "roofline_other": dict(roof),
            "roofline_step": {"note": prose, "frac": 0.55}, "parity": {"vs_fused_restatement": par, "step_f32_vs_f64_mode": par}}
  return {"metric": "moment_matched_rollout_step_elements_per_sec", "value": 23920.24,
          "unit": "rollout step-elements/s (B*H per rollout second)", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 10.7022,
          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
          "config": {"workload": prose, "N": 2000, "d": 8, "D": 8, "H": 40, "B_per_gpu": 256, "B_total": 256, "recipe": "baseline",
                     "value_is": prose, "offdiag_items_one_rollout": {"collapsed": 1, "meaning": prose}},
          "segments_ms": regime["segments_ms"], "roofline": roof, "roofline_other": dict(roof), "roofline_step": regime["roofline_step"],
          "roofline_q_stage": {"per_kernel": {f"k{i}": {"hbm_bytes": 1.0e8} for i in range(8)}},
          "cpu_baseline": {"value": 0.0766, "unit": "rollout step-elements/s (B*H per rollout second)", "cores": 256, "kind": "port", "sample": prose},
          "cpu_baseline_matched": {"value": 0.2184, "unit": "x", "cores": 256, "kind": "port", "sample": prose},
          "parity": dict(par, vs_fused_restatement=par, step_f32_vs_f64_mode=par),
          "next_rows": {"f-1": {"forward_ms": 10.78, "forward_backward_ms": 34.54, "what": prose, "B": 256}},
          "regimes": {n: dict(regime) for n in ("baseline", "pilco", "worst")}}


def test_the_bench_line_is_one_short_strict_json_line(tmp_path, capsys):
  """the driver's record of round 4 came back `parsed: null` because bench.py printed a 22 KB line.
  The line is now compact (numbers; prose and per-regime detail go to bench_detail.json), strict JSON (no NaN / Infinity),
  one line, and carries what the contract asks for."""
  bench = _load("bench_line_t", os.path.join(ROOT, "bench.py"))
  res = _canned_result()
  assert len(json.dumps(res, default=float)) > 15000                                 # the canned result is round-4 sized
  detail = tmp_path / "detail.json"
  bench.emit(res, detail_path=str(detail))
  printed = capsys.readouterr().out
  assert printed.endswith("\n") and printed.count("\n") == 1
  line = printed.strip()
  assert len(line) < 6000 and len(line) <= bench.LINE_LIMIT

  def refuse(tok):
    raise AssertionError(f"non-strict JSON constant {tok} in the bench line")
  got = json.loads(line, parse_constant=refuse)
  for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
    assert k in got, k
  assert got["value"] == 23920.24 and got["steps"] == 20 and got["warmup"] == 5 and got["config"]["N"] == 2000
  r = got["roofline"]
  assert r["bound"] == "mfma" and r["frac"] == 0.832 and r["kernel_ms"] == 6.1346 and r["traffic"] == 590144000.0
  assert r["issue_frac"] is None                                                     # NaN -> null
  assert r["pipes"] == {"mfma_bf16": 0.1762, "valu_f32": 0.3354} and abs(r["mfma_busy_frac"] - 0.2) < 1e-6
  assert not any("definition" in k or k == "flops_model" for k in r)                 # no prose in the roofline block
  assert got["cpu_baseline"]["cores"] == 256 and got["cpu_baseline"]["kind"] == "port" and len(got["cpu_baseline"]["sample"]) <= 160
  assert got["parity"]["max_abs_err"] == {"f1": 6.6e-08, "Sff": None, "cross_pre": 4.3e-07} and "vs" not in got["parity"]
  assert got["regimes"]["pilco"] == {"value": 23920.24, "ms_per_step": 10.7, "kernel_ms": 6.1346, "frac": 0.832}
  assert got["detail"] == "bench_detail.json"
  # the side file keeps everything, also as strict JSON
  full = json.loads(detail.read_text(), parse_constant=refuse)
  assert "issue_frac_definition" in full["roofline"] and set(full["regimes"]) == {"baseline", "pilco", "worst"}
  assert full["roofline"]["issue_ceiling_ms"] is None
  # a block that would push the line over the limit is dropped before the contract keys are
  res["next_rows"] = {f"row{i}": {f"k{j}": 1.234567 for j in range(40)} for i in range(20)}
  got2 = json.loads(bench.render_line(res))
  assert "next_rows" not in got2 and "roofline" in got2 and "cpu_baseline" in got2 and len(bench.render_line(res)) <= bench.LINE_LIMIT


def test_every_committed_bench_result_renders_to_a_short_line():
  """The full results kept under profiles/ (one JSON object per file, whatever config) all render within the limit."""
  import glob
  bench = _load("bench_line_t2", os.path.join(ROOT, "bench.py"))
  seen = 0
  for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[45]_bench_*.json"))):
    with open(p) as fh:
      txt = fh.read().strip()
    try:
      res = json.loads(txt if txt.startswith("{") and "\n" not in txt else txt.splitlines()[-1]) if txt else None
    except json.JSONDecodeError:
      res = json.loads(txt)
    if not isinstance(res, dict) or "metric" not in res:
      continue
    line = bench.render_line(res)
    got = json.loads(line)
    assert len(line) < 6000 and "\n" not in line and got["metric"] == res["metric"] and "roofline" in got, p
    seen += 1
  assert seen >= 5
