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
