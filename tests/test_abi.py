"""The C-ABI library loads on a CPU-only host and exports exactly what include/*.h declares;
argument validation happens before any HIP call (so it is testable without a GPU)."""
import ctypes
import os
import re

from gpflowpilco_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gpflowpilco_mm.h")


def declared_functions():
  text = open(HEADER).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(?:int|size_t)\s+(mm_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
  lib = _lib.lib()
  names = declared_functions()
  assert len(names) >= 10
  for name in names:
    assert hasattr(lib, name), f"{name} declared in the header but not exported"
  assert set(_lib.SIGNATURES) == set(names), "ctypes signatures out of sync with the header"
  assert lib.mm_abi_version() == 2


def test_header_constants_match_python():
  text = open(HEADER).read()
  for name in ("MM_F32", "MM_F64", "MM_DMAX", "MM_M_ALIGN", "MM_FULL_OUTPUT_COV", "MM_MODEL_UNCERTAINTY",
               "MM_FORCE_GENERIC", "MM_STAGE_DIAG", "MM_STAGE_OFFDIAG", "MM_STAGE_FINALIZE",
               "MM_FORCE_WORST_TIER", "MM_WORKSPACE_CURRENT", "MM_FORCE_ROUTE", "MM_NO_ROUTE", "MM_SUMS_CURRENT"):
    m = re.search(rf"#define\s+{name}\s+(\d+)", text)
    assert m and int(m.group(1)) == getattr(_lib, name), name


def test_size_queries():
  lib = _lib.lib()
  flags = _lib.MM_FULL_OUTPUT_COV | _lib.MM_MODEL_UNCERTAINTY
  with_c = lib.mm_packed_model_bytes(8, 2000, 8, _lib.MM_F32, 1)
  without = lib.mm_packed_model_bytes(8, 2000, 8, _lib.MM_F32, 0)
  assert with_c - without >= 8 * 2048 * 2048 * 8            # C is stored in f64, padded to 2048
  assert lib.mm_packed_model_bytes(8, 2000, 33, _lib.MM_F32, 1) == 0      # d > MM_DMAX
  ws_full = lib.mm_workspace_bytes(256, 8, 2000, 8, _lib.MM_F32, flags)
  ws_diag = lib.mm_workspace_bytes(256, 8, 2000, 8, _lib.MM_F32, _lib.MM_MODEL_UNCERTAINTY)
  assert 0 < ws_diag < ws_full < 4 << 30
  assert lib.mm_workspace_bytes(0, 8, 2000, 8, _lib.MM_F32, flags) == 0
  # a B-shard is a pointer offset: workspace grows (about) linearly with B
  ws2 = lib.mm_workspace_bytes(512, 8, 2000, 8, _lib.MM_F32, flags)
  assert abs(ws2 / ws_full - 2.0) < 0.05


def test_argument_validation_without_gpu():
  lib = _lib.lib()
  buf = (ctypes.c_char * 64)()
  p = ctypes.addressof(buf)
  assert lib.mm_pack_model(None, 0, 1, 1, 1, 0, p, p, p, p, None, None, None) == -1      # MM_E_ARG
  assert lib.mm_pack_model(p, 64, 1, 16, 40, 0, p, p, p, p, None, None, None) == -2      # MM_E_DIM
  assert lib.mm_pack_model(p, 64, 1, 16, 4, 7, p, p, p, p, None, None, None) == -3       # MM_E_DTYPE
  assert lib.mm_pack_model(p, 64, 1, 16, 4, 0, p, p, p, p, None, None, None) == -4       # buffer too small
  assert lib.mm_moment_match(p, 64, 1, 16, 4, 0, 2, None, p, 3, 0.0, p, p, p, p, 64, None, None) == -1
  assert lib.mm_moment_match(p, 64, 1, 16, 4, 0, 2, p, p, 3, 0.0, p, p, p, p, 64, None, None) == -4
  assert lib.mm_euler_update(2, 40, 0, 1.0, p, p, p, p, p, p, p, None) == -2
  assert lib.mm_rollout_closed(p, 64, 3, 16, 4, 0, 2, 5, 1.0, 3, 0.0, p, p, None, None, p, 64, None, None) == -6
  assert lib.mm_expected_cost(0, 4, 0, p, p, p, p, p, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
  monkeypatch.setattr(_lib, "_lib", None)
  monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
  try:
    _lib.lib()
  except _lib.MomentMatchingLibraryError as e:
    assert "no CPU" in str(e)
  else:
    raise AssertionError("expected MomentMatchingLibraryError")


def test_argument_validation_of_the_gradient_entry_points_without_gpu():
  """The reverse-sweep / tape entry points validate before any HIP call too."""
  lib = _lib.lib()
  buf = (ctypes.c_char * 64)()
  p = ctypes.addressof(buf)
  act = (ctypes.c_int32 * 1)(1)
  F64, F32 = _lib.MM_F64, _lib.MM_F32
  # sizes: cartpole wiring (nx 4, one angle), H = 30
  tape = lib.mm_compose_tape_bytes(1, 30, 4, 1, 100, F64)
  assert tape > 31 * lib.mm_compose_workspace_bytes(1, 4, 1, F64) + 30 * lib.mm_workspace_bytes(1, 4, 100, 6, F64, 3) > 0     # keeps the drift's q stage
  big = lib.mm_compose_tape_bytes(64, 50, 4, 1, 4000, F64)
  assert 0 < big < 51 * lib.mm_compose_workspace_bytes(64, 4, 1, F64) + (1 << 20)      # too large to keep: recomputed in the reverse sweep
  assert lib.mm_compose_tape_bytes(1, 30, 4, 5, 100, F64) == 0                         # more angles than state dims
  assert lib.mm_compose_backward_workspace_bytes(1, 4, 1, 100) > lib.mm_moment_match_backward_bytes(1, 4, 100, 6, 3) > 0
  assert lib.mm_policy_grad_bytes(2, 30, 5) == 2 * (30 * 5 + 30 + 5 + 2) * 8
  # backward of the composed rollout: f64 only, pointers required, shapes must compose, small policy only
  args = lambda dtype=F64, tapep=p, pol_M=30, drift_d=6: (
      p, 64, 4, 100, drift_d, p, 64, pol_M, 5, dtype, 1, 30, 1.0, 4, 1, act, 2.0, -0.5, p, p, tapep, 64, p, p, None, None,
      p, 64, p, 64, None, None)
  assert lib.mm_rollout_composed_backward(*args(dtype=F32)) == -3                 # MM_E_DTYPE
  assert lib.mm_rollout_composed_backward(*args(tapep=None)) == -1                # MM_E_ARG
  assert lib.mm_rollout_composed_backward(*args(drift_d=7)) == -6                 # MM_E_STATE: shapes do not compose
  assert lib.mm_rollout_composed_backward(*args(pol_M=300)) == -2                 # MM_E_DIM: more than 256 policy centres
  assert lib.mm_rollout_composed_backward(*args()) == -4                          # tape too small
  # backward of one match
  mb = lambda dtype=F64, mu=p, d=4: (p, 64, 2, 16, d, dtype, 2, mu, p, 3, p, p, p, p, p, 0, p, 64, p, 64, None, None)
  assert lib.mm_moment_match_backward(*mb(dtype=F32, d=12)) == -3                 # f32 packs: d <= 8 only
  assert lib.mm_moment_match_backward(*mb(dtype=F32)) == -4                       # ... where they get as far as the size check
  assert lib.mm_bwd_f32_supported(8) == 1 and lib.mm_bwd_f32_supported(9) == 0
  # the off-diagonal aggregates alone: f32 packs with d <= 8 only
  pa = lambda dtype=F32, d=4, mu=p: (p, 64, 3, 16, d, dtype, 2, mu, p, 3, p, 64, p, 64, p, 64, None, None)
  assert lib.mm_backward_pair_aggregates(*pa(dtype=F64)) == -3 and lib.mm_backward_pair_aggregates(*pa(d=9)) == -3
  assert lib.mm_backward_pair_aggregates(*pa(mu=None)) == -1 and lib.mm_backward_pair_aggregates(*pa()) == -4
  assert lib.mm_backward_pair_aggregates_bytes(2, 3, 16, 4, 3) > 0 and lib.mm_backward_pair_aggregates_bytes(2, 3, 16, 9, 3) == 0
  assert lib.mm_moment_match_backward(*mb(mu=None)) == -1
  assert lib.mm_moment_match_backward(*mb(d=40)) == -2
  assert lib.mm_moment_match_backward(*mb()) == -4                                 # workspaces too small
  # value + sums in one pass: the same checks as the two calls it stands for; the tape keeps the sums where they fit
  ws = lambda dtype=F64, mu=p, d=4: (p, 64, 2, 16, d, dtype, 2, mu, p, 3, 0.0, p, p, p, p, 64, p, 64, None, None)
  assert lib.mm_moment_match_with_sums(*ws(mu=None)) == -1 and lib.mm_moment_match_with_sums(*ws(d=40)) == -2
  assert lib.mm_moment_match_with_sums(*ws(dtype=F32, d=12)) == -3                 # f32 packs: d <= 8 only
  assert lib.mm_moment_match_with_sums(*ws()) == -4                                # workspaces too small
  assert tape >= (31 * lib.mm_compose_workspace_bytes(1, 4, 1, F64) + 30 * lib.mm_workspace_bytes(1, 4, 100, 6, F64, 3)
                  + 30 * lib.mm_moment_match_backward_bytes_dtype(1, 4, 100, 6, F64, 3))
