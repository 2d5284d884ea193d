"""ctypes binding of ``libgpflowpilco_mm.so`` (include/gpflowpilco_mm.h).

There is NO CPU fallback: if the HIP library is missing the import of any op
raises, loudly.  Build it with ``gpflowpilco_amd/csrc/build.sh`` (or
``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPFLOWPILCO_MM_LIB: an alternative build of the same library (A/B experiments of kernel variants: csrc/build.sh
# with OUT=... and extra -D flags); the default is the in-tree build
LIB_PATH = os.environ.get("GPFLOWPILCO_MM_LIB") or os.path.join(_HERE, "libgpflowpilco_mm.so")

MM_F32, MM_F64 = 0, 1
MM_DMAX = 32
MM_M_ALIGN = 128
MM_FULL_OUTPUT_COV = 1
MM_MODEL_UNCERTAINTY = 2
MM_FORCE_GENERIC = 4
MM_STAGE_DIAG = 8
MM_STAGE_OFFDIAG = 16
MM_STAGE_FINALIZE = 32
MM_FORCE_WORST_TIER = 64
MM_WORKSPACE_CURRENT = 128
MM_FORCE_ROUTE = 256
MM_NO_ROUTE = 512
MM_SUMS_CURRENT = 1024

ERRORS = {
    -1: "MM_E_ARG: NULL pointer or non-positive size",
    -2: "MM_E_DIM: input dimension d exceeds MM_DMAX=32 or unsupported shape",
    -3: "MM_E_DTYPE: dtype must be MM_F32 or MM_F64",
    -4: "MM_E_WORKSPACE: workspace/packed buffer too small",
    -5: "MM_E_NO_C: model_uncertainty requested but the model was packed without C",
    -6: "MM_E_STATE: Euler update / closed rollout needs d == L and a full output covariance",
}

# name -> (restype, argtypes); exactly the symbols include/gpflowpilco_mm.h declares
SIGNATURES = {
    "mm_abi_version": (C.c_int, []),
    "mm_packed_model_bytes": (C.c_size_t, [C.c_int] * 5),
    "mm_pack_model": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]
                      + [C.c_void_p] * 6 + [C.c_void_p]),
    "mm_pack_perm": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mm_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "mm_moment_match": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                  C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "mm_q_forward": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "mm_Q_reduce_forward": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_double, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "mm_euler_update": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double] + [C.c_void_p] * 8),
    "mm_expected_cost": (C.c_int, [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6),
    "mm_offdiag_stats": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "mm_route_estimates": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "mm_compose_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "mm_rollout_composed": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                      C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_void_p]),
    "mm_compose_tape_bytes": (C.c_size_t, [C.c_int] * 6),
    "mm_rollout_composed_taped": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                            C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_void_p]),
    "mm_compose_backward_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "mm_policy_grad_bytes": (C.c_size_t, [C.c_int] * 3),
    "mm_rollout_composed_backward": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                               C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                               C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_size_t, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                               C.c_void_p, C.c_void_p]),
    "mm_bwd_f32_supported": (C.c_int, [C.c_int]),
    "mm_backward_pair_aggregates_bytes": (C.c_size_t, [C.c_int] * 5),
    "mm_backward_pair_aggregates": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                              C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "mm_moment_match_backward_bytes": (C.c_size_t, [C.c_int] * 5),
    "mm_moment_match_backward_bytes_dtype": (C.c_size_t, [C.c_int] * 6),
    "mm_moment_match_backward": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_void_p]),
    "mm_moment_match_with_sums": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                            C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_void_p]),
    "mm_backward_bytes": (C.c_size_t, [C.c_int] * 5),
    "mm_backward_sums": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mm_pathwise_eval": (C.c_int, [C.c_int] * 6 + [C.c_void_p] * 12),
    "mm_pathwise_rollout": (C.c_int, [C.c_int] * 7 + [C.c_double] + [C.c_void_p] * 13),
    "mm_pathwise_eval_jac": (C.c_int, [C.c_int] * 6 + [C.c_void_p] * 13),
    "mm_pathwise_eval_bound": (C.c_int, [C.c_int] * 6 + [C.c_void_p] * 13),
    "mm_pathwise_tape_bytes": (C.c_size_t, [C.c_int] * 6),
    "mm_pathwise_policy_rollout": (C.c_int, [C.c_int] * 5 + [C.c_double, C.c_int, C.c_int] + [C.c_void_p] * 10
                                   + [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_double]
                                   + [C.c_void_p] * 4 + [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "mm_pathwise_backward_scratch_bytes": (C.c_size_t, [C.c_int] * 3),
    "mm_pathwise_policy_rollout_backward": (C.c_int, [C.c_int] * 3 + [C.c_double, C.c_int, C.c_int, C.c_void_p]
                                            + [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_double]
                                            + [C.c_void_p] * 2 + [C.c_void_p, C.c_size_t] + [C.c_void_p] * 3
                                            + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "mm_rollout_closed": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_double, C.c_int, C.c_double,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
}

_lib = None


class MomentMatchingLibraryError(RuntimeError):
  pass


def lib():
  """Load the shared library once; raise if it is not built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise MomentMatchingLibraryError(
          f"{LIB_PATH} is missing: the HIP extension is not built and there is no CPU "
          "fallback. Run gpflowpilco_amd/csrc/build.sh (needs hipcc).")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
      fn = getattr(handle, name)
      fn.restype = res
      fn.argtypes = args
    _lib = handle
  return _lib


def check(rc: int, what: str):
  if rc == 0:
    return
  if rc < 0:
    raise ValueError(f"{what}: {ERRORS.get(rc, rc)}")
  raise MomentMatchingLibraryError(f"{what}: HIP error {rc}")
