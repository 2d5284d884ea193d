// Reverse sweep of the moment-matched policy rollout and of one GP moment match, as device code (gfx950) -- SURVEY.md
// rows f-1 x f-2.  The reference gets these gradients from tf.GradientTape over the whole closure
// (gpflow_pilco/utils/optimizers.py:51-56, loops/pilco.py:192-220); here the arithmetic of csrc/mm_adjoint.h runs one
// workgroup per batch element (or per (element, latent | pair) item), f64 in LDS:
//   k_gp_bwd_items + k_gp_bwd_sum : d(f1, Sff, cross)/d(mu, Sigma) of a frozen model from mm_backward_sums' M-sized sums
//                                   (mm_moment_match_backward: the native form of autodiff.moment_match_backward)
//   k_compose_tail_bwd            : cost + encoding adjoint at x_{h+1}, then the adjoint of forward_sde's bookkeeping + Euler
//   k_policy_head_bwd_small       : NormalCDF head adjoint, then the policy match's adjoint w.r.t. its input moments AND the
//                                   packed policy (Z, beta, Lambda, variance, mean) -- the policy-parameter gradient
//   k_compose_encode_bwd0         : the encoding adjoint at x_0 (gradient w.r.t. the initial state)
// mm_rollout_composed_backward replays the tape of mm_rollout_composed_taped backwards: per step 9 launches (the drift's
// q stage again for its workspace, two sweeps of mm_backward_sums, items, sum, tail, head + policy).
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"
#include "mm_compose.h"
#include "mm_adjoint.h"

// mm_backward.hip / mm_bwd_f32.hip
int mm_backward_sums_impl(const char* pk, const MMModelLayout& ml, const char* ws, const MMWorkspaceLayout& wl, int L, int M, int d,
                          int B, const double* mu, int flags, bool with_unc, bool diag_only, double* out, hipStream_t s);
extern "C" int mm_bwd_f32_supported(int d);
size_t mm_bwd_f32_slab_bytes(int B, int Po, int Mp, int d);
int mm_launch_bwd_offdiag_f32(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl,
                              int B, int L, int M, int d, int flags, const float* mu, double* slab, double* pagg, int32_t* status,
                              hipStream_t stream, int stages, int agg_threads = 512);
int mm_launch_cast_f32_f64(const float* x, double* y, size_t n, hipStream_t stream);

// stage profile (tools/profile_c1_stages.py; -DMM_STAGE_PROFILE builds only): cycles per stage of block 0
#ifdef MM_STAGE_PROFILE
__device__ long long* mmb_stage_prof = nullptr;
extern "C" void mm_stage_profile_set_bwd(void* device_buffer) {
  hipMemcpyToSymbol(HIP_SYMBOL(mmb_stage_prof), &device_buffer, sizeof(void*));
}
#define MMB_PROF_CTX(c_) long long mmb_last_ = clock64(); (c_).prof = mmb_stage_prof; (c_).last = &mmb_last_
#else
#define MMB_PROF_CTX(c_) do {} while (0)
#endif

enum { MMB_MODE_ALL = 0, MMB_MODE_SWEEPS = 1, MMB_MODE_CHAIN = 2 };

// f32 packs, all stages in one call: the aggregate chain of the off-diagonal pairs (re-reduce of the routed items 0.6 ms + full
// moment GEMM 0.5 ms + k_pair_agg 1.4 ms at C3 shape: latency-bound, few or short workgroups) runs on a SIDE STREAM beside the diagonal pairs' f64 sweep,
// which leaves 96 VGPRs per SIMD lane and 112 KB of LDS per CU free: with 256-thread workgroups the chain hides completely
// (value + sums 35.9 -> 34.2 ms; with 512-thread workgroups, which cannot co-reside, 35.5).  (mm_fork.h)
#include "mm_fork.h"


#define MMB_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// GP match backward: items and their sum
// ---------------------------------------------------------------------------------------------------------------------
// grid (L + P, B), 256 threads.  items [B][L + P][d^2 + d]; cbuf [B][L][Mp].
// col [B][col_pairs][3 + d][Mp]: col_pairs = P, or L with pagg [B][Po][mma_pair_agg_len(d)] for the off-diagonal pairs.
__global__ __launch_bounds__(256) void k_gp_bwd_items(int L, int M, int Mp, int d, int P, int with_unc, int full_cov,
                                                      const double* __restrict__ Z, const double* __restrict__ ls2,
                                                      const double* __restrict__ mu, const double* __restrict__ Sigma,
                                                      const double* __restrict__ latmat, const double* __restrict__ w,
                                                      const double* __restrict__ q, const double* __restrict__ col,
                                                      const double* __restrict__ row, int col_pairs,
                                                      const double* __restrict__ pagg, const double* __restrict__ f1raw,
                                                      const double* __restrict__ pre, int nchunk, int nmom,
                                                      const double* __restrict__ g_f1,
                                                      const double* __restrict__ g_Sff, const double* __restrict__ g_cross,
                                                      double* __restrict__ items, double* __restrict__ cbuf, int32_t* status) {
  extern __shared__ double sm[];
  const int item = blockIdx.x, b = blockIdx.y, Po = P - L, nsff = full_cov ? L * L : L;
  bool ok = true;
  double* out = items + ((size_t)b * (L + P) + item) * (d * d + d);
  mma_gp_item_bwd(MMADevCtx(), item, L, M, Mp, d, P, with_unc != 0, Z, ls2, mu + (size_t)b * d, Sigma + (size_t)b * d * d,
                  latmat + (size_t)b * L * (2 * d * d + 2), w + (size_t)b * L * Mp, q + (size_t)b * L * Mp,
                  col + (size_t)b * col_pairs * (3 + d) * Mp, row + (size_t)b * Po * 2 * Mp, g_f1 + (size_t)b * L,
                  g_Sff + (size_t)b * nsff, full_cov, g_cross + (size_t)b * d * L, out, out + d * d,
                  cbuf + ((size_t)b * L + (item < L ? item : 0)) * Mp, sm, &ok,
                  pagg ? pagg + (size_t)b * Po * mma_pair_agg_len(d) : nullptr, f1raw + (size_t)b * L,
                  (pre && item < nmom) ? pre + ((size_t)b * nmom + item) * nchunk * 3 * mma_gp_ncol(d) : nullptr, nchunk);
  if (!ok && threadIdx.x == 0 && status) { atomicMax(status, (int)gridDim.y - b); status[1] = item; }
}

// Partial moment sums of the items over chunks of centres (mma_gp_item_moments): grid (nchunk, nmom, B), 256 threads;
// nmom = L + (pagg ? L : P) items need moments (latents, then pairs -- with aggregates only the diagonal pairs).
// pre [B][nmom][nchunk][3 nc].  For M >= 512 the items kernel's own loops over all centres (one workgroup per item) were
// 0.6 ms at C3 shape; 8 chunks x 16 items x B workgroups do the same sums from LDS-staged tiles.
__global__ __launch_bounds__(256) void k_gp_bwd_moments(int L, int M, int Mp, int d, int P, int with_unc, int full_cov, int chunk,
                                                        const double* __restrict__ Z, const double* __restrict__ mu,
                                                        const double* __restrict__ latmat, const double* __restrict__ w,
                                                        const double* __restrict__ q, const double* __restrict__ col,
                                                        const double* __restrict__ row, int col_pairs, int have_pagg,
                                                        const double* __restrict__ g_f1, const double* __restrict__ g_Sff,
                                                        const double* __restrict__ g_cross, double* __restrict__ pre) {
  extern __shared__ double sm[];
  const int ch = blockIdx.x, item = blockIdx.y, b = blockIdx.z, Po = P - L, nsff = full_cov ? L * L : L, nc = mma_gp_ncol(d);
  const int m0 = ch * chunk, m1 = m0 + chunk < M ? m0 + chunk : M;
  double* out = pre + (((size_t)b * gridDim.y + item) * gridDim.x + ch) * 3 * nc;
  mma_gp_item_moments(MMADevCtx(), item, m0, m1, L, M, Mp, d, P, with_unc != 0, Z, mu + (size_t)b * d,
                      latmat + (size_t)b * L * (2 * d * d + 2), w + (size_t)b * L * Mp, q + (size_t)b * L * Mp,
                      col + (size_t)b * col_pairs * (3 + d) * Mp, row + (size_t)b * Po * 2 * Mp, g_f1 + (size_t)b * L,
                      g_Sff + (size_t)b * nsff, full_cov, g_cross + (size_t)b * d * L, have_pagg != 0, out, sm);
}

// chunk of centres per workgroup of k_gp_bwd_moments (0: the items kernel sums over the centres itself)
// (d <= 8, C3 shape, B = 256: chunks of 512 / 256 / 128 / 64 / 32 centres -> the chain rule of one match takes 2.77 / 1.46 /
// 1.00 / 0.99 / 1.44 ms: short workgroups with four barrier phases each, so more of them in flight wins until the partials dominate)
#ifndef MM_GP_CHUNK8
#define MM_GP_CHUNK8 128
#endif
static inline int mm_gp_moment_chunk(int M, int d) {
  if (M < 512 || d > 16) return 0;
  return d <= 8 ? MM_GP_CHUNK8 : 128;
}

// grid B, 64 threads: gmu [B][d] = sum of the items; gS [B][d][d] (+)= the symmetrised sum.
__global__ __launch_bounds__(64) void k_gp_bwd_sum(int nitems, int d, const double* __restrict__ items,
                                                   double* __restrict__ gmu, double* __restrict__ gS, int accumulate_S) {
  const int b = blockIdx.x, lane = threadIdx.x, st = d * d + d;
  const double* it = items + (size_t)b * nitems * st;
  for (int idx = lane; idx < d * d; idx += 64) {
    const int i = idx / d, j = idx - i * d;
    double s = 0.0;
    for (int t = 0; t < nitems; ++t) s += 0.5 * (it[(size_t)t * st + i * d + j] + it[(size_t)t * st + j * d + i]);
    if (accumulate_S) gS[(size_t)b * d * d + idx] += s; else gS[(size_t)b * d * d + idx] = s;
  }
  for (int k = lane; k < d; k += 64) {
    double s = 0.0;
    for (int t = 0; t < nitems; ++t) s += it[(size_t)t * st + d * d + k];
    gmu[(size_t)b * d + k] = s;
  }
}

// MM_WORKSPACE_CURRENT is a promise of the caller ("the q stage of exactly this state is still on the workspace"): verify the part
// of it the device can see -- the q stage stamped the mean it read into the workspace (MMWorkspaceLayout::mu64) -- and flag a stale
// workspace through the status word ([0] = B - b, [1] = -1) instead of differentiating another state's q stage silently
// (code -1: the q-stage workspace; -2: the kept sums of mm_moment_match_with_sums, stamped with the mean they were swept for)
template <typename T>
__global__ void k_check_workspace_current(const T* __restrict__ mu, const double* __restrict__ mu64, int B, int d, int32_t* status,
                                          int code) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * d) return;
  // a NaN state (the documented result of a non-PD step earlier in a rollout) is not a stale workspace: NaN on both sides
  // matches; and a failure already recorded in status[0] (the non-PD element itself) is never replaced by this code
  const double a = (double)mu[i], bq = mu64[i];
  const bool same = (a == bq) || (a != a && bq != bq);
  if (!same && atomicCAS(status, 0, B - i / d) == 0) status[1] = code;
}
template <typename T>
__global__ void k_stamp_state(const T* __restrict__ mu, double* __restrict__ stamp, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stamp[i] = (double)mu[i];
}

struct MMGpBwdLayout {
  size_t sums, items, cbuf, f1, cross, slab, pagg, mu64, S64, pre, stamp, total;
};
// dtype MM_F64: the sums of all P pairs; MM_F32 (d <= 8): the L diagonal pairs' sums, the off-diagonal pairs' remainder slabs
// and aggregates (mm_bwd_f32.hip), and f64 copies of the f32 state
static inline MMGpBwdLayout mm_gp_bwd_layout(int B, int L, int M, int d, int dtype, int flags) {
  MMGpBwdLayout o;
  const size_t A = 256;
  const int Mp = mm_round_up_int(M, MM_M_ALIGN), P = mm_num_pairs(L, flags), Po = P - L;
  const bool f32 = dtype == MM_F32;
  size_t off = 0;
  o.sums = off;  off = mm_align_up(off + (f32 ? (size_t)B * L * (3 + d) * Mp * 8 : mm_backward_bytes(B, L, M, d, flags)), A);
  o.items = off; off = mm_align_up(off + (size_t)B * (L + P) * (d * d + d) * 8, A);
  o.cbuf = off;  off = mm_align_up(off + (size_t)B * L * Mp * 8, A);
  o.f1 = off;    off = mm_align_up(off + (size_t)B * L * 8, A);
  o.cross = off; off = mm_align_up(off + (size_t)B * d * L * 8, A);
  o.slab = off;  off = mm_align_up(off + (f32 ? mm_bwd_f32_slab_bytes(B, Po, Mp, d) : 0), A);
  o.pagg = off;  off = mm_align_up(off + (f32 ? (size_t)B * Po * mma_pair_agg_len(d) * 8 : 0), A);
  o.mu64 = off;  off = mm_align_up(off + (f32 ? (size_t)B * d * 8 : 0), A);
  o.S64 = off;   off = mm_align_up(off + (f32 ? (size_t)B * d * d * 8 : 0), A);
  const int chunk = mm_gp_moment_chunk(M, d);
  // (items that need moment partials: the latents and the pairs that do not come as aggregates -- f32 packs: the diagonal ones)
  o.pre = off;   off = mm_align_up(off + (chunk ? (size_t)B * (f32 ? 2 * L : L + P) * ((M + chunk - 1) / chunk) * 3 * mma_gp_ncol(d) * 8 : 0), A);
  o.stamp = off; off = mm_align_up(off + (size_t)B * d * 8, A);   // the mean the kept sums were swept for (MM_SUMS_CURRENT is verified)
  o.total = off;
  return o;
}

// enough for either pack type of the model (an f32 pack with d <= 8 runs mm_bwd_f32.hip; anything else the f64 sweeps)
extern "C" size_t mm_moment_match_backward_bytes(int B, int L, int M, int d, int flags) {
  if (B <= 0 || L <= 0 || M <= 0 || d <= 0 || d > MM_DMAX) return 0;
  const size_t n64 = mm_gp_bwd_layout(B, L, M, d, MM_F64, flags).total;
  const size_t n32 = mm_bwd_f32_supported(d) ? mm_gp_bwd_layout(B, L, M, d, MM_F32, flags).total : 0;
  return n64 > n32 ? n64 : n32;
}

// what one pack type needs (0: that pack type has no backward of its own)
extern "C" size_t mm_moment_match_backward_bytes_dtype(int B, int L, int M, int d, int dtype, int flags) {
  if (B <= 0 || L <= 0 || M <= 0 || d <= 0 || d > MM_DMAX) return 0;
  if (dtype == MM_F64) return mm_gp_bwd_layout(B, L, M, d, MM_F64, flags).total;
  if (dtype == MM_F32 && mm_bwd_f32_supported(d)) return mm_gp_bwd_layout(B, L, M, d, MM_F32, flags).total;
  return 0;
}

// (g_f1 [B,L], g_Sff [B,L,L] | [B,L], g_cross [B,d,L]) -> g_mu [B,d], g_Sigma [B,d,d] (symmetric; += if accumulate_Sigma);
// gradients are f64 for either pack type, (mu, Sigma) have the pack's type.
// Re-runs the q stage for (mu, Sigma) on `workspace`, then the M x M sweeps, the items and their sum.
// workspace_is_current: `workspace` already holds the q stage of exactly this (mu, Sigma, flags) -- the reverse sweep of a
// rollout whose tape kept the drift's workspace per step -- so the q stage is not run again
static int mm_moment_match_backward_impl(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                                         const void* mu, const void* Sigma, int flags,
                                         const void* g_f1, const void* g_Sff, const void* g_cross,
                                         void* g_mu, void* g_Sigma, int accumulate_Sigma,
                                         void* workspace, size_t workspace_bytes, void* bwd_ws, size_t bwd_ws_bytes,
                                         int32_t* status, void* stream, bool workspace_is_current, bool skip_sum = false,
                                         int mode = MMB_MODE_ALL, bool verify_sums = true /* check the caller's MM_*_CURRENT promises on the device */) {
  // mode: MMB_MODE_ALL = sweeps + chain rule; MMB_MODE_SWEEPS = everything that does NOT depend on the incoming gradient (the two
  // M x M sweeps, the full moment GEMM, the pair aggregates: mm_moment_match_with_sums leaves them on bwd_ws);
  // MMB_MODE_CHAIN = the chain rule alone on sums a MMB_MODE_SWEEPS call left on bwd_ws (MM_SUMS_CURRENT)
  const bool do_sweeps = mode != MMB_MODE_CHAIN, do_chain = mode != MMB_MODE_SWEEPS;
  if (!packed || !mu || !Sigma || !workspace || !bwd_ws) return MM_E_ARG;
  if (do_chain && (!g_f1 || !g_Sff || !g_cross || !g_mu || !g_Sigma)) return MM_E_ARG;
  if (L <= 0 || M <= 0 || d <= 0 || B <= 0) return MM_E_ARG;
  if (d > MM_DMAX) return MM_E_DIM;
  if (dtype != MM_F64 && dtype != MM_F32) return MM_E_DTYPE;
  const bool f32 = dtype == MM_F32;
  // MM_STAGE_* (measurement, as in mm_Q_reduce_forward): DIAG = the f64 sweep (f64 packs: of every pair), OFFDIAG = the f32
  // remainder sweep, FINALIZE = everything M-sized and smaller (moment GEMM, aggregates, item moments, items, sum)
  int stages = flags & (MM_STAGE_DIAG | MM_STAGE_OFFDIAG | MM_STAGE_FINALIZE);
  if (!stages || mode != MMB_MODE_ALL) stages = MM_STAGE_DIAG | MM_STAGE_OFFDIAG | MM_STAGE_FINALIZE;
  flags &= ~(MM_STAGE_DIAG | MM_STAGE_OFFDIAG | MM_STAGE_FINALIZE);
  if (f32 && !mm_bwd_f32_supported(d)) return MM_E_DTYPE;     // d > 8: differentiate through an f64 pack of the model
  const MMGpBwdLayout bl = mm_gp_bwd_layout(B, L, M, d, dtype, flags);
  if (bwd_ws_bytes < bl.total) return MM_E_WORKSPACE;
  const MMModelLayout ml = mm_model_layout(L, M, d, dtype, 1);
  if (packed_bytes < ml.Cm) return MM_E_WORKSPACE;
  const bool with_unc = (flags & MM_MODEL_UNCERTAINTY) != 0;
  if (with_unc && packed_bytes < ml.total) return MM_E_NO_C;
  const MMWorkspaceLayout wl = mm_workspace_layout(B, L, M, d, dtype, flags);
  if (workspace_bytes < wl.total) return MM_E_WORKSPACE;
  char* bw = (char*)bwd_ws; const char* pk = (const char*)packed; char* ws = (char*)workspace;
  hipStream_t s = (hipStream_t)stream;
  int rc = 0;
  if (!workspace_is_current) {
    rc = mm_q_forward(packed, packed_bytes, L, M, d, dtype, B, mu, Sigma, flags | MM_ISTAGE_NO_M56, bw + bl.f1, bw + bl.cross, nullptr,
                      workspace, workspace_bytes, status, stream);
    if (rc) return rc;
  } else if (status && mode != MMB_MODE_SWEEPS && verify_sums) {   // (SWEEPS: mm_moment_match_with_sums has just run the q stage itself;
                                                                    //  !verify: the composed rollout's own tape holds the workspace)
    const int n = B * d;
    if (f32) hipLaunchKernelGGL((k_check_workspace_current<float>), dim3((n + 255) / 256), dim3(256), 0, s, (const float*)mu,
                                (const double*)(ws + wl.mu64), B, d, status, -1);
    else hipLaunchKernelGGL((k_check_workspace_current<double>), dim3((n + 255) / 256), dim3(256), 0, s, (const double*)mu,
                            (const double*)(ws + wl.mu64), B, d, status, -1);
    MMB_CHECK();
  }
  if (mode == MMB_MODE_CHAIN && status && verify_sums) {   // MM_SUMS_CURRENT is a promise too: the sums carry the mean they belong to
    const int n = B * d;
    if (f32) hipLaunchKernelGGL((k_check_workspace_current<float>), dim3((n + 255) / 256), dim3(256), 0, s, (const float*)mu,
                                (const double*)(bw + bl.stamp), B, d, status, -2);
    else hipLaunchKernelGGL((k_check_workspace_current<double>), dim3((n + 255) / 256), dim3(256), 0, s, (const double*)mu,
                            (const double*)(bw + bl.stamp), B, d, status, -2);
    MMB_CHECK();
  }
  const int P = wl.P, Mp = wl.Mp;
  const double* mu64 = (const double*)mu;
  const double* S64 = (const double*)Sigma;
  const double* pagg = nullptr;
  if (f32) {
    rc = mm_launch_cast_f32_f64((const float*)mu, (double*)(bw + bl.mu64), (size_t)B * d, s);
    if (rc) return rc;
    rc = mm_launch_cast_f32_f64((const float*)Sigma, (double*)(bw + bl.S64), (size_t)B * d * d, s);
    if (rc) return rc;
    mu64 = (const double*)(bw + bl.mu64);
    S64 = (const double*)(bw + bl.S64);
    MMFork* fork = (do_sweeps && wl.Po > 0 && stages == (MM_STAGE_DIAG | MM_STAGE_OFFDIAG | MM_STAGE_FINALIZE)) ? mm_fork_get(s) : nullptr;
    if (fork) {
      // remainder sweep first; then [moment GEMM + pair aggregates on the side stream] beside [the diagonal sweep]
      rc = mm_launch_bwd_offdiag_f32(pk, ml, ws, wl, B, L, M, d, flags, (const float*)mu, (double*)(bw + bl.slab),
                                     (double*)(bw + bl.pagg), status, s, MM_STAGE_OFFDIAG | MM_ISTAGE_NO_ROUTE);
      if (rc) return rc;
      {
        hipError_t ef = hipEventRecord(fork->fork, s);
        if (ef == hipSuccess) ef = hipStreamWaitEvent(fork->s2, fork->fork, 0);
        if (ef != hipSuccess) return (int)ef;
      }
      rc = mm_launch_bwd_offdiag_f32(pk, ml, ws, wl, B, L, M, d, flags, (const float*)mu, (double*)(bw + bl.slab),
                                     (double*)(bw + bl.pagg), status, fork->s2, MM_ISTAGE_ROUTE | MM_STAGE_FINALIZE, 256);
      // (joined whatever happened on the side stream: an error return must not strand it, least of all under capture)
      hipError_t ej = hipEventRecord(fork->join, fork->s2);
      if (!rc && ej == hipSuccess)
        rc = mm_backward_sums_impl(pk, ml, ws, wl, L, M, d, B, mu64, flags, with_unc, true, (double*)(bw + bl.sums), s);
      if (ej == hipSuccess) ej = hipStreamWaitEvent(s, fork->join, 0);
      if (rc) return rc;
      if (ej != hipSuccess) return (int)ej;
      pagg = (const double*)(bw + bl.pagg);
    } else {
    if (do_sweeps && (stages & MM_STAGE_DIAG)) {
      rc = mm_backward_sums_impl(pk, ml, ws, wl, L, M, d, B, mu64, flags, with_unc, true, (double*)(bw + bl.sums), s);
      if (rc) return rc;
    }
    if (wl.Po > 0) {
      if (do_sweeps) {
        rc = mm_launch_bwd_offdiag_f32(pk, ml, ws, wl, B, L, M, d, flags, (const float*)mu, (double*)(bw + bl.slab),
                                       (double*)(bw + bl.pagg), status, s, stages);
        if (rc) return rc;
      }
      pagg = (const double*)(bw + bl.pagg);
    }
    }
  } else if (do_sweeps && (stages & (MM_STAGE_DIAG | MM_STAGE_OFFDIAG))) {
    rc = mm_backward_sums(packed, packed_bytes, L, M, d, dtype, B, mu, flags, workspace, workspace_bytes, bw + bl.sums,
                          mm_backward_bytes(B, L, M, d, flags), stream);
    if (rc) return rc;
  }
  if (mode == MMB_MODE_SWEEPS && verify_sums) {
    const int n = B * d;
    if (f32) hipLaunchKernelGGL((k_stamp_state<float>), dim3((n + 255) / 256), dim3(256), 0, s, (const float*)mu, (double*)(bw + bl.stamp), n);
    else hipLaunchKernelGGL((k_stamp_state<double>), dim3((n + 255) / 256), dim3(256), 0, s, (const double*)mu, (double*)(bw + bl.stamp), n);
    MMB_CHECK();
  }
  if (!do_chain || !(stages & MM_STAGE_FINALIZE)) return 0;
  const int col_pairs = f32 ? L : P;
  const double* col = (const double*)(bw + bl.sums);
  const double* row = col + (size_t)B * P * (3 + d) * Mp;       // (f64 packs; not read with aggregates)
  const size_t shm = (size_t)mma_gp_item_scratch(d, 256) * sizeof(double);
  if (shm > 160 * 1024) return MM_E_DIM;
  if (shm > 64 * 1024) {                                   // d >= 20: more than the default dynamic LDS limit
    hipError_t ea = hipFuncSetAttribute((const void*)k_gp_bwd_items, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (ea != hipSuccess) return (int)ea;
  }
  const int chunk = mm_gp_moment_chunk(M, d), nchunk = chunk ? (M + chunk - 1) / chunk : 0;
  const int nmom = L + (pagg ? L : P);
  const double* pre = nullptr;
  if (chunk) {
    const size_t shm_m = (size_t)mma_gp_moments_scratch(d, 256, chunk) * sizeof(double);
    hipLaunchKernelGGL(k_gp_bwd_moments, dim3(nchunk, nmom, B), dim3(256), shm_m, s, L, M, Mp, d, P, with_unc ? 1 : 0,
                       (flags & MM_FULL_OUTPUT_COV) ? 1 : 0, chunk, (const double*)(pk + ml.Z64), mu64,
                       (const double*)(ws + wl.latmat), (const double*)(ws + wl.w64), (const double*)(ws + wl.q64), col, row,
                       col_pairs, pagg ? 1 : 0, (const double*)g_f1, (const double*)g_Sff, (const double*)g_cross,
                       (double*)(bw + bl.pre));
    MMB_CHECK();
    pre = (const double*)(bw + bl.pre);
  }
  hipLaunchKernelGGL(k_gp_bwd_items, dim3(L + P, B), dim3(256), shm, s, L, M, Mp, d, P, with_unc ? 1 : 0,
                     (flags & MM_FULL_OUTPUT_COV) ? 1 : 0, (const double*)(pk + ml.Z64), (const double*)(pk + ml.ls2),
                     mu64, S64, (const double*)(ws + wl.latmat), (const double*)(ws + wl.w64),
                     (const double*)(ws + wl.q64), col, row, col_pairs, pagg, (const double*)(ws + wl.f1raw),
                     pre, nchunk, nmom, (const double*)g_f1, (const double*)g_Sff, (const double*)g_cross,
                     (double*)(bw + bl.items), (double*)(bw + bl.cbuf), status);
  MMB_CHECK();
  if (skip_sum) return 0;                                   // the consumer sums the items itself (k_policy_head_bwd_small)
  hipLaunchKernelGGL(k_gp_bwd_sum, dim3(B), dim3(64), 0, s, L + P, d, (const double*)(bw + bl.items), (double*)g_mu,
                     (double*)g_Sigma, accumulate_Sigma);
  MMB_CHECK();
  return 0;
}

extern "C" int mm_moment_match_backward(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                                        const void* mu, const void* Sigma, int flags,
                                        const void* g_f1, const void* g_Sff, const void* g_cross,
                                        void* g_mu, void* g_Sigma, int accumulate_Sigma,
                                        void* workspace, size_t workspace_bytes, void* bwd_ws, size_t bwd_ws_bytes,
                                        int32_t* status, void* stream) {
  return mm_moment_match_backward_impl(packed, packed_bytes, L, M, d, dtype, B, mu, Sigma,
                                       flags & ~(MM_WORKSPACE_CURRENT | MM_SUMS_CURRENT), g_f1, g_Sff,
                                       g_cross, g_mu, g_Sigma, accumulate_Sigma, workspace, workspace_bytes, bwd_ws, bwd_ws_bytes,
                                       status, stream, (flags & MM_WORKSPACE_CURRENT) != 0, false,
                                       (flags & MM_SUMS_CURRENT) ? MMB_MODE_CHAIN : MMB_MODE_ALL);
}

// ---------------------------------------------------------------------------------------------------------------------
// Value AND sums in one pass.  Nothing the backward's M x M sweeps compute depends on the incoming gradient: the column sums of
// the diagonal pairs (K, c, cC, U) and the off-diagonal aggregates are functions of the state alone, and they CONTAIN the forward's
// sums:   Sff_aa = sum_j (w_j c_j + q_j cC_j) + var + jitter,   Sff_aa' = pagg[0] - (sum w)(sum w')   (f64 packs: sum_j w'_j c_j).
// A caller that will differentiate therefore runs the q stage and the BACKWARD's sweeps once, reads the value off them, and its
// backward is the chain rule alone (MM_SUMS_CURRENT) -- the forward's own two sweeps are never run.
// ---------------------------------------------------------------------------------------------------------------------
// one wave per (b, pair); col [B][col_pairs][3 + d][Mp]; w, q [B][L][Mp]
template <typename T>
__global__ __launch_bounds__(256) void k_sff_from_sums(const double* __restrict__ col, int col_pairs, const double* __restrict__ pagg,
                                                       int nT, const double* __restrict__ w, const double* __restrict__ q,
                                                       const double* __restrict__ f1raw, const double* __restrict__ var,
                                                       int B, int L, int M, int Mp, int d, int P, int full, int with_unc,
                                                       double jitter, T* __restrict__ Sff) {
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (idx >= B * P) return;
  const int b = idx / P, p = idx - b * P;
  int a = p, a2 = p;
  if (p >= L) { int r = p - L, i = 0; while (r >= L - 1 - i) { r -= L - 1 - i; ++i; } a = i; a2 = i + 1 + r; }
  double s = 0.0;
  if (p < col_pairs) {
    const double* c = col + ((size_t)b * col_pairs + p) * (size_t)(3 + d) * Mp;
    const double* w2 = w + ((size_t)b * L + a2) * Mp;
    const double* q2 = q + ((size_t)b * L + a2) * Mp;
    const bool cterm = (a == a2) && with_unc;
    for (int j = lane; j < M; j += 64) {
      s = fma(w2[j], c[(size_t)Mp + j], s);
      if (cterm) s = fma(q2[j], c[(size_t)2 * Mp + j], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  } else {
    s = pagg[((size_t)b * (P - L) + (p - L)) * nT] - f1raw[(size_t)b * L + a] * f1raw[(size_t)b * L + a2];
  }
  if (lane != 0) return;
  if (a == a2) {
    if (with_unc) s += var[a];
    s += jitter;
    if (full) Sff[((size_t)b * L + a) * L + a] = (T)s; else Sff[(size_t)b * L + a] = (T)s;
  } else {
    Sff[((size_t)b * L + a) * L + a2] = (T)s;
    Sff[((size_t)b * L + a2) * L + a] = (T)s;
  }
}

// stamp: leave the mean on bwd_ws for the backward's check of MM_SUMS_CURRENT (the composed rollout's own tape needs none)
int mm_moment_match_with_sums_impl(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                                   const void* mu, const void* Sigma, int flags, double jitter,
                                   void* f1, void* Sff, void* cross_pre,
                                   void* workspace, size_t workspace_bytes, void* bwd_ws, size_t bwd_ws_bytes,
                                   int32_t* status, void* stream, bool stamp) {
  if (!packed || !mu || !Sigma || !f1 || !Sff || !cross_pre || !workspace || !bwd_ws) return MM_E_ARG;
  if (L <= 0 || M <= 0 || d <= 0 || B <= 0) return MM_E_ARG;
  if (d > MM_DMAX) return MM_E_DIM;
  if (dtype != MM_F64 && dtype != MM_F32) return MM_E_DTYPE;
  if (dtype == MM_F32 && !mm_bwd_f32_supported(d)) return MM_E_DTYPE;
  flags &= ~(MM_STAGE_DIAG | MM_STAGE_OFFDIAG | MM_STAGE_FINALIZE | MM_WORKSPACE_CURRENT | MM_SUMS_CURRENT);
  int rc = mm_q_forward(packed, packed_bytes, L, M, d, dtype, B, mu, Sigma, flags | MM_ISTAGE_NO_M56, f1, cross_pre, nullptr, workspace,
                        workspace_bytes, status, stream);
  if (rc) return rc;
  rc = mm_moment_match_backward_impl(packed, packed_bytes, L, M, d, dtype, B, mu, Sigma, flags, nullptr, nullptr, nullptr, nullptr,
                                     nullptr, 0, workspace, workspace_bytes, bwd_ws, bwd_ws_bytes, status, stream, true, false,
                                     MMB_MODE_SWEEPS, stamp);
  if (rc) return rc;
  const bool f32 = dtype == MM_F32;
  const MMGpBwdLayout bl = mm_gp_bwd_layout(B, L, M, d, dtype, flags);
  const MMModelLayout ml = mm_model_layout(L, M, d, dtype, 1);
  const MMWorkspaceLayout wl = mm_workspace_layout(B, L, M, d, dtype, flags);
  const char* bw = (const char*)bwd_ws; const char* ws = (const char*)workspace; const char* pk = (const char*)packed;
  const int n = B * wl.P, full = (flags & MM_FULL_OUTPUT_COV) ? 1 : 0, with_unc = (flags & MM_MODEL_UNCERTAINTY) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
#define MM_SFF_LAUNCH(T_)                                                                                                        \
  hipLaunchKernelGGL((k_sff_from_sums<T_>), dim3((n + 3) / 4), dim3(256), 0, s, (const double*)(bw + bl.sums), f32 ? L : wl.P,   \
                     (f32 && wl.Po > 0) ? (const double*)(bw + bl.pagg) : (const double*)nullptr, mma_pair_agg_len(d),            \
                     (const double*)(ws + wl.w64), (const double*)(ws + wl.q64), (const double*)(ws + wl.f1raw),                  \
                     (const double*)(pk + ml.var), B, L, M, wl.Mp, d, wl.P, full, with_unc, jitter, (T_*)Sff)
  if (f32) MM_SFF_LAUNCH(float); else MM_SFF_LAUNCH(double);
#undef MM_SFF_LAUNCH
  MMB_CHECK();
  return 0;
}

extern "C" int mm_moment_match_with_sums(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                                         const void* mu, const void* Sigma, int flags, double jitter,
                                         void* f1, void* Sff, void* cross_pre,
                                         void* workspace, size_t workspace_bytes, void* bwd_ws, size_t bwd_ws_bytes,
                                         int32_t* status, void* stream) {
  return mm_moment_match_with_sums_impl(packed, packed_bytes, L, M, d, dtype, B, mu, Sigma, flags, jitter, f1, Sff, cross_pre,
                                        workspace, workspace_bytes, bwd_ws, bwd_ws_bytes, status, stream, true);
}

// ---------------------------------------------------------------------------------------------------------------------
// composed rollout: carry of adjoints between the kernels of the reverse sweep (per batch element, f64)
// ---------------------------------------------------------------------------------------------------------------------
struct MMCarryLayout {
  size_t cm, cS;                  // [B][nx], [B][nx][nx]   adjoint of the state x_h (direct terms; total after the encoding adjoint)
  size_t cme, cSee, cSxe;         // [B][ne], [B][ne][ne], [B][nx][ne]   adjoint of the encoding of x_h
  size_t ccp, cSdd, cmd;          // [B][ne], [B][nd][nd], [B][nd]       adjoint of cpol and of the drift's input moments
  size_t cdf1, cdSff, cdcross;    // [B][nx], [B][nx][nx], [B][nd][nx]   adjoint of the drift's outputs
  size_t total;
};
static inline MMCarryLayout mm_carry_layout(int B, int nx, int na) {
  MMCarryLayout o;
  const size_t A = 256;
  const int ne = nx + na, nd = ne + 1;
  size_t off = 0;
  o.cm = off;      off = mm_align_up(off + (size_t)B * nx * 8, A);
  o.cS = off;      off = mm_align_up(off + (size_t)B * nx * nx * 8, A);
  o.cme = off;     off = mm_align_up(off + (size_t)B * ne * 8, A);
  o.cSee = off;    off = mm_align_up(off + (size_t)B * ne * ne * 8, A);
  o.cSxe = off;    off = mm_align_up(off + (size_t)B * nx * ne * 8, A);
  o.ccp = off;     off = mm_align_up(off + (size_t)B * ne * 8, A);
  o.cSdd = off;    off = mm_align_up(off + (size_t)B * nd * nd * 8, A);
  o.cmd = off;     off = mm_align_up(off + (size_t)B * nd * 8, A);
  o.cdf1 = off;    off = mm_align_up(off + (size_t)B * nx * 8, A);
  o.cdSff = off;   off = mm_align_up(off + (size_t)B * nx * nx * 8, A);
  o.cdcross = off; off = mm_align_up(off + (size_t)B * nd * nx * 8, A);
  o.total = off;
  return o;
}

// k_compose_tail_bwd: grid B, 64 threads.  `first`: the last step of the rollout (no adjoint arrives from a later step).
//   x1 = x_{h+1}; me1 / See1: its encoding (tape slot h + 1); Sxe, cp, Sdd, dcross: tape slot h.
__global__ __launch_bounds__(64) void k_compose_tail_bwd(MMComposeDims D, double dt, int first, const double* __restrict__ x1m,
                                                         const double* __restrict__ x1S, const double* __restrict__ me1,
                                                         const double* __restrict__ See1, const double* __restrict__ target,
                                                         const double* __restrict__ precis, const double* __restrict__ gcost,
                                                         const double* __restrict__ Sxe, const double* __restrict__ cp,
                                                         const double* __restrict__ Sdd, const double* __restrict__ dcross,
                                                         double* cm, double* cS, double* cme, double* cSee, double* cSxe,
                                                         double* ccp, double* cSdd, double* cdf1, double* cdSff, double* cdcross) {
  extern __shared__ double sm[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int nx = D.nx, ne = D.ne, nd = D.nd;
  MMADevCtx c;
  double* gme = sm; double* gSee = gme + ne; double* gSxe = gSee + ne * ne; double* gm1 = gSxe + nx * ne; double* gS1 = gm1 + nx;
  double* wk = gS1 + nx * nx;
  cm += (size_t)b * nx; cS += (size_t)b * nx * nx; cme += (size_t)b * ne; cSee += (size_t)b * ne * ne; cSxe += (size_t)b * nx * ne;
  for (int i = lane; i < ne; i += 64) gme[i] = first ? 0.0 : cme[i];
  for (int i = lane; i < ne * ne; i += 64) gSee[i] = first ? 0.0 : cSee[i];
  for (int i = lane; i < nx * ne; i += 64) gSxe[i] = first ? 0.0 : cSxe[i];
  for (int i = lane; i < nx; i += 64) gm1[i] = first ? 0.0 : cm[i];
  for (int i = lane; i < nx * nx; i += 64) gS1[i] = first ? 0.0 : cS[i];
  __syncthreads();
  // this step's cost statistic: the expected cost of the ENCODED new state (loops/pilco.py:199-205)
  mma_cost_bwd(c, ne, me1 + (size_t)b * ne, See1 + (size_t)b * ne * ne, target, precis, gcost[b], gme, gSee, wk);
  mma_encode_bwd(c, D, x1m + (size_t)b * nx, x1S + (size_t)b * nx * nx, gme, gSee, gSxe, gm1, gS1, wk);
  mma_step_bwd(c, D, dt, Sxe + (size_t)b * nx * ne, cp + (size_t)b * ne, Sdd + (size_t)b * nd * nd, dcross + (size_t)b * nd * nx,
               gm1, gS1, cSxe, ccp + (size_t)b * ne, cSdd + (size_t)b * nd * nd, cdf1 + (size_t)b * nx,
               cdSff + (size_t)b * nx * nx, cdcross + (size_t)b * nd * nx, wk);
  for (int i = lane; i < nx; i += 64) cm[i] = gm1[i];
  for (int i = lane; i < nx * nx; i += 64) cS[i] = gS1[i];
}
static inline size_t mm_tail_bwd_lds(int nx, int ne, int nd) {
  int wk = mma_cost_bwd_scratch(ne);
  const int e = mma_encode_bwd_scratch(nx, ne - nx), st = mma_step_bwd_scratch(nx, nd);
  if (e > wk) wk = e;
  if (st > wk) wk = st;
  return (size_t)(ne + ne * ne + nx * ne + nx + nx * nx + wk + 8) * sizeof(double);
}

// k_policy_head_bwd_small: grid B, 256 threads.  Reads cmd, cSdd, ccp; writes cme, cSee (the adjoint of the encoding of x_h
// through the policy and the joint); accumulates the packed policy's gradient into gpar [B][mm_policy_grad_len].
__global__ __launch_bounds__(256) void k_policy_head_bwd_small(int M, int ne, double scale, double shift,
                                                               const double* __restrict__ Z, const double* __restrict__ beta,
                                                               const double* __restrict__ ls2, const double* __restrict__ var,
                                                               const double* __restrict__ me, const double* __restrict__ See,
                                                               const double* __restrict__ pf1, const double* __restrict__ pSff,
                                                               const double* __restrict__ pcross, const double* __restrict__ cmd,
                                                               const double* __restrict__ cSdd, const double* __restrict__ ccp,
                                                               double* __restrict__ cme, double* __restrict__ cSee,
                                                               double* __restrict__ gpar, int32_t* status,
                                                               const double* __restrict__ items, int nitems) {
  extern __shared__ double sm[];
  const int b = blockIdx.x, nd = ne + 1;
  MMADevCtx c;
  MMB_PROF_CTX(c);
  double* gme = sm; double* gSee = gme + ne; double* gpc = gSee + ne * ne; double* gmu = gpc + ne; double* gSig = gmu + ne;
  double* hw = gSig + ne * ne;             // head scratch: ne + 4
  double* dmd = hw + ne + 4;               // [nd], [nd][nd]: adjoint of the drift's input moments
  double* dSd = dmd + nd;
  double* wk = dSd + nd * nd;
  if (items) {
    // the drift match's (latent | pair) items are summed here (what k_gp_bwd_sum does as a launch of its own): g md = the
    // sum, g Sdd = the bookkeeping's part (cSdd) + the symmetrised sum
    const int st = nd * nd + nd;
    const double* it = items + (size_t)b * nitems * st;
    for (int idx = threadIdx.x; idx < nd * nd; idx += 256) {
      const int i = idx / nd, j = idx - i * nd;
      double sv = 0.0;
      for (int t = 0; t < nitems; ++t) sv += 0.5 * (it[(size_t)t * st + i * nd + j] + it[(size_t)t * st + j * nd + i]);
      dSd[idx] = cSdd[(size_t)b * nd * nd + idx] + sv;
    }
    for (int k = threadIdx.x; k < nd; k += 256) {
      double sv = 0.0;
      for (int t = 0; t < nitems; ++t) sv += it[(size_t)t * st + nd * nd + k];
      dmd[k] = sv;
    }
  } else {
    for (int idx = threadIdx.x; idx < nd * nd; idx += 256) dSd[idx] = cSdd[(size_t)b * nd * nd + idx];
    for (int k = threadIdx.x; k < nd; k += 256) dmd[k] = cmd[(size_t)b * nd + k];
  }
  __syncthreads();
  mma_head_bwd(c, ne, scale, shift, pf1[b], pSff[b], pcross + (size_t)b * ne, See + (size_t)b * ne * ne, dmd, dSd,
               ccp + (size_t)b * ne, gme, gSee, gpc, hw);
  const double gpf1 = hw[ne], gpSff = hw[ne + 1];
  bool ok = true;
  c.stamp(8);
  mma_policy_small_bwd<MMADevCtx, 8>(c, M, ne, Z, beta, ls2, var[0], me + (size_t)b * ne, See + (size_t)b * ne * ne, gpf1, gpSff, gpc, gmu, gSig,
                       gpar + (size_t)b * ((size_t)M * ne + M + ne + 2), wk, &ok);
  c.stamp(9);
  for (int k = threadIdx.x; k < ne; k += 256) cme[(size_t)b * ne + k] = gme[k] + gmu[k];
  for (int idx = threadIdx.x; idx < ne * ne; idx += 256) cSee[(size_t)b * ne * ne + idx] = gSee[idx] + gSig[idx];
  if (!ok && threadIdx.x == 0 && status) { atomicMax(status, (int)gridDim.x - b); status[1] = 0; }
}
static inline size_t mm_policy_bwd_lds(int M, int ne) {
  const int nd = ne + 1;
  return (size_t)(4 * ne + 2 * ne * ne + 4 + nd + nd * nd + mma_policy_small_bwd_scratch(M, ne, 256) + 8) * sizeof(double);
}

// k_compose_encode_bwd0: grid B, 64 threads: the adjoint of the initial state.
__global__ __launch_bounds__(64) void k_compose_encode_bwd0(MMComposeDims D, const double* __restrict__ x0m, const double* __restrict__ x0S,
                                                            const double* __restrict__ cm, const double* __restrict__ cS,
                                                            const double* __restrict__ cme, const double* __restrict__ cSee,
                                                            const double* __restrict__ cSxe, double* __restrict__ g_mx0,
                                                            double* __restrict__ g_Sxx0) {
  extern __shared__ double sm[];
  const int b = blockIdx.x, lane = threadIdx.x, nx = D.nx, ne = D.ne;
  double* gm = sm; double* gS = gm + nx; double* wk = gS + nx * nx;
  for (int i = lane; i < nx; i += 64) gm[i] = cm[(size_t)b * nx + i];
  for (int i = lane; i < nx * nx; i += 64) gS[i] = cS[(size_t)b * nx * nx + i];
  __syncthreads();
  mma_encode_bwd(MMADevCtx(), D, x0m + (size_t)b * nx, x0S + (size_t)b * nx * nx, cme + (size_t)b * ne, cSee + (size_t)b * ne * ne,
                 cSxe + (size_t)b * nx * ne, gm, gS, wk);
  for (int i = lane; i < nx; i += 64) g_mx0[(size_t)b * nx + i] = gm[i];
  for (int idx = lane; idx < nx * nx; idx += 64) {
    const int r = idx / nx, cc = idx - r * nx;
    g_Sxx0[(size_t)b * nx * nx + idx] = 0.5 * (gS[idx] + gS[cc * nx + r]);
  }
}

__global__ void k_zero_f64(double* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.0;
}

struct MMComposeBwdLayout { size_t carry, gp, total; };
static inline MMComposeBwdLayout mm_compose_bwd_layout(int B, int nx, int na, int Md) {
  MMComposeBwdLayout o;
  const int nd = nx + na + 1;
  o.carry = 0;
  o.gp = mm_align_up(mm_carry_layout(B, nx, na).total, 256);
  o.total = o.gp + mm_gp_bwd_layout(B, nx, Md, nd, MM_F64, MM_FULL_OUTPUT_COV | MM_MODEL_UNCERTAINTY).total;
  return o;
}

extern "C" size_t mm_compose_backward_workspace_bytes(int B, int nx, int na, int drift_M) {
  if (B <= 0 || nx <= 0 || nx > MMC_NX || na <= 0 || na > MMC_NA || na > nx || drift_M <= 0) return 0;
  if (2 * na + (nx - na) + 1 > MMC_ND) return 0;
  return mm_compose_bwd_layout(B, nx, na, drift_M).total;
}

extern "C" size_t mm_policy_grad_bytes(int B, int policy_M, int policy_d) {
  if (B <= 0 || policy_M <= 0 || policy_d <= 0) return 0;
  return (size_t)B * mm_policy_grad_len(policy_M, policy_d) * sizeof(double);
}

// Reverse sweep over the tape of mm_rollout_composed_taped (same models, shapes and constants).  f64 only.
//   g_cost   [H][B]: d loss / d cost[h][b] (the loss of pilco.py:199-205 is the plain sum: all ones)
//   g_policy [B][M d + M + d + 2] (out, overwritten): per batch element, gradient w.r.t. the PACKED policy -- Z [M][d],
//            beta [M], ls2 = lengthscales^2 [d], variance, mean_c; the chain to (q_mu, Z, lengthscales, ...) through
//            beta = Kuu^-1 u is the caller's (gpflowpilco_amd/autodiff.py does it with autograd on the 30 x 30 precompute)
//   g_mx0 [B][nx], g_Sxx0 [B][nx][nx] (out, optional): gradient w.r.t. the initial state (symmetric)
// The policy: M <= 256 centres on ne <= 8 encoded dims (k_policy_head_bwd_small: one workgroup per element, the M x M block
// and the M-sized vectors in LDS -- mm_policy_bwd_lds: 120 KB at M = 256, ne = 8); else MM_E_DIM.
extern "C" int mm_rollout_composed_backward(const void* drift_packed, size_t drift_bytes, int drift_L, int drift_M, int drift_d,
                                            const void* policy_packed, size_t policy_bytes, int policy_M, int policy_d,
                                            int dtype, int B, int H, double dt, int nx, int na, const int32_t* active_dims,
                                            double head_scale, double head_shift, const void* target, const void* precis,
                                            const void* tape, size_t tape_bytes, const void* g_cost,
                                            void* g_policy, void* g_mx0, void* g_Sxx0,
                                            void* ws_drift, size_t ws_drift_bytes, void* ws_bwd, size_t ws_bwd_bytes,
                                            int32_t* status, void* stream) {
  if (!drift_packed || !policy_packed || !tape || !g_cost || !g_policy || !ws_drift || !ws_bwd || !target || !precis) return MM_E_ARG;
  if (B <= 0 || H <= 0 || drift_M <= 0 || policy_M <= 0) return MM_E_ARG;
  if (dtype != MM_F64) return MM_E_DTYPE;
  if ((g_mx0 == nullptr) != (g_Sxx0 == nullptr)) return MM_E_ARG;
  MMComposeDims D;
  int rc = mm_compose_dims(nx, na, active_dims, D);
  if (rc) return rc;
  const int ne = D.ne, nd = D.nd;
  if (drift_L != nx || drift_d != nd || policy_d != ne) return MM_E_STATE;
  if (policy_M > 256 || ne > 8) return MM_E_DIM;
  const MMTapeLayout tl = mm_tape_layout(B, H, nx, na, drift_M, dtype);
  if (tape_bytes < tl.total) return MM_E_WORKSPACE;
  const MMComposeBwdLayout bl = mm_compose_bwd_layout(B, nx, na, drift_M);
  if (ws_bwd_bytes < bl.total) return MM_E_WORKSPACE;
  const MMModelLayout pl = mm_model_layout(1, policy_M, ne, dtype, 1);
  if (policy_bytes < pl.Cm) return MM_E_WORKSPACE;
  const MMComposeLayout cl = mm_compose_layout(B, nx, na, dtype);
  const MMCarryLayout kl = mm_carry_layout(B, nx, na);
  hipStream_t s = (hipStream_t)stream;
  char* bw = (char*)ws_bwd; char* cw = bw + bl.carry; char* gw = bw + bl.gp;
  const char* tp = (const char*)tape; const char* pp = (const char*)policy_packed;
  auto cr = [&](size_t off) { return (double*)(cw + off); };
  const double* xm = (const double*)(tp + tl.xm); const double* xS = (const double*)(tp + tl.xS);
  const size_t npar = (size_t)B * mm_policy_grad_len(policy_M, ne);
  hipLaunchKernelGGL(k_zero_f64, dim3((unsigned)((npar + 255) / 256)), dim3(256), 0, s, (double*)g_policy, npar);
  MMB_CHECK();
  const size_t lds_tail = mm_tail_bwd_lds(nx, ne, nd), lds_pol = mm_policy_bwd_lds(policy_M, ne);
  if (lds_pol > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k_policy_head_bwd_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pol);
    if (e != hipSuccess) return (int)e;
  }
  const int dflags = MM_FULL_OUTPUT_COV | MM_MODEL_UNCERTAINTY;
  const MMGpBwdLayout gbl = mm_gp_bwd_layout(B, nx, drift_M, nd, MM_F64, dflags);
  const size_t gp_bytes = gbl.total;
  for (int h = H - 1; h >= 0; --h) {
    const char* sl = tp + (size_t)h * tl.slot_bytes; const char* sn = tp + (size_t)(h + 1) * tl.slot_bytes;
    hipLaunchKernelGGL(k_compose_tail_bwd, dim3(B), dim3(64), lds_tail, s, D, dt, h == H - 1 ? 1 : 0,
                       xm + (size_t)(h + 1) * B * nx, xS + (size_t)(h + 1) * B * nx * nx, (const double*)(sn + cl.me),
                       (const double*)(sn + cl.See), (const double*)target, (const double*)precis,
                       (const double*)g_cost + (size_t)h * B, (const double*)(sl + cl.Sxe), (const double*)(sl + cl.cpol),
                       (const double*)(sl + cl.Sdd), (const double*)(sl + cl.dcross), cr(kl.cm), cr(kl.cS), cr(kl.cme),
                       cr(kl.cSee), cr(kl.cSxe), cr(kl.ccp), cr(kl.cSdd), cr(kl.cdf1), cr(kl.cdSff), cr(kl.cdcross));
    MMB_CHECK();
    // the drift's match: (g df1, g dSff, g dcross) -> g md (assigned), g Sdd (accumulated onto the bookkeeping's part)
    const bool kept = tl.ws_stride != 0;                   // the tape holds this step's q-stage workspace
    const bool sums = tl.gp_stride != 0;                   // ... and the sums of its backward sweeps: chain rule alone
    void* wsd = kept ? (void*)(const_cast<char*>(tp) + tl.ws + (size_t)h * tl.ws_stride) : ws_drift;
    char* gslot = sums ? const_cast<char*>(tp) + tl.gp + (size_t)h * tl.gp_stride : gw;
    rc = mm_moment_match_backward_impl(drift_packed, drift_bytes, nx, drift_M, nd, dtype, B, sl + cl.md, sl + cl.Sdd, dflags,
                                       cr(kl.cdf1), cr(kl.cdSff), cr(kl.cdcross), cr(kl.cmd), cr(kl.cSdd), 1, wsd,
                                       kept ? tl.ws_stride : ws_drift_bytes, gslot, sums ? tl.gp_stride : gp_bytes, status, stream,
                                       kept, true, sums ? MMB_MODE_CHAIN : MMB_MODE_ALL, false /* the tape is this rollout's own */);
    if (rc) return rc;
    hipLaunchKernelGGL(k_policy_head_bwd_small, dim3(B), dim3(256), lds_pol, s, policy_M, ne, head_scale, head_shift,
                       (const double*)(pp + pl.Z64), (const double*)(pp + pl.beta64), (const double*)(pp + pl.ls2),
                       (const double*)(pp + pl.var), (const double*)(sl + cl.me), (const double*)(sl + cl.See),
                       (const double*)(sl + cl.pf1), (const double*)(sl + cl.pSff), (const double*)(sl + cl.pcross),
                       (const double*)cr(kl.cmd), (const double*)cr(kl.cSdd), (const double*)cr(kl.ccp), cr(kl.cme), cr(kl.cSee),
                       (double*)g_policy, status, (const double*)(gslot + gbl.items), nx + nx * (nx + 1) / 2);
    MMB_CHECK();
  }
  if (g_mx0) {
    const size_t lds0 = (size_t)(nx + nx * nx + mma_encode_bwd_scratch(nx, na) + 8) * sizeof(double);
    hipLaunchKernelGGL(k_compose_encode_bwd0, dim3(B), dim3(64), lds0, s, D, xm, xS, (const double*)cr(kl.cm), (const double*)cr(kl.cS),
                       (const double*)cr(kl.cme), (const double*)cr(kl.cSee), (const double*)cr(kl.cSxe), (double*)g_mx0, (double*)g_Sxx0);
    MMB_CHECK();
  }
  return 0;
}
