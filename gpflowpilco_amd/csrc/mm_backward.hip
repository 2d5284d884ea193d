// Backward of the fused reduce w.r.t. the input moments, stage A: the M x M
// sweeps.  The reference differentiates the whole rollout with tf.GradientTape
// (gpflow_pilco/utils/optimizers.py:52-56 through moment_matching/models.py:200-299); here the
// M^2-sized part of that derivative is reduced on the GPU to M-sized vectors, and the remaining
// O(M d^2) algebra is done by torch autograd on a surrogate (gpflowpilco_amd/autodiff.py).
//
// With Omega_ij = (w_i w'_j + [a == a'] C_ij q_i q_j) exp(delta_ij), E_ij = expm1(delta_ij):
//   column sums (thread = column j, rows uniform):
//     Ksum_j = sum_i Omega_ij     csum_j = sum_i w_i E_ij     Usum_j = sum_i Omega_ij zeta_i   [d]
//     cC_j   = sum_i C_ij q_i exp(delta_ij)                    (diagonal pairs only)
//   row sums (thread = row i, columns uniform; off-diagonal pairs only -- diagonal pairs are symmetric):
//     Rsum_i = sum_j Omega_ij     rsum_i = sum_j E_ij w'_j
// f64 mode only (first version; a plain VALU sweep like k_qred_generic: correctness first, the
// MFMA treatment of the forward kernels is the next step).  Must follow mm_q_forward /
// mm_moment_match on the same workspace (it reads w, q and the streamed operands from it).
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"

__device__ __forceinline__ void mmb_decode_pair(int p, int L, int& a, int& a2) {
  if (p < L) { a = p; a2 = p; return; }
  int r = p - L, i = 0;
  while (r >= L - 1 - i) { r -= L - 1 - i; ++i; }
  a = i; a2 = i + 1 + r;
}

// ROWS == false: grid (Mp/256, P, B), thread owns column j.  out_col [B][P][3 + d][Mp]: K, c, cC, U.
// ROWS == true : grid (Mp/256, Po, B), thread owns row i of pair L + blockIdx.y.  out_row [B][Po][2][Mp].
template <int DK, bool ROWS>
__global__ __launch_bounds__(256) void k_bwd_sums(const double* __restrict__ Z64, const double* __restrict__ Zc, int Kz,
                                                  const double* __restrict__ Cm, const double* __restrict__ mu,
                                                  int L, int M, int Mp, int d, int P,
                                                  const double* __restrict__ w, const double* __restrict__ q,
                                                  const double* __restrict__ rowD, const double* __restrict__ colD,
                                                  const double* __restrict__ rowO, const double* __restrict__ colO,
                                                  double* __restrict__ out) {
  const int Po = P - L;
  const int lp = blockIdx.y, b = blockIdx.z;
  const int p = ROWS ? L + lp : lp;
  int a, a2;
  mmb_decode_pair(p, L, a, a2);
  const bool diag = p < L;
  const int t = blockIdx.x * 256 + threadIdx.x;      // column j (or row i)
  if (t >= Mp) return;
  const double* ra = diag ? rowD + ((size_t)b * L + p) * Mp : rowO + ((size_t)b * Po + (p - L)) * Mp;
  const double* cb = diag ? colD + ((size_t)b * L + p) * (size_t)(d + 1) * Mp
                          : colO + ((size_t)b * Po + (p - L)) * (size_t)(d + 1) * Mp;
  const double* wr = w + ((size_t)b * L + a) * Mp;
  const double* wc = w + ((size_t)b * L + a2) * Mp;
  const double* qr = q + ((size_t)b * L + a) * Mp;
  const double* zrow = Zc + (size_t)a * Mp * Kz;
  if (!ROWS) {
    double g[DK], U[DK];
#pragma unroll
    for (int k = 0; k < DK; ++k) { g[k] = (k < d) ? cb[(size_t)k * Mp + t] : 0.0; U[k] = 0.0; }
    const double gam = cb[(size_t)d * Mp + t];
    const bool withC = diag && (Cm != nullptr);
    double Ks = 0.0, cs = 0.0, cC = 0.0, UC[DK];
#pragma unroll
    for (int k = 0; k < DK; ++k) UC[k] = 0.0;
    for (int i = 0; i < M; ++i) {
      double delta = ra[i] + gam;
#pragma unroll
      for (int k = 0; k < DK; ++k) if (k < d) delta += zrow[(size_t)i * Kz + k] * g[k];
      const double E = expm1(delta), e = E + 1.0;
      const double wi = wr[i];
      const double om = wi * e;                       // Omega_ij / w'_j
      Ks += om;
      cs += wi * E;
      double omC = 0.0;
      if (withC) { omC = Cm[((size_t)a * Mp + i) * Mp + t] * qr[i] * e; cC += omC; }
#pragma unroll
      for (int k = 0; k < DK; ++k) {
        if (k < d) {
          const double zi = Z64[((size_t)a * M + i) * d + k] - mu[(size_t)b * d + k];
          U[k] += om * zi;
          UC[k] += omC * zi;
        }
      }
    }
    const double wj = wc[t], qj = withC ? qr[t] : 0.0;
    double* o = out + ((size_t)b * P + p) * (size_t)(3 + d) * Mp;
    o[t] = Ks * wj + cC * qj;
    o[(size_t)Mp + t] = cs;
    o[(size_t)2 * Mp + t] = cC;
#pragma unroll
    for (int k = 0; k < DK; ++k) if (k < d) o[(size_t)(3 + k) * Mp + t] = U[k] * wj + UC[k] * qj;
  } else {
    double zi[DK];
#pragma unroll
    for (int k = 0; k < DK; ++k) zi[k] = (k < d) ? zrow[(size_t)t * Kz + k] : 0.0;
    const double rho = ra[t], wi = wr[t];
    double Rs = 0.0, rs = 0.0;
    for (int j = 0; j < M; ++j) {
      double delta = rho + cb[(size_t)d * Mp + j];
#pragma unroll
      for (int k = 0; k < DK; ++k) if (k < d) delta += zi[k] * cb[(size_t)k * Mp + j];
      const double E = expm1(delta);
      const double wj = wc[j];
      Rs += wj * (E + 1.0);
      rs += wj * E;
    }
    double* o = out + ((size_t)b * Po + lp) * (size_t)2 * Mp;
    o[t] = Rs * wi;
    o[(size_t)Mp + t] = rs;
  }
}

extern "C" size_t mm_backward_bytes(int B, int L, int M, int d, int flags) {
  if (B <= 0 || L <= 0 || M <= 0 || d <= 0 || d > MM_DMAX) return 0;
  const int Mp = mm_round_up_int(M, MM_M_ALIGN), P = mm_num_pairs(L, flags), Po = P - L;
  return ((size_t)B * P * (3 + d) * Mp + (size_t)B * Po * 2 * Mp) * sizeof(double);
}

// out: [B][P][3 + d][Mp] column sums followed by [B][Po][2][Mp] row sums (f64).
extern "C" int mm_backward_sums(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                                const void* mu, int flags, const void* workspace, size_t workspace_bytes,
                                void* out, size_t out_bytes, void* stream) {
  if (!packed || !mu || !workspace || !out) return MM_E_ARG;
  if (L <= 0 || M <= 0 || d <= 0 || B <= 0) return MM_E_ARG;
  if (d > MM_DMAX) return MM_E_DIM;
  if (dtype != MM_F64) return MM_E_DTYPE;              // f64 mode only (see file header)
  const MMModelLayout ml = mm_model_layout(L, M, d, dtype, 1);
  if (packed_bytes < ml.Cm) return MM_E_WORKSPACE;
  const bool has_C = packed_bytes >= ml.total;
  const bool with_unc = (flags & MM_MODEL_UNCERTAINTY) != 0;
  if (with_unc && !has_C) return MM_E_NO_C;
  const MMWorkspaceLayout wl = mm_workspace_layout(B, L, M, d, dtype, flags);
  if (workspace_bytes < wl.total) return MM_E_WORKSPACE;
  if (out_bytes < mm_backward_bytes(B, L, M, d, flags)) return MM_E_WORKSPACE;
  const char* pk = (const char*)packed; const char* ws = (const char*)workspace;
  hipStream_t s = (hipStream_t)stream;
  const double* Cm = with_unc ? (const double*)(pk + ml.Cm) : nullptr;
  double* out_col = (double*)out;
  double* out_row = out_col + (size_t)B * wl.P * (3 + d) * wl.Mp;
#define MMB_ARGS (const double*)(pk + ml.Z64), (const double*)(pk + ml.Zc64), ml.Kz, Cm, (const double*)mu, L, M, wl.Mp, d, wl.P, \
                 (const double*)(ws + wl.w64), (const double*)(ws + wl.q64), (const double*)(ws + wl.rowD),                 \
                 (const double*)(ws + wl.colD), (const double*)(ws + wl.rowO), (const double*)(ws + wl.colO)
#define MMB_LAUNCH(DK_)                                                                                             \
  do {                                                                                                              \
    hipLaunchKernelGGL((k_bwd_sums<DK_, false>), dim3((wl.Mp + 255) / 256, wl.P, B), dim3(256), 0, s, MMB_ARGS, out_col); \
    if (wl.Po > 0)                                                                                                  \
      hipLaunchKernelGGL((k_bwd_sums<DK_, true>), dim3((wl.Mp + 255) / 256, wl.Po, B), dim3(256), 0, s, MMB_ARGS, out_row); \
  } while (0)
  if (d <= 4) MMB_LAUNCH(4);
  else if (d <= 8) MMB_LAUNCH(8);
  else if (d <= 16) MMB_LAUNCH(16);
  else MMB_LAUNCH(32);
#undef MMB_LAUNCH
#undef MMB_ARGS
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
