// Accuracy contract of the f32 pack's off-diagonal pairs (gfx950): estimate, decide, re-reduce in f64.
//
// The reference evaluates <K_Zx K_xZ'> in float64 throughout (gpflow_pilco/utils/kernel_expectation.py:158-165: the
// exp_mahalanobis term of every (i, j)).  The f32 pack takes 1 + b + b^2/2 of e^{b_ij} from f64 moments and reduces the
// remainder  S_rem = sum_ij what_i what'_j r(b_ij)  in f32 (mm_mfma.hip; the backward's aggregates likewise in
// mm_bwd_f32.hip).  Each entry then carries a relative rounding error ~2^-24 (what_i, what'_j and A_i are stored in f32;
// the split product and the polynomial are f32), and the weights what = beta q alternate in sign: where
// sum |what what' r(b)| >> |S| -- ill-conditioned Kuu (beta up to 1e5), wide states -- those roundings are no longer small
// against the block's own scale (measured: 1e-2 of it for state std 0.25 at lengthscales 0.5-1.4, tools/route_study.py).
//
// Nothing tells the caller today; this file does.  The tile kernels carry, per (b, pair, row panel), the running sum
//     E2 = sum over lane blocks of (sum_rows what_i^2) what'_j^2 (max|b|^3 (1 + X + X^2))^2
// -- the variance of S_rem under independent relative roundings, with rho(x) = |r(x)| + |x r'(x)| <= (2/3) |x|^3 e^|x|
// taken at the block's max|b| -- and
//   k_route_decide : est = 2^-24 (2/3) sqrt(E2) per (b, pair); the item is ROUTED when est > MM_ROUTE_TOL x scale_b, scale_b =
//                    max over the batch element's off-diagonal pairs of |s12 - f1 f1'| (the covariance up to the remainder,
//                    already exact in f64 from the q stage); routed items go to a compact list, their count to the
//                    workspace (mm_offdiag_stats) and to status[2] (forward) / status[3] (backward);
//   k_route_f64    : persistent workgroups over (listed item, 256-row panel, column chunk): the same sum from f64 operands -- A_i = G^T zeta_i
//                    re-derived from the pair matrix, the unrounded f64 weights whR / whC of k_pairvec, the f64 centred
//                    inducing inputs -- thread = row, columns streamed through scalar loads (wave-uniform); it OVERWRITES the
//                    f32 kernel's slab entries.  AGG: the backward's 1 + 2d + 3d^2 remainder aggregates instead of the sum.
// On the estimator: measured 7-50 x above the actual f32 error on four regimes (BASELINE recipe, pilco, wide states, the
// random-shape draws of tests/test_gpu_backward_f32.py): nothing is routed on the BASELINE / pilco recipes (20 x margin),
// everything on the ill-conditioned draws whose f32 error was 1e-2; what stays in f32 is within ~4e-5 of its block's scale.
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"
#include "mm_fork.h"
#include "mm_mono.h"
#include "mm_adjoint.h"

__device__ __forceinline__ void mmx_decode_pair_o(int lp, int L, int& a, int& a2) {
  int r = lp, i = 0;
  while (r >= L - 1 - i) { r -= L - 1 - i; ++i; }
  a = i; a2 = i + 1 + r;
}

// est = 2^-24 (2/3) sqrt(E2): E2 from the sweep (rho at every block's max|b|, its e^|x| part, max(1 + X + X^2, e^X), at the lane's max)
__device__ __forceinline__ double mmx_est(double e2, unsigned int, double) {
  return 5.9604644775390625e-8 * (2.0 / 3.0) * sqrt(e2);
}

// grid B, 64 threads.  slot: 1 = forward, 2 = backward (rcount[slot] and status[1 + slot] receive the number of routed items;
// rcount[0] and rcount[slot] were zeroed by the tile kernel of this pass)
__global__ __launch_bounds__(64) void k_route_decide(const float* __restrict__ estO, int npanel, const double* __restrict__ s12,
                                                     const double* __restrict__ f1raw, const unsigned int* __restrict__ amax,
                                                     const double* __restrict__ zmax2, int L, int Po, double tol, int force,
                                                     int* __restrict__ rlist, int* __restrict__ rcount, int* __restrict__ rflag,
                                                     int slot, int32_t* __restrict__ status, const float* __restrict__ estS) {
  const int b = blockIdx.x, lane = threadIdx.x;
  auto zmax2a2 = [&](int po) { int a, a2; mmx_decode_pair_o(po, L, a, a2); return zmax2[a2]; };
  double sc = 0.0;
  for (int po = lane; po < Po; po += 64) {
    int a, a2;
    mmx_decode_pair_o(po, L, a, a2);
    const double v = fabs(s12[(size_t)b * Po + po] - f1raw[(size_t)b * L + a] * f1raw[(size_t)b * L + a2]);
    sc = v > sc ? v : sc;                                   // (a NaN never wins: nothing is routed for a failed batch element)
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(sc, off, 64); sc = o > sc ? o : sc; }
  int total = 0;
  for (int po0 = 0; po0 < Po; po0 += 64) {
    const int po = po0 + lane;
    bool r = false;
    if (po < Po) {
      const float* e = estO + ((size_t)b * Po + po) * npanel;
      double e2 = 0.0;
      for (int pn = 0; pn < npanel; ++pn) e2 += (double)e[pn];
      // + what the skipped tiles of a collapsed item leave out (the forward only: mm_common.h, MM_C6_SYS2)
      if (estS) e2 += (double)estS[(size_t)b * Po + po];
      const double est = mmx_est(e2, amax[(size_t)b * Po + po], zmax2a2(po));
      r = force || est > tol * sc;
      if (rflag) rflag[(size_t)b * Po + po] = r ? 1 : 0;
    }
    const unsigned long long bal = __ballot(r);
    const int n = __popcll(bal);
    if (n) {
      int base = 0;
      if (lane == 0) base = atomicAdd(rcount, n);
      base = __shfl(base, 0, 64);
      if (r) rlist[base + __popcll(bal & ((1ull << lane) - 1ull))] = b * Po + po;
      total += n;
    }
  }
  if (lane == 0 && total) {
    atomicAdd(rcount + slot, total);
    if (status) atomicAdd(status + 1 + slot, total);
  }
}

// Diagnostic: per (b, off-diagonal pair) {est, scale_b} as k_route_decide sees them (after a forward / backward sweep).
__global__ __launch_bounds__(64) void k_route_report(const float* __restrict__ estO, int npanel, const double* __restrict__ s12,
                                                     const double* __restrict__ f1raw, const unsigned int* __restrict__ amax,
                                                     const double* __restrict__ zmax2, int L, int Po, double* __restrict__ out,
                                                     const float* __restrict__ estS) {
  const int b = blockIdx.x, lane = threadIdx.x;
  auto zmax2a2 = [&](int po) { int a, a2; mmx_decode_pair_o(po, L, a, a2); return zmax2[a2]; };
  double sc = 0.0;
  for (int po = lane; po < Po; po += 64) {
    int a, a2;
    mmx_decode_pair_o(po, L, a, a2);
    const double v = fabs(s12[(size_t)b * Po + po] - f1raw[(size_t)b * L + a] * f1raw[(size_t)b * L + a2]);
    sc = v > sc ? v : sc;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(sc, off, 64); sc = o > sc ? o : sc; }
  for (int po = lane; po < Po; po += 64) {
    const float* e = estO + ((size_t)b * Po + po) * npanel;
    double e2 = 0.0;
    for (int pn = 0; pn < npanel; ++pn) e2 += (double)e[pn];
    if (estS) e2 += (double)estS[(size_t)b * Po + po];
    out[((size_t)b * Po + po) * 2 + 0] = mmx_est(e2, amax[(size_t)b * Po + po], zmax2a2(po));
    out[((size_t)b * Po + po) * 2 + 1] = sc;
  }
}

extern "C" int mm_route_estimates(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B, int flags,
                                  const void* workspace, size_t workspace_bytes, double* out, void* stream) {
  if (!packed || !workspace || !out || L <= 0 || M <= 0 || d <= 0 || d > MM_DMAX || B <= 0) return MM_E_ARG;
  if (dtype != MM_F32) return MM_E_DTYPE;
  const MMWorkspaceLayout wl = mm_workspace_layout(B, L, M, d, dtype, flags);
  const MMModelLayout ml = mm_model_layout(L, M, d, dtype, 1);
  if (workspace_bytes < wl.total || packed_bytes < ml.Cm) return MM_E_WORKSPACE;
  if (wl.Po <= 0) return 0;
  const char* ws = (const char*)workspace;
  if (const int rj = mm_fork_join_wait((hipStream_t)stream)) return rj;
  hipLaunchKernelGGL(k_route_report, dim3(B), dim3(64), 0, (hipStream_t)stream, (const float*)(ws + wl.estO),
                     (wl.Mp + MM_PANEL_ROWS - 1) / MM_PANEL_ROWS, (const double*)(ws + wl.s12), (const double*)(ws + wl.f1raw),
                     (const unsigned int*)(ws + wl.amax), (const double*)((const char*)packed + ml.zmax2), L, wl.Po, out,
                     d <= 8 ? (const float*)(ws + wl.estS) : (const float*)nullptr);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// r(x) = e^x - 1 - x - x^2/2 in f64 for any x: Taylor to x^16 on |x| < 1/2 (truncation 0.5^14 / 17! x^3: 2e-19), else
// expm1 minus the two terms (absolute error ~1e-16 e^|x|, against r >= 0.02 there)
__device__ __forceinline__ double mmx_rem(double x) {
  if (fabs(x) < 0.5) {
    double p = 4.7794773323873853e-14;                      // 1/16!
    p = fma(p, x, 7.6471637318198165e-13);                  // 1/15!
    p = fma(p, x, 1.1470745597729725e-11);                  // 1/14!
    p = fma(p, x, 1.6059043836821613e-10);                  // 1/13!
    p = fma(p, x, 2.08767569878681e-09);                    // 1/12!
    p = fma(p, x, 2.505210838544172e-08);                   // 1/11!
    p = fma(p, x, 2.755731922398589e-07);                   // 1/10!
    p = fma(p, x, 2.7557319223985893e-06);                  // 1/9!
    p = fma(p, x, 2.48015873015873e-05);                    // 1/8!
    p = fma(p, x, 1.984126984126984e-04);                   // 1/7!
    p = fma(p, x, 1.388888888888889e-03);                   // 1/6!
    p = fma(p, x, 8.333333333333333e-03);                   // 1/5!
    p = fma(p, x, 4.1666666666666664e-02);                  // 1/4!
    p = fma(p, x, 1.6666666666666666e-01);                  // 1/3!
    return (x * x) * (x * p);
  }
  return expm1(x) - fma(0.5 * x, x, x);
}

__device__ __forceinline__ double mmx_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// grid: persistent (any size), 256 threads.  out: !AGG: partB [B][P][NS] (entry (b, L + lp, panel) is assigned);
// AGG: slab [B][Po][npanel][1 + 2d + 3d^2] (assigned), laid out as k_bwd_rem_f32 writes it (column side centred at zbar_a').
template <int DK, bool AGG>
__global__ __launch_bounds__(256) void k_route_f64(const int* __restrict__ rlist, const int* __restrict__ rcount,
                                                   const double* __restrict__ Zt64, const double* __restrict__ Zc64, int Kz,
                                                   const double* __restrict__ mu64, const double* __restrict__ pairmat,
                                                   const double* __restrict__ whR, const double* __restrict__ whC,
                                                   const unsigned int* __restrict__ amax, const unsigned int* __restrict__ amaxc,
                                                   const double* __restrict__ zmax2,
                                                   int allow_collapse, int L, int M, int Mp, int d, int P, int Po, int npanel,
                                                   int ncc, int NS, double* __restrict__ out) {
  constexpr int NB2 = AGG ? DK * (DK + 1) / 2 : 1;
  constexpr int CS = DK + 2;                                // LDS row stride of a staged column (16-byte aligned rows)
  __shared__ double Gs[DK * DK];
  __shared__ __align__(16) double colbuf[64 * CS];
  __shared__ double red[4];
  __shared__ double Tw[AGG ? 4 * (1 + 2 * DK + 3 * DK * DK) : 1];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  // work unit = (listed item, row panel, column chunk cc of ncc): columns [cc Mc, (cc + 1) Mc) (AGG: ncc == 1)
  const int upi = npanel * ncc, Mc = (M + ncc - 1) / ncc;
  const int nwork = __builtin_amdgcn_readfirstlane(rcount[0]) * upi;
  for (int w = blockIdx.x; w < nwork; w += gridDim.x) {
    const int item = __builtin_amdgcn_readfirstlane(rlist[w / upi]), wu = w % upi, panel = wu / ncc, cc = wu - panel * ncc;
    const int j0 = cc * Mc, j1 = (j0 + Mc < M) ? j0 + Mc : M;
    const int b = item / Po, lp = item - b * Po;
    int a, a2;
    mmx_decode_pair_o(lp, L, a, a2);
    const double* pm = pairmat + ((size_t)b * P + (L + lp)) * (d * d + 1);
    __syncthreads();                                        // (the previous work item's Gs / Tw have been read)
    for (int idx = tid; idx < d * d; idx += 256) Gs[idx] = pm[idx];
    __syncthreads();
    const int row = panel * 256 + tid;
    const bool live = row < M;
    const int rr = live ? row : 0;
    // zeta_i = z_i - mu: the row monomials of the aggregates (AGG); zb_i: the rows as the q stage centred them for the bilinear
    // form -- at the latent's centroid, or at mu where k_pairvec kept them there (mm_mono.h)
    const bool rcen = mm_rows_recentred(amax[(size_t)b * Po + lp]);
    double zeta[DK], A[DK];
#pragma unroll
    for (int k = 0; k < DK; ++k)
      zeta[k] = (k < d && live) ? Zt64[((size_t)a * d + (k < d ? k : 0)) * Mp + rr] - mu64[(size_t)b * d + (k < d ? k : 0)] : 0.0;
    {
      double zb[DK];
#pragma unroll
      for (int k = 0; k < DK; ++k) zb[k] = (k < d && live && rcen) ? Zc64[((size_t)a * Mp + rr) * Kz + (k < d ? k : 0)] : zeta[k];
#pragma unroll
      for (int k = 0; k < DK; ++k) {
        double s = 0.0;
        if (k < d) {
#pragma unroll
          for (int l = 0; l < DK; ++l) if (l < d) s = fma(Gs[l * d + k], zb[l], s);      // A_i = G^T zb_i
        }
        A[k] = s;
      }
    }
    const double wr = live ? whR[((size_t)b * Po + lp) * Mp + rr] : 0.0;
    // the forward's collapsed items carry p6 in their moments (k_spoly, k_spoly56): the same predicate, the same coefficients
    // (an item with at least one collapsed row group has the cubic term of EVERY row in its f64 moments: mm_mono.h, k_spoly)
    bool coll = false;
    if (!AGG && allow_collapse && zmax2 != nullptr && d <= 8) coll = mm_item_collapsed(amaxc[(size_t)b * Po + lp]);
    // (order 3 of p6 stays with the f64 moments; orders 4, 5, 6 -- s56, from f32 moments -- are dropped for a routed item by
    // k_finalize, so the re-reduce keeps them: r - C0 x^3)
    const double sub0 = coll ? (double)MM_C6_C0 : 0.0, sub1 = 0.0;
    const double* zc = Zc64 + (size_t)a2 * Mp * Kz;         // wave-uniform from here on: scalar loads
    const double* wc = whC + ((size_t)b * Po + lp) * Mp;
    double B0 = 0.0, B1[AGG ? DK : 1], B2[NB2];
#pragma unroll
    for (int k = 0; k < (AGG ? DK : 1); ++k) B1[k] = 0.0;
#pragma unroll
    for (int k = 0; k < NB2; ++k) B2[k] = 0.0;
    // columns in chunks of 64 through LDS: coalesced loads of (zc'_j [d], what'_j), then every lane reads the same address per
    // column (broadcast).  (Wave-uniform scalar loads straight from memory were tried first: one s_load per column, a fresh
    // cache line every time -- 0.5 us per column, 130 us for a 250-column unit against 17 us of arithmetic.)
    for (int jc = j0; jc < j1; jc += 64) {
    __syncthreads();                                        // (the previous chunk has been consumed)
    for (int idx = tid; idx < 64 * (DK + 1); idx += 256) {
      const int c = idx / (DK + 1), k = idx - c * (DK + 1), jj = jc + c;
      double v = 0.0;
      if (jj < j1) v = k == DK ? wc[jj] : (k < d ? zc[(size_t)jj * Kz + k] : 0.0);
      colbuf[c * CS + k] = v;                               // columns past the chunk's end: zero weight, zero inputs
    }
    __syncthreads();
#pragma unroll 2
    for (int c = 0; c < 64; ++c) {
      double z[DK];
#pragma unroll
      for (int k = 0; k < DK; ++k) z[k] = colbuf[c * CS + k];
      const double wj = colbuf[c * CS + DK];
      double x = 0.0;
#pragma unroll
      for (int k = 0; k < DK; ++k) x = fma(A[k], z[k], x);
      x = fmin(x, MM_EXP_CAP_F64);                          // (mm_common.h: exponent caps)
      double r = mmx_rem(x);
      if (!AGG) r -= (x * x) * x * fma(sub1, x, sub0);
      const double v = wj * r;
      B0 += v;
      if constexpr (AGG) {
        int t = 0;
#pragma unroll
        for (int l = 0; l < DK; ++l) {
          const double u = v * z[l];
          B1[l] += u;
#pragma unroll
          for (int l2 = l; l2 < DK; ++l2, ++t) B2[t] = fma(u, z[l2], B2[t]);
        }
      }
    }
    }
    if constexpr (!AGG) {
      const double s = mmx_wave_sum(wr * B0);
      if (lane == 0) red[wv] = s;
      __syncthreads();
      if (tid == 0) out[((size_t)b * P + (L + lp)) * NS + wu] = (red[0] + red[1]) + (red[2] + red[3]);
    } else {
      const int nT = mma_pair_agg_len(d);
      const int oR1 = 1, oR2 = 1 + d, oK1 = 1 + d + d * d, oK2 = 1 + 2 * d + d * d, oXC = 1 + 2 * d + 2 * d * d;
      double* tw = Tw + wv * nT;
      auto emit = [&](int idx, double val) {
        const double s = mmx_wave_sum(val);
        if (lane == 0) tw[idx] = s;
      };
      const double w0 = wr * B0;
      emit(0, w0);
#pragma unroll
      for (int k = 0; k < DK; ++k) {
        if (k >= d) break;
        const double wk = w0 * zeta[k];
        emit(oR1 + k, wk);
#pragma unroll
        for (int k2 = 0; k2 < DK; ++k2) if (k2 < d) emit(oR2 + k * d + k2, wk * zeta[k2]);
        emit(oK1 + k, wr * B1[k]);
#pragma unroll
        for (int l = 0; l < DK; ++l) if (l < d) emit(oXC + k * d + l, (wr * zeta[k]) * B1[l]);
      }
      {
        int t = 0;
#pragma unroll
        for (int l = 0; l < DK; ++l)
#pragma unroll
          for (int l2 = l; l2 < DK; ++l2, ++t) {
            if (l2 < d) {                                   // (l <= l2 < d)
              const double s = mmx_wave_sum(wr * B2[t]);
              if (lane == 0) { tw[oK2 + l * d + l2] = s; tw[oK2 + l2 * d + l] = s; }
            }
          }
      }
      __syncthreads();
      double* o = out + (((size_t)b * Po + lp) * npanel + panel) * nT;
      for (int idx = tid; idx < nT; idx += 256) o[idx] = (Tw[idx] + Tw[nT + idx]) + (Tw[2 * nT + idx] + Tw[3 * nT + idx]);
    }
  }
}

// Decide + re-reduce after the f32 tile kernel of this pass (forward: mm_mfma.hip, agg = 0, out = the workspace's partB;
// backward: mm_bwd_f32.hip, agg = 1, out = its remainder slab).  The tile kernel has zeroed rcount[0] and rcount[slot].
int mm_launch_route(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl, int B, int L, int M,
                    int d, int flags, int agg, double* out, int32_t* status, hipStream_t stream) {
  if (wl.Po <= 0 || (flags & MM_NO_ROUTE)) return 0;
  if (const int rj = mm_fork_join_wait(stream)) return rj;   // s12 (the decision's scale) may still be on the q stage's side stream
  const int npanel = (wl.Mp + MM_PANEL_ROWS - 1) / MM_PANEL_ROWS;
  int* rlist = (int*)(ws + wl.rlist);
  int* rcount = (int*)(ws + wl.rcount);
  hipLaunchKernelGGL(k_route_decide, dim3(B), dim3(64), 0, stream, (const float*)(ws + wl.estO), npanel,
                     (const double*)(ws + wl.s12), (const double*)(ws + wl.f1raw), (const unsigned int*)(ws + wl.amax),
                     (const double*)(packed + ml.zmax2), L, wl.Po, (double)MM_ROUTE_TOL,
                     (flags & MM_FORCE_ROUTE) ? 1 : 0, rlist, rcount, agg ? (int*)nullptr : (int*)(ws + wl.rflag), agg ? 2 : 1, status,
                     (!agg && d <= 8) ? (const float*)(ws + wl.estS) : (const float*)nullptr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  const int ncc = agg ? 1 : mm_route_ncc(wl.NS, npanel);
  long long nw = (long long)B * wl.Po * npanel * ncc;
  const int grid = (int)(nw < 2048 ? nw : 2048);
  const double* zmax2 = mm_moment_deg(d) >= 4 ? (const double*)(packed + ml.zmax2) : nullptr;
  const int allow = (flags & MM_FORCE_WORST_TIER) ? 0 : 1;
#define MMX_ARGS                                                                                                             \
  (const int*)rlist, (const int*)rcount, (const double*)(packed + ml.Zt64), (const double*)(packed + ml.Zc64), ml.Kz,       \
  (const double*)(ws + wl.mu64), (const double*)(ws + wl.pairmat), (const double*)(ws + wl.whR), (const double*)(ws + wl.whC), \
  (const unsigned int*)(ws + wl.amax), (const unsigned int*)(ws + wl.amaxc), zmax2, allow, L, M, wl.Mp, d, wl.P, wl.Po, npanel, ncc, wl.NS, out
  if (agg) {
    if (d > 8) return MM_E_DIM;
    if (d <= 4) hipLaunchKernelGGL((k_route_f64<4, true>), dim3(grid), dim3(256), 0, stream, MMX_ARGS);
    else hipLaunchKernelGGL((k_route_f64<8, true>), dim3(grid), dim3(256), 0, stream, MMX_ARGS);
  } else {
    if (d <= 4) hipLaunchKernelGGL((k_route_f64<4, false>), dim3(grid), dim3(256), 0, stream, MMX_ARGS);
    else if (d <= 8) hipLaunchKernelGGL((k_route_f64<8, false>), dim3(grid), dim3(256), 0, stream, MMX_ARGS);
    else if (d <= 16) hipLaunchKernelGGL((k_route_f64<16, false>), dim3(grid), dim3(256), 0, stream, MMX_ARGS);
    else hipLaunchKernelGGL((k_route_f64<32, false>), dim3(grid), dim3(256), 0, stream, MMX_ARGS);
  }
#undef MMX_ARGS
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
