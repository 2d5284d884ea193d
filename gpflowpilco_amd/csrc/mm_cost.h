// Closed-form expected saturating cost of one Gaussian (GaussianObjective, gpflow_pilco/components.py:26-37), as a
// device function of ONE 64-lane wave: shared by k_expected_cost (mm_kernels.hip) and the fused end-of-step kernel
// of the composed rollout (mm_compose.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

// dynamic LDS the body needs: [d][d+2] + [d][d] + [d] doubles
static inline size_t mm_cost_lds_bytes(int d) { return (size_t)(d * (d + 2) + d * d + d) * sizeof(double); }

//   cost = -det(I + S W)^-1/2 exp(-0.5 err^T W (I + S W)^-1 err): Gaussian elimination with partial pivoting on
//   [I + S W | err] in LDS (f64).  n: the element; smem: mm_cost_lds_bytes(d) of LDS.
// d <= 8: the elimination in registers, one matrix entry per lane (lane = 8 i + j), partial pivoting kept, broadcasts by
// wave shuffles: no LDS, no barrier.  The LDS version below took 6.6 us of the 14.3 us k_compose_tail at cartpole sizes
// (tools/profile_c1_stages.py): a barrier + a serial pivot search per column and a serial back substitution.
template <typename T>
__device__ __forceinline__ void mm_expected_cost_body8(int d, const T* mean, const T* cov, const T* target, const T* precis,
                                                       T* cost, int n, int lane) {
  const int i = lane >> 3, j = lane & 7;
  const bool in = i < d && j < d;
  const double w = in ? (double)precis[i * d + j] : 0.0;
  const double s = in ? (double)cov[((size_t)n * d + i) * d + j] : 0.0;
  const double ei = i < d ? (double)mean[(size_t)n * d + i] - (double)target[i] : 0.0;
  double a = (i == j) ? 1.0 : 0.0;                          // I + S W
#pragma unroll
  for (int k = 0; k < 8; ++k) a = fma(__shfl(s, 8 * i + k, 64), __shfl(w, 8 * k + j, 64), a);
  double b = ei, det = 1.0;                                 // right-hand side of row i, replicated along the row
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < d) {
      double best = fabs(__shfl(a, 9 * k, 64));
      int p = k;
#pragma unroll
      for (int r = 1; r < 8; ++r) {
        if (k + r < d) {
          const double v = fabs(__shfl(a, 8 * (k + r) + k, 64));
          if (v > best) { best = v; p = k + r; }
        }
      }
      if (p != k) {                                         // (wave-uniform)
        const int si = i == k ? p : (i == p ? k : i);
        a = __shfl(a, 8 * si + j, 64); b = __shfl(b, 8 * si + j, 64);
        det = -det;
      }
      const double piv = __shfl(a, 9 * k, 64), ip = 1.0 / piv;
      det *= piv;
      const double rk = __shfl(a, 8 * k + j, 64), bk = __shfl(b, 8 * k, 64), ci = __shfl(a, 8 * i + k, 64);
      if (i == k) { a = rk * ip; b = bk * ip; }
      else { a = fma(-ci * ip, rk, a); b = fma(-ci * ip, bk, b); }
    }
  }
  // b = (I + S W)^-1 err in row i;  dist2 = err^T W b
  double t = in ? ei * w * __shfl(b, 8 * j, 64) : 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  if (lane == 0) cost[n] = (T)(-rsqrt(det) * exp(-0.5 * t));
}

template <typename T>
__device__ __forceinline__ void mm_expected_cost_body(int d, const T* mean, const T* cov, const T* target, const T* precis,
                                                      T* cost, int n, int lane, double* smem) {
  if (d <= 8) { mm_expected_cost_body8<T>(d, mean, cov, target, precis, cost, n, lane); return; }
  const int dp = d + 2;
  double* A = smem;               // [d][d+2]: I + S W | err | (pad)
  double* W = A + d * dp;         // [d][d]
  double* e0 = W + d * d;         // [d] err
  __shared__ int piv;
  __shared__ double detv;
  for (int idx = lane; idx < d * d; idx += 64) W[idx] = (double)precis[idx];
  if (lane < d) e0[lane] = (double)mean[(size_t)n * d + lane] - (double)target[lane];
  if (lane == 0) detv = 1.0;
  __syncthreads();
  for (int idx = lane; idx < d * d; idx += 64) {
    const int i = idx / d, j = idx - i * d;
    double s = (i == j) ? 1.0 : 0.0;
    for (int k = 0; k < d; ++k) s += (double)cov[((size_t)n * d + i) * d + k] * W[k * d + j];
    A[i * dp + j] = s;
  }
  if (lane < d) A[lane * dp + d] = e0[lane];
  __syncthreads();
  for (int k = 0; k < d; ++k) {
    if (lane == 0) {
      int p = k; double best = fabs(A[k * dp + k]);
      for (int i = k + 1; i < d; ++i) { const double v = fabs(A[i * dp + k]); if (v > best) { best = v; p = i; } }
      piv = p;
    }
    __syncthreads();
    const int p = piv;
    if (p != k) {
      for (int j = lane; j <= d; j += 64) { const double t = A[k * dp + j]; A[k * dp + j] = A[p * dp + j]; A[p * dp + j] = t; }
      if (lane == 0) detv = -detv;
    }
    __syncthreads();
    const double akk = A[k * dp + k];
    if (lane == 0) detv *= akk;
    // eliminate rows below k: entries (i, j), i > k, j > k (incl. the rhs column)
    const int nr = d - 1 - k, nc = d - k;       // columns k+1 .. d
    __syncthreads();
    for (int idx = lane; idx < nr * nc; idx += 64) {
      const int i = k + 1 + idx / nc, j = k + 1 + idx % nc;
      A[i * dp + j] -= (A[i * dp + k] / akk) * A[k * dp + j];
    }
    __syncthreads();
  }
  // back substitution (serial, d <= 32): y = (I + S W)^-1 err
  if (lane == 0) {
    for (int i = d - 1; i >= 0; --i) {
      double s = A[i * dp + d];
      for (int j = i + 1; j < d; ++j) s -= A[i * dp + j] * A[j * dp + d];
      A[i * dp + d] = s / A[i * dp + i];
    }
    double dist2 = 0.0;
    for (int i = 0; i < d; ++i) {
      double s = 0.0;
      for (int j = 0; j < d; ++j) s += W[i * d + j] * A[j * dp + d];
      dist2 += e0[i] * s;
    }
    cost[n] = (T)(-rsqrt(detv) * exp(-0.5 * dist2));
  }
}
