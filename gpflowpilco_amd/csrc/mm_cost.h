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
template <typename T>
__device__ __forceinline__ void mm_expected_cost_body(int d, const T* mean, const T* cov, const T* target, const T* precis,
                                                      T* cost, int n, int lane, double* smem) {
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
