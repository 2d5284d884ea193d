// Device helpers shared by the translation units of the library (no relocatable device code: header only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

// In-place inverse of a symmetric positive definite d x d matrix held in LDS (row stride dp),
// by one 64-lane workgroup.  A: in SPD (full storage) -> out inverse (full).  Y: scratch.
// Returns log det(A) (same value in every lane); *ok is cleared if a pivot is not positive.
// Several waves of a workgroup may call it at once, each on its own (A, Y): the barriers are workgroup barriers, and
// every wave executes the same number of them.
// d <= 8: the whole inversion in registers, one matrix entry per lane (lane = 8 i + j, identity outside d x d):
// Gauss-Jordan without pivoting (the matrix is SPD: every pivot is a Schur complement > 0), row / column / pivot
// broadcasts by wave shuffles -- no LDS round trip and no barrier per elimination step.  At cartpole sizes the
// barrier-per-step version below was 5.7 us of k_policy_match_small and most of k_prep (tools/profile_c1_stages.py).
// Same contract as mm_spd_inverse (one workgroup barrier on entry, one on exit, executed by every wave).
__device__ __forceinline__ double mm_spd_inverse_d8(double* A, int d, int dp, bool* ok) {
  const int lane = threadIdx.x & 63, i = lane >> 3, j = lane & 7;
  __syncthreads();
  const bool in = i < d && j < d;
  double a = in ? A[i * dp + j] : (i == j ? 1.0 : 0.0);
  double pk = 1.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < d) {
      const double p = __shfl(a, 9 * k, 64);
      if (!(p > 0.0)) *ok = false;
      if (lane == k) pk = p;
      const double ip = 1.0 / p;
      const double rk = __shfl(a, 8 * k + j, 64);           // a(k, j)
      const double ci = __shfl(a, 8 * i + k, 64);           // a(i, k)
      if (i == k) a = (j == k) ? ip : rk * ip;
      else a = (j == k) ? -ci * ip : fma(-ci * ip, rk, a);
    }
  }
  a = 0.5 * (a + __shfl(a, 8 * j + i, 64));                 // exactly symmetric (callers keep upper triangles only)
  double logdet = lane < d ? log(pk) : 0.0;                 // one logarithm per lane
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) logdet += __shfl_xor(logdet, off, 64);
  if (in) A[i * dp + j] = a;
  __syncthreads();
  return logdet;
}

__device__ __forceinline__ double mm_spd_inverse(double* A, double* Y, int d, int dp, bool* ok) {
  if (d <= 8) return mm_spd_inverse_d8(A, d, dp, ok);
  const int lane = threadIdx.x & 63;
  // d <= 8: lane owns entry (li, lj) for the whole factorisation (one integer division instead of one per step)
  const bool one = d * d <= 64;
  const int li = lane / d, lj = lane - li * d;
  for (int k = 0; k < d; ++k) {
    __syncthreads();
    const double akk = A[k * dp + k];
    if (!(akk > 0.0)) *ok = false;
    const double lkk = sqrt(akk);
    __syncthreads();
    if (lane > k && lane < d) A[lane * dp + k] /= lkk;
    if (lane == k) A[k * dp + k] = lkk;
    __syncthreads();
    if (one) {
      if (li < d && lj > k && li >= lj) A[li * dp + lj] -= A[li * dp + k] * A[lj * dp + k];
    } else {
      for (int idx = lane; idx < d * d; idx += 64) {
        const int i = idx / d, j = idx - i * d;
        if (j > k && i >= j) A[i * dp + j] -= A[i * dp + k] * A[j * dp + k];
      }
    }
  }
  __syncthreads();
  // log det = 2 sum_k log L_kk: lane k takes one logarithm (d <= 32 <= 64 lanes), then a wave sum
  double logdet = lane < d ? log(A[lane * dp + lane]) : 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) logdet += __shfl_xor(logdet, off, 64);
  logdet *= 2.0;
  // Y = L^-1 (lower): lane c owns column c.
  if (lane < d) {
    const int c = lane;
    for (int i = 0; i < d; ++i) {
      double v = 0.0;
      if (i == c) v = 1.0 / A[i * dp + i];
      else if (i > c) {
        double s = 0.0;
        for (int k = c; k < i; ++k) s += A[i * dp + k] * Y[k * dp + c];
        v = -s / A[i * dp + i];
      }
      Y[i * dp + c] = v;
    }
  }
  __syncthreads();
  // A <- Y^T Y
  for (int idx = lane; idx < d * d; idx += 64) {
    const int i = idx / d, j = idx - i * d;
    const int k0 = i > j ? i : j;
    double s = 0.0;
    for (int k = k0; k < d; ++k) s += Y[k * dp + i] * Y[k * dp + j];
    A[i * dp + j] = s;
  }
  __syncthreads();
  return logdet;
}

