// Reverse-mode (adjoint) arithmetic of the composed rollout step and of one GP moment match -- SURVEY.md rows f-1 x f-2.
//
// The reference differentiates the whole policy-loss closure with tf.GradientTape (gpflow_pilco/utils/optimizers.py:51-56
// through loops/pilco.py:192-220); here every stage of the step has its hand-derived adjoint:
//   mma_encode_bwd        moment_matching/components.py:19-57 + maths.py:143-176   (sincos moments, Cov(x, e))
//   mma_cost_bwd          components.py:26-37                                      (expected saturating cost)
//   mma_step_bwd          dynamics/forward_sde.py:105-131 + solvers.py:110-135     (Cov(x, f) bookkeeping + Euler)
//   mma_head_bwd          moment_matching/bijectors.py:39-69 + gaussian.py:53-83   (NormalCDF head, chain rule, joint)
//   mma_policy_small_bwd  moment_matching/models.py:200-299, mean-only, one latent -- w.r.t. the input moments AND the
//                         packed policy (Z, beta, Lambda, variance, mean)
//   mma_gp_item_bwd       the same handler for a frozen multi-output model (the drift): the M-sized part of
//                         d(f1, Sff, cross)/d(mu, Sigma) from the M x M sums of mm_backward_sums (mm_backward.hip)
// All in f64.  The functions are written once for an execution context `Ctx` (lane(), nl(), sync()): a HIP workgroup
// on the device (MMADevCtx) and one host thread in the CPU build that tests/ uses to check the arithmetic against
// autograd / finite differences without a GPU (MMAHostCtx; tests/hostcheck -- not part of the product library).
// Phases of independent outputs are separated by ctx.sync(); no cross-lane intrinsics.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_compose.h"
#include "mm_mono.h"

#define MMA_FN template <class Ctx> __host__ __device__ inline

// one wave of a workgroup working on its own data (d x d items handled by several waves at once): LDS traffic of a
// single wave is processed in order, so a wave-scope fence (no reordering by the compiler) is all a phase boundary needs
struct MMAWaveCtx {
  __device__ void stamp(int) const {}
  __device__ int lane() const { return (int)(threadIdx.x & 63u); }
  __device__ int nl() const { return 64; }
  __device__ void sync() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
};
struct MMADevCtx {
  // optional stage profile (tools/profile_small.py): cycles between consecutive stamp() calls, accumulated per stage id
  long long* prof = nullptr;
  long long* last = nullptr;
  __device__ void stamp(int k) const {
    if (prof && threadIdx.x == 0 && blockIdx.x == 0) { const long long t = clock64(); prof[k] += t - *last; *last = t; }
  }
  __device__ int lane() const { return (int)threadIdx.x; }
  __device__ int nl() const { return (int)blockDim.x; }
  __device__ void sync() const { __syncthreads(); }
  // groups = waves: independent small items are dealt to them and run concurrently in sub()
  __device__ int group() const { return (int)(threadIdx.x >> 6); }
  __device__ int ngroups() const { return (int)(blockDim.x >> 6); }
  __device__ MMAWaveCtx sub() const { return MMAWaveCtx(); }
  // sum of v[k] over all lanes of the workgroup, result in every lane; scratch: ngroups() * N doubles
  template <int N>
  __device__ void reduce(double (&v)[N], double* scratch) const {
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
    }
    const int g = group(), ng = ngroups();
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
      for (int k = 0; k < N; ++k) scratch[g * N + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
      double s = 0.0;
      for (int t = 0; t < ng; ++t) s += scratch[t * N + k];
      v[k] = s;
    }
    __syncthreads();
  }
};
struct MMAHostCtx {
  __host__ __device__ void stamp(int) const {}
  __host__ __device__ int lane() const { return 0; }
  __host__ __device__ int nl() const { return 1; }
  __host__ __device__ void sync() const {}
  __host__ __device__ int group() const { return 0; }
  __host__ __device__ int ngroups() const { return 1; }
  __host__ __device__ MMAHostCtx sub() const { return MMAHostCtx(); }
  template <int N>
  __host__ __device__ void reduce(double (&)[N], double*) const {}
};

#define MMA_INV_SQRT_2PI 0.39894228040143267794
#define MMA_INV_2PI 0.15915494309189533577

// ---- small dense helpers (row-major; no sync inside unless said) -------------------------------------------------
// C [n, m] (ldc) = A [n, k] (lda) * B [k, m] (ldb)
MMA_FN void mma_mm(Ctx c, int n, int k, int m, const double* A, int lda, const double* B, int ldb, double* C, int ldc) {
  for (int idx = c.lane(); idx < n * m; idx += c.nl()) {
    const int i = idx / m, j = idx - i * m;
    double s = 0.0;
    for (int t = 0; t < k; ++t) s = fma(A[i * lda + t], B[t * ldb + j], s);
    C[i * ldc + j] = s;
  }
}

// In-place inverse of an SPD d x d matrix (row stride dp), log det returned (every lane); Y: d x dp scratch.
// *ok cleared on a non-positive pivot.  Syncs inside; A must be complete (synced) on entry, is complete on exit.
MMA_FN double mma_spd_inverse(Ctx c, double* A, double* Y, int d, int dp, bool* ok) {
  const int lane = c.lane(), nl = c.nl();
  for (int k = 0; k < d; ++k) {
    c.sync();
    const double akk = A[k * dp + k];
    if (!(akk > 0.0)) *ok = false;
    const double lkk = sqrt(akk);
    c.sync();
    for (int i = k + 1 + lane; i < d; i += nl) A[i * dp + k] /= lkk;
    if (lane == 0) A[k * dp + k] = lkk;
    c.sync();
    for (int idx = lane; idx < d * d; idx += nl) {
      const int i = idx / d, j = idx - i * d;
      if (j > k && i >= j) A[i * dp + j] -= A[i * dp + k] * A[j * dp + k];
    }
  }
  c.sync();
  double logdet = 0.0;
  for (int k = 0; k < d; ++k) logdet += log(A[k * dp + k]);
  logdet *= 2.0;
  for (int cc = lane; cc < d; cc += nl) {            // Y = L^-1, one column per lane
    for (int i = 0; i < d; ++i) {
      double v = 0.0;
      if (i == cc) v = 1.0 / A[i * dp + i];
      else if (i > cc) {
        double s = 0.0;
        for (int k = cc; k < i; ++k) s += A[i * dp + k] * Y[k * dp + cc];
        v = -s / A[i * dp + i];
      }
      Y[i * dp + cc] = v;
    }
  }
  c.sync();
  for (int idx = lane; idx < d * d; idx += nl) {     // A <- Y^T Y
    const int i = idx / d, j = idx - i * d;
    const int k0 = i > j ? i : j;
    double s = 0.0;
    for (int k = k0; k < d; ++k) s += Y[k * dp + i] * Y[k * dp + j];
    A[i * dp + j] = s;
  }
  c.sync();
  return logdet;
}

// Gauss-Jordan-free solve of A X = R for nr right-hand sides by elimination with partial pivoting on the augmented
// matrix Aug [n][n + nr] (row stride ld); on exit the right block holds X.  Returns det(A) (every lane).  Syncs inside.
MMA_FN double mma_solve_pivot(Ctx c, double* Aug, int n, int nr, int ld) {
  const int lane = c.lane(), nl = c.nl();
  double det = 1.0;
  for (int k = 0; k < n; ++k) {
    c.sync();
    int p = k; double best = fabs(Aug[k * ld + k]);      // every lane finds the same pivot
    for (int i = k + 1; i < n; ++i) { const double v = fabs(Aug[i * ld + k]); if (v > best) { best = v; p = i; } }
    c.sync();
    if (p != k) {
      for (int j = lane; j < n + nr; j += nl) { const double t = Aug[k * ld + j]; Aug[k * ld + j] = Aug[p * ld + j]; Aug[p * ld + j] = t; }
      det = -det;
    }
    c.sync();
    const double akk = Aug[k * ld + k];
    det *= akk;
    const int nrow = n - 1 - k, ncol = n + nr - 1 - k;   // columns k+1 .. n+nr-1
    for (int idx = lane; idx < nrow * ncol; idx += nl) {
      const int i = k + 1 + idx / ncol, j = k + 1 + idx % ncol;
      Aug[i * ld + j] -= (Aug[i * ld + k] / akk) * Aug[k * ld + j];
    }
  }
  c.sync();
  for (int r = lane; r < nr; r += nl) {                  // back substitution, one right-hand side per lane
    for (int i = n - 1; i >= 0; --i) {
      double s = Aug[i * ld + n + r];
      for (int j = i + 1; j < n; ++j) s -= Aug[i * ld + j] * Aug[j * ld + n + r];
      Aug[i * ld + n + r] = s / Aug[i * ld + i];
    }
  }
  c.sync();
  return det;
}

// ---- device fast paths for n <= 8: one matrix entry per lane of ONE wave (lane = 8 i + j), broadcasts by wave shuffles ----
// The phase-per-barrier versions above are latency chains on the device (an SPD inverse of a 6 x 6 matrix: ~20 barrier
// phases, 5 us); these take ~0.5 us.  Chosen by overload for the device contexts; the host context keeps the generic code.
#if defined(__HIPCC__)
// Gauss-Jordan inverse of an SPD matrix held in `a` (identity outside d x d); returns the entry of the inverse, sets *logdet
__device__ inline double mma_gj_spd8(double a, int d, bool* ok, double* logdet) {
  const int lane = (int)(threadIdx.x & 63u), i = lane >> 3, j = lane & 7;
  double pk = 1.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < d) {
      const double p = __shfl(a, 9 * k, 64);
      if (!(p > 0.0)) *ok = false;
      if (lane == k) pk = p;
      const double ip = 1.0 / p;
      const double rk = __shfl(a, 8 * k + j, 64), ci = __shfl(a, 8 * i + k, 64);
      if (i == k) a = (j == k) ? ip : rk * ip;
      else a = (j == k) ? -ci * ip : fma(-ci * ip, rk, a);
    }
  }
  a = 0.5 * (a + __shfl(a, 8 * j + i, 64));
  double ld = lane < d ? log(pk) : 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ld += __shfl_xor(ld, off, 64);
  *logdet = ld;
  return a;
}

__device__ inline double mma_spd_inverse(MMAWaveCtx c, double* A, double* Y, int d, int dp, bool* ok) {
  if (d > 8) return mma_spd_inverse<MMAWaveCtx>(c, A, Y, d, dp, ok);
  const int lane = c.lane(), i = lane >> 3, j = lane & 7;
  const bool in = i < d && j < d;
  c.sync();
  double ld;
  const double a = mma_gj_spd8(in ? A[i * dp + j] : (i == j ? 1.0 : 0.0), d, ok, &ld);
  if (in) A[i * dp + j] = a;
  c.sync();
  return ld;
}

// whole workgroup: wave 0 does the work, the log-determinant reaches every lane through Y[0]
__device__ inline double mma_spd_inverse(MMADevCtx c, double* A, double* Y, int d, int dp, bool* ok) {
  if (d > 8) return mma_spd_inverse<MMADevCtx>(c, A, Y, d, dp, ok);
  c.sync();
  if (c.group() == 0) {
    const int lane = c.lane(), i = lane >> 3, j = lane & 7;
    const bool in = i < d && j < d;
    double ld;
    bool okw = true;
    const double a = mma_gj_spd8(in ? A[i * dp + j] : (i == j ? 1.0 : 0.0), d, &okw, &ld);
    if (in) A[i * dp + j] = a;
    if (lane == 0) { Y[0] = ld; Y[1] = okw ? 1.0 : 0.0; }
  }
  c.sync();
  const double ld = Y[0];
  if (Y[1] == 0.0) *ok = false;
  c.sync();
  return ld;
}

// A X = R, n <= 8 unknowns and nr <= 8 right-hand sides, partial pivoting: wave 0, one entry of A and one of R per lane
__device__ inline double mma_solve_pivot(MMADevCtx c, double* Aug, int n, int nr, int ld) {
  if (n > 8 || nr > 8) return mma_solve_pivot<MMADevCtx>(c, Aug, n, nr, ld);
  c.sync();
  if (c.group() == 0) {
    const int lane = c.lane(), i = lane >> 3, j = lane & 7;
    double a = (i < n && j < n) ? Aug[i * ld + j] : (i == j ? 1.0 : 0.0);
    double b = (i < n && j < nr) ? Aug[i * ld + n + j] : 0.0;
    double det = 1.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < n) {
        double best = fabs(__shfl(a, 9 * k, 64));
        int p = k;
#pragma unroll
        for (int r = 1; r < 8; ++r) {
          if (k + r < n) {
            const double v = fabs(__shfl(a, 8 * (k + r) + k, 64));
            if (v > best) { best = v; p = k + r; }
          }
        }
        if (p != k) {
          const int si = i == k ? p : (i == p ? k : i);
          a = __shfl(a, 8 * si + j, 64); b = __shfl(b, 8 * si + j, 64);
          det = -det;
        }
        const double piv = __shfl(a, 9 * k, 64), ip = 1.0 / piv;
        det *= piv;
        const double rk = __shfl(a, 8 * k + j, 64), bk = __shfl(b, 8 * k + j, 64), ci = __shfl(a, 8 * i + k, 64);
        if (i == k) { a = rk * ip; b = bk * ip; }
        else { a = fma(-ci * ip, rk, a); b = fma(-ci * ip, bk, b); }
      }
    }
    if (i < n && j < nr) Aug[i * ld + n + j] = b;
    if (lane == 0) Aug[0] = det;                          // (the eliminated left block is not read again: carries det)
  }
  c.sync();
  const double det = Aug[0];
  c.sync();
  return det;
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// mma_encode_bwd: adjoint of the trigonometric encoding (mmc_encode_body in mm_compose.hip)
//   (m [nx], S [nx, nx])  ->  me = [sin a, cos a, x_inactive] [ne], See [ne, ne], Sxe [nx, ne]
// in: gme, gSee, gSxe (adjoints of the outputs; gSee need not be symmetric);  gm, gS: ACCUMULATED (+=).
// sm: scratch of mma_encode_bwd_scratch(nx, na) doubles.
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int mma_encode_bwd_scratch(int nx, int na) { return 5 * na + 4 * na * na + nx * 2 * na; }

MMA_FN void mma_encode_bwd(Ctx c, const MMComposeDims& D, const double* m, const double* S, const double* gme,
                           const double* gSee, const double* gSxe, double* gm, double* gS, double* sm) {
  const int lane = c.lane(), nl = c.nl();
  const int nx = D.nx, na = D.na, nb = D.nb, ne = D.ne, n2 = 2 * na;
  double* s1 = sm; double* c1 = s1 + na; double* gs1 = c1 + na; double* gc1 = gs1 + na; double* gv = gc1 + na;
  double* pai = gv + na; double* paj = pai + na * na; double* pv = paj + na * na; double* ps = pv + na * na;
  double* gSxy = ps + na * na;                        // [nx][n2]
  for (int i = lane; i < na; i += nl) {
    const int r = D.active[i];
    const double ev = exp(-0.5 * S[r * nx + r]);
    s1[i] = ev * sin(m[r]); c1[i] = ev * cos(m[r]);
  }
  // adjoint of Sxy = Cov(x, [sin a, cos a]): from Sxe[:, :n2] and from the Sby / Sby^T blocks of See
  for (int idx = lane; idx < nx * n2; idx += nl) {
    const int r = idx / n2, k = idx - r * n2;
    double g = gSxe[r * ne + k];
    const int sl = D.slot[r];
    if (sl >= na) { const int ib = sl - na; g += gSee[(n2 + ib) * ne + k] + gSee[k * ne + n2 + ib]; }
    gSxy[idx] = g;
  }
  c.sync();
  for (int i = lane; i < na; i += nl) {
    const int ri = D.active[i];
    double a1 = gme[i], a2 = gme[na + i];
    for (int j = 0; j < na; ++j) {
      a1 -= (gSee[i * ne + j] + gSee[j * ne + i]) * s1[j] + (gSee[i * ne + na + j] + gSee[(na + j) * ne + i]) * c1[j];
      a2 -= (gSee[(na + i) * ne + na + j] + gSee[(na + j) * ne + na + i]) * c1[j]
          + (gSee[j * ne + na + i] + gSee[(na + i) * ne + j]) * s1[j];
    }
    for (int r = 0; r < nx; ++r) {
      a1 -= gSxy[r * n2 + na + i] * S[r * nx + ri];
      a2 += gSxy[r * n2 + i] * S[r * nx + ri];
    }
    gs1[i] = a1; gc1[i] = a2;
  }
  for (int idx = lane; idx < na * na; idx += nl) {     // ordered pairs (i, j): adjoints of s2, c2, sc
    const int i = idx / na, j = idx - i * na;
    const int ri = D.active[i], rj = D.active[j];
    const double ai = m[ri], aj = m[rj], vi = S[ri * nx + ri], vj = S[rj * nx + rj];
    const double sij = 0.5 * (S[ri * nx + rj] + S[rj * nx + ri]);
    const double A = exp(-0.5 * (vi + vj) - sij), Bm = exp(-0.5 * (vi + vj) + sij);
    const double si = sin(ai), ci = cos(ai), sj = sin(aj), cj = cos(aj);
    const double gs2 = gSee[i * ne + j], gc2 = gSee[(na + i) * ne + na + j];
    const double gsc = gSee[i * ne + na + j] + gSee[(na + j) * ne + i];
    const double gB = 0.5 * (gs2 + gc2), gA = 0.5 * (gc2 - gs2);            // adjoints of B cos(ai - aj), A cos(ai + aj)
    const double cp_ = cos(ai + aj), sp_ = sin(ai + aj), cm_ = cos(ai - aj), sm_ = sin(ai - aj);
    const double gBm = gB * cm_ + 0.5 * gsc * (si * cj - sj * ci);
    const double gAe = gA * cp_ + 0.5 * gsc * (si * cj + sj * ci);
    pai[idx] = -gA * A * sp_ - gB * Bm * sm_ + 0.5 * gsc * (ci * cj * (Bm + A) + sj * si * (Bm - A));
    paj[idx] = -gA * A * sp_ + gB * Bm * sm_ + 0.5 * gsc * (-si * sj * (Bm + A) - cj * ci * (Bm - A));
    pv[idx] = -0.5 * (gAe * A + gBm * Bm);
    ps[idx] = -gAe * A + gBm * Bm;
  }
  c.sync();
  for (int i = lane; i < na; i += nl) {
    double ga = gs1[i] * c1[i] - gc1[i] * s1[i];
    double g2 = -0.5 * (gs1[i] * s1[i] + gc1[i] * c1[i]);
    for (int j = 0; j < na; ++j) { ga += pai[i * na + j] + paj[j * na + i]; g2 += pv[i * na + j] + pv[j * na + i]; }
    gm[D.active[i]] += ga;
    gv[i] = g2;
  }
  for (int k = lane; k < nb; k += nl) gm[D.inactive[k]] += gme[n2 + k];
  c.sync();
  for (int idx = lane; idx < nx * nx; idx += nl) {
    const int r = idx / nx, cc = idx - r * nx;
    const int sr = D.slot[r], sc = D.slot[cc];
    double g = 0.0;
    if (sc >= na) {
      g += gSxe[r * ne + n2 + (sc - na)];
      if (sr >= na) g += gSee[(n2 + sr - na) * ne + n2 + (sc - na)];
    } else {
      g += gSxy[r * n2 + sc] * c1[sc] - gSxy[r * n2 + na + sc] * s1[sc];
      if (sr < na) {
        g += 0.5 * (ps[sr * na + sc] + ps[sc * na + sr]);
        if (sr == sc) g += gv[sr];
      }
    }
    gS[idx] += g;
  }
  c.sync();
}

// ---------------------------------------------------------------------------------------------------------------------
// mma_cost_bwd: cost = -det(I + S W)^-1/2 exp(-1/2 e^T W (I + S W)^-1 e), e = mean - target (components.py:29-37).
// With Q = (I + W S)^-1 W (symmetric):  d cost = cost (-1/2 <Q, dS> - de^T Q e + 1/2 (Qe)^T dS (Qe)).
// gmean, gcov ACCUMULATED with weight gc.  Returns the cost.  sm: n (2n + 3) doubles.
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int mma_cost_bwd_scratch(int n) { return n * (2 * n + 3); }

MMA_FN double mma_cost_bwd(Ctx c, int n, const double* mean, const double* cov, const double* target, const double* W,
                           double gc, double* gmean, double* gcov, double* sm) {
  const int lane = c.lane(), nl = c.nl(), ld = 2 * n;
  double* Aug = sm;                 // [n][2n]: I + W S | W  ->  . | Q
  double* e = Aug + n * ld;         // [n]
  double* Qe = e + n;               // [n]
  for (int idx = lane; idx < n * n; idx += nl) {
    const int i = idx / n, j = idx - i * n;
    double s = i == j ? 1.0 : 0.0;
    for (int k = 0; k < n; ++k) s = fma(W[i * n + k], cov[k * n + j], s);
    Aug[i * ld + j] = s;
    Aug[i * ld + n + j] = W[idx];
  }
  for (int i = lane; i < n; i += nl) e[i] = mean[i] - target[i];
  const double det = mma_solve_pivot(c, Aug, n, n, ld);
  for (int i = lane; i < n; i += nl) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s = fma(0.5 * (Aug[i * ld + n + j] + Aug[j * ld + n + i]), e[j], s);
    Qe[i] = s;
  }
  c.sync();
  double dist2 = 0.0;
  for (int i = 0; i < n; ++i) dist2 = fma(e[i], Qe[i], dist2);
  const double cost = -exp(-0.5 * dist2) / sqrt(det);
  const double f = gc * cost;
  for (int i = lane; i < n; i += nl) gmean[i] -= f * Qe[i];
  for (int idx = lane; idx < n * n; idx += nl) {
    const int i = idx / n, j = idx - i * n;
    const double q = 0.5 * (Aug[i * ld + n + j] + Aug[j * ld + n + i]);
    gcov[idx] += f * 0.5 * (Qe[i] * Qe[j] - q);
  }
  c.sync();
  return cost;
}

// ---------------------------------------------------------------------------------------------------------------------
// mma_step_bwd: adjoint of mmc_step_body (forward_sde.py:105-131 + solvers.py:110-135)
//   Sxd = rows [Sxe_r, Sxe_r . cp] (encoded dims) | rows of Sdd (the other dims);  Sxf = Sxd dcross;
//   m' = m + dt df1;  S' = S + dt (Sxf + Sxf^T) + dt^2 dSff.
// in : gm1 [nx], gS1 [nx, nx] (adjoint of the new state)
// out: gSxe [nx, ne], gcp [ne], gSdd [nd, nd], gdf1 [nx], gdSff [nx, nx], gdcross [nd, nx]  (all ASSIGNED);
//      the adjoint of (m, S) through the direct terms is (gm1, gS1) itself.
// sm: nx (2 nd + nx) doubles.
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int mma_step_bwd_scratch(int nx, int nd) { return nx * (2 * nd + nx); }

MMA_FN void mma_step_bwd(Ctx c, const MMComposeDims& D, double dt, const double* Sxe, const double* cp, const double* Sdd,
                         const double* dcross, const double* gm1, const double* gS1, double* gSxe, double* gcp, double* gSdd,
                         double* gdf1, double* gdSff, double* gdcross, double* sm) {
  const int lane = c.lane(), nl = c.nl();
  const int nx = D.nx, na = D.na, ne = D.ne, nd = D.nd, n2 = 2 * na;
  double* Sxd = sm; double* gSxd = Sxd + nx * nd; double* gSxf = gSxd + nx * nd;
  for (int idx = lane; idx < nx * nd; idx += nl) {
    const int r = idx / nd, k = idx - r * nd;
    const int sl = D.slot[r];
    double v;
    if (sl < na) {
      if (k < ne) v = Sxe[r * ne + k];
      else { double s = 0.0; for (int l = 0; l < ne; ++l) s = fma(Sxe[r * ne + l], cp[l], s); v = s; }
    } else {
      v = Sdd[(n2 + (sl - na)) * nd + k];
    }
    Sxd[idx] = v;
  }
  for (int idx = lane; idx < nx * nx; idx += nl) {
    const int r = idx / nx, cc = idx - r * nx;
    gSxf[idx] = dt * (gS1[idx] + gS1[cc * nx + r]);
    gdSff[idx] = dt * dt * gS1[idx];
  }
  for (int i = lane; i < nx; i += nl) gdf1[i] = dt * gm1[i];
  c.sync();
  for (int idx = lane; idx < nx * nd; idx += nl) {        // gSxd = gSxf dcross^T
    const int r = idx / nd, k = idx - r * nd;
    double s = 0.0;
    for (int cc = 0; cc < nx; ++cc) s = fma(gSxf[r * nx + cc], dcross[k * nx + cc], s);
    gSxd[idx] = s;
  }
  for (int idx = lane; idx < nd * nx; idx += nl) {        // gdcross = Sxd^T gSxf
    const int k = idx / nx, cc = idx - k * nx;
    double s = 0.0;
    for (int r = 0; r < nx; ++r) s = fma(Sxd[r * nd + k], gSxf[r * nx + cc], s);
    gdcross[idx] = s;
  }
  c.sync();
  for (int idx = lane; idx < nx * ne; idx += nl) {
    const int r = idx / ne, k = idx - r * ne;
    gSxe[idx] = D.slot[r] < na ? gSxd[r * nd + k] + gSxd[r * nd + ne] * cp[k] : 0.0;
  }
  for (int l = lane; l < ne; l += nl) {
    double s = 0.0;
    for (int i = 0; i < na; ++i) { const int r = D.active[i]; s = fma(gSxd[r * nd + ne], Sxe[r * ne + l], s); }
    gcp[l] = s;
  }
  for (int idx = lane; idx < nd * nd; idx += nl) {
    const int i = idx / nd, k = idx - i * nd;
    gSdd[idx] = (i >= n2 && i < ne) ? gSxd[D.inactive[i - n2] * nd + k] : 0.0;
  }
  c.sync();
}

// ---------------------------------------------------------------------------------------------------------------------
// mma_head_bwd: adjoint of k_compose_policy (bijectors.py:39-69 1-D branch, gaussian.py:53-83)
//   vx = max(pSff, 0), z = pf1 / sqrt(vx + 1), y1 = Phi(z), y2 = y1 - 2 T(z, 1 / sqrt(1 + 2 vx)),
//   mu_u = scale (y1 + shift), Suu = scale^2 (y2 - y1^2), cp = pcross scale phi(z) / sqrt(vx + 1), Seu = See cp,
//   md = [me, mu_u], Sdd = [[See, Seu], [Seu^T, Suu]].
// in : gmd [nd], gSdd [nd, nd], gcp [ne]  (adjoints of md, Sdd and of cp's use in the step's bookkeeping)
// out: gme [ne], gSee [ne, ne], gpcross [ne] ASSIGNED;  hs[0..1] = gpf1, gpSff.   sm: ne + 4 doubles (hs = sm + ne).
// ---------------------------------------------------------------------------------------------------------------------
MMA_FN void mma_head_bwd(Ctx c, int ne, double scale, double shift, double pf1, double pSff, const double* pcross,
                         const double* See, const double* gmd, const double* gSdd, const double* gcp,
                         double* gme, double* gSee, double* gpcross, double* sm) {
  const int lane = c.lane(), nl = c.nl(), nd = ne + 1;
  double* gct = sm; double* hs = sm + ne;
  const double vx = pSff > 0.0 ? pSff : 0.0;
  const double isq = 1.0 / sqrt(vx + 1.0), z = isq * pf1;
  const double aa = 1.0 / sqrt(1.0 + 2.0 * vx);
  const double phi = MMA_INV_SQRT_2PI * exp(-0.5 * z * z);
  const double y1 = 0.5 * erfc(-z * 0.70710678118654752440);
  const double head_pre = isq * phi * scale;
  for (int l = lane; l < ne; l += nl) {                  // total adjoint of cp: direct + through Seu = See cp
    double s = gcp[l];
    for (int k = 0; k < ne; ++k) s = fma(See[k * ne + l], gSdd[k * nd + ne] + gSdd[ne * nd + k], s);
    gct[l] = s;
  }
  c.sync();
  for (int k = lane; k < ne; k += nl) { gme[k] = gmd[k]; gpcross[k] = gct[k] * head_pre; }
  for (int idx = lane; idx < ne * ne; idx += nl) {
    const int k = idx / ne, l = idx - k * ne;
    gSee[idx] = gSdd[k * nd + l] + (gSdd[k * nd + ne] + gSdd[ne * nd + k]) * (pcross[l] * head_pre);
  }
  if (lane == 0) {
    double ghp = 0.0;
    for (int k = 0; k < ne; ++k) ghp = fma(gct[k], pcross[k], ghp);
    const double gmu_u = gmd[ne], gSuu = gSdd[ne * nd + ne];
    const double gy2 = scale * scale * gSuu;
    const double gy1 = scale * gmu_u - 2.0 * scale * scale * y1 * gSuu + gy2;
    const double gT = -2.0 * gy2;
    const double PhiAz = 0.5 * erfc(-aa * z * 0.70710678118654752440);
    const double gz = gy1 * phi - gT * phi * (PhiAz - 0.5) - ghp * z * head_pre;
    const double gaa = gT * exp(-0.5 * z * z * (1.0 + aa * aa)) * MMA_INV_2PI / (1.0 + aa * aa);
    const double gisq = ghp * scale * phi + gz * pf1;
    const double gvx = -0.5 * gisq * isq * isq * isq - gaa * aa * aa * aa;
    hs[0] = gz * isq;
    hs[1] = pSff > 0.0 ? gvx : 0.0;
  }
  (void)shift;
  c.sync();
}

// ---------------------------------------------------------------------------------------------------------------------
// mma_policy_small_bwd: adjoint of k_policy_match_small (one latent, mean-only; moment_matching/models.py:200-299 with
// model_uncertainty = False) w.r.t. the input moments (mu, Sigma) AND the packed model (Z, beta, ls2 = Lambda, var, mean).
//   P = (Sigma + Lambda)^-1, S0 = (Sigma + Lambda/2)^-1, E = sym(Lambda^-1 Sigma P), T = Lambda/2 - (Lambda/2) S0 (Lambda/2),
//   G = Lambda^-1 T Lambda^-1, zeta_m = z_m - mu, q_m = var |Lambda|^1/2 |Sigma + Lambda|^-1/2 exp(-zeta^T P zeta / 2),
//   w = beta q, f1 = sum w + c, cross = P sum w zeta, r_m = -zeta^T (E - G) zeta / 2,
//   Sff = sum_ij w_i expm1(r_i + r_j + cst + zeta_i^T G zeta_j) w_j.
// in : gf1, gSff, gcross [d].   out: gmu [d], gSig [d, d] (symmetric) ASSIGNED;  gpar (mm_policy_grad_len) ACCUMULATED.
// Sigma: [d, d], the lower triangle is read (as the forward does).  sm: mma_policy_small_bwd_scratch(M, d, nl) doubles.
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int mma_policy_nsub(int M, int nl) { const int n = nl / M; return n < 1 ? 1 : n; }
__host__ __device__ inline int mma_policy_small_bwd_scratch(int M, int d, int nl) {
  const int dp = d + 1, ns = mma_policy_nsub(M, nl);
  return 12 * d * dp + M * (5 * d + 8) + ns * M * (d + 2) + 4 * d + 16 + 3 * (nl > d * d + 2 * d + 2 ? nl : d * d + 2 * d + 2) + 16;
}

template <class Ctx, int DK>
__host__ __device__ inline void mma_policy_small_bwd(Ctx c, int M, int d, const double* Z, const double* beta, const double* ls2, double var,
                                 const double* mu, const double* Sigma, double gf1, double gSff, const double* gcross,
                                 double* gmu, double* gSig, double* gpar, double* sm, bool* ok) {
  const int lane = c.lane(), nl = c.nl(), dp = d + 1, msz = d * dp, ns = mma_policy_nsub(M, nl);
  double* Sg = sm;            double* Pm = Sg + msz;     double* S0 = Pm + msz;     double* Yw = S0 + msz;
  double* Em = Yw + msz;      double* Gm = Em + msz;     double* Tm = Gm + msz;
  double* Pb = Tm + msz;      double* Eb = Pb + msz;     double* Gb = Eb + msz;     double* W1 = Gb + msz;   double* W2 = W1 + msz;
  double* zs = W2 + msz;                 // [M][d] zeta
  double* Pz = zs + M * d;               // [M][d] P zeta
  double* gz = Pz + M * d;               // [M][d] G zeta
  double* dz = gz + M * d;               // [M][d] (E - G) zeta
  double* zb = dz + M * d;               // [M][d] adjoint of zeta
  double* ws = zb + M * d;  double* qs = ws + M;  double* rs = qs + M;  double* cs = rs + M;  double* Ks = cs + M;
  double* wb = Ks + M;      double* mb = wb + M;  double* rb = mb + M;       // adjoints of w, maha, r
  double* part = rb + M;                 // [ns][M][d + 2]
  double* sv = part + ns * M * (d + 2);  // [d] s = sum w zeta
  double* sb = sv + d;                   // [d] adjoint of s = P gcross
  double* lamb = sb + d;                 // [d] adjoint of Lambda
  double* Vb = lamb + d;                 // [d] adjoint of V
  double* sc = Vb + d;                   // scalars [16]
  double* slc = sc + 16;                 // [slices][d * d][3] partial matrix sums over the centres

  for (int idx = lane; idx < d * d; idx += nl) {
    const int i = idx / d, j = idx - i * d;
    const double s = i >= j ? Sigma[i * d + j] : Sigma[j * d + i];
    Sg[i * dp + j] = s;
    Pm[i * dp + j] = s + (i == j ? ls2[i] : 0.0);
    S0[i * dp + j] = s + (i == j ? 0.5 * ls2[i] : 0.0);
  }
  c.sync();
  c.stamp(0);
  const double ldA = mma_spd_inverse(c, Pm, Yw, d, dp, ok);
  const double ldS = mma_spd_inverse(c, S0, Yw, d, dp, ok);
  c.stamp(1);
  double sl = 0.0;
  for (int k = 0; k < d; ++k) sl += log(ls2[k]);
  const double lognorm = log(var) + 0.5 * sl - 0.5 * ldA;
  const double cst = -0.5 * ldS - 0.5 * (sl + d * 0.6931471805599453) + ldA;
  for (int idx = lane; idx < d * d; idx += nl) {
    const int i = idx / d, j = idx - i * d;
    double s1 = 0.0, t1 = 0.0, tt = 0.0;
    for (int k = 0; k < d; ++k) {
      s1 += Sg[i * dp + k] * Pm[k * dp + j];
      t1 += Sg[j * dp + k] * Pm[k * dp + i];
      tt += S0[i * dp + k] * Sg[k * dp + j];
    }
    Em[i * dp + j] = 0.5 * (s1 / ls2[i] + t1 / ls2[j]);
    Yw[i * dp + j] = 0.5 * ls2[i] * tt;
  }
  c.sync();
  for (int idx = lane; idx < d * d; idx += nl) {
    const int i = idx / d, j = idx - i * d;
    Tm[i * dp + j] = 0.5 * (Yw[i * dp + j] + Yw[j * dp + i]);
    Gm[i * dp + j] = Tm[i * dp + j] / (ls2[i] * ls2[j]);
  }
  for (int k = lane; k < d; k += nl) {
    double s = 0.0;
    for (int l = 0; l < d; ++l) s = fma(Pm[k * dp + l], gcross[l], s);
    sb[k] = s;
  }
  c.sync();
  c.stamp(2);
  // ---- per centre: forward quantities ---------------------------------------------------------------------------------
  for (int m = lane; m < M; m += nl) {
    double maha = 0.0, r1 = 0.0;
    for (int k = 0; k < d; ++k) zs[m * d + k] = Z[m * d + k] - mu[k];
    for (int i = 0; i < d; ++i) {
      double tp = 0.0, te = 0.0, tg = 0.0;
      for (int k = 0; k < d; ++k) {
        const double zk = zs[m * d + k];
        tp = fma(Pm[i * dp + k], zk, tp); te = fma(Em[i * dp + k], zk, te); tg = fma(Gm[i * dp + k], zk, tg);
      }
      Pz[m * d + i] = tp; gz[m * d + i] = tg; dz[m * d + i] = te - tg;
      maha = fma(zs[m * d + i], tp, maha);
      r1 = fma(zs[m * d + i], te - tg, r1);
    }
    qs[m] = exp(lognorm - 0.5 * maha);
    ws[m] = beta[m] * qs[m];
    rs[m] = -0.5 * r1;
  }
  c.sync();
  c.stamp(3);
  // ---- M x M sweep: c_i = sum_j E_ij w_j, K_i = sum_j Omega_ij, U_i = sum_j Omega_ij zeta_j -----------------------------
  for (int idx = lane; idx < ns * M; idx += nl) {
    const int i = idx % M, sub = idx / M;
    double ci = 0.0, Ki = 0.0;
    double* pu = part + (size_t)idx * (d + 2);
    for (int k = 0; k < d; ++k) pu[k] = 0.0;
    for (int j = sub; j < M; j += ns) {
      double delta = rs[i] + rs[j] + cst;
      for (int k = 0; k < d; ++k) delta = fma(zs[i * d + k], gz[j * d + k], delta);
      const double Ex = expm1(fmin(delta, MM_EXP_CAP_F64));                    // (mm_common.h: exponent caps)
      const double om = ws[i] * ws[j] * (Ex + 1.0);
      ci = fma(Ex, ws[j], ci); Ki += om;
      for (int k = 0; k < d; ++k) pu[k] = fma(om, zs[j * d + k], pu[k]);
    }
    pu[d] = ci; pu[d + 1] = Ki;
  }
  c.sync();
  c.stamp(4);
  // combine the sub-sweeps (into the slot of sub-sweep 0): U_i [d], c_i, K_i
  for (int idx = lane; idx < M * (d + 2); idx += nl) {
    const int i = idx / (d + 2), k = idx - i * (d + 2);
    double v = 0.0;
    for (int sub = 0; sub < ns; ++sub) v += part[(size_t)(sub * M + i) * (d + 2) + k];
    part[(size_t)i * (d + 2) + k] = v;
  }
  c.sync();
  for (int i = lane; i < M; i += nl) {                  // per-centre scalar adjoints
    const double ci = part[(size_t)i * (d + 2) + d], Ki = part[(size_t)i * (d + 2) + d + 1];
    cs[i] = ci; Ks[i] = Ki;
    double sz = 0.0;
    for (int k = 0; k < d; ++k) sz = fma(sb[k], zs[i * d + k], sz);
    const double wbar = gf1 + 2.0 * gSff * ci + sz;
    wb[i] = wbar; mb[i] = -0.5 * wbar * ws[i]; rb[i] = 2.0 * gSff * Ki;
  }
  c.sync();
  for (int idx = lane; idx < M * d; idx += nl) {        // adjoint of zeta_i, one (centre, dimension) per lane
    const int i = idx / d, k = idx - i * d;
    double gu = 0.0;
    for (int l = 0; l < d; ++l) gu = fma(Gm[k * dp + l], part[(size_t)i * (d + 2) + l], gu);
    zb[idx] = 2.0 * gSff * gu + ws[i] * sb[k] + 2.0 * mb[i] * Pz[idx] - rb[i] * dz[idx];
  }
  c.sync();
  c.stamp(5);
  // ---- sums over the centres, in slices of the centre range (one (entry, slice) per lane, then one lane per entry): the
  // d x d matrices sum mbar zz^T, sum rbar zz^T, X = sum zeta U^T, the vectors s = sum w zeta, sum zbar, and two scalars ------
  {
    const int dd = d * d, ne_ = dd + 2 * d + 2, nsl = nl / ne_ < 1 ? 1 : nl / ne_;
    for (int idx = lane; idx < nsl * ne_; idx += nl) {
      const int e = idx % ne_, sl2 = idx / ne_;
      double v0 = 0.0, v1 = 0.0, v2 = 0.0;
      if (e < dd) {
        const int a = e / d, b = e - a * d;
        for (int i = sl2; i < M; i += nsl) {
          const double zz = zs[i * d + a] * zs[i * d + b];
          v0 = fma(mb[i], zz, v0); v1 = fma(rb[i], zz, v1);
          v2 = fma(zs[i * d + a], part[(size_t)i * (d + 2) + b], v2);
        }
      } else if (e < dd + d) {
        const int k = e - dd;
        for (int i = sl2; i < M; i += nsl) v0 = fma(ws[i], zs[i * d + k], v0);
      } else if (e < dd + 2 * d) {
        const int k = e - dd - d;
        for (int i = sl2; i < M; i += nsl) v0 += zb[i * d + k];
      } else if (e == dd + 2 * d) {
        for (int i = sl2; i < M; i += nsl) v0 = fma(wb[i], ws[i], v0);
      } else {
        for (int i = sl2; i < M; i += nsl) v0 += Ks[i];
      }
      slc[(size_t)idx * 3] = v0; slc[(size_t)idx * 3 + 1] = v1; slc[(size_t)idx * 3 + 2] = v2;
    }
    c.sync();
    for (int e = lane; e < ne_; e += nl) {
      double v0 = 0.0, v1 = 0.0, v2 = 0.0;
      for (int t = 0; t < nsl; ++t) { v0 += slc[(size_t)(t * ne_ + e) * 3]; v1 += slc[(size_t)(t * ne_ + e) * 3 + 1]; v2 += slc[(size_t)(t * ne_ + e) * 3 + 2]; }
      if (e < dd) {
        const int a = e / d, b = e - a * d;
        Pb[a * dp + b] = v0;           // sum mbar zeta zeta^T            (+ sym(gcross s^T) + from E, below)
        Eb[a * dp + b] = -0.5 * v1;    // adjoint of E
        Gb[a * dp + b] = 0.5 * v1;     // adjoint of G: + gSff sym(X), below
        W1[a * dp + b] = v2;           // X
      } else if (e < dd + d) sv[e - dd] = v0;
      else if (e < dd + 2 * d) gmu[e - dd - d] = -v0;
      else if (e == dd + 2 * d) sc[0] = v0;              // adjoint of lognorm
      else sc[1] = gSff * v0;                            // adjoint of cst
    }
  }
  c.sync();
  c.stamp(6);
  const double lbar = sc[0], cbar = sc[1];
  const double ldAb = -0.5 * lbar + cbar, ldSb = -0.5 * cbar;
  for (int idx = lane; idx < d * d; idx += nl) {
    const int a = idx / d, b = idx - a * d;
    Gb[a * dp + b] += 0.5 * gSff * (W1[a * dp + b] + W1[b * dp + a]);
    Pb[a * dp + b] += 0.5 * (gcross[a] * sv[b] + gcross[b] * sv[a]);
  }
  c.sync();
  // from E = sym(Lambda^-1 Sigma P):  Pbar += sym(Sigma Lambda^-1 Ebar);  W2 = Sigma P (for the Lambda and Sigma terms)
  for (int idx = lane; idx < d * d; idx += nl) {
    const int a = idx / d, b = idx - a * d;
    double s = 0.0, t = 0.0;
    for (int k = 0; k < d; ++k) { s = fma(Sg[a * dp + k] / ls2[k], Eb[k * dp + b], s); t = fma(Sg[a * dp + k], Pm[k * dp + b], t); }
    W1[a * dp + b] = s;              // Sigma Lambda^-1 Ebar
    W2[a * dp + b] = t;              // Sigma P
  }
  c.sync();
  for (int idx = lane; idx < d * d; idx += nl) {
    const int a = idx / d, b = idx - a * d;
    Pb[a * dp + b] += 0.5 * (W1[a * dp + b] + W1[b * dp + a]);
  }
  c.sync();
  // Abar = -P Pbar P + ldAbar P;   Yw = Pbar P
  mma_mm(c, d, d, d, Pb, dp, Pm, dp, Yw, dp);
  c.sync();
  mma_mm(c, d, d, d, Pm, dp, Yw, dp, W1, dp);          // W1 = P Pbar P
  // Tbar = Gbar o 1 / (lam_i lam_j)  (into Tm's adjoint slot: reuse Eb after its last use below -> use Yw later)
  c.sync();
  for (int k = lane; k < d; k += nl) {
    // Lambda adjoint, part 1: lognorm, cst, A = Sigma + Lambda, E's Lambda^-1, G's Lambda^-1
    double g = 0.5 * lbar / ls2[k] - 0.5 * cbar / ls2[k] + (-W1[k * dp + k] + ldAb * Pm[k * dp + k]);
    double se = 0.0, sg = 0.0;
    for (int l = 0; l < d; ++l) { se = fma(W2[k * dp + l], Eb[l * dp + k], se); sg = fma(Gb[k * dp + l], Gm[k * dp + l], sg); }
    g += -se / (ls2[k] * ls2[k]) - 2.0 * sg / ls2[k];
    lamb[k] = g;
  }
  for (int idx = lane; idx < d * d; idx += nl) {
    const int a = idx / d, b = idx - a * d;
    // gSig so far: Abar + sym(Lambda^-1 Ebar P)
    double s = 0.0, t = 0.0;
    for (int k = 0; k < d; ++k) { s = fma(Eb[a * dp + k], Pm[k * dp + b], s); t = fma(Eb[b * dp + k], Pm[k * dp + a], t); }
    gSig[a * d + b] = -W1[a * dp + b] + ldAb * Pm[a * dp + b] + 0.5 * (s / ls2[a] + t / ls2[b]);
    Yw[a * dp + b] = Gb[a * dp + b] / (ls2[a] * ls2[b]);                 // Tbar
  }
  c.sync();
  // T = V - V S0 V, V = Lambda / 2:  Vbar_k = Tbar_kk - 2 (S0 V Tbar)_kk;  S0bar = -V Tbar V;  Bbar = S0 (V Tbar V) S0 + ldSbar S0
  for (int idx = lane; idx < d * d; idx += nl) {
    const int a = idx / d, b = idx - a * d;
    W2[a * dp + b] = 0.25 * ls2[a] * Yw[a * dp + b] * ls2[b];            // V Tbar V
  }
  for (int k = lane; k < d; k += nl) {
    double s = 0.0;
    for (int l = 0; l < d; ++l) s = fma(S0[k * dp + l] * 0.5 * ls2[l], Yw[l * dp + k], s);
    Vb[k] = Yw[k * dp + k] - 2.0 * s;
  }
  c.sync();
  mma_mm(c, d, d, d, W2, dp, S0, dp, W1, dp);
  c.sync();
  mma_mm(c, d, d, d, S0, dp, W1, dp, W2, dp);          // W2 = S0 (V Tbar V) S0
  c.sync();
  for (int idx = lane; idx < d * d; idx += nl) {
    const int a = idx / d, b = idx - a * d;
    gSig[a * d + b] += 0.5 * (W2[a * dp + b] + W2[b * dp + a]) + ldSb * S0[a * dp + b];
  }
  c.stamp(7);
  // ---- the packed model's gradient (accumulated over the steps of a rollout) ---------------------------------------------
  for (int idx = lane; idx < M * d; idx += nl) gpar[idx] += zb[idx];
  for (int i = lane; i < M; i += nl) gpar[M * d + i] += wb[i] * qs[i];
  for (int k = lane; k < d; k += nl) {
    const double vbar = Vb[k] + W2[k * dp + k] + ldSb * S0[k * dp + k];
    gpar[M * d + M + k] += lamb[k] + 0.5 * vbar;
  }
  if (lane == 0) {
    gpar[M * d + M + d] += lbar / var;
    gpar[M * d + M + d + 1] += gf1;
  }
  c.sync();
}

// ---------------------------------------------------------------------------------------------------------------------
// Off-diagonal pair aggregates of an f32 model (csrc/mm_bwd_f32.hip).  With Omega_ij = what_i what'_j e^{b_ij},
// b_ij = zeta_i^T G zc'_j (zeta_i = z_i - mu, zc' = the column latent's inducing inputs centred at their centroid):
//     T[alpha, beta] = sum_ij Omega_ij zeta_i^alpha zc'_j^beta,   |alpha| + |beta| <= 2,
// stored as (N0 | r1 [d] | R2 [d, d] | k1' [d] | K2' [d, d] | XC' [d, d]) = mma_pair_agg_len(d) doubles.  The polynomial
// part 1 + b + b^2/2 of e^b is exact from the moments of the two weight vectors (the forward's k_wmom_gemm tables, to
// degree 4):   sum_ij what_i what'_j zeta_i^alpha zc'_j^beta b_ij^n = < M~_{|alpha|+n}[alpha, .], G^{(x)n} Q_{|beta|+n}[beta, .] >,
// M~_k = sum_i what_i zeta_i^{(x)k} (binomial shift of the table's zc-moments by dmu = mu - zbar_a), Q_k = sum_j what'_j zc'_j^{(x)k};
// the remainder r(b) = e^b - 1 - b - b^2/2 is reduced by the tile kernel.  mma_pair_convert then re-centres the column side
// at mu (zeta'_j = zc'_j - dmu2, dmu2 = mu - zbar_a').
//   mR, mC: packed moments (graded colex, mm_mono.h) of the row / column weight vector, degree <= 4.
//   sm: mma_pair_poly_scratch(d) doubles.  T is ASSIGNED.
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int mma_pair_agg_len(int d) { return 1 + 2 * d + 3 * d * d; }
__host__ __device__ inline int mma_ipow(int b, int e) { int r = 1; for (int i = 0; i < e; ++i) r *= b; return r; }
__host__ __device__ inline int mma_pair_poly_scratch(int d) {
  int n = 0;
  for (int k = 0; k <= 4; ++k) n += mma_ipow(d, k);
  return n + mma_ipow(d, 4) + 8;
}

//   rtab: the packed model's rank table for DK = 8 (MMModelLayout::rtab; nullptr: ranks are computed).
MMA_FN void mma_pair_poly(Ctx c, int d, const double* G, const double* dmu, const double* mR, const double* mC, double* T,
                          double* sm, const short* rtab = nullptr) {
  const int lane = c.lane(), nl = c.nl();
  int moff[5], toff[6];
  toff[0] = 0;
  for (int k = 0; k <= 4; ++k) { moff[k] = mm_mono_off(k, d); toff[k + 1] = toff[k] + mma_ipow(d, k); }
  // rank of an (unsorted) index tuple of length k inside its degree block: table lookup (block k starts at (8^k - 8) / 7)
  auto rank_of = [&](const int (&ix)[4], int k) -> int {
    if (k == 0) return 0;
    if (rtab) {
      int pos = 0;
      for (int t = 0; t < k; ++t) pos |= ix[t] << (3 * t);
      return (int)rtab[(((1 << (3 * k)) - 8) / 7) + pos];
    }
    return mm_mono_rank_unsorted(ix[0], ix[1], ix[2], ix[3], k);
  };
  double* Mt = sm;                 // M~_k, k = 0..4, full tensors (index 0 most significant)
  double* Q = sm + toff[5];        // Q_t expanded, then G applied along its trailing indices
  // ---- M~_k[i_1..i_k] = sum_{S subset [k]} (-1)^{k - |S|} m_{|S|}[i_S] prod_{t not in S} dmu[i_t] ------------------------------
  // the tensors are symmetric: the shift is evaluated once per monomial (packed, into Q as scratch), then expanded
  const int nmono = mm_mono_off(5, d);
  for (int idx = lane; idx < nmono; idx += nl) {
    int k = 0;
    while (idx >= mm_mono_off(k + 1, d)) ++k;
    int ix[4] = {0, 0, 0, 0};
    mm_mono_unrank(idx - moff[k], k, ix);
    double acc = 0.0;
    for (int S = 0; S < (1 << k); ++S) {
      int sel[4] = {0, 0, 0, 0}, ns = 0;
      double pr = 1.0;
      for (int t = 0; t < k; ++t) {
        if (S & (1 << t)) sel[ns++] = ix[t];               // ix is sorted, so is every sub-tuple
        else pr *= -dmu[ix[t]];
      }
      acc = fma(pr, mR[moff[ns] + rank_of(sel, ns)], acc);
    }
    Q[idx] = acc;
  }
  c.sync();
  for (int idx = lane; idx < toff[5]; idx += nl) {
    int k = 0;
    while (idx >= toff[k + 1]) ++k;
    int rem = idx - toff[k], ix[4] = {0, 0, 0, 0};
    for (int t = k - 1; t >= 0; --t) { ix[t] = rem % d; rem /= d; }
    Mt[idx] = Q[moff[k] + rank_of(ix, k)];
  }
  c.sync();
  const int nT = mma_pair_agg_len(d);
  for (int idx = lane; idx < nT; idx += nl) T[idx] = 0.0;
  c.sync();
  // ---- total column order t = s + n: Q_t expanded; n = 0, 1, 2: G applied to the last n indices; s = t - n <= 2 ----------------
  const int oR1 = 1, oR2 = 1 + d, oK1 = 1 + d + d * d, oK2 = 1 + 2 * d + d * d, oXC = 1 + 2 * d + 2 * d * d;
  for (int t = 0; t <= 4; ++t) {
    const int dt = mma_ipow(d, t);
    for (int idx = lane; idx < dt; idx += nl) {
      int rem = idx, ix[4] = {0, 0, 0, 0};
      for (int u = t - 1; u >= 0; --u) { ix[u] = rem % d; rem /= d; }
      Q[idx] = mC[moff[t] + rank_of(ix, t)];
    }
    c.sync();
    for (int n = 0; n <= 2 && n <= t; ++n) {
      if (n > 0) {
        // apply G along index (t - n) (0-based from the left): fibre stride d^(n - 1); one lane owns a whole fibre
        const int st = mma_ipow(d, n - 1), nf = dt / d;
        for (int f = lane; f < nf; f += nl) {
          const int lo = f % st, hi = f / st, base = hi * st * d + lo;
          double v[8], o[8];
          for (int l = 0; l < d; ++l) v[l] = Q[base + l * st];
          for (int k = 0; k < d; ++k) {
            double a = 0.0;
            for (int l = 0; l < d; ++l) a = fma(G[k * d + l], v[l], a);
            o[k] = a;
          }
          for (int k = 0; k < d; ++k) Q[base + k * st] = o[k];
        }
        c.sync();
      }
      const int s = t - n;
      if (s > 2) continue;
      const double fac = n == 2 ? 0.5 : 1.0;
      const int dn = mma_ipow(d, n), ds = mma_ipow(d, s);
      // outputs (alpha, beta), |alpha| = ra <= 2 - s: += fac < M~_{ra + n}[alpha, .], Q[beta, .] >
      for (int ra = 0; ra + s <= 2; ++ra) {
        const int da = mma_ipow(d, ra), nout = da * ds;
        const double* Mk = Mt + toff[ra + n];
        const int obase = (ra == 0 && s == 0) ? 0 : (ra == 1 && s == 0) ? oR1 : (ra == 2) ? oR2 : (ra == 0 && s == 1) ? oK1
                          : (ra == 0 && s == 2) ? oK2 : oXC;
        for (int o = lane; o < nout; o += nl) {
          const int al = o / ds, be = o - al * ds;
          double a = 0.0;
          for (int k = 0; k < dn; ++k) a = fma(Mk[al * dn + k], Q[be * dn + k], a);
          T[obase + o] += fac * a;              // (alpha, beta) flat = alpha * d^s + beta: each output owned by one lane
        }
      }
      c.sync();
    }
  }
}

// T (zeta, zc') -> aggregates (zeta, zeta'), zeta'_j = zc'_j - dmu2; in place.  One sync inside, none at the end.
MMA_FN void mma_pair_convert(Ctx c, int d, const double* dmu2, double* T) {
  const int lane = c.lane(), nl = c.nl();
  const int oR1 = 1, oK1 = 1 + d + d * d, oK2 = 1 + 2 * d + d * d, oXC = 1 + 2 * d + 2 * d * d;
  const double N0 = T[0];
  for (int idx = lane; idx < d * d; idx += nl) {
    const int k = idx / d, l = idx - k * d;
    T[oK2 + idx] += -dmu2[k] * T[oK1 + l] - T[oK1 + k] * dmu2[l] + dmu2[k] * dmu2[l] * N0;
    T[oXC + idx] -= T[oR1 + k] * dmu2[l];
  }
  c.sync();
  for (int k = lane; k < d; k += nl) T[oK1 + k] -= dmu2[k] * N0;
}

// T (zc, .) -> (zeta, .), zeta_i = zc_i - dmu: the ROW side's counterpart of mma_pair_convert, for the polynomial part of an item
// whose bilinear form runs on rows centred at the centroid (mma_pair_poly called with a zero shift; mm_mono.h) while the
// aggregates' row monomials are those of zeta.  In place; one sync inside, none at the end.
MMA_FN void mma_pair_convert_rows(Ctx c, int d, const double* dmu, double* T) {
  const int lane = c.lane(), nl = c.nl();
  const int oR1 = 1, oR2 = 1 + d, oK1 = 1 + d + d * d, oXC = 1 + 2 * d + 2 * d * d;
  const double N0 = T[0];
  for (int idx = lane; idx < d * d; idx += nl) {
    const int k = idx / d, l = idx - k * d;
    T[oR2 + idx] += -dmu[k] * T[oR1 + l] - T[oR1 + k] * dmu[l] + dmu[k] * dmu[l] * N0;
    T[oXC + idx] -= dmu[k] * T[oK1 + l];
  }
  c.sync();
  for (int k = lane; k < d; k += nl) T[oR1 + k] -= dmu[k] * N0;
}

// ---------------------------------------------------------------------------------------------------------------------
// mma_gp_item_bwd: one (latent | kernel pair) item of d(f1, Sff, cross)/d(mu, Sigma) of a frozen model, from the M-sized
// sums of mm_backward_sums (csrc/mm_backward.hip).  Pair order: the L diagonal pairs, then a < a' row by row.
//   col  [P][3 + d][Mp]: Ksum_j, csum_j, cC_j, Usum_j[d]  (columns: latent a' of the pair)
//   row  [Po][2][Mp]   : Rsum_i, rsum_i                   (rows: latent a; diagonal pairs: = Ksum, csum)
//   latmat [L][2 d^2 + 2]: P_a = (Sigma + Lambda_a)^-1 (k_prep);  w, q [L][Mp];  Z [L][M][d];  ls2 [L][d].
// item < L: latent a = item;  item >= L: pair p = item - L.   out: gS_item [d, d], gmu_item [d] (ASSIGNED; the caller sums
// the items and symmetrises).  The formulas are those of autodiff.moment_match_backward (moment form).
// cbuf: M doubles private to the item (coefficient vector of a latent item);  sm: mma_gp_item_scratch(d, nl) doubles.
// pagg != nullptr (f32 models, csrc/mm_bwd_f32.hip): the off-diagonal pairs come as AGGREGATES instead of M-sized vectors,
//   pagg [Po][mma_pair_agg_len(d)] = sum_ij Omega_ij (1 | zeta_i | zeta_i zeta_i^T | zeta'_j | zeta'_j zeta'_j^T | zeta_i zeta'_j^T),
//   zeta = z - mu; `col` then holds the L diagonal pairs only and `row` is not read; f1raw [L] = sum_i w_i per latent.
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int mma_gp_ncol(int d) { return 1 + d + d * d; }
__host__ __device__ inline int mma_gp_nslice(int d, int nl) { const int n = nl / mma_gp_ncol(d); return n < 1 ? 1 : n; }
__host__ __device__ inline int mma_gp_item_scratch(int d, int nl) {
  const int dp = d + 1, nc = mma_gp_ncol(d), ns = mma_gp_nslice(d, nl);
  return 3 * ns * nc + 3 * nc + 10 * d * dp + 6 * d + 8;
}

__host__ __device__ inline void mma_decode_pair(int p, int L, int& a, int& a2) {
  if (p < L) { a = p; a2 = p; return; }
  int r = p - L, i = 0;
  while (r >= L - 1 - i) { r -= L - 1 - i; ++i; }
  a = i; a2 = i + 1 + r;
}

// raw moments sum_m coef[m] (1, z_m, z_m z_m^T) -> out [1 + d + d^2]
MMA_FN void mma_raw_moments(Ctx c, int M, int d, const double* Z, const double* coef, double* part, double* out) {
  const int lane = c.lane(), nl = c.nl(), nc = mma_gp_ncol(d), ns = mma_gp_nslice(d, nl);
  for (int idx = lane; idx < ns * nc; idx += nl) {
    const int k = idx % nc, s = idx / nc;
    const int i = k <= d ? k - 1 : (k - 1 - d) / d, j = k <= d ? 0 : (k - 1 - d) % d;
    // four independent chains: the loads of four centres are in flight together (a single chain is one global-memory
    // latency per centre on the device)
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    auto term = [&](int m) -> double {
      return k == 0 ? 1.0 : k <= d ? Z[(size_t)m * d + i] : Z[(size_t)m * d + i] * Z[(size_t)m * d + j];
    };
    int m = s;
    for (; m + 3 * ns < M; m += 4 * ns) {
      const double c0 = coef[m], c1 = coef[m + ns], c2 = coef[m + 2 * ns], c3 = coef[m + 3 * ns];
      const double f0 = term(m), f1 = term(m + ns), f2 = term(m + 2 * ns), f3 = term(m + 3 * ns);
      acc[0] = fma(c0, f0, acc[0]); acc[1] = fma(c1, f1, acc[1]); acc[2] = fma(c2, f2, acc[2]); acc[3] = fma(c3, f3, acc[3]);
    }
    for (; m < M; m += ns) acc[0] = fma(coef[m], term(m), acc[0]);
    part[idx] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  }
  c.sync();
  for (int k = lane; k < nc; k += nl) {
    double s = 0.0;
    for (int t = 0; t < ns; ++t) s += part[t * nc + k];
    out[k] = s;
  }
  c.sync();
}

// coefficient of centre m in the moment sums of latent item a: (g_f1 + e_m + sum_pairs g_p rsum/csum_m) w_m + 2 g_aa cC_m q_m
// (with aggregates the off-diagonal pairs' share is added to the moments instead: mma_gp_item_bwd)
template <class GP>
__host__ __device__ inline double mma_gp_latent_coef(int a, int m, int L, int Mp, int d, int P, bool with_unc, bool have_pagg,
                                                     const double* Za, const double* v0, double pvmu, double gfa, double gdiag,
                                                     GP gpair, const double* wa, const double* qa, const double* col,
                                                     const double* row) {
  double e = -pvmu;
  for (int k = 0; k < d; ++k) e = fma(Za[(size_t)m * d + k], v0[k], e);
  double pw = 2.0 * gdiag * col[((size_t)a * (3 + d) + 1) * Mp + m];         // diagonal pair: rsum = csum
  if (!have_pagg) {
    for (int p = L; p < P; ++p) {
      int r, s2; mma_decode_pair(p, L, r, s2);
      if (r == a) pw = fma(gpair(p), row[((size_t)(p - L) * 2 + 1) * Mp + m], pw);
      else if (s2 == a) pw = fma(gpair(p), col[((size_t)p * (3 + d) + 1) * Mp + m], pw);
    }
  }
  double cm = (gfa + e + pw) * wa[m];
  if (with_unc) cm = fma(2.0 * gdiag * col[((size_t)a * (3 + d) + 2) * Mp + m], qa[m], cm);
  return cm;
}

// Partial moment sums of one (latent | pair) item over the centres [m0, m1): what mma_gp_item_bwd otherwise forms in loops
// over all M centres (one workgroup per item: the slow part at M = 2000).  out [3 nc] (nc = 1 + d + d^2):
//   latent item: cmom | wmom | unused;     pair item (not aggregated): Rmom | Kmom | X [d, d], u [d].
// sm: mma_gp_moments_scratch(d, nl, m1 - m0) doubles.  Items that come as aggregates need no moments (not called).
__host__ __device__ inline int mma_gp_moments_scratch(int d, int nl, int n) {
  return (2 + d) * n + 2 * n * d + 3 * mma_gp_nslice(d, nl) * mma_gp_ncol(d) + d + 8;
}

MMA_FN void mma_gp_item_moments(Ctx c, int item, int m0, int m1, int L, int M, int Mp, int d, int P, bool with_unc,
                                const double* Z, const double* mu, const double* latmat, const double* w, const double* q,
                                const double* col, const double* row, const double* g_f1, const double* g_Sff, int full_cov,
                                const double* g_cross, bool have_pagg, double* out, double* sm) {
  const int lane = c.lane(), nl = c.nl(), nc = mma_gp_ncol(d), ns = mma_gp_nslice(d, nl), n = m1 - m0, lat = 2 * d * d + 2;
  double* cA = sm;                 // [n]
  double* cB = cA + n;             // [n]
  double* Uc = cB + n;             // [d][n]
  double* Zra = Uc + d * n;        // [n][d] the row side's inducing inputs
  double* Zcb = Zra + n * d;       // [n][d] the column side's
  double* part = Zcb + n * d;      // [3][ns][nc]
  double* v0 = part + 3 * ns * nc; // [d]
  auto gpair = [&](int p) -> double {
    if (!full_cov) return g_Sff[p];
    int a, a2; mma_decode_pair(p, L, a, a2);
    return p < L ? g_Sff[a * L + a] : g_Sff[a * L + a2] + g_Sff[a2 * L + a];
  };
  const bool latent = item < L;
  int a, b;
  if (latent) { a = item; b = item; } else mma_decode_pair(item - L, L, a, b);
  const double* Za = Z + (size_t)a * M * d; const double* Zb = Z + (size_t)b * M * d;
  for (int idx = lane; idx < n * d; idx += nl) { Zra[idx] = Za[(size_t)m0 * d + idx]; Zcb[idx] = Zb[(size_t)m0 * d + idx]; }
  if (latent) {
    const double* Pa = latmat + (size_t)a * lat;
    for (int k = lane; k < d; k += nl) {
      double s = 0.0;
      for (int l = 0; l < d; ++l) s = fma(Pa[k * d + l], g_cross[l * L + a], s);
      v0[k] = s;
    }
    c.sync();
    double pvmu = 0.0;
    for (int k = 0; k < d; ++k) pvmu = fma(v0[k], mu[k], pvmu);
    const double gfa = g_f1[a], gdiag = gpair(a);
    for (int i = lane; i < n; i += nl) {
      cA[i] = mma_gp_latent_coef(a, m0 + i, L, Mp, d, P, with_unc, have_pagg, Za, v0, pvmu, gfa, gdiag, gpair, w + (size_t)a * Mp,
                                 q + (size_t)a * Mp, col, row);
      cB[i] = w[(size_t)a * Mp + m0 + i];
    }
  } else {
    const int p = item - L;
    const double* colp = col + (size_t)p * (3 + d) * Mp;
    const double* Rsum = p < L ? colp : row + (size_t)(p - L) * 2 * Mp;
    for (int i = lane; i < n; i += nl) { cA[i] = Rsum[m0 + i]; cB[i] = colp[m0 + i]; }
    for (int idx = lane; idx < d * n; idx += nl) { const int k = idx / n, i = idx - k * n; Uc[idx] = colp[(size_t)(3 + k) * Mp + m0 + i]; }
  }
  c.sync();
  const int nq = d * d + d;
  const bool same_tab = a == b;                 // latent item or diagonal pair: one monomial per centre serves both sums
  const bool same_coef = !latent && a == b;     // diagonal pair: Rsum = Ksum (the pair is symmetric), Rmom = Kmom
  for (int idx = lane; idx < ns * nc; idx += nl) {
    const int k = idx % nc, sl = idx / nc;
    const int i = k <= d ? k - 1 : (k - 1 - d) / d, j = k <= d ? 0 : (k - 1 - d) % d;
    // X / u column k < nq: X[i2][j2] = sum U_i2 z'_j2 (k < d^2), u[i2] = sum U_i2
    const int i2 = k < d * d ? k / d : k - d * d, j2 = k < d * d ? k % d : 0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int m = sl; m < n; m += ns) {
      const double fa = k == 0 ? 1.0 : k <= d ? Zra[m * d + i] : Zra[m * d + i] * Zra[m * d + j];
      s0 = fma(cA[m], fa, s0);
      if (!same_coef) {
        const double fb = same_tab ? fa : (k == 0 ? 1.0 : k <= d ? Zcb[m * d + i] : Zcb[m * d + i] * Zcb[m * d + j]);
        s1 = fma(cB[m], fb, s1);
      }
      if (!latent && k < nq) s2 = fma(Uc[i2 * n + m], k < d * d ? Zcb[m * d + j2] : 1.0, s2);
    }
    part[idx] = s0; part[ns * nc + idx] = same_coef ? s0 : s1; part[2 * ns * nc + idx] = s2;
  }
  c.sync();
  for (int k = lane; k < 3 * nc; k += nl) {
    const int which = k / nc, kk = k - which * nc;
    double s = 0.0;
    for (int t = 0; t < ns; ++t) s += part[which * ns * nc + t * nc + kk];
    out[k] = s;
  }
  c.sync();
}

MMA_FN void mma_gp_item_bwd(Ctx c, int item, int L, int M, int Mp, int d, int P, bool with_unc,
                            const double* Z, const double* ls2, const double* mu, const double* Sigma, const double* latmat,
                            const double* w, const double* q, const double* col, const double* row,
                            const double* g_f1, const double* g_Sff, int full_cov, const double* g_cross,
                            double* gS_item, double* gmu_item, double* cbuf, double* sm, bool* ok,
                            const double* pagg = nullptr, const double* f1raw = nullptr,
                            const double* pre = nullptr, int nchunk = 0) {
  const int lane = c.lane(), nl = c.nl(), dp = d + 1, msz = d * dp, nc = mma_gp_ncol(d), ns = mma_gp_nslice(d, nl);
  const int Po = P - L, lat = 2 * d * d + 2, na = mma_pair_agg_len(d);
  double* part = sm;                        // [3][ns][nc]
  double* mom0 = part + 3 * ns * nc;        // [nc]
  double* mom1 = mom0 + nc;                 // [nc]
  double* mom2 = mom1 + nc;                 // [nc]  (X [d, d] and u [d] for pair items)
  double* A0 = mom2 + nc;                   // d x dp work matrices
  double* A1 = A0 + msz; double* A2 = A1 + msz; double* A3 = A2 + msz; double* A4 = A3 + msz; double* A5 = A4 + msz;
  double* A6 = A5 + msz; double* A7 = A6 + msz; double* A8 = A7 + msz; double* A9 = A8 + msz;
  double* v0 = A9 + msz; double* v1 = v0 + d; double* v2 = v1 + d; double* v3 = v2 + d; double* v4 = v3 + d; double* v5 = v4 + d;
  auto gpair = [&](int p) -> double {
    if (!full_cov) return g_Sff[p];
    int a, a2; mma_decode_pair(p, L, a, a2);
    return p < L ? g_Sff[a * L + a] : g_Sff[a * L + a2] + g_Sff[a2 * L + a];
  };
  // second moment centred at mu from raw moments: m2 - m1 mu^T - mu m1^T + m0 mu mu^T
  auto second = [&](const double* mm, int i, int j) -> double {
    return mm[1 + d + i * d + j] - mm[1 + i] * mu[j] - mu[i] * mm[1 + j] + mm[0] * mu[i] * mu[j];
  };
  // pre != nullptr: the moment sums come as nchunk partials [nchunk][3 nc] of mma_gp_item_moments
  auto from_partials = [&]() {
    for (int k = lane; k < 3 * nc; k += nl) {
      double sacc = 0.0;
      for (int ch = 0; ch < nchunk; ++ch) sacc += pre[(size_t)ch * 3 * nc + k];
      (k < nc ? mom0[k] : k < 2 * nc ? mom1[k - nc] : mom2[k - 2 * nc]) = sacc;
    }
    c.sync();
  };
  if (item < L) {
    // ---- latent a: F_a = c0 lognorm - 1/2 <Pa, C2(mu)> + v^T Pa s - w0 (Pa v)^T mu --------------------------------------
    const int a = item;
    const double* Pa = latmat + (size_t)a * lat;          // [d][d]
    const double* Za = Z + (size_t)a * M * d;
    const double* wa = w + (size_t)a * Mp;
    const double* qa = q + (size_t)a * Mp;
    for (int k = lane; k < d; k += nl) {                  // pv = Pa v,  v = g_cross[:, a]
      double s = 0.0;
      for (int l = 0; l < d; ++l) s = fma(Pa[k * d + l], g_cross[l * L + a], s);
      v0[k] = s;
    }
    c.sync();
    double pvmu = 0.0;
    for (int k = 0; k < d; ++k) pvmu = fma(v0[k], mu[k], pvmu);
    const double gfa = g_f1[a], gdiag = gpair(a);
    if (pre) {
      from_partials();
    } else {
      for (int m = lane; m < M; m += nl)
        cbuf[m] = mma_gp_latent_coef(a, m, L, Mp, d, P, with_unc, pagg != nullptr, Za, v0, pvmu, gfa, gdiag, gpair, wa, qa, col, row);
      c.sync();
      mma_raw_moments(c, M, d, Za, cbuf, part, mom0);       // cmom
      mma_raw_moments(c, M, d, Za, wa, part, mom1);         // wmom
    }
    const double w0 = mom1[0];
    // the off-diagonal pairs' share of the coefficient moments, from the aggregates: with E = e - 1,
    //   sum_ij w_i w'_j E_ij phi(zeta_i) = sum_ij Omega_ij phi(zeta_i) - (sum_i w_i phi(zeta_i)) (sum_j w'_j)
    // (mu-centred: dc0, dc1 [d] in v1, dC2 [d, d] in A4)
    double dc0 = 0.0;
    if (pagg) {
      for (int idx = lane; idx < d * d + d; idx += nl) {
        const bool isv = idx >= d * d;
        const int i = isv ? idx - d * d : idx / d, j = isv ? 0 : idx - i * d;
        double acc = 0.0;
        for (int p = L; p < P; ++p) {
          int r, s2; mma_decode_pair(p, L, r, s2);
          if (r != a && s2 != a) continue;
          const double* ag = pagg + (size_t)(p - L) * na;
          const double Wo = f1raw[r == a ? s2 : r];                          // sum of the partner latent's weights
          const double* v1p = r == a ? ag + 1 : ag + 1 + d + d * d;          // r1 | k1
          const double* m2p = r == a ? ag + 1 + d : ag + 1 + 2 * d + d * d;  // R2 | K2
          const double own = isv ? mom1[1 + i] - w0 * mu[i] : second(mom1, i, j);
          acc = fma(gpair(p), (isv ? v1p[i] : m2p[i * d + j]) - own * Wo, acc);
        }
        if (isv) v1[i] = acc; else A4[i * dp + j] = acc;
      }
      for (int p = L; p < P; ++p) {
        int r, s2; mma_decode_pair(p, L, r, s2);
        if (r != a && s2 != a) continue;
        dc0 = fma(gpair(p), pagg[(size_t)(p - L) * na] - w0 * f1raw[r == a ? s2 : r], dc0);
      }
      c.sync();
    }
    const double c0 = mom0[0] + dc0;
    for (int idx = lane; idx < d * d; idx += nl) {
      const int i = idx / d, j = idx - i * d;
      const double si = mom1[1 + i] - w0 * mu[i], sj = mom1[1 + j] - w0 * mu[j];       // s_det
      const double vi = g_cross[i * L + a], vj = g_cross[j * L + a];
      const double c2 = second(mom0, i, j) + (pagg ? A4[i * dp + j] : 0.0);
      A0[i * dp + j] = -0.5 * c2 + 0.5 * (vi * sj + si * vj);                           // Abar
      A1[i * dp + j] = Pa[i * d + j];
    }
    c.sync();
    mma_mm(c, d, d, d, A0, dp, A1, dp, A2, dp);
    c.sync();
    mma_mm(c, d, d, d, A1, dp, A2, dp, A3, dp);          // Pa Abar Pa
    c.sync();
    for (int idx = lane; idx < d * d; idx += nl) {
      const int i = idx / d, j = idx - i * d;
      gS_item[idx] = -A3[i * dp + j] - 0.5 * c0 * A1[i * dp + j];
    }
    for (int k = lane; k < d; k += nl) {
      double s = 0.0;
      for (int l = 0; l < d; ++l) s = fma(A1[k * dp + l], mom0[1 + l] - mom0[0] * mu[l] + (pagg ? v1[l] : 0.0), s);
      gmu_item[k] = s - w0 * v0[k];
    }
    c.sync();
    return;
  }
  // ---- pair p = (a, b): F_p = g [K0 const - 1/2 <Dr, R2> - 1/2 <Dc, K2> + <G, X - u mu^T> - mu^T G m1c] -------------------
  const int p = item - L;
  int a, b; mma_decode_pair(p, L, a, b);
  const double gp = gpair(p);
  const double* La = ls2 + (size_t)a * d; const double* Lb = ls2 + (size_t)b * d;
  const double* Za = Z + (size_t)a * M * d; const double* Zb = Z + (size_t)b * M * d;
  const double* colp = col + (size_t)p * (3 + d) * Mp;
  const double* Ksum = colp;
  const double* Rsum = p < L ? colp : row + (size_t)(p - L) * 2 * Mp;
  (void)Po;
  const bool agg = pagg != nullptr && p >= L;
  // mu-centred aggregates of the pair: R0 = K0 = sum Omega;  r1 (v1), k1 (v2), u (v3) [d];  R2 (A5), K2 (A6), XC (mom2) [d, d]
  if (agg) {
    const double* ag = pagg + (size_t)(p - L) * na;
    for (int idx = lane; idx < d * d; idx += nl) {
      const int i = idx / d, j = idx - i * d;
      A5[i * dp + j] = ag[1 + d + idx];
      A6[i * dp + j] = ag[1 + 2 * d + d * d + idx];
      mom2[idx] = ag[1 + 2 * d + 2 * d * d + idx];
    }
    for (int k = lane; k < d; k += nl) { v1[k] = ag[1 + k]; v2[k] = ag[1 + d + d * d + k]; v3[k] = ag[1 + k]; }
    if (lane == 0) { mom0[0] = ag[0]; mom1[0] = ag[0]; }
    c.sync();
  } else {
  if (pre) from_partials();
  else {
  mma_raw_moments(c, M, d, Za, Rsum, part, mom0);        // Rmom
  mma_raw_moments(c, M, d, Zb, Ksum, part, mom1);        // Kmom
  // X [d, d] = sum_j U_j z'_j^T, u [d] = sum_j U_j  (into mom2: X at [0, d^2), u at [d^2, d^2 + d))
  {
    const int nq = d * d + d, nsl = mma_gp_nslice(d, nl);
    for (int idx = lane; idx < nsl * nq; idx += nl) {
      const int k = idx % nq, s = idx / nq;
      const int i = k < d * d ? k / d : k - d * d, j = k < d * d ? k % d : 0;
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      auto term = [&](int m) -> double { return colp[(size_t)(3 + i) * Mp + m] * (k < d * d ? Zb[(size_t)m * d + j] : 1.0); };
      int m = s;
      for (; m + 3 * nsl < M; m += 4 * nsl) {
        const double t0 = term(m), t1 = term(m + nsl), t2 = term(m + 2 * nsl), t3 = term(m + 3 * nsl);
        acc[0] += t0; acc[1] += t1; acc[2] += t2; acc[3] += t3;
      }
      for (; m < M; m += nsl) acc[0] += term(m);
      part[idx] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    c.sync();
    for (int k = lane; k < nq; k += nl) {
      double s = 0.0;
      for (int t = 0; t < nsl; ++t) s += part[t * nq + k];
      mom2[k] = s;
    }
    c.sync();
  }
  }
  // centre at mu: XC = X - u mu^T (in place; u is copied out first)
  for (int k = lane; k < d; k += nl) {
    v1[k] = mom0[1 + k] - mom0[0] * mu[k];
    v2[k] = mom1[1 + k] - mom1[0] * mu[k];
    v3[k] = mom2[d * d + k];
  }
  for (int idx = lane; idx < d * d; idx += nl) {
    const int i = idx / d, j = idx - i * d;
    A5[i * dp + j] = second(mom0, i, j);
    A6[i * dp + j] = second(mom1, i, j);
  }
  c.sync();
  for (int idx = lane; idx < d * d; idx += nl) mom2[idx] -= v3[idx / d] * mu[idx % d];
  c.sync();
  }
  const double* XC = mom2; const double* u = v3;
  const double R0 = mom0[0], K0 = mom1[0];
  // Svi = (Sigma + V)^-1
  for (int idx = lane; idx < d * d; idx += nl) {
    const int i = idx / d, j = idx - i * d;
    const double s = i >= j ? Sigma[i * d + j] : Sigma[j * d + i];
    A0[i * dp + j] = s + (i == j ? La[i] * Lb[i] / (La[i] + Lb[i]) : 0.0);
    A1[i * dp + j] = latmat[(size_t)a * lat + i * d + j];              // Pa
    A2[i * dp + j] = latmat[(size_t)b * lat + i * d + j];              // Pb
  }
  c.sync();
  mma_spd_inverse(c, A0, A9, d, dp, ok);
  for (int idx = lane; idx < d * d; idx += nl) {
    const int i = idx / d, j = idx - i * d;
    const double Vi = La[i] * Lb[i] / (La[i] + Lb[i]), Vj = La[j] * Lb[j] / (La[j] + Lb[j]);
    const double Wm = Vi * A0[i * dp + j];                              // V Svi
    const double T = (i == j ? Vi : 0.0) - Wm * Vj;
    A3[i * dp + j] = Wm;
    A4[i * dp + j] = T;
    const double iab = 1.0 / (La[i] * Lb[j]), iaa = 1.0 / (La[i] * La[j]), ibb = 1.0 / (Lb[i] * Lb[j]);
    const double R2 = A5[i * dp + j], K2 = A6[i * dp + j];
    const double xg_ij = XC[i * d + j] * iab;
    const double xg_ji = XC[j * d + i] / (La[j] * Lb[i]);
    A7[i * dp + j] = 0.5 * (xg_ij + xg_ji) + 0.5 * R2 * iaa + 0.5 * K2 * ibb;     // Tbar
  }
  c.sync();
  // gS_p = Wm^T Tbar Wm - 1/2 Pa R2 Pa - 1/2 Pb K2 Pb + K0 (1/2 (Pa + Pb) - 1/2 Svi)
  mma_mm(c, d, d, d, A7, dp, A3, dp, A8, dp);            // Tbar Wm
  c.sync();
  for (int idx = lane; idx < d * d; idx += nl) {          // A7 <- Wm^T (Tbar Wm)
    const int i = idx / d, j = idx - i * d;
    double s = 0.0;
    for (int k = 0; k < d; ++k) s = fma(A3[k * dp + i], A8[k * dp + j], s);
    A9[i * dp + j] = s;
  }
  c.sync();
  mma_mm(c, d, d, d, A5, dp, A1, dp, A7, dp);            // R2 Pa
  mma_mm(c, d, d, d, A6, dp, A2, dp, A8, dp);            // K2 Pb
  c.sync();
  mma_mm(c, d, d, d, A1, dp, A7, dp, A5, dp);            // Pa R2 Pa
  mma_mm(c, d, d, d, A2, dp, A8, dp, A6, dp);            // Pb K2 Pb
  c.sync();
  for (int idx = lane; idx < d * d; idx += nl) {
    const int i = idx / d, j = idx - i * d;
    gS_item[idx] = gp * (A9[i * dp + j] - 0.5 * A5[i * dp + j] - 0.5 * A6[i * dp + j]
                         + K0 * (0.5 * (A1[i * dp + j] + A2[i * dp + j]) - 0.5 * A0[i * dp + j]));
  }
  // gmu_p = Dr (R1 - R0 mu) + Dc (K1 - K0 mu) - G^T u - G m1c,  Dr = Lam_a^-1 - Pa - T / (La La), G = T / (La Lb)
  for (int i = lane; i < d; i += nl) {
    double s = 0.0;
    for (int j = 0; j < d; ++j) {
      const double T = A4[i * dp + j];
      const double Dr = (i == j ? 1.0 / La[i] : 0.0) - A1[i * dp + j] - T / (La[i] * La[j]);
      const double Dc = (i == j ? 1.0 / Lb[i] : 0.0) - A2[i * dp + j] - T / (Lb[i] * Lb[j]);
      const double r1 = v1[j], k1 = v2[j];
      s += Dr * r1 + Dc * k1 - (A4[j * dp + i] / (La[j] * Lb[i])) * u[j] - (T / (La[i] * Lb[j])) * k1;
    }
    gmu_item[i] = gp * s;
  }
  c.sync();
  (void)v4; (void)v5; (void)R0;
}
