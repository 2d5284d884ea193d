// Pathwise POLICY rollout and its reverse sweep on gfx950 -- SURVEY.md row f-3 completed (rounds 1-3 had the drift
// evaluation and a drift-only Euler fold).
//
// What the reference's only caller of sample paths runs and differentiates (PathwisePILCO._policy_loss_closure,
// gpflow_pilco/loops/pilco.py:263-298; the tensor branch of forward_sde, dynamics/forward_sde.py:23-31; Euler.step,
// dynamics/solvers.py:50-65; the mean over samples and the gradient tape: examples/cartpole_swingup/train_utils.py:108-135).
// Per sample path s and step h:
//     e  = encoder(x_h)              TrigonometricEncoder on tensors: [sin a, cos a, x_inactive]   (components.py:44-75)
//     u  = scale (Phi(f_pol(e)) + shift),  f_pol(e) = sum_m beta_m k(e, z_m) + c                   (models/core.py:60-71: the
//                                          KernelRegressor's predictive MEAN through Chain[Scale, Shift, NormalCDF])
//     x_{h+1} = x_h + dt f_s([e, u])     f_s: the s-th drift sample path (mm_pathwise.hip: a weight stream, HBM-bound)
//     cost[h][s] = -exp(-(enc(x_{h+1}) - t)^T W (enc(x_{h+1}) - t) / 2)                              (components.py:39-41)
// Forward: per step one small kernel (k_pw_head: Euler update of the previous step, cost, encoder, policy -> the drift's
// input) and one stream pass; the taped forward makes the stream pass also emit d f_s / d d_s [S][nx][nd] (its JAC variant).
// Backward: samples are independent, so the WHOLE reverse sweep is one kernel (k_pw_policy_bwd: thread = sample, loop over
// the steps backwards, the state's adjoint in registers) that reads the tape -- states, drift inputs, Jacobians: no second
// pass over the weight stream -- and accumulates the packed policy's gradient per wave (fixed-order wave sums -> per-wave
// slabs -> k_pw_grad_sum: deterministic).  All small algebra in f64 whatever the paths' element type.
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"
#include "mm_compose.h"

int mm_pathwise_launch(int S, int L, int M, int K, int d, int dtype, const void* x, const void* omega_t, const void* phase,
                       const void* zs_t, const void* hz, const double* x_scale, const double* prior_scale, const double* variance,
                       const double* mean_c, const void* wb, void* f_out, void* jac, hipStream_t s);

#define MMP_NE 24            // largest encoded dimension (2 na + nb, na <= 8, nx <= 16)
#define MMP_POLICY_MMAX 256

struct MMPwTapeLayout {
  size_t x;      // [H + 1][S][nx] T   states
  size_t din;    // [H][S][nd] T       drift inputs (e_h, u_h)
  size_t f;      // [S][nx] T          the current step's drift sample (scratch)
  size_t jac;    // [H][S][nx][nd] T   d f / d d per step (0 bytes when not differentiating)
  size_t total;
};
static inline MMPwTapeLayout mm_pw_tape_layout(int S, int H, int nx, int na, int dtype, int with_jac) {
  MMPwTapeLayout o;
  const size_t es = mm_elem_size(dtype), A = 256;
  const int nd = nx + na + 1;
  size_t off = 0;
  o.x = off;   off = mm_align_up(off + (size_t)(H + 1) * S * nx * es, A);
  o.din = off; off = mm_align_up(off + (size_t)H * S * nd * es, A);
  o.f = off;   off = mm_align_up(off + (size_t)S * nx * es, A);
  o.jac = off; off = mm_align_up(off + (with_jac ? (size_t)H * S * nx * nd * es : 0), A);
  o.total = off;
  return o;
}

__device__ __forceinline__ void mmp_encode(const MMComposeDims& D, const double* x, double* e) {
  for (int i = 0; i < D.na; ++i) { double sn, cs; sincos(x[D.active[i]], &sn, &cs); e[i] = sn; e[D.na + i] = cs; }
  for (int i = 0; i < D.nb; ++i) e[2 * D.na + i] = x[D.inactive[i]];
}
// adjoint of the encoder: ge [ne] -> gx [nx] (ACCUMULATED)
__device__ __forceinline__ void mmp_encode_bwd(const MMComposeDims& D, const double* x, const double* ge, double* gx) {
  for (int i = 0; i < D.na; ++i) {
    double sn, cs;
    sincos(x[D.active[i]], &sn, &cs);
    gx[D.active[i]] += cs * ge[i] - sn * ge[D.na + i];
  }
  for (int i = 0; i < D.nb; ++i) gx[D.inactive[i]] += ge[2 * D.na + i];
}
// cost = -exp(-err^T W err / 2) of an encoded state; gq != NULL: also d cost / d e (W need not be symmetric)
__device__ __forceinline__ double mmp_cost(int ne, const double* e, const double* target, const double* precis, double* gq) {
  double err[MMP_NE], q = 0.0;
  for (int i = 0; i < ne; ++i) err[i] = e[i] - target[i];
  for (int i = 0; i < ne; ++i) {
    double r = 0.0;
    for (int j = 0; j < ne; ++j) r = fma(precis[i * ne + j], err[j], r);
    q = fma(err[i], r, q);
  }
  const double c = -exp(-0.5 * q);
  if (gq) {
    for (int i = 0; i < ne; ++i) {
      double r = 0.0;
      for (int j = 0; j < ne; ++j) r = fma(precis[i * ne + j] + precis[j * ne + i], err[j], r);
      gq[i] = -0.5 * c * r;                                  // d c = -c/2 dq,  dq = err^T (W + W^T) de
    }
  }
  return c;
}

// policy block in LDS: Z [M][ne] | beta [M] | 1 / ls2 [ne]; var, mean in registers
struct MMPwPolicy { const double* Z; const double* beta; const double* ils2; double var, mean; int M; };

__device__ __forceinline__ double mmp_policy_mean(int ne, const MMPwPolicy& P, const double* e) {
  double f = P.mean;
  for (int m = 0; m < P.M; ++m) {
    double r2 = 0.0;
    for (int k = 0; k < ne; ++k) { const double t = e[k] - P.Z[m * ne + k]; r2 = fma(t * t, P.ils2[k], r2); }
    f = fma(P.beta[m], P.var * exp(-0.5 * r2), f);
  }
  return f;
}
__device__ __forceinline__ double mmp_ndtr(double x) { return 0.5 * erfc(-x * 0.7071067811865476); }

template <typename T>
__device__ __forceinline__ void mmp_stage_policy(const double* Zg, const double* bg, const double* ls2g, int M, int ne, double* sm) {
  for (int i = threadIdx.x; i < M * ne; i += blockDim.x) sm[i] = Zg[i];
  for (int i = threadIdx.x; i < M; i += blockDim.x) sm[M * ne + i] = bg[i];
  for (int i = threadIdx.x; i < ne; i += blockDim.x) sm[M * ne + M + i] = 1.0 / ls2g[i];
  __syncthreads();
}

// k_pw_head: grid ceil(S / 256), thread = sample.  h in [0, H]:
//   h > 0: x_h = x_{h-1} + dt f_{h-1} -> tape; cost[h-1][s] of its encoding;    h < H: the drift input (e_h, u_h) -> tape.
template <typename T>
__global__ __launch_bounds__(256) void k_pw_head(MMComposeDims D, int S, int h, int H, double dt, const T* __restrict__ xprev,
                                                 const T* __restrict__ f, T* __restrict__ xcur, T* __restrict__ din,
                                                 T* __restrict__ cost, const T* __restrict__ target, const T* __restrict__ precis,
                                                 const double* __restrict__ pZ, const double* __restrict__ pbeta,
                                                 const double* __restrict__ pls2, const double* __restrict__ pvar,
                                                 const double* __restrict__ pmean, int pM, double scale, double shift) {
  extern __shared__ double sm[];
  const int nx = D.nx, ne = D.ne, nd = D.nd;
  double* pol = sm;                                        // M ne + M + ne
  double* tg = pol + pM * ne + pM + ne;                    // [ne]
  double* W = tg + ne;                                     // [ne][ne]
  for (int i = threadIdx.x; i < ne; i += 256) tg[i] = (double)target[i];
  for (int i = threadIdx.x; i < ne * ne; i += 256) W[i] = (double)precis[i];
  mmp_stage_policy<T>(pZ, pbeta, pls2, pM, ne, pol);
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  double x[MMC_NX], e[MMP_NE];
  if (h > 0) {
    for (int i = 0; i < nx; ++i) {
      x[i] = (double)xprev[(size_t)s * nx + i] + dt * (double)f[(size_t)s * nx + i];   // Euler.step, solvers.py:50-65
      xcur[(size_t)s * nx + i] = (T)x[i];
    }
  } else {
    for (int i = 0; i < nx; ++i) x[i] = (double)xcur[(size_t)s * nx + i];
  }
  // (the state every later step reads is the STORED one: rounded to T once)
  for (int i = 0; i < nx; ++i) x[i] = (double)(T)x[i];
  mmp_encode(D, x, e);
  if (h > 0) cost[(size_t)(h - 1) * S + s] = (T)mmp_cost(ne, e, tg, W, nullptr);
  if (h < H) {
    const MMPwPolicy P{pol, pol + pM * ne, pol + pM * ne + pM, pvar[0], pmean[0], pM};
    const double u = scale * (mmp_ndtr(mmp_policy_mean(ne, P, e)) + shift);
    for (int i = 0; i < ne; ++i) din[(size_t)s * nd + i] = (T)e[i];
    din[(size_t)s * nd + ne] = (T)u;
  }
}

__device__ __forceinline__ double mmp_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// k_pw_policy_bwd: grid ceil(S / 256), thread = sample, all H steps backwards.  gpart [nwaves][npar] (ASSIGNED): per wave the
// sum over its 64 samples and all steps of the packed policy's gradient (dZ [M][ne], dbeta [M], dls2 [ne], dvar, dmean);
// g_x0 [S][nx] (optional).  g_cost [H][S] f64.
template <typename T>
__global__ __launch_bounds__(256) void k_pw_policy_bwd(MMComposeDims D, int S, int H, double dt, const T* __restrict__ xs,
                                                       const T* __restrict__ dins, const T* __restrict__ jacs,
                                                       const double* __restrict__ g_cost, const T* __restrict__ target,
                                                       const T* __restrict__ precis, const double* __restrict__ pZ,
                                                       const double* __restrict__ pbeta, const double* __restrict__ pls2,
                                                       const double* __restrict__ pvar, const double* __restrict__ pmean, int pM,
                                                       double scale, double shift, double* __restrict__ gpart,
                                                       double* __restrict__ g_x0) {
  extern __shared__ double sm[];
  const int nx = D.nx, ne = D.ne, nd = D.nd, npar = pM * ne + pM + ne + 2;
  double* pol = sm;
  double* tg = pol + pM * ne + pM + ne;
  double* W = tg + ne;
  double* acc = W + ne * ne;                               // [4 waves][npar]
  for (int i = threadIdx.x; i < ne; i += 256) tg[i] = (double)target[i];
  for (int i = threadIdx.x; i < ne * ne; i += 256) W[i] = (double)precis[i];
  for (int i = threadIdx.x; i < 4 * npar; i += 256) acc[i] = 0.0;
  mmp_stage_policy<T>(pZ, pbeta, pls2, pM, ne, pol);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* wacc = acc + wv * npar;
  const int s = blockIdx.x * 256 + threadIdx.x;
  const bool live = s < S;
  const int sr = live ? s : S - 1;                         // (idle lanes recompute the last sample with zero adjoints)
  const MMPwPolicy P{pol, pol + pM * ne, pol + pM * ne + pM, pvar[0], pmean[0], pM};
  double gx[MMC_NX];
  for (int i = 0; i < nx; ++i) gx[i] = 0.0;
  for (int h = H - 1; h >= 0; --h) {
    double x1[MMC_NX], e1[MMP_NE], ge[MMP_NE];
    // adjoint of x_{h+1}: what later steps left in gx, plus this step's cost of its encoding
    for (int i = 0; i < nx; ++i) x1[i] = (double)xs[((size_t)(h + 1) * S + sr) * nx + i];
    mmp_encode(D, x1, e1);
    mmp_cost(ne, e1, tg, W, ge);
    const double gc = live ? g_cost[(size_t)h * S + sr] : 0.0;
    for (int i = 0; i < ne; ++i) ge[i] *= gc;
    mmp_encode_bwd(D, x1, ge, gx);
    // x_{h+1} = x_h + dt f(d_h):  g d = dt J^T g x_{h+1}
    double gd[MMP_NE + 1], e[MMP_NE];
    const T* J = jacs + ((size_t)h * S + sr) * (size_t)nx * nd;
    for (int k = 0; k < nd; ++k) {
      double r = 0.0;
      for (int i = 0; i < nx; ++i) r = fma((double)J[i * nd + k], gx[i], r);
      gd[k] = dt * r;
    }
    const T* dn = dins + ((size_t)h * S + sr) * nd;
    for (int i = 0; i < ne; ++i) e[i] = (double)dn[i];
    // policy head: u = scale (Phi(fp) + shift);  g fp = g u scale phi(fp)
    const double fp = mmp_policy_mean(ne, P, e);
    const double gfp = gd[ne] * scale * 0.3989422804014327 * exp(-0.5 * fp * fp);
    // policy mean fp = sum_m beta_m var exp(-sum_k (e_k - z_mk)^2 / (2 ls2_k)) + c: parameters (wave sums) and input
    double gvar = 0.0, gls2[MMP_NE];
    for (int k = 0; k < ne; ++k) { ge[k] = gd[k]; gls2[k] = 0.0; }
    for (int m = 0; m < pM; ++m) {
      double r2 = 0.0, df[MMP_NE];
      for (int k = 0; k < ne; ++k) { df[k] = e[k] - P.Z[m * ne + k]; r2 = fma(df[k] * df[k], P.ils2[k], r2); }
      const double km = exp(-0.5 * r2);                    // k_m / var
      const double t = gfp * P.beta[m] * P.var * km;       // g fp * beta_m k_m
      gvar = fma(gfp * P.beta[m], km, gvar);
      const double sb = mmp_wave_sum(gfp * P.var * km);    // d / d beta_m
      if (lane == 0) wacc[pM * ne + m] += sb;
      for (int k = 0; k < ne; ++k) {
        const double tz = t * df[k] * P.ils2[k];           // d k_m / d z_mk = k_m (e_k - z_mk) / ls2_k = - d k_m / d e_k
        ge[k] -= tz;
        gls2[k] = fma(0.5 * tz * df[k], P.ils2[k], gls2[k]);   // d k_m / d ls2_k = k_m (e_k - z_mk)^2 / (2 ls2_k^2)
        const double sz = mmp_wave_sum(tz);
        if (lane == 0) wacc[m * ne + k] += sz;
      }
    }
    for (int k = 0; k < ne; ++k) {
      const double sl = mmp_wave_sum(gls2[k]);
      if (lane == 0) wacc[pM * ne + pM + k] += sl;
    }
    {
      const double sv = mmp_wave_sum(gvar), sc = mmp_wave_sum(gfp);
      if (lane == 0) { wacc[pM * ne + pM + ne] += sv; wacc[pM * ne + pM + ne + 1] += sc; }
    }
    // d_h = (enc(x_h), u): adjoint of x_h = that of x_{h+1} (identity part of the Euler step) + the encoder's
    double x0[MMC_NX];
    for (int i = 0; i < nx; ++i) x0[i] = (double)xs[((size_t)h * S + sr) * nx + i];
    mmp_encode_bwd(D, x0, ge, gx);
  }
  if (g_x0 && live) for (int i = 0; i < nx; ++i) g_x0[(size_t)s * nx + i] = gx[i];
  __syncthreads();
  double* o = gpart + ((size_t)blockIdx.x * 4) * npar;
  for (int i = threadIdx.x; i < 4 * npar; i += 256) o[i] = acc[i];
}

// fixed-order sum of the per-wave slabs -> g_policy [npar]
__global__ __launch_bounds__(256) void k_pw_grad_sum(const double* __restrict__ gpart, int nslab, int npar, double* __restrict__ g_policy) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= npar) return;
  double sacc = 0.0;
  for (int w = 0; w < nslab; ++w) sacc += gpart[(size_t)w * npar + p];
  g_policy[p] = sacc;
}

static int mmp_check(int S, int M, int K, int dtype, int H, int nx, int na, const int32_t* active_dims, int policy_M, MMComposeDims& D) {
  if (S <= 0 || M <= 0 || K <= 0 || H <= 0 || policy_M <= 0) return MM_E_ARG;
  if (dtype != MM_F32 && dtype != MM_F64) return MM_E_DTYPE;
  const int rc = mm_compose_dims(nx, na, active_dims, D);
  if (rc) return rc;
  if (D.nd > 8 || policy_M > MMP_POLICY_MMAX) return MM_E_DIM;   // the Jacobian pass of the weight stream: nd <= 8
  return 0;
}
static inline size_t mmp_head_lds(int pM, int ne) { return (size_t)(pM * ne + pM + ne + ne + ne * ne + 8) * sizeof(double); }

extern "C" size_t mm_pathwise_tape_bytes(int S, int H, int nx, int na, int dtype, int with_jacobians) {
  if (S <= 0 || H <= 0 || nx <= 0 || nx > MMC_NX || na <= 0 || na > MMC_NA || na > nx) return 0;
  return mm_pw_tape_layout(S, H, nx, na, dtype, with_jacobians).total;
}

template <typename T>
static int mmp_rollout_t(const MMComposeDims& D, int S, int M, int K, int dtype, int H, double dt, const void* omega_t,
                         const void* phase, const void* zs_t, const void* hz, const double* x_scale, const double* prior_scale,
                         const double* variance, const double* mean_c, const void* wb, const char* pp, const MMModelLayout& pl,
                         int policy_M, double scale, double shift, const T* target, const T* precis, const T* x0, T* cost,
                         char* tape, const MMPwTapeLayout& tl, hipStream_t s) {
  const int nx = D.nx, ne = D.ne, nd = D.nd;
  T* xs = (T*)(tape + tl.x); T* dins = (T*)(tape + tl.din); T* f = (T*)(tape + tl.f);
  T* jac = tl.jac != tl.total ? (T*)(tape + tl.jac) : nullptr;
  hipError_t e = hipMemcpyAsync(xs, x0, (size_t)S * nx * sizeof(T), hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return (int)e;
  const size_t lds = mmp_head_lds(policy_M, ne);
  const dim3 grid((S + 255) / 256);
  for (int h = 0; h <= H; ++h) {
    hipLaunchKernelGGL((k_pw_head<T>), grid, dim3(256), lds, s, D, S, h, H, dt, h > 0 ? xs + (size_t)(h - 1) * S * nx : (const T*)nullptr,
                       (const T*)f, xs + (size_t)h * S * nx, h < H ? dins + (size_t)h * S * nd : (T*)nullptr, cost, target, precis,
                       (const double*)(pp + pl.Z64), (const double*)(pp + pl.beta64), (const double*)(pp + pl.ls2),
                       (const double*)(pp + pl.var), (const double*)(pp + pl.meanc), policy_M, scale, shift);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (h == H) break;
    const int rc = mm_pathwise_launch(S, nx, M, K, nd, dtype, dins + (size_t)h * S * nd, omega_t, phase, zs_t, hz, x_scale,
                                      prior_scale, variance, mean_c, wb, f, jac ? jac + (size_t)h * S * nx * nd : nullptr, s);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int mm_pathwise_policy_rollout(int S, int M, int K, int dtype, int H, double dt, int nx, int na,
                                          const int32_t* active_dims, const void* omega_t, const void* phase, const void* zs_t,
                                          const void* hz, const double* x_scale, const double* prior_scale,
                                          const double* variance, const double* mean_c, const void* wb,
                                          const void* policy_packed, size_t policy_bytes, int policy_M, double head_scale,
                                          double head_shift, const void* target, const void* precis, const void* x0, void* cost,
                                          void* tape, size_t tape_bytes, int with_jacobians, void* stream) {
  MMComposeDims D;
  int rc = mmp_check(S, M, K, dtype, H, nx, na, active_dims, policy_M, D);
  if (rc) return rc;
  if (!omega_t || !phase || !zs_t || !hz || !x_scale || !prior_scale || !variance || !wb || !policy_packed || !target || !precis ||
      !x0 || !cost || !tape) return MM_E_ARG;
  const MMPwTapeLayout tl = mm_pw_tape_layout(S, H, nx, na, dtype, with_jacobians);
  if (tape_bytes < tl.total) return MM_E_WORKSPACE;
  const MMModelLayout pl = mm_model_layout(1, policy_M, D.ne, MM_F64, 1);   // the f64 blocks precede the T blocks in every pack
  if (policy_bytes < pl.Zc64) return MM_E_WORKSPACE;
  const char* pp = (const char*)policy_packed;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MM_F64)
    return mmp_rollout_t<double>(D, S, M, K, dtype, H, dt, omega_t, phase, zs_t, hz, x_scale, prior_scale, variance, mean_c, wb, pp,
                                 pl, policy_M, head_scale, head_shift, (const double*)target, (const double*)precis,
                                 (const double*)x0, (double*)cost, (char*)tape, tl, s);
  return mmp_rollout_t<float>(D, S, M, K, dtype, H, dt, omega_t, phase, zs_t, hz, x_scale, prior_scale, variance, mean_c, wb, pp, pl,
                              policy_M, head_scale, head_shift, (const float*)target, (const float*)precis, (const float*)x0,
                              (float*)cost, (char*)tape, tl, s);
}

extern "C" size_t mm_pathwise_backward_scratch_bytes(int S, int policy_M, int ne) {
  if (S <= 0 || policy_M <= 0 || ne <= 0) return 0;
  return (size_t)((S + 255) / 256) * 4 * (size_t)(policy_M * ne + policy_M + ne + 2) * sizeof(double);
}

extern "C" int mm_pathwise_policy_rollout_backward(int S, int dtype, int H, double dt, int nx, int na, const int32_t* active_dims,
                                                   const void* policy_packed, size_t policy_bytes, int policy_M,
                                                   double head_scale, double head_shift, const void* target, const void* precis,
                                                   const void* tape, size_t tape_bytes, const void* g_cost, void* g_policy,
                                                   void* g_x0, void* scratch, size_t scratch_bytes, void* stream) {
  MMComposeDims D;
  int rc = mmp_check(S, 1, 1, dtype, H, nx, na, active_dims, policy_M, D);
  if (rc) return rc;
  if (!policy_packed || !target || !precis || !tape || !g_cost || !g_policy || !scratch) return MM_E_ARG;
  const MMPwTapeLayout tl = mm_pw_tape_layout(S, H, nx, na, dtype, 1);
  if (tape_bytes < tl.total) return MM_E_WORKSPACE;
  const int ne = D.ne, npar = policy_M * ne + policy_M + ne + 2;
  if (scratch_bytes < mm_pathwise_backward_scratch_bytes(S, policy_M, ne)) return MM_E_WORKSPACE;
  const MMModelLayout pl = mm_model_layout(1, policy_M, ne, MM_F64, 1);
  if (policy_bytes < pl.Zc64) return MM_E_WORKSPACE;
  const char* pp = (const char*)policy_packed; const char* tp = (const char*)tape;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = mmp_head_lds(policy_M, ne) + (size_t)4 * npar * sizeof(double);
  if (lds > 160 * 1024) return MM_E_DIM;
  const dim3 grid((S + 255) / 256);
#define MMP_BWD(T_)                                                                                                              \
  do {                                                                                                                          \
    if (lds > 64 * 1024) {                                                                                                      \
      hipError_t ea = hipFuncSetAttribute((const void*)k_pw_policy_bwd<T_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (ea != hipSuccess) return (int)ea;                                                                                     \
    }                                                                                                                           \
    hipLaunchKernelGGL((k_pw_policy_bwd<T_>), grid, dim3(256), lds, s, D, S, H, dt, (const T_*)(tp + tl.x), (const T_*)(tp + tl.din), \
                       (const T_*)(tp + tl.jac), (const double*)g_cost, (const T_*)target, (const T_*)precis,                    \
                       (const double*)(pp + pl.Z64), (const double*)(pp + pl.beta64), (const double*)(pp + pl.ls2),              \
                       (const double*)(pp + pl.var), (const double*)(pp + pl.meanc), policy_M, head_scale, head_shift,           \
                       (double*)scratch, (double*)g_x0);                                                                        \
  } while (0)
  if (dtype == MM_F64) MMP_BWD(double); else MMP_BWD(float);
#undef MMP_BWD
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_pw_grad_sum, dim3((npar + 255) / 256), dim3(256), 0, s, (const double*)scratch, (int)grid.x * 4, npar,
                     (double*)g_policy);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
