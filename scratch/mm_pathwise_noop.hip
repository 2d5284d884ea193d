// Pathwise (decoupled-sampling) GP evaluation for sample rollouts on gfx950 -- SURVEY.md row f-3,
// BASELINE.json configs[4].  Replaces the arithmetic of gpflow_sampling's predict_f_samples as the
// reference calls it (gpflow_pilco/models/svgp.py:124-130 via loops/pilco.py:263-298 and the
// tensor branch of forward_sde, dynamics/forward_sde.py:23-31, under Euler.step, solvers.py:50-65):
//
//   f_s,a(x_s) = scale_a sum_k w[s,a,k] cos(2 pi (omega_t[a,:,k] . x_s + phase[a,k]))
//              + var_a   sum_m v[s,a,m] exp2(zs_t[a,:,m] . xs - hz[a,m] - hx)  + mean_a
//   with omega_t = omega / 2 pi (revolutions: v_cos_f32's native unit), xs = x_s * x_scale_a,
//   x_scale = sqrt(log2 e) / ls_a, zs_t = z_m * x_scale_a, hz = |zs|^2 / 2, hx = |xs|^2 / 2, so that
//   the SE-ARD kernel exp(-|x - z|^2 / (2 ls^2)) is a bare v_exp_f32.  Shared operands are stored
//   k-major ([d][K], [d][M]) so that a wave's 16-byte loads are contiguous.
//
// Every (sample, latent) owns K + M weights that are used exactly once per step: the kernel is a
// weight stream, HBM-bound (S L (K+M) sizeof(T) bytes per step; C5 per GPU: 0.8 GB).  The weights
// arrive as ONE blocked stream wb[g][a][tb][sl][BT] (g = group of NS samples, tb = block of BT terms:
// K/BT prior blocks then M/BT update blocks, sl = sample in the group): one pass of a wave reads
// NS * BT contiguous elements (4 KB) and consecutive passes are consecutive in memory.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../gpflowpilco_amd/csrc/mm_common.h"

#define MM_PW_NS 4      // samples per workgroup (and per block of the weight stream)
#ifndef PW_COS          // overridable for ablation builds (scratch/): which part of the kernel costs what
#define PW_COS(x_) pw_cos(x_)
#define PW_EXP(x_) pw_exp(x_)
#endif

template <typename T> struct PwVec;
template <> struct PwVec<float> { typedef float4 type; static constexpr int W = 4; };
template <> struct PwVec<double> { typedef double2 type; static constexpr int W = 2; };

// cos(2 pi x) and 2^x: the transcendental unit in f32 (v_cos_f32 takes revolutions), ocml in f64
__device__ __forceinline__ float pw_cos(float x) { return __builtin_amdgcn_cosf(x - floorf(x)); }
__device__ __forceinline__ double pw_cos(double x) { return cospi(2.0 * x); }
__device__ __forceinline__ float pw_exp(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ double pw_exp(double x) { return exp2(x); }

template <typename T>
__device__ __forceinline__ void pw_unpack(const float4& v, T (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <typename T>
__device__ __forceinline__ void pw_unpack(const double2& v, T (&o)[2]) { o[0] = v.x; o[1] = v.y; }

// grid: ceil(S / NS) workgroups of 256 threads = 4 waves.  Wave wv of a workgroup owns the latents
// wv, wv + 4, ... of the workgroup's NS samples and streams their K + M weights alone: the only
// cross-lane step is one shuffle reduction per (latent, sample) -- no LDS, no barriers.  The weight
// loads are software-pipelined two passes ahead (PMC: 71 % of wave cycles were s_waitcnt with one
// pass in flight).  EULER != 0: x_out = x + dt f (needs d == L), else f_out = f.
template <typename T, int DK>
__global__ __launch_bounds__(256) void k_pathwise(int S, int L, int M, int K, int d,
                                                  const T* __restrict__ x,        // [S,d]
                                                  const T* __restrict__ omega,    // [L,d,K] revolutions
                                                  const T* __restrict__ phase,    // [L,K]   revolutions
                                                  const T* __restrict__ zs,       // [L,d,M] scaled
                                                  const T* __restrict__ hz,       // [L,M]
                                                  const double* __restrict__ xscale,  // [L,d]
                                                  const double* __restrict__ pscale,  // [L]
                                                  const double* __restrict__ var,     // [L]
                                                  const double* __restrict__ meanc,   // [L] or null
                                                  const T* __restrict__ wb,       // blocked weights
                                                  T* __restrict__ out,            // [S,L] (f or x_next)
                                                  T* __restrict__ traj,           // optional [S,L]
                                                  int euler, double dt) {
  typedef typename PwVec<T>::type VT;
  constexpr int W = PwVec<T>::W, NS = MM_PW_NS, BT = 64 * W;     // BT terms per pass of a wave
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = blockIdx.x, s0 = g * NS;
  const int nbK = K / BT, nbM = M / BT, NB = nbK + nbM;
  // inputs of the NS samples (clamped index for the ragged tail; those results are not stored)
  T xr[NS][DK];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int row = (s0 + s < S) ? s0 + s : S - 1;
#pragma unroll
    for (int k = 0; k < DK; ++k) xr[s][k] = (k < d) ? x[(size_t)row * d + k] : (T)0;
  }

  for (int a = wv; a < L; a += 4) {
    T accp[NS], accu[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { accp[s] = (T)0; accu[s] = (T)0; }

    // ---- prior part: random Fourier features ------------------------------------------------
    {
      const T* om = omega + (size_t)a * K * d;
      const T* ph = phase + (size_t)a * K;
      const T* wrow = wb + (((size_t)g * L + a) * NB) * NS * BT + lane * W;
      VT wq[3][NS];
#pragma unroll
      for (int pf = 0; pf < 2; ++pf)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          wq[pf][s] = (pf < nbK) ? *reinterpret_cast<const VT*>(wrow + ((size_t)pf * NS + s) * BT) : VT{};
      for (int pidx = 0; pidx < nbK; ++pidx) {
        const int k0 = lane * W + pidx * BT;
#pragma unroll
        for (int s = 0; s < NS; ++s)
          wq[2][s] = (pidx + 2 < nbK) ? *reinterpret_cast<const VT*>(wrow + ((size_t)(pidx + 2) * NS + s) * BT) : VT{};
        T wv4[NS][W];
#pragma unroll
        for (int s = 0; s < NS; ++s) pw_unpack<T>(wq[0][s], wv4[s]);
        T cv[DK][W], bv[W];
#pragma unroll
        for (int k = 0; k < DK; ++k) {
          if (k < d) { for (int j = 0; j < W; ++j) cv[k][j] = (T)(0.001 * k); }
          else {
#pragma unroll
            for (int j = 0; j < W; ++j) cv[k][j] = (T)0;
          }
        }
        pw_unpack<T>(*reinterpret_cast<const VT*>(ph + k0), bv);
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            T arg = bv[j];
#pragma unroll
            for (int k = 0; k < DK; ++k) arg += cv[k][j] * xr[s][k];
            accp[s] += wv4[s][j] * PW_COS(arg);
          }
#pragma unroll
        for (int s = 0; s < NS; ++s) { wq[0][s] = wq[1][s]; wq[1][s] = wq[2][s]; }
      }
    }
    // ---- update part: kernel basis at the inducing points -----------------------------------
    {
      T xsc[NS][DK], hx[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        T h = (T)0;
#pragma unroll
        for (int k = 0; k < DK; ++k) {
          const T sc = (k < d) ? (T)xscale[a * d + k] : (T)0;
          xsc[s][k] = xr[s][k] * sc; h += xsc[s][k] * xsc[s][k];
        }
        hx[s] = (T)0.5 * h;
      }
      const T* zz = zs + (size_t)a * M * d;
      const T* hh = hz + (size_t)a * M;
      const T* vrow = wb + (((size_t)g * L + a) * NB + nbK) * NS * BT + lane * W;
      VT vq[3][NS];
#pragma unroll
      for (int pf = 0; pf < 2; ++pf)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          vq[pf][s] = (pf < nbM) ? *reinterpret_cast<const VT*>(vrow + ((size_t)pf * NS + s) * BT) : VT{};
      for (int pidx = 0; pidx < nbM; ++pidx) {
        const int m0 = lane * W + pidx * BT;
#pragma unroll
        for (int s = 0; s < NS; ++s)
          vq[2][s] = (pidx + 2 < nbM) ? *reinterpret_cast<const VT*>(vrow + ((size_t)(pidx + 2) * NS + s) * BT) : VT{};
        T vv[NS][W];
#pragma unroll
        for (int s = 0; s < NS; ++s) pw_unpack<T>(vq[0][s], vv[s]);
        T cv[DK][W], hv[W];
#pragma unroll
        for (int k = 0; k < DK; ++k) {
          if (k < d) { for (int j = 0; j < W; ++j) cv[k][j] = (T)(0.001 * k); }
          else {
#pragma unroll
            for (int j = 0; j < W; ++j) cv[k][j] = (T)0;
          }
        }
        pw_unpack<T>(*reinterpret_cast<const VT*>(hh + m0), hv);
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            T arg = -hv[j] - hx[s];
#pragma unroll
            for (int k = 0; k < DK; ++k) arg += cv[k][j] * xsc[s][k];
            accu[s] += vv[s][j] * PW_EXP(arg);
          }
#pragma unroll
        for (int s = 0; s < NS; ++s) { vq[0][s] = vq[1][s]; vq[1][s] = vq[2][s]; }
      }
    }
    // ---- wave reduction (f64), one value per sample -----------------------------------------
    const double ps = pscale[a], vr = var[a], mc = meanc ? meanc[a] : 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      double t = ps * (double)accp[s] + vr * (double)accu[s];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
      if (lane == 0 && s0 + s < S) {
        double f = t + mc;
        if (euler) f = (double)xr[s][a < DK ? a : 0] + dt * f;
        out[(size_t)(s0 + s) * L + a] = (T)f;
        if (traj) traj[(size_t)(s0 + s) * L + a] = (T)f;
      }
    }
  }
}

template <typename T>
static int pw_launch(int S, int L, int M, int K, int d, const T* x, const T* omega, const T* phase, const T* zs,
                     const T* hz, const double* xscale, const double* pscale, const double* var,
                     const double* meanc, const T* wb, T* out, T* traj, int euler, double dt, hipStream_t s) {
  dim3 grid((S + MM_PW_NS - 1) / MM_PW_NS);
#define PW_LAUNCH(DK_) hipLaunchKernelGGL((k_pathwise<T, DK_>), grid, dim3(256), 0, s, S, L, M, K, d, x, omega, phase, \
                                          zs, hz, xscale, pscale, var, meanc, wb, out, traj, euler, dt)
  if (d <= 4) PW_LAUNCH(4);
  else if (d <= 8) PW_LAUNCH(8);
  else if (d <= 16) PW_LAUNCH(16);
  else PW_LAUNCH(32);
#undef PW_LAUNCH
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

static int pw_check(int S, int L, int M, int K, int d, int dtype) {
  if (S <= 0 || L <= 0 || M <= 0 || K <= 0 || d <= 0) return MM_E_ARG;
  if (d > MM_DMAX) return MM_E_DIM;
  if (dtype != MM_F32 && dtype != MM_F64) return MM_E_DTYPE;
  const int BT = 64 * (dtype == MM_F64 ? 2 : 4);
  if (K % BT || M % BT) return MM_E_DIM;        // the host pads K and M with zero weights
  return 0;
}

extern "C" int mm_pathwise_eval(int S, int L, int M, int K, int d, int dtype,
                                const void* x, const void* omega_t, const void* phase, const void* zs_t,
                                const void* hz, const double* x_scale, const double* prior_scale,
                                const double* variance, const double* mean_c, const void* wb, void* f_out,
                                void* stream) {
  int rc = pw_check(S, L, M, K, d, dtype);
  if (rc) return rc;
  if (!x || !omega_t || !phase || !zs_t || !hz || !x_scale || !prior_scale || !variance || !wb || !f_out)
    return MM_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MM_F64)
    return pw_launch<double>(S, L, M, K, d, (const double*)x, (const double*)omega_t, (const double*)phase,
                             (const double*)zs_t, (const double*)hz, x_scale, prior_scale, variance, mean_c,
                             (const double*)wb, (double*)f_out, nullptr, 0, 0.0, s);
  return pw_launch<float>(S, L, M, K, d, (const float*)x, (const float*)omega_t, (const float*)phase,
                          (const float*)zs_t, (const float*)hz, x_scale, prior_scale, variance, mean_c,
                          (const float*)wb, (float*)f_out, nullptr, 0, 0.0, s);
}

extern "C" int mm_pathwise_rollout(int S, int L, int M, int K, int d, int dtype, int H, double dt,
                                   void* x, void* x_tmp, const void* omega_t, const void* phase, const void* zs_t,
                                   const void* hz, const double* x_scale, const double* prior_scale,
                                   const double* variance, const double* mean_c, const void* wb,
                                   void* traj, void* stream) {
  int rc = pw_check(S, L, M, K, d, dtype);
  if (rc) return rc;
  if (H <= 0 || !x || !x_tmp || !omega_t || !phase || !zs_t || !hz || !x_scale || !prior_scale || !variance || !wb)
    return MM_E_ARG;
  if (d != L) return MM_E_STATE;
  hipStream_t s = (hipStream_t)stream;
  const size_t es = mm_elem_size(dtype), stride = (size_t)S * d * es;
  char* cur = (char*)x; char* nxt = (char*)x_tmp;
  for (int h = 0; h < H; ++h) {
    char* tr = traj ? (char*)traj + (size_t)h * stride : nullptr;
    if (dtype == MM_F64)
      rc = pw_launch<double>(S, L, M, K, d, (const double*)cur, (const double*)omega_t, (const double*)phase,
                             (const double*)zs_t, (const double*)hz, x_scale, prior_scale, variance, mean_c,
                             (const double*)wb, (double*)nxt, (double*)tr, 1, dt, s);
    else
      rc = pw_launch<float>(S, L, M, K, d, (const float*)cur, (const float*)omega_t, (const float*)phase,
                            (const float*)zs_t, (const float*)hz, x_scale, prior_scale, variance, mean_c,
                            (const float*)wb, (float*)nxt, (float*)tr, 1, dt, s);
    if (rc) return rc;
    char* t = cur; cur = nxt; nxt = t;
  }
  if (cur != (char*)x) {   // odd number of steps: the result sits in x_tmp
    hipError_t e = hipMemcpyAsync(x, cur, stride, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}
