// Pathwise (decoupled-sampling) GP evaluation for sample rollouts on gfx950 -- SURVEY.md row f-3,
// BASELINE.json configs[4].  Replaces the arithmetic of gpflow_sampling's predict_f_samples as the
// reference calls it (gpflow_pilco/models/svgp.py:124-130 via loops/pilco.py:263-298 and the
// tensor branch of forward_sde, dynamics/forward_sde.py:23-31, under Euler.step, solvers.py:50-65):
//
//   f_s,a(x_s) = scale_a sum_k w[s,a,k] cos(omega[a,k] . x_s + phase[a,k])
//              + var_a   sum_m v[s,a,m] exp(zs[a,m] . xs - hz[a,m] - hx)  + mean_a
//   with xs = x_s / ls_a, zs = z_m / ls_a, hz = |zs|^2 / 2, hx = |xs|^2 / 2      (SE-ARD kernel)
//
// Every (sample, latent) owns K + M weights that are used exactly once per step: the kernel is a
// weight stream, HBM-bound (S L (K+M) sizeof(T) bytes per step; C5 per GPU: 0.8 GB).  A workgroup
// owns NS consecutive samples so that the shared operands (omega, zs: L2-resident) are fetched once
// per NS weight streams; weights are read with 16-byte loads, coalesced along k / m.
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"

#define MM_PW_NS 4      // samples per workgroup

template <typename T> struct PwVec;
template <> struct PwVec<float> { typedef float4 type; static constexpr int W = 4; };
template <> struct PwVec<double> { typedef double2 type; static constexpr int W = 2; };

__device__ __forceinline__ float pw_cos(float x) { return cosf(x); }
__device__ __forceinline__ double pw_cos(double x) { return cos(x); }
__device__ __forceinline__ float pw_exp(float x) { return __expf(x); }
__device__ __forceinline__ double pw_exp(double x) { return exp(x); }

template <typename T>
__device__ __forceinline__ void pw_unpack(const float4& v, T (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <typename T>
__device__ __forceinline__ void pw_unpack(const double2& v, T (&o)[2]) { o[0] = v.x; o[1] = v.y; }

// grid: ceil(S / NS) workgroups of 256 threads.  K and M must be multiples of the vector width
// (the host pads); EULER != 0: x_out = x + dt f (needs d == L), else f_out = f.
template <typename T, int DK>
__global__ __launch_bounds__(256) void k_pathwise(int S, int L, int M, int K, int d,
                                                  const T* __restrict__ x,        // [S,d]
                                                  const T* __restrict__ omega,    // [L,K,d]
                                                  const T* __restrict__ phase,    // [L,K]
                                                  const T* __restrict__ zs,       // [L,M,d]
                                                  const T* __restrict__ hz,       // [L,M]
                                                  const double* __restrict__ ls,  // [L,d]
                                                  const double* __restrict__ pscale,  // [L]
                                                  const double* __restrict__ var,     // [L]
                                                  const double* __restrict__ meanc,   // [L] or null
                                                  const T* __restrict__ w,        // [S,L,K]
                                                  const T* __restrict__ v,        // [S,L,M]
                                                  T* __restrict__ out,            // [S,L] (f or x_next)
                                                  T* __restrict__ traj,           // optional [S,L]
                                                  int euler, double dt) {
  typedef typename PwVec<T>::type VT;
  constexpr int W = PwVec<T>::W, NS = MM_PW_NS;
  const int s0 = blockIdx.x * NS, tid = threadIdx.x;
  __shared__ double red[NS][4];
  __shared__ T xsh[NS][MM_DMAX];
  for (int i = tid; i < NS * d; i += 256) {
    const int s = i / d, k = i - s * d;
    xsh[s][k] = (s0 + s < S) ? x[(size_t)(s0 + s) * d + k] : (T)0;
  }
  __syncthreads();

  for (int a = 0; a < L; ++a) {
    // per-sample scaled inputs for this latent
    T xr[NS][DK], xsc[NS][DK], hx[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      T h = (T)0;
#pragma unroll
      for (int k = 0; k < DK; ++k) {
        const T xv = (k < d) ? xsh[s][k] : (T)0;
        const T sc = (k < d) ? (T)(1.0 / ls[a * d + k]) : (T)0;
        xr[s][k] = xv; xsc[s][k] = xv * sc; h += xsc[s][k] * xsc[s][k];
      }
      hx[s] = (T)0.5 * h;
    }
    T accp[NS], accu[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { accp[s] = (T)0; accu[s] = (T)0; }

    // ---- prior part: random Fourier features ------------------------------------------------
    const T* om = omega + (size_t)a * K * d;
    const T* ph = phase + (size_t)a * K;
    for (int k0 = tid * W; k0 < K; k0 += 256 * W) {
      T wv[NS][W];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int ss = (s0 + s < S) ? s0 + s : S - 1;
        pw_unpack<T>(*reinterpret_cast<const VT*>(w + ((size_t)ss * L + a) * K + k0), wv[s]);
      }
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const T* c = om + (size_t)(k0 + j) * d;
        T cv[DK];
#pragma unroll
        for (int k = 0; k < DK; ++k) cv[k] = (k < d) ? c[k] : (T)0;
        const T b = ph[k0 + j];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          T arg = b;
#pragma unroll
          for (int k = 0; k < DK; ++k) arg += cv[k] * xr[s][k];
          accp[s] += wv[s][j] * pw_cos(arg);
        }
      }
    }
    // ---- update part: kernel basis at the inducing points -----------------------------------
    const T* zz = zs + (size_t)a * M * d;
    const T* hh = hz + (size_t)a * M;
    for (int m0 = tid * W; m0 < M; m0 += 256 * W) {
      T vv[NS][W];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int ss = (s0 + s < S) ? s0 + s : S - 1;
        pw_unpack<T>(*reinterpret_cast<const VT*>(v + ((size_t)ss * L + a) * M + m0), vv[s]);
      }
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const T* c = zz + (size_t)(m0 + j) * d;
        T cv[DK];
#pragma unroll
        for (int k = 0; k < DK; ++k) cv[k] = (k < d) ? c[k] : (T)0;
        const T hzv = hh[m0 + j];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          T arg = -hzv - hx[s];
#pragma unroll
          for (int k = 0; k < DK; ++k) arg += cv[k] * xsc[s][k];
          accu[s] += vv[s][j] * pw_exp(arg);
        }
      }
    }
    // ---- workgroup reduction (f64), one value per sample ------------------------------------
    const double ps = pscale[a], vr = var[a], mc = meanc ? meanc[a] : 0.0;
    double tot[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      double t = ps * (double)accp[s] + vr * (double)accu[s];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
      tot[s] = t;
    }
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
      for (int s = 0; s < NS; ++s) red[s][tid >> 6] = tot[s];
    }
    __syncthreads();
    if (tid < NS && s0 + tid < S) {
      double f = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3] + mc;
      if (euler) f = (double)xsh[tid][a] + dt * f;
      out[(size_t)(s0 + tid) * L + a] = (T)f;
      if (traj) traj[(size_t)(s0 + tid) * L + a] = (T)f;
    }
  }
}

template <typename T>
static int pw_launch(int S, int L, int M, int K, int d, const T* x, const T* omega, const T* phase, const T* zs,
                     const T* hz, const double* ls, const double* pscale, const double* var, const double* meanc,
                     const T* w, const T* v, T* out, T* traj, int euler, double dt, hipStream_t s) {
  dim3 grid((S + MM_PW_NS - 1) / MM_PW_NS);
#define PW_LAUNCH(DK_) hipLaunchKernelGGL((k_pathwise<T, DK_>), grid, dim3(256), 0, s, S, L, M, K, d, x, omega, phase, \
                                          zs, hz, ls, pscale, var, meanc, w, v, out, traj, euler, dt)
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
  const int W = dtype == MM_F64 ? 2 : 4;
  if (K % W || M % W) return MM_E_DIM;          // the host pads K and M with zero weights
  return 0;
}

extern "C" int mm_pathwise_eval(int S, int L, int M, int K, int d, int dtype,
                                const void* x, const void* omega, const void* phase, const void* zs, const void* hz,
                                const double* ls, const double* prior_scale, const double* variance,
                                const double* mean_c, const void* w, const void* v, void* f_out, void* stream) {
  int rc = pw_check(S, L, M, K, d, dtype);
  if (rc) return rc;
  if (!x || !omega || !phase || !zs || !hz || !ls || !prior_scale || !variance || !w || !v || !f_out) return MM_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MM_F64)
    return pw_launch<double>(S, L, M, K, d, (const double*)x, (const double*)omega, (const double*)phase,
                             (const double*)zs, (const double*)hz, ls, prior_scale, variance, mean_c,
                             (const double*)w, (const double*)v, (double*)f_out, nullptr, 0, 0.0, s);
  return pw_launch<float>(S, L, M, K, d, (const float*)x, (const float*)omega, (const float*)phase,
                          (const float*)zs, (const float*)hz, ls, prior_scale, variance, mean_c,
                          (const float*)w, (const float*)v, (float*)f_out, nullptr, 0, 0.0, s);
}

extern "C" int mm_pathwise_rollout(int S, int L, int M, int K, int d, int dtype, int H, double dt,
                                   void* x, void* x_tmp, const void* omega, const void* phase, const void* zs,
                                   const void* hz, const double* ls, const double* prior_scale,
                                   const double* variance, const double* mean_c, const void* w, const void* v,
                                   void* traj, void* stream) {
  int rc = pw_check(S, L, M, K, d, dtype);
  if (rc) return rc;
  if (H <= 0 || !x || !x_tmp || !omega || !phase || !zs || !hz || !ls || !prior_scale || !variance || !w || !v)
    return MM_E_ARG;
  if (d != L) return MM_E_STATE;
  hipStream_t s = (hipStream_t)stream;
  const size_t es = mm_elem_size(dtype), stride = (size_t)S * d * es;
  char* cur = (char*)x; char* nxt = (char*)x_tmp;
  for (int h = 0; h < H; ++h) {
    char* tr = traj ? (char*)traj + (size_t)h * stride : nullptr;
    if (dtype == MM_F64)
      rc = pw_launch<double>(S, L, M, K, d, (const double*)cur, (const double*)omega, (const double*)phase,
                             (const double*)zs, (const double*)hz, ls, prior_scale, variance, mean_c,
                             (const double*)w, (const double*)v, (double*)nxt, (double*)tr, 1, dt, s);
    else
      rc = pw_launch<float>(S, L, M, K, d, (const float*)cur, (const float*)omega, (const float*)phase,
                            (const float*)zs, (const float*)hz, ls, prior_scale, variance, mean_c,
                            (const float*)w, (const float*)v, (float*)nxt, (float*)tr, 1, dt, s);
    if (rc) return rc;
    char* t = cur; cur = nxt; nxt = t;
  }
  if (cur != (char*)x) {   // odd number of steps: the result sits in x_tmp
    hipError_t e = hipMemcpyAsync(x, cur, stride, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}
