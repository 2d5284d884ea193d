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
#include <atomic>
#include "mm_common.h"

#define MM_PW_NS 4      // samples per workgroup (and per block of the weight stream)
#ifndef PW_RING
#define PW_RING 3       // register sets of the weight stream: PW_RING - 1 blocks in flight per wave
#endif
#ifndef PW_LDS_WAVES
#define PW_LDS_WAVES 8   // waves per workgroup of the LDS-resident kernel (one workgroup per CU)
#endif
#ifndef PW_COS          // overridable for ablation builds (scratch/): which part of the kernel costs what
#define PW_COS(x_) pw_cos(x_)
#define PW_EXP(x_) pw_exp(x_)
#endif

typedef float pwf2 __attribute__((ext_vector_type(2)));
#define PW_B2(x_) ((pwf2){(float)(x_), (float)(x_)})
template <typename T> struct PwVec;
template <> struct PwVec<float> { typedef float4 type; static constexpr int W = 4; };
template <> struct PwVec<double> { typedef double2 type; static constexpr int W = 2; };

// cos(2 pi x) and 2^x: the transcendental unit in f32 (v_cos_f32 takes revolutions), ocml in f64
__device__ __forceinline__ float pw_cos(float x) { return __builtin_amdgcn_cosf(x - floorf(x)); }
__device__ __forceinline__ double pw_cos(double x) { return cospi(2.0 * x); }
__device__ __forceinline__ float pw_exp(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ double pw_exp(double x) { return exp2(x); }
// (cos, sin)(2 pi x) of one argument (the Jacobian pass needs both): one range reduction, two v_*_f32
__device__ __forceinline__ void pw_sincos(float x, float& c, float& s) {
  const float r = x - floorf(x);
  c = __builtin_amdgcn_cosf(r); s = __builtin_amdgcn_sinf(r);
}
__device__ __forceinline__ void pw_sincos(double x, double& c, double& s) { sincospi(2.0 * x, &s, &c); }

template <typename T>
__device__ __forceinline__ void pw_unpack(const float4& v, T (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <typename T>
__device__ __forceinline__ void pw_unpack(const double2& v, T (&o)[2]) { o[0] = v.x; o[1] = v.y; }

// Sum over the 64 lanes of a wave by DPP row operations (no LDS crossbar round trips): result valid
// in lane 63.  Fixed order => bitwise reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double pw_dpp_add(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)u, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)(u >> 32), CTRL, ROW_MASK, 0xf, false);
  const unsigned long long w = ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;
  return v + __builtin_bit_cast(double, w);          // lanes outside ROW_MASK / without a source add +0.0
}
__device__ __forceinline__ double pw_wave_sum63(double v) {
  v = pw_dpp_add<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
  v = pw_dpp_add<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
  v = pw_dpp_add<0x141, 0xf>(v);     // row_half_mirror
  v = pw_dpp_add<0x140, 0xf>(v);     // row_mirror: every lane of a row holds the row sum
  v = pw_dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
  v = pw_dpp_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return v;
}

// the same reduction in f32 (the Jacobian entries of an f32 path: 1 + d values per (latent, sample) -- in f64 the reductions
// alone would cost a third of the stream pass)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float pw_dpp_addf(float v) {
  const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, o);
}
__device__ __forceinline__ float pw_wave_sum63(float v) {
  v = pw_dpp_addf<0xB1, 0xf>(v);
  v = pw_dpp_addf<0x4E, 0xf>(v);
  v = pw_dpp_addf<0x141, 0xf>(v);
  v = pw_dpp_addf<0x140, 0xf>(v);
  v = pw_dpp_addf<0x142, 0xa>(v);
  v = pw_dpp_addf<0x143, 0xc>(v);
  return v;
}

// JAC (both kernels): the same pass also emits the per-sample Jacobian d f_{s,a} / d x_s [S][L][d] -- what the reverse sweep of
// a differentiated sample rollout needs (the reference differentiates the pathwise loss with a gradient tape:
// examples/cartpole_swingup/train_utils.py:108-135 through loops/pilco.py:263-298):
//   prior  : d/dx_k w cos(2 pi arg) = -2 pi omega_k w sin(2 pi arg)       -> accJ_k += (w sin) omega_k, factor -2 pi scale
//   update : d/dx_k v 2^arg = ln2 v 2^arg (c_k - xscale_k^2 x_k), c = z xscale^2  -> accJ_k += (v 2^arg) c_k, then
//            ln2 var (accJ_k - xscale_k^2 x_k accu)
// d more FMAs and (prior blocks) one more transcendental per term and sample; 1 + d wave sums per (latent, sample).  The pass
// is then VALU-bound, not HBM-bound: 1.41 x the plain pass at the C5 shape (0.50 of the HBM peak against 0.70); the same pass in
// packed f32 over term pairs (parity-split accumulators, 256 VGPRs + scratch, 1300 register moves) ran 1.9 x: not kept.
// grid: ceil(S / NS) workgroups of 256 threads = 4 waves.  Wave wv of a workgroup owns the latents
// wv, wv + 4, ... of the workgroup's NS samples and streams their K + M weights alone: the only
// cross-lane step is one shuffle reduction per (latent, sample) -- no LDS, no barriers.  The weight
// loads are software-pipelined two passes ahead (PMC: 71 % of wave cycles were s_waitcnt with one
// pass in flight).  EULER != 0: x_out = x + dt f (needs d == L), else f_out = f.
// CND: the pass also accumulates the ABSOLUTE terms, abs[s,a] = scale_a sum_k |w cos| + var_a sum_m |v 2^arg| -- what the rounding of
// a T-typed weight stream and T-typed basis values does to f[s,a] is within ~2 eps_T of it (mm_pathwise_eval_bound).
template <typename T, int DK, bool JAC, bool CND>
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
                                                  T* __restrict__ jac,            // JAC: [S,L,d]
                                                  T* __restrict__ cnd,            // CND: [S,L]
                                                  int euler, double dt) {
  typedef typename PwVec<T>::type VT;
  constexpr int W = PwVec<T>::W, NS = MM_PW_NS, BT = 64 * W;     // BT terms per pass of a wave
  constexpr int NJ = JAC ? DK : 1;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = blockIdx.x, s0 = g * NS;
  const int nbK = K / BT, nbM = M / BT, NB = nbK + nbM;
  // inputs of the NS samples (clamped index for the ragged tail; those results are not stored)
  T xr[NS][DK];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int row = (s0 + s < S) ? s0 + s : S - 1;
#pragma unroll
    for (int k = 0; k < DK; ++k) {
      const T v = x[(size_t)row * d + (k < d ? k : 0)];      // unconditional load, then select
      xr[s][k] = (k < d) ? v : (T)0;
    }
  }

  for (int a = wv; a < L; a += 4) {
    T accp[NS], accu[NS], accJ[NS][NJ], absp[NS], absu[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      accp[s] = (T)0; accu[s] = (T)0; absp[s] = (T)0; absu[s] = (T)0;
#pragma unroll
      for (int k = 0; k < NJ; ++k) accJ[s][k] = (T)0;
    }

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
          wq[pf][s] = *reinterpret_cast<const VT*>(wrow + ((size_t)(pf < nbK ? pf : nbK - 1) * NS + s) * BT);
      for (int pidx = 0; pidx < nbK; ++pidx) {
        const int k0 = lane * W + pidx * BT;
#pragma unroll
        for (int s = 0; s < NS; ++s)
          wq[2][s] = *reinterpret_cast<const VT*>(wrow + ((size_t)(pidx + 2 < nbK ? pidx + 2 : nbK - 1) * NS + s) * BT);
        T wv4[NS][W];
#pragma unroll
        for (int s = 0; s < NS; ++s) pw_unpack<T>(wq[0][s], wv4[s]);
        T cv[DK][W], bv[W];
#pragma unroll
        for (int k = 0; k < DK; ++k) {
          pw_unpack<T>(*reinterpret_cast<const VT*>(om + (size_t)(k < d ? k : 0) * K + k0), cv[k]);
          if (k >= d) {
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
            if constexpr (JAC) {
              T cs, sn;
              pw_sincos(arg, cs, sn);
              accp[s] += wv4[s][j] * cs;
              const T t = wv4[s][j] * sn;
#pragma unroll
              for (int k = 0; k < DK; ++k) accJ[s][k] += t * cv[k][j];
            } else if constexpr (CND) {
              const T c = PW_COS(arg);
              accp[s] += wv4[s][j] * c;                       // (the plain pass's own expression: the value stays bit-equal)
              absp[s] += fabs(wv4[s][j]) * fabs(c);
            } else {
              accp[s] += wv4[s][j] * PW_COS(arg);
            }
          }
#pragma unroll
        for (int s = 0; s < NS; ++s) { wq[0][s] = wq[1][s]; wq[1][s] = wq[2][s]; }
      }
      if constexpr (JAC) {
        // ONE accumulator set for both halves: the prior sums are rescaled so that the update half's final factor
        // ln2 var xscale_k applies to the total (no second [NS][DK] register set)
#pragma unroll
        for (int k = 0; k < DK; ++k) {
          const double scv = xscale[a * d + (k < d ? k : 0)];
          const double den = 0.6931471805599453 * var[a] * scv;
          const T fj = (k < d && den != 0.0) ? (T)(-6.283185307179586 * pscale[a] / den) : (T)0;
#pragma unroll
          for (int s = 0; s < NS; ++s) accJ[s][k] *= fj;
        }
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
          const T scv = (T)xscale[a * d + (k < d ? k : 0)];
          const T sc = (k < d) ? scv : (T)0;
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
          vq[pf][s] = *reinterpret_cast<const VT*>(vrow + ((size_t)(pf < nbM ? pf : nbM - 1) * NS + s) * BT);
      for (int pidx = 0; pidx < nbM; ++pidx) {
        const int m0 = lane * W + pidx * BT;
#pragma unroll
        for (int s = 0; s < NS; ++s)
          vq[2][s] = *reinterpret_cast<const VT*>(vrow + ((size_t)(pidx + 2 < nbM ? pidx + 2 : nbM - 1) * NS + s) * BT);
        T vv[NS][W];
#pragma unroll
        for (int s = 0; s < NS; ++s) pw_unpack<T>(vq[0][s], vv[s]);
        T cv[DK][W], hv[W];
#pragma unroll
        for (int k = 0; k < DK; ++k) {
          pw_unpack<T>(*reinterpret_cast<const VT*>(zz + (size_t)(k < d ? k : 0) * M + m0), cv[k]);
          if (k >= d) {
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
            const T e = PW_EXP(arg);
            const T t = vv[s][j] * e;
            accu[s] += t;
            if constexpr (CND) absu[s] += fabs(vv[s][j]) * e;
            if constexpr (JAC) {
#pragma unroll
              for (int k = 0; k < DK; ++k) accJ[s][k] += t * cv[k][j];
            }
          }
#pragma unroll
        for (int s = 0; s < NS; ++s) { vq[0][s] = vq[1][s]; vq[1][s] = vq[2][s]; }
      }
      if constexpr (JAC) {
        // arg = zs . xs - hz - hx with xs = x xscale: d arg / d x_k = xscale_k (zs_k - xs_k)
        const T fj = (T)(0.6931471805599453 * var[a]);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int k = 0; k < DK; ++k) {
            const T scv = (T)xscale[a * d + (k < d ? k : 0)];
            const T sc = (k < d) ? scv : (T)0;
            accJ[s][k] = fj * sc * (accJ[s][k] - xsc[s][k] * accu[s]);
          }
      }
    }
    // ---- wave reduction (f64), one value per sample -----------------------------------------
    const double ps = pscale[a], vr = var[a], mc = meanc ? meanc[a] : 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const double t = pw_wave_sum63(ps * (double)accp[s] + vr * (double)accu[s]);
      if (lane == 63 && s0 + s < S) {
        double f = t + mc;
        if (euler) f = (double)x[(size_t)(s0 + s) * d + a] + dt * f;     // re-read: no dynamic register indexing
        out[(size_t)(s0 + s) * L + a] = (T)f;
        if (traj) traj[(size_t)(s0 + s) * L + a] = (T)f;
      }
      if constexpr (CND) {
        const double ab = pw_wave_sum63(ps * (double)absp[s] + vr * (double)absu[s]);
        if (lane == 63 && s0 + s < S) cnd[(size_t)(s0 + s) * L + a] = (T)ab;
      }
      if constexpr (JAC) {
#pragma unroll
        for (int k = 0; k < DK; ++k) {
          const T jv = pw_wave_sum63(accJ[s][k]);
          if (lane == 63 && s0 + s < S && k < d) jac[((size_t)(s0 + s) * L + a) * d + k] = jv;
        }
      }
    }
  }
}

// LDS-resident variant (used when the shared operands of one latent fit in LDS, C5: 108 KB):
// one 512-thread workgroup per CU loads omega_t / phase / zs_t / hz of ITS latent into LDS once
// and then streams many sample groups through it, so the only global traffic in the loop is the
// weight stream itself (operand re-reads through L2 cost 39 % of the plain kernel's time).
// grid = L * nW workgroups; wave w of workgroup (a, i) handles sample groups i*8 + w, + nW*8, ...
template <typename T, int DK, bool JAC, bool CND>
__global__ __launch_bounds__(64 * PW_LDS_WAVES) void k_pathwise_lds(int S, int L, int M, int K, int d, int nW,
                                                      const T* __restrict__ x, const T* __restrict__ omega,
                                                      const T* __restrict__ phase, const T* __restrict__ zs,
                                                      const T* __restrict__ hz, const double* __restrict__ xscale,
                                                      const double* __restrict__ pscale, const double* __restrict__ var,
                                                      const double* __restrict__ meanc, const T* __restrict__ wb,
                                                      T* __restrict__ out, T* __restrict__ traj, T* __restrict__ jac,
                                                      T* __restrict__ cnd, int euler, double dt) {
  typedef typename PwVec<T>::type VT;
  constexpr int W = PwVec<T>::W, NS = MM_PW_NS, BT = 64 * W, NWAVE = PW_LDS_WAVES;
  constexpr int NJ = JAC ? DK : 1;
  extern __shared__ __attribute__((aligned(16))) char pw_smem[];
  // [DK + 1][K + M]: rows 0..d-1 vectors (rows d..DK-1 zero: the k loops below are unconditional),
  // row DK the scalars (phase | hz)
  T* op = reinterpret_cast<T*>(pw_smem);
  const int a = blockIdx.x % L, wgi = blockIdx.x / L;
  const int KT = K + M, nbK = K / BT, NB = KT / BT;
  for (int idx = threadIdx.x * W; idx < (DK + 1) * KT; idx += 64 * PW_LDS_WAVES * W) {
    const int row = idx / KT, col = idx - row * KT;     // KT % W == 0: a vector never straddles rows
    T v[W];
#pragma unroll
    for (int j = 0; j < W; ++j) v[j] = (T)0;
    if (row < d) {
      pw_unpack<T>(*reinterpret_cast<const VT*>((col < K) ? omega + ((size_t)a * d + row) * K + col
                                                          : zs + ((size_t)a * d + row) * M + (col - K)), v);
      // update rows carry z * x_scale^2, so that both halves of the stream use the raw x_s
      const T f = (col < K) ? (T)1 : (T)xscale[a * d + row];
#pragma unroll
      for (int j = 0; j < W; ++j) v[j] *= f;
    } else if (row == DK) {
      pw_unpack<T>(*reinterpret_cast<const VT*>((col < K) ? phase + (size_t)a * K + col
                                                          : hz + (size_t)a * M + (col - K)), v);
    }
#pragma unroll
    for (int j = 0; j < W; ++j) op[idx + j] = v[j];
  }
  __syncthreads();
  // the wave index as a scalar: everything derived from it (sample group, x_s, stream base) is wave-uniform
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int ngroups = (S + NS - 1) / NS;
  T sc[DK];
#pragma unroll
  for (int k = 0; k < DK; ++k) {
    const T v = (T)xscale[a * d + (k < d ? k : 0)];
    sc[k] = (k < d) ? v : (T)0;
  }
  const double ps = pscale[a], vr = var[a], mc = meanc ? meanc[a] : 0.0;

  // The wave's work is ONE flat sequence of weight blocks (its groups back to back, NB blocks each),
  // consumed through PW_RING register sets in rotation: blocks i + 1 .. i + PW_RING - 1 are in flight
  // while block i is reduced -- across group boundaries too, so a new group does not start with an empty pipeline.
  // All loads are unconditional (past the end the last block is re-read): the compiler's s_waitcnt
  // for a block then covers exactly the loads issued up to it.
  const int g_first = wgi * NWAVE + wv, g_stride = nW * NWAVE;
  if (g_first >= ngroups) return;
  const int n_mine = (ngroups - g_first + g_stride - 1) / g_stride;
  const int total = n_mine * NB;
  int ld_g = g_first, ld_tb = 0;                        // next block to load
  auto load_next = [&](VT (&q)[NS]) {
    const T* ptr = wb + (((size_t)ld_g * L + a) * NB + ld_tb) * NS * BT + lane * W;
#pragma unroll
    for (int s = 0; s < NS; ++s) q[s] = *reinterpret_cast<const VT*>(ptr + (size_t)s * BT);
    if (ld_tb + 1 < NB) ++ld_tb;
    else if (ld_g + g_stride < ngroups) { ld_g += g_stride; ld_tb = 0; }
  };
  int cg = g_first, ctb = 0;                            // block being consumed
  T xr[NS][DK], hx[NS], accp[NS], accu[NS], accJ[NS][NJ], absp[NS], absu[NS];
  // PK (the Jacobian pass of an f32 stream): the NS samples of a group in PAIRS -- every per-(term, sample) operation of the pass
  // (the d FMAs of the argument, the d FMAs of the Jacobian sums, the weight product, the sums) is one v_pk_*_f32 over a sample
  // pair with the shared operand (omega_k / c_k of the term) broadcast: half the FMA-class instructions of a pass that is
  // VALU-bound.  Pairs over SAMPLES need no extra accumulators (pairs over terms did: 256 VGPRs + scratch)
  constexpr bool PK = JAC && sizeof(T) == 4 && NS % 2 == 0;
  constexpr int NP = PK ? NS / 2 : 1, NJP = PK ? DK : 1;
  pwf2 xr2[NP][NJP], hx2[NP], accp2[NP], accu2[NP], accJ2[NP][NJP];
  auto consume = [&](const VT (&q)[NS]) {
    const int s0 = cg * NS;
    if (ctb == 0) {                                     // new group: its NS states (wave-uniform loads)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int row = (s0 + s < S) ? s0 + s : S - 1;
        T h = (T)0;
#pragma unroll
        for (int k = 0; k < DK; ++k) {
          const T v = x[(size_t)row * d + (k < d ? k : 0)];
          xr[s][k] = (k < d) ? v : (T)0;
          const T xs = xr[s][k] * sc[k];
          h += xs * xs;
        }
        hx[s] = (T)0.5 * h;
        accp[s] = (T)0; accu[s] = (T)0; absp[s] = (T)0; absu[s] = (T)0;
#pragma unroll
        for (int k = 0; k < NJ; ++k) accJ[s][k] = (T)0;
      }
      if constexpr (PK) {
#pragma unroll
        for (int sp = 0; sp < NP; ++sp) {
          hx2[sp] = (pwf2){(float)hx[2 * sp], (float)hx[2 * sp + 1]};
          accp2[sp] = (pwf2){0.0f, 0.0f}; accu2[sp] = (pwf2){0.0f, 0.0f};
#pragma unroll
          for (int k = 0; k < DK; ++k) {
            xr2[sp][k] = (pwf2){(float)xr[2 * sp][k], (float)xr[2 * sp + 1][k]};
            accJ2[sp][k] = (pwf2){0.0f, 0.0f};
          }
        }
      }
    }
    T wv4[NS][W], cv[DK][W], sv[W];
#pragma unroll
    for (int s = 0; s < NS; ++s) pw_unpack<T>(q[s], wv4[s]);
    const int col = ctb * BT + lane * W;
#pragma unroll
    for (int k = 0; k < DK; ++k) pw_unpack<T>(*reinterpret_cast<const VT*>(op + (size_t)k * KT + col), cv[k]);
    pw_unpack<T>(*reinterpret_cast<const VT*>(op + (size_t)DK * KT + col), sv);
    if (PK && ctb < nbK) {                              // prior block, sample pairs
      if constexpr (PK) {
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
          for (int sp = 0; sp < NP; ++sp) {
            pwf2 arg = PW_B2(sv[j]);
#pragma unroll
            for (int k = 0; k < DK; ++k) arg = __builtin_elementwise_fma(PW_B2(cv[k][j]), xr2[sp][k], arg);
            float c0, s0f, c1, s1f;
            pw_sincos(arg[0], c0, s0f); pw_sincos(arg[1], c1, s1f);
            const pwf2 w2 = {(float)wv4[2 * sp][j], (float)wv4[2 * sp + 1][j]};
            accp2[sp] = __builtin_elementwise_fma(w2, (pwf2){c0, c1}, accp2[sp]);
            const pwf2 t2 = w2 * (pwf2){s0f, s1f};
#pragma unroll
            for (int k = 0; k < DK; ++k) accJ2[sp][k] = __builtin_elementwise_fma(t2, PW_B2(cv[k][j]), accJ2[sp][k]);
          }
        if (ctb == nbK - 1) {
          const double den = 0.6931471805599453 * vr;
          const float fj = den != 0.0 ? (float)(-6.283185307179586 * ps / den) : 0.0f;
#pragma unroll
          for (int sp = 0; sp < NP; ++sp)
#pragma unroll
            for (int k = 0; k < DK; ++k) accJ2[sp][k] = accJ2[sp][k] * PW_B2(fj);
        }
      }
    } else if (PK) {                                    // update block, sample pairs
      if constexpr (PK) {
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
          for (int sp = 0; sp < NP; ++sp) {
            pwf2 arg = PW_B2(-(float)sv[j]) - hx2[sp];
#pragma unroll
            for (int k = 0; k < DK; ++k) arg = __builtin_elementwise_fma(PW_B2(cv[k][j]), xr2[sp][k], arg);
            const pwf2 e2 = {pw_exp(arg[0]), pw_exp(arg[1])};
            const pwf2 t2 = (pwf2){(float)wv4[2 * sp][j], (float)wv4[2 * sp + 1][j]} * e2;
            accu2[sp] = accu2[sp] + t2;
#pragma unroll
            for (int k = 0; k < DK; ++k) accJ2[sp][k] = __builtin_elementwise_fma(t2, PW_B2(cv[k][j]), accJ2[sp][k]);
          }
      }
    } else if (ctb < nbK) {                             // wave-uniform: prior block
#pragma unroll
      for (int j = 0; j < W; ++j)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          T arg = sv[j];
#pragma unroll
          for (int k = 0; k < DK; ++k) arg += cv[k][j] * xr[s][k];
          if constexpr (JAC) {
            T cs, sn;
            pw_sincos(arg, cs, sn);
            accp[s] += wv4[s][j] * cs;
            const T t = wv4[s][j] * sn;
#pragma unroll
            for (int k = 0; k < DK; ++k) accJ[s][k] += t * cv[k][j];
          } else if constexpr (CND) {
            const T c = PW_COS(arg);
            accp[s] += wv4[s][j] * c;                         // (the plain pass's own expression: the value stays bit-equal)
            absp[s] += fabs(wv4[s][j]) * fabs(c);
          } else {
            accp[s] += wv4[s][j] * PW_COS(arg);
          }
        }
      if constexpr (JAC) {
        if (ctb == nbK - 1) {
          // last prior block: d/dx of w cos(2 pi arg) = -2 pi omega w sin.  ONE accumulator set for both halves: the prior sums
          // are rescaled so that the update half's final factor ln2 var applies to the total
          const double den = 0.6931471805599453 * vr;
          const T fj = den != 0.0 ? (T)(-6.283185307179586 * ps / den) : (T)0;
#pragma unroll
          for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int k = 0; k < DK; ++k) accJ[s][k] *= fj;
        }
      }
    } else {                                            // update block
#pragma unroll
      for (int j = 0; j < W; ++j)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          T arg = -sv[j] - hx[s];
#pragma unroll
          for (int k = 0; k < DK; ++k) arg += cv[k][j] * xr[s][k];
          const T e = PW_EXP(arg);
          const T t = wv4[s][j] * e;
          accu[s] += t;
          if constexpr (CND) absu[s] += fabs(wv4[s][j]) * e;
          if constexpr (JAC) {
#pragma unroll
            for (int k = 0; k < DK; ++k) accJ[s][k] += t * cv[k][j];
          }
        }
    }
    if (ctb == NB - 1) {                                // group done: reduce, Euler update, store
      if constexpr (PK) {
#pragma unroll
        for (int sp = 0; sp < NP; ++sp)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            accp[2 * sp + hh] = (T)accp2[sp][hh]; accu[2 * sp + hh] = (T)accu2[sp][hh];
#pragma unroll
            for (int k = 0; k < DK; ++k) accJ[2 * sp + hh][PK ? k : 0] = (T)accJ2[sp][k][hh];
          }
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const double t = pw_wave_sum63(ps * (double)accp[s] + vr * (double)accu[s]);
        if (lane == 63 && s0 + s < S) {
          double f = t + mc;
          // x_a re-read (d == L here): indexing the register array with the runtime latent index would
          // put it in scratch memory
          if (euler) f = (double)x[(size_t)(s0 + s) * d + a] + dt * f;
          out[(size_t)(s0 + s) * L + a] = (T)f;
          if (traj) traj[(size_t)(s0 + s) * L + a] = (T)f;
        }
        if constexpr (CND) {
          const double ab = pw_wave_sum63(ps * (double)absp[s] + vr * (double)absu[s]);
          if (lane == 63 && s0 + s < S) cnd[(size_t)(s0 + s) * L + a] = (T)ab;
        }
        if constexpr (JAC) {
          // update rows carry c = z xscale^2 and arg = c . x - hz - hx: d arg / d x_k = c_k - xscale_k^2 x_k
          const T fj = (T)(0.6931471805599453 * vr);
#pragma unroll
          for (int k = 0; k < DK; ++k) {
            const T jv = pw_wave_sum63(fj * (accJ[s][k] - sc[k] * sc[k] * xr[s][k] * accu[s]));
            if (lane == 63 && s0 + s < S && k < d) jac[((size_t)(s0 + s) * L + a) * d + k] = jv;
          }
        }
      }
      ctb = 0; cg += g_stride;
    } else {
      ++ctb;
    }
  };
  VT q[PW_RING][NS];
#pragma unroll
  for (int r = 0; r < PW_RING - 1; ++r) load_next(q[r]);
  for (int i = 0; i < total; i += PW_RING) {
#pragma unroll
    for (int r = 0; r < PW_RING; ++r) {
      load_next(q[(r + PW_RING - 1) % PW_RING]);
      if (r == 0 || i + r < total) consume(q[r]);
    }
  }
}

template <typename T>
static int pw_launch(int S, int L, int M, int K, int d, const T* x, const T* omega, const T* phase, const T* zs,
                     const T* hz, const double* xscale, const double* pscale, const double* var,
                     const double* meanc, const T* wb, T* out, T* traj, int euler, double dt, hipStream_t s,
                     T* jac = nullptr, T* cnd = nullptr) {
  if (jac && cnd) return MM_E_ARG;                  // (one extra output per pass)
  if (jac && d > 8) return MM_E_DIM;               // the Jacobian pass keeps (1 + d) accumulators per sample: d <= 8
  // LDS-resident operands when one latent's (d + 1) x (K + M) block fits (<= 144 KB)
  const int dk = d <= 4 ? 4 : d <= 8 ? 8 : d <= 16 ? 16 : 32;
  const size_t lds_bytes = (size_t)(dk + 1) * (K + M) * sizeof(T);      // rows d..dk-1 are zero padding
  if (lds_bytes <= 144 * 1024) {
    const int ngroups = (S + MM_PW_NS - 1) / MM_PW_NS;
    int nW = 256 / L; if (nW < 1) nW = 1;                           // ~ one workgroup per CU
    while (nW > 1 && (nW - 1) * PW_LDS_WAVES >= ngroups) --nW;                 // no idle workgroups on small S
#define PW_LAUNCH_LDS_(DK_, JAC_, CND_)                                                             \
    do {                                                                                            \
      /* raise the dynamic-LDS limit once per instantiation and device (the call is slow: not per launch) */ \
      static std::atomic<unsigned long long> lds_set{0ull};                                         \
      int dev_ = 0;                                                                                 \
      if (hipGetDevice(&dev_) != hipSuccess) dev_ = 64;                                             \
      const unsigned long long bit_ = (dev_ >= 0 && dev_ < 64) ? (1ull << dev_) : 0ull;             \
      if (!bit_ || !(lds_set.load() & bit_)) {                                                      \
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pathwise_lds<T, DK_, JAC_, CND_>), \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024); \
        if (ea != hipSuccess) return (int)ea;                                                       \
        lds_set.fetch_or(bit_);                                                                     \
      }                                                                                             \
      hipLaunchKernelGGL((k_pathwise_lds<T, DK_, JAC_, CND_>), dim3(L * nW), dim3(64 * PW_LDS_WAVES), lds_bytes, s, S, L, M, K, d, nW, \
                         x, omega, phase, zs, hz, xscale, pscale, var, meanc, wb, out, traj, jac, cnd, euler, dt); \
    } while (0)
#define PW_LAUNCH_LDS(DK_) do { if (jac) PW_LAUNCH_LDS_(DK_, true, false); else if (cnd) PW_LAUNCH_LDS_(DK_, false, true); else PW_LAUNCH_LDS_(DK_, false, false); } while (0)
#define PW_LAUNCH_LDS_NJ(DK_) do { if (cnd) PW_LAUNCH_LDS_(DK_, false, true); else PW_LAUNCH_LDS_(DK_, false, false); } while (0)
    if (d <= 4) PW_LAUNCH_LDS(4);
    else if (d <= 8) PW_LAUNCH_LDS(8);
    else if (d <= 16) PW_LAUNCH_LDS_NJ(16);
    else PW_LAUNCH_LDS_NJ(32);
#undef PW_LAUNCH_LDS_NJ
#undef PW_LAUNCH_LDS
#undef PW_LAUNCH_LDS_
    hipError_t el = hipGetLastError();
    return el == hipSuccess ? 0 : (int)el;
  }
  dim3 grid((S + MM_PW_NS - 1) / MM_PW_NS);
#define PW_LAUNCH_(DK_, JAC_, CND_) hipLaunchKernelGGL((k_pathwise<T, DK_, JAC_, CND_>), grid, dim3(256), 0, s, S, L, M, K, d, x, omega, phase, \
                                                zs, hz, xscale, pscale, var, meanc, wb, out, traj, jac, cnd, euler, dt)
#define PW_LAUNCH(DK_) do { if (jac) PW_LAUNCH_(DK_, true, false); else if (cnd) PW_LAUNCH_(DK_, false, true); else PW_LAUNCH_(DK_, false, false); } while (0)
#define PW_LAUNCH_NJ(DK_) do { if (cnd) PW_LAUNCH_(DK_, false, true); else PW_LAUNCH_(DK_, false, false); } while (0)
  if (d <= 4) PW_LAUNCH(4);
  else if (d <= 8) PW_LAUNCH(8);
  else if (d <= 16) PW_LAUNCH_NJ(16);
  else PW_LAUNCH_NJ(32);
#undef PW_LAUNCH_NJ
#undef PW_LAUNCH
#undef PW_LAUNCH_
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

extern "C" int mm_pathwise_eval_bound(int S, int L, int M, int K, int d, int dtype,
                                      const void* x, const void* omega_t, const void* phase, const void* zs_t,
                                      const void* hz, const double* x_scale, const double* prior_scale,
                                      const double* variance, const double* mean_c, const void* wb, void* f_out,
                                      void* abs_out, void* stream) {
  int rc = pw_check(S, L, M, K, d, dtype);
  if (rc) return rc;
  if (!x || !omega_t || !phase || !zs_t || !hz || !x_scale || !prior_scale || !variance || !wb || !f_out || !abs_out)
    return MM_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MM_F64)
    return pw_launch<double>(S, L, M, K, d, (const double*)x, (const double*)omega_t, (const double*)phase,
                             (const double*)zs_t, (const double*)hz, x_scale, prior_scale, variance, mean_c,
                             (const double*)wb, (double*)f_out, nullptr, 0, 0.0, s, nullptr, (double*)abs_out);
  return pw_launch<float>(S, L, M, K, d, (const float*)x, (const float*)omega_t, (const float*)phase,
                          (const float*)zs_t, (const float*)hz, x_scale, prior_scale, variance, mean_c,
                          (const float*)wb, (float*)f_out, nullptr, 0, 0.0, s, nullptr, (float*)abs_out);
}

// one evaluation f [S,L] (and, jac != NULL, d f / d x [S,L,d]) for the other translation units (mm_pathwise_policy.hip)
int mm_pathwise_launch(int S, int L, int M, int K, int d, int dtype, const void* x, const void* omega_t, const void* phase,
                       const void* zs_t, const void* hz, const double* x_scale, const double* prior_scale, const double* variance,
                       const double* mean_c, const void* wb, void* f_out, void* jac, hipStream_t s) {
  int rc = pw_check(S, L, M, K, d, dtype);
  if (rc) return rc;
  if (dtype == MM_F64)
    return pw_launch<double>(S, L, M, K, d, (const double*)x, (const double*)omega_t, (const double*)phase,
                             (const double*)zs_t, (const double*)hz, x_scale, prior_scale, variance, mean_c,
                             (const double*)wb, (double*)f_out, nullptr, 0, 0.0, s, (double*)jac);
  return pw_launch<float>(S, L, M, K, d, (const float*)x, (const float*)omega_t, (const float*)phase,
                          (const float*)zs_t, (const float*)hz, x_scale, prior_scale, variance, mean_c,
                          (const float*)wb, (float*)f_out, nullptr, 0, 0.0, s, (float*)jac);
}

extern "C" int mm_pathwise_eval_jac(int S, int L, int M, int K, int d, int dtype,
                                    const void* x, const void* omega_t, const void* phase, const void* zs_t,
                                    const void* hz, const double* x_scale, const double* prior_scale,
                                    const double* variance, const double* mean_c, const void* wb, void* f_out,
                                    void* jac_out, void* stream) {
  if (!x || !omega_t || !phase || !zs_t || !hz || !x_scale || !prior_scale || !variance || !wb || !f_out || !jac_out)
    return MM_E_ARG;
  return mm_pathwise_launch(S, L, M, K, d, dtype, x, omega_t, phase, zs_t, hz, x_scale, prior_scale, variance, mean_c, wb,
                            f_out, jac_out, (hipStream_t)stream);
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
