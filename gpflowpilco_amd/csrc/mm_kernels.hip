// Moment-matched GP propagation for MI355X (gfx950): hand-written HIP kernels + C ABI.
//
// Replaces, behind include/gpflowpilco_mm.h, the per-step arithmetic of
//   gpflow_pilco/moment_matching/models.py:200-299 (_mm_gauss_svgp_mo; _so and GPR are L == 1)
//   gpflow_pilco/utils/kernel_expectation.py:72-247 (<K_Zx K_xZ'> pair kernel and fan-outs)
//   gpflow <k(x,Z)> (third party)   and   gpflow_pilco/dynamics/solvers.py:110-135.
//
// Algorithm (DESIGN.md "Centred fused reduce"): the reference materialises
// Q = eKuffu [B,L,M,L,M] and runs two triangular solves per step.  Here, with
// beta_a = Kuu_a^-1 u_a and C_a = Kuu_a^-1 S_a Kuu_a^-1 - Kuu_a^-1 precomputed once,
//   f1_a     = sum_i w_i,                          w_i = beta_i q_i
//   Sff_aa'  = sum_ij w_i (exp(delta_ij) - 1) w'_j  + [a==a'] (sigma_a^2 + sum_ij C_ij q_i exp(delta_ij) q_j)
//   delta_ij = log Q_ij - log q_i - log q'_j = const + rho_i + gamma_j + zeta_i^T G zeta_j
// which is algebraically f2 - f1 f1^T (+ E[Var f]) of models.py:244-261 but never forms Q
// and has no catastrophic cancellation (delta -> 0 as Sigma -> 0).
//
// Stages per moment match (all enqueue-only):
//   k_prep      one wave per (b, pair) / (b, latent): d x d Cholesky algebra in f64 (LDS)
//   k_qvec      one workgroup per (b, latent): q_i, w_i, f1, Sigma^-1 Cov(x,f)
//   k_pairvec   per (b, pair, m): rho_m, g_m = G zeta_m, gamma'_m  (streamed operands of the reduce; k_pairvec_reg for d <= 8)
//   k_wmom_gemm, k_spoly (mm_moments.hip; f32 models): the polynomial part of the off-diagonal sums from f64 moments
//   k_qred_*    the M x M fused reduce: diagonal pairs (a == a', incl. the C-weighted term) always in
//               f64, off-diagonal pairs in T (generic VALU kernel here; f32 MFMA kernel in mm_mfma.hip)
//   k_finalize  deterministic sum of the partial slabs -> Sff
//   k_euler     MomentMatchingEuler.step
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"
#include "mm_cost.h"
#include "mm_dev.h"
#include "mm_mono.h"

#include <mutex>
#include <unordered_map>
#include "mm_fork.h"

#define MM_ABI_VERSION 2

namespace {
struct MMForkKey {
  int dev; hipStream_t stream;
  bool operator==(const MMForkKey& o) const { return dev == o.dev && stream == o.stream; }
};
struct MMForkKeyHash {
  size_t operator()(const MMForkKey& k) const { return std::hash<const void*>()((const void*)k.stream) * 31u + (size_t)k.dev; }
};
std::mutex g_fork_mu;                                       // guards the table only (lookup / insert), never an enqueue
std::unordered_map<MMForkKey, MMFork*, MMForkKeyHash> g_forks;   // entries live for the process (a few dozen bytes + one stream each)

MMFork* mm_fork_lookup(hipStream_t stream, bool create) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  const MMForkKey key{dev, stream};
  std::lock_guard<std::mutex> g(g_fork_mu);
  auto it = g_forks.find(key);
  if (it != g_forks.end()) return it->second;               // (nullptr: creation failed once for this stream -- not retried)
  if (!create) return nullptr;
  MMFork* f = new MMFork();
  const bool ok = hipStreamCreateWithFlags(&f->s2, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&f->fork, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&f->join, hipEventDisableTiming) == hipSuccess;
  if (!ok) { delete f; f = nullptr; }
  g_forks.emplace(key, f);
  return f;
}
}  // namespace

MMFork* mm_fork_get(hipStream_t stream) { return mm_fork_lookup(stream, true); }

int mm_fork_join_wait(hipStream_t stream) {
  MMFork* fork = mm_fork_lookup(stream, false);               // a side stream is never a key: in order with itself, nothing to wait for
  if (!fork) return 0;
  const hipError_t e = hipStreamWaitEvent(stream, fork->join, 0);
  return e == hipSuccess ? 0 : (int)e;
}

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void mm_decode_pair(int p, int L, int& a, int& a2) {
  // pairs [0, L) are the diagonal (a, a); the rest enumerate a < a' row by row.
  if (p < L) { a = p; a2 = p; return; }
  int r = p - L;
  int i = 0;
  while (r >= L - 1 - i) { r -= L - 1 - i; ++i; }
  a = i; a2 = i + 1 + r;
}

__device__ __forceinline__ double mm_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum over a 256-thread workgroup; result valid in thread 0. `red` holds >= 4 doubles.
__device__ __forceinline__ double mm_block_sum256(double v, double* red) {
  v = mm_wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float mm_expm1(float x) { return expm1f(x); }
__device__ __forceinline__ double mm_expm1(double x) { return expm1(x); }

// ---------------------------------------------------------------------------------------------
// Pack order (MMModelLayout::perm): per latent the inducing points are sorted by the norm of (z - zbar) / lengthscale.
// Every sum of the path runs over all points of a latent, so the order is free; |b_ij| <= |G| |zeta_i| |zeta_j| makes tiles of
// sorted points homogeneous: along the C3 rollout the mean degree of the diagonal sweep's tier polynomial falls from 9.8 to
// 8.1 FMAs per entry (with the column recentring of k_pairvec_reg), and 83 % instead of 32 % of the dense off-diagonal wave
// tiles have max|b| <= 1/4 (scratch study).
//   k_pack_key   : keys (one workgroup per latent)
//   k_pack_rank  : rank by counting, ties by index (a permutation for ANY keys; non-finite keys sort last) -> perm
//   k_pack_gather: Z64, beta64 in packed order; the other pack kernels read THOSE
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_key(char* packed, MMModelLayout lay, int M, int d, const double* __restrict__ Z,
                                                  const double* __restrict__ ls) {
  const int a = blockIdx.x, tid = threadIdx.x;
  __shared__ double red[4];
  __shared__ double zb[MM_DMAX];
  for (int k = 0; k < d; ++k) {
    double s = 0.0;
    for (int m = tid; m < M; m += 256) s += Z[((size_t)a * M + m) * d + k];
    s = mm_block_sum256(s, red);
    if (tid == 0) zb[k] = s / (double)M;
    __syncthreads();
  }
  double* key = (double*)(packed + lay.skey) + (size_t)a * lay.Mp;
  for (int m = tid; m < M; m += 256) {
    double s2 = 0.0;
    for (int k = 0; k < d; ++k) { const double v = (Z[((size_t)a * M + m) * d + k] - zb[k]) / ls[a * d + k]; s2 = fma(v, v, s2); }
    key[m] = (s2 == s2 && s2 < 1.0e300) ? s2 : 1.0e300;
  }
}

__global__ __launch_bounds__(256) void k_pack_rank(char* packed, MMModelLayout lay, int M, int sorted) {
  // grid (Mp / 256, L)
  const int a = blockIdx.y, tid = threadIdx.x, i = blockIdx.x * 256 + tid;
  int* perm = (int*)(packed + lay.perm) + (size_t)a * lay.Mp;
  if (!sorted) { if (i < lay.Mp) perm[i] = i; return; }
  const double* key = (const double*)(packed + lay.skey) + (size_t)a * lay.Mp;
  __shared__ double kj[256];
  const double ki = i < M ? key[i] : 0.0;
  int r = 0;
  for (int j0 = 0; j0 < M; j0 += 256) {
    __syncthreads();
    kj[tid] = j0 + tid < M ? key[j0 + tid] : 0.0;
    __syncthreads();
    const int n = M - j0 < 256 ? M - j0 : 256;
    for (int t = 0; t < n; ++t) { const double k2 = kj[t]; r += (k2 < ki || (k2 == ki && j0 + t < i)) ? 1 : 0; }
  }
  if (i < M) perm[r] = i;
  else if (i < lay.Mp) perm[i] = i;
}

__global__ __launch_bounds__(256) void k_pack_gather(char* packed, MMModelLayout lay, int M, int d, const double* __restrict__ Z,
                                                     const double* __restrict__ beta) {
  // grid (ceil(M / 256), L)
  const int a = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int src = ((const int*)(packed + lay.perm))[(size_t)a * lay.Mp + m];
  double* Z64 = (double*)(packed + lay.Z64) + ((size_t)a * M + m) * d;
  for (int k = 0; k < d; ++k) Z64[k] = Z[((size_t)a * M + src) * d + k];
  ((double*)(packed + lay.beta64))[(size_t)a * M + m] = beta[(size_t)a * M + src];
}

// ---------------------------------------------------------------------------------------------
// k_pack_model: raw f64 model -> packed buffer (centred/padded T copies for the reduce kernels)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_pack_vectors(char* packed, MMModelLayout lay, int L, int M, int d,
                               const double* __restrict__ ls,
                               const double* __restrict__ var,
                               const double* __restrict__ mean_c) {
  // one workgroup per latent a; Z: the inducing inputs in PACKED order (k_pack_gather wrote Z64 and beta64)
  const int a = blockIdx.x, tid = threadIdx.x;
  const double* Z = (const double*)(packed + lay.Z64);
  double* zbar = (double*)(packed + lay.zbar) + (size_t)a * d;
  double* ls2 = (double*)(packed + lay.ls2) + (size_t)a * d;
  double* Zc64 = (double*)(packed + lay.Zc64) + (size_t)a * lay.Mp * lay.Kz;
  T* Zc = (T*)(packed + lay.Zc) + (size_t)a * lay.Mp * lay.Kz;
  __shared__ double red[4];
  __shared__ double zb[MM_DMAX];
  for (int k = 0; k < d; ++k) {
    double s = 0.0;
    for (int m = tid; m < M; m += 256) s += Z[((size_t)a * M + m) * d + k];
    s = mm_block_sum256(s, red);
    if (tid == 0) zb[k] = s / (double)M;
  }
  __syncthreads();
  if (tid < d) { zbar[tid] = zb[tid]; const double l = ls[a * d + tid]; ls2[tid] = l * l; }
  if (tid == 0) {
    ((double*)(packed + lay.var))[a] = var[a];
    ((double*)(packed + lay.meanc))[a] = mean_c ? mean_c[a] : 0.0;
  }
  {
    double* Zt = (double*)(packed + lay.Zt64) + (size_t)a * d * lay.Mp;
    for (int idx = tid; idx < d * lay.Mp; idx += 256) {
      const int k = idx / lay.Mp, m = idx - k * lay.Mp;
      Zt[idx] = m < M ? Z[((size_t)a * M + m) * d + k] : 0.0;
    }
  }
  for (int idx = tid; idx < lay.Mp * lay.Kz; idx += 256) {
    const int m = idx / lay.Kz, k = idx - m * lay.Kz;
    double v = 0.0;
    if (m < M && k < d) v = Z[((size_t)a * M + m) * d + k] - zb[k];
    Zc64[idx] = v;
    if (sizeof(T) != 8) Zc[idx] = (T)v;
  }
  if (sizeof(T) != 8) {
    // moment table: every monomial of zc of total degree <= mm_moment_deg(d), graded colex order (mm_mono.h);
    // zero beyond M and beyond the last column
    double* Zm = (double*)(packed + lay.Zm) + (size_t)a * lay.Mp * lay.KMp;
    const int deg = d <= 8 ? 4 : 2;                       // mm_moment_deg(d) (host function)
    const int ncol = mm_mono_off(deg + 1, d);
    for (int idx = tid; idx < lay.Mp * lay.KMp; idx += 256) {
      const int m = idx / lay.KMp, c = idx - m * lay.KMp;
      double v = 0.0;
      if (m < M && c < ncol) {
        int n = 0;
        while (mm_mono_off(n + 1, d) <= c) ++n;                // degree of column c
        int k[4] = {0, 0, 0, 0};
        mm_mono_unrank(c - mm_mono_off(n, d), n, k);
        v = 1.0;
        for (int t = 0; t < n; ++t) v *= Z[((size_t)a * M + m) * d + k[t]] - zb[k[t]];
      }
      Zm[idx] = v;
    }
    // rank table of k_spoly (latent 0 writes it): position = index tuple base DK, value = colex rank or -1
    if (a == 0) {
      short* rt = (short*)(packed + lay.rtab);
      const int lb = d <= 8 ? 3 : 5;
      size_t base = 0;
      for (int kk = 1; kk <= deg; ++kk) {
        const int sz = 1 << (lb * kk);
        for (int idx = tid; idx < sz; idx += 256) {
          int dg[4] = {0, 0, 0, 0};
          bool in = true;
          for (int t = 0; t < kk; ++t) { dg[t] = (idx >> (lb * t)) & ((1 << lb) - 1); in = in && dg[t] < d; }
          for (int i = 1; i < kk; ++i)
            for (int j = i; j > 0 && dg[j - 1] > dg[j]; --j) { const int t2 = dg[j]; dg[j] = dg[j - 1]; dg[j - 1] = t2; }
          rt[base + idx] = in ? (short)mm_mono_rank(dg, kk) : (short)-1;
        }
        base += sz;
      }
    }
  }
  {
    // max_m |zc_m|^2 (the column side of the Cauchy-Schwarz bound on |b_ij|)
    double zm2 = 0.0;
    for (int m = tid; m < M; m += 256) {
      double s2 = 0.0;
      for (int k = 0; k < d; ++k) { const double v = Z[((size_t)a * M + m) * d + k] - zb[k]; s2 += v * v; }
      zm2 = s2 > zm2 ? s2 : zm2;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(zm2, off, 64); zm2 = o > zm2 ? o : zm2; }
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = zm2;
    __syncthreads();
    if (tid == 0) ((double*)(packed + lay.zmax2))[a] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    // the same per 32-point column tile (MMModelLayout::zt2)
    float* zt2 = (float*)(packed + lay.zt2) + (size_t)a * (lay.Mp / 32);
    for (int t = tid; t < lay.Mp / 32; t += 256) {
      double mx = 0.0;
      for (int j = 0; j < 32; ++j) {
        const int m = t * 32 + j;
        if (m >= M) break;
        double s2 = 0.0;
        for (int k = 0; k < d; ++k) { const double v = Z[((size_t)a * M + m) * d + k] - zb[k]; s2 += v * v; }
        mx = s2 > mx ? s2 : mx;
      }
      zt2[t] = (float)(mx * 1.000001);
    }
  }
  if (sizeof(T) != 8) {
    // bf16 3-way split of the centred inputs for the f32 MFMA kernel, tile-and-part major:
    // [Mp / 32 column tiles][3 (h,m,l)][32 columns][8 nd8] -- one part of one 32-column tile is ONE contiguous
    // 512 nd8-byte block, so each half-wave of a tile load reads a fully coalesced segment
    unsigned short* Zs3 = (unsigned short*)(packed + lay.Zs3) + (size_t)a * lay.Mp * 24 * lay.nd8;
    const int kw = 8 * lay.nd8;
    for (int idx = tid; idx < lay.Mp * kw; idx += 256) {
      const int m = idx / kw, k = idx - m * kw;
      float v = 0.0f;
      if (m < M && k < d) v = (float)(Z[((size_t)a * M + m) * d + k] - zb[k]);
      const __bf16 h = (__bf16)v;
      float r = v - (float)h;
      const __bf16 mm = (__bf16)r;
      r -= (float)mm;
      const __bf16 l = (__bf16)r;
      unsigned short* o = Zs3 + ((size_t)(m >> 5) * 3 * 32 + (m & 31)) * kw + k;
      o[0] = __builtin_bit_cast(unsigned short, h);
      o[32 * kw] = __builtin_bit_cast(unsigned short, mm);
      o[64 * kw] = __builtin_bit_cast(unsigned short, l);
    }
    if (d <= 8) {
      // degree-2 monomial images for the backward's aggregate product (MMModelLayout::Zq2)
      unsigned short* Zq = (unsigned short*)(packed + lay.Zq2) + (size_t)a * (lay.Mp / 32) * 4096;
      const int nslot = 1 + d + d * (d + 1) / 2;
      for (int idx = tid; idx < (lay.Mp / 32) * 2048; idx += 256) {       // one (hi, lo) pair per index
        int r = idx;
        const int t = r & 7; r >>= 3;
        const int ln = r & 63; r >>= 6;
        const int nb = r & 1; r >>= 1;
        const int s2 = r & 1; r >>= 1;
        const int ct = r;
        const int slot = 32 * nb + (ln & 31), hh = ln >> 5;
        const int m = 32 * ct + 16 * s2 + 8 * (t >> 2) + 4 * hh + (t & 3);
        float v = 0.0f;
        if (m < M && slot < nslot) {
          if (slot == 0) v = 1.0f;
          else if (slot <= d) v = (float)(Z[((size_t)a * M + m) * d + slot - 1] - zb[slot - 1]);
          else {
            int qq = slot - 1 - d, l0 = 0;
            while (qq >= d - l0) { qq -= d - l0; ++l0; }
            const int l1 = l0 + qq;
            v = (float)((Z[((size_t)a * M + m) * d + l0] - zb[l0]) * (Z[((size_t)a * M + m) * d + l1] - zb[l1]));
          }
        }
        const __bf16 h = (__bf16)v;
        const __bf16 lo = (__bf16)(v - (float)h);
        unsigned short* o = Zq + ((((size_t)(ct * 2 + s2) * 2 + nb) * 2) * 64 + ln) * 8 + t;
        o[0] = __builtin_bit_cast(unsigned short, h);
        o[512] = __builtin_bit_cast(unsigned short, lo);
      }
    }
  }
}

__global__ void k_pack_C(char* packed, MMModelLayout lay, int L, int M, const double* __restrict__ C) {
  // grid (Mp/256, Mp, L): zero-padded f64 copy of C in packed order (rows and columns)
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y, a = blockIdx.z;
  if (j >= lay.Mp) return;
  double* Cm = (double*)(packed + lay.Cm) + ((size_t)a * lay.Mp + i) * lay.Mp;
  const int* pm = (const int*)(packed + lay.perm) + (size_t)a * lay.Mp;   // packed position -> the caller's index
  double v = 0.0;
  if (i < M && j < M) v = C[((size_t)a * M + pm[i]) * M + pm[j]];
  Cm[j] = v;
}

// ---------------------------------------------------------------------------------------------
// k_prep: per (b, pair) and per (b, latent) d x d algebra, f64, one wave each
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(192) void k_prep(const double* __restrict__ ls2, const double* __restrict__ var,
                                              int L, int d, int P,
                                              const T* __restrict__ mu, const T* __restrict__ Sigma,
                                              double* __restrict__ pairmat, double* __restrict__ latmat,
                                              unsigned int* __restrict__ amax, int32_t* status, int pairs_pass) {
  // pairs_pass == 0: the L latent items (blockIdx.x = a); pairs_pass == 1: the P pair items, which take
  // (Sigma + Lambda_a)^-1 and its log-determinant from the latent pass instead of refactorising them (one
  // Cholesky-inverse per pair instead of three) -- both with 64 threads; pairs_pass == 2: all P + L items in ONE launch,
  // pairs self-contained -- for small problems, where a launch boundary costs more than the factorisations: 192
  // threads, the three factorisations of a pair item run CONCURRENTLY, one per wave (B = 1 rollouts are a chain of
  // dependent launches: the critical path of this kernel is their step time)
  extern __shared__ double smem[];
  const int dp = d + 1, msz = d * dp;
  double* Sg = smem;            // Sigma_b (symmetrised from the lower triangle)
  double* A0 = Sg + msz;        // (Sigma + V)^-1
  double* A1 = A0 + msz;        // (Sigma + Lambda_a)^-1
  double* A2 = A1 + msz;        // (Sigma + Lambda_a')^-1
  double* Tm = A2 + msz;        // T
  double* Y = Tm + msz;         // scratch of wave 0; waves 1, 2 (single-launch mode): Y + msz, Y + 2 msz
  __shared__ double lds3[3];
  __shared__ int okw[3];
  const int tid = threadIdx.x, nt = blockDim.x, wv = tid >> 6;
  const int item = pairs_pass == 0 ? P + blockIdx.x : blockIdx.x, b = blockIdx.y;
  bool ok = true;
  const T* Sb = Sigma + (size_t)b * d * d;
  for (int idx = tid; idx < d * d; idx += nt) {
    const int i = idx / d, j = idx - i * d;
    Sg[i * dp + j] = (double)(i >= j ? Sb[i * d + j] : Sb[j * d + i]);
  }
  if (tid < 3) okw[tid] = 1;
  __syncthreads();
  if (item >= P) {
    // latent item: P_a = (Sigma + Lambda_a)^-1, lognorm_a = log var + sum log ls - 0.5 logdet
    const int a = item - P;
    const double* la = ls2 + a * d;
    for (int idx = tid; idx < d * d; idx += nt) {
      const int i = idx / d, j = idx - i * d;
      A1[i * dp + j] = Sg[i * dp + j] + (i == j ? la[i] : 0.0);
      A0[i * dp + j] = i == j ? 1.0 : 0.0;                   // (waves 1, 2 of the single-launch mode factorise an identity)
      A2[i * dp + j] = i == j ? 1.0 : 0.0;
    }
    double* Aw = wv == 0 ? A1 : wv == 1 ? A0 : A2;
    const double ldw = mm_spd_inverse(Aw, Y + wv * msz, d, dp, &ok);
    if (wv == 0 && !ok) okw[0] = 0;
    if (tid == 0) lds3[0] = ldw;
    __syncthreads();
    ok = okw[0] != 0;
    const double ld = lds3[0];
    double* out = latmat + ((size_t)b * L + a) * (2 * d * d + 2);
    for (int idx = tid; idx < d * d; idx += nt) { const int i = idx / d, j = idx - i * d; out[idx] = A1[i * dp + j]; }
    {
      double sl = (tid < d) ? log(la[tid]) : 0.0;            // wave 0: one logarithm per lane, wave sum
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sl += __shfl_xor(sl, off, 64);
      if (tid == 0) {
        out[d * d] = log(var[a]) + 0.5 * sl - 0.5 * ld;
        out[d * d + 1] = ld;
      }
    }
    // E_a = sym(Lambda_a^-1 Sigma P_a) (= Lambda_a^-1 - P_a, in product form: no cancellation for small Sigma)
    for (int idx = tid; idx < d * d; idx += nt) {
      const int i = idx / d, j = idx - i * d;
      double s1 = 0.0, t1 = 0.0;
      for (int k = 0; k < d; ++k) {
        s1 += Sg[i * dp + k] * A1[k * dp + j];
        t1 += Sg[j * dp + k] * A1[k * dp + i];
      }
      out[d * d + 2 + idx] = 0.5 * (s1 / la[i] + t1 / la[j]);
    }
  } else {
    int a, a2;
    mm_decode_pair(item, L, a, a2);
    if (amax && item >= L && tid == 0) {                    // k_pairvec max-es |A_i|^2 into both: amax, and amaxc right behind it
      amax[(size_t)b * (P - L) + (item - L)] = 0u;
      amax[((size_t)gridDim.y + b) * (P - L) + (item - L)] = 0u;
    }
    const double* la = ls2 + a * d;
    const double* lb = ls2 + a2 * d;
    for (int idx = tid; idx < d * d; idx += nt) {
      const int i = idx / d, j = idx - i * d;
      const double s = Sg[i * dp + j];
      const double v = la[i] * lb[i] / (la[i] + lb[i]);   // kernel_expectation.py:119
      A0[i * dp + j] = s + (i == j ? v : 0.0);
      if (pairs_pass != 1) {      // single-launch mode: the two log-determinants are factorised here
        A1[i * dp + j] = s + (i == j ? la[i] : 0.0);
        A2[i * dp + j] = s + (i == j ? lb[i] : 0.0);
      }
    }
    double ldS, ldA, ldB;
    if (pairs_pass == 1) {
      ldS = mm_spd_inverse(A0, Y, d, dp, &ok);
      ldA = latmat[((size_t)b * L + a) * (2 * d * d + 2) + d * d + 1];
      ldB = latmat[((size_t)b * L + a2) * (2 * d * d + 2) + d * d + 1];
    } else {                       // single-launch mode (small problems): self-contained, one factorisation per wave
      double* Aw = wv == 0 ? A0 : wv == 1 ? A1 : A2;
      const double ldw = mm_spd_inverse(Aw, Y + wv * msz, d, dp, &ok);
      if (!ok) okw[wv] = 0;
      if ((tid & 63) == 0) lds3[wv] = ldw;
      __syncthreads();
      ok = okw[0] && okw[1] && okw[2];
      ldS = lds3[0]; ldA = lds3[1]; ldB = lds3[2];
    }
    // T = V S^-1 Sigma (product form: no cancellation), symmetrised below
    for (int idx = tid; idx < d * d; idx += nt) {
      const int i = idx / d, j = idx - i * d;
      double s = 0.0;
      for (int k = 0; k < d; ++k) s += A0[i * dp + k] * Sg[k * dp + j];
      Y[i * dp + j] = (la[i] * lb[i] / (la[i] + lb[i])) * s;
    }
    __syncthreads();
    for (int idx = tid; idx < d * d; idx += nt) {
      const int i = idx / d, j = idx - i * d;
      Tm[i * dp + j] = 0.5 * (Y[i * dp + j] + Y[j * dp + i]);
    }
    __syncthreads();
    double* out = pairmat + ((size_t)b * P + item) * (d * d + 1);
    // G = Lambda_a^-1 T Lambda_a'^-1   (rho_i, gamma_j follow from G and the per-latent E_a: k_pairvec)
    for (int idx = tid; idx < d * d; idx += nt) {
      const int i = idx / d, j = idx - i * d;
      out[idx] = Tm[i * dp + j] / (la[i] * lb[j]);
    }
    {
      // 0.5 sum_k [log V_k - log Lam_a,k - log Lam_a',k] = -0.5 sum_k log(Lam_a,k + Lam_a',k): one logarithm per lane
      double lsum = (tid < d) ? log(la[tid] + lb[tid]) : 0.0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off, 64);
      // log kappa_ab - lognorm_a - lognorm_a'  (the variances cancel)
      if (tid == 0) out[d * d] = -0.5 * ldS - 0.5 * lsum + 0.5 * ldA + 0.5 * ldB;
    }
  }
  if (!ok && tid == 0 && status) {
    atomicMax(status, (int)gridDim.y - b);   // B - b: the host decodes the smallest failing b
    status[1] = item;
  }
}

// ---------------------------------------------------------------------------------------------
// k_qvec: q_i, w_i = beta_i q_i, f1, Sigma^-1 Cov(x, f).  One workgroup per (latent, b).
// ---------------------------------------------------------------------------------------------
template <typename T, int DK>
__global__ __launch_bounds__(256) void k_qvec(const double* __restrict__ Zt64, const double* __restrict__ beta64,
                                              const double* __restrict__ meanc,
                                              int L, int M, int Mp, int d,
                                              const T* __restrict__ mu, const double* __restrict__ latmat,
                                              double* __restrict__ w64, double* __restrict__ q64, T* __restrict__ w,
                                              double* __restrict__ f1raw, double* __restrict__ rho1,
                                              T* __restrict__ f1, T* __restrict__ cross, T* __restrict__ q_out,
                                              double* __restrict__ mu64, double* __restrict__ lq,
                                              const int* __restrict__ perm) {
  const int a = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  // P_a = (Sigma + Lambda_a)^-1 and E_a, zero padded to DK x DK: compile-time LDS offsets (wide broadcast reads)
  __shared__ double Pa[DK * DK], Ea[DK * DK];
  __shared__ double mub[DK];
  __shared__ double sv[DK + 1];
  const double* lm = latmat + ((size_t)b * L + a) * (2 * d * d + 2);
  for (int idx = tid; idx < DK * DK; idx += 256) {
    const int i = idx / DK, k = idx - i * DK;
    const bool in = i < d && k < d;
    const int src = in ? i * d + k : 0;
    const double pv = lm[src], ev = lm[d * d + 2 + src];
    Pa[idx] = in ? pv : 0.0; Ea[idx] = in ? ev : 0.0;
  }
  if (tid < DK) mub[tid] = tid < d ? (double)mu[(size_t)b * d + tid] : 0.0;
  if (a == 0 && tid < d) mu64[(size_t)b * d + tid] = (double)mu[(size_t)b * d + tid];   // for mm_route.hip (the Q stage gets no mu)
  __syncthreads();
  const double lognorm = lm[d * d];
  // d <= 8: the two SYMMETRIC matrices live in registers as their upper triangles (2 x 36 doubles at d = 8), broadcast
  // once per workgroup with v_readlane; a quadratic form is then sum_i z_i (M_ii z_i + 2 sum_{k>i} M_ik z_k):
  // 44 register FMAs instead of 72 with one broadcast LDS read each (the loop was bound by those reads)
  constexpr bool QREG = DK <= 8;
  constexpr int NT = DK * (DK + 1) / 2;
  double Pr[QREG ? NT : 1], Er[QREG ? NT : 1];
  if constexpr (QREG) {
    const int ln = tid & 63;
    const double pmine = ln < DK * DK ? Pa[ln] : 0.0, emine = ln < DK * DK ? Ea[ln] : 0.0;
    const int plo = __double2loint(pmine), phi = __double2hiint(pmine);
    const int elo = __double2loint(emine), ehi = __double2hiint(emine);
    int t = 0;
#pragma unroll
    for (int i = 0; i < DK; ++i)
#pragma unroll
      for (int k = i; k < DK; ++k, ++t) {
        double pv = __hiloint2double(__builtin_amdgcn_readlane(phi, i * DK + k), __builtin_amdgcn_readlane(plo, i * DK + k));
        double ev = __hiloint2double(__builtin_amdgcn_readlane(ehi, i * DK + k), __builtin_amdgcn_readlane(elo, i * DK + k));
        asm volatile("" : "+v"(pv), "+v"(ev));                 // vector registers (see k_pairvec_reg)
        Pr[QREG ? t : 0] = pv; Er[QREG ? t : 0] = ev;
      }
  }
  double acc_f = 0.0;
  double acc_s[DK];
#pragma unroll
  for (int k = 0; k < DK; ++k) acc_s[k] = 0.0;
  double* wb64 = w64 + ((size_t)b * L + a) * Mp;
  double* qb64 = q64 + ((size_t)b * L + a) * Mp;
  double* r1 = rho1 + ((size_t)b * L + a) * Mp;
  double* lqb = lq + ((size_t)b * L + a) * Mp;
  // two operand sets: the next chunk's loads (dimension-major table: coalesced over m; index-clamped, so unconditional) are in
  // flight while this chunk is computed -- the loop used to issue its 8 + 1 loads and wait for them in every one of its 8 chunks
  auto loadz = [&](int m, double (&z)[DK], double& bt) {
    const int mc = m < M ? m : M - 1;
#pragma unroll
    for (int k = 0; k < DK; ++k) z[k] = Zt64[((size_t)a * d + (k < d ? k : 0)) * Mp + mc];
    bt = beta64[(size_t)a * M + mc];
  };
  auto chunk = [&](int m, double (&z)[DK], double bt) __attribute__((always_inline)) {
    double wv = 0.0, qv = 0.0, rv = 0.0, lqv = -1.0e30;
    if (m < M) {
#pragma unroll
      for (int k = 0; k < DK; ++k) {
        asm volatile("" : "+v"(z[k]));
        z[k] = (k < d) ? z[k] - mub[k] : 0.0;
      }
      double maha = 0.0;
      // opaque copies of the LDS pointers: hoisted out of the row loop, the two matrices would take
      // 4 DK^2 registers (346 VGPRs at d = 8, scratch beyond)
      typedef const __attribute__((address_space(3))) double* lds_cptr;     // stays a ds_read (not flat) access
      lds_cptr Pl = (lds_cptr)Pa; lds_cptr El = (lds_cptr)Ea;
      asm volatile("" : "+v"(Pl), "+v"(El));
      auto row = [&](int i) {
        double t = 0.0, te = 0.0;
#pragma unroll
        for (int k = 0; k < DK; ++k) { t = fma(Pl[i * DK + k], z[k], t); te = fma(El[i * DK + k], z[k], te); }
        maha = fma(z[i], t, maha);
        rv = fma(z[i], te, rv);                              // zeta^T E_a zeta
      };
      if constexpr (QREG) {
        int t = 0;
#pragma unroll
        for (int i = 0; i < DK; ++i) {
          const double pd = Pr[QREG ? t : 0] * z[i], ed = Er[QREG ? t : 0] * z[i];
          ++t;
          double tp = 0.0, te = 0.0;
#pragma unroll
          for (int k = i + 1; k < DK; ++k, ++t) { tp = fma(Pr[QREG ? t : 0], z[k], tp); te = fma(Er[QREG ? t : 0], z[k], te); }
          maha = fma(z[i], fma(2.0, tp, pd), maha);
          rv = fma(z[i], fma(2.0, te, ed), rv);                // zeta^T E_a zeta
        }
      } else {
#pragma unroll 2
        for (int i = 0; i < DK; ++i) row(i);
      }
      lqv = lognorm - 0.5 * maha;
      qv = exp(lqv);
      wv = bt * qv;
      acc_f += wv;
#pragma unroll
      for (int k = 0; k < DK; ++k) acc_s[k] += wv * z[k];
      if (q_out) q_out[((size_t)b * L + a) * M + perm[(size_t)a * Mp + m]] = (T)qv;   // the caller's order (MMModelLayout::perm)
    }
    wb64[m] = wv;
    qb64[m] = qv;
    r1[m] = rv;
    lqb[m] = lqv;
  };
  {
    double zA[DK], zB[DK], bA, bB;
    loadz(tid, zA, bA);
    for (int m = tid; m < Mp; m += 512) {
      loadz(m + 256 < Mp ? m + 256 : m, zB, bB);
      chunk(m, zA, bA);
      if (m + 256 >= Mp) break;
      loadz(m + 512 < Mp ? m + 512 : m + 256, zA, bA);
      chunk(m + 256, zB, bB);
    }
  }
  // the DK + 1 workgroup sums together: wave butterflies, one LDS stage, one barrier (fixed order: reproducible)
  __shared__ double red9[4][DK + 1];
  {
    double v[DK + 1];
#pragma unroll
    for (int k = 0; k < DK; ++k) v[k] = acc_s[k];
    v[DK] = acc_f;
#pragma unroll
    for (int k = 0; k <= DK; ++k) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    }
    if ((tid & 63) == 0) {
#pragma unroll
      for (int k = 0; k <= DK; ++k) red9[tid >> 6][k] = v[k];
    }
  }
  __syncthreads();
  if (tid <= DK) sv[tid] = (red9[0][tid] + red9[1][tid]) + (red9[2][tid] + red9[3][tid]);
  __syncthreads();
  if (tid == 0) { f1[(size_t)b * L + a] = (T)(sv[DK] + meanc[a]); f1raw[(size_t)b * L + a] = sv[DK]; }
  if (tid < d) {
    // Sigma^-1 Cov(x, f_a) = P_a sum_i w_i (z_i - mu)      (models.py:263-277)
    double s = 0.0;
    for (int k = 0; k < d; ++k) s += Pa[tid * DK + k] * sv[k];
    cross[((size_t)b * d + tid) * L + a] = (T)s;
  }
}

// ---------------------------------------------------------------------------------------------
// k_pairvec: streamed operands of the reduce for d > 8 (d <= 8: k_pairvec_reg below).  grid (nsplit, P, B)
// ---------------------------------------------------------------------------------------------
template <typename T, int DK>
__global__ __launch_bounds__(256) void k_pairvec(const double* __restrict__ Zt64, const double* __restrict__ zbar,
                                                 const double* __restrict__ ls2, int L, int M, int Mp, int d, int P,
                                                 const T* __restrict__ mu, const double* __restrict__ pairmat,
                                                 const double* __restrict__ rho1,
                                                 double* __restrict__ rowD, double* __restrict__ colD,
                                                 T* __restrict__ rowO, T* __restrict__ colO,
                                                 const double* __restrict__ w64, double* __restrict__ whR,
                                                 double* __restrict__ whC, unsigned int* __restrict__ amax,
                                                 const double* __restrict__ q64, double* __restrict__ qhR,
                                                 double* __restrict__ qhC, int with_unc, int nblk,
                                                 const double* __restrict__ lq, const double* __restrict__ beta64,
                                                 const double* __restrict__ zmax2, unsigned short* __restrict__ /* wsp: d <= 8 only */,
                                                 unsigned char* __restrict__ /* gflag */, float* __restrict__ /* gmax2 */, int /* allow */, int p0) {
  static_assert(DK > 8, "d <= 8 takes k_pairvec_reg");
  // a workgroup owns the 256-row chunks blockIdx.x, blockIdx.x + gridDim.x, ... of one (b, pair): the
  // pair's matrix is fetched once per workgroup, not once per chunk
  const int p = p0 + (int)blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  int a, a2;
  mm_decode_pair(p, L, a, a2);
  // With zeta = z - mu, A_i = G^T zeta_i (row side, latent a), g_j = G zeta'_j (column side, latent a'):
  //   zeta_i^T D_row zeta_i   = rho1_a[i]  - sum_k zeta_ik  A_ik Lam_a',k / Lam_a,k
  //   zeta'_j^T D_col zeta'_j = rho1_a'[j] - sum_k zeta'_jk g_jk Lam_a,k  / Lam_a',k
  // (D_row = E_a - Lam_a^-1 T Lam_a^-1, T Lam_a^-1 zeta = Lam_a' A; rho1 = zeta^T E zeta comes from k_qvec),
  // so only G is needed per pair.  As in k_pairvec_reg the SYMMETRIC T = Lam_a G Lam_a' is used on scaled vectors
  // (s = zeta / Lam_a, s' = zeta' / Lam_a'; u = T s, v = T s'; A_i = u_i / Lam_a',i, g_i = v_i / Lam_a,i): one row of
  // T read from LDS feeds both mat-vecs, where G needed a row for g and a (strided) column for A -- the loop was
  // bound by those broadcast reads (C4 shard: 289 LDS instructions per 256-row chunk and wave).  In registers T would
  // take DK (DK + 1) VGPRs, and a prefetched second operand set halves the occupancy at these d: loads are in place.
  const double* r1a = rho1 + ((size_t)b * L + a) * Mp;
  const double* r1b = rho1 + ((size_t)b * L + a2) * Mp;
  __shared__ __align__(16) double Ts[DK * DK];
  // mu | 1/Lam_a | 1/Lam_a' | (mu - zbar_a') / Lam_a' | (mu - zbar_a) / Lam_a | t0 = T (mu - zbar_a) / Lam_a | c00 = t0 . [3]
  __shared__ double vecs[7][DK];
  __shared__ int rcs;
  const double* pm = pairmat + ((size_t)b * P + p) * (d * d + 1);
  for (int idx = tid; idx < DK * DK; idx += 256) {
    const int i = idx / DK, k = idx - i * DK;
    const bool in = i < d && k < d;
    Ts[idx] = in ? pm[i * d + k] * ls2[a * d + i] * ls2[a2 * d + k] : 0.0;
  }
  if (tid < DK) {
    const int k = tid < d ? tid : 0;
    const double mv = (double)mu[(size_t)b * d + k];
    const double la = ls2[a * d + k], lb = ls2[a2 * d + k];
    vecs[0][tid] = tid < d ? mv : 0.0;
    vecs[1][tid] = tid < d ? 1.0 / la : 0.0;
    vecs[2][tid] = tid < d ? 1.0 / lb : 0.0;
    vecs[3][tid] = tid < d ? (mv - zbar[a2 * d + k]) / lb : 0.0;   // the A operand is centred at zbar_a', not at mu_b
    vecs[4][tid] = tid < d ? (mv - zbar[a * d + k]) / la : 0.0;
  }
  __syncthreads();
  const double cst = pm[d * d];
  const bool diag = p < L;
  const int Po = P - L;
  // f32 off-diagonal format (mm_mfma.hip): see k_pairvec_reg
  const bool f32off = !diag && sizeof(T) == 4;
  if (f32off) {
    // rows recentred at zbar_a (mm_mono.h; as in k_pairvec_reg)
    if (tid < DK) {
      double t = 0.0;
      for (int k = 0; k < DK; ++k) t = fma(Ts[tid * DK + k], vecs[4][k], t);
      vecs[5][tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
      double sa2 = 0.0, c00 = 0.0;
      for (int i = 0; i < DK; ++i) { const double sa = vecs[5][i] * vecs[2][i]; sa2 = fma(sa, sa, sa2); c00 = fma(vecs[5][i], vecs[3][i], c00); }
      const bool rc = sa2 * zmax2[a2] <= MM_RECENTRE_CMAX * MM_RECENTRE_CMAX;
      rcs = rc ? 1 : 0;
      vecs[6][0] = rc ? c00 : 0.0;
      if (!rc) for (int i = 0; i < DK; ++i) vecs[5][i] = 0.0;
    }
    __syncthreads();
  }
  const bool recentred = f32off ? rcs != 0 : true;
  T* rO = rowO + ((size_t)b * Po + (f32off ? p - L : 0)) * (size_t)(d + 1) * Mp;
  T* cO = colO + ((size_t)b * Po + (f32off ? p - L : 0)) * Mp;
  double* hR = whR + ((size_t)b * Po + (f32off ? p - L : 0)) * Mp;
  double* hC = whC + ((size_t)b * Po + (f32off ? p - L : 0)) * Mp;
  // diagonal pairs (p < L) stream f64 operands, off-diagonal pairs of the f64 mode likewise
  double* raD = rowD + ((size_t)b * L + (diag ? p : 0)) * Mp;
  double* cbD = colD + ((size_t)b * L + (diag ? p : 0)) * (size_t)(d + 1) * Mp;
  T* raO = rowO + ((size_t)b * Po + (diag ? 0 : p - L)) * Mp;
  T* cbO = colO + ((size_t)b * Po + (diag ? 0 : p - L)) * (size_t)(d + 1) * Mp;
  float a2max = 0.0f;                                         // max_i |A_i|^2 over this thread's rows (f32 off-diagonal pairs)
  for (int mblk = blockIdx.x; mblk < nblk; mblk += gridDim.x) {
    const int m = mblk * 256 + tid;
    if (m >= Mp) continue;
    // (laundered LDS pointers: hoisted out of the chunk loop, the five vectors alone take 10 DK VGPRs)
    typedef const __attribute__((address_space(3))) double* lds_cptr;
    lds_cptr vl = (lds_cptr)&vecs[0][0], tl = (lds_cptr)Ts;
    asm volatile("" : "+v"(vl), "+v"(tl));
    // scaled centred inducing inputs of row / column index m (zero beyond d -- the scale vectors are -- or M)
    double zr[DK], zc[DK];
#pragma unroll
    for (int k = 0; k < DK; ++k) {
      const int kk = k < d ? k : 0;
      const double vr = Zt64[((size_t)a * d + kk) * Mp + m];     // dimension-major: coalesced over m
      const double vc = Zt64[((size_t)a2 * d + kk) * Mp + m];
      zr[k] = (m < M) ? (vr - vl[k]) * vl[DK + k] : 0.0;
      zc[k] = (m < M) ? (vc - vl[k]) * vl[2 * DK + k] : 0.0;
    }
    const double r1r = r1a[m], r1c = r1b[m];
    // weights in the log domain: beta exp(log q + ...) as ONE exponential (q alone underflows where the rest overflows)
    const int mc = m < M ? m : M - 1;
    const double lqr = lq[((size_t)b * L + a) * Mp + m], lqc = lq[((size_t)b * L + a2) * Mp + m];
    const double btr = beta64[(size_t)a * M + mc], btc = beta64[(size_t)a2 * M + mc];
    double tA = 0.0, tg = 0.0, corrA = 0.0, corrg = 0.0;
    double asq = 0.0, cj = vl[6 * DK];
#pragma unroll 2      // not fully: the compiler would hoist all the LDS reads (> 256 VGPRs)
    for (int i = 0; i < DK; ++i) {
      double u = 0.0, v = 0.0;
#pragma unroll
      for (int k = 0; k < DK; ++k) {
        const double t = tl[i * DK + k];
        u = fma(t, zr[k], u);
        v = fma(t, zc[k], v);
      }
      tA = fma(zr[i], u, tA);
      tg = fma(zc[i], v, tg);
      corrA = fma(vl[3 * DK + i], u, corrA);
      corrg = fma(vl[4 * DK + i], v, corrg);
      // f32 off-diagonal pairs: rows recentred at zbar_a (t0 is zero otherwise and for the other formats)
      const double t0i = f32off ? vl[5 * DK + i] : 0.0;
      cj = fma(t0i, zc[i], cj);
      const double av = (u + t0i) * vl[2 * DK + i], gv = v * vl[DK + i];
      asq = fma(av, av, asq);
      if (i < d) {
        if (f32off) rO[(size_t)i * Mp + m] = (T)av;           // zero for the padding rows m >= M
        else if (diag) cbD[(size_t)i * Mp + m] = gv;
        else cbO[(size_t)i * Mp + m] = (T)gv;
      }
    }
    const double rho_q = (m < M) ? r1r - tA : 0.0;              // zeta_i^T D_row zeta_i
    const double gam_q = (m < M) ? r1c - tg : 0.0;              // zeta'_j^T D_col zeta'_j
    if (f32off) {
      a2max = fmaxf(a2max, (float)asq * 1.000001f);           // rounded up: the bound must not be under-estimated
      double whr = 0.0, whc = 0.0;
      if (m < M) {
        whr = btr * exp(fmin(lqr - 0.5 * rho_q + cst - corrA, (double)MM_EXP_CAP_F32));
        whc = btc * exp(fmin(lqc - 0.5 * gam_q - cj, (double)MM_EXP_CAP_F32));
      }
      rO[(size_t)d * Mp + m] = (T)whr;
      cO[m] = (T)whc;
      hR[m] = whr;
      hC[m] = whc;
    } else {
      // gamma' = gamma + const - (mu - zbar_a)^T g   so that   delta = rho_i + gamma'_j + zc_i . g_j
      const double rowv = -0.5 * rho_q;
      const double colv = (m < M) ? -0.5 * gam_q + cst - corrg : 0.0;
      if (diag) {
        raD[m] = rowv; cbD[(size_t)d * Mp + m] = colv;
        // factored weights of the f64 MFMA reduce (mm_f64.hip): e^{delta_ij} = e^{rho_i} e^{gamma'_j} e^{zc_i . g_j};
        // u = q with model uncertainty (the fused sum runs over q_i q_j D_ij e^{delta}), w without
        const size_t qi = ((size_t)b * L + p) * Mp + m;
        const double u = (m < M) ? (with_unc ? 1.0 : btr) : 0.0;
        qhR[qi] = u * exp(fmin(lqr + rowv, MM_EXP_CAP_F64));
        qhC[qi] = u * exp(fmin(lqr + colv, MM_EXP_CAP_F64));
      } else { raO[m] = (T)rowv; cbO[(size_t)d * Mp + m] = (T)colv; }
    }
  }
  if (f32off && amax) {
    // one atomicMax per wave: non-negative floats are ordered like their bit patterns (max is order independent)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a2max = fmaxf(a2max, __shfl_down(a2max, off, 64));
    if ((tid & 63) == 0) atomicMax(amax + (size_t)b * Po + (p - L), recentred ? __float_as_uint(a2max) : MM_AMAX_NOT_RECENTRED);
  }
}

// ---------------------------------------------------------------------------------------------
// k_pairvec_reg: k_pairvec for d <= 8, the pair's matrix in REGISTERS.  grid (nsplit, P, B), 256 threads.
//
// G = Lam_a^-1 T Lam_a'^-1 with T symmetric: a thread keeps the DK (DK + 1) / 2 distinct entries of T = Lam_a G Lam_a'
// (72 VGPRs at d = 8; G itself would take 128, which the compiler answered by re-broadcasting it with v_readlane in
// every chunk) and works on SCALED vectors, s = zeta / Lam_a (row side), s' = zeta' / Lam_a' (column side):
//     u = T s,  v = T s';    A_i = u_i / Lam_a',i,  g_i = v_i / Lam_a,i,
//     zeta^T (Lam_a^-1 T Lam_a^-1) zeta = s . u,     zeta'^T (Lam_a'^-1 T Lam_a'^-1) zeta' = s' . v = s'^T T s',
// so a quadratic form whose vector is not an output costs 36 products instead of 64 + 8.  The three modes (diagonal
// pair; f32 off-diagonal pair; f64 off-diagonal pair) are straight-line code behind a workgroup-uniform branch;
// the chunk loop is unrolled twice over two operand sets (the next chunk's loads are in flight, no register copies);
// the per-latent vectors stay in LDS (laundered pointer: hoisted, the 5 x 8 doubles cost 80 VGPRs).
// ---------------------------------------------------------------------------------------------
// MODE 0: diagonal pair; 1: f32 off-diagonal pair; 2: f64 off-diagonal pair.  One instantiation per mode behind the
// kernel's workgroup-uniform branch: operand pointers and temporaries of the other modes are never live.
template <typename T, int DK, int MODE>
__device__ __forceinline__ void mm_pairvec_reg_body(const double* __restrict__ Zt64, const double* __restrict__ zbar,
                                                    const double* __restrict__ ls2, int L, int M, int Mp, int d, int P,
                                                    const T* __restrict__ mu, const double* __restrict__ pairmat,
                                                    const double* __restrict__ rho1,
                                                    double* __restrict__ rowD, double* __restrict__ colD,
                                                    T* __restrict__ rowO, T* __restrict__ colO,
                                                    const double* __restrict__ w64, double* __restrict__ whR,
                                                    double* __restrict__ whC, unsigned int* __restrict__ amax,
                                                    const double* __restrict__ q64, double* __restrict__ qhR,
                                                    double* __restrict__ qhC, int with_unc, int nblk, int a, int a2,
                                                    double (*vecs)[DK], const double* __restrict__ lq,
                                                    const double* __restrict__ beta64, int p, const double* __restrict__ zmax2,
                                                    unsigned short* __restrict__ wsp, unsigned char* __restrict__ gflag, float* __restrict__ gmax2,
                                                    int allow) {
  constexpr bool diag = MODE == 0;
  const int b = blockIdx.z, tid = threadIdx.x;
  const int Po = P - L;
  const double* zA = Zt64 + (size_t)a * d * Mp;
  const double* zB = Zt64 + (size_t)a2 * d * Mp;
  const double* r1a = rho1 + ((size_t)b * L + a) * Mp;
  const double* r1b = rho1 + ((size_t)b * L + a2) * Mp;
  // weights in the log domain: every factored weight is beta exp(log q + ...) as ONE exponential -- q alone underflows to 0
  // where e^{rho} overflows (lengthscales far below the state's distance to the inducing point), and 0 x inf is a NaN
  const double* wa = lq + ((size_t)b * L + a) * Mp;        // log q of the row latent
  const double* wb = lq + ((size_t)b * L + a2) * Mp;       // ... of the column latent
  const double* ba = beta64 + (size_t)a * M;
  const double* bb = beta64 + (size_t)a2 * M;
  struct Ops { double zr[DK], zc[diag ? 1 : DK], r1r, r1c, wr, wc, br, bc; };
  auto load_ops = [&](int mblk, Ops& o) {
    int m = mblk * 256 + tid;
    m = m < Mp ? m : Mp - 1;
#pragma unroll
    for (int k = 0; k < DK; ++k) {
      const unsigned off = (unsigned)((k < d ? k : 0) * Mp + m);   // index-clamped: the scale vectors are zero beyond d
      o.zr[k] = zA[off];                                     // dimension-major: coalesced over m
      if constexpr (!diag) o.zc[k] = zB[off];
    }
    o.r1r = r1a[m];
    o.wr = wa[m];
    const int mb = m < M ? m : M - 1;
    o.br = ba[mb];
    if constexpr (!diag) { o.r1c = r1b[m]; o.wc = wb[m]; o.bc = bb[mb]; }
  };
  Ops oA, oB;
  load_ops(blockIdx.x, oA);
  const double* pm = pairmat + ((size_t)b * P + p) * (d * d + 1);
  constexpr int NT = DK * (DK + 1) / 2;
  double Tr[NT];
  auto tsym = [](int i, int k) { return i <= k ? i * DK - i * (i - 1) / 2 + (k - i) : k * DK - k * (k - 1) / 2 + (i - k); };
  {
    // lane l of every wave loads padded entry l (one coalesced load), then the values are broadcast with v_readlane
    // and pinned to vector registers
    const int ln = tid & 63, li = ln / DK, lk = ln - li * DK;
    const bool lin = li < d && lk < d && ln < DK * DK;
    const double mine = lin ? pm[li * d + lk] * ls2[a * d + li] * ls2[a2 * d + lk] : 0.0;
    const int mlo = __double2loint(mine), mhi = __double2hiint(mine);
#pragma unroll
    for (int i = 0; i < DK; ++i)
#pragma unroll
      for (int k = i; k < DK; ++k) {
        double t = __hiloint2double(__builtin_amdgcn_readlane(mhi, i * DK + k), __builtin_amdgcn_readlane(mlo, i * DK + k));
        asm volatile("" : "+v"(t));
        Tr[tsym(i, k)] = t;
      }
  }
  if (tid < DK) {
    const int k = tid < d ? tid : 0;
    const double mv = (double)mu[(size_t)b * d + k];
    const double la = ls2[a * d + k], lb = ls2[a2 * d + k];
    vecs[0][tid] = tid < d ? mv : 0.0;
    vecs[1][tid] = tid < d ? 1.0 / la : 0.0;
    vecs[2][tid] = tid < d ? 1.0 / lb : 0.0;
    vecs[3][tid] = tid < d ? (mv - zbar[a2 * d + k]) / lb : 0.0;   // the A operand is centred at zbar_a', not at mu_b
    vecs[4][tid] = tid < d ? (mv - zbar[a * d + k]) / la : 0.0;
  }
  __syncthreads();
  bool recentred = true;
  if constexpr (MODE == 0) {
    // DIAGONAL pairs: the A operand of the sweep is zeta_i = z_i - zbar_a (model constant); the column operand was g_j = G (z_j - mu),
    // so b_ij = zeta_i . g_j kept the state's offset from the centroid on the column side.  Recentred there too,
    //     g'_j = G zeta_j = g_j + h,   h = G (mu - zbar_a) = t0 / Lam_a,   delta_ij = (rho_i - zeta_i . h) + gamma'_j + zeta_i . g'_j:
    // the row-only term joins rho_i (one more term in the row weight's single exponential), b'_ij = zeta_i G zeta_j is smaller
    // on average (lower range tiers in the sweeps; with the pack's norm order the mean tier degree of the C3 rollout falls from
    // 9.8 to 8.1 FMAs per entry) and every consumer of (rowD, colD, qhR, qhC) evaluates the same delta_ij.  Taken where the
    // moved exponent is harmless, |h| max|zeta| <= MM_RECENTRE_CMAX, else the operands stay as they were.
    double h2 = 0.0, t0[DK];
#pragma unroll
    for (int i = 0; i < DK; ++i) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < DK; ++k) t = fma(Tr[tsym(i, k)], vecs[4][k], t);
      t0[i] = t;
      const double hi = t * vecs[1][i];
      h2 = fma(hi, hi, h2);
    }
    recentred = h2 * zmax2[a] <= MM_RECENTRE_CMAX * MM_RECENTRE_CMAX;
    if (tid == 0) {
#pragma unroll
      for (int i = 0; i < DK; ++i) vecs[5][i] = recentred ? t0[i] : 0.0;
    }
    __syncthreads();
  }
  if constexpr (MODE == 1) {
    // rows recentred at zbar_a (mm_mono.h): t0 = T (mu - zbar_a) / Lam_a, A_i = (u_i + t0_i) / Lam_a',i, and the column weight
    // takes e^{-c_j}, c_j = sum_k t0_k zc'_jk / Lam_a',k.  Every thread forms t0 (workgroup-uniform), thread 0 publishes it.
    double sa2 = 0.0, c00 = 0.0, t0[DK];
#pragma unroll
    for (int i = 0; i < DK; ++i) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < DK; ++k) t = fma(Tr[tsym(i, k)], vecs[4][k], t);
      t0[i] = t;
      const double sa = t * vecs[2][i];
      sa2 = fma(sa, sa, sa2);
      c00 = fma(t, vecs[3][i], c00);
    }
    recentred = sa2 * zmax2[a2] <= MM_RECENTRE_CMAX * MM_RECENTRE_CMAX;
    if (tid == 0) {
#pragma unroll
      for (int i = 0; i < DK; ++i) vecs[5][i] = recentred ? t0[i] : 0.0;
      vecs[6][0] = recentred ? c00 : 0.0;
    }
    __syncthreads();
  }
  const double cst = pm[d * d];
  float a2max = 0.0f;                                         // max_i |A_i|^2 over this thread's rows (f32 off-diagonal pairs)
  float a2cmax = 0.0f, a2in = 0.0f;                           // ... over the rows of COLLAPSED groups alone (mm_mono.h: row-group collapse)
  const bool cancoll = MODE == 1 && allow && recentred;       // no group is collapsed where the caller rules it out or the rows stay at mu
  const double zm2c = MODE == 1 ? zmax2[a2] : 0.0;
  typedef const __attribute__((address_space(3))) double* lds_cptr;
  // rows i >= d of a (d + 1)-row operand do not exist: their (zero) value goes to row d, which the chunk's last
  // store overwrites -- no branch per row, and no extra store at all when d == DK
  auto rowi = [&](int i) { return (unsigned)((i < d ? i : d) * Mp); };
  const unsigned rowd = (unsigned)(d * Mp);

  auto body = [&](int mblk, const Ops& o) {
    const int m = mblk * 256 + tid;
    if (m >= Mp) return;
    lds_cptr vl = (lds_cptr)&vecs[0][0];
    asm volatile("" : "+v"(vl));
    const bool live = m < M;
    if constexpr (MODE == 0) {
      double* cbD = colD + ((size_t)b * L + p) * (size_t)(d + 1) * Mp;
      double* raD = rowD + ((size_t)b * L + p) * Mp;
      double* qR = qhR + ((size_t)b * L + p) * Mp;
      double* qC = qhC + ((size_t)b * L + p) * Mp;
      if (live) {
        double sv[DK];
#pragma unroll
        for (int k = 0; k < DK; ++k) sv[k] = (o.zr[k] - vl[k]) * vl[DK + k];
        double tq = 0.0, corr = 0.0, zh = 0.0;
#pragma unroll
        for (int i = 0; i < DK; ++i) {
          double u = 0.0;
#pragma unroll
          for (int k = 0; k < DK; ++k) u = fma(Tr[tsym(i, k)], sv[k], u);
          tq = fma(sv[i], u, tq);
          corr = fma(vl[4 * DK + i], u, corr);               // (mu - zbar_a) . g
          zh = fma(sv[i] + vl[4 * DK + i], vl[5 * DK + i], zh);   // zeta . h  (t0 = 0 where the pair stays as it was)
          cbD[rowi(i) + m] = (u + vl[5 * DK + i]) * vl[DK + i];   // g'_i = (G zeta)_i
        }
        // gamma' = gamma + const - (mu - zbar_a)^T g   so that   delta = rho_i + gamma'_j + zc_i . g_j  (= rho_i - zeta_i . h + gamma'_j + zeta_i . g'_j)
        const double row0 = -0.5 * (o.r1r - tq);
        const double rowv = row0 - zh;
        const double colv = row0 + cst - corr;
        // factored weights of the f64 MFMA reduce (mm_f64.hip): e^{delta_ij} = e^{rho_i} e^{gamma'_j} e^{zc_i . g_j};
        // u = q with model uncertainty (the fused sum runs over q_i q_j D_ij e^{delta}), w without
        const double uw = with_unc ? 1.0 : o.br;             // the weight of the fused sum is q with model uncertainty, else w = beta q
        raD[m] = rowv; cbD[rowd + m] = colv;
        qR[m] = uw * exp(fmin(o.wr + rowv, MM_EXP_CAP_F64)); qC[m] = uw * exp(fmin(o.wr + colv, MM_EXP_CAP_F64));
      } else {
#pragma unroll
        for (int i = 0; i < DK; ++i) cbD[rowi(i) + m] = 0.0;
        raD[m] = 0.0; cbD[rowd + m] = 0.0; qR[m] = 0.0; qC[m] = 0.0;
      }
    } else if constexpr (MODE == 1) {
      // f32 off-diagonal format (mm_mfma.hip).  With b_ij = A_i . zc^{a'}_j,
      //   delta_ij = rho'_i + gamma_j + b_ij,  rho'_i = rho_i + const - A_i . (mu - zbar_a'),
      // exp(delta) - 1 = e^{rho'_i} e^{gamma_j} (expm1(b_ij) + 1) - 1: the M x M tile is a pure bilinear form in
      // what_i = w_i e^{rho'_i}, what'_j = w'_j e^{gamma_j}; the f64 weights feed the moment GEMM (mm_moments.hip).
      T* rO = rowO + ((size_t)b * Po + (p - L)) * (size_t)(d + 1) * Mp;
      T* cO = colO + ((size_t)b * Po + (p - L)) * Mp;
      double* hR = whR + ((size_t)b * Po + (p - L)) * Mp;
      double* hC = whC + ((size_t)b * Po + (p - L)) * Mp;
      double whr = 0.0, whc = 0.0;
      bool rowok = true;                                     // (padding rows: A_i = 0)
      if (live) {
        double sr[DK], sc[DK];
        double cj = vl[6 * DK];                              // c_j = t0 . (zc'_j / Lam_a'): the row shift's column factor
#pragma unroll
        for (int k = 0; k < DK; ++k) {
          const double muk = vl[k];
          sr[k] = (o.zr[k] - muk) * vl[DK + k]; sc[k] = (o.zc[k] - muk) * vl[2 * DK + k];
          cj = fma(vl[5 * DK + k], sc[k], cj);
        }
        double tA = 0.0, tg = 0.0, corrA = 0.0, asq = 0.0;
#pragma unroll
        for (int i = 0; i < DK; ++i) {
          double u = 0.0, vh = 0.0;
#pragma unroll
          for (int k = 0; k < DK; ++k) {
            const double t = Tr[tsym(i, k)];
            u = fma(t, sr[k], u);
            if (k > i) vh = fma(t, sc[k], vh);               // s'^T T s' from the upper triangle
          }
          tA = fma(sr[i], u, tA);
          tg = fma(sc[i], fma(2.0, vh, Tr[tsym(i, i)] * sc[i]), tg);
          const double av = (u + vl[5 * DK + i]) * vl[2 * DK + i];   // A_i = G^T (z_i - zbar_a)  (G^T (z_i - mu) if not recentred)
          asq = fma(av, av, asq);
          corrA = fma(vl[3 * DK + i], u, corrA);
          rO[rowi(i) + m] = (T)av;
        }
        const float a2row = (float)asq * 1.000001f;          // rounded up: the bound must not be under-estimated
        a2max = fmaxf(a2max, a2row);
        rowok = mm_collapse_bound2(__float_as_uint(a2row), zm2c) <= MM_COLLAPSE_BOUND2;
        a2in = a2row;
        whr = o.br * exp(fmin(o.wr - 0.5 * (o.r1r - tA) + cst - corrA, (double)MM_EXP_CAP_F32));
        whc = o.bc * exp(fmin(o.wc - 0.5 * (o.r1c - tg) - cj, (double)MM_EXP_CAP_F32));
      } else {
#pragma unroll
        for (int i = 0; i < DK; ++i) rO[rowi(i) + m] = (T)0;  // zero rows: b = 0 in the padding
        a2in = 0.0f;
      }
      rO[rowd + m] = (T)whr; cO[m] = (T)whc; hR[m] = whr; hC[m] = whc;
      if (wsp) {
        // ROW-GROUP COLLAPSE (mm_mono.h): this wave's 64 rows are one group; collapsed when every row's own Cauchy-Schwarz
        // bound is <= 1/2.  (the lanes are converged here: m < Mp is wave-uniform, Mp % 128 == 0)
        const bool inner = cancoll && __all(rowok);
        float g2 = a2in;                                     // the group's max |A_i|^2 (MMWorkspaceLayout::gmax2)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) g2 = fmaxf(g2, __shfl_xor(g2, off, 64));
        if ((tid & 63) == 0) {
          const size_t gi = ((size_t)b * Po + (p - L)) * (size_t)(Mp / MM_GROUP_ROWS) + (m >> 6);
          gflag[gi] = inner ? 1 : 0;
          gmax2[gi] = g2;
        }
        if (inner) a2cmax = fmaxf(a2cmax, a2in);
        // bf16 2-way split of both weights: the A operand of the degree-4/5/6 moment GEMM (mm_moments6.hip), [side][h, m][Mp];
        // the ROW weight of a group that is not collapsed is zero there
        unsigned short* sp = wsp + ((size_t)b * Po + (p - L)) * 4 * (size_t)Mp + m;
        const float fr = inner ? (float)whr : 0.0f, fc = (float)whc;
        const __bf16 rh = (__bf16)fr, ch = (__bf16)fc;
        const __bf16 rm = (__bf16)(fr - (float)rh), cm = (__bf16)(fc - (float)ch);
        sp[0] = __builtin_bit_cast(unsigned short, rh); sp[Mp] = __builtin_bit_cast(unsigned short, rm);
        sp[2 * (size_t)Mp] = __builtin_bit_cast(unsigned short, ch); sp[3 * (size_t)Mp] = __builtin_bit_cast(unsigned short, cm);
      }
    } else {
      // off-diagonal pair of the f64 mode: rho_i (row), g_j and gamma'_j (column)
      T* raO = rowO + ((size_t)b * Po + (p - L)) * Mp;
      T* cbO = colO + ((size_t)b * Po + (p - L)) * (size_t)(d + 1) * Mp;
      if (live) {
        double sr[DK], sc[DK];
#pragma unroll
        for (int k = 0; k < DK; ++k) { const double muk = vl[k]; sr[k] = (o.zr[k] - muk) * vl[DK + k]; sc[k] = (o.zc[k] - muk) * vl[2 * DK + k]; }
        double tA = 0.0, tg = 0.0, corrg = 0.0;
#pragma unroll
        for (int i = 0; i < DK; ++i) {
          double v = 0.0, uh = 0.0;
#pragma unroll
          for (int k = 0; k < DK; ++k) {
            const double t = Tr[tsym(i, k)];
            v = fma(t, sc[k], v);
            if (k > i) uh = fma(t, sr[k], uh);
          }
          tg = fma(sc[i], v, tg);
          tA = fma(sr[i], fma(2.0, uh, Tr[tsym(i, i)] * sr[i]), tA);
          corrg = fma(vl[4 * DK + i], v, corrg);
          cbO[rowi(i) + m] = (T)(v * vl[DK + i]);            // g_i
        }
        raO[m] = (T)(-0.5 * (o.r1r - tA));
        cbO[rowd + m] = (T)(-0.5 * (o.r1c - tg) + cst - corrg);
      } else {
#pragma unroll
        for (int i = 0; i < DK; ++i) cbO[rowi(i) + m] = (T)0;
        raO[m] = (T)0; cbO[rowd + m] = (T)0;
      }
    }
  };

  const int step = (int)gridDim.x;
  for (int mblk = blockIdx.x; mblk < nblk; mblk += 2 * step) {
    // unconditional (clamped) prefetch: past the end it re-reads the current chunk
    load_ops(mblk + step < nblk ? mblk + step : mblk, oB);
    body(mblk, oA);
    if (mblk + step >= nblk) break;
    load_ops(mblk + 2 * step < nblk ? mblk + 2 * step : mblk + step, oA);
    body(mblk + step, oB);
  }
  if (MODE == 1 && amax) {
    // one atomicMax per wave: non-negative floats are ordered like their bit patterns (max is order independent)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a2max = fmaxf(a2max, __shfl_down(a2max, off, 64));
    // rows left centred at mu: marked (mm_mono.h), which also keeps the item out of every collapse predicate
    if ((tid & 63) == 0) atomicMax(amax + (size_t)b * Po + (p - L), recentred ? __float_as_uint(a2max) : MM_AMAX_NOT_RECENTRED);
    // amaxc ([B][Po] right behind amax): the same over the collapsed groups
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a2cmax = fmaxf(a2cmax, __shfl_down(a2cmax, off, 64));
    if ((tid & 63) == 0 && a2cmax > 0.0f) atomicMax(amax + ((size_t)gridDim.z + b) * Po + (p - L), __float_as_uint(a2cmax));
  }
}

template <typename T, int DK>
__global__ __launch_bounds__(256, 2) void k_pairvec_reg(const double* __restrict__ Zt64, const double* __restrict__ zbar,
                                                        const double* __restrict__ ls2, int L, int M, int Mp, int d, int P,
                                                        const T* __restrict__ mu, const double* __restrict__ pairmat,
                                                        const double* __restrict__ rho1,
                                                        double* __restrict__ rowD, double* __restrict__ colD,
                                                        T* __restrict__ rowO, T* __restrict__ colO,
                                                        const double* __restrict__ w64, double* __restrict__ whR,
                                                        double* __restrict__ whC, unsigned int* __restrict__ amax,
                                                        const double* __restrict__ q64, double* __restrict__ qhR,
                                                        double* __restrict__ qhC, int with_unc, int nblk,
                                                        const double* __restrict__ lq, const double* __restrict__ beta64,
                                                        const double* __restrict__ zmax2, unsigned short* __restrict__ wsp,
                                                        unsigned char* __restrict__ gflag, float* __restrict__ gmax2, int allow, int p0) {
  // (grid.y = the pairs [p0, p0 + gridDim.y): the q stage launches the diagonal pairs' operands first -- the diagonal sweep needs
  // nothing else -- and the off-diagonal pairs' on the side stream beside that sweep)
  static_assert(DK <= 8, "register form: d <= 8");
  // mu | 1/Lam_a | 1/Lam_a' | (mu - zbar_a') / Lam_a' | (mu - zbar_a) / Lam_a | t0 = T (mu - zbar_a) / Lam_a | c00 = t0 . [3]
  __shared__ double vecs[7][DK];
  const int p = p0 + (int)blockIdx.y;
  int a, a2;
  mm_decode_pair(p, L, a, a2);
#define MM_PV_BODY(MODE_)                                                                                           \
  mm_pairvec_reg_body<T, DK, MODE_>(Zt64, zbar, ls2, L, M, Mp, d, P, mu, pairmat, rho1, rowD, colD, rowO, colO, w64, \
                                    whR, whC, amax, q64, qhR, qhC, with_unc, nblk, a, a2, vecs, lq, beta64, p, zmax2, wsp, gflag, gmax2, allow)
  if (p < L) MM_PV_BODY(0);
  else if (sizeof(T) == 4) MM_PV_BODY(1);
  else MM_PV_BODY(2);
#undef MM_PV_BODY
}

// ---------------------------------------------------------------------------------------------
// k_qred_generic: portable VALU fused reduce over the pairs [p0, p0 + gridDim.y).
//   f64 instantiation: the diagonal pairs of every mode (with the C-weighted term) and all
//   pairs of the f64 mode; f32 instantiation: off-diagonal pairs (fallback / cross-check of
//   the MFMA kernel).  thread = one column j, workgroup = 256 columns x MM_GEN_ROWS rows of
//   one (b, pair).  grid (nrb * ncb, npairs, B); row/col operand arrays are indexed by the
//   LOCAL pair index, the partial slabs by the global one.
// ---------------------------------------------------------------------------------------------
template <typename T, int DK, bool ROWVEC>
__global__ __launch_bounds__(256) void k_qred_generic(const T* __restrict__ Zc, int Kz, const double* __restrict__ Cm,
                                                      int L, int Mp, int d, int P, int NS, int ncb, int p0,
                                                      const T* __restrict__ w, const T* __restrict__ q,
                                                      const T* __restrict__ rowA, const T* __restrict__ colB,
                                                      double* __restrict__ partB, double* __restrict__ partC,
                                                      const unsigned char* __restrict__ gflag, const unsigned int* __restrict__ amaxc,
                                                      const double* __restrict__ zmax2) {
  const int cbk = blockIdx.x % ncb, rbk = blockIdx.x / ncb;
  const int lp = blockIdx.y, np = gridDim.y, p = p0 + lp, b = blockIdx.z, tid = threadIdx.x;
  int a, a2;
  mm_decode_pair(p, L, a, a2);
  const int j = cbk * MM_GEN_COLS + tid;
  const bool jv = j < Mp;
  const int jj = jv ? j : 0;
  // ROWVEC (f32 off-diagonal layout): the row side carries the vector A_i and rho'_i, the column
  // side is the centred inducing input zc_j of latent a' plus gamma_j.  Otherwise the column side
  // carries g_j, gamma'_j and the row side is zc_i of latent a plus rho_i.
  const T* cb = ROWVEC ? colB + ((size_t)b * np + lp) * Mp
                       : colB + ((size_t)b * np + lp) * (size_t)(d + 1) * Mp;
  T g[DK];
#pragma unroll
  for (int k = 0; k < DK; ++k)
    g[k] = (k < d) ? (ROWVEC ? Zc[((size_t)a2 * Mp + jj) * Kz + k] : cb[(size_t)k * Mp + jj]) : (T)0;
  // ROWVEC: pure bilinear tile b_ij = A_i . zc_j with the factored weights what_i (row A, entry d)
  // and what'_j (colB), see k_pairvec / mm_mfma.hip; else delta = rho_i + gamma'_j + zc_i . g_j
  const T gam = ROWVEC ? (T)0 : cb[(size_t)d * Mp + jj];
  const T wj = jv ? (ROWVEC ? cb[jj] : w[((size_t)b * L + a2) * Mp + jj]) : (T)0;
  const bool withC = (Cm != nullptr) && (a == a2);
  const T qj = (withC && jv) ? q[((size_t)b * L + a2) * Mp + jj] : (T)0;
  const T* ra = ROWVEC ? rowA + ((size_t)b * np + lp) * (size_t)(d + 1) * Mp
                       : rowA + ((size_t)b * np + lp) * Mp;
  const T* wr = w + ((size_t)b * L + a) * Mp;
  const T* qr = q + ((size_t)b * L + a) * Mp;
  const T* zrow = Zc + (size_t)a * Mp * Kz;
  const int i0 = rbk * MM_GEN_ROWS;
  const int i1 = (i0 + MM_GEN_ROWS < Mp) ? i0 + MM_GEN_ROWS : Mp;
  T accB = (T)0;
  double sumB = 0.0, sumC = 0.0;
  // ROWVEC: the rows of a COLLAPSED row group (mm_mono.h) have p6 (mm_common.h) of the remainder in the moments (mm_moments.hip,
  // mm_moments6.hip); the other rows of an item with such a group the cubic term C0 b^3 alone
  const unsigned char* gf = (ROWVEC && gflag != nullptr && zmax2 != nullptr) ? gflag + ((size_t)b * np + lp) * (size_t)(Mp / MM_GROUP_ROWS) : nullptr;
  const bool icoll = gf != nullptr && mm_item_collapsed(amaxc[(size_t)b * np + lp]);
  for (int i = i0; i < i1; ++i) {
    const bool coll = icoll && gf[i >> 6] != 0;
    T delta = ROWVEC ? (T)0 : ra[i] + gam;
#pragma unroll
    for (int k = 0; k < DK; ++k)
      if (k < d) delta += (ROWVEC ? ra[(size_t)k * Mp + i] : zrow[(size_t)i * Kz + k]) * g[k];
    // ROWVEC (f32 off-diagonal pairs): only the remainder expm1(b) - b - b^2/2 is reduced here, the
    // rest comes from the f64 weight moments (k_spoly); evaluated in f64 (portable cross-check kernel)
    const double dd = fmin((double)delta, sizeof(T) == 8 ? MM_EXP_CAP_F64 : (double)MM_EXP_CAP_F32);   // (mm_common.h: exponent caps)
    const T e = ROWVEC ? (T)(expm1(dd) - dd - 0.5 * dd * dd - (coll ? MM_C6_POLY_F64(dd) : (icoll ? (double)MM_C6_C0 * dd * dd * dd : 0.0)))
                       : (T)expm1(dd);
    accB += (ROWVEC ? ra[(size_t)d * Mp + i] : wr[i]) * e;
    if (withC) {
      const double cij = Cm[((size_t)a * Mp + i) * Mp + jj];
      const double qi = (double)qr[i];
      sumC += cij * (qi * (double)e + qi);      // C_ij q_i exp(delta_ij)   (f64 only)
    }
    if (((i - i0) & 15) == 15) { sumB += (double)accB; accB = (T)0; }
  }
  sumB += (double)accB;
  sumB *= (double)wj;
  sumC *= (double)qj;
  __shared__ double red[4];
  const double tb = mm_block_sum256(sumB, red);
  if (tid == 0) partB[((size_t)b * P + p) * NS + blockIdx.x] = tb;
  if (withC) {
    const double tc = mm_block_sum256(sumC, red);
    if (tid == 0) partC[((size_t)b * L + a) * NS + blockIdx.x] = tc;
  }
}

// ---------------------------------------------------------------------------------------------
// k_finalize: Sff from the partial slabs (fixed summation order => bitwise reproducible)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ partB, const double* __restrict__ partC,
                                                  const double* __restrict__ var, int B, int L, int P, int NS,
                                                  int nsB_diag, int nsB_off, int nsC, int full, int with_unc,
                                                  double jitter, const double* __restrict__ f1raw,
                                                  const double* __restrict__ s12, const double* __restrict__ s56, int diag_factored,
                                                  const int* __restrict__ rflag, int ns_routed, T* __restrict__ Sff) {
  // one wave per (b, pair): lanes stride over the slab (coalesced), fixed butterfly => reproducible
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (idx >= B * P) return;
  const int b = idx / P, p = idx - b * P;
  int a, a2;
  mm_decode_pair(p, L, a, a2);
  const double* pb = partB + ((size_t)b * P + p) * NS;
  int ns = (a == a2) ? nsB_diag : nsB_off;
  // an item the accuracy contract re-reduced in f64 (mm_route.hip): its slab holds the route kernel's npanel x ncc partial sums
  const bool routed = a != a2 && rflag != nullptr && rflag[(size_t)b * (P - L) + (p - L)];
  if (routed) ns = ns_routed;
  double s = 0.0;
  for (int k = lane; k < ns; k += 64) s += pb[k];
  if (a == a2 && with_unc) {
    const double* pc = partC + ((size_t)b * L + a) * NS;
    for (int k = lane; k < nsC; k += 64) s += pc[k];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane != 0) return;
  // f32 mode: the tile kernel reduced only the remainder; the moments supply 1 + b + b^2/2, and c0 b^3 + c1 b^4 of a collapsed item (k_spoly)
  // (orders 5 and 6 of a collapsed item: s56, from f32 moments -- a routed item has them in its f64 re-reduce instead)
  if (a != a2 && s12) s += s12[(size_t)b * (P - L) + (p - L)] - f1raw[(size_t)b * L + a] * f1raw[(size_t)b * L + a2];
  if (a != a2 && s56 && !routed) s += s56[(size_t)b * (P - L) + (p - L)];
  if (a == a2) {
    // factored f64 reduce: the slabs hold sum_ij u_i u_j [D_ij] e^{delta_ij}; minus (sum_i w_i)^2 gives the centred sum
    if (diag_factored) s -= f1raw[(size_t)b * L + a] * f1raw[(size_t)b * L + a];
    if (with_unc) s += var[a];                 // models.py:254-261
    s += jitter;                               // models.py:293-296
    if (full) Sff[((size_t)b * L + a) * L + a] = (T)s;
    else Sff[(size_t)b * L + a] = (T)s;
  } else {
    Sff[((size_t)b * L + a) * L + a2] = (T)s;
    Sff[((size_t)b * L + a2) * L + a] = (T)s;
  }
}

// ---------------------------------------------------------------------------------------------
// k_euler: MomentMatchingEuler.step (solvers.py:110-135), one workgroup per b, in-place safe
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void k_euler(int d, double dt, const T* __restrict__ mu, const T* __restrict__ Sigma,
                                              const T* __restrict__ f1, const T* __restrict__ Sff,
                                              const T* __restrict__ cross, T* mu_out, T* Sigma_out,
                                              T* traj_mu, T* traj_Sigma) {
  extern __shared__ double smem[];
  double* Sg = smem;          // [d][d]
  double* Cr = Sg + d * d;    // [d][d] cross_pre
  double* Sxf = Cr + d * d;   // [d][d]
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int idx = lane; idx < d * d; idx += 64) {
    Sg[idx] = (double)Sigma[(size_t)b * d * d + idx];
    Cr[idx] = (double)cross[(size_t)b * d * d + idx];
  }
  __syncthreads();
  for (int idx = lane; idx < d * d; idx += 64) {
    const int i = idx / d, j = idx - i * d;
    double s = 0.0;
    for (int k = 0; k < d; ++k) s += Sg[i * d + k] * Cr[k * d + j];   // gaussian.py:38-39
    Sxf[idx] = s;
  }
  __syncthreads();
  for (int idx = lane; idx < d * d; idx += 64) {
    const int i = idx / d, j = idx - i * d;
    const double v = Sg[idx] + dt * (Sxf[i * d + j] + Sxf[j * d + i]) + dt * dt * (double)Sff[(size_t)b * d * d + idx];
    Sigma_out[(size_t)b * d * d + idx] = (T)v;
    if (traj_Sigma) traj_Sigma[(size_t)b * d * d + idx] = (T)v;
  }
  if (lane < d) {
    const double v = (double)mu[(size_t)b * d + lane] + dt * (double)f1[(size_t)b * d + lane];
    mu_out[(size_t)b * d + lane] = (T)v;
    if (traj_mu) traj_mu[(size_t)b * d + lane] = (T)v;
  }
}

// ---------------------------------------------------------------------------------------------
// k_expected_cost: closed-form expected saturating cost (GaussianObjective, components.py:26-37)
//   cost = -det(I + S W)^-1/2 exp(-0.5 err^T W (I + S W)^-1 err),  one 64-lane workgroup per element.
//   Gaussian elimination with partial pivoting on [I + S W | err] in LDS (f64).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void k_expected_cost(int d, const T* __restrict__ mean, const T* __restrict__ cov,
                                                      const T* __restrict__ target, const T* __restrict__ precis,
                                                      T* __restrict__ cost) {
  extern __shared__ double smem[];
  mm_expected_cost_body<T>(d, mean, cov, target, precis, cost, (int)blockIdx.x, (int)threadIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// f32 MFMA reduce kernels (mm_mfma.hip)
extern "C" int mm_mfma_supported(int d);
// off-diagonal pairs, f32: fills partB[b][p >= L][0 .. mm_mfma_num_slots(Mp))
int mm_mfma_num_slots(int Mp);
// f64 MFMA reduce (mm_f64.hip): diagonal pairs of both modes, off-diagonal pairs of the f64 mode
int mm_f64_num_slots(int Mp, int diag);
int mm_launch_qred_f64_both(const double* Zc, int Kz, const double* Cm, const double* beta, int M, int L, int Mp, int d, int P, int NS,
                            int Po, int B, int force_worst, const double* qhR, const double* qhC, const double* rowD,
                            const double* colD, const double* w64, const double* q64, const double* rowO, const double* colO,
                            double* partB, double* partC, hipStream_t stream, bool* launched);
int mm_launch_qred_f64(const double* Zc, int Kz, const double* Cm, const double* beta, int M, int L, int Mp, int d,
                       int P, int NS, int p0, int npairs, int B, int diag, int lowp, int force_worst,
                       const double* w, const double* q, const double* rowA, const double* colB,
                       double* partB, double* partC, hipStream_t stream);
int mm_launch_qred_mfma(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl,
                        int B, int L, int d, int flags, hipStream_t stream);
// f32 mode: estimate-driven f64 re-reduce of the (b, pair) items the f32 sweep is not good for (mm_route.hip)
int mm_launch_route(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl, int B, int L, int M,
                    int d, int flags, int agg, double* out, int32_t* status, hipStream_t stream);
// f32 mode: weight moments against the monomial tables + their per-(b, pair) contraction (mm_moments.hip): fills s12
int mm_launch_moments(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl,
                      int B, int L, int d, const void* mu_f32, int flags, hipStream_t stream);

// f32 mode, d <= 8: the degree-5/6 monomial tables and the contraction's index tables (mm_moments6.hip)
int mm_launch_pack56(char* packed, const MMModelLayout& lay, int L, int M, int d, const double* Z, hipStream_t s);

#define MM_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

static int mm_check_common(const void* packed, int L, int M, int d, int dtype, int B) {
  if (!packed) return MM_E_ARG;
  if (L <= 0 || M <= 0 || d <= 0 || B <= 0) return MM_E_ARG;
  if (d > MM_DMAX) return MM_E_DIM;
  if (dtype != MM_F32 && dtype != MM_F64) return MM_E_DTYPE;
  return 0;
}

static inline int mm_dk(int d) { return d <= 4 ? 4 : d <= 8 ? 8 : d <= 16 ? 16 : 32; }

extern "C" int mm_abi_version(void) { return MM_ABI_VERSION; }

extern "C" size_t mm_packed_model_bytes(int L, int M, int d, int dtype, int with_C) {
  if (L <= 0 || M <= 0 || d <= 0 || d > MM_DMAX) return 0;
  return mm_model_layout(L, M, d, dtype, with_C).total;
}

extern "C" size_t mm_workspace_bytes(int B, int L, int M, int d, int dtype, int flags) {
  if (B <= 0 || L <= 0 || M <= 0 || d <= 0 || d > MM_DMAX) return 0;
  return mm_workspace_layout(B, L, M, d, dtype, flags).total;
}

template <typename T>
static int mm_pack_model_t(char* packed, const MMModelLayout& lay, int L, int M, int d,
                           const double* Z, const double* ls, const double* var, const double* beta,
                           const double* C, const double* mean_c, hipStream_t s) {
  const int sorted = M > MM_SORT_MIN_M ? 1 : 0;
  if (sorted) {
    hipLaunchKernelGGL(k_pack_key, dim3(L), dim3(256), 0, s, packed, lay, M, d, Z, ls);
    MM_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_pack_rank, dim3((lay.Mp + 255) / 256, L), dim3(256), 0, s, packed, lay, M, sorted);
  MM_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_pack_gather, dim3((M + 255) / 256, L), dim3(256), 0, s, packed, lay, M, d, Z, beta);
  MM_CHECK_LAUNCH();
  hipLaunchKernelGGL((k_pack_vectors<T>), dim3(L), dim3(256), 0, s, packed, lay, L, M, d, ls, var, mean_c);
  MM_CHECK_LAUNCH();
  if (sizeof(T) == 4 && d <= 8) {
    const int rc = mm_launch_pack56(packed, lay, L, M, d, (const double*)(packed + lay.Z64), s);
    if (rc) return rc;
  }
  if (C) {
    hipLaunchKernelGGL(k_pack_C, dim3((lay.Mp + 255) / 256, lay.Mp, L), dim3(256), 0, s, packed, lay, L, M, C);
    MM_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int mm_pack_model(void* packed, size_t packed_bytes, int L, int M, int d, int dtype,
                             const double* Z, const double* lengthscales, const double* variance,
                             const double* beta, const double* C, const double* mean_c, void* stream) {
  int rc = mm_check_common(packed, L, M, d, dtype, 1);
  if (rc) return rc;
  if (!Z || !lengthscales || !variance || !beta) return MM_E_ARG;
  const MMModelLayout lay = mm_model_layout(L, M, d, dtype, C != nullptr);
  if (packed_bytes < lay.total) return MM_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MM_F64) return mm_pack_model_t<double>((char*)packed, lay, L, M, d, Z, lengthscales, variance, beta, C, mean_c, s);
  return mm_pack_model_t<float>((char*)packed, lay, L, M, d, Z, lengthscales, variance, beta, C, mean_c, s);
}

extern "C" int mm_pack_perm(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int32_t* perm, void* stream) {
  int rc = mm_check_common(packed, L, M, d, dtype, 1);
  if (rc) return rc;
  if (!perm) return MM_E_ARG;
  const MMModelLayout lay = mm_model_layout(L, M, d, dtype, 0);       // (perm lies before C: the offset does not depend on with_C)
  if (packed_bytes < lay.total) return MM_E_WORKSPACE;
  const hipError_t e = hipMemcpy2DAsync(perm, (size_t)M * 4, (const char*)packed + lay.perm, (size_t)lay.Mp * 4, (size_t)M * 4, (size_t)L,
                                        hipMemcpyDeviceToDevice, (hipStream_t)stream);
  return e == hipSuccess ? 0 : (int)e;
}

template <typename T, int DK>
static int mm_q_forward_t(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl,
                          int L, int M, int d, int B, const T* mu, const T* Sigma, int flags,
                          T* f1, T* cross, T* q_out, int32_t* status, hipStream_t s, bool joins) {
  // joins: the caller runs the Q stage in the same call (and so joins the side stream this q stage may open)
  const double* ls2 = (const double*)(packed + ml.ls2);
  const double* var = (const double*)(packed + ml.var);
  const double* Z64 = (const double*)(packed + ml.Z64);
  double* pairmat = (double*)(ws + wl.pairmat);
  double* latmat = (double*)(ws + wl.latmat);
  const size_t shm = (size_t)8 * d * (d + 1) * sizeof(double);
  unsigned int* amax = (sizeof(T) == 4 && wl.Po > 0) ? (unsigned int*)(ws + wl.amax) : nullptr;
  if ((long long)(wl.P + L) * B <= 4096) {
    hipLaunchKernelGGL((k_prep<T>), dim3(wl.P + L, B), dim3(192), shm, s,
                       ls2, var, L, d, wl.P, mu, Sigma, pairmat, latmat, amax, status, 2);
    MM_CHECK_LAUNCH();
  } else {
    hipLaunchKernelGGL((k_prep<T>), dim3(L, B), dim3(64), shm, s,
                       ls2, var, L, d, wl.P, mu, Sigma, pairmat, latmat, amax, status, 0);
    MM_CHECK_LAUNCH();
    hipLaunchKernelGGL((k_prep<T>), dim3(wl.P, B), dim3(64), shm, s,
                       ls2, var, L, d, wl.P, mu, Sigma, pairmat, latmat, amax, status, 1);
    MM_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL((k_qvec<T, DK>), dim3(L, B), dim3(256), 0, s,
                     (const double*)(packed + ml.Zt64), (const double*)(packed + ml.beta64), (const double*)(packed + ml.meanc),
                     L, M, wl.Mp, d, mu, latmat, (double*)(ws + wl.w64), (double*)(ws + wl.q64), (T*)(ws + wl.w),
                     (double*)(ws + wl.f1raw), (double*)(ws + wl.rho1), f1, cross, q_out, (double*)(ws + wl.mu64), (double*)(ws + wl.lq),
                     (const int*)(packed + ml.perm));
  MM_CHECK_LAUNCH();
  {
    // The streamed operands of the reduces, one launch per kind of pair: the diagonal pairs' first (all the diagonal sweep needs),
    // then the off-diagonal pairs' and -- f32 packs -- the moment chain behind them (k_wmom_perm, k_wmom_gemm, k_spoly: the exact
    // polynomial part of the off-diagonal sums, mm_moments.hip; s12 is read after both sweeps only).  Where the call can fork, the
    // second group runs on the SIDE STREAM beside the diagonal pairs' sweep (mm_fork.h): HBM-bound operand writes and a small f64
    // GEMM next to an f64 issue-bound kernel.  mm_Q_reduce_t joins before the off-diagonal sweep.
    const int nblk = (wl.Mp + 255) / 256;                  // 256-row chunks = wsum slots per (b, pair)
    auto nsplit_for = [&](int npairs) {
      long long per = (long long)npairs * B;               // workgroups per chunk split
      int ns = (int)((4096 + per - 1) / per);              // >= 16 workgroups per CU, else one per (b, pair)
      if (ns > nblk) ns = nblk;
      return ns < 1 ? 1 : ns;
    };
#define MM_PAIRVEC_ARGS                                                                                                     \
    (const double*)(packed + ml.Zt64), (const double*)(packed + ml.zbar), ls2, L, M, wl.Mp, d, wl.P, mu, pairmat,            \
    (const double*)(ws + wl.rho1), (double*)(ws + wl.rowD), (double*)(ws + wl.colD), (T*)(ws + wl.rowO), (T*)(ws + wl.colO), \
    (const double*)(ws + wl.w64), (double*)(ws + wl.whR), (double*)(ws + wl.whC), amax,                                      \
    (const double*)(ws + wl.q64), (double*)(ws + wl.qhR), (double*)(ws + wl.qhC), (flags & MM_MODEL_UNCERTAINTY) ? 1 : 0, nblk, \
    (const double*)(ws + wl.lq), (const double*)(packed + ml.beta64), (const double*)(packed + ml.zmax2),                       \
    ((sizeof(T) == 4 && d <= 8) ? (unsigned short*)(ws + wl.wsp) : (unsigned short*)nullptr),                                \
    (unsigned char*)(ws + wl.gflag), (float*)(ws + wl.gmax2), ((flags & (MM_FORCE_WORST_TIER | MM_ISTAGE_NO_M56)) ? 0 : 1)
    auto pairvec = [&](int p0, int npairs, hipStream_t st) {
      if (npairs <= 0) return;
      if constexpr (DK <= 8) {
        hipLaunchKernelGGL((k_pairvec_reg<T, DK>), dim3(nsplit_for(npairs), npairs, B), dim3(256), 0, st, MM_PAIRVEC_ARGS, p0);
      } else {
        hipLaunchKernelGGL((k_pairvec<T, DK>), dim3(nsplit_for(npairs), npairs, B), dim3(256), 0, st, MM_PAIRVEC_ARGS, p0);
      }
    };
    pairvec(0, L, s);
    MM_CHECK_LAUNCH();
    if (wl.Po > 0) {
      // (only a call that joins itself forks -- mm_moment_match, the rollouts: a stand-alone mm_q_forward returns with everything it
      // enqueued on the caller's stream, so that the caller's stream order -- and its allocator's -- covers the workspace)
      // (small problems: nothing to hide behind, and launches are what counts)
      MMFork* fork = (joins && (long long)wl.P * B >= 512) ? mm_fork_get(s) : nullptr;
      hipStream_t s2 = s;
      if (fork) {
        hipError_t ef = hipEventRecord(fork->fork, s);
        if (ef == hipSuccess) ef = hipStreamWaitEvent(fork->s2, fork->fork, 0);
        if (ef != hipSuccess) return (int)ef;
        s2 = fork->s2;
      }
      pairvec(L, wl.Po, s2);
      int rcs = 0;
      { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) rcs = (int)e_; }
      if (!rcs && sizeof(T) == 4) rcs = mm_launch_moments(packed, ml, ws, wl, B, L, d, (const void*)mu, flags, s2);
      if (fork) {
        // the join is recorded whatever happened in between: an error return must not leave the side stream unjoined (under
        // capture that would strand the capture), and the caller's stream waits for it before this call's error is reported
        hipError_t ej = hipEventRecord(fork->join, fork->s2);
        if (rcs) { if (ej == hipSuccess) (void)hipStreamWaitEvent(s, fork->join, 0); return rcs; }
        if (ej != hipSuccess) return (int)ej;
      }
      if (rcs) return rcs;
    }
#undef MM_PAIRVEC_ARGS
  }
  MM_CHECK_LAUNCH();
  return 0;
}

template <typename T, int DK>
static int mm_Q_reduce_t(const char* packed, const MMModelLayout& ml, bool has_C, char* ws, const MMWorkspaceLayout& wl,
                         int L, int M, int d, int B, int flags, double jitter, T* Sff, int32_t* status, hipStream_t s) {
  const int with_unc = (flags & MM_MODEL_UNCERTAINTY) ? 1 : 0;
  const int full = (flags & MM_FULL_OUTPUT_COV) ? 1 : 0;
  if (with_unc && !has_C) return MM_E_NO_C;
  const double* Cm = with_unc ? (const double*)(packed + ml.Cm) : nullptr;
  double* partB = (double*)(ws + wl.partB);
  double* partC = (double*)(ws + wl.partC);
  const int nrb = (wl.Mp + MM_GEN_ROWS - 1) / MM_GEN_ROWS, ncb = (wl.Mp + MM_GEN_COLS - 1) / MM_GEN_COLS;
  const bool generic = (flags & MM_FORCE_GENERIC) != 0;
  const bool use_mfma32 = sizeof(T) == 4 && !generic && mm_mfma_supported(d);
  const int ns_gen = nrb * ncb;
  const int nsB_diag = generic ? ns_gen : mm_f64_num_slots(wl.Mp, 1), nsC = nsB_diag;
  const int nsB_off = generic ? ns_gen : (sizeof(T) == 4 ? (use_mfma32 ? mm_mfma_num_slots(wl.Mp) : ns_gen)
                                                         : mm_f64_num_slots(wl.Mp, 0));
  int stages = flags & (MM_STAGE_DIAG | MM_STAGE_OFFDIAG | MM_STAGE_FINALIZE);
  if (!stages) stages = MM_STAGE_DIAG | MM_STAGE_OFFDIAG | MM_STAGE_FINALIZE;
  // the f32 sweep leaves its own error estimate per (b, pair); items beyond MM_ROUTE_TOL are re-reduced in f64 (mm_route.hip).
  // A function of the flags alone: the stage calls of one match (bench.py) agree on it.  Not with the forced worst tier (its
  // tile ranges are fake)
  const bool routes = use_mfma32 && wl.Po > 0 && !(flags & (MM_FORCE_WORST_TIER | MM_NO_ROUTE));
  // small f64 models: both kinds of pairs in one launch (mm_f64.hip) -- the two sweeps then run side by side on the device
  bool both = false;
  if (sizeof(T) == 8 && !generic && wl.Po > 0 && (stages & (MM_STAGE_DIAG | MM_STAGE_OFFDIAG)) == (MM_STAGE_DIAG | MM_STAGE_OFFDIAG)) {
    if (const int rj = mm_fork_join_wait(s)) return rj;     // (it reads the off-diagonal operands too)
    const int rc = mm_launch_qred_f64_both((const double*)(packed + ml.Zc64), ml.Kz, Cm, (const double*)(packed + ml.beta64), M, L,
                                           wl.Mp, d, wl.P, wl.NS, wl.Po, B, (flags & MM_FORCE_WORST_TIER) ? 1 : 0,
                                           (const double*)(ws + wl.qhR), (const double*)(ws + wl.qhC), (const double*)(ws + wl.rowD),
                                           (const double*)(ws + wl.colD), (const double*)(ws + wl.w64), (const double*)(ws + wl.q64),
                                           (const double*)(ws + wl.rowO), (const double*)(ws + wl.colO), partB, partC, s, &both);
    if (rc) return rc;
  }
  // (1) diagonal pairs: always f64
  if (!both && (stages & MM_STAGE_DIAG)) {
    if (generic) {
      hipLaunchKernelGGL((k_qred_generic<double, DK, false>), dim3(nrb * ncb, L, B), dim3(256), 0, s,
                         (const double*)(packed + ml.Zc64), ml.Kz, Cm, L, wl.Mp, d, wl.P, wl.NS, ncb, 0,
                         (const double*)(ws + wl.w64), (const double*)(ws + wl.q64),
                         (const double*)(ws + wl.rowD), (const double*)(ws + wl.colD), partB, partC,
                         (const unsigned char*)nullptr, (const unsigned int*)nullptr, (const double*)nullptr);
      MM_CHECK_LAUNCH();
    } else {
      const int rc = mm_launch_qred_f64((const double*)(packed + ml.Zc64), ml.Kz, Cm,
                                        (const double*)(packed + ml.beta64), M, L, wl.Mp, d, wl.P, wl.NS,
                                        0, L, B, 1, sizeof(T) == 4 ? 1 : 0, (flags & MM_FORCE_WORST_TIER) ? 1 : 0,
                                        (const double*)(ws + wl.qhR), (const double*)(ws + wl.qhC),   // factored weights
                                        (const double*)(ws + wl.rowD), (const double*)(ws + wl.colD),
                                        partB, partC, s);
      if (rc) return rc;
    }
  }
  // (the off-diagonal pairs' operands and the moment chain may still be on the q stage's side stream)
  if (wl.Po > 0 && (stages & (MM_STAGE_OFFDIAG | MM_STAGE_FINALIZE))) {
    if (const int rj = mm_fork_join_wait(s)) return rj;
  }
  // (2) off-diagonal pairs in T
  if (!both && wl.Po > 0 && (stages & MM_STAGE_OFFDIAG)) {
    if (use_mfma32) {
      int rc = mm_launch_qred_mfma(packed, ml, ws, wl, B, L, d, flags, s);
      if (rc) return rc;
      if (routes) {
        // (mm_launch_route joins the q stage's moment chain on the side stream: s12 is the route decision's scale)
        rc = mm_launch_route(packed, ml, ws, wl, B, L, M, d, flags, 0, partB, status, s);
        if (rc) return rc;
      }
    } else if (sizeof(T) == 8 && !generic) {
      const int rc = mm_launch_qred_f64((const double*)(packed + ml.Zc64), ml.Kz, nullptr,
                                        (const double*)(packed + ml.beta64), M, L, wl.Mp, d, wl.P, wl.NS,
                                        L, wl.Po, B, 0, 0, (flags & MM_FORCE_WORST_TIER) ? 1 : 0, (const double*)(ws + wl.w64), (const double*)(ws + wl.q64),
                                        (const double*)(ws + wl.rowO), (const double*)(ws + wl.colO),
                                        partB, partC, s);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL((k_qred_generic<T, DK, sizeof(T) == 4>), dim3(nrb * ncb, wl.Po, B), dim3(256), 0, s,
                         (const T*)(packed + ml.Zc), ml.Kz, (const double*)nullptr, L, wl.Mp, d, wl.P, wl.NS, ncb, L,
                         (const T*)(ws + wl.w), (const T*)nullptr, (const T*)(ws + wl.rowO),
                         (const T*)(ws + wl.colO), partB, partC,
                         (sizeof(T) == 4 && d <= 8 && !(flags & MM_FORCE_WORST_TIER)) ? (const unsigned char*)(ws + wl.gflag) : (const unsigned char*)nullptr,
                         (const unsigned int*)(ws + wl.amaxc),
                         (sizeof(T) == 4 && mm_moment_deg(d) >= 4) ? (const double*)(packed + ml.zmax2) : (const double*)nullptr);
      MM_CHECK_LAUNCH();
    }
  }
  if (stages & MM_STAGE_FINALIZE) {
    const int n = B * wl.P;
    hipLaunchKernelGGL((k_finalize<T>), dim3((n + 3) / 4), dim3(256), 0, s,
                       partB, partC, (const double*)(packed + ml.var), B, L, wl.P, wl.NS,
                       nsB_diag, nsB_off, nsC, full, with_unc, jitter, (const double*)(ws + wl.f1raw),
                       sizeof(T) == 4 ? (const double*)(ws + wl.s12) : (const double*)nullptr,
                       (sizeof(T) == 4 && d <= 8 && wl.Po > 0) ? (const double*)(ws + wl.s56) : (const double*)nullptr, generic ? 0 : 1,
                       routes ? (const int*)(ws + wl.rflag) : (const int*)nullptr,
                       mm_mfma_num_slots(wl.Mp) * mm_route_ncc(wl.NS, mm_mfma_num_slots(wl.Mp)), Sff);
    MM_CHECK_LAUNCH();
  }
  return 0;
}

// Whether C is present is inferred from the size of the packed buffer (C is its last section).
template <typename T, int DK>
static int mm_moment_match_t(const char* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                             const T* mu, const T* Sigma, int flags, double jitter,
                             T* f1, T* Sff, T* cross, T* q_out, char* ws, size_t ws_bytes,
                             int32_t* status, hipStream_t s, bool do_q, bool do_Q) {
  const int with_unc = (flags & MM_MODEL_UNCERTAINTY) ? 1 : 0;
  const MMModelLayout ml = mm_model_layout(L, M, d, dtype, 1);
  if (packed_bytes < ml.Cm) return MM_E_WORKSPACE;
  const bool has_C = packed_bytes >= ml.total;
  if (with_unc && !has_C) return MM_E_NO_C;
  const MMWorkspaceLayout wl = mm_workspace_layout(B, L, M, d, dtype, flags);
  if (ws_bytes < wl.total) return MM_E_WORKSPACE;
  int rc = 0;
  if (do_q) {
    // the q stage forks only where the Q stage of this very call joins: every stage requested (MM_STAGE_* all set or none)
    const int st = flags & (MM_STAGE_DIAG | MM_STAGE_OFFDIAG | MM_STAGE_FINALIZE);
    const bool joins = do_Q && (st == 0 || (st & (MM_STAGE_OFFDIAG | MM_STAGE_FINALIZE)) != 0);
    rc = mm_q_forward_t<T, DK>(packed, ml, ws, wl, L, M, d, B, mu, Sigma, flags, f1, cross, q_out, status, s, joins);
    if (rc) return rc;
  }
  if (do_Q) rc = mm_Q_reduce_t<T, DK>(packed, ml, has_C, ws, wl, L, M, d, B, flags, jitter, Sff, status, s);
  return rc;
}

#define MM_DISPATCH(T_, CALL)                                             \
  switch (mm_dk(d)) {                                                     \
    case 4: return CALL(T_, 4);                                           \
    case 8: return CALL(T_, 8);                                           \
    case 16: return CALL(T_, 16);                                         \
    default: return CALL(T_, 32);                                         \
  }

extern "C" int mm_moment_match(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                               const void* mu, const void* Sigma, int flags, double jitter,
                               void* f1, void* Sff, void* cross_pre,
                               void* workspace, size_t workspace_bytes, int32_t* status, void* stream) {
  int rc = mm_check_common(packed, L, M, d, dtype, B);
  if (rc) return rc;
  if (!mu || !Sigma || !f1 || !Sff || !cross_pre || !workspace) return MM_E_ARG;
  hipStream_t s = (hipStream_t)stream;
#define CALL_MM(T_, DK_) mm_moment_match_t<T_, DK_>((const char*)packed, packed_bytes, L, M, d, dtype, B, (const T_*)mu, (const T_*)Sigma, \
    flags, jitter, (T_*)f1, (T_*)Sff, (T_*)cross_pre, (T_*)nullptr, (char*)workspace, workspace_bytes, status, s, true, true)
  if (dtype == MM_F64) { MM_DISPATCH(double, CALL_MM) }
  MM_DISPATCH(float, CALL_MM)
#undef CALL_MM
}

extern "C" int mm_q_forward(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                            const void* mu, const void* Sigma, int flags,
                            void* f1, void* cross_pre, void* q_out,
                            void* workspace, size_t workspace_bytes, int32_t* status, void* stream) {
  int rc = mm_check_common(packed, L, M, d, dtype, B);
  if (rc) return rc;
  if (!mu || !Sigma || !f1 || !cross_pre || !workspace) return MM_E_ARG;
  hipStream_t s = (hipStream_t)stream;
#define CALL_Q(T_, DK_) mm_moment_match_t<T_, DK_>((const char*)packed, packed_bytes, L, M, d, dtype, B, (const T_*)mu, (const T_*)Sigma, \
    flags, 0.0, (T_*)f1, (T_*)nullptr, (T_*)cross_pre, (T_*)q_out, (char*)workspace, workspace_bytes, status, s, true, false)
  if (dtype == MM_F64) { MM_DISPATCH(double, CALL_Q) }
  MM_DISPATCH(float, CALL_Q)
#undef CALL_Q
}

extern "C" int mm_Q_reduce_forward(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                                   int flags, double jitter, void* Sff,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  int rc = mm_check_common(packed, L, M, d, dtype, B);
  if (rc) return rc;
  if (!Sff || !workspace) return MM_E_ARG;
  hipStream_t s = (hipStream_t)stream;
#define CALL_QQ(T_, DK_) mm_moment_match_t<T_, DK_>((const char*)packed, packed_bytes, L, M, d, dtype, B, (const T_*)nullptr, (const T_*)nullptr, \
    flags, jitter, (T_*)nullptr, (T_*)Sff, (T_*)nullptr, (T_*)nullptr, (char*)workspace, workspace_bytes, nullptr, s, false, true)
  if (dtype == MM_F64) { MM_DISPATCH(double, CALL_QQ) }
  MM_DISPATCH(float, CALL_QQ)
#undef CALL_QQ
}

template <typename T>
static int mm_euler_t(int B, int d, double dt, const T* mu, const T* Sigma, const T* f1, const T* Sff,
                      const T* cross, T* mu_out, T* Sigma_out, T* tmu, T* tS, hipStream_t s) {
  const size_t shm = (size_t)3 * d * d * sizeof(double);
  hipLaunchKernelGGL((k_euler<T>), dim3(B), dim3(64), shm, s, d, dt, mu, Sigma, f1, Sff, cross, mu_out, Sigma_out, tmu, tS);
  MM_CHECK_LAUNCH();
  return 0;
}

extern "C" int mm_euler_update(int B, int d, int dtype, double dt, const void* mu, const void* Sigma,
                               const void* f1, const void* Sff, const void* cross_pre,
                               void* mu_out, void* Sigma_out, void* stream) {
  if (B <= 0 || d <= 0) return MM_E_ARG;
  if (d > MM_DMAX) return MM_E_DIM;
  if (dtype != MM_F32 && dtype != MM_F64) return MM_E_DTYPE;
  if (!mu || !Sigma || !f1 || !Sff || !cross_pre || !mu_out || !Sigma_out) return MM_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MM_F64)
    return mm_euler_t<double>(B, d, dt, (const double*)mu, (const double*)Sigma, (const double*)f1, (const double*)Sff,
                              (const double*)cross_pre, (double*)mu_out, (double*)Sigma_out, nullptr, nullptr, s);
  return mm_euler_t<float>(B, d, dt, (const float*)mu, (const float*)Sigma, (const float*)f1, (const float*)Sff,
                           (const float*)cross_pre, (float*)mu_out, (float*)Sigma_out, nullptr, nullptr, s);
}

template <typename T, int DK>
static int mm_rollout_t(const char* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B, int H, double dt, int flags,
                        double jitter, T* mu, T* Sigma, T* tmu, T* tS, char* ws, size_t ws_bytes,
                        int32_t* status, hipStream_t s) {
  const MMWorkspaceLayout wl = mm_workspace_layout(B, L, M, d, dtype, flags);
  if (ws_bytes < wl.total) return MM_E_WORKSPACE;
  T* f1 = (T*)(ws + wl.f1s);
  T* Sff = (T*)(ws + wl.Sffs);
  T* cr = (T*)(ws + wl.crs);
  for (int h = 0; h < H; ++h) {
    int rc = mm_moment_match_t<T, DK>(packed, packed_bytes, L, M, d, dtype, B, mu, Sigma, flags, jitter, f1, Sff, cr, (T*)nullptr,
                                      ws, ws_bytes, status, s, true, true);
    if (rc) return rc;
    rc = mm_euler_t<T>(B, d, dt, mu, Sigma, f1, Sff, cr, mu, Sigma,
                       tmu ? tmu + (size_t)h * B * d : nullptr, tS ? tS + (size_t)h * B * d * d : nullptr, s);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int mm_rollout_closed(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B, int H,
                                 double dt, int flags, double jitter, void* mu, void* Sigma,
                                 void* traj_mu, void* traj_Sigma, void* workspace, size_t workspace_bytes,
                                 int32_t* status, void* stream) {
  int rc = mm_check_common(packed, L, M, d, dtype, B);
  if (rc) return rc;
  if (H <= 0 || !mu || !Sigma || !workspace) return MM_E_ARG;
  if (d != L) return MM_E_STATE;
  if (!(flags & MM_FULL_OUTPUT_COV)) return MM_E_STATE;
  hipStream_t s = (hipStream_t)stream;
#define CALL_R(T_, DK_) mm_rollout_t<T_, DK_>((const char*)packed, packed_bytes, L, M, d, dtype, B, H, dt, flags, jitter, (T_*)mu, (T_*)Sigma, \
    (T_*)traj_mu, (T_*)traj_Sigma, (char*)workspace, workspace_bytes, status, s)
  if (dtype == MM_F64) { MM_DISPATCH(double, CALL_R) }
  MM_DISPATCH(float, CALL_R)
#undef CALL_R
}

extern "C" int mm_expected_cost(int N, int d, int dtype, const void* mean, const void* cov,
                                const void* target, const void* precis, void* cost, void* stream) {
  if (N <= 0 || d <= 0) return MM_E_ARG;
  if (d > MM_DMAX) return MM_E_DIM;
  if (dtype != MM_F32 && dtype != MM_F64) return MM_E_DTYPE;
  if (!mean || !cov || !target || !precis || !cost) return MM_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t shm = mm_cost_lds_bytes(d);
  if (dtype == MM_F64)
    hipLaunchKernelGGL((k_expected_cost<double>), dim3(N), dim3(64), shm, s, d, (const double*)mean, (const double*)cov,
                       (const double*)target, (const double*)precis, (double*)cost);
  else
    hipLaunchKernelGGL((k_expected_cost<float>), dim3(N), dim3(64), shm, s, d, (const float*)mean, (const float*)cov,
                       (const float*)target, (const float*)precis, (float*)cost);
  MM_CHECK_LAUNCH();
  return 0;
}
