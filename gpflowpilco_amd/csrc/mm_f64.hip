// f64 MFMA fused reduce (gfx950): the diagonal kernel pairs (a == a') of every mode, and the
// off-diagonal pairs of the f64 mode.
//
// Diagonal pairs carry the C-weighted term  sum_ij C_ij q_i exp(delta_ij) q_j  of
// models.py:254-261 (tr(W) and sum(W o q_cov)); C = Kuu^-1 S Kuu^-1 - Kuu^-1 has norm up to
// 1/jitter = 1e6, so any per-entry f32 rounding of exp(delta) is amplified beyond use
// (DESIGN.md "fp32 error budget") -- these pairs are reduced in f64 in both modes.
//
//   tile      : 64 x 64 entries per workgroup (4 waves, 32 x 32 each = 2 x 2 MFMA tiles of
//               v_mfma_f64_16x16x4_f64; one extra k-step adds rho_i + gamma'_j)
//   diag mode : only tile pairs it <= jt are visited (Q_aa and C_a are symmetric; strictly
//               upper tiles count twice); the C tile is loaded ONCE into registers and the
//               workgroup loops over its chunk of the batch -- C traffic is M^2*8 B per chunk
//               instead of per batch element;
//   expm1     : branch-free f64 (k ln2 + r reduction, degree-13 polynomial, ldexp), relative
//               error ~2e-16 of expm1 itself for every argument;
//   output    : per (b, pair, tile) partial sums -> slab (deterministic, no atomics).
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mm_decode_pair_f(int p, int L, int& a, int& a2) {
  if (p < L) { a = p; a2 = p; return; }
  int r = p - L, i = 0;
  while (r >= L - 1 - i) { r -= L - 1 - i; ++i; }
  a = i; a2 = i + 1 + r;
}

// expm1, any argument: x = k ln2 + r, degree-12 Taylor of expm1(r) on |r| <= ln2/2 (truncation
// 0.35^12/13! = 5e-16 relative), 2^k (1 + p) - 1 = fma(2^k, p, 2^k - 1).  No clamp: v_cvt_i32_f64
// saturates and v_ldexp_f64 over/underflows to inf/0, which is the right limit.
__device__ __forceinline__ double mm_expm1_f64(double x) {
  const double kf = rint(x * 1.4426950408889634);
  double r = fma(-kf, 6.93147180369123816490e-01, x);
  r = fma(-kf, 1.90821492927058770002e-10, r);
  double q = 2.08767569878681e-09;              // 1/12!
  q = fma(q, r, 2.505210838544172e-08);         // 1/11!
  q = fma(q, r, 2.755731922398589e-07);         // 1/10!
  q = fma(q, r, 2.7557319223985893e-06);        // 1/9!
  q = fma(q, r, 2.48015873015873e-05);          // 1/8!
  q = fma(q, r, 1.984126984126984e-04);         // 1/7!
  q = fma(q, r, 1.388888888888889e-03);         // 1/6!
  q = fma(q, r, 8.333333333333333e-03);         // 1/5!
  q = fma(q, r, 4.1666666666666664e-02);        // 1/4!
  q = fma(q, r, 1.6666666666666666e-01);        // 1/3!
  q = fma(q, r, 0.5);
  q = fma(q, r, 1.0);
  const double p = q * r;                       // expm1(r)
  const double s = ldexp(1.0, (int)kf);
  return fma(s, p, s - 1.0);                    // 2^k (1 + p) - 1
}

// expm1 without range reduction: Taylor to degree DEG, x * (1 + x/2! + ... + x^(DEG-1)/DEG!).
// Truncation relative to expm1(x): |x|^DEG / (DEG+1)!.  Tiers used by the kernel (wave-uniform):
//   |x| <= 0.25: DEG 10 (2.4e-14, f32 mode) or 12 (1e-17, f64 mode);  |x| <= 0.75 / 0.5: DEG 15.
__device__ constexpr double mm_inv_fact(int n) { double f = 1.0; for (int i = 2; i <= n; ++i) f *= i; return 1.0 / f; }

template <int DEG>
__device__ __forceinline__ double mm_expm1_f64_poly(double x) {
  double q = mm_inv_fact(DEG);
#pragma unroll
  for (int k = DEG - 1; k >= 1; --k) q = fma(q, x, mm_inv_fact(k));
  return q * x;
}
__device__ __forceinline__ double mm_expm1_f64_p10(double x) { return mm_expm1_f64_poly<10>(x); }
__device__ __forceinline__ double mm_expm1_f64_p12(double x) { return mm_expm1_f64_poly<12>(x); }
__device__ __forceinline__ double mm_expm1_f64_p15(double x) { return mm_expm1_f64_poly<15>(x); }

// KS4: number of K=4 MFMA steps covering the d input dimensions.
// grid: x = tile pairs (diag: nt(nt+1)/2 upper pairs; else nt*nt), y = pairs of this launch,
//       z = batch chunks.  Row/col operand arrays are indexed by the local pair index.
template <int KS4, bool DIAG, bool WITHC>
__global__ __launch_bounds__(256, 2) void k_qred_f64_mfma(const double* __restrict__ Zc, int Kz,
                                                          const double* __restrict__ Cm,
                                                          int L, int Mp, int d, int P, int NS, int p0,
                                                          int B, int bchunk, double small_limit,
                                                          const double* __restrict__ w,
                                                          const double* __restrict__ q,
                                                          const double* __restrict__ rowA,
                                                          const double* __restrict__ colB,
                                                          double* __restrict__ partB,
                                                          double* __restrict__ partC) {
  const int nt = Mp / MM_F64_TILE;
  int it, jt;
  if (DIAG) {                       // blockIdx.x -> (it <= jt), row-major over the upper triangle
    int r = blockIdx.x; it = 0;
    while (r >= nt - it) { r -= nt - it; ++it; }
    jt = it + r;
  } else {
    it = blockIdx.x / nt; jt = blockIdx.x - it * nt;
  }
  const int lp = blockIdx.y, np = gridDim.y, p = p0 + lp;
  int a, a2;
  mm_decode_pair_f(p, L, a, a2);
  constexpr bool withC = DIAG && WITHC;
  const double sym = (DIAG && it != jt) ? 2.0 : 1.0;

  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  const int rbase = it * MM_F64_TILE + (wv >> 1) * 32;
  const int cbase = jt * MM_F64_TILE + (wv & 1) * 32;

  // b-independent A operands (centred inducing inputs of latent a)
  const double* zr = Zc + (size_t)a * Mp * Kz;
  double areg[2][KS4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int s = 0; s < KS4; ++s) {
      const int k = 4 * s + kq;
      const double v = zr[(size_t)(rbase + rt * 16 + l15) * Kz + (k < Kz ? k : Kz - 1)];
      areg[rt][s] = (k < Kz) ? v : 0.0;
    }
  // C tile -> registers (element (row(rt, r), col(ct)) in the MFMA accumulator layout)
  double creg[2][2][4];
  if (withC) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          creg[rt][ct][r] = Cm[((size_t)a * Mp + rbase + rt * 16 + kq + 4 * r) * Mp + cbase + ct * 16 + l15];
  }
  size_t boff[KS4];
#pragma unroll
  for (int s = 0; s < KS4; ++s) {
    const int k = 4 * s + kq;
    boff[s] = (size_t)(k < d ? k : d) * Mp;
  }
  const size_t goff = (size_t)d * Mp;

  // in f64 mode only |delta| <= 0.5 takes the unreduced polynomial (full f64 accuracy there)
  const double SMALL_LIMIT = small_limit;
  __shared__ double red[8];
  const int b0 = blockIdx.z * bchunk;
  const int b1 = (b0 + bchunk < B) ? b0 + bchunk : B;
  // per-b operand pointers advance by constant strides (no 64-bit multiplies in the loop)
  const double* ra = rowA + ((size_t)b0 * np + lp) * Mp;
  const double* cb = colB + ((size_t)b0 * np + lp) * (size_t)(d + 1) * Mp;
  const double* wr = w + ((size_t)b0 * L + a) * Mp;
  const double* wc = w + ((size_t)b0 * L + a2) * Mp;
  const double* qr = q + ((size_t)b0 * L + a) * Mp;
  const size_t st_ra = (size_t)np * Mp, st_cb = (size_t)np * (d + 1) * Mp, st_w = (size_t)L * Mp;
  // lane-constant selectors of the extra k-step: A = (rho_i, 1, 0, 0), B = (1, gamma'_j, 0, 0)
  const double selA1 = (kq == 1) ? 1.0 : 0.0, selB0 = (kq == 0) ? 1.0 : 0.0;
  const double mA0 = (kq == 0) ? 1.0 : 0.0, mB1 = (kq == 1) ? 1.0 : 0.0;
  for (int b = b0; b < b1; ++b, ra += st_ra, cb += st_cb, wr += st_w, wc += st_w, qr += st_w) {

    double ax[2], bx[2], breg[2][KS4], wj[2], qj[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const double rv = ra[rbase + rt * 16 + l15];
      ax[rt] = fma(rv, mA0, selA1);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int col = cbase + ct * 16 + l15;
#pragma unroll
      for (int s = 0; s < KS4; ++s) breg[ct][s] = cb[boff[s] + col];
      const double gv = cb[goff + col];
      bx[ct] = fma(gv, mB1, selB0);
      wj[ct] = wc[col];
      qj[ct] = withC ? qr[col] : 0.0;      // a == a2 on the diagonal
    }
    double wi[2][4], qi[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rbase + rt * 16 + kq + 4 * r;
        wi[rt][r] = wr[row];
        qi[rt][r] = withC ? qr[row] : 0.0;
      }

    f64x4 cacc[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f64x4 c = {0.0, 0.0, 0.0, 0.0};
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[rt], bx[ct], c, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < KS4; ++s)
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[rt][s], breg[ct][s], c, 0, 0, 0);
        cacc[rt][ct] = c;
      }
    // wave-uniform choice of the expm1 form (the |.| test runs on the high dwords only)
    double mx = 0.0;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmax(mx, fabs(cacc[rt][ct][r]));
    const bool tiny = !__any(mx > 0.25);
    const bool small = !__any(mx > SMALL_LIMIT);
    double sB = 0.0, sC = 0.0;
#define MM_F64_ACCUM(EXPM1_)                                                              \
    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) {                                    \
      double pB = 0.0, pC = 0.0;                                                          \
      _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                    \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                   \
          const double e = EXPM1_(cacc[rt][ct][r]);                                       \
          pB = fma(wi[rt][r], e, pB);                                                     \
          if (withC) pC = fma(creg[rt][ct][r], fma(qi[rt][r], e, qi[rt][r]), pC);         \
        }                                                                                 \
      sB = fma(pB, wj[ct], sB);                                                           \
      sC = fma(pC, qj[ct], sC);                                                           \
    }
    if (tiny) {
      if (SMALL_LIMIT > 0.6) { MM_F64_ACCUM(mm_expm1_f64_p10) } else { MM_F64_ACCUM(mm_expm1_f64_p12) }
    } else if (small) {
      MM_F64_ACCUM(mm_expm1_f64_p15)
    } else {
      MM_F64_ACCUM(mm_expm1_f64)
    }
#undef MM_F64_ACCUM
    // workgroup reduction of (sB, sC) for this b.  First fold across the two lane halves so that
    // lanes 0-31 carry sB partials and lanes 32-63 sC partials, then 5 butterfly steps on ONE value.
    double v;
    {
      const bool lo = lane < 32;
      const double keep = lo ? sB : sC, give = lo ? sC : sB;
      v = keep + __shfl_xor(give, 32, 64);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if (lane == 0) red[wv] = v;            // sB of this wave
    if (lane == 32) red[4 + wv] = v;       // sC of this wave
    __syncthreads();
    if (threadIdx.x == 0) {
      partB[((size_t)b * P + p) * NS + blockIdx.x] = sym * (red[0] + red[1] + red[2] + red[3]);
      if (withC) partC[((size_t)b * L + a) * NS + blockIdx.x] = sym * (red[4] + red[5] + red[6] + red[7]);
    }
  }
}

int mm_f64_num_slots(int Mp, int diag) {
  const int nt = Mp / MM_F64_TILE;
  return diag ? nt * (nt + 1) / 2 : nt * nt;
}

// Launch over `npairs` pairs starting at global pair index p0.  diag != 0: pairs are (a, a).
int mm_launch_qred_f64(const double* Zc, int Kz, const double* Cm, int L, int Mp, int d, int P, int NS,
                       int p0, int npairs, int B, int diag, double small_limit,
                       const double* w, const double* q, const double* rowA, const double* colB,
                       double* partB, double* partC, hipStream_t stream) {
  if (npairs <= 0) return 0;
  const int nslots = mm_f64_num_slots(Mp, diag);
  if (nslots > NS) return MM_E_WORKSPACE;
  // batch chunk: enough workgroups to fill the chip (>= ~8 per CU), as few C re-reads as possible
  long long per_chunk = (long long)nslots * npairs;
  int nchunk = (int)((2048 + per_chunk - 1) / per_chunk);
  if (nchunk < 1) nchunk = 1;
  if (nchunk > B) nchunk = B;
  if (!diag) nchunk = B < 64 ? B : 64;       // no C reuse to protect: more, shorter workgroups
  const int bchunk = (B + nchunk - 1) / nchunk;
  nchunk = (B + bchunk - 1) / bchunk;
  dim3 grid(nslots, npairs, nchunk);
  const int ks4 = (d + 3) / 4;
#define MM_LAUNCH_F64(KS_, DG_)                                                                    \
  do {                                                                                             \
    if (DG_ && Cm != nullptr)                                                                      \
      hipLaunchKernelGGL((k_qred_f64_mfma<KS_, DG_, true>), grid, dim3(256), 0, stream, Zc, Kz, Cm, L, Mp, d, \
                         P, NS, p0, B, bchunk, small_limit, w, q, rowA, colB, partB, partC);      \
    else                                                                                           \
      hipLaunchKernelGGL((k_qred_f64_mfma<KS_, DG_, false>), grid, dim3(256), 0, stream, Zc, Kz, Cm, L, Mp, d, \
                         P, NS, p0, B, bchunk, small_limit, w, q, rowA, colB, partB, partC);      \
  } while (0)
#define MM_LAUNCH_F64_KS(DG_)                                   \
  do {                                                          \
    if (ks4 <= 1) MM_LAUNCH_F64(1, DG_);                        \
    else if (ks4 == 2) MM_LAUNCH_F64(2, DG_);                   \
    else if (ks4 == 3) MM_LAUNCH_F64(3, DG_);                   \
    else if (ks4 == 4) MM_LAUNCH_F64(4, DG_);                   \
    else if (ks4 <= 6) MM_LAUNCH_F64(6, DG_);                   \
    else MM_LAUNCH_F64(8, DG_);                                 \
  } while (0)
  if (diag) MM_LAUNCH_F64_KS(true); else MM_LAUNCH_F64_KS(false);
#undef MM_LAUNCH_F64_KS
#undef MM_LAUNCH_F64
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
