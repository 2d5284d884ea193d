// f64 MFMA fused reduce (gfx950): the diagonal kernel pairs (a == a') of every mode, and the
// off-diagonal pairs of the f64 mode.
//
// Diagonal pairs carry the C-weighted term  sum_ij C_ij q_i exp(delta_ij) q_j  of
// models.py:254-261 (tr(W) and sum(W o q_cov)); C = Kuu^-1 S Kuu^-1 - Kuu^-1 has norm up to
// 1/jitter = 1e6, so any per-entry f32 rounding of exp(delta) is amplified beyond use
// (DESIGN.md "fp32 error budget") -- these pairs are reduced in f64 in both modes.
//
//   tile      : 64 x 64 entries per workgroup (4 waves, 32 x 32 each = 2 x 2 MFMA tiles of
//               v_mfma_f64_16x16x4_f64; rho_i + gamma'_j initialises the accumulator)
//   diag mode : only tile pairs it <= jt are visited (Q_aa and C_a are symmetric; strictly
//               upper tiles count twice); the C tile is loaded ONCE into registers and the
//               workgroup loops over its chunk of the batch -- C traffic is M^2*8 B per chunk
//               instead of per batch element; with model uncertainty ONE fused sum
//               sum_ij q_i q_j (D_ij expm1(delta_ij) + C_ij) = sum_ij qh_i qh'_j D_ij e^{b_ij} - (sum_i w_i)^2,
//               D = C + beta beta^T, with the rank-one parts of delta_ij = rho_i + gamma'_j + b_ij FACTORED into the
//               weights (qh_i = q_i e^{rho_i}, qh'_j = q_j e^{gamma'_j}: k_pairvec): no accumulator initialisation,
//               |b| < |delta| (lower Taylor tiers), 10 fewer operand loads per lane and batch element, and only
//               the D tile in registers;
//   expm1     : wave-uniform Taylor tiers by the tile's range (degree 6/7/8/9/10/15 for an f32 model,
//               7/8/9/10/12/15 for an f64 model), Horner steps vertical over 8 entries; beyond the last
//               tier k ln2 + r reduction + v_ldexp_f64 (relative error ~2e-16 for every argument);
//   pipeline  : operands of batch element b + 1 prefetched into a second register set; per-thread
//               partials of 16 batch elements staged in LDS and reduced together;
//   grid      : 1-D, remapped so that an XCD owns a contiguous (pair, chunk, tile) range;
//   output    : per (b, pair, tile) partial sums -> slab (deterministic, no atomics).
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"
#include "mm_exp_f64.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
// Taylor degrees of the f32-mode (LOWP) tiers |x| <= 1/64, 1/16, 1/4 (the half steps 1/32, 1/8 take D0 + 1, D1 + 1)
#ifndef MM_LOWP_D0
#define MM_LOWP_D0 6
#define MM_LOWP_D1 8
#define MM_LOWP_D2 10
#endif
#ifndef MM_F64_TARGET_WGS
#define MM_F64_TARGET_WGS 8192
#endif
#define MM_F64_NB 16      // batch elements whose partial sums are staged in LDS between workgroup reductions
// diagonal pairs with the rank-one terms factored into the weights need ~100 fewer VGPRs: three waves per SIMD
// up to this many K = 4 steps (d <= 4 MM_F64_3WAVE_KS4)
#ifndef MM_F64_3WAVE_KS4
#define MM_F64_3WAVE_KS4 2
#endif

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
  x = fmin(x, MM_EXP_CAP_F64);                   // (mm_common.h: a zero weight must meet a finite factor)
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

// Per-b operands of one wave's 32 x 32 sub-tile (prefetched one batch element ahead).
template <int KS4, bool DIAG>
struct MMF64Operands {
  double rho[DIAG ? 1 : 2][DIAG ? 1 : 4];      // rho_i of the rows (rt, kq + 4 r)      (off-diagonal pairs only)
  double rw[2][4];       // row weight: factored q_i e^{rho_i} / w_i e^{rho_i} (diagonal pairs) or w_i
  double breg[2][KS4];   // MFMA B operand: g_j components of column (ct, l15)
  double gam[DIAG ? 1 : 2];                    // gamma'_j                                (off-diagonal pairs only)
  double cw[2];          // column weight: factored (diagonal pairs) or w'_j
};

// KS4: number of K=4 MFMA steps covering the d input dimensions.  LOWP: f32 mode (the diagonal pairs
// of an f32 model: expm1 to ~1e-14 relative instead of ~1e-17).
// grid: 1-D over (pair of this launch, batch chunk, tile pair); tile pairs: nt(nt+1)/2 upper pairs on
//       the diagonal, else nt*nt.  Row/col operand arrays are indexed by the local pair index.
//
// Diagonal pairs with model uncertainty reduce ONE fused sum
//     sum_ij q_i q_j (D_ij expm1(delta_ij) + C_ij),   D = C + beta beta^T   (w_i = beta_i q_i),
// i.e. sum w expm1 w' + sum C q exp(delta) q' with two FMAs per entry instead of three; D is formed
// once per workgroup from the C tile.  delta_ij = rho_i + gamma'_j + zc_i . g_j: the two O(1) terms
// initialise the MFMA accumulator (one add per entry; an extra K step would cost a whole MFMA).
// (the kernel's body as a device function of the block index `orig`: k_qred_f64_mfma below runs it for one kind of pair,
// k_qred_f64_both for the diagonal AND the off-diagonal pairs of a small f64 model in one launch)
template <int KS4, bool DIAG, bool WITHC, bool LOWP>
__device__ __forceinline__ void mmq_f64_body(const double* __restrict__ Zc, int Kz,
                                             const double* __restrict__ Cm,
                                             const double* __restrict__ beta, int M,
                                             int L, int Mp, int d, int P, int NS, int p0,
                                             int B, int bchunk, int np, int nslots, int nchunk, int nwork,
                                             int force_worst,
                                             const double* __restrict__ w,
                                             const double* __restrict__ q,
                                             const double* __restrict__ rowA,
                                             const double* __restrict__ colB,
                                             double* __restrict__ partB,
                                             double* __restrict__ partC, int orig, double (*stage)[256], double (*part16)[16]) {
  // 1-D grid, XCD-aware: workgroups b and b + 8 share an XCD (round-robin dispatch), so the remap
  // hands every XCD one contiguous range of work items ordered (pair, batch chunk, tile): the
  // per-batch-element operands of one latent stay in ONE XCD's L2.  Bijective for any nwork.
  const int nt = Mp / MM_F64_TILE;
  const int xcd = orig & 7, slot = orig >> 3;
  const int qn = nwork >> 3, rn = nwork & 7;
  const int wi = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  const int tile = wi % nslots;
  const int chunk = (wi / nslots) % nchunk;
  const int lp = wi / (nslots * nchunk);
  const int p = p0 + lp;
  int it, jt;
  if (DIAG) {                       // tile -> (it <= jt), row-major over the upper triangle
    int r = tile; it = 0;
    while (r >= nt - it) { r -= nt - it; ++it; }
    jt = it + r;
  } else {
    it = tile / nt; jt = tile - it * nt;
  }
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
  // D = C + beta beta^T tile -> registers (element (row(rt, r), col(ct)) in the MFMA accumulator layout)
  double dreg[2][2][4];
  if (withC) {
    double bcol[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int col = cbase + ct * 16 + l15;
      bcol[ct] = beta[(size_t)a * M + (col < M ? col : M - 1)];
      if (col >= M) bcol[ct] = 0.0;
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rbase + rt * 16 + kq + 4 * r;
        double brow = beta[(size_t)a * M + (row < M ? row : M - 1)];
        if (row >= M) brow = 0.0;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const double cv = Cm[((size_t)a * Mp + row) * Mp + cbase + ct * 16 + l15];
          dreg[rt][ct][r] = fma(brow, bcol[ct], cv);
        }
      }
  }
  size_t boff[KS4];
#pragma unroll
  for (int s = 0; s < KS4; ++s) {
    const int k = 4 * s + kq;
    boff[s] = (size_t)(k < d ? k : d) * Mp;
  }
  const size_t goff = (size_t)d * Mp;

  const int b0 = chunk * bchunk;
  const int b1 = (b0 + bchunk < B) ? b0 + bchunk : B;
  // per-b operand pointers advance by constant strides (no 64-bit multiplies in the loop)
  const double* ra = rowA + ((size_t)b0 * np + lp) * Mp;
  const double* cb = colB + ((size_t)b0 * np + lp) * (size_t)(d + 1) * Mp;
  // diagonal pairs: `w` / `q` are the factored row / column weights (workspace qhR / qhC), indexed by latent
  const double* rwp = w + ((size_t)b0 * L + a) * Mp;
  const double* cwp = (DIAG ? q : w) + ((size_t)b0 * L + a2) * Mp;
  const size_t st_ra = (size_t)np * Mp, st_cb = (size_t)np * (d + 1) * Mp, st_w = (size_t)L * Mp;

  auto load_ops = [&](MMF64Operands<KS4, DIAG>& o) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rbase + rt * 16 + kq + 4 * r;
        if constexpr (!DIAG) o.rho[rt][r] = ra[row];
        o.rw[rt][r] = rwp[row];
      }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int col = cbase + ct * 16 + l15;
#pragma unroll
      for (int s = 0; s < KS4; ++s) {
        o.breg[ct][s] = cb[boff[s] + col];     // k >= d reads the (finite) gamma row against a zero A operand
      }
      if constexpr (!DIAG) o.gam[ct] = cb[goff + col];
      o.cw[ct] = cwp[col];
    }
  };

  // sum the staged partials of batch elements [bs, bs + nb): 16 threads per element sum 16 values
  // each (rotated start: LDS bank spread), then one thread per element sums the 16 partials.
  // Fixed order => bitwise reproducible.
  auto flush = [&](int bs, int nb) {
    __syncthreads();
    const int bl = threadIdx.x >> 4, j = threadIdx.x & 15;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += stage[bl][j * 16 + ((k + j) & 15)];
    part16[bl][j] = acc;
    __syncthreads();
    if ((int)threadIdx.x < nb) {
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) tot += part16[threadIdx.x][k];
      const int b = bs + threadIdx.x;
      partB[((size_t)b * P + p) * NS + tile] = sym * tot;
      if (withC) partC[((size_t)b * L + a) * NS + tile] = 0.0;    // fused into partB
    }
  };

  auto reduce_b = [&](int b, const MMF64Operands<KS4, DIAG>& o) {
    f64x4 cacc[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f64x4 c;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = DIAG ? 0.0 : o.rho[rt][r] + o.gam[ct];
#pragma unroll
        for (int s = 0; s < KS4; ++s)
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[rt][s], o.breg[ct][s], c, 0, 0, 0);
        cacc[rt][ct] = c;
      }
    // wave-uniform choice of the expm1 form.  |x| is ordered like its high dword (sign cleared) as an
    // unsigned integer, so the range test runs on 32-bit integer max; comparing against the high
    // dword of the (power-of-two) limits with >= errs to the higher-degree side at the boundary.
    unsigned int mxh = mm_absmax_hi32(cacc);                       // (mm_exp_f64.h: one v_max3_f32 per entry pair)
#define MM_HI32(x_) ((unsigned int)(__builtin_bit_cast(unsigned long long, (double)(x_)) >> 32))
    if (force_worst) mxh = 0x7ff00000u;                          // MM_FORCE_WORST_TIER: wave-uniform override
    double sv = 0.0;
    // DIAG (factored weights): the entry contributes [D_ij] e^{b_ij} (the caller subtracts (sum w)^2 once);
    // off-diagonal pairs of the f64 mode: expm1(delta_ij)
#define MM_F64_ACCUM(EXPM1_)                                                              \
    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) {                                    \
      double pv = 0.0;                                                                    \
      _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                    \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                   \
          const double e = DIAG ? mm_exp_f64(cacc[rt][ct][r]) : EXPM1_(cacc[rt][ct][r]);  \
          if (withC) pv = fma(dreg[rt][ct][r] * e, o.rw[rt][r], pv);                      \
          else pv = fma(o.rw[rt][r], e, pv);                                              \
        }                                                                                 \
      sv = fma(pv, o.cw[ct], sv);                                                         \
    }
    // Taylor tiers: the Horner steps run "vertically" over the 8 entries of a column block, so
    // consecutive v_fma_f64 are independent (a per-entry chain stalls on the f64 FMA latency with
    // only two waves per SIMD to cover it).
#define MM_F64_ACCUM_POLY(DEG_)                                                           \
    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) {                                    \
      double pp[8], xv[8];                                                                \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                     \
        xv[i] = cacc[i >> 2][ct][i & 3];                                                  \
        pp[i] = fma(mm_inv_fact(DEG_), xv[i], mm_inv_fact(DEG_ - 1));                     \
      }                                                                                   \
      _Pragma("unroll") for (int k = DEG_ - 2; k >= 1; --k)                               \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) pp[i] = fma(pp[i], xv[i], mm_inv_fact(k)); \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) pp[i] = DIAG ? fma(pp[i], xv[i], 1.0) : pp[i] * xv[i]; \
      double pa = 0.0, pb = 0.0;                                                          \
      if (withC) {                                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) pp[i] *= dreg[i >> 2][ct][i & 3];   \
      }                                                                                   \
      _Pragma("unroll") for (int i = 0; i < 8; i += 2) {                                  \
        pa = fma(pp[i], o.rw[i >> 2][i & 3], pa);                                         \
        pb = fma(pp[i + 1], o.rw[(i + 1) >> 2][(i + 1) & 3], pb);                         \
      }                                                                                   \
      sv = fma(pa + pb, o.cw[ct], sv);                                                    \
    }
    // the same with a coefficient table (near-minimax LOWP tiers): degree N_
#define MM_F64_ACCUM_POLYC(N_, TBL_)                                                      \
    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) {                                    \
      double pp[8], xv[8];                                                                \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                     \
        xv[i] = cacc[i >> 2][ct][i & 3];                                                  \
        pp[i] = fma(MMExpMM::TBL_[N_ - 1], xv[i], MMExpMM::TBL_[N_ - 2]);                 \
      }                                                                                   \
      _Pragma("unroll") for (int k = N_ - 3; k >= 0; --k)                                 \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) pp[i] = fma(pp[i], xv[i], MMExpMM::TBL_[k]); \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) pp[i] = DIAG ? fma(pp[i], xv[i], 1.0) : pp[i] * xv[i]; \
      double pa = 0.0, pb = 0.0;                                                          \
      if (withC) {                                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) pp[i] *= dreg[i >> 2][ct][i & 3];   \
      }                                                                                   \
      _Pragma("unroll") for (int i = 0; i < 8; i += 2) {                                  \
        pa = fma(pp[i], o.rw[i >> 2][i & 3], pa);                                         \
        pb = fma(pp[i + 1], o.rw[(i + 1) >> 2][(i + 1) & 3], pb);                         \
      }                                                                                   \
      sv = fma(pa + pb, o.cw[ct], sv);                                                    \
    }
    // Taylor degree by range: absolute truncation |x|^(DEG+1) / (DEG+1)!
    //   LOWP (f32 model): 1/64 -> 6 (5e-17), 1/32 -> 7 (2e-17), 1/16 -> 8 (4e-17), 1/8 -> 9 (3e-16), 1/4 -> 10 (6e-15), 3/4 -> 15
    //   f64 model       : 1/64 -> 7 (7e-20), 1/32 -> 8 (1e-19), 1/16 -> 9 (3e-19), 1/8 -> 10 (2e-18), 1/4 -> 12 (2e-18), 1/2 -> 15
    // Along the C3 rollout the 32 x 32 wave tiles of the diagonal pairs have max |delta| in (1/64, 1/32] 22-33 %,
    // (1/32, 1/16] 12-20 %, (1/16, 1/8] 47-55 %, (1/8, 1/4] 0-12 %: the half steps save one FMA per entry on 80 %.
    constexpr bool MMX = LOWP && MM_LOWP_MINIMAX;
    if (!__any(mxh >= MM_HI32(0.015625))) {
      if (MMX) { MM_F64_ACCUM_POLYC(5, t0) } else if (LOWP) { MM_F64_ACCUM_POLY(MM_LOWP_D0) } else { MM_F64_ACCUM_POLY(7) }
    } else if (!__any(mxh >= MM_HI32(0.03125))) {
      if (MMX) { MM_F64_ACCUM_POLYC(6, t1) } else if (LOWP) { MM_F64_ACCUM_POLY(MM_LOWP_D0 + 1) } else { MM_F64_ACCUM_POLY(8) }
    } else if (!__any(mxh >= MM_HI32(0.0625))) {
      if (MMX) { MM_F64_ACCUM_POLYC(7, t2) } else if (LOWP) { MM_F64_ACCUM_POLY(MM_LOWP_D1) } else { MM_F64_ACCUM_POLY(9) }
    } else if (!__any(mxh >= MM_HI32(0.125))) {
      if (MMX) { MM_F64_ACCUM_POLYC(8, t3) } else if (LOWP) { MM_F64_ACCUM_POLY(MM_LOWP_D1 + 1) } else { MM_F64_ACCUM_POLY(10) }
    } else if (!__any(mxh >= MM_HI32(0.25))) {
      if (MMX) { MM_F64_ACCUM_POLYC(9, t4) } else if (LOWP) { MM_F64_ACCUM_POLY(MM_LOWP_D2) } else { MM_F64_ACCUM_POLY(12) }
    } else if (!__any(mxh >= (LOWP ? MM_HI32(0.75) : MM_HI32(0.5)))) {
      if (MMX) { MM_F64_ACCUM_POLYC(12, t5) } else { MM_F64_ACCUM_POLY(15) }
    } else {
      MM_F64_ACCUM(mm_expm1_f64)
    }
#undef MM_F64_ACCUM
#undef MM_F64_ACCUM_POLY
#undef MM_F64_ACCUM_POLYC
#undef MM_HI32
    // the workgroup reduction is deferred: per-thread partials of MM_F64_NB batch elements are
    // staged in LDS and reduced together (no cross-lane traffic or barrier per batch element)
    stage[(b - b0) % MM_F64_NB][threadIdx.x] = sv;
    if ((b - b0) % MM_F64_NB == MM_F64_NB - 1 || b == b1 - 1) flush(b - (b - b0) % MM_F64_NB, (b - b0) % MM_F64_NB + 1);
  };

  // two operand sets in registers: batch element b + 1 is in flight while b is reduced
  // (the prefetch is unconditional -- past the end it re-reads the last element -- so that the
  // compiler's s_waitcnt for the current set does not have to cover a maybe-not-issued prefetch)
  MMF64Operands<KS4, DIAG> o0, o1;
  auto advance = [&](bool more) {
    const size_t m = more ? 1 : 0;
    ra += m * st_ra; cb += m * st_cb; rwp += m * st_w; cwp += m * st_w;
  };
  load_ops(o0);
  for (int b = b0; b < b1; b += 2) {
    advance(b + 1 < b1);
    load_ops(o1);
    reduce_b(b, o0);
    advance(b + 2 < b1);
    load_ops(o0);
    if (b + 1 < b1) reduce_b(b + 1, o1);
  }
}

template <int KS4, bool DIAG, bool WITHC, bool LOWP>
__global__ __launch_bounds__(256, (KS4 >= 6 ? 1 : (DIAG && KS4 <= MM_F64_3WAVE_KS4 ? 3 : 2))) void k_qred_f64_mfma(const double* __restrict__ Zc, int Kz,
                                                          const double* __restrict__ Cm,
                                                          const double* __restrict__ beta, int M,
                                                          int L, int Mp, int d, int P, int NS, int p0,
                                                          int B, int bchunk, int np, int nslots, int nchunk, int nwork,
                                                          int force_worst,
                                                          const double* __restrict__ w,
                                                          const double* __restrict__ q,
                                                          const double* __restrict__ rowA,
                                                          const double* __restrict__ colB,
                                                          double* __restrict__ partB,
                                                          double* __restrict__ partC) {
  __shared__ double stage[MM_F64_NB][256];
  __shared__ double part16[MM_F64_NB][16];
  mmq_f64_body<KS4, DIAG, WITHC, LOWP>(Zc, Kz, Cm, beta, M, L, Mp, d, P, NS, p0, B, bchunk, np, nslots, nchunk, nwork, force_worst,
                                       w, q, rowA, colB, partB, partC, (int)blockIdx.x, stage, part16);
}

// Both reduces of a SMALL f64 model in one launch (cartpole sizes: every kernel of the step costs ~ 4-5 us whatever it does, so
// the step is its number of launches): blocks [0, nwork_d) are the diagonal pairs' work items, the rest the off-diagonal pairs'.
struct MMF64Side {
  int p0, bchunk, np, nslots, nchunk, nwork;
  const double *w, *q, *rowA, *colB;
};
template <int KS4, bool WITHC>
__global__ __launch_bounds__(256, (KS4 >= 6 ? 1 : 2)) void k_qred_f64_both(const double* __restrict__ Zc, int Kz,
                                                                           const double* __restrict__ Cm,
                                                                           const double* __restrict__ beta, int M, int L, int Mp, int d,
                                                                           int P, int NS, int B, int force_worst, MMF64Side sd,
                                                                           MMF64Side so, double* __restrict__ partB,
                                                                           double* __restrict__ partC) {
  __shared__ double stage[MM_F64_NB][256];
  __shared__ double part16[MM_F64_NB][16];
  const int orig = (int)blockIdx.x;
  if (orig < sd.nwork)
    mmq_f64_body<KS4, true, WITHC, false>(Zc, Kz, Cm, beta, M, L, Mp, d, P, NS, sd.p0, B, sd.bchunk, sd.np, sd.nslots, sd.nchunk, sd.nwork,
                                          force_worst, sd.w, sd.q, sd.rowA, sd.colB, partB, partC, orig, stage, part16);
  else
    mmq_f64_body<KS4, false, false, false>(Zc, Kz, nullptr, beta, M, L, Mp, d, P, NS, so.p0, B, so.bchunk, so.np, so.nslots, so.nchunk,
                                           so.nwork, force_worst, so.w, so.q, so.rowA, so.colB, partB, partC, orig - sd.nwork, stage,
                                           part16);
}

int mm_f64_num_slots(int Mp, int diag) {
  const int nt = Mp / MM_F64_TILE;
  return diag ? nt * (nt + 1) / 2 : nt * nt;
}

// Launch over `npairs` pairs starting at global pair index p0.  diag != 0: pairs are (a, a).
// lowp != 0: the model is f32 (diagonal pairs only need ~1e-14 relative accuracy of expm1).
// batch chunking of one side (the rule of mm_launch_qred_f64 below)
static void mm_f64_side_chunks(int nslots, int npairs, int B, int diag, int& bchunk, int& nchunk) {
  long long per_chunk = (long long)nslots * npairs;
  nchunk = (int)((MM_F64_TARGET_WGS + per_chunk - 1) / per_chunk);
  if (nchunk > B / 16) nchunk = B / 16;
  if (nchunk < 1) nchunk = 1;
  if (!diag) {
    nchunk = (int)((6144 + per_chunk - 1) / per_chunk);
    const int cap = (B + 3) / 4;
    if (nchunk > cap) nchunk = cap;
    if (nchunk < 1) nchunk = 1;
  }
  bchunk = (B + nchunk - 1) / nchunk;
  nchunk = (B + bchunk - 1) / bchunk;
}

// Diagonal AND off-diagonal pairs of an f64 model in ONE launch, where that is worth more than the diagonal kernel's third wave
// per SIMD: small models (<= MM_F64_BOTH_MAX_WGS workgroups in total, d <= 16).  Returns 1 if it launched, 0 if the caller should
// use the two launches, < 0 / > 0 on error.
#ifndef MM_F64_BOTH_MAX_WGS
#define MM_F64_BOTH_MAX_WGS 1024
#endif
int mm_launch_qred_f64_both(const double* Zc, int Kz, const double* Cm, const double* beta, int M, int L, int Mp, int d, int P, int NS,
                            int Po, int B, int force_worst, const double* qhR, const double* qhC, const double* rowD,
                            const double* colD, const double* w64, const double* q64, const double* rowO, const double* colO,
                            double* partB, double* partC, hipStream_t stream, bool* launched) {
  *launched = false;
  if (Po <= 0 || d > 16) return 0;
  MMF64Side sd, so;
  sd.p0 = 0; sd.np = L; sd.nslots = mm_f64_num_slots(Mp, 1);
  so.p0 = L; so.np = Po; so.nslots = mm_f64_num_slots(Mp, 0);
  if (sd.nslots > NS || so.nslots > NS) return MM_E_WORKSPACE;
  mm_f64_side_chunks(sd.nslots, L, B, 1, sd.bchunk, sd.nchunk);
  mm_f64_side_chunks(so.nslots, Po, B, 0, so.bchunk, so.nchunk);
  const long long nd = (long long)sd.nslots * L * sd.nchunk, no = (long long)so.nslots * Po * so.nchunk;
  if (nd + no > MM_F64_BOTH_MAX_WGS) return 0;
  sd.nwork = (int)nd; so.nwork = (int)no;
  sd.w = qhR; sd.q = qhC; sd.rowA = rowD; sd.colB = colD;
  so.w = w64; so.q = q64; so.rowA = rowO; so.colB = colO;
  const int ks4 = (d + 3) / 4;
  dim3 grid((unsigned)(nd + no));
#define MM_BOTH_(KS_, WC_) hipLaunchKernelGGL((k_qred_f64_both<KS_, WC_>), grid, dim3(256), 0, stream, Zc, Kz, Cm, beta, M, L, Mp, d, P, \
                                             NS, B, force_worst, sd, so, partB, partC)
#define MM_BOTH(KS_) do { if (Cm) MM_BOTH_(KS_, true); else MM_BOTH_(KS_, false); } while (0)
  if (ks4 <= 1) MM_BOTH(1); else if (ks4 == 2) MM_BOTH(2); else if (ks4 == 3) MM_BOTH(3); else MM_BOTH(4);
#undef MM_BOTH
#undef MM_BOTH_
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  *launched = true;
  return 0;
}

int mm_launch_qred_f64(const double* Zc, int Kz, const double* Cm, const double* beta, int M, int L, int Mp, int d,
                       int P, int NS, int p0, int npairs, int B, int diag, int lowp, int force_worst,
                       const double* w, const double* q, const double* rowA, const double* colB,
                       double* partB, double* partC, hipStream_t stream) {
  if (npairs <= 0) return 0;
  const int nslots = mm_f64_num_slots(Mp, diag);
  if (nslots > NS) return MM_E_WORKSPACE;
  // batch chunk: enough workgroups to fill the chip evenly (>= ~32 per CU), as few C re-reads as possible
  long long per_chunk = (long long)nslots * npairs;
  int nchunk = (int)((MM_F64_TARGET_WGS + per_chunk - 1) / per_chunk);    // >= 16 rounds of 512 resident workgroups: short tail
  if (nchunk > B / 16) nchunk = B / 16;      // keep >= 16 batch elements per C tile load
  if (nchunk < 1) nchunk = 1;
  if (!diag) {                               // no C reuse to protect: ~24 workgroups per CU, but chunks long
    nchunk = (int)((6144 + per_chunk - 1) / per_chunk);   // enough (>= 4) for the operand prefetch to pay
    const int cap = (B + 3) / 4;
    if (nchunk > cap) nchunk = cap;
    if (nchunk < 1) nchunk = 1;
  }
  const int bchunk = (B + nchunk - 1) / nchunk;
  nchunk = (B + bchunk - 1) / bchunk;
  const long long nwork_ll = (long long)nslots * npairs * nchunk;
  if (nwork_ll > 0x7fffffffLL) return MM_E_DIM;
  const int nwork = (int)nwork_ll;
  dim3 grid(nwork);
  const int ks4 = (d + 3) / 4;
#define MM_LAUNCH_F64_(KS_, DG_, WC_, LP_)                                                                  \
  hipLaunchKernelGGL((k_qred_f64_mfma<KS_, DG_, WC_, LP_>), grid, dim3(256), 0, stream, Zc, Kz, Cm, beta, M, \
                     L, Mp, d, P, NS, p0, B, bchunk, npairs, nslots, nchunk, nwork, force_worst, w, q, rowA, colB, partB, partC)
#define MM_LAUNCH_F64(KS_, DG_)                                                  \
  do {                                                                           \
    const bool wc = DG_ && Cm != nullptr;                                        \
    if (wc && lowp) MM_LAUNCH_F64_(KS_, DG_, true, true);                        \
    else if (wc) MM_LAUNCH_F64_(KS_, DG_, true, false);                          \
    else if (lowp) MM_LAUNCH_F64_(KS_, DG_, false, true);                        \
    else MM_LAUNCH_F64_(KS_, DG_, false, false);                                 \
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
#undef MM_LAUNCH_F64_
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
