// Backward of the fused reduce w.r.t. the input moments, stage A: the M x M
// sweeps.  The reference differentiates the whole rollout with tf.GradientTape
// (gpflow_pilco/utils/optimizers.py:52-56 through moment_matching/models.py:200-299); here the
// M^2-sized part of that derivative is reduced on the GPU to M-sized vectors, and the remaining
// O(M d^2) algebra is done by torch autograd on a surrogate (gpflowpilco_amd/autodiff.py).
//
// With Omega_ij = (w_i w'_j + [a == a'] C_ij q_i q_j) exp(delta_ij), E_ij = expm1(delta_ij):
//   column sums (thread = column j, rows uniform):
//     Ksum_j = sum_i Omega_ij     csum_j = sum_i w_i E_ij     Usum_j = sum_i Omega_ij zeta_i   [d]
//     cC_j   = sum_i C_ij q_i exp(delta_ij)                    (diagonal pairs only)
//   row sums (thread = row i, columns uniform; off-diagonal pairs only -- diagonal pairs are symmetric):
//     Rsum_i = sum_j Omega_ij     rsum_i = sum_j E_ij w'_j
// The sums are taken in f64 on an f64 pack (f32 models differentiate through the f64 pack of the same
// model, autodiff.py).  Two kernels: k_bwd_mfma (below; f64 matrix pipe, d <= 31) and the first version
// k_bwd_sums (one thread per column, kept as the cross-check behind MM_FORCE_GENERIC and for d = 32).
// Must follow mm_q_forward / mm_moment_match on the same workspace (w, q and the streamed operands).
#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>
#include "mm_common.h"
#include "mm_fork.h"
#include "mm_exp_f64.h"

__device__ __forceinline__ void mmb_decode_pair(int p, int L, int& a, int& a2) {
  if (p < L) { a = p; a2 = p; return; }
  int r = p - L, i = 0;
  while (r >= L - 1 - i) { r -= L - 1 - i; ++i; }
  a = i; a2 = i + 1 + r;
}

__device__ constexpr double mmb_inv_fact(int n) { double f = 1.0; for (int i = 2; i <= n; ++i) f *= i; return 1.0 / f; }

// ROWS == false: grid (Mp/256, P, B), thread owns column j.  out_col [B][P][3 + d][Mp]: K, c, cC, U.
// ROWS == true : grid (Mp/256, Po, B), thread owns row i of pair L + blockIdx.y.  out_row [B][Po][2][Mp].
template <int DK, bool ROWS>
__global__ __launch_bounds__(256) void k_bwd_sums(const double* __restrict__ Z64, const double* __restrict__ Zc, int Kz,
                                                  const double* __restrict__ Cm, const double* __restrict__ mu,
                                                  int L, int M, int Mp, int d, int P,
                                                  const double* __restrict__ w, const double* __restrict__ q,
                                                  const double* __restrict__ rowD, const double* __restrict__ colD,
                                                  const double* __restrict__ rowO, const double* __restrict__ colO,
                                                  double* __restrict__ out) {
  const int Po = P - L;
  const int lp = blockIdx.y, b = blockIdx.z;
  const int p = ROWS ? L + lp : lp;
  int a, a2;
  mmb_decode_pair(p, L, a, a2);
  const bool diag = p < L;
  const int t = blockIdx.x * 256 + threadIdx.x;      // column j (or row i)
  if (t >= Mp) return;
  const double* ra = diag ? rowD + ((size_t)b * L + p) * Mp : rowO + ((size_t)b * Po + (p - L)) * Mp;
  const double* cb = diag ? colD + ((size_t)b * L + p) * (size_t)(d + 1) * Mp
                          : colO + ((size_t)b * Po + (p - L)) * (size_t)(d + 1) * Mp;
  const double* wr = w + ((size_t)b * L + a) * Mp;
  const double* wc = w + ((size_t)b * L + a2) * Mp;
  const double* qr = q + ((size_t)b * L + a) * Mp;
  const double* zrow = Zc + (size_t)a * Mp * Kz;
  if (!ROWS) {
    double g[DK], U[DK];
#pragma unroll
    for (int k = 0; k < DK; ++k) { g[k] = (k < d) ? cb[(size_t)k * Mp + t] : 0.0; U[k] = 0.0; }
    const double gam = cb[(size_t)d * Mp + t];
    const bool withC = diag && (Cm != nullptr);
    double Ks = 0.0, cs = 0.0, cC = 0.0, UC[DK];
#pragma unroll
    for (int k = 0; k < DK; ++k) UC[k] = 0.0;
    for (int i = 0; i < M; ++i) {
      double delta = ra[i] + gam;
#pragma unroll
      for (int k = 0; k < DK; ++k) if (k < d) delta += zrow[(size_t)i * Kz + k] * g[k];
      const double E = expm1(fmin(delta, MM_EXP_CAP_F64)), e = E + 1.0;     // (mm_common.h: exponent caps)
      const double wi = wr[i];
      const double om = wi * e;                       // Omega_ij / w'_j
      Ks += om;
      cs += wi * E;
      double omC = 0.0;
      if (withC) { omC = Cm[((size_t)a * Mp + i) * Mp + t] * qr[i] * e; cC += omC; }
#pragma unroll
      for (int k = 0; k < DK; ++k) {
        if (k < d) {
          const double zi = Z64[((size_t)a * M + i) * d + k] - mu[(size_t)b * d + k];
          U[k] += om * zi;
          UC[k] += omC * zi;
        }
      }
    }
    const double wj = wc[t], qj = withC ? qr[t] : 0.0;
    double* o = out + ((size_t)b * P + p) * (size_t)(3 + d) * Mp;
    o[t] = Ks * wj + cC * qj;
    o[(size_t)Mp + t] = cs;
    o[(size_t)2 * Mp + t] = cC;
#pragma unroll
    for (int k = 0; k < DK; ++k) if (k < d) o[(size_t)(3 + k) * Mp + t] = U[k] * wj + UC[k] * qj;
  } else {
    double zi[DK];
#pragma unroll
    for (int k = 0; k < DK; ++k) zi[k] = (k < d) ? zrow[(size_t)t * Kz + k] : 0.0;
    const double rho = ra[t], wi = wr[t];
    double Rs = 0.0, rs = 0.0;
    for (int j = 0; j < M; ++j) {
      double delta = rho + cb[(size_t)d * Mp + j];
#pragma unroll
      for (int k = 0; k < DK; ++k) if (k < d) delta += zi[k] * cb[(size_t)k * Mp + j];
      const double E = expm1(fmin(delta, MM_EXP_CAP_F64));
      const double wj = wc[j];
      Rs += wj * (E + 1.0);
      rs += wj * E;
    }
    double* o = out + ((size_t)b * Po + lp) * (size_t)2 * Mp;
    o[t] = Rs * wi;
    o[(size_t)Mp + t] = rs;
  }
}

// ---------------------------------------------------------------------------------------------
// MFMA version of the same sums (d <= 31).  Structure of the forward k_qred_f64_mfma, turned into a
// COLUMN-OWNER sweep: a workgroup owns 64 columns of one (b, pair) and loops over all 64-row tiles,
// so every column sum stays in registers and nothing is reduced across workgroups (no atomics:
// bitwise reproducible).  Per wave and iteration a 32 x 32 sub-tile:
//   delta  : v_mfma_f64_16x16x4_f64, accumulator initialised with rho_i + gamma'_j (as in the forward);
//   expm1  : wave-uniform Taylor tiers of the forward's f64 mode;
//   U, K   : U_j[k] = sum_i Omega_ij zc_i[k] and K_j = sum_i Omega_ij are ONE more MFMA product,
//            Omega^T (zc | 1): the forward accumulator layout (lane = column, registers = rows
//            kq + 4 r) is exactly the A-operand layout of the transposed product, K step r -- no data
//            movement; zeta_i = zc_i + (zbar_a - mu_b) is corrected from K at the end;
//   c, cC  : one FMA per entry each, reduced over the row groups through LDS at the end.
// SWAP: the row sums of an off-diagonal pair are the column sums of the transposed tile; the same
// kernel runs with the operand roles exchanged (A operand = g_j of latent a', B operand = zc_i of
// latent a) and only the two scalar sums.
// ---------------------------------------------------------------------------------------------
typedef double f64x4b __attribute__((ext_vector_type(4)));

template <int DEG>
__device__ __forceinline__ void mmb_expm1_poly8(const double (&x)[8], double (&e)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = fma(mmb_inv_fact(DEG), x[i], mmb_inv_fact(DEG - 1));
#pragma unroll
  for (int k = DEG - 2; k >= 1; --k)
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = fma(e[i], x[i], mmb_inv_fact(k));
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] *= x[i];
}

// expm1 for any argument (k ln2 + r reduction), the forward's general path
__device__ __forceinline__ double mmb_expm1_any(double x) {
  x = fmin(x, MM_EXP_CAP_F64);                   // (mm_common.h: a zero weight must meet a finite factor)
  const double kf = rint(x * 1.4426950408889634);
  double r = fma(-kf, 6.93147180369123816490e-01, x);
  r = fma(-kf, 1.90821492927058770002e-10, r);
  double q = mmb_inv_fact(12);
#pragma unroll
  for (int k = 11; k >= 1; --k) q = fma(q, r, mmb_inv_fact(k));
  const double p = q * r;
  const double s = ldexp(1.0, (int)kf);
  return fma(s, p, s - 1.0);
}

// KS4: K = 4 steps covering d; NU: 16-wide blocks of the U product ((d + 1) <= 16 NU).
// grid (Mp / 64, npairs, B).  SWAP == false: pairs [0, P), out_col [B][P][3 + d][Mp].
//                             SWAP == true : pairs L + blockIdx.y, out_row [B][Po][2][Mp].
// (the kernel's body as a function of the block index: k_bwd_mfma runs one kind of sweep, k_bwd_mfma_both the column AND the row
// sweep of a small f64 model in one launch)
template <int KS4, int NU, bool SWAP>
__device__ __forceinline__ void mmb_mfma_body(const double* __restrict__ Zc, int Kz,
                                              const double* __restrict__ zbar, const double* __restrict__ Cm,
                                              const double* __restrict__ mu, int L, int Mp, int d, int P,
                                              const double* __restrict__ w, const double* __restrict__ q,
                                              const double* __restrict__ rowD, const double* __restrict__ colD,
                                              const double* __restrict__ rowO, const double* __restrict__ colO,
                                              double* __restrict__ out, int B, int nwork, int p0, int orig) {
  const int Po = P - L;
  // 1-D grid, XCD-aware (blocks i and i + 8 share an XCD: consecutive work items go to the same XCD), batch element fastest:
  // the B workgroups of one (pair, column tile) sweep the same 64-column strip of C_a (Mp x 64 doubles) at the same pace and
  // share it in that XCD's L2.  In (column tile, pair, b) launch order every workgroup streamed its strip from HBM --
  // B L Mp^2 x 8 bytes = 8.6 GB at C3 shape with B = 32: the kernel ran at HBM speed, not at its arithmetic.
  const int xcd = orig & 7, slotx = orig >> 3;
  const int qn = nwork >> 3, rn = nwork & 7;
  const int wi = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slotx;
  const int b = wi % B, rest = wi / B;
  const int ntc = Mp / 64;
  const int jt = rest % ntc, lp = rest / ntc;
  const int p = SWAP ? L + lp : p0 + lp;               // (p0: the column sums of the pairs [p0, P); the diagonal ones have k_bwd_diag)
  int a, a2;
  mmb_decode_pair(p, L, a, a2);
  const bool diag = p < L;
  const bool withC = !SWAP && diag && (Cm != nullptr);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  const int rh = wv >> 1, cw = wv & 1;                 // row half / column half of the 64 x 64 tile
  const int cbase = jt * 64 + cw * 32;
  const int nt = Mp / 64;

  // delta_ij = rho_i + gamma'_j + zc_i . g_j   (rho, zc: latent a, row index i;  g, gamma': latent a', column j)
  const double* rho = diag ? rowD + ((size_t)b * L + p) * Mp : rowO + ((size_t)b * Po + (p - L)) * Mp;
  const double* gcol = diag ? colD + ((size_t)b * L + p) * (size_t)(d + 1) * Mp
                            : colO + ((size_t)b * Po + (p - L)) * (size_t)(d + 1) * Mp;
  const double* zc_a = Zc + (size_t)a * Mp * Kz;
  const double* w_a = w + ((size_t)b * L + a) * Mp;
  const double* w_a2 = w + ((size_t)b * L + a2) * Mp;
  const double* q_a = q + ((size_t)b * L + a) * Mp;
  const double* Ca = Cm ? Cm + (size_t)a * Mp * Mp : nullptr;      // this latent's C (32-bit indices below: Mp^2 < 2^32)

  // ---- the workgroup's fixed side: its 32 "columns" per wave ---------------------------------
  // !SWAP: columns = j (latent a'):  B operand g_j, init gamma'_j, weights w'_j, q_j
  //  SWAP: columns = i (latent a) :  B operand zc_i, init rho_i, weight w_i
  double bfix[2][KS4], cinit[2], cwt[2], cq[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = cbase + ct * 16 + l15;
#pragma unroll
    for (int s = 0; s < KS4; ++s) {
      const int k = 4 * s + kq;
      const double v = SWAP ? zc_a[(size_t)col * Kz + (k < Kz ? k : Kz - 1)] : gcol[(size_t)(k < d ? k : d) * Mp + col];
      bfix[ct][s] = (SWAP ? k < Kz : k < d) ? v : 0.0;
    }
    cinit[ct] = SWAP ? rho[col] : gcol[(size_t)d * Mp + col];
    cwt[ct] = SWAP ? w_a[col] : w_a2[col];
    cq[ct] = withC ? q_a[col] : 0.0;
  }

  f64x4b Uacc[2][NU];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int u = 0; u < NU; ++u) Uacc[ct][u] = (f64x4b){0.0, 0.0, 0.0, 0.0};
  double pc[2] = {0.0, 0.0}, pC[2] = {0.0, 0.0}, pK[2] = {0.0, 0.0};
  double zmul[NU], zadd[NU];                              // column k = 16 u + l15 of (zc | 1 | 0 ...)
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int k = 16 * u + l15;
    zmul[u] = (k < d && k < Kz) ? 1.0 : 0.0;
    zadd[u] = (k == d) ? 1.0 : 0.0;
  }

  // The A operand and the accumulator init of iteration it + 1 are fetched while iteration it evaluates its
  // polynomials (they are the first thing an iteration needs; with two waves per SIMD their latency was a
  // stall at the top of every iteration).  Unconditional, index-clamped loads.
  auto load_first = [&](int it, double (&ar)[2][KS4], double (&ri)[2][4]) {
    const int rb = it * 64 + rh * 32;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
      for (int s = 0; s < KS4; ++s) {
        const int k = 4 * s + kq, row = rb + rt * 16 + l15;
        // 24-bit multiplies (full rate; row, Mp, Kz < 2^24): the 64-bit index products were ~40 quarter-rate
        // integer multiplies per iteration
        ar[rt][s] = SWAP ? gcol[__umul24(k < d ? k : d, Mp) + row] : zc_a[__umul24(row, Kz) + (k < Kz ? k : Kz - 1)];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rb + rt * 16 + kq + 4 * r;
        ri[rt][r] = SWAP ? gcol[__umul24(d, Mp) + row] : rho[row];
      }
    }
  };
  // (24 more live registers: kept where they fit -- KS4 <= 2 at two waves per SIMD, every NU >= 2 variant)
  constexpr bool PFB = KS4 <= 2 || NU >= 2;
  double arow_n[2][KS4], rinit_n[2][4];
  if (PFB) load_first(0, arow_n, rinit_n);

  // MMB_EXPERIMENT_HALF (timing experiment only, WRONG results): diagonal pairs visit the tiles it <= jt alone -- what a
  // symmetric sweep would compute, WITHOUT the row-side sums it would have to add per tile: a lower bound on its time
#ifdef MMB_EXPERIMENT_HALF
  const int nt_run = (!SWAP && diag) ? jt + 1 : nt;
#else
  const int nt_run = nt;
#endif
  for (int it = 0; it < nt_run; ++it) {
    const int rbase = it * 64 + rh * 32;
    // ---- this iteration's 32 "rows" -------------------------------------------------------------
    double arow[2][KS4], rinit[2][4], rwt[2][4], rq[2][4];
    if (!PFB) load_first(it, arow_n, rinit_n);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
      for (int s = 0; s < KS4; ++s) {
        const int k = 4 * s + kq;
        arow[rt][s] = (SWAP ? k < d : k < Kz) ? arow_n[rt][s] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rbase + rt * 16 + kq + 4 * r;
        rinit[rt][r] = rinit_n[rt][r];
        rwt[rt][r] = SWAP ? w_a2[row] : w_a[row];
        rq[rt][r] = q_a[row];                                // only read under withC: loaded unconditionally (no branch)
      }
    }
    // C tile and the B operand of the U product: issued here, consumed after the polynomial (their
    // latency overlaps with the delta MFMAs and the expm1 evaluation)
    double creg[2][2][4], zB[NU][8];
    if (withC) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rbase + rt * 16 + kq + 4 * r;
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) creg[rt][ct][r] = Ca[__umul24(row, Mp) + cbase + ct * 16 + l15];
        }
    }
    if (!SWAP) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int k = 16 * u + l15;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = rbase + (i >> 2) * 16 + 4 * (i & 3) + kq;
          const double zv = zc_a[__umul24(row, Kz) + (k < Kz ? k : Kz - 1)];
          zB[u][i] = fma(zv, zmul[u], zadd[u]);              // (zc | 1 | 0): lane-constant select as one FMA
        }
      }
    }
    // ---- delta tiles ------------------------------------------------------------------------------
    f64x4b acc[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f64x4b c;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = rinit[rt][r] + cinit[ct];
#pragma unroll
        for (int s = 0; s < KS4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[rt][s], bfix[ct][s], c, 0, 0, 0);
        acc[rt][ct] = c;
      }
    if (PFB) load_first(it + 1 < nt ? it + 1 : it, arow_n, rinit_n);
    const unsigned int mxh = mm_absmax_hi32(acc);                      // (mm_exp_f64.h)
#define MMB_HI32(x_) ((unsigned int)(__builtin_bit_cast(unsigned long long, (double)(x_)) >> 32))
    // Taylor degree by range (absolute truncation |x|^(D+1)/(D+1)! <= 2e-18): 1/64 -> 7, 1/32 -> 8, 1/16 -> 9,
    // 1/8 -> 10, 1/4 -> 12, 1/2 -> 15; the half steps as in mm_f64.hip (most tiles of a rollout sit in them)
    const int tier = !__any(mxh >= MMB_HI32(0.015625)) ? 0 : !__any(mxh >= MMB_HI32(0.03125)) ? 1
                   : !__any(mxh >= MMB_HI32(0.0625)) ? 2 : !__any(mxh >= MMB_HI32(0.125)) ? 3
                   : !__any(mxh >= MMB_HI32(0.25)) ? 4 : !__any(mxh >= MMB_HI32(0.5)) ? 5 : 6;
#undef MMB_HI32
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      double x[8], E[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = acc[i >> 2][ct][i & 3];
      if (tier == 0) mmb_expm1_poly8<7>(x, E);
      else if (tier == 1) mmb_expm1_poly8<8>(x, E);
      else if (tier == 2) mmb_expm1_poly8<9>(x, E);
      else if (tier == 3) mmb_expm1_poly8<10>(x, E);
      else if (tier == 4) mmb_expm1_poly8<12>(x, E);
      else if (tier == 5) mmb_expm1_poly8<15>(x, E);
      else {
#pragma unroll
        for (int i = 0; i < 8; ++i) E[i] = mmb_expm1_any(x[i]);
      }
      double om[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const double e = E[i] + 1.0;
        const double wi = rwt[i >> 2][i & 3];
        pc[ct] = fma(wi, E[i], pc[ct]);                       // sum_i w_i E_ij      (SWAP: sum_j w'_j E_ij)
        if (SWAP) {
          pK[ct] = fma(wi, e, pK[ct]);                        // sum_j w'_j e_ij
        } else {
          double o = wi * cwt[ct];                            // w_i w'_j
          if (withC) {
            const double cqi = creg[i >> 2][ct][i & 3] * rq[i >> 2][i & 3];
            pC[ct] = fma(cqi, e, pC[ct]);                     // sum_i C_ij q_i e_ij
            o = fma(cqi, cq[ct], o);
          }
          om[i] = o * e;                                      // Omega_ij
        }
      }
      if (!SWAP) {
        // U_j[k] += sum_i Omega_ij zc_i[k], K_j in column k == d:  A = Omega^T (register r = K step r)
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i)
            Uacc[ct][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(om[i], zB[u][i], Uacc[ct][u], 0, 0, 0);
      }
    }
  }

  // ---- finish: combine the row groups / row halves through LDS, write this workgroup's 64 columns ----
  __shared__ double Us[2][2][2][NU][4][64];      // [rh][cw][ct][u][r][lane]
  __shared__ double Ps[3][2][2][2][64];          // [which][rh][cw][ct][lane]
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) Us[rh][cw][ct][u][r][lane] = SWAP ? 0.0 : Uacc[ct][u][r];
    Ps[0][rh][cw][ct][lane] = pc[ct];
    Ps[1][rh][cw][ct][lane] = pC[ct];
    Ps[2][rh][cw][ct][lane] = pK[ct];
  }
  __syncthreads();
  // scalar sums: thread t < 64 owns column t of the tile
  if (threadIdx.x < 64) {
    const int c = threadIdx.x, ccw = c >> 5, cct = (c >> 4) & 1, cl = c & 15;
    double sc = 0.0, sC = 0.0, sK = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        sc += Ps[0][h][ccw][cct][16 * g + cl];
        sC += Ps[1][h][ccw][cct][16 * g + cl];
        sK += Ps[2][h][ccw][cct][16 * g + cl];
      }
    const int col = jt * 64 + c;
    if (SWAP) {
      double* o = out + ((size_t)b * Po + lp) * (size_t)2 * Mp;
      o[col] = sK * w_a[col];                  // R_i = w_i sum_j w'_j e_ij
      o[(size_t)Mp + col] = sc;                // r_i = sum_j w'_j E_ij
    } else {
      double* o = out + ((size_t)b * P + p) * (size_t)(3 + d) * Mp;
      o[(size_t)Mp + col] = sc;
      o[(size_t)2 * Mp + col] = sC;
    }
  }
  if (!SWAP) {
    // U / K: D layout of the transposed product: lane (l15 = k, kq), register r  <->  column kq + 4 r
    double* o = out + ((size_t)b * P + p) * (size_t)(3 + d) * Mp;
    for (int idx = threadIdx.x; idx < 64 * (d + 1); idx += 256) {
      const int c = idx & 63, k = idx >> 6;                   // k == d: K_j
      const int ccw = c >> 5, cct = (c >> 4) & 1, j16 = c & 15, ckq = j16 & 3, cr = j16 >> 2;
      const int u = k >> 4, kl = k & 15;
      const int ln = 16 * ckq + kl, lnK = 16 * ckq + (d & 15), uK = d >> 4;
      const double val = Us[0][ccw][cct][u][cr][ln] + Us[1][ccw][cct][u][cr][ln];
      const int col = jt * 64 + c;
      if (k == d) {
        o[col] = val;
      } else {
        const double Kj = Us[0][ccw][cct][uK][cr][lnK] + Us[1][ccw][cct][uK][cr][lnK];
        o[(size_t)(3 + k) * Mp + col] = fma(zbar[a * d + k] - mu[(size_t)b * d + k], Kj, val);
      }
    }
  }
}


template <int KS4, int NU, bool SWAP>
__global__ __launch_bounds__(256, (NU >= 2 ? 1 : 2)) void k_bwd_mfma(const double* __restrict__ Zc, int Kz,
                                                     const double* __restrict__ zbar, const double* __restrict__ Cm,
                                                     const double* __restrict__ mu, int L, int Mp, int d, int P,
                                                     const double* __restrict__ w, const double* __restrict__ q,
                                                     const double* __restrict__ rowD, const double* __restrict__ colD,
                                                     const double* __restrict__ rowO, const double* __restrict__ colO,
                                                     double* __restrict__ out, int B, int nwork, int p0) {
  mmb_mfma_body<KS4, NU, SWAP>(Zc, Kz, zbar, Cm, mu, L, Mp, d, P, w, q, rowD, colD, rowO, colO, out, B, nwork, p0, (int)blockIdx.x);
}

// Column sums (all pairs) and row sums (off-diagonal pairs) of a SMALL f64 model in one launch: the two sweeps are independent and
// a cartpole-sized one does not fill the device, so side by side they take the time of the longer one (9.7 + 6.5 -> ~ 10 us).
template <int KS4, int NU>
__global__ __launch_bounds__(256, (NU >= 2 ? 1 : 2)) void k_bwd_mfma_both(const double* __restrict__ Zc, int Kz,
                                                     const double* __restrict__ zbar, const double* __restrict__ Cm,
                                                     const double* __restrict__ mu, int L, int Mp, int d, int P,
                                                     const double* __restrict__ w, const double* __restrict__ q,
                                                     const double* __restrict__ rowD, const double* __restrict__ colD,
                                                     const double* __restrict__ rowO, const double* __restrict__ colO,
                                                     double* __restrict__ out_col, double* __restrict__ out_row, int B,
                                                     int nw_col, int nw_row) {
  const int orig = (int)blockIdx.x;
  if (orig < nw_col) mmb_mfma_body<KS4, NU, false>(Zc, Kz, zbar, Cm, mu, L, Mp, d, P, w, q, rowD, colD, rowO, colO, out_col, B, nw_col, 0, orig);
  else mmb_mfma_body<KS4, NU, true>(Zc, Kz, zbar, Cm, mu, L, Mp, d, P, w, q, rowD, colD, rowO, colO, out_row, B, nw_row, 0, orig - nw_col);
}

// ---------------------------------------------------------------------------------------------
// Diagonal pairs (a == a'), factored form (the forward's: mm_f64.hip).  e^{delta_ij} = e^{rho_i} e^{gamma'_j} e^{b_ij} with the
// rank-one parts in the workspace's factored weights qhR_i = u_i e^{rho_i}, qhC_j = u_j e^{gamma'_j} (u = q with model
// uncertainty, w without), so the polynomial argument is b_ij = zc_i . g_j alone (lower range tiers, no accumulator
// initialisation, e^b directly instead of expm1 + 1) and every sum is taken WITHOUT the column's factor, applied once per
// column at the end:
//     t_ij = qhR_i e^{b_ij}
//     with C:    S_j = sum_i beta_i t_ij,  Cs_j = sum_i C_ij t_ij,  om_ij = (C_ij + beta_i beta_j) t_ij
//                c_j = e^{gamma'_j} S_j - sum_i w_i,   cC_j = e^{gamma'_j} Cs_j,   (K_j | U_j) = qhC_j sum_i om_ij (1 | zc_i)
//     without:   om_ij = t_ij (qhR = w_i e^{rho_i}),  c_j = e^{gamma'_j} sum_i t_ij - sum_i w_i,  (K_j | U_j) = qhC_j sum_i t_ij (1 | zc_i)
// Per entry: the tier polynomial (5 ... 12 FMAs) + 5 operations, against expm1 tiers on |delta| (7 ... 15) + 7 before.
// LOWP (f32 packs): the forward's near-minimax tiers (1e-15 absolute); f64 packs: Taylor tiers (2e-18), as the forward.
// Same work decomposition, launch order and epilogue as k_bwd_mfma<., ., false>.
// ---------------------------------------------------------------------------------------------
#ifndef MMB_DIAG_WAVES
#define MMB_DIAG_WAVES 2
#endif
// NB: batch elements per workgroup -- the C tile, beta_i, zc_i are the same for every batch element; a workgroup that sweeps for
// NB of them issues those loads once (grid: ceil(B / NB) groups; an odd last group repeats its element and writes it once).
// Measured at C3 shape, B = 256 (same box): NB 1 at two waves per SIMD (205 VGPRs) 14.6 ms; NB 2 at two waves (256 VGPRs +
// 192 B scratch) 17.1; NB 2 at one wave 20.8; NB 1 at three waves (168 VGPRs + 168 B scratch) 23.7 -- the default stays
#ifndef MMB_DIAG_NB
#define MMB_DIAG_NB 1
#endif
template <int KS4, int NU, bool WITHC, bool LOWP, int NB>
__global__ __launch_bounds__(256, (NU >= 2 ? 1 : (KS4 <= 2 ? MMB_DIAG_WAVES : 2))) void k_bwd_diag(const double* __restrict__ Zc, int Kz,
                                                     const double* __restrict__ zbar, const double* __restrict__ Cm,
                                                     const double* __restrict__ beta, const double* __restrict__ mu,
                                                     int L, int M, int Mp, int d, int P,
                                                     const double* __restrict__ qhR, const double* __restrict__ qhC,
                                                     const double* __restrict__ colD, const double* __restrict__ f1raw,
                                                     double* __restrict__ out, int B, int nwork) {
  const int orig = blockIdx.x;
  const int xcd = orig & 7, slotx = orig >> 3;
  const int qn = nwork >> 3, rn = nwork & 7;
  const int wi = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slotx;
  // (the divisions run on the vector unit: readfirstlane puts the work item's coordinates -- and every base pointer derived
  // from them -- back into scalar registers, which is what lets the loads below take an SGPR base)
  const int BG = (B + NB - 1) / NB;
  const int bg = __builtin_amdgcn_readfirstlane(wi % BG), rest = wi / BG;
  const int ntc = Mp / 64;
  const int jt = __builtin_amdgcn_readfirstlane(rest % ntc), a = __builtin_amdgcn_readfirstlane(rest / ntc);
  const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rh = wv >> 1, cw = wv & 1;
  const int cbase = jt * 64 + cw * 32;
  const int nt = Mp / 64;

  int bb[NB];
  const double* gcol[NB];
  const double* qR[NB];
#pragma unroll
  for (int e = 0; e < NB; ++e) {
    bb[e] = bg * NB + e < B ? bg * NB + e : B - 1;
    gcol[e] = colD + ((size_t)bb[e] * L + a) * (size_t)(d + 1) * Mp;
    qR[e] = qhR + ((size_t)bb[e] * L + a) * Mp;
  }
  const double* zc_a = Zc + (size_t)a * Mp * Kz;
  const double* be = beta + (size_t)a * M;
  const double* Ca = WITHC ? Cm + (size_t)a * Mp * Mp : nullptr;

  double bfix[NB][2][KS4], bcol[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = cbase + ct * 16 + l15;
#pragma unroll
    for (int e = 0; e < NB; ++e)
#pragma unroll
      for (int s = 0; s < KS4; ++s) {
        const int k = 4 * s + kq;
        const double v = gcol[e][(size_t)(k < d ? k : d) * Mp + col];
        bfix[e][ct][s] = (k < d) ? v : 0.0;
      }
    bcol[ct] = (WITHC && col < M) ? be[col] : 0.0;
  }
  f64x4b Uacc[NB][2][NU];
  double pS[NB][2], pC[NB][2];
#pragma unroll
  for (int e = 0; e < NB; ++e)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int u = 0; u < NU; ++u) Uacc[e][ct][u] = (f64x4b){0.0, 0.0, 0.0, 0.0};
      pS[e][ct] = 0.0; pC[e][ct] = 0.0;
    }
  double zmul[NU], zadd[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int k = 16 * u + l15;
    zmul[u] = (k < d && k < Kz) ? 1.0 : 0.0;
    zadd[u] = (k == d) ? 1.0 : 0.0;
  }
  // Addresses: every load of the sweep is  (uniform base, advanced per iteration by scalar adds) [per-lane 32-bit offset, fixed
  // for the whole sweep] + a compile-time constant  (global_load with an SGPR base): the per-load 64-bit index arithmetic of
  // the first version was ~120 integer VALU instructions per iteration
  unsigned int offA[KS4], offZ[NU];
#pragma unroll
  for (int s = 0; s < KS4; ++s) { const int k = 4 * s + kq; offA[s] = (unsigned)((rh * 32 + l15) * Kz + (k < Kz ? k : Kz - 1)); }
#pragma unroll
  for (int u = 0; u < NU; ++u) { const int k = 16 * u + l15; offZ[u] = (unsigned)((rh * 32 + kq) * Kz + (k < Kz ? k : Kz - 1)); }
  const unsigned int offR = (unsigned)(rh * 32 + kq);                                   // rows kq + 4 r of the wave's half
  const unsigned int offC = (unsigned)((rh * 32 + kq) * Mp + cbase + l15);
  auto load_first = [&](int it, double (&ar)[2][KS4]) {
    const double* zu = zc_a + (size_t)it * 64 * Kz;                                     // uniform
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int s = 0; s < KS4; ++s) ar[rt][s] = (zu + rt * 16 * Kz)[offA[s]];
  };
  double arow_n[2][KS4];
  load_first(0, arow_n);

  // TAIL: the row tile reaches beyond M (beta is not padded): clamped, masked loads there, plain ones in the body of the sweep
  auto sweep_tile = [&](int it, auto tail_tag) {
    constexpr bool tail = decltype(tail_tag)::value;
    const double* bu = be + it * 64;                                                     // uniform bases of this row tile
    const double* zu = zc_a + (size_t)it * 64 * Kz;
    const double* cu = WITHC ? Ca + (size_t)it * 64 * Mp : nullptr;
    double arow[2][KS4], rq[NB][2][4], rb_[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
      for (int s = 0; s < KS4; ++s) arow[rt][s] = (4 * s + kq < Kz) ? arow_n[rt][s] : 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int e = 0; e < NB; ++e) rq[e][rt][r] = (qR[e] + it * 64 + rt * 16 + 4 * r)[offR];
        if (WITHC) {
          if (!tail) rb_[rt][r] = (bu + rt * 16 + 4 * r)[offR];
          else {
            const int row = it * 64 + rh * 32 + rt * 16 + kq + 4 * r;
            const double bv = be[row < M ? row : M - 1];
            rb_[rt][r] = row < M ? bv : 0.0;
          }
        }
      }
    }
    double creg[2][2][4], zB[NU][8];
    if (WITHC) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) creg[rt][ct][r] = (cu + (size_t)(rt * 16 + 4 * r) * Mp + ct * 16)[offC];
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const double zv = (zu + ((i >> 2) * 16 + 4 * (i & 3)) * Kz)[offZ[u]];
        zB[u][i] = fma(zv, zmul[u], zadd[u]);
      }
    load_first(it + 1 < nt ? it + 1 : it, arow_n);
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      f64x4b acc[2][2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          f64x4b c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < KS4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[rt][s], bfix[e][ct][s], c, 0, 0, 0);
          acc[rt][ct] = c;
        }
      const unsigned int mxh = mm_absmax_hi32(acc);
#define MMB_HI32(x_) ((unsigned int)(__builtin_bit_cast(unsigned long long, (double)(x_)) >> 32))
      const int tier = !__any(mxh >= MMB_HI32(0.015625)) ? 0 : !__any(mxh >= MMB_HI32(0.03125)) ? 1
                     : !__any(mxh >= MMB_HI32(0.0625)) ? 2 : !__any(mxh >= MMB_HI32(0.125)) ? 3
                     : !__any(mxh >= MMB_HI32(0.25)) ? 4 : !__any(mxh >= (LOWP ? MMB_HI32(0.75) : MMB_HI32(0.5))) ? 5 : 6;
#undef MMB_HI32
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        double x[8], ex[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = acc[i >> 2][ct][i & 3];
        // e^x of the 8 entries, Horner steps vertical (independent FMAs)
#define MMB_POLYC(N_, TBL_)                                                                                         \
        { _Pragma("unroll") for (int i = 0; i < 8; ++i) ex[i] = fma(MMExpMM::TBL_[N_ - 1], x[i], MMExpMM::TBL_[N_ - 2]); \
          _Pragma("unroll") for (int k = N_ - 3; k >= 0; --k)                                                        \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) ex[i] = fma(ex[i], x[i], MMExpMM::TBL_[k]);                 \
          _Pragma("unroll") for (int i = 0; i < 8; ++i) ex[i] = fma(ex[i], x[i], 1.0); }
#define MMB_POLYT(DEG_)                                                                                             \
        { _Pragma("unroll") for (int i = 0; i < 8; ++i) ex[i] = fma(mmb_inv_fact(DEG_), x[i], mmb_inv_fact(DEG_ - 1)); \
          _Pragma("unroll") for (int k = DEG_ - 2; k >= 1; --k)                                                      \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) ex[i] = fma(ex[i], x[i], mmb_inv_fact(k));                  \
          _Pragma("unroll") for (int i = 0; i < 8; ++i) ex[i] = fma(ex[i], x[i], 1.0); }
        constexpr bool MMX = LOWP && MM_LOWP_MINIMAX;
        if (tier == 0) { if (MMX) MMB_POLYC(5, t0) else MMB_POLYT(7) }
        else if (tier == 1) { if (MMX) MMB_POLYC(6, t1) else MMB_POLYT(8) }
        else if (tier == 2) { if (MMX) MMB_POLYC(7, t2) else MMB_POLYT(9) }
        else if (tier == 3) { if (MMX) MMB_POLYC(8, t3) else MMB_POLYT(10) }
        else if (tier == 4) { if (MMX) MMB_POLYC(9, t4) else MMB_POLYT(12) }
        else if (tier == 5) { if (MMX) MMB_POLYC(12, t5) else MMB_POLYT(15) }
        else {
#pragma unroll
          for (int i = 0; i < 8; ++i) ex[i] = mm_exp_f64(x[i]);
        }
#undef MMB_POLYC
#undef MMB_POLYT
        double om[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const double t = rq[e][i >> 2][i & 3] * ex[i];
          if (WITHC) {
            const double cij = creg[i >> 2][ct][i & 3], bi = rb_[i >> 2][i & 3];
            pS[e][ct] = fma(bi, t, pS[e][ct]);
            pC[e][ct] = fma(cij, t, pC[e][ct]);
            om[i] = fma(bi, bcol[ct], cij) * t;
          } else {
            om[i] = t;
          }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i)
            Uacc[e][ct][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(om[i], zB[u][i], Uacc[e][ct][u], 0, 0, 0);
      }
    }
  };
  const int nt_full = M / 64;
  for (int it = 0; it < nt_full; ++it) sweep_tile(it, std::false_type{});
  for (int it = nt_full; it < nt; ++it) sweep_tile(it, std::true_type{});

  __shared__ double Us[2][2][2][NU][4][64];      // [rh][cw][ct][u][r][lane]
  __shared__ double Ps[2][2][2][2][64];          // [which][rh][cw][ct][lane]
#pragma unroll
  for (int e = 0; e < NB; ++e) {
    if (e > 0) {
      if (bg * NB + e >= B) break;                 // (uniform) the odd last group's repeated element
      __syncthreads();
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) Us[rh][cw][ct][u][r][lane] = Uacc[e][ct][u][r];
      Ps[0][rh][cw][ct][lane] = pS[e][ct];
      Ps[1][rh][cw][ct][lane] = pC[e][ct];
    }
    __syncthreads();
    const int b = bb[e];
    const double* qC = qhC + ((size_t)b * L + a) * Mp;
    double* o = out + ((size_t)b * P + a) * (size_t)(3 + d) * Mp;      // out [B][P][3 + d][Mp]: the diagonal pairs are the first L
    const double F = f1raw[(size_t)b * L + a];
    // U / K (and, without C, c from K): D layout of the transposed product: lane (l15 = k, kq), register r <-> column kq + 4 r
    for (int idx = threadIdx.x; idx < 64 * (d + 1); idx += 256) {
      const int c = idx & 63, k = idx >> 6;                   // k == d: K_j
      const int ccw = c >> 5, cct = (c >> 4) & 1, j16 = c & 15, ckq = j16 & 3, cr = j16 >> 2;
      const int u = k >> 4, kl = k & 15;
      const int ln = 16 * ckq + kl, lnK = 16 * ckq + (d & 15), uK = d >> 4;
      const int col = jt * 64 + c;
      const double qc = qC[col];
      const double val = Us[0][ccw][cct][u][cr][ln] + Us[1][ccw][cct][u][cr][ln];
      if (k == d) {
        o[col] = qc * val;
        if (!WITHC) {
          o[(size_t)Mp + col] = fma(exp(fmin(gcol[e][(size_t)d * Mp + col], MM_EXP_CAP_F64)), val, -F);
          o[(size_t)2 * Mp + col] = 0.0;
        }
      } else {
        const double Kj = Us[0][ccw][cct][uK][cr][lnK] + Us[1][ccw][cct][uK][cr][lnK];
        o[(size_t)(3 + k) * Mp + col] = qc * fma(zbar[a * d + k] - mu[(size_t)b * d + k], Kj, val);
      }
    }
    if (WITHC && threadIdx.x < 64) {
      const int c = threadIdx.x, ccw = c >> 5, cct = (c >> 4) & 1, cl = c & 15;
      double sS = 0.0, sC = 0.0;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          sS += Ps[0][h][ccw][cct][16 * g + cl];
          sC += Ps[1][h][ccw][cct][16 * g + cl];
        }
      const int col = jt * 64 + c;
      const double eg = exp(fmin(gcol[e][(size_t)d * Mp + col], MM_EXP_CAP_F64));
      o[(size_t)Mp + col] = fma(eg, sS, -F);
      o[(size_t)2 * Mp + col] = eg * sC;
    }
  }
}

extern "C" size_t mm_backward_bytes(int B, int L, int M, int d, int flags) {
  if (B <= 0 || L <= 0 || M <= 0 || d <= 0 || d > MM_DMAX) return 0;
  const int Mp = mm_round_up_int(M, MM_M_ALIGN), P = mm_num_pairs(L, flags), Po = P - L;
  return ((size_t)B * P * (3 + d) * Mp + (size_t)B * Po * 2 * Mp) * sizeof(double);
}

// The sweeps on an already validated layout.  diag_only: the L diagonal pairs alone, out [B][L][3 + d][Mp] (the f32-model
// backward: its off-diagonal pairs are aggregated by mm_bwd_f32.hip; the diagonal operands rowD / colD / w64 / q64 and
// Zc64 are f64 in both pack types).  mu: [B][d] f64.
int mm_backward_sums_impl(const char* pk, const MMModelLayout& ml, const char* ws, const MMWorkspaceLayout& wl, int L, int M, int d,
                          int B, const double* mu, int flags, bool with_unc, bool diag_only, double* out, hipStream_t s) {
  const double* Cm = with_unc ? (const double*)(pk + ml.Cm) : nullptr;
  const int Pk = diag_only ? L : wl.P, Pok = diag_only ? 0 : wl.Po;       // what the kernels see as P / Po
  if (Pok > 0) { if (const int rj = mm_fork_join_wait(s)) return rj; }   // (the off-diagonal operands may still be on the q stage's side stream)
  double* out_col = out;
  double* out_row = out_col + (size_t)B * Pk * (3 + d) * wl.Mp;
  if (!(flags & MM_FORCE_GENERIC) && d <= 31) {
    const double* Zc = (const double*)(pk + ml.Zc64);
    const double* zb = (const double*)(pk + ml.zbar);
    const int ks4 = (d + 3) / 4, nu = (d + 16) / 16;          // (d + 1) <= 16 nu
    // diagonal pairs: k_bwd_diag (factored form); off-diagonal pairs of an f64 pack: column sums k_bwd_mfma<., ., false> from
    // pair L on, row sums <., ., true>.  MM_FORCE_WORST_TIER keeps the unfactored kernel for every pair (its cross-check)
    // ... and so do models of one or two column tiles (cartpole sizes: the sweeps are launch-bound there, and an f64 pack's diagonal and
    // off-diagonal pairs are ONE launch of the unfactored kernel against two)
    const bool old_diag = (flags & MM_FORCE_WORST_TIER) != 0 || (wl.Mp <= 128 && !diag_only);
    const int pc0 = old_diag ? 0 : L;
    const int nbd = nu == 1 ? MMB_DIAG_NB : 1, nbg = (B + nbd - 1) / nbd;       // batch elements per workgroup of k_bwd_diag
    const long long nwd = (long long)(wl.Mp / 64) * L * nbg, nwc = (long long)(wl.Mp / 64) * (Pk - pc0) * B,
                    nwr = (long long)(wl.Mp / 64) * Pok * B;
    if (nwd > 0x7fffffffLL || nwc > 0x7fffffffLL || nwr > 0x7fffffffLL) return MM_E_DIM;
    const int nw_diag = (int)nwd, nw_col = (int)nwc, nw_row = (int)nwr;
    const bool lowp = diag_only;                       // the f32 packs (their diagonal pairs alone run here): the forward's LOWP tiers
#define MMB_M_ARGS Zc, ml.Kz, zb, Cm, mu, L, wl.Mp, d, Pk, (const double*)(ws + wl.w64),                              \
                   (const double*)(ws + wl.q64), (const double*)(ws + wl.rowD), (const double*)(ws + wl.colD),       \
                   (const double*)(ws + wl.rowO), (const double*)(ws + wl.colO)
#define MMB_D_ARGS Zc, ml.Kz, zb, Cm, (const double*)(pk + ml.beta64), mu, L, M, wl.Mp, d, Pk, (const double*)(ws + wl.qhR),  \
                   (const double*)(ws + wl.qhC), (const double*)(ws + wl.colD), (const double*)(ws + wl.f1raw), out_col, B, nw_diag
#define MMB_D_LAUNCH(KS_, NU_)                                                                                      \
    do {                                                                                                            \
      if (Cm && lowp) hipLaunchKernelGGL((k_bwd_diag<KS_, NU_, true, true, (NU_ == 1 ? MMB_DIAG_NB : 1)>), dim3(nw_diag), dim3(256), 0, s, MMB_D_ARGS);    \
      else if (Cm) hipLaunchKernelGGL((k_bwd_diag<KS_, NU_, true, false, (NU_ == 1 ? MMB_DIAG_NB : 1)>), dim3(nw_diag), dim3(256), 0, s, MMB_D_ARGS);      \
      else if (lowp) hipLaunchKernelGGL((k_bwd_diag<KS_, NU_, false, true, (NU_ == 1 ? MMB_DIAG_NB : 1)>), dim3(nw_diag), dim3(256), 0, s, MMB_D_ARGS);    \
      else hipLaunchKernelGGL((k_bwd_diag<KS_, NU_, false, false, (NU_ == 1 ? MMB_DIAG_NB : 1)>), dim3(nw_diag), dim3(256), 0, s, MMB_D_ARGS);             \
    } while (0)
#define MMB_M_LAUNCH(KS_, NU_)                                                                                      \
    do {                                                                                                            \
      if (old_diag && Pok > 0 && (long long)nw_col + nw_row <= 1024) {                                              \
        hipLaunchKernelGGL((k_bwd_mfma_both<KS_, NU_>), dim3(nw_col + nw_row), dim3(256), 0, s, MMB_M_ARGS, out_col, out_row, B, \
                           nw_col, nw_row);                                                                         \
        break;                                                                                                      \
      }                                                                                                             \
      if (!old_diag) MMB_D_LAUNCH(KS_, NU_);                                                                        \
      if (nw_col > 0)                                                                                               \
        hipLaunchKernelGGL((k_bwd_mfma<KS_, NU_, false>), dim3(nw_col), dim3(256), 0, s, MMB_M_ARGS, out_col, B, nw_col, pc0); \
      if (Pok > 0)                                                                                                  \
        hipLaunchKernelGGL((k_bwd_mfma<KS_, NU_, true>), dim3(nw_row), dim3(256), 0, s, MMB_M_ARGS, out_row, B, nw_row, 0); \
    } while (0)
    if (ks4 <= 1) MMB_M_LAUNCH(1, 1);
    else if (ks4 == 2) MMB_M_LAUNCH(2, 1);
    else if (ks4 == 3) MMB_M_LAUNCH(3, 1);
    else if (ks4 == 4) { if (nu == 1) MMB_M_LAUNCH(4, 1); else MMB_M_LAUNCH(4, 2); }
    else if (ks4 <= 6) MMB_M_LAUNCH(6, 2);
    else MMB_M_LAUNCH(8, 2);
#undef MMB_M_LAUNCH
#undef MMB_D_LAUNCH
#undef MMB_D_ARGS
#undef MMB_M_ARGS
    hipError_t em = hipGetLastError();
    return em == hipSuccess ? 0 : (int)em;
  }
#define MMB_ARGS (const double*)(pk + ml.Z64), (const double*)(pk + ml.Zc64), ml.Kz, Cm, mu, L, M, wl.Mp, d, Pk,      \
                 (const double*)(ws + wl.w64), (const double*)(ws + wl.q64), (const double*)(ws + wl.rowD),                 \
                 (const double*)(ws + wl.colD), (const double*)(ws + wl.rowO), (const double*)(ws + wl.colO)
#define MMB_LAUNCH(DK_)                                                                                             \
  do {                                                                                                              \
    hipLaunchKernelGGL((k_bwd_sums<DK_, false>), dim3((wl.Mp + 255) / 256, Pk, B), dim3(256), 0, s, MMB_ARGS, out_col); \
    if (Pok > 0)                                                                                                    \
      hipLaunchKernelGGL((k_bwd_sums<DK_, true>), dim3((wl.Mp + 255) / 256, Pok, B), dim3(256), 0, s, MMB_ARGS, out_row); \
  } while (0)
  if (d <= 4) MMB_LAUNCH(4);
  else if (d <= 8) MMB_LAUNCH(8);
  else if (d <= 16) MMB_LAUNCH(16);
  else MMB_LAUNCH(32);
#undef MMB_LAUNCH
#undef MMB_ARGS
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// out: [B][P][3 + d][Mp] column sums followed by [B][Po][2][Mp] row sums (f64).
extern "C" int mm_backward_sums(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                                const void* mu, int flags, const void* workspace, size_t workspace_bytes,
                                void* out, size_t out_bytes, void* stream) {
  if (!packed || !mu || !workspace || !out) return MM_E_ARG;
  if (L <= 0 || M <= 0 || d <= 0 || B <= 0) return MM_E_ARG;
  if (d > MM_DMAX) return MM_E_DIM;
  if (dtype != MM_F64) return MM_E_DTYPE;              // f64 packs (f32 packs: mm_moment_match_backward)
  const MMModelLayout ml = mm_model_layout(L, M, d, dtype, 1);
  if (packed_bytes < ml.Cm) return MM_E_WORKSPACE;
  const bool has_C = packed_bytes >= ml.total;
  const bool with_unc = (flags & MM_MODEL_UNCERTAINTY) != 0;
  if (with_unc && !has_C) return MM_E_NO_C;
  const MMWorkspaceLayout wl = mm_workspace_layout(B, L, M, d, dtype, flags);
  if (workspace_bytes < wl.total) return MM_E_WORKSPACE;
  if (out_bytes < mm_backward_bytes(B, L, M, d, flags)) return MM_E_WORKSPACE;
  return mm_backward_sums_impl((const char*)packed, ml, (const char*)workspace, wl, L, M, d, B, (const double*)mu, flags, with_unc,
                               false, (double*)out, (hipStream_t)stream);
}
