// Shared host/device layout helpers for the moment-matching kernels (gfx950).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/gpflowpilco_mm.h"

static inline size_t mm_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int mm_round_up_int(int x, int a) { return (x + a - 1) / a * a; }
static inline size_t mm_elem_size(int dtype) { return dtype == MM_F64 ? 8 : 4; }

// Number of (a, a') kernel pairs that are reduced: all a <= a' with full output
// covariance (models.py:244-248), only a == a' otherwise (:249-252).
// internal stage bits of mm_launch_bwd_offdiag_f32 (mm_bwd_f32.hip), beside the public MM_STAGE_*: the re-reduce of the routed items
// can be split off the remainder sweep (it then runs on the side stream with the aggregate chain, mm_compose_bwd.hip)
#define MM_ISTAGE_NO_ROUTE (1 << 20)      /* with MM_STAGE_OFFDIAG: the sweep alone */
#define MM_ISTAGE_ROUTE (1 << 21)         /* the route decision + the f64 re-reduce of the routed items alone */
// q stage of a call that will NOT run the forward's off-diagonal reduce on this workspace (the backward's own q stage, the
// value-and-gradient call: their sums come from the backward's sweeps): the degree-5/6 moment chain (1.4 ms at C3) is skipped and
// s56 is poisoned with NaN, so that a forward reduce run on such a workspace by mistake fails loudly instead of reading stale sums
#define MM_ISTAGE_NO_M56 (1 << 22)

static inline int mm_num_pairs(int L, int flags) {
  return (flags & MM_FULL_OUTPUT_COV) ? L * (L + 1) / 2 : L;
}

// Moment table of the f32 mode: every monomial of the centred inducing inputs up to total degree
// mm_moment_deg(d), graded (degree 0, 1, 2, ...), colex order of the sorted index tuple inside a degree
// (rank of k1 <= ... <= kn is sum_t C(k_t + t - 1, t)).  Degree 4 for d <= 8: the off-diagonal sums are then
// collapsed to moments up to the quartic term (mm_moments.hip); degree 2 beyond (the table would have
// C(d + 4, 4) columns: 4845 at d = 16).
static inline int mm_moment_deg(int d) { return d <= 8 ? 4 : 2; }
static inline long long mm_binom(int n, int k) {
  if (k < 0 || k > n) return 0;
  long long r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return r;
}
// number of monomials of degree exactly n in d variables, and of degree < n (offset of the degree-n block)
static inline int mm_mono_count(int n, int d) { return (int)mm_binom(d + n - 1, n); }
static inline int mm_mono_offset(int n, int d) { int o = 0; for (int m = 0; m < n; ++m) o += mm_mono_count(m, d); return o; }
// entries of the rank table: sum_{k=1..deg} DK^k with DK = 8 (d <= 8) or 32
static inline size_t mm_rank_table_entries(int d) {
  const int dk = d <= 8 ? 8 : 32, deg = mm_moment_deg(d);
  size_t n = 0, p = 1;
  for (int k = 1; k <= deg; ++k) { p *= dk; n += p; }
  return n;
}
// columns of the table, rounded up to the 16-wide MFMA tile
static inline int mm_moment_cols(int d) { return mm_round_up_int(mm_mono_offset(mm_moment_deg(d) + 1, d), 16); }
// degree-5 and degree-6 monomials (d <= 8: the collapse's bf16 tables, mm_moments6.hip): degree 5 first, then degree 6, colex
// inside a degree; rounded up to the 128-column tile of the bf16 GEMM.  0 for d > 8 (no collapse).
// Layout: [degree 5 | pad to 128][degree 6 | pad to 128] -- a 128-column block of the GEMM holds one degree only, so that the
// degree-6 blocks can be formed for the rows that need them alone (MM_C6_X5_2 below).
// (the table also carries the degree-4 monomials, in front: [degree 4 | pad][degree 5 | pad][degree 6 | pad] -- the quartic term of
// p6 comes from the bf16 GEMM too; the names keep their "56")
static inline int mm_moment56_off5(int d) { return d <= 8 ? mm_round_up_int(mm_mono_count(4, d), 128) : 0; }
static inline int mm_moment56_off6(int d) { return d <= 8 ? mm_moment56_off5(d) + mm_round_up_int(mm_mono_count(5, d), 128) : 0; }
static inline int mm_moment56_cols(int d) { return d <= 8 ? mm_moment56_off6(d) + mm_round_up_int(mm_mono_count(6, d), 128) : 0; }
// Index tables of the degree-5/6 contraction (k_spoly56, mm_moments6.hip; functions of d alone, written at pack time), with
// sym(k) = mm_mono_count(k, d) sorted index tuples of length k in colex rank order:
//   ins  [m = 0..5][J < sym(m)][8] i16: rank inside sym(m + 1) of the tuple J with the index j inserted (0 for j >= d)
//   last [k = 0..5][I < sym(k)]    i16: the largest index of I (0 for k = 0): appending i >= last keeps I sorted, rank += C(i + k, k + 1)
//   mult2 [sym(2)], mult3 [sym(3)] f32: multinomial n! / prod(count!) of the tuple (a packed entry stands for that many tensor entries)
struct MMTab56 { int ins[6], last[6], n_i16, mult2, mult3, n_f32; size_t bytes; };
static inline MMTab56 mm_tab56(int d) {
  MMTab56 t;
  int o = 0;
  for (int m = 0; m < 6; ++m) { t.ins[m] = o; o += mm_mono_count(m, d) * 8; }
  for (int k = 0; k < 6; ++k) { t.last[k] = o; o += mm_mono_count(k, d); }
  t.n_i16 = mm_round_up_int(o, 8);
  t.mult2 = 0; t.mult3 = mm_mono_count(2, d); t.n_f32 = mm_mono_count(2, d) + mm_mono_count(3, d);
  t.bytes = (size_t)t.n_i16 * 2 + (size_t)t.n_f32 * 4;
  return t;
}

// Packed model: byte offsets inside the caller-owned device buffer.
struct MMModelLayout {
  int Mp, Kz, nd8, KMp;
  size_t Z64;     // [L][M][d]  f64 raw inducing inputs (prep stages run in f64)
  size_t Zt64;    // [L][d][Mp] f64 the same, dimension-major and zero padded: a wave reading one
                  //            dimension of 64 consecutive points is one 512-byte segment (k_qvec, k_pairvec)
  size_t zbar;    // [L][d]     f64 per-latent centroid of Z (centres the MFMA A operand)
  size_t ls2;     // [L][d]     f64 squared lengthscales (Lambda)
  size_t var;     // [L]        f64 kernel variances
  size_t meanc;   // [L]        f64 Constant mean (zeros for Zero)
  size_t beta64;  // [L][M]     f64 Kuu^-1 u
  size_t Zc64;    // [L][Mp][Kz] f64 centred inducing inputs, zero padded (rows >= M, cols >= d)
  size_t Zc;      // [L][Mp][Kz] T   same, element type T (aliases Zc64 when T is f64)
  size_t Zs3;     // [L][Mp/32][3][32][8 nd8] bf16: Zc split into bf16 parts (h, m, l), tile-and-part major, f32 mode only
  size_t Zm;      // [L][Mp][KMp] f64: the monomials of zc up to degree mm_moment_deg(d) per inducing point (graded colex,
                  // zero padded) -- the table the weight moments sum_m what_m zc_m^alpha are taken against (f32 mode only)
  size_t zmax2;   // [L] f64 max_m |zc_m|^2 (Cauchy-Schwarz bound on |b_ij| that admits a (b, pair) to the collapse)
  size_t rtab;    // int16: for k = 1..deg the rank (inside the degree-k block of the moment table) of every index tuple
                  // encoded base DK = 8 (d <= 8) or 32 in a flat position, -1 where a digit is >= d; blocks of DK^k
                  // entries one after another (k_spoly's lookups: no integer arithmetic per tensor entry)
  size_t Zq2;     // [L][Mp/32][2 (s)][2 (nb)][2 (hi, lo)][64 lanes][8] bf16 (f32 mode, d <= 8): the monomials of zc up to degree 2
                  // (slot 0: 1; 1..d: zc_l; then zc_l zc_l', l <= l', row-major; 64 slots, zero beyond) as the B operand of
                  // v_mfma_f32_32x32x16_bf16 -- lane (slot & 31, h) of image (tile, s, nb = slot >> 5, part) holds K slots
                  // t = 0..7 = columns 16 s + 8 (t >> 2) + 4 h + (t & 3) of the tile (mm_bwd_f32.hip); 8 KB per 32 columns
  size_t Zm56;    // [L][2 (h, m)][N56p][Mp] bf16 (f32 mode, d <= 8): the degree-5 and degree-6 monomials of zc, 2-way split, one
                  // monomial's Mp values contiguous (the B operand of k_wmom56_gemm reads 8 consecutive m per lane)
  size_t tab56;   // index tables of k_spoly56 (MMTab56; f32 mode, d <= 8): i16 block, then the f32 multinomials
  size_t perm;    // [L][Mp] int32: the caller's index of the inducing point at packed position m (identity beyond M and for packs
                  // of M <= MM_SORT_MIN_M points).  Every packed array is in PACKED order: per latent the points are sorted by
                  // |(z - zbar) / lengthscale| (mm_kernels.hip: k_pack_key / k_pack_rank) -- the outputs are sums over the points
                  // and do not depend on it, but |b_ij| <= |G| |zeta_i| |zeta_j|, so tiles of sorted points are homogeneous and the
                  // reduce kernels' per-tile range tiers (and the f32 sweep's screening) see smaller maxima; q_out is written
                  // through perm in the caller's order
  size_t skey;    // [L][Mp] f64 scratch of the pack: the sort keys
  size_t zt2;     // [L][Mp/32] f32: max |zc_m|^2 over the 32 points of a column tile, rounded up (with the pack in norm order nearly
                  // ascending): the f32 sweep skips the column tiles whose Cauchy-Schwarz bound with the wave's own rows is inside
                  // the collapsed range without touching them (mm_mfma.hip)
  size_t Cm;      // [L][Mp][Mp] f64 Kuu^-1 S Kuu^-1 - Kuu^-1, zero padded (absent: == total).
                  // Always f64: with Kuu jitter 1e-6 its norm reaches 1e6.
  size_t total;
};
// packs of at most this many inducing points keep the caller's order (four 64-point tiles: nothing to gain, and the policy
// packs of the composed rollout -- M <= 256 -- return gradients per packed centre: include/gpflowpilco_mm.h g_policy)
#define MM_SORT_MIN_M 256

static inline MMModelLayout mm_model_layout(int L, int M, int d, int dtype, int with_C) {
  MMModelLayout o;
  const size_t es = mm_elem_size(dtype), A = 256;
  o.Mp = mm_round_up_int(M, MM_M_ALIGN);
  o.Kz = mm_round_up_int(d, 2);
  o.nd8 = (d + 7) / 8;
  o.KMp = mm_moment_cols(d);
  size_t off = 0;
  o.Z64 = off;    off = mm_align_up(off + (size_t)L * M * d * 8, A);
  o.Zt64 = off;   off = mm_align_up(off + (size_t)L * d * o.Mp * 8, A);
  o.zbar = off;   off = mm_align_up(off + (size_t)L * d * 8, A);
  o.ls2 = off;    off = mm_align_up(off + (size_t)L * d * 8, A);
  o.var = off;    off = mm_align_up(off + (size_t)L * 8, A);
  o.meanc = off;  off = mm_align_up(off + (size_t)L * 8, A);
  o.beta64 = off; off = mm_align_up(off + (size_t)L * M * 8, A);
  o.Zc64 = off;   off = mm_align_up(off + (size_t)L * o.Mp * o.Kz * 8, A);
  o.Zc = o.Zc64;
  if (dtype != MM_F64) { o.Zc = off; off = mm_align_up(off + (size_t)L * o.Mp * o.Kz * es, A); }
  o.Zs3 = off;
  if (dtype != MM_F64) off = mm_align_up(off + (size_t)L * o.Mp * 24 * o.nd8 * 2, A);
  o.Zm = off;
  if (dtype != MM_F64) off = mm_align_up(off + (size_t)L * o.Mp * o.KMp * 8, A);
  o.zmax2 = off;  off = mm_align_up(off + (size_t)L * 8, A);
  o.rtab = off;   off = mm_align_up(off + mm_rank_table_entries(d) * 2, A);
  o.Zq2 = off;
  if (dtype != MM_F64 && d <= 8) off = mm_align_up(off + (size_t)L * (o.Mp / 32) * 8192, A);
  o.Zm56 = off;
  if (dtype != MM_F64 && d <= 8) off = mm_align_up(off + (size_t)L * 2 * mm_moment56_cols(d) * o.Mp * 2, A);
  o.tab56 = off;
  if (dtype != MM_F64 && d <= 8) off = mm_align_up(off + mm_tab56(d).bytes, A);
  o.perm = off;   off = mm_align_up(off + (size_t)L * o.Mp * 4, A);
  o.skey = off;   off = mm_align_up(off + (size_t)L * o.Mp * 8, A);
  o.zt2 = off;    off = mm_align_up(off + (size_t)L * (o.Mp / 32) * 4, A);
  o.Cm = off;
  if (with_C) off = mm_align_up(off + (size_t)L * o.Mp * o.Mp * 8, A);
  o.total = off;
  return o;
}

// k_wmom_gemm: slices of the m range (split-K) per output tile; k_spoly adds the partial moment vectors
#define MM_MOM_SPLIT 2
// THE MOMENT COLLAPSE (d <= 8; round 5: to degree 6).  For a collapsed (b, pair) the polynomial
//     p6(x) = x^3 (C0 + C1 x + C2 x^2 + C3 x^3)  ~  r(x) = e^x - 1 - x - x^2/2     on |x| <= MM_C6_MAX = 1/4
// (near-minimax, |p6 - r| <= 5.8e-10 there: it IS the tile kernel's degree-3 tier, MMRem<3>) is taken from weight moments
// against model-constant monomial tables -- degrees <= 3 in f64 (k_wmom_gemm, k_spoly), degrees 4, 5 and 6 from a bf16 2-way
// split GEMM on the matrix pipe with f32 accumulation and a partially-symmetric f32 contraction (mm_moments6.hip) -- and the
// tile kernel skips every wave tile whose max|b| <= 1/4 after a one-MFMA screening product, reducing the correction r - p6 on
// the others.  Round 4 stopped at degree 4 (first tier |b| <= 1/20): on the BASELINE recipe 9 % of the wave tiles could be
// skipped and 18 % of the items were worth collapsing; with rows recentred (mm_mono.h) and degree 6, 92 % of the tiles have
// max|b| <= 1/4 and 53 % of the items are WHOLLY inside by their Cauchy-Schwarz bound alone (tools/tile_hist_baseline.py).
// Accuracy of what the skipped tiles leave out (tools/collapse6_study.py, BASELINE recipe at C3, 56 items): <= 1.7e-6 of the
// batch element's largest off-diagonal covariance (rms 3.7e-7; systematic, cancels the way the sum itself does), rounding of
// the degree-5/6 moments from the 2-way split <= 1.1e-8.
// An item is collapsed when max_i |A_i|^2 * max_j |zc_j|^2 <= MM_COLLAPSE_BOUND2 (Cauchy-Schwarz bound on |b_ij|): (1/2)^2 --
// items with a bound in (1/4, 1/2] have 98 % of their tiles under 1/4, items beyond 1/2 a third.
#ifndef MM_COLLAPSE_BOUND2
#define MM_COLLAPSE_BOUND2 0.25f
#endif
// ... and not every collapsed item needs every order: with the item's Cauchy-Schwarz bound X on |b|, dropping the degree-6 term
// of p6 leaves C3 X^6 <= 8.3e-11 at X = 1/16, dropping degrees 5 and 6 C2 X^5 + C3 X^6 <= 8.2e-11 at X = 1/40 -- below the
// polynomial's own 5.8e-10.  k_wmom_perm orders a latent's rows [degree 6 | degree 5 only | degree 4 only | not collapsed], the
// bf16 GEMM forms the degree-5 column blocks for the first two classes and the degree-6 blocks for the first, k_spoly56 contracts
// what exists.  (The pilco recipe's items sit at X <= 0.05: without this the degree-5/6 work -- all of it useless there -- took
// the step from 5.0 to 6.2 ms.)
#define MM_C6_X5_2 (1.0f / 256.0f)      /* X <= 1/16: no degree-6 term */
#define MM_C6_X4_2 (1.0f / 1600.0f)     /* X <= 1/40: neither degree 5 nor 6 (the f64 cubic and the bf16 quartic moments only).  (At 1/32 the dropped
                                           C2 X^5, priced like p6's own error in the route estimate, sent 0.6 % of the pilco recipe's
                                           items -- |what| |what'| ~ 1e7 x the covariance scale there -- to the f64 re-reduce: +0.5 ms) */
#define MM_C6_MAX 0.25f
#define MM_C6_C0 1.666663289e-01f
#define MM_C6_C1 4.166659713e-02f
#define MM_C6_C2 8.350561373e-03f
#define MM_C6_C3 1.391559141e-03f
// what a skipped entry leaves out, as an equivalent relative rounding for the route estimate (mm_route.hip): the independence
// model with a per-entry amplitude of 1e-10 (the polynomial's error oscillates with amplitude 5.8e-10, but it is a smooth
// function of b and cancels under the alternating weights: measured total / (|what|_2 |what'|_2) <= 4e-11,
// tools/collapse6_study.py).  estS = MM_C6_SYS2 * min(1, (4 X)^3)^2 * sum what_i^2 * sum what'_j^2 (X = the item's
// Cauchy-Schwarz bound: near 0 the error of p6 is the perturbation of its leading coefficient, 3.4e-7 |x|^3) is added to the sweep's E2:
// (1e-10 / (2^-24 * 2/3))^2
#define MM_C6_SYS2 6.33e-6f
// p6 as every f64 consumer evaluates it (k_spoly's coefficients, the routed re-reduce, the portable kernel): the f32 literals
// widened, so that the moments and the tile kernel's corrected tiers subtract the same polynomial
#define MM_C6_POLY_F64(x_) (((x_) * (x_)) * (x_) * fma(fma(fma((double)MM_C6_C3, (x_), (double)MM_C6_C2), (x_), (double)MM_C6_C1), (x_), (double)MM_C6_C0))
// First tier of the f32 remainder: near-minimax r(x) = expm1(x) - x - x^2/2 ~ x^3 (C0 + C1 x) on |x| <= MM_TIER1_MAX
// (tools/minimax_remainder.py).  It is the polynomial the moment collapse takes from the f64 moments, so its
// approximation error is SYSTEMATIC (it does not average out over the M^2 entries the way rounding does) and sets the
// f32 mode's error floor at wide states: range 1/16 -> 2.2e-8 |x|, 1/20 -> 9.0e-9, 1/24 -> 4.3e-9, 1/32 -> 1.4e-9.
// Measured at C3 (B = 256; wide-state 10-step rollout f32 vs f64 mode | BASELINE-recipe step | pilco step):
//   1/16: 2.9e-6 | 11.12 ms | 4.72 ms     1/20: 1.2e-6 | 11.35 | 4.84     1/24: 5.3e-7 | 11.45 | 5.17     1/32: 1.5e-7 | 11.97 | 5.95
// 1/20 restores the 2e-6 bound of tests/test_gpu_fullsize.py (it had been widened to 1e-5 at 1/16) for +2 % / +2.6 %.
#ifndef MM_TIER1_DIV
#define MM_TIER1_DIV 20
#endif
#if MM_TIER1_DIV == 16
#define MM_TIER1_MAX 0.0625f
#define MM_REM1_C0 1.666936278e-01f
#define MM_REM1_C1 4.167173430e-02f
#elif MM_TIER1_DIV == 20
#define MM_TIER1_MAX 0.05f
#define MM_REM1_C0 1.666839272e-01f
#define MM_REM1_C1 4.166988656e-02f
#elif MM_TIER1_DIV == 24
#define MM_TIER1_MAX 0.041666668f
#define MM_REM1_C0 1.666786522e-01f
#define MM_REM1_C1 4.166889191e-02f
#elif MM_TIER1_DIV == 32
#define MM_TIER1_MAX 0.03125f
#define MM_REM1_C0 1.666734070e-01f
#define MM_REM1_C1 4.166791216e-02f
#else
#error "MM_TIER1_DIV must be 16, 20, 24 or 32"
#endif
// ... and when the bound itself says every |b_ij| <= MM_C6_MAX, the whole remainder of the (b, pair) is inside the
// collapsed range: the tile kernel's workgroup writes zero partials and leaves
#define MM_INSIDE_BOUND2 (0.998f * MM_C6_MAX * MM_C6_MAX)
// Accuracy contract of the f32 off-diagonal reduce.  The f32 tile kernels carry, beside their sums, an
// estimate of the rounding error of each (b, pair)'s remainder sum under an independent-rounding model,
//     est = 2^-24 * (2/3) * sqrt( sum over lane blocks of (sum_rows what_i^2) what'_j^2 (max|b|^3 (1 + X + X^2))^2 )
// (rho(x) = |r(x)| + |x| |r'(x)| <= (2/3) |x|^3 e^|x| bounds what a relative error 2^-24 in what_i, what'_j, b_ij does to
// what_i what'_j r(b_ij)); measured 7-50 x above the actual error (tools/route_study.py).  An item whose estimate exceeds
// MM_ROUTE_TOL x (the largest |off-diagonal covariance| of its batch element, taken from the f64 moments: s12 - f1 f1') is
// re-reduced in f64 (mm_route.hip) and its slab overwritten: the ROUNDING error of what stays in f32 is then within ~4e-5 of
// the block's own scale (est / 7).  3e-4: on the BASELINE recipe at C3 (7168 items per step) 4-7 items per step exceed it
// (65 at 1e-4, none at 1e-3), every item of the ill-conditioned wide draws does (their est / scale is 5e-3 .. 0.2).
#ifndef MM_ROUTE_TOL
#define MM_ROUTE_TOL 3.0e-4
#endif
// the f64 re-reduce of the forward splits an item's columns over this many work units per row panel (latency of a routed
// item: 256 rows x M / MM_ROUTE_CSPLIT columns of f64 VALU work per workgroup); their partial sums take slots
// [0, npanel * ncc) of the item's slab, ncc = min(MM_ROUTE_CSPLIT, NS / npanel)
#define MM_ROUTE_CSPLIT 8
static inline int mm_route_ncc(int NS, int npanel) { int n = NS / npanel; return n < 1 ? 1 : (n > MM_ROUTE_CSPLIT ? MM_ROUTE_CSPLIT : n); }
// Exponent caps of the reduce kernels.  Every entry Q_ij = q_i q_j e^{delta_ij} is bounded by var_a var_a', but its factors are
// not: with lengthscales far below the state's distance to an inducing point (the reference's tests draw them down to 0.01) the
// bilinear part b_ij reaches several hundred while q_i q_j is e^{-thousands}.  Whenever b_ij is that large the entry itself is
// negligible (b_ij <= (zeta_i^T G zeta_i + zeta_j^T G zeta_j) / 2 and the weights carry e^{-zeta^T P zeta / 2} with P >= G), so the
// exponents are capped below the overflow of their type: a zero weight then multiplies a finite number instead of inf.
#define MM_EXP_CAP_F64 700.0
#define MM_EXP_CAP_F32 80.0f
// Rows per workgroup of the generic reduce kernel / columns per workgroup.
#define MM_GEN_ROWS 64
#define MM_GEN_COLS 256
// f32 MFMA kernel: row panel per workgroup (4 waves x 64 rows).
#define MM_PANEL_ROWS 256
// f64 MFMA kernel: 64 x 64 tiles
#define MM_F64_TILE 64

// Diagonal pairs (a == a', p < L) are always reduced in f64 (they carry the C-weighted
// term, whose conditioning rules out f32); off-diagonal pairs (p >= L) in T.
struct MMWorkspaceLayout {
  int Mp, P, Po, NS;
  size_t pairmat;  // [B][P][d^2 + 1] f64: G, const
  size_t latmat;   // [B][L][2 d^2 + 2] f64: P_a = (Sigma + Lambda_a)^-1, log-normaliser, log det(Sigma + Lambda_a),
                   //                   E_a = sym(Lambda_a^-1 Sigma P_a)
  size_t rho1;     // [B][L][Mp] f64  zeta_i^T E_a zeta_i (the pair-independent part of rho_i / gamma_j)
  size_t w64;      // [B][L][Mp] f64  beta_i q_i
  size_t q64;      // [B][L][Mp] f64  q_i = <k_a(x, z_i)>
  size_t w;        // = w64 (the f64 mode's generic kernel reads it; an f32 copy used to be written and was read by nothing)
  size_t lq;       // [B][L][Mp] f64  log q_i (-1e30 in the padding): k_pairvec forms every factored weight as ONE exponential
                   //                 beta_i exp(log q_i + rho_i) -- with short lengthscales q_i underflows to 0 where e^{rho_i}
                   //                 overflows (the reference's own test designs: lengthscales down to 0.01), and 0 x inf is a NaN
  size_t rowD;     // [B][L][Mp] f64        rho_i          diagonal pairs
  size_t colD;     // [B][L][d+1][Mp] f64   g_j, gamma'_j  diagonal pairs
  size_t qhR;      // [B][L][Mp] f64  diagonal pairs, factored weights of the f64 MFMA reduce: u_i e^{rho_i}, u = q with
  size_t qhC;      // [B][L][Mp] f64  model uncertainty (w without): column side u_j e^{gamma'_j}; then delta = zc_i . g_j only
  size_t rowO;     // off-diagonal pairs.  f64: [B][Po][Mp] rho_i            f32: [B][Po][d+1][Mp] A_i, what_i
  size_t colO;     //                      f64: [B][Po][d+1][Mp] g_j, gamma'_j  f32: [B][Po][Mp] what'_j
  size_t f1raw;    // [B][L] f64      sum_i w_i (f1 without the mean)
  size_t whR;      // [B][Po][Mp] f64  what_i  = w_i e^{rho'_i}   (f32 mode: row weights, unrounded)
  size_t whC;      // [B][Po][Mp] f64  what'_j = w'_j e^{gamma_j}
  size_t mom;      // [B][Po][2][MM_MOM_SPLIT][KMp] f64  partial sums over m of what_m zc_m^alpha (all monomials of the
                   //                  table): row side, column side; MM_MOM_SPLIT slices of the m range
  size_t amax;     // [B][Po] u32  bits of max_i |A_i|^2 (f32, >= 0: ordered like the integer), zeroed by k_prep
  size_t amaxc;    // [B][Po] u32  the same over the rows of the COLLAPSED row groups alone (0: none), zeroed by k_prep (== amax + 4 B Po)
  size_t gflag;    // [B][Po][Mp/64] u8 (f32 mode, d <= 8): 1 = the 64-row group is collapsed -- orders 4, 5, 6 of its rows' p6 come from
                   //              the f32 moments (k_pairvec_reg writes zero bf16 row weights for the other groups), the sweep screens /
                   //              corrects its tiles; 0 = the sweep reduces r(b) on every tile of the group -- minus the cubic term
                   //              C0 b^3 where the item has collapsed groups at all: the f64 moments then carry that term for EVERY
                   //              row (mm_mono.h: MM_GROUP_ROWS)
  size_t gperm;    // [L][(L-1) B] i32 + [3][L] i32 (f32 mode): per latent the rows of its moment GEMM ordered [collapsed to degree 6 |
                   //              to degree 5 | to degree 4 | not collapsed], and the counts {collapsed, needing degree 5, needing
                   //              degree 6} -- every column block of the GEMMs is formed for the rows that read it only
  size_t s12;      // [B][Po] f64  the polynomial part of the off-diagonal sums from the moments:
                   //              orders 0..2 always, order 3 as well where the (b, pair) is collapsed (orders 4..6: s56)
  size_t gmax2;    // [B][Po][Mp/64] f32 (f32 mode, d <= 8): max_i |A_i|^2 over the rows of the group, rounded up (k_pairvec_reg): the
                   //              sweep's Cauchy-Schwarz skipping reads it instead of the rows (a collapsed group all of whose
                   //              tiles are inside the collapsed range loads nothing at all)
  size_t partB;    // [B][P][NS] f64 partial sums of w_i expm1(delta_ij) w_j
  size_t partC;    // [B][L][NS] f64 partial sums of C_ij q_i expm1(delta_ij) q_j  (+ q^T C q)
  size_t mu64;     // [B][d] f64    the state mean as the q stage read it (mm_route.hip re-derives A_i = G^T (z_i - mu) in f64)
  size_t estO;     // [B][Po][ceil(Mp/256)] f32 (f32 mode): per row panel the tile kernel's running estimate of its own rounding error,
                   //              sum over lane blocks of (sum_rows what_i^2) what'_j^2 (max|b|^3)^2 (1 + X + X^2)^2 (mm_route.hip)
  size_t rlist;    // [B Po] i32   the (b, off-diagonal pair) items k_route_decide hands to the f64 re-reduce
  size_t rcount;   // [4] i32      {entries of rlist (current pass), items routed by the last forward, by the last backward, 0}
  size_t rflag;    // [B Po] i32   1 where the last forward routed the item (k_finalize then sums the route kernel's slots)
  size_t wsp;      // [B][Po][2 (row, column side)][2 (h, m)][Mp] bf16 (f32 mode, d <= 8): the factored weights, 2-way split, as the A
                   //              operand of the degree-5/6 moment GEMM (k_pairvec_reg writes them beside whR / whC)
  size_t mom56;    // [B][Po][2][N56p] f32: sum_m what_m zc_m^alpha over the degree-5 and degree-6 monomials (collapsed items)
  size_t estS;     // [B][Po] f32: MM_C6_SYS2 sum what^2 sum what'^2 of a collapsed item (0 otherwise): what its skipped tiles leave
                   //              out, in the units of the sweep's error estimate estO (k_spoly56 writes, k_route_decide adds)
  size_t ilist;    // i32 [4] counts {items collapsed to degree 6, to degree 5, to degree 4, 0} + [B Po] item indices: the degree-6 and
                   //              degree-5 items from the front (in that order), the degree-4 items from the back (k_item_classes:
                   //              the work lists of k_spoly56 / k_spoly4, mm_moments6.hip)
  size_t s56;      // [B][Po] f64: C1 <N_4, G^4 Q_4> + C2 <N_5, G^5 Q_5> + C3 <N_6, G^6 Q_6> of a collapsed item (0 otherwise): the degree-4..6
                   //              part of p6 from the f32 moments.  Kept apart from s12: an item the accuracy contract re-reduces in f64 (mm_route.hip) takes
                   //              those two orders from the re-reduce instead (k_finalize skips s56 where rflag is set)
  size_t f1s;      // [B][L] T      rollout scratch outputs
  size_t Sffs;     // [B][L][L] T
  size_t crs;      // [B][d][L] T
  size_t total;
};

static inline MMWorkspaceLayout mm_workspace_layout(int B, int L, int M, int d, int dtype, int flags) {
  MMWorkspaceLayout o;
  const size_t es = mm_elem_size(dtype), A = 256;
  o.Mp = mm_round_up_int(M, MM_M_ALIGN);
  o.P = mm_num_pairs(L, flags);
  o.Po = o.P - L;
  // partial-sum slots per (b, pair): the largest of what the reduce kernels write
  const int nrb = (o.Mp + MM_GEN_ROWS - 1) / MM_GEN_ROWS;
  const int ncb = (o.Mp + MM_GEN_COLS - 1) / MM_GEN_COLS;
  const int nt = o.Mp / MM_F64_TILE;
  int ns = nrb * ncb;                                          // generic kernel
  if (nt * (nt + 1) / 2 > ns) ns = nt * (nt + 1) / 2;          // f64 MFMA kernel, diagonal pairs
  if (dtype == MM_F64 && o.Po > 0 && nt * nt > ns) ns = nt * nt;  // f64 MFMA kernel, off-diagonal
  o.NS = ns;
  size_t off = 0;
  o.pairmat = off; off = mm_align_up(off + (size_t)B * o.P * (d * d + 1) * 8, A);
  o.latmat = off;  off = mm_align_up(off + (size_t)B * L * (2 * d * d + 2) * 8, A);
  o.rho1 = off;    off = mm_align_up(off + (size_t)B * L * o.Mp * 8, A);
  o.w64 = off;     off = mm_align_up(off + (size_t)B * L * o.Mp * 8, A);
  o.q64 = off;     off = mm_align_up(off + (size_t)B * L * o.Mp * 8, A);
  o.w = o.w64;
  o.lq = off;      off = mm_align_up(off + (size_t)B * L * o.Mp * 8, A);
  o.rowD = off;    off = mm_align_up(off + (size_t)B * L * o.Mp * 8, A);
  o.colD = off;    off = mm_align_up(off + (size_t)B * L * (d + 1) * o.Mp * 8, A);
  o.qhR = off;     off = mm_align_up(off + (size_t)B * L * o.Mp * 8, A);
  o.qhC = off;     off = mm_align_up(off + (size_t)B * L * o.Mp * 8, A);
  const size_t nrow = dtype == MM_F64 ? 1 : (size_t)(d + 1), ncol = dtype == MM_F64 ? (size_t)(d + 1) : 1;
  o.rowO = off;    off = mm_align_up(off + (size_t)B * o.Po * nrow * o.Mp * es, A);
  o.colO = off;    off = mm_align_up(off + (size_t)B * o.Po * ncol * o.Mp * es, A);
  o.f1raw = off;   off = mm_align_up(off + (size_t)B * L * 8, A);
  const size_t nwh = dtype == MM_F64 ? 0 : (size_t)B * o.Po * o.Mp;
  o.whR = off;     off = mm_align_up(off + nwh * 8, A);
  o.whC = off;     off = mm_align_up(off + nwh * 8, A);
  o.mom = off;     off = mm_align_up(off + (dtype == MM_F64 ? 0 : (size_t)B * o.Po * 2 * MM_MOM_SPLIT * mm_moment_cols(d) * 8), A);
  o.amax = off;    off = off + (size_t)B * o.Po * 4;
  o.amaxc = off;   off = mm_align_up(off + (size_t)B * o.Po * 4, A);
  o.gflag = off;   off = mm_align_up(off + (dtype == MM_F64 ? 0 : (size_t)B * o.Po * (o.Mp / 64)), A);
  o.gmax2 = off;   off = mm_align_up(off + (dtype == MM_F64 ? 0 : (size_t)B * o.Po * (o.Mp / 64) * 4), A);
  o.gperm = off;   off = mm_align_up(off + (dtype == MM_F64 || L < 2 ? 0 : ((size_t)L * (L - 1) * B + 3 * L) * 4), A);
  o.s12 = off;     off = mm_align_up(off + (size_t)B * o.Po * 8, A);
  o.partB = off;   off = mm_align_up(off + (size_t)B * o.P * o.NS * 8, A);
  o.partC = off;   off = mm_align_up(off + (size_t)B * L * o.NS * 8, A);
  const size_t nro = dtype == MM_F64 ? 0 : (size_t)B * o.Po;
  o.mu64 = off;    off = mm_align_up(off + (size_t)B * d * 8, A);
  o.estO = off;    off = mm_align_up(off + nro * ((o.Mp + MM_PANEL_ROWS - 1) / MM_PANEL_ROWS) * 4, A);
  o.rlist = off;   off = mm_align_up(off + nro * 4, A);
  o.rcount = off;  off = mm_align_up(off + 16, A);
  o.rflag = off;   off = mm_align_up(off + nro * 4, A);
  const size_t n56 = (dtype == MM_F64 || d > 8) ? 0 : (size_t)B * o.Po;
  o.wsp = off;     off = mm_align_up(off + n56 * 4 * o.Mp * 2, A);
  o.mom56 = off;   off = mm_align_up(off + n56 * 2 * mm_moment56_cols(d) * 4, A);
  o.estS = off;    off = mm_align_up(off + n56 * 4, A);
  o.s56 = off;     off = mm_align_up(off + n56 * 8, A);
  o.ilist = off;   off = mm_align_up(off + n56 * 4 + 16, A);
  o.f1s = off;     off = mm_align_up(off + (size_t)B * L * es, A);
  o.Sffs = off;    off = mm_align_up(off + (size_t)B * L * L * es, A);
  o.crs = off;     off = mm_align_up(off + (size_t)B * d * L * es, A);
  o.total = off;
  return o;
}
