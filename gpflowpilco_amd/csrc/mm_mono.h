// Monomial indexing of the moment tables (host + device): graded order, colex rank of the sorted index tuple
// inside a degree.  See mm_common.h (mm_moment_deg / mm_mono_offset) and mm_moments.hip.
#pragma once
#include <hip/hip_runtime.h>

// C(n, k) for the small arguments used here (n <= 40, k <= 5)
__host__ __device__ __forceinline__ int mm_binom_i(int n, int k) {
  if (k < 0 || k > n) return 0;
  int r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return r;
}

// offset of the degree-n block: number of monomials of degree < n in d variables
__host__ __device__ __forceinline__ int mm_mono_off(int n, int d) {
  int o = 0;
  for (int m = 0; m < n; ++m) o += mm_binom_i(d + m - 1, m);
  return o;
}

// rank of the multiset {k[0] <= k[1] <= ... <= k[n-1]} inside the degree-n block
__host__ __device__ __forceinline__ int mm_mono_rank(const int* k, int n) {
  int r = 0;
  for (int t = 0; t < n; ++t) r += mm_binom_i(k[t] + t, t + 1);
  return r;
}

// rank of an UNSORTED index tuple of length n <= 4 (sorted with a tiny network first)
__host__ __device__ __forceinline__ int mm_mono_rank_unsorted(int k0, int k1, int k2, int k3, int n) {
  // unused slots must be passed as values that keep the order (the caller passes n)
  int k[4] = {k0, k1, k2, k3};
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && k[j - 1] > k[j]; --j) { const int t = k[j]; k[j] = k[j - 1]; k[j - 1] = t; }
  return mm_mono_rank(k, n);
}

// inverse of mm_mono_rank: k[0..n-1] sorted ascending
__host__ __device__ __forceinline__ void mm_mono_unrank(int r, int n, int* k) {
  for (int t = n - 1; t >= 0; --t) {
    int v = 0;
    while (mm_binom_i(v + 1 + t, t + 1) <= r) ++v;
    k[t] = v;
    r -= mm_binom_i(v + t, t + 1);
  }
}

// Row centring of the f32 off-diagonal operands (round 5).  With zeta_i = z_i - mu the exponent's bilinear part is
//     zeta_i^T G zc'_j = zc_i^T G zc'_j - dmu^T G zc'_j,     dmu = mu - zbar_a,
// and the second term depends on the column alone: k_pairvec folds e^{-dmu^T G zc'_j} into the column weight and hands the tile
// kernels A_i = G^T zc_i, ROWS CENTRED AT THE LATENT'S CENTROID.  |zc_i| is smaller than |z_i - mu| wherever the state is not at
// the centroid (BASELINE recipe, mu ~ U[0,1]^d: wave tiles with max|b| <= 1/4 go from 84 % to 92 %, items whose Cauchy-Schwarz
// bound is <= 1/4 from 40 % to 53 %), and the weight moments need no binomial shift.  The fold is only taken where its exponent
// is harmless: |dmu^T G zc'_j| <= |G^T dmu| max_j |zc'_j| <= MM_RECENTRE_CMAX (with lengthscales far below the state's distance
// to the centroid it reaches hundreds and the weight would leave the f32 range); an item that keeps its rows centred at mu
// is marked by amax = FLT_MAX, which also keeps it out of every collapse predicate (they all go through mm_collapse_bound2).
#define MM_RECENTRE_CMAX 16.0
#define MM_AMAX_NOT_RECENTRED 0x7f7fffffu
__host__ __device__ __forceinline__ bool mm_rows_recentred(unsigned int amax_bits) { return amax_bits != MM_AMAX_NOT_RECENTRED; }

// Cauchy-Schwarz bound (squared) on |b_ij| of one (b, off-diagonal pair): max_i |A_i|^2 (k_pairvec, f32 bits, rounded
// up) times max_j |zc'_j|^2 (pack time).  ONE definition: k_spoly, the f32 tile kernel and the portable kernel must
// agree on which (b, pair) is collapsed.
__device__ __forceinline__ float mm_collapse_bound2(unsigned int amax_bits, double zmax2) {
  return __uint_as_float(amax_bits) * ((float)zmax2 * 1.000001f);
}
// ROW-GROUP COLLAPSE.  The predicate is taken per group of MM_GROUP_ROWS = 64 consecutive rows (one wave of k_pairvec_reg, one
// wave panel of the tile kernel): a group is collapsed when the bound of ITS rows, max_{i in group} |A_i|^2 max_j |zc'_j|^2, is
// <= MM_COLLAPSE_BOUND2.  With the pack in norm order (MMModelLayout::perm) the rows of large |A_i| are the last groups, and an
// item whose overall bound is beyond 1/2 -- "dense" before: 11 % of the BASELINE recipe's items, 70 % of the sweep's time --
// keeps 62 % of its rows collapsed.  gflag (workspace) is the ONE record of the decision; amaxc = the max of |A_i|^2 over the
// collapsed groups (0: no collapsed group) gives the item's collapse degree and screening margin through mm_collapse_bound2.
// Who carries what for an item with collapsed groups: the CUBIC term C0 b^3 of every row, collapsed or not, is in the f64
// moments (k_wmom_gemm, k_spoly: sum_ij what_i what'_j b_ij^3 = <N_3, G^(x)3 Q_3> is an identity, exact for any b; the f64
// GEMM needs no second weight set); orders 4, 5, 6 of p6 for the rows of the collapsed groups alone (f32 moments from the bf16
// GEMM, whose row weights k_pairvec_reg zeroes for the other groups); the sweep reduces r - p6 on the collapsed groups' tiles
// that are not skipped and r - C0 b^3 on every tile of the others; the routed f64 re-reduce r - C0 b^3 on every row.
#define MM_GROUP_ROWS 64
__host__ __device__ __forceinline__ bool mm_item_collapsed(unsigned int amaxc_bits) { return amaxc_bits != 0u; }
