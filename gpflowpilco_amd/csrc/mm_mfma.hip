// f32 MFMA fused reduce of the off-diagonal kernel pairs (a < a') on gfx950.
//
// For every (b, pair) the reference's eKuffu slice Q_aa' (utils/kernel_expectation.py:72-187, then
// models.py:219-248) is never stored; with b_ij = A_i . zc_j, what_i = w_i e^{rho'_i}, what'_j = w'_j e^{gamma_j}
//     S = sum_ij w_i expm1(delta_ij) w'_j,   delta_ij = rho'_i + gamma_j + b_ij
//       = sum_ij what_i what'_j (1 + b_ij + b_ij^2/2)  - (sum w)(sum w')        exact, f64 moments (O(M d^2))
//       + sum_ij what_i what'_j r(b_ij),   r(x) = expm1(x) - x - x^2/2            THIS kernel, f32, O(M^2)
// Where the model's input dimension allows the moment tables (d <= 8) and a (b, pair)'s Cauchy-Schwarz bound on |b_ij| is
// <= 1/2 (MM_COLLAPSE_BOUND2: nearly all of its tiles are then inside 1/4), the (b, pair) is COLLAPSED (mm_common.h,
// mm_moments.hip, mm_moments6.hip): p6(x) = x^3 (C0 + .. + C3 x^3) -- this kernel's own degree-3 tier, |p6 - r| <= 5.8e-10 on
// |x| <= 1/4 -- is taken from weight moments as well, every wave tile whose max|b| is inside 1/4 (known after ONE screening
// MFMA per 32 x 32 block, the (h, h + m) part of the split product) contributes nothing and is skipped, and the other tiles
// reduce the correction r(x) - p6(x) with the same range tiers.
// The bilinear part runs on the bf16 matrix pipe as a 3-way split product with f32 accuracy
// (v_mfma_f32_32x32x16_bf16), the remainder polynomial + weighted reduction on the VALU in packed
// f32 (v_pk_fma_f32); MFMA and f32 FMA-class VALU time add on a gfx950 SIMD (tools/ubench_overlap.hip).
//
// Work decomposition: workgroup = 4 waves = 256 rows of one (b, pair); a wave owns 64 rows
// (two 32x32 MFMA row tiles; split A operands, rho' (the MFMA C operand) and the 32 row weights
// stay in registers) and sweeps all columns in tiles of 32, software-prefetching the next
// tile's pre-split inducing inputs (two 16-B loads per lane) and gamma_j / w_j.  Partial sums
// are flushed to f64 once per column tile and written to a slab (no atomics: bitwise
// reproducible).  The 1-D grid is remapped so that the row panels of one (b, pair) land on the
// same XCD and share its L2.
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"
#include "mm_mono.h"
#include "mm_f32_tile.h"

__device__ __forceinline__ void mm_decode_pair_o(int p, int L, int& a, int& a2) {
  int r = p - L, i = 0;
  while (r >= L - 1 - i) { r -= L - 1 - i; ++i; }
  a = i; a2 = i + 1 + r;
}

// largest tile range max|b| for which the 2-way (h + m) product is used without the (h,l)/(l,h) terms
// (measured at C3: 1/64, 1/32 and 1/16 all leave the Sff error at 3.5e-6; 1/32 keeps a factor 4 in hand)
#ifndef MM_TWO_WAY_MAX
#define MM_TWO_WAY_MAX 0.03125f
#endif
// LDS-staged sweep: workgroups per (b, pair), each paying the 64 KB fill for its share of the row panels.  Measured
// at C3 (off-diagonal segment, pilco / BASELINE recipe): 1 -> 0.574 / 6.66 ms, 2 -> 0.444 / 6.27, 4 -> 0.380 / 6.19,
// 8 -> 0.420 / 6.31: with most (b, pair) items leaving at once (wholly inside the collapsed range) the few that sweep
// are the grid's tail, and finer items balance it.  Round 5, with a workgroup's panels interleaved over the (norm-ordered) rows
// and the Cauchy-Schwarz skipping (BASELINE recipe): 1 -> 1.135, 2 -> 0.99, 4 -> 1.03, 8 -> 1.09 ms
#ifndef MM_F32_PPW_DIV
#define MM_F32_PPW_DIV 2
#endif
// 0: the sweep without its rounding-error estimate (A/B measurement of what the accuracy contract costs: nothing is ever routed)
#ifndef MM_ROUTE_EST
#define MM_ROUTE_EST 1
#endif
#ifndef MM_F32_WAVES
#define MM_F32_WAVES 2
#endif
// sum_r w_r * x_r^3 * R_DEG(x_r) over the 16 register pairs of a wave tile.  The Horner steps run
// "vertically" over the pairs so that consecutive v_pk_fma_f32 are independent.
// CC: the (b, pair) is collapsed -- its moments already carry p6 = x^3 (C0 + C1 x + C2 x^2 + C3 x^3) (mm_common.h), which is
// subtracted from the four leading coefficients at compile time (the corrected polynomial costs no instruction; the
// differences are exact in f32: Sterbenz)
template <int DEG, int CC>
struct MMRemC {
  // CC: what the moments already carry for these rows -- 0: nothing beyond order 2; 1: p6 (a collapsed row group); 2: the cubic
  // term C0 x^3 alone (the other row groups of an item that has collapsed ones: mm_mono.h)
  static constexpr float get(int k) {
    const float c6[4] = {MM_C6_C0, MM_C6_C1, MM_C6_C2, MM_C6_C3};
    return MMRem<DEG>::c[k] - ((CC == 1 && k < 4) ? c6[k] : ((CC == 2 && k == 0) ? c6[0] : 0.0f));
  }
};
template <int DEG, int CC>
__device__ __forceinline__ f32x2 mm_weighted_rem(const f32x2 (&xx)[16], const f32x2 (&wrow)[2][8]) {
  f32x2 parts[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  // two halves of 8 register pairs (one 32-row MFMA tile each): 8 independent chains cover the packed-FMA latency,
  // and the temporaries of one half are dead before the other starts (register pressure)
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    f32x2 pp[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
      pp[r] = mm_pkfma(MM_PK((MMRemC<DEG, CC>::get(DEG))), xx[8 * hh + r], MM_PK((MMRemC<DEG, CC>::get(DEG - 1))));
#pragma unroll
    for (int k = DEG - 2; k >= 0; --k)
#pragma unroll
      for (int r = 0; r < 8; ++r)
        pp[r] = mm_pkfma(pp[r], xx[8 * hh + r], MM_PK((MMRemC<DEG, CC>::get(k))));
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const f32x2 wx = wrow[hh][r] * xx[8 * hh + r];                   // w_i * x
      const f32x2 tv = (xx[8 * hh + r] * xx[8 * hh + r]) * pp[r];      // x^2 * R(x)
      parts[r & 3] = mm_pkfma(tv, wx, parts[r & 3]);                   // += w_i * x^3 * R(x)
    }
  }
  return (parts[0] + parts[1]) + (parts[2] + parts[3]);
}

// First tier with the row weight folded into the two coefficients (w c0, w c1 kept per row):
//   w x^3 (c0 + c1 x) = (x^2 x) * fma(w c1, x, w c0):  four packed ops per entry pair instead of five.
__device__ __forceinline__ f32x2 mm_weighted_rem1(const f32x2 (&xx)[16], const f32x2 (&wc0)[2][8],
                                                  const f32x2 (&wc1)[2][8]) {
  f32x2 parts[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    f32x2 t3[8], rw[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) t3[r] = xx[8 * hh + r] * xx[8 * hh + r];
#pragma unroll
    for (int r = 0; r < 8; ++r) rw[r] = mm_pkfma(wc1[hh][r], xx[8 * hh + r], wc0[hh][r]);
#pragma unroll
    for (int r = 0; r < 8; ++r) t3[r] = t3[r] * xx[8 * hh + r];
#pragma unroll
    for (int r = 0; r < 8; ++r) parts[r & 3] = mm_pkfma(t3[r], rw[r], parts[r & 3]);
  }
  return (parts[0] + parts[1]) + (parts[2] + parts[3]);
}

// ND8: number of 8-wide blocks of input dimensions (d <= 8 * ND8).
//
// The bilinear form A_i . zc_j runs on the bf16 matrix pipe as a 3-way split product
// (h/m/l bf16 parts, six cross terms packed into three K=16 MFMAs per 8 dims: f32-equivalent
// accuracy), because v_mfma_f32_32x32x2_f32 was measured to serialise with the VALU
// on a SIMD (it shares the f32 FMA datapath) while the bf16 MFMA overlaps with it.
//   stationary operand (registers, split once per sweep): A_i = G^T zeta_i of the wave's 64 rows and
//   their factored weights what_i; streaming operand: the model's pre-split centred inducing inputs
//   of latent a' (b-independent, L2-resident) and what'_j per column.
// LDSZ: the (h, m) parts of the column latent's split inputs -- all the screening product and the 2-way product read --
// are staged ONCE per workgroup in LDS (Mp * 32 ND8 bytes: 64 KB at C3) and shared by its four waves: a collapsed
// (b, pair) is a sweep of one MFMA + one range check per 32 x 32 block, which from L2 would be bound by the
// 1 KB per wave tile of operand traffic (measured 2.0 ms at C3 against 0.4 ms of MFMA time).
template <int ND8, bool LDSZ>
__global__ __launch_bounds__(256, (ND8 >= 4 ? 1 : MM_F32_WAVES)) void k_qred_f32_mfma(const unsigned short* __restrict__ Zs3,
                                                          int L, int Mp, int d, int P, int Po, int NS,
                                                          int npanel, int ppw, int nwork, int force_worst,
                                                          const unsigned int* __restrict__ amax,
                                                          const unsigned int* __restrict__ amaxc,
                                                          const unsigned char* __restrict__ gflag, const float* __restrict__ gmax2,
                                                          const double* __restrict__ zmax2, const float* __restrict__ zt2,
                                                          const float* __restrict__ rowO,
                                                          const float* __restrict__ colO,
                                                          double* __restrict__ partB, float* __restrict__ estO,
                                                          int* __restrict__ rcount) {
  // XCD-aware remap of the 1-D grid (blocks b and b+8 share an XCD): consecutive work items
  // go to the same XCD.  Bijective for any nwork (cdna guide T1).
  const int orig = blockIdx.x;
  if (orig == 0 && threadIdx.x == 0 && rcount) { rcount[0] = 0; rcount[1] = 0; }   // this pass's route list (k_route_decide follows)
  const int xcd = orig & 7, slot = orig >> 3;
  const int qn = nwork >> 3, rn = nwork & 7;
  const int wi = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  // a workgroup owns ppw consecutive row panels of one (b, pair) (all of them when the operand image is staged in
  // LDS: the 64 KB fill is then paid once per (b, pair)); npw = ceil(npanel / ppw) workgroups per (b, pair)
  const int npw = (npanel + ppw - 1) / ppw;
  const int pgrp = wi % npw;
  const int t = wi / npw;
  const int lp = t % Po, b = t / Po;
  const int p = L + lp;
  int a, a2;
  mm_decode_pair_o(p, L, a, a2);

  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  if constexpr (ND8 == 1) {
    // Cauchy-Schwarz already bounds every |b_ij| of this (b, pair) by MM_C6_MAX: the remainder is p6 everywhere, which the
    // moments carry (k_spoly, k_spoly56) -- every tile would be skipped: no sweep at all
    if (!force_worst && zmax2 && mm_collapse_bound2(amax[(size_t)b * Po + lp], zmax2[a2]) <= MM_INSIDE_BOUND2) {
      for (int panel = (int)threadIdx.x; panel < npanel; panel += 256) {
        if (panel % npw != pgrp) continue;                   // (this workgroup's panels: pgrp, pgrp + npw, ...)
        partB[((size_t)b * P + p) * NS + panel] = 0.0;
        if (estO) estO[((size_t)b * Po + lp) * npanel + panel] = 0.0f;        // exact (f64 moments): nothing to estimate
      }
      return;
    }
  }
  // Row-group collapse (mm_mono.h): does any wave of this workgroup have a tile to visit?  A collapsed group whose Cauchy-Schwarz
  // bound with the column latent's LARGEST point is inside the collapsed range has none (gmax2: its rows' max |A_i|^2 from
  // k_pairvec_reg) -- with the pack in norm order those are the first panels of most items; a workgroup of such groups alone
  // writes its zero partials and leaves before the 64 KB fill
  const bool icoll_wg = (ND8 == 1) && !force_worst && zmax2 != nullptr && zt2 != nullptr && gmax2 != nullptr &&
                        mm_item_collapsed(amaxc[(size_t)b * Po + lp]);
  const size_t grp0 = ((size_t)b * Po + lp) * (size_t)(Mp / MM_GROUP_ROWS);
  if constexpr (ND8 == 1) {
    if (icoll_wg) {
      bool need = false;
      for (int panel = pgrp; panel < npanel; panel += npw) {
        const int g = (panel * MM_PANEL_ROWS + wv * 64) >> 6;
        if (g < Mp / MM_GROUP_ROWS)
          need = need || !(gflag[grp0 + g] != 0 && mm_collapse_bound2(__float_as_uint(gmax2[grp0 + g]), zmax2[a2]) <= MM_INSIDE_BOUND2);
      }
      if (!__syncthreads_or(need ? 1 : 0)) {
        for (int panel = (int)threadIdx.x; panel < npanel; panel += 256) {
          if (panel % npw != pgrp) continue;
          partB[((size_t)b * P + p) * NS + panel] = 0.0;
          if (estO) estO[((size_t)b * Po + lp) * npanel + panel] = 0.0f;
        }
        return;
      }
    }
  }
  extern __shared__ __align__(16) char zlds[];     // LDSZ: [Mp/32 tiles][2 parts (h, m)][32 columns][16 ND8 bytes]
  if constexpr (LDSZ) {
    // the first two parts of every tile are contiguous in the packed model ([tile][h, m, l][32][16 ND8]): copy
    // 1024 ND8 of every 1536 ND8 bytes, 16 bytes per thread and pass
    const char* src = reinterpret_cast<const char*>(Zs3 + (size_t)a2 * Mp * (24 * ND8));
    const int nchunk = (Mp >> 5) * 64 * ND8;         // 16-byte chunks
    // batches of 8 chunks per thread: all loads of a batch are issued before its LDS stores (a load -> store loop
    // would pay one memory latency per chunk)
    for (int c0 = 0; c0 < nchunk; c0 += 8 * 256) {
      u32x4 tmp[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int c = c0 + u * 256 + (int)threadIdx.x;
        c = c < nchunk ? c : nchunk - 1;
        const int tile = c / (64 * ND8), within = c - tile * (64 * ND8);
        tmp[u] = *reinterpret_cast<const u32x4*>(src + (size_t)tile * (1536 * ND8) + (size_t)within * 16);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + u * 256 + (int)threadIdx.x;
        if (c < nchunk) *reinterpret_cast<u32x4*>(zlds + (size_t)c * 16) = tmp[u];
      }
    }
    __syncthreads();
  }
  __shared__ double red[4];
  __shared__ float redf[4];
  // the workgroup's panels are INTERLEAVED (pgrp, pgrp + npw, ...): with the pack in norm order the rows of large |A_i| -- the
  // tiles that are not skipped -- are the last panels, and consecutive panels would give one workgroup of four all of an item's work
  for (int panel = pgrp; panel < npanel; panel += npw) {
  const int row0 = panel * MM_PANEL_ROWS + wv * 64;
  double sum = 0.0;
  // running estimate of the sweep's own rounding error (mm_common.h: MM_ROUTE_TOL; mm_route.hip): per lane -- one column,
  // the lane's 32 rows -- sum over the reduced tiles of (max|b|^3 what'_j)^2, two tiles per packed instruction; the rows' sum
  // of squares and the (1 + X + X^2) factor of rho (X = the lane's largest |b|: one v_max3 per tile pair) once per sweep
  f32x2 est2 = {0.0f, 0.0f};
  float rowsq = 0.0f, xall = 0.0f;
  // icoll: the item has collapsed groups -- then the CUBIC term of every row is in the f64 moments (k_spoly), and a group that
  // is not collapsed reduces r(b) - C0 b^3 on every tile.  coll: this wave's 64 rows are a COLLAPSED row group (mm_mono.h:
  // k_pairvec_reg decided, with the Cauchy-Schwarz bound of the group's rows, |b_ij| <= |A_i| |zc_j|); g2: its rows' max |A_i|^2
  const bool icoll = icoll_wg;
  const bool coll = icoll && row0 < Mp && gflag[grp0 + (row0 >> 6)] != 0;
  const float g2 = coll ? gmax2[grp0 + (row0 >> 6)] : 0.0f;
  // (a collapsed group with every tile inside the collapsed range: nothing to load, nothing to add)
  const bool wave_inside = coll && mm_collapse_bound2(__float_as_uint(g2), zmax2[a2]) <= MM_INSIDE_BOUND2;
  if (row0 < Mp && !wave_inside) {   // Mp % 128 == 0, so a wave's 64 rows are all inside or all outside
    const float* ra = rowO + ((size_t)b * Po + lp) * (size_t)(d + 1) * Mp;   // [d+1][Mp]: A_i, what_i
    const float* wcf = colO + ((size_t)b * Po + lp) * Mp;                    // what'_j
    // pre-split centred inducing inputs of latent a': [Mp/32][3 (h,m,l)][32][8 ND8] bf16
    const unsigned short* zs = Zs3 + (size_t)a2 * Mp * (24 * ND8);
    // bound2: the bound over all collapsed groups of the item (the screening margin)
    const float bound2 = zmax2 ? mm_collapse_bound2(amaxc[(size_t)b * Po + lp], zmax2[a2]) : 3.0e38f;
    // a tile is skipped on the screening product alone: its error is <= 2^-9 sum_k |A_k||Z_k| <= 2^-9 sqrt(bound2)
    const float thr_skip = MM_C6_MAX - 0.00390625f * __builtin_sqrtf(bound2) - 1e-6f;

    // ---- stationary operands --------------------------------------------------------------
    bf16x8 a1[2][ND8], a2v[2][ND8], a3[2][ND8];
    f32x2 wrow[2][8], wc0[2][8], wc1[2][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = row0 + rt * 32 + l31;
#pragma unroll
      for (int nb = 0; nb < ND8; ++nb) {
        unsigned int hh[8], mm[8], ll[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = nb * 8 + j;
          const float v = ra[(size_t)(k < d ? k : d) * Mp + row];
          mm_split3(k < d ? v : 0.0f, hh[j], mm[j], ll[j]);
        }
        u32x4 ph, pm, pl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ph[j] = hh[2 * j] | (hh[2 * j + 1] << 16);
          pm[j] = mm[2 * j] | (mm[2 * j + 1] << 16);
          pl[j] = ll[2 * j] | (ll[2 * j + 1] << 16);
        }
        // MFMA1: (m,m) | (m,h)   MFMA3: (h,m) | (h,h)   MFMA2: (h,l) | (l,h)     [lane half 0 | 1]
        // MFMA1 + MFMA3 alone are the 2-way (16-bit) product; MFMA2 adds the 2^-16 terms
        a1[rt][nb] = __builtin_bit_cast(bf16x8, pm);
        a2v[rt][nb] = __builtin_bit_cast(bf16x8, h ? pl : ph);
        a3[rt][nb] = __builtin_bit_cast(bf16x8, ph);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rr = row0 + rt * 32 + 8 * g + 4 * h;
        const float4 v = *reinterpret_cast<const float4*>(ra + (size_t)d * Mp + rr);   // factored row weights
        wrow[rt][2 * g + 0] = (f32x2){v.x, v.y};
        wrow[rt][2 * g + 1] = (f32x2){v.z, v.w};
        if constexpr (ND8 == 1) {
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
            wc0[rt][2 * g + q2] = wrow[rt][2 * g + q2] * MM_PK(MMRem<1>::c[0]);
            wc1[rt][2 * g + q2] = wrow[rt][2 * g + q2] * MM_PK(MMRem<1>::c[1]);
          }
        }
      }
    }
    {
      f32x2 rs = {0.0f, 0.0f};
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 8; ++r) rs = mm_pkfma(wrow[rt][r], wrow[rt][r], rs);
      rowsq = rs[0] + rs[1];
    }

    // ---- streaming operands: lane half 0 reads parts (m, l), half 1 reads (h, h) ------------
    // lane byte offsets inside the latent's [Mp/32][3][32][8 ND8] bf16 block; the tile offset is wave-uniform and each
    // half-wave reads one contiguous 512 ND8-byte part of the tile
    const unsigned int tile_bytes = 32u * 48u * ND8, part_bytes = 32u * 16u * ND8;
    const unsigned int offA = (h ? 0u : 1u) * part_bytes + (unsigned int)l31 * (16u * ND8);
    const unsigned int offB = (h ? 0u : 2u) * part_bytes + (unsigned int)l31 * (16u * ND8);
    const char* zbase = reinterpret_cast<const char*>(zs);
    const int nct = Mp >> 5;                     // Mp % 128 == 0: nct is a multiple of 4

    // zA: the (m | h) parts (from LDS when staged); zB: the (l | h) parts, always from L2 -- prefetched with the tile
    // for a (b, pair) that is not collapsed, fetched on demand for a collapsed one (where few tiles get that far)
    auto load_zA = [&](int ct, u32x4 (&zA)[ND8]) {
#pragma unroll
      for (int nb = 0; nb < ND8; ++nb) {
        if constexpr (LDSZ) zA[nb] = *reinterpret_cast<const u32x4*>(zlds + (size_t)ct * (1024 * ND8) + offA + nb * 16);
        else zA[nb] = *reinterpret_cast<const u32x4*>(zbase + (size_t)ct * tile_bytes + offA + nb * 16);
      }
    };
    auto load_zB = [&](int ct, u32x4 (&zB)[ND8]) {
#pragma unroll
      for (int nb = 0; nb < ND8; ++nb) zB[nb] = *reinterpret_cast<const u32x4*>(zbase + (size_t)ct * tile_bytes + offB + nb * 16);
    };

    // b_ij in two stages.  Stage 1: the 2-way split product (h + m parts, 2^-17 relative): enough when the
    // whole tile has |b| <= MM_TWO_WAY_MAX (1/32) -- the kernel only reduces r(b) = O(b^3), whose sensitivity to an error
    // in b is b^2/2.  Stage 2 (wave-uniform, only for larger tiles): the (h,l) and (l,h) terms.
    // screening product = MFMA3 alone: (h, m) | (h, h) = A_h . (Z_h + Z_m), |error| <= 2^-9 sum_k |A_k||Z_k|
    auto mfma_tile_screen = [&](const u32x4 (&zA)[ND8], f32x16 (&acc)[2]) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < ND8; ++nb)
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[rt][nb], __builtin_bit_cast(bf16x8, zA[nb]), c, 0, 0, 0);
        acc[rt] = c;
      }
    };
    // + MFMA1: (m, m) | (m, h): together with the screening product the 2-way (h + m) product
    auto mfma_tile_m = [&](const u32x4 (&zA)[ND8], f32x16 (&acc)[2]) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f32x16 c = acc[rt];
#pragma unroll
        for (int nb = 0; nb < ND8; ++nb)
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[rt][nb], __builtin_bit_cast(bf16x8, zA[nb]), c, 0, 0, 0);
        acc[rt] = c;
      }
    };
    auto mfma_tile_l = [&](const u32x4 (&zB)[ND8], f32x16 (&acc)[2]) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f32x16 c = acc[rt];
#pragma unroll
        for (int nb = 0; nb < ND8; ++nb)
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2v[rt][nb], __builtin_bit_cast(bf16x8, zB[nb]), c, 0, 0, 0);
        acc[rt] = c;
      }
    };
    // max |b| of a finished tile: one v_max3_f32 |a|, |b|, m per entry pair (needs -fno-honor-nans)
    auto tile_max = [&](const f32x16 (&acc)[2]) {
      // four independent v_max3 chains (a single chain of 16 dependent ops is the critical path of a skipped tile)
      float m4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int r = 0; r < 16; ++r)
        m4[r & 3] = fmaxf(fmaxf(m4[r & 3], fabsf(acc[r >> 3][2 * (r & 7)])), fabsf(acc[r >> 3][2 * (r & 7) + 1]));
      return fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    };
    // b_ij = acc: a pure bilinear form (rho'_i, gamma_j live in the weights); tile range -> tier.
    // (sub0, sub1) = (c0, c1) of the first tier for a collapsed (b, pair) -- already in the moments -- else (0, 0).
    // collm (compile time): the (b, pair) is collapsed -- the moments already carry c0 x^3 + c1 x^4 of the first tier
    auto reduce_tile = [&](auto collm, const f32x16 (&acc)[2], float mx, float wc) {
      constexpr int CC = decltype(collm)::value;            // 0 / 1 / 2: MMRemC
      f32x2 xx[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) xx[r] = (f32x2){acc[r >> 3][2 * (r & 7)], acc[r >> 3][2 * (r & 7) + 1]};
      f32x2 part2;
      // wave-uniform tier choice (ballots, no cross-lane reduction)
      if (CC == 0 && !__any(mx > MM_TIER1_MAX)) {
        // (the folded coefficients cost 64 more VGPRs: only where the operand registers leave room, d <= 8)
        if constexpr (ND8 == 1) part2 = mm_weighted_rem1(xx, wc0, wc1);
        else part2 = mm_weighted_rem<1, 0>(xx, wrow);
      } else if (!__any(mx > MM_C6_MAX)) {
        // a collapsed row group has nothing left to add below 1/4: its moments carry this tier's own polynomial
        if constexpr (CC == 1) part2 = (f32x2){0.0f, 0.0f};
        else part2 = mm_weighted_rem<3, CC>(xx, wrow);
      } else if (!__any(mx > 0.5f)) {
        part2 = mm_weighted_rem<4, CC>(xx, wrow);
      } else if (!__any(mx > 1.0f)) {
        part2 = mm_weighted_rem<5, CC>(xx, wrow);
      } else {
        part2 = (f32x2){0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int e2 = 0; e2 < 2; ++e2) {
            // |b| > 1: exp2 on the transcendental unit, then the three leading terms subtracted
            const float x = fminf(xx[r][e2], MM_EXP_CAP_F32);     // (mm_common.h: exponent caps)
            const float xs = fminf(fmaxf(x, -1.0f), 1.0f);
            const float big = (__builtin_amdgcn_exp2f(x * 1.44269504f) - 1.0f) - fmaf(0.5f * x, x, x);
            float e = (fabsf(x) <= 1.0f) ? mm_rem_p5(xs) : big;
            if constexpr (CC == 1) e -= (x * x) * x * fmaf(fmaf(fmaf(MM_C6_C3, x, MM_C6_C2), x, MM_C6_C1), x, MM_C6_C0);
            if constexpr (CC == 2) e -= (x * x) * (x * MM_C6_C0);
            part2[e2] = fmaf(wrow[r >> 3][r & 7][e2], e, part2[e2]);
          }
      }
      sum += (double)wc * (double)(part2[0] + part2[1]);
    };
    // one wave tile: screening product; a collapsed (b, pair) stops here when the whole tile is inside the collapsed
    // range; else the rest of the split product, the range tier and the weighted reduction
    // collc (compile time): SCREENED sweep -- the screening check, and neither the column weight nor the (l | h) parts
    // prefetched (few tiles of such an item need them, and a prefetched global load would put its latency on every tile
    // of a sweep that is otherwise 2 MFMAs long); collm: collapsed coefficients (reduce_tile)
    // (emx, ewc): what the error estimate takes from this tile -- its max|b| of the lane and the column weight; zeros for a
    // tile that contributes exact moments only
    auto process_tile = [&](auto collc, auto collm, int ct, const u32x4 (&zA)[ND8], u32x4 (&zB)[ND8], float wc, float& emx, float& ewc) __attribute__((always_inline)) {
      constexpr bool CM = decltype(collc)::value;
      f32x16 acc[2];
      emx = 0.0f; ewc = 0.0f;
      mfma_tile_screen(zA, acc);
      if constexpr (CM) {
        const float ms = tile_max(acc);
        if (!__any(ms > thr_skip)) return;
      }
      mfma_tile_m(zA, acc);
      const float mx = force_worst ? 2.0f : tile_max(acc);       // MM_FORCE_WORST_TIER: wave-uniform override
      if constexpr (CM) {
        if (!__any(mx > MM_C6_MAX)) return;                           // inside the collapsed range: all in the moments
      }
      if constexpr (CM) wc = wcf[ct * 32 + l31];
      emx = mx; ewc = wc;
      if (__any(mx > MM_TWO_WAY_MAX)) {
        if constexpr (CM) load_zB(ct, zB);
        mfma_tile_l(zB, acc);
      }
      reduce_tile(collm, acc, mx, wc);
    };

    // two-stage register ping-pong: tile ct + 1 is in flight while tile ct is reduced.  (Issuing the MFMAs
    // of tile ct + 1 interleaved with the v_max3 range check of tile ct -- a second accumulator set,
    // 182 VGPRs -- was measured: 4.27 ms against 4.25 ms; three waves per SIMD already overlap the two.)
    // A collapsed row group skips, without touching them, the column tiles whose Cauchy-Schwarz bound with ITS rows is inside the
    // collapsed range: max_i |A_i|^2 (this wave's rows) x max_j |zc'_j|^2 (the tile's 32 points, MMModelLayout::zt2) <=
    // MM_INSIDE_BOUND2.  The pack's norm order makes those tiles a PREFIX: the sweep starts at the first tile that is not
    // (BASELINE recipe: two thirds of the screened wave tiles; the screening product + range check was half the sweep's time).
    int ct_first = 0;
    if (coll) {
      const float a2w = g2 * 1.000001f;                     // (gmax2 is rounded up from f64; the f32-rounded operands add < 2e-7)
      const float* zt = zt2 + (size_t)a2 * nct;
      int cf = nct;
      for (int c0 = 0; c0 < nct; c0 += 64) {
        const int c = c0 + lane;
        const bool outside = c < nct && a2w * zt[c < nct ? c : 0] > MM_INSIDE_BOUND2;
        const unsigned long long bal = __ballot(outside);
        if (bal) { cf = c0 + (int)__builtin_ctzll(bal); break; }
      }
      ct_first = cf & ~1;
    }
    auto sweep = [&](auto collc, auto collm) __attribute__((always_inline)) {
      constexpr bool CM = decltype(collc)::value;
      u32x4 zA0[ND8], zB0[ND8], zA1[ND8], zB1[ND8];
      float w0, w1;
      w0 = w1 = 0.0f;
      const int ct0 = CM ? ct_first : 0;
      if (ct0 >= nct) return;
      load_zA(ct0, zA0);
      if constexpr (!CM) { load_zB(0, zB0); w0 = wcf[l31]; }
      for (int ct = ct0; ct < nct; ct += 2) {
        load_zA(ct + 1, zA1);
        if constexpr (!CM) { load_zB(ct + 1, zB1); w1 = wcf[(ct + 1) * 32 + l31]; }
        float emx0, ewc0, emx1, ewc1;
        process_tile(collc, collm, ct, zA0, zB0, w0, emx0, ewc0);
        const int cn = ct + 2 < nct ? ct + 2 : ct;               // clamped: the last pass re-reads its own tile
        load_zA(cn, zA0);
        if constexpr (!CM) { load_zB(cn, zB0); w0 = wcf[cn * 32 + l31]; }
        process_tile(collc, collm, ct + 1, zA1, zB1, w1, emx1, ewc1);
#if MM_ROUTE_EST
        {
          const f32x2 emx = {emx0, emx1}, ewc = {ewc0, ewc1};
          const f32x2 u = (emx * emx) * (emx * ewc);             // both tiles at once: 4 packed instructions per tile pair
          est2 = mm_pkfma(u, u, est2);
          xall = fmaxf(fmaxf(xall, emx0), emx1);
        }
#endif
      }
    };
    // A (b, pair) is collapsed only where its bound lets the screening skip most tiles (MM_COLLAPSE_BOUND2 = (1/2)^2: on the
    // BASELINE recipe items with a bound in (1/4, 1/2] have 98 % of their wave tiles under 1/4, items beyond 1/2 a third,
    // tools/tile_hist_baseline.py).
    // (Two instantiations, not a run-time flag: a non-collapsed item must not pay for the collapsed coefficients -- as a flag
    // they cost the exp2 branch 4 more ops per entry: forced-worst C3 12.5 -> 16.8 ms)
    // (the third: the dense row groups of an item that has collapsed ones, cubic term in the moments)
    if (coll) sweep(mm_true{}, mm_int<1>{});
    else if (icoll) sweep(mm_false{}, mm_int<2>{});
    else sweep(mm_false{}, mm_int<0>{});
  }
  // workgroup reduction -> slab
  float estl;
  {
    const float xf = fminf(xall, 8.0f);                     // (beyond |b| = 8 the estimate is astronomically large anyway)
    const float pf = fmaxf(fmaf(xf, xf, xf) + 1.0f, __expf(xf));   // e^X <= 1 + X + X^2 only up to X = 1.79: the larger of the two (X <= 8)
    estl = ((est2[0] + est2[1]) * rowsq) * (pf * pf);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sum += __shfl_down(sum, off, 64); estl += __shfl_down(estl, off, 64); }
  __syncthreads();                               // (the previous panel's red[] has been read)
  if (lane == 0) { red[wv] = sum; redf[wv] = estl; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partB[((size_t)b * P + p) * NS + panel] = red[0] + red[1] + red[2] + red[3];
    if (estO) estO[((size_t)b * Po + lp) * npanel + panel] = (redf[0] + redf[1]) + (redf[2] + redf[3]);
  }
  }
}

extern "C" int mm_mfma_supported(int d) { return d >= 1 && d <= 32; }

int mm_mfma_num_slots(int Mp) { return (Mp + MM_PANEL_ROWS - 1) / MM_PANEL_ROWS; }

int mm_launch_qred_mfma(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl,
                        int B, int L, int d, int flags, hipStream_t stream) {
  const int npanel = mm_mfma_num_slots(wl.Mp);
  const int force_worst = (flags & MM_FORCE_WORST_TIER) ? 1 : 0;
  const unsigned int* amax = (const unsigned int*)(ws + wl.amax);
  const double* zmax2 = mm_moment_deg(d) >= 4 ? (const double*)(packed + ml.zmax2) : nullptr;
  // the (h, m) parts of one latent in LDS when they fit beside a second workgroup (<= 64 KB): d <= 8 and M <= 2048
  const size_t zbytes = (size_t)wl.Mp * 32 * ml.nd8;
  const bool ldsz = ml.nd8 == 1 && zbytes <= 65536;
  // LDS-staged: one workgroup per (b, pair) (the fill is paid once) as long as that still leaves >= 8 rounds of
  // 512 resident workgroups; else one per panel
  int ppw = 1;
  if (ldsz && (long long)wl.Po * B >= 4096) ppw = (npanel + MM_F32_PPW_DIV - 1) / MM_F32_PPW_DIV;
  const long long nwork_ll = (long long)((npanel + ppw - 1) / ppw) * wl.Po * B;
  if (nwork_ll <= 0 || nwork_ll > 0x7fffffffLL) return MM_E_DIM;
  const int nwork = (int)nwork_ll;
  const unsigned short* Zs3 = (const unsigned short*)(packed + ml.Zs3);
  const float* rowO = (const float*)(ws + wl.rowO);
  const float* colO = (const float*)(ws + wl.colO);
  double* partB = (double*)(ws + wl.partB);
  float* estO = (float*)(ws + wl.estO);
#define MM_LAUNCH_ND(ND_, LZ_, SH_)                                                                          \
  hipLaunchKernelGGL((k_qred_f32_mfma<ND_, LZ_>), dim3(nwork), dim3(256), SH_, stream, Zs3, L, wl.Mp, d,    \
                     wl.P, wl.Po, wl.NS, npanel, ppw, nwork, force_worst, amax, (const unsigned int*)(ws + wl.amaxc), \
                     (const unsigned char*)(ws + wl.gflag), (const float*)(ws + wl.gmax2), zmax2, (const float*)(packed + ml.zt2), rowO, colO, partB, estO, (int*)(ws + wl.rcount))
  switch (ml.nd8) {
    case 1: if (ldsz) MM_LAUNCH_ND(1, true, zbytes); else MM_LAUNCH_ND(1, false, 0); break;
    case 2: MM_LAUNCH_ND(2, false, 0); break;
    case 3: MM_LAUNCH_ND(3, false, 0); break;
    default: MM_LAUNCH_ND(4, false, 0); break;
  }
#undef MM_LAUNCH_ND
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
