// Shared pieces of the f32 tile kernels (mm_mfma.hip: the forward's off-diagonal reduce; mm_bwd_f32.hip: the backward's
// remainder aggregates): packed-f32 helpers, the bf16 split, and the range-tiered near-minimax remainder polynomials.
#pragma once
#include <hip/hip_runtime.h>
#include "mm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Packed f32 FMA (v_pk_fma_f32): gfx950 issues a wave64 VALU instruction over 4 cycles, so the
// f32 vector peak (64 FLOP/clk/SIMD) is only reached with two FMAs per lane per instruction.
__device__ __forceinline__ f32x2 mm_pkfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#define MM_PK(c_) ((f32x2){(c_), (c_)})

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct mm_true { static constexpr bool value = true; };
struct mm_false { static constexpr bool value = false; };
template <int V> struct mm_int { static constexpr int value = V; };

__device__ __forceinline__ unsigned int mm_f2bf(float x) {
  const __bf16 b = (__bf16)x;                                  // v_cvt_pk_bf16_f32, round to nearest even
  return (unsigned int)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float mm_bf2f(unsigned int u) { return __builtin_bit_cast(float, u << 16); }

// x = h + m + l with h, m, l bf16 (8 + 8 + 8 significand bits: exact to ~2^-24 |x|)
__device__ __forceinline__ void mm_split3(float x, unsigned int& h, unsigned int& m, unsigned int& l) {
  h = mm_f2bf(x);
  float r = x - mm_bf2f(h);
  m = mm_f2bf(r);
  r -= mm_bf2f(m);
  l = mm_f2bf(r);
}

// The tile kernel reduces only the REMAINDER r(x) = expm1(x) - x - x^2/2 of every entry: the constant,
// linear and quadratic parts of sum_ij what_i what'_j (1 + expm1(b_ij)) are taken exactly from f64 moments
// of the weights (k_wmom_gemm / k_spoly in mm_moments.hip).  In f32 the sum is ill-conditioned
// (sum |what_i E_ij what'_j| >> |S|: weights of +-17 at M = 2000), and it is the rounding of the linear
// term that costs the digits; r(x) = O(x^3) carries the same relative rounding but is 1e-3..1e-5 of it
// (tools/error_budget.py: 2.5e-3 -> 1e-8 of max|Sff| at C3).
//
// Near-minimax R(x) ~ r(x) / x^3 by range tier, |x^3 R(x) - r(x)| <= 5e-8 |x| with fused f32 Horner
// steps (tools/minimax_remainder.py):  |x| <= 1/20 (MM_TIER1_MAX): degree 1,  <= 1/4: 3,  <= 1/2: 4,  <= 1: 5.
// Along the C3 rollout 90-100 % of the 64 x 32 wave tiles are in the first tier (tools/tier_stats.py) -- which is why,
// for d <= 8, that tier's own approximant c0 x^3 + c1 x^4 is taken from degree-3/4 moments as well (the collapse,
// mm_moments.hip): a collapsed (b, pair) only visits tiles with max|b| > 1/20 and reduces r(x) - c0 x^3 - c1 x^4 there.
template <int DEG> struct MMRem;
template <> struct MMRem<1> {
  static constexpr float c[2] = {MM_REM1_C0, MM_REM1_C1};      // mm_common.h: shared with the moment collapse
};
template <> struct MMRem<3> {
  static constexpr float c[4] = {MM_C6_C0, MM_C6_C1, MM_C6_C2, MM_C6_C3};   // mm_common.h: it IS the collapse's polynomial p6
};
template <> struct MMRem<4> {
  static constexpr float c[5] = {1.666666716e-01f, 4.166586325e-02f, 8.333111182e-03f, 1.398149878e-03f, 1.998390071e-04f};
};
template <> struct MMRem<5> {
  static constexpr float c[6] = {1.666671634e-01f, 4.166677967e-02f, 8.330268785e-03f, 1.388406614e-03f,
                                 2.037364029e-04f, 2.544890958e-05f};
};


__device__ __forceinline__ float mm_rem_p5(float x) {       // r(x) on [-1, 1]
  float p = fmaf(MMRem<5>::c[5], x, MMRem<5>::c[4]);
  p = fmaf(p, x, MMRem<5>::c[3]); p = fmaf(p, x, MMRem<5>::c[2]);
  p = fmaf(p, x, MMRem<5>::c[1]); p = fmaf(p, x, MMRem<5>::c[0]);
  return (x * x) * p * x;
}

