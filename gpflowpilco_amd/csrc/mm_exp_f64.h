// e^x in f64 for the diagonal-pair sweeps (forward: mm_f64.hip, backward: mm_backward.hip): near-minimax range tiers and the
// any-argument form.
#pragma once
#include <hip/hip_runtime.h>
#include "mm_common.h"

// LOWP tiers: near-minimax e^x ~ 1 + x (c[0] + c[1] x + ... + c[N-1] x^(N-1)) on |x| <= h, ONE DEGREE BELOW the Taylor
// polynomial it replaces at the same 1e-15 level (tools/minimax_exp_f64.py; max |p - e^x| in the comments): one f64 FMA
// less per entry in every tier (three in the last).  MM_LOWP_MINIMAX 0 restores the Taylor tiers.
#ifndef MM_LOWP_MINIMAX
#define MM_LOWP_MINIMAX 1
#endif
struct MMExpMM {
  static constexpr double t0[5] = {1.00000000000000044e+00, 4.99999999964337805e-01, 1.66666666653991380e-01, 4.16671387780754715e-02,
                                   8.33342673612048890e-03};                                           // h = 1/64: 7.8e-16
  static constexpr double t1[6] = {1.00000000000002021e+00, 5.00000000000006217e-01, 1.66666666501118582e-01, 4.16666666339412517e-02,
                                   8.33367240573600759e-03, 1.38894006327464504e-03};                  // 1/32: 9.1e-17
  static constexpr double t2[7] = {1.00000000000000000e+00, 5.00000000000271227e-01, 1.66666666666765689e-01, 4.16666662452447947e-02,
                                   8.33333324931659694e-03, 1.38907499636817516e-03, 1.98438976588123547e-04};   // 1/16: 5.5e-17
  static constexpr double t3[8] = {9.99999999999994227e-01, 4.99999999999998057e-01, 1.66666666671587443e-01, 4.16666666676764191e-02,
                                   8.33333219859749494e-03, 1.38888871922507188e-03, 1.98509570295033318e-04,
                                   2.48131262574843270e-05};                                           // 1/8: 8.3e-17
  static constexpr double t4[9] = {1.00000000000000044e+00, 4.99999999999706179e-01, 1.66666666666557495e-01, 4.16666667123390885e-02,
                                   8.33333334251792034e-03, 1.38888668227472043e-03, 1.98412387042712872e-04, 2.48435978552550916e-05,
                                   2.76035419601197339e-06};                                           // 1/4: 6.1e-16
  static constexpr double t5[12] = {9.99999999999984124e-01, 4.99999999999993894e-01, 1.66666666667467184e-01, 4.16666666668425861e-02,
                                    8.33333332188715588e-03, 1.38888888712763070e-03, 1.98412768356629729e-04, 2.48015956038952326e-05,
                                    2.75552447171072000e-06, 2.75553088264872583e-07, 2.53470020964661828e-08,
                                    2.11189104196695848e-09};                                          // 3/4: 9.6e-16
};

// e^x, any argument, by the same reduction: 2^k (1 + p).  The diagonal pairs reduce q^_i [D_ij] e^{b_ij} q^'_j with the O(1)
// parts of delta_ij factored into the weights, so b_ij alone can be far below -36 where e^b < 2^-53: expm1(b) + 1 is then
// exactly 0 and the entry -- whose weights are correspondingly large -- was lost (found on the reference's own single-output
// test design: one entry of 3e-10 on a variance of 3.5e-5, tests/golden/refdesign_svgp_so.npz)
__device__ __forceinline__ double mm_exp_f64(double x) {
  x = fmin(x, MM_EXP_CAP_F64);
  const double kf = rint(x * 1.4426950408889634);
  double r = fma(-kf, 6.93147180369123816490e-01, x);
  r = fma(-kf, 1.90821492927058770002e-10, r);
  double q = 2.08767569878681e-09;              // 1/12!
  q = fma(q, r, 2.505210838544172e-08);
  q = fma(q, r, 2.755731922398589e-07);
  q = fma(q, r, 2.7557319223985893e-06);
  q = fma(q, r, 2.48015873015873e-05);
  q = fma(q, r, 1.984126984126984e-04);
  q = fma(q, r, 1.388888888888889e-03);
  q = fma(q, r, 8.333333333333333e-03);
  q = fma(q, r, 4.1666666666666664e-02);
  q = fma(q, r, 1.6666666666666666e-01);
  q = fma(q, r, 0.5);
  q = fma(q, r, 1.0);
  const double s = ldexp(1.0, (int)kf);         // (x << 0: 2^k underflows to 0, the right limit)
  return fma(s, q * r, s);                      // 2^k (1 + expm1(r))
}

// max |x| over the 16 entries of a wave's 2 x 2 accumulator tiles, as the bit pattern of the HIGH DWORD with the sign cleared
// (ordered like |x| for doubles; the callers compare it with the high dword of their power-of-two limits).  Read as an f32
// the high dword keeps that order for every finite double below 2^1017 (sign | the exponent's top 8 bits as the f32 exponent
// | the rest as mantissa; doubles below 2^-1015 may read as 0), so the max runs as ONE v_max3_f32 |a|, |b|, m per entry PAIR
// on two chains -- 10 instructions per batch element and wave where v_and + v_max3_u32 took 24.
// (__double2hiint, not bit_cast<u64>(vec[r]) >> 32: on a vector element the latter compiles to a test of element 0 only with
// this toolchain -- tools/bitcast_repro.hip)
typedef double mm_f64x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned int mm_absmax_hi32(const mm_f64x4_t (&acc)[2][2]) {
  float m0 = 0.0f, m1 = 0.0f;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const mm_f64x4_t cv = acc[rt][ct];
      m0 = fmaxf(fmaxf(m0, fabsf(__int_as_float(__double2hiint(cv[0])))), fabsf(__int_as_float(__double2hiint(cv[1]))));
      m1 = fmaxf(fmaxf(m1, fabsf(__int_as_float(__double2hiint(cv[2])))), fabsf(__int_as_float(__double2hiint(cv[3]))));
    }
  return __float_as_uint(fmaxf(m0, m1));
}
