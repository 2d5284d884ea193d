// f32 MFMA fused reduce of the off-diagonal kernel pairs (a < a') on gfx950.
//
// For every (b, pair) the kernel evaluates, without ever storing the M x M block Q_aa'
// (the reference's eKuffu slice, utils/kernel_expectation.py:72-187, then
// models.py:219-248),
//     S = sum_ij w_i * expm1(delta_ij) * w'_j ,   delta_ij = rho_i + gamma'_j + zc_i . g_j
// The bilinear part runs on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32 FMA chain,
// one extra k-step carries (rho_i, 1) x (1, gamma'_j)), the expm1 + weighted reduction on the
// VALU, which runs concurrently with the MFMA pipe of the other resident waves.
//
// Work decomposition: workgroup = 4 waves = 256 rows of one (b, pair); a wave owns 64 rows
// (two 32x32 MFMA row tiles, A operands and the 32 row weights stay in registers) and sweeps
// all columns in tiles of 32, software-prefetching the next tile's B operands from the
// k-major colO stream ([k][Mp]: two coalesced 128-B segments per k-step).  Partial sums are
// flushed to f64 once per column tile and written to a slab (no atomics: bitwise
// reproducible).  The 1-D grid is remapped so that the row panels of one (b, pair) -- which
// all stream the same colO block -- land on the same XCD and share its L2.
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Packed f32 FMA (v_pk_fma_f32): gfx950 issues a wave64 VALU instruction over 4 cycles, so the
// f32 vector peak (64 FLOP/clk/SIMD) is only reached with two FMAs per lane per instruction.
__device__ __forceinline__ f32x2 mm_pkfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#define MM_PK(c_) ((f32x2){(c_), (c_)})

__device__ __forceinline__ void mm_decode_pair_o(int p, int L, int& a, int& a2) {
  int r = p - L, i = 0;
  while (r >= L - 1 - i) { r -= L - 1 - i; ++i; }
  a = i; a2 = i + 1 + r;
}

// expm1 on [-1, 1]: Taylor to degree 10 (truncation 1/11! relative, below f32 rounding).
// Returns x * P(x); relative error ~1e-7 of expm1(x) itself (not of 1 + expm1(x)), which is
// what the centred reduce needs (DESIGN.md "fp32 error budget").
__device__ __forceinline__ f32x2 mm_expm1_small2(f32x2 x) {
  f32x2 p = MM_PK(2.7557319e-7f);
  p = mm_pkfma(p, x, MM_PK(2.7557319e-6f));
  p = mm_pkfma(p, x, MM_PK(2.4801587e-5f));
  p = mm_pkfma(p, x, MM_PK(1.9841270e-4f));
  p = mm_pkfma(p, x, MM_PK(1.3888889e-3f));
  p = mm_pkfma(p, x, MM_PK(8.3333333e-3f));
  p = mm_pkfma(p, x, MM_PK(4.1666667e-2f));
  p = mm_pkfma(p, x, MM_PK(1.6666667e-1f));
  p = mm_pkfma(p, x, MM_PK(0.5f));
  p = mm_pkfma(p, x, MM_PK(1.0f));
  return p * x;
}

__device__ __forceinline__ float mm_expm1_small(float x) {
  float p = 2.7557319e-7f;             // 1/10!
  p = fmaf(p, x, 2.7557319e-6f);       // 1/9!
  p = fmaf(p, x, 2.4801587e-5f);       // 1/8!
  p = fmaf(p, x, 1.9841270e-4f);       // 1/7!
  p = fmaf(p, x, 1.3888889e-3f);       // 1/6!
  p = fmaf(p, x, 8.3333333e-3f);       // 1/5!
  p = fmaf(p, x, 4.1666667e-2f);       // 1/4!
  p = fmaf(p, x, 1.6666667e-1f);       // 1/3!
  p = fmaf(p, x, 0.5f);
  p = fmaf(p, x, 1.0f);
  return p * x;
}

template <int KS>
__global__ __launch_bounds__(256, 2) void k_qred_f32_mfma(const float* __restrict__ Zc, int Kz,
                                                          int L, int Mp, int d, int P, int Po, int NS,
                                                          int npanel, int nwork,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ rowO,
                                                          const float* __restrict__ colO,
                                                          double* __restrict__ partB) {
  // XCD-aware remap of the 1-D grid (blocks b and b+8 share an XCD): consecutive work items
  // go to the same XCD.  Bijective for any nwork (cdna guide T1).
  const int orig = blockIdx.x;
  const int xcd = orig & 7, slot = orig >> 3;
  const int qn = nwork >> 3, rn = nwork & 7;
  const int wi = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  const int panel = wi % npanel;
  const int t = wi / npanel;
  const int lp = t % Po, b = t / Po;
  const int p = L + lp;
  int a, a2;
  mm_decode_pair_o(p, L, a, a2);

  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int row0 = panel * MM_PANEL_ROWS + wv * 64;
  double sum = 0.0;
  if (row0 < Mp) {   // Mp % 128 == 0, so a wave's 64 rows are all inside or all outside
    const float* zr = Zc + (size_t)a * Mp * Kz;
    const float* ra = rowO + ((size_t)b * Po + lp) * Mp;
    const float* wr = w + ((size_t)b * L + a) * Mp;
    const float* wc = w + ((size_t)b * L + a2) * Mp;
    const float* cb = colO + ((size_t)b * Po + lp) * (size_t)(d + 1) * Mp;

    float areg[2][KS], ax[2];
    f32x2 wrow[2][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = row0 + rt * 32 + l31;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        // unconditional load at a clamped index, then select (no branch around the load)
        const int k = 2 * s + h;
        const float v = zr[(size_t)row * Kz + (k < Kz ? k : Kz - 1)];
        areg[rt][s] = (k < Kz) ? v : 0.0f;
      }
      const float rv = ra[row];
      ax[rt] = h ? 1.0f : rv;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(wr + row0 + rt * 32 + 8 * g + 4 * h);
        wrow[rt][2 * g + 0] = (f32x2){v.x, v.y};
        wrow[rt][2 * g + 1] = (f32x2){v.z, v.w};
      }
    }

    const int nct = Mp >> 5;
    // B-operand row offsets: k-step s reads row min(2s + h, d) of the k-major colO block.  Rows
    // beyond d - 1 meet a zero A operand (Zc is zero padded), so any finite value is fine there.
    size_t boff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 2 * s + h;
      boff[s] = (size_t)(k < d ? k : d) * Mp;
    }
    const size_t goff = (size_t)d * Mp;
    float bcur[KS], bxc, wcc;
    // prologue: operands of column tile 0
#pragma unroll
    for (int s = 0; s < KS; ++s) bcur[s] = cb[boff[s] + l31];
    {
      const float gv = cb[goff + l31];
      bxc = h ? gv : 1.0f;
    }
    wcc = wc[l31];

    for (int ct = 0; ct < nct; ++ct) {
      // prefetch tile ct + 1 (clamped: the last iteration re-reads its own tile)
      const int cn = ((ct + 1 < nct) ? ct + 1 : ct) * 32 + l31;
      float bnxt[KS], bxn, wcn;
#pragma unroll
      for (int s = 0; s < KS; ++s) bnxt[s] = cb[boff[s] + cn];
      {
        const float gv = cb[goff + cn];
        bxn = h ? gv : 1.0f;
      }
      wcn = wc[cn];

      f32x16 acc[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[rt], bxc, c, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < KS; ++s)
          c = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[rt][s], bcur[s], c, 0, 0, 0);
        acc[rt] = c;
      }
      // range check of the tile (wave-uniform): the polynomial covers |delta| <= 1
      float mx = 0.0f;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(acc[rt][r]));
      f32x2 part2 = {0.0f, 0.0f};
      if (!__any(mx > 1.0f)) {
        // Horner steps run "vertically" over the 16 register pairs so that consecutive
        // v_pk_fma_f32 are independent (a dependent pair costs a wait state + the FMA latency).
        f32x2 xx[16], pp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          xx[r] = (f32x2){acc[r >> 3][2 * (r & 7)], acc[r >> 3][2 * (r & 7) + 1]};
          pp[r] = mm_pkfma(MM_PK(2.7557319e-7f), xx[r], MM_PK(2.7557319e-6f));
        }
#define MM_HORNER_STEP(c_)                                            \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) pp[r] = mm_pkfma(pp[r], xx[r], MM_PK(c_));
        MM_HORNER_STEP(2.4801587e-5f)
        MM_HORNER_STEP(1.9841270e-4f)
        MM_HORNER_STEP(1.3888889e-3f)
        MM_HORNER_STEP(8.3333333e-3f)
        MM_HORNER_STEP(4.1666667e-2f)
        MM_HORNER_STEP(1.6666667e-1f)
        MM_HORNER_STEP(0.5f)
        MM_HORNER_STEP(1.0f)
#undef MM_HORNER_STEP
        f32x2 parts[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const f32x2 wx = wrow[r >> 3][r & 7] * xx[r];          // w_i * x
          parts[r & 3] = mm_pkfma(wx, pp[r], parts[r & 3]);       // += w_i * x * P(x)
        }
        part2 = (parts[0] + parts[1]) + (parts[2] + parts[3]);
      } else {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            // |delta| > 1: exp2 on the transcendental unit (relative error ~ |x| * 6e-8)
            const float x = acc[rt][r];
            const float xs = fminf(fmaxf(x, -1.0f), 1.0f);
            const float big = __builtin_amdgcn_exp2f(x * 1.44269504f) - 1.0f;
            const float e = (fabsf(x) <= 1.0f) ? mm_expm1_small(xs) : big;
            part2[r & 1] = fmaf(wrow[rt][r >> 1][r & 1], e, part2[r & 1]);
          }
      }
      sum += (double)(part2[0] + part2[1]) * (double)wcc;
#pragma unroll
      for (int s = 0; s < KS; ++s) bcur[s] = bnxt[s];
      bxc = bxn; wcc = wcn;
    }
  }
  // workgroup reduction -> slab
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  __shared__ double red[4];
  if (lane == 0) red[wv] = sum;
  __syncthreads();
  if (threadIdx.x == 0) partB[((size_t)b * P + p) * NS + panel] = red[0] + red[1] + red[2] + red[3];
}

extern "C" int mm_mfma_supported(int d) { return d >= 1 && d <= 32; }

int mm_mfma_num_slots(int Mp) { return (Mp + MM_PANEL_ROWS - 1) / MM_PANEL_ROWS; }

int mm_launch_qred_mfma(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl,
                        int B, int L, int d, hipStream_t stream) {
  const int npanel = mm_mfma_num_slots(wl.Mp);
  const long long nwork_ll = (long long)npanel * wl.Po * B;
  if (nwork_ll <= 0 || nwork_ll > 0x7fffffffLL) return MM_E_DIM;
  const int nwork = (int)nwork_ll;
  const float* Zc = (const float*)(packed + ml.Zc);
  const float* w = (const float*)(ws + wl.w);
  const float* rowO = (const float*)(ws + wl.rowO);
  const float* colO = (const float*)(ws + wl.colO);
  double* partB = (double*)(ws + wl.partB);
  const int ks = (d + 1) / 2;
#define MM_LAUNCH_KS(KS_)                                                                         \
  hipLaunchKernelGGL((k_qred_f32_mfma<KS_>), dim3(nwork), dim3(256), 0, stream, Zc, ml.Kz, L,     \
                     wl.Mp, d, wl.P, wl.Po, wl.NS, npanel, nwork, w, rowO, colO, partB)
  if (ks <= 1) MM_LAUNCH_KS(1);
  else if (ks == 2) MM_LAUNCH_KS(2);
  else if (ks == 3) MM_LAUNCH_KS(3);
  else if (ks == 4) MM_LAUNCH_KS(4);
  else if (ks <= 6) MM_LAUNCH_KS(6);
  else if (ks <= 8) MM_LAUNCH_KS(8);
  else if (ks <= 12) MM_LAUNCH_KS(12);
  else MM_LAUNCH_KS(16);
#undef MM_LAUNCH_KS
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
