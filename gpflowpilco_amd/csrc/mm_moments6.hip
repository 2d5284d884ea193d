// Degree-4, -5 and -6 weight moments of the collapsed f32 off-diagonal pairs and their contraction (gfx950, d <= 8).
//
// mm_common.h ("THE MOMENT COLLAPSE"): a collapsed (b, pair) takes p6(x) = x^3 (C0 + C1 x + C2 x^2 + C3 x^3) ~ r(x) on |x| <= 1/4
// from weight moments.  Degrees <= 3 are f64 (mm_moments.hip).  Degrees 5 and 6 contribute <= ~3e-4 of the batch element's
// covariance scale on the items that are collapsed, degree 4 <= 6e-3, so f32 accuracy is ample -- emulated on the BASELINE recipe the
// 2-way split's rounding is <= 1.1e-8 (degrees 5, 6) and <= 4.7e-7 (degree 4) of that scale, where the CUBIC term would cost 1.1e-5
// (it stays f64) -- and an f64 GEMM over their 330 + 792 + 1716 columns (d = 8) would cost 6x the f64 moment GEMM that is left.  Here:
//
//   k_pack_zm56   : the monomials of zc of degree 4, 5 and 6, bf16 2-way split (h, m), monomial-major [L][2][N56p][Mp], and the
//                   contraction's index tables (MMTab56) -- pack time;
//   k_wmom56_gemm : mom56[(b, pair, side)][c] = sum_m what_m zc_m^alpha(c) over the COLLAPSED rows of every latent's GEMM
//                   (k_wmom_perm puts them first): the three products hh + hm + mh on v_mfma_f32_32x32x16_bf16 with f32
//                   accumulation (tools/collapse6_study.py: rounding <= 1.1e-8 of the covariance scale; hh alone would be 1e-5).
//                   128 x 128 output tile per workgroup (4 waves x 64 x 64), K blocks of 32 through a double-buffered LDS image
//                   filled by global_load_lds_dwordx4 (XOR-swizzled on the source side: conflict-free ds_read_b128 fragments);
//                   both operands are read 8 consecutive m per lane (what is [row][m], the table [monomial][m]): no transposes;
//   k_spoly56     : s56[b][po] = C1 <N_4, G^{(x)4} Q_4> + C2 <N_5, G^{(x)5} Q_5> + C3 <N_6, G^{(x)6} Q_6> from the PACKED symmetric moments, G applied one
//                   index at a time on tensors symmetric in the transformed and in the untransformed indices separately
//                   (tools/spoly56_proto.py: 0.54 M FMA per item at d = 8 against 15 M for full tensors), f32, one MM6_THREADS-thread
//                   workgroup per collapsed (b, pair); also estS (mm_common.h: MM_C6_SYS2).
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"
#include "mm_mono.h"
#include "mm_f32_tile.h"
#include <atomic>
#include <type_traits>

__device__ __forceinline__ void mm6_decode_pair_o(int lp, int L, int& a, int& a2) {
  int r = lp, i = 0;
  while (r >= L - 1 - i) { r -= L - 1 - i; ++i; }
  a = i; a2 = i + 1 + r;
}

// ---------------------------------------------------------------------------------------------
// pack time
// ---------------------------------------------------------------------------------------------
// grid (N56p, L), 256 threads: one monomial column of one latent
__global__ __launch_bounds__(256) void k_pack_zm56(char* packed, MMModelLayout lay, int L, int M, int d, int N56p,
                                                   const double* __restrict__ Z) {
  const int c = blockIdx.x, a = blockIdx.y, tid = threadIdx.x;
  const int n4 = mm_binom_i(d + 3, 4), n5 = mm_binom_i(d + 4, 5), n6 = mm_binom_i(d + 5, 6);
  const int off5 = ((n4 + 127) / 128) * 128;            // mm_moment56_off5 / _off6 (d): a 128-column block holds one degree only
  const int off6 = off5 + ((n5 + 127) / 128) * 128;
  const double* zbar = (const double*)(packed + lay.zbar) + (size_t)a * d;
  unsigned short* oh = (unsigned short*)(packed + lay.Zm56) + (((size_t)a * 2 + 0) * N56p + c) * lay.Mp;
  unsigned short* om = (unsigned short*)(packed + lay.Zm56) + (((size_t)a * 2 + 1) * N56p + c) * lay.Mp;
  int n = 0, k[6] = {0, 0, 0, 0, 0, 0};
  if (c < n4) { n = 4; mm_mono_unrank(c, 4, k); }
  else if (c >= off5 && c < off5 + n5) { n = 5; mm_mono_unrank(c - off5, 5, k); }
  else if (c >= off6 && c < off6 + n6) { n = 6; mm_mono_unrank(c - off6, 6, k); }
  for (int m = tid; m < lay.Mp; m += 256) {
    float vf = 0.0f;
    if (n && m < M) {
      double v = 1.0;
      for (int t = 0; t < n; ++t) v *= Z[((size_t)a * M + m) * d + k[t]] - zbar[k[t]];
      vf = (float)v;
    }
    const __bf16 h = (__bf16)vf;
    const __bf16 mid = (__bf16)(vf - (float)h);
    oh[m] = __builtin_bit_cast(unsigned short, h);
    om[m] = __builtin_bit_cast(unsigned short, mid);
  }
}

// one workgroup: the index tables (functions of d alone)
__global__ __launch_bounds__(256) void k_pack_tab56(char* packed, MMModelLayout lay, int d, MMTab56 t) {
  short* i16 = (short*)(packed + lay.tab56);
  float* f32 = (float*)(packed + lay.tab56 + (size_t)t.n_i16 * 2);
  const int tid = threadIdx.x;
  for (int m = 0; m < 6; ++m) {
    const int ns = mm_binom_i(d + m - 1, m);
    for (int idx = tid; idx < ns * 8; idx += 256) {
      const int J = idx >> 3, j = idx & 7;
      int k[6] = {0, 0, 0, 0, 0, 0}, kk[7];
      mm_mono_unrank(J, m, k);
      short r = 0;
      if (j < d) {
        int pos = 0;                                   // insert j keeping the order
        while (pos < m && k[pos] <= j) ++pos;
        for (int u = 0; u < pos; ++u) kk[u] = k[u];
        kk[pos] = j;
        for (int u = pos; u < m; ++u) kk[u + 1] = k[u];
        r = (short)mm_mono_rank(kk, m + 1);
      }
      i16[t.ins[m] + idx] = r;
    }
    for (int I = tid; I < ns; I += 256) {
      int k[6] = {0, 0, 0, 0, 0, 0};
      mm_mono_unrank(I, m, k);
      i16[t.last[m] + I] = (short)(m ? k[m - 1] : 0);
    }
  }
  for (int n = 2; n <= 3; ++n) {
    const int ns = mm_binom_i(d + n - 1, n);
    for (int I = tid; I < ns; I += 256) {
      int k[6] = {0, 0, 0, 0, 0, 0};
      mm_mono_unrank(I, n, k);
      double mult = 1.0;                               // n! / prod over runs of equal indices (run length)!
      int run = 0;
      for (int u = 0; u < n; ++u) {
        mult *= (double)(u + 1);
        run = (u > 0 && k[u] == k[u - 1]) ? run + 1 : 1;
        mult /= (double)run;
      }
      f32[(n == 2 ? t.mult2 : t.mult3) + I] = (float)mult;
    }
  }
}

int mm_launch_pack56(char* packed, const MMModelLayout& lay, int L, int M, int d, const double* Z, hipStream_t s) {
  const int N56p = mm_moment56_cols(d);
  if (N56p <= 0) return 0;
  hipLaunchKernelGGL(k_pack_zm56, dim3(N56p, L), dim3(256), 0, s, packed, lay, L, M, d, N56p, Z);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_pack_tab56, dim3(1), dim3(256), 0, s, packed, lay, d, mm_tab56(d));
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// ---------------------------------------------------------------------------------------------
// k_wmom56_gemm
// ---------------------------------------------------------------------------------------------
#define MM6_TB 128            // output tile: rows (weight vectors) and columns (monomials) per workgroup
#define MM6_KB 32             // m per LDS stage: 64-byte rows
// one stage: [A | B][part (h, m)][128 rows][64 B] = 32 KB; two stages
#define MM6_STAGE_BYTES 32768

__global__ __launch_bounds__(256, 2) void k_wmom56_gemm(const unsigned short* __restrict__ wsp, const unsigned short* __restrict__ Zm56,
                                                        int N56p, int L, int Mp, int B, int Po, int nrb, int ncb, int nwork, int cb5, int cb6,
                                                        const int* __restrict__ gperm, float* __restrict__ mom56) {
  // work item -> (latent, column block, row block), row block fastest: the workgroups that run together on an XCD share a table
  // column block (and, across column blocks, the latent's weight rows)
  const int orig = blockIdx.x;
  const int xcd = orig & 7, slot = orig >> 3;
  const int qn = nwork >> 3, rn = nwork & 7;
  int wi = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  const int rb = wi % nrb; wi /= nrb;
  const int cb = wi % ncb; wi /= ncb;
  const int a = wi;
  const int R = (L - 1) * B;
  // rows of this latent's GEMM that read this column block (k_wmom_perm: they come first): the degree-4 blocks [0, cb5) are read by
  // every collapsed item, the degree-5 blocks [cb5, cb6) by those collapsed to degree >= 5, the degree-6 blocks by those to degree 6
  const int ncoll = gperm[(size_t)L * R + (cb < cb5 ? 0 : (cb < cb6 ? 1 : 2)) * L + a];
  if (rb * MM6_TB >= ncoll) return;
  const int* perm = gperm + (size_t)a * R;
  extern __shared__ __align__(1024) char lds[];        // 2 stages x 32 KB, then 128 ints
  int* orow = reinterpret_cast<int*>(lds + 2 * MM6_STAGE_BYTES);   // per tile row: element offset of its mom56 row, or -1
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  // row r of the latent's GEMM -> (b, pair, side): as k_wmom_gemm
  auto row_item = [&](int r) {
    const int which = r / B, b = r - which * B;
    const int ap = which < a ? which : which + 1;
    const int lo = ap < a ? ap : a, hi = ap < a ? a : ap;
    const int po = lo * (L - 1) - lo * (lo - 1) / 2 + (hi - lo - 1);
    return ((b * Po + po) << 1) | (ap < a ? 1 : 0);
  };
  if (tid < MM6_TB) {
    const int r = rb * MM6_TB + tid;
    orow[tid] = r < ncoll ? row_item(perm[r]) : -1;
  }
  // ---- staging: waves 0, 1 fill the A image (wave = part), waves 2, 3 the B image; 8 global_load_lds_dwordx4 each per stage.
  // One instruction writes 1 KB = 16 rows x 64 B, lane -> (row = lane >> 2, physical 16-B chunk = lane & 3); the chunk holds the
  // logical chunk (lane & 3) ^ ((row >> 2) & 3): the fragment reads below are then conflict-free (MI355X_MICROARCH.md: the four
  // 16-lane groups of a ds_read_b128 would otherwise land on 4 of the 16 slots of a bank row)
  const bool isB = wv >= 2;
  const int part = wv & 1;
  unsigned int src_off[8];                               // element offset of this lane's 8 rows (chunk included) at m = 0
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = i * 16 + (lane >> 2);
    const int lc = (lane & 3) ^ ((row >> 2) & 3);
    if (isB) {
      src_off[i] = (unsigned int)((((size_t)a * 2 + part) * N56p + (size_t)cb * MM6_TB + row) * Mp) + lc * 8;
    } else {
      int r = rb * MM6_TB + row;
      r = r < ncoll ? r : ncoll - 1;                     // rows past the end recompute the last one (not stored)
      src_off[i] = (unsigned int)(((size_t)row_item(perm[r]) * 2 + part) * Mp) + lc * 8;
    }
  }
  const unsigned short* src = isB ? Zm56 : wsp;
  auto stage = [&](int kb, int buf) {
    char* dst = lds + buf * MM6_STAGE_BYTES + (isB ? 16384 : 0) + part * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + src_off[i] + kb * MM6_KB),
                                       (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
  };
  // ---- fragments: wave (wr, wc) owns rows wr * 64 .. + 63, columns wc * 64 .. + 63: 2 x 2 tiles of 32 x 32
  const int wr = wv >> 1, wc = wv & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[rt][ct][e] = 0.0f;
  int fragA[2], fragB[2];                                // byte offset of the lane's row inside a part image (k-step chunk added below)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    fragA[t] = (wr * 64 + t * 32 + l31) * 64;
    fragB[t] = 16384 + (wc * 64 + t * 32 + l31) * 64;
  }
  const int swz = (l31 >> 2) & 3;                        // ((row >> 2) & 3): 64 and 32 are multiples of 16 rows
  const int nkb = Mp / MM6_KB;
  stage(0, 0);
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();                                     // stage kb has landed (vmcnt(0) + barrier); the other buffer has been read
    if (kb + 1 < nkb) stage(kb + 1, (kb + 1) & 1);
    const char* base = lds + (kb & 1) * MM6_STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ((ks * 2 + h) ^ swz) * 16;
      bf16x8 Ah[2], Am[2], Th[2], Tm[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        Ah[t] = *reinterpret_cast<const bf16x8*>(base + fragA[t] + ch);
        Am[t] = *reinterpret_cast<const bf16x8*>(base + 8192 + fragA[t] + ch);
        Th[t] = *reinterpret_cast<const bf16x8*>(base + fragB[t] + ch);
        Tm[t] = *reinterpret_cast<const bf16x8*>(base + 8192 + fragB[t] + ch);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am[rt], Th[ct], acc[rt][ct], 0, 0, 0);
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[rt], Tm[ct], acc[rt][ct], 0, 0, 0);
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[rt], Th[ct], acc[rt][ct], 0, 0, 0);
        }
    }
  }
  // accumulator element e of lane (l31, h): row 8 (e >> 2) + 4 h + (e & 3), column l31
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int item = orow[wr * 64 + rt * 32 + 8 * (e >> 2) + 4 * h + (e & 3)];
      if (item < 0) continue;
      float* o = mom56 + (size_t)item * N56p + (size_t)cb * MM6_TB + wc * 64 + l31;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) o[ct * 32] = acc[rt][ct][e];
    }
}

// ---------------------------------------------------------------------------------------------
// k_spoly56: grid (Po, B), MM6_THREADS threads.  MEET IN THE MIDDLE (tools/spoly56_proto.py: contract_mitm): three indices of
// Q_n are carried to the row side with G,
//     X[I][J] = sum_l G_{i1 l1} G_{i2 l2} G_{i3 l3} Q_{l1 l2 l3 J},          I in sym(3), J in sym(n - 3),
// the other n - 3 indices of N_n to the column side with G^T,  Y[J][I] = sum_m N_{I m} G_{m1 j1} .. ,  and
// <N_n, G^{(x)n} Q_n> = sum_{I, J} mult(I) mult(J) X[I][J] Y[J][I]  (the multinomials are folded into X as it is stored; Y's last
// step is fused with the dot and never stored).  Every step T_{k+1}[I + {i}][J'] = sum_j G[i][j] T_k[I][J' + {j}], i >= max(I), has
// sym(n - k - 1) >= 36 values of J' (d = 8): LANES run over J' -- the eight gather offsets ins[J'][.] are loaded once per step and
// lane -- and the WAVE's I, its largest index and the output row are scalars: no per-lane predicate, no integer division, the
// tuple-rank arithmetic on the scalar unit.  Measured at C3, BASELINE recipe (6400 collapsed items, tools/q_stage_kernels.py): one
// flat loop over (I, J') pairs with per-lane bounds 1.19 ms; this form with G in 64 scalar registers (spilled to VGPR lanes)
// 1.11 ms; G in vector registers, 12 waves 0.88 ms (8 waves 1.03); G output-fastest per side (aligned register pairs for the packed
// FMAs: 90 -> 27 moves per step, 202 -> 138 VGPRs) 0.83 ms -- of which 0.16 ms are the loads and the launch of 7168 workgroups.
// ---------------------------------------------------------------------------------------------
// MODE 0: store T_{k+1}; 1: X's last step (k = 2 -> 3): store mult3(I3) multJ(J') value, row stride xs; 2: Y's last step fused
// with the dot against X (acc += value * X[J'][row])
#ifndef MM6_THREADS
#define MM6_THREADS 768       // 12 waves = 3 per SIMD (138 VGPRs with two chunks per unit).  Measured: 8 waves 0.94 ms, 12 waves 0.83,
#endif                        // 16 waves with one chunk per unit 0.83
#define MM6_WAVES (MM6_THREADS / 64)
// NC: chunks of 64 values of J' per work unit (lane handles J' = c 64 NC + lane + 64 q, q < NC): the unit's scalar work -- its
// decode, the tuple's largest index, the per-i branches and row offsets -- is shared by the NC chunks, and their FMA chains
// interleave (measured: the pass is bound by instructions per unit, ~180 of which ~32 are the packed FMAs)
#ifndef MM6_NC
#define MM6_NC 2
#endif
template <int K, bool TRANSG, int MODE, int NC = MM6_NC, int NW = MM6_WAVES>
__device__ __forceinline__ void mm6_step(const float* __restrict__ Tin, int nJin, int nI, float* __restrict__ Tout, int nJp,
                                         const short* __restrict__ ins, int d, const float (&G)[64], const float* __restrict__ multJ,
                                         int xs, float& acc, int wave, int lane) {
  const int chunks = (nJp + 64 * NC - 1) / (64 * NC);
  // the outputs i = I0 .. 7 of one (I, J'): eight-term dot products of the gathered v with the matrix the caller loaded into G as
  // G[j * 8 + i] = (coefficient of v_j in output i) -- output index fastest, so that the packed FMA over an output pair (i, i + 1)
  // reads an aligned register pair (row-major, the X side needed a v_mov pair per packed FMA: 90 of them per step) -- the chains
  // interleaved (one chain per i inside its own branch ran at the FMA's latency, not its issue rate)
  auto dots = [&](auto i0c, const float (&v)[NC][8], float (&s)[NC][8]) __attribute__((always_inline)) {
    constexpr int I0 = decltype(i0c)::value;
#pragma unroll
    for (int q = 0; q < NC; ++q)
#pragma unroll
      for (int i = I0; i < 8; ++i) s[q][i] = G[0 * 8 + i] * v[q][0];
#pragma unroll
    for (int j = 1; j < 8; ++j)
#pragma unroll
      for (int q = 0; q < NC; ++q)
#pragma unroll
        for (int i = I0; i < 8; ++i) s[q][i] = fmaf(G[j * 8 + i], v[q][j], s[q][i]);
  };
  // lanes past the end of J' (of chunk q) gather a valid (clamped) entry and store / add nothing
  auto body = [&](int I, int t, const int (&Jp)[NC], const bool (&jv)[NC], const int (&off)[NC][8], const float (&mj)[NC]) __attribute__((always_inline)) {
    const float* tin = Tin + I * nJin;
    float v[NC][8], s[NC][8];
#pragma unroll
    for (int q = 0; q < NC; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[q][j] = tin[off[q][j]];
    // only i >= t (the largest index of I) is needed; t is wave-uniform: three variants of the product
    if (t < 4) dots(std::integral_constant<int, 0>{}, v, s);
    else if (t < 6) dots(std::integral_constant<int, 4>{}, v, s);
    else dots(std::integral_constant<int, 6>{}, v, s);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i >= t && i < d) {                             // scalar compare: t, d are wave-uniform
        const int orow = I + mm_binom_i(i + K, K + 1);   // rank of I with i appended (colex)
        float m3 = 1.0f;
        if constexpr (MODE == 1 && K == 2) {
          // I = (a, t) with a = I - C(t + 1, 2): multinomial of the sorted triple (a, t, i)
          const int a = I - ((t * (t + 1)) >> 1);
          m3 = a == t ? (i == t ? 1.0f : 3.0f) : (i == t ? 3.0f : 6.0f);
        } else if constexpr (MODE == 1 && K == 1) {
          m3 = i == t ? 1.0f : 2.0f;                     // I = (t): multinomial of the sorted pair (t, i)
        }
#pragma unroll
        for (int q = 0; q < NC; ++q) {
          if constexpr (MODE == 0) { if (jv[q]) Tout[orow * nJp + Jp[q]] = s[q][i]; }
          else if constexpr (MODE == 1) { if (jv[q]) Tout[orow * xs + Jp[q]] = (m3 * mj[q]) * s[q][i]; }
          else { if (jv[q]) acc = fmaf(s[q][i], Tout[Jp[q] * xs + orow], acc); }   // (Tout = X here)
        }
      }
    }
  };
  // work unit = (chunk group c of 64 NC values of J', I), I fastest: the waves take units round robin (the group's gather offsets
  // are reloaded when c changes: one 16-byte load per chunk)
  const int nunit = chunks * nI;
  const float rnI = 1.0f / (float)nI;
  int cprev = -1, off[NC][8], Jc[NC];
  bool jv[NC];
  float mj[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) { Jc[q] = 0; jv[q] = false; mj[q] = 0.0f; }
  for (int u = wave; u < nunit; u += NW) {
    int c = K == 0 ? u : (int)(((float)u + 0.5f) * rnI);   // u / nI (exact: u < 2^20)
    c = __builtin_amdgcn_readfirstlane(c);
    const int I = K == 0 ? 0 : __builtin_amdgcn_readfirstlane(u - c * nI);
    if (c != cprev) {
      cprev = c;
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const int Jp = (c * NC + q) * 64 + lane;
        jv[q] = Jp < nJp;
        Jc[q] = jv[q] ? Jp : nJp - 1;
        const short4 ia = *reinterpret_cast<const short4*>(ins + (size_t)Jc[q] * 8);
        const short4 ib = *reinterpret_cast<const short4*>(ins + (size_t)Jc[q] * 8 + 4);
        off[q][0] = ia.x; off[q][1] = ia.y; off[q][2] = ia.z; off[q][3] = ia.w;
        off[q][4] = ib.x; off[q][5] = ib.y; off[q][6] = ib.z; off[q][7] = ib.w;
        if constexpr (MODE == 1) mj[q] = multJ[Jc[q]];
      }
    }
    // largest index of the K-tuple of rank I: the tuples whose largest index is t have ranks [C(t + K - 1, K), C(t + K, K))
    int t = 0;
    if constexpr (K > 0) {
#pragma unroll
      for (int q = 1; q < 8; ++q) t += (I >= mm_binom_i(q + K - 1, K)) ? 1 : 0;
    }
    if (jv[0]) body(I, t, Jc, jv, off, mj);              // (chunk 0 of a group is never wholly past the end: jv[0] is a lane mask)
  }
}

// Work lists by collapse degree (MMWorkspaceLayout::ilist): the contractions run over the items that need them, the degree-6/5 ones
// in workgroups of MM6_THREADS threads with 160 KB of LDS (one per CU), the degree-4 ones -- every item of the pilco recipe --
// in workgroups of 256 threads with 13 KB (k_spoly4): as ONE grid over all (b, pair) a degree-4 item cost a 768-thread, whole-CU
// workgroup of its own (pilco recipe: 0.53 ms for 0.03 ms of arithmetic).  An item without a collapsed row group gets its zeros here.
__global__ __launch_bounds__(256) void k_item_classes(const unsigned int* __restrict__ amaxc, const double* __restrict__ zmax2,
                                                      int L, int Po, int n, int allow, int* __restrict__ ilist,
                                                      double* __restrict__ s56, float* __restrict__ estS) {
  const int idx = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  int cls = 3;
  if (idx < n) {
    int a, a2;
    mm6_decode_pair_o(idx % Po, L, a, a2);
    const unsigned int ac = amaxc[idx];
    const float x2 = mm_collapse_bound2(ac, zmax2[a2]);
    cls = (!allow || !mm_item_collapsed(ac)) ? 3 : (x2 > MM_C6_X5_2 ? 0 : (x2 > MM_C6_X4_2 ? 1 : 2));
    if (cls == 3) { s56[idx] = 0.0; estS[idx] = 0.0f; }
  }
  // slots: class 0 and 1 items from the front of two separate runs is not possible without the totals, so: class 0/1 items are
  // appended at the FRONT cursor (count[0] + count[1] grows; the kernel reads each item's class again from its bound), class 2
  // items at the BACK cursor
  const unsigned long long b01 = __ballot(cls == 0 || cls == 1), b2 = __ballot(cls == 2);
  int base01 = 0, base2 = 0;
  if (lane == 0) {
    if (b01) base01 = atomicAdd(ilist + 0, (int)__popcll(b01));
    if (b2) base2 = atomicAdd(ilist + 2, (int)__popcll(b2));
  }
  base01 = __shfl(base01, 0, 64); base2 = __shfl(base2, 0, 64);
  const unsigned long long below = (1ull << lane) - 1ull;
  if (cls == 0 || cls == 1) ilist[4 + base01 + __popcll(b01 & below)] = idx;
  if (cls == 2) ilist[4 + n - 1 - (base2 + __popcll(b2 & below))] = idx;
}

// what the skipped tiles of a collapsed item leave out, in the units of the sweep's error estimate (mm_common.h: MM_C6_SYS2):
// |p6 - r| equioscillates with amplitude 5.8e-10 on [-1/4, 1/4]; near 0 it is the perturbation of the leading coefficient,
// (1/6 - C0) |x|^3 = 3.4e-7 |x|^3: an item whose bound X is far inside 1/4 leaves out (4 X)^3 of the amplitude; an item that
// leaves degree 6 (and 5) out adds C3 X^6 (+ C2 X^5) per entry, with the same cancellation under the weights (a smooth function
// of b): 1e-10 / 5.8e-10 of the amplitude, as for p6 itself
__device__ __forceinline__ float mm6_estS(float bound2, bool need5, bool need6, double r2, double c2) {
  const double X = sqrt((double)bound2), X3 = X * X * X;
  double amp = fmin(1.0, 64.0 * X3);                                                 // in units of 1e-10
  if (!need6) amp += (double)MM_C6_C3 * X3 * X3 / 5.8e-10;
  if (!need5) amp += (double)MM_C6_C2 * X3 * X * X / 5.8e-10;
  return (float)fmin((double)MM_C6_SYS2 * (amp * amp) * r2 * c2, 3.0e38);
}

// grid: persistent over the degree-6 / degree-5 items of ilist (any size; one workgroup per CU fits)
__global__ __launch_bounds__(MM6_THREADS) void k_spoly56(const float* __restrict__ mom56, int N56p, const double* __restrict__ pairmat,
                                                 const double* __restrict__ zmax2, const unsigned int* __restrict__ amaxc,
                                                 const unsigned char* __restrict__ gflag,
                                                 const double* __restrict__ whR, const double* __restrict__ whC,
                                                 const char* __restrict__ tab, MMTab56 tb, int L, int d, int P, int Mp,
                                                 const int* __restrict__ ilist,
                                                 double* __restrict__ s56, float* __restrict__ estS) {
  extern __shared__ __align__(16) float sm6[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int Po = P - L;
  const int nitems = __builtin_amdgcn_readfirstlane(ilist[0]);
  for (int li = blockIdx.x; li < nitems; li += gridDim.x) {
  const size_t item = (size_t)__builtin_amdgcn_readfirstlane(ilist[4 + li]);
  const int b = (int)(item / Po), po = (int)(item - (size_t)b * Po);
  int a, a2;
  mm6_decode_pair_o(po, L, a, a2);
  // (the bound of the item's COLLAPSED row groups -- mm_mono.h: the moments carry their rows alone)
  const float bound2 = mm_collapse_bound2(amaxc[item], zmax2[a2]);
  int sy[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) sy[k] = mm_binom_i(d + k - 1, k);
  // LDS (floats): nq [2][sy5 + sy6] | X [sy3][sy3 + 1] | A [sy2 sy4] | Bf [sy1 sy5] | red
  const int n56 = sy[4] + sy[5] + sy[6];                // (degrees 4, 5, 6 of one side)
  const int off5 = ((sy[4] + 127) / 128) * 128;         // mm_moment56_off5 / _off6 (d): first columns of the degree-5 / -6 blocks in mom56
  const int off6 = off5 + ((sy[5] + 127) / 128) * 128;
  const bool need5 = bound2 > MM_C6_X4_2, need6 = bound2 > MM_C6_X5_2;   // (mm_common.h: what the item's bound leaves negligible)
  const int xs = sy[3] + 1;                              // X's row stride: odd at d = 8 (121): the fused dot reads it along a column
  float* nq = sm6;
  float* X = nq + 2 * n56;
  float* A = X + sy[3] * xs;
  float* Bf = A + sy[2] * sy[4];
  float* red = sm6 + ((2 * n56 + sy[3] * xs + sy[2] * sy[4] + sy[1] * sy[5] + 1) & ~1);   // [MM6_WAVES][3] doubles (8-byte aligned)
  for (int idx = tid; idx < 2 * n56; idx += MM6_THREADS) {
    const int side = idx >= n56, c = idx - side * n56;
    const int dg = c < sy[4] ? 4 : (c < sy[4] + sy[5] ? 5 : 6);
    const int col = dg == 4 ? c : (dg == 5 ? off5 + (c - sy[4]) : off6 + (c - sy[4] - sy[5]));
    float v = 0.0f;
    if (dg == 4 || (dg == 5 ? need5 : need6)) v = mom56[(item * 2 + side) * N56p + col];
    nq[idx] = v;
  }
  // G in VECTOR registers (through LDS): as 64 scalar registers beside the loop's own scalars it spilled to VGPR lanes
  // (780 v_readlane / v_writelane in the kernel, 47 per work unit)
  float* Gl = sm6 + ((2 * n56 + sy[3] * xs + sy[2] * sy[4] + sy[1] * sy[5] + 6 * MM6_WAVES + 2 + 3) & ~3);   // [128], 16-byte aligned
  float* Bf4 = Gl + 128;                                 // n = 4: T_1 [sym1][sym3]
  float* X4 = Bf4 + sy[1] * sy[3];                       //        X [sym2][sym2 + 1]
  if (tid < 64) {
    const int i = tid >> 3, j = tid & 7;
    const double* pm = pairmat + ((size_t)b * P + (L + po)) * (d * d + 1);
    const float g = (i < d && j < d) ? (float)pm[i * d + j] : 0.0f;
    Gl[tid] = g;                                         // row-major: the Y side's (G^T applied: output j' takes G[m][j'])
    Gl[64 + j * 8 + i] = g;                              // transposed: the X side's (output i takes G[i][j])
  }
  // what the skipped tiles leave out, in the units of the sweep's error estimate (mm_common.h: MM_C6_SYS2)
  double s2r = 0.0, s2c = 0.0;
  {
    const double* hr = whR + item * Mp;
    const double* hc = whC + item * Mp;
    const unsigned char* gf = gflag + item * (size_t)(Mp / MM_GROUP_ROWS);
    for (int m = tid; m < Mp; m += MM6_THREADS) {
      const double x = gf[m >> 6] ? hr[m] : 0.0, y = hc[m];     // (the rows of the collapsed groups)
      s2r = fma(x, x, s2r); s2c = fma(y, y, s2c);
    }
  }
  __syncthreads();
  float G[64];
  auto load_G = [&](int transposed) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float4 g4 = *reinterpret_cast<const float4*>(Gl + 64 * transposed + 4 * q);
      G[4 * q] = g4.x; G[4 * q + 1] = g4.y; G[4 * q + 2] = g4.z; G[4 * q + 3] = g4.w;
    }
  };
  const short* tabi = (const short*)tab;
  const float* mult2 = (const float*)(tab + (size_t)tb.n_i16 * 2) + tb.mult2;
  const float* mult3 = (const float*)(tab + (size_t)tb.n_i16 * 2) + tb.mult3;
  float acc5 = 0.0f, acc6 = 0.0f, dummy = 0.0f;
  // ---- n = 6: X = three indices of Q_6 (column side of nq) through G; Y = three indices of N_6 through G^T, fused dot
  if (need6) {
    const float* Q6 = nq + n56 + sy[4] + sy[5];
    const float* N6 = nq + sy[4] + sy[5];
    load_G(1);
    mm6_step<0, false, 0>(Q6, sy[6], 1, Bf, sy[5], tabi + tb.ins[5], d, G, nullptr, 0, dummy, wave, lane);
    __syncthreads();
    mm6_step<1, false, 0>(Bf, sy[5], sy[1], A, sy[4], tabi + tb.ins[4], d, G, nullptr, 0, dummy, wave, lane);
    __syncthreads();
    mm6_step<2, false, 1>(A, sy[4], sy[2], X, sy[3], tabi + tb.ins[3], d, G, mult3, xs, dummy, wave, lane);
    __syncthreads();
    load_G(0);
    mm6_step<0, true, 0>(N6, sy[6], 1, Bf, sy[5], tabi + tb.ins[5], d, G, nullptr, 0, dummy, wave, lane);
    __syncthreads();
    mm6_step<1, true, 0>(Bf, sy[5], sy[1], A, sy[4], tabi + tb.ins[4], d, G, nullptr, 0, dummy, wave, lane);
    __syncthreads();
    mm6_step<2, true, 2>(A, sy[4], sy[2], X, sy[3], tabi + tb.ins[3], d, G, nullptr, xs, acc6, wave, lane);
    __syncthreads();
  }
  // ---- n = 5 (X [sym3][sym2] = three indices of Q_5 through G; Y = two indices of N_5 through G^T, fused dot) and n = 4 (every
  // collapsed item: X4 [sym2][sym2] = two indices of Q_4; Y = two indices of N_4) in the SAME barrier phases: the n = 4 steps are
  // 1 / 8 / 1 / 8 work units on buffers of their own (Bf4, X4: 9 KB) -- as a pass of its own they cost four nearly empty rounds
  // per item (0.15 ms at C3); merged, a few waves take one more unit per phase
  float acc4 = 0.0f;
  {
    const float* Q5 = nq + n56 + sy[4];
    const float* N5 = nq + sy[4];
    const float* Q4 = nq + n56;
    const float* N4 = nq;
    const int xs4 = sy[2] + 1;
    load_G(1);
    if (need5) mm6_step<0, false, 0>(Q5, sy[5], 1, Bf, sy[4], tabi + tb.ins[4], d, G, nullptr, 0, dummy, wave, lane);
    mm6_step<0, false, 0>(Q4, sy[4], 1, Bf4, sy[3], tabi + tb.ins[3], d, G, nullptr, 0, dummy, wave, lane);
    __syncthreads();
    if (need5) mm6_step<1, false, 0>(Bf, sy[4], sy[1], A, sy[3], tabi + tb.ins[3], d, G, nullptr, 0, dummy, wave, lane);
    mm6_step<1, false, 1, 1>(Bf4, sy[3], sy[1], X4, sy[2], tabi + tb.ins[2], d, G, mult2, xs4, dummy, wave, lane);   // (J' = sym2 <= 36: one chunk)
    __syncthreads();
    if (need5) {
      mm6_step<2, false, 1, 1>(A, sy[3], sy[2], X, sy[2], tabi + tb.ins[2], d, G, mult2, xs, dummy, wave, lane);
      __syncthreads();
    }
    load_G(0);
    if (need5) mm6_step<0, true, 0>(N5, sy[5], 1, Bf, sy[4], tabi + tb.ins[4], d, G, nullptr, 0, dummy, wave, lane);
    mm6_step<0, true, 0>(N4, sy[4], 1, Bf4, sy[3], tabi + tb.ins[3], d, G, nullptr, 0, dummy, wave, lane);
    __syncthreads();
    if (need5) mm6_step<1, true, 2>(Bf, sy[4], sy[1], X, sy[3], tabi + tb.ins[3], d, G, nullptr, xs, acc5, wave, lane);
    mm6_step<1, true, 2, 1>(Bf4, sy[3], sy[1], X4, sy[2], tabi + tb.ins[2], d, G, nullptr, xs4, acc4, wave, lane);
  }
  double tot = (double)MM_C6_C1 * (double)acc4 + (double)MM_C6_C2 * (double)acc5 + (double)MM_C6_C3 * (double)acc6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    tot += __shfl_down(tot, off, 64); s2r += __shfl_down(s2r, off, 64); s2c += __shfl_down(s2c, off, 64);
  }
  double* redd = reinterpret_cast<double*>(red);         // [MM6_WAVES][3]
  if (lane == 0) { redd[wave * 3 + 0] = tot; redd[wave * 3 + 1] = s2r; redd[wave * 3 + 2] = s2c; }
  __syncthreads();
  if (tid == 0) {
    double t = 0.0, r2 = 0.0, c2 = 0.0;
    for (int w = 0; w < MM6_WAVES; ++w) { t += redd[w * 3]; r2 += redd[w * 3 + 1]; c2 += redd[w * 3 + 2]; }
    s56[item] = t;
    estS[item] = mm6_estS(bound2, need5, need6, r2, c2);
  }
  __syncthreads();                                       // (the next item of this workgroup reuses the LDS image)
  }
}

// ---------------------------------------------------------------------------------------------
// k_spoly4: the degree-4 items (bound X <= 1/40: orders 5 and 6 are below p6's own error) -- s56 = C1 <N_4, G^{(x)4} Q_4>, the n = 4
// steps of k_spoly56 alone: 256 threads, 13 KB of LDS, persistent over the back part of ilist
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_spoly4(const float* __restrict__ mom56, int N56p, const double* __restrict__ pairmat,
                                                const double* __restrict__ zmax2, const unsigned int* __restrict__ amaxc,
                                                const unsigned char* __restrict__ gflag,
                                                const double* __restrict__ whR, const double* __restrict__ whC,
                                                const char* __restrict__ tab, MMTab56 tb, int L, int d, int P, int Mp, int ntotal,
                                                const int* __restrict__ ilist,
                                                double* __restrict__ s56, float* __restrict__ estS) {
  extern __shared__ __align__(16) float sm4[];
  constexpr int NW = 4;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int Po = P - L;
  int sy[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) sy[k] = mm_binom_i(d + k - 1, k);
  const int xs4 = sy[2] + 1;
  // LDS (floats): nq [2][sy4] | Bf4 [sy1][sy3] | X4 [sy2][sy2 + 1] | Gl [128] (16-byte aligned) | red [NW][3] doubles
  float* nq = sm4;
  float* Bf4 = nq + 2 * sy[4];
  float* X4 = Bf4 + sy[1] * sy[3];
  float* Gl = sm4 + ((2 * sy[4] + sy[1] * sy[3] + sy[2] * xs4 + 3) & ~3);
  double* redd = reinterpret_cast<double*>(Gl + 128);
  const short* tabi = (const short*)tab;
  const float* mult2 = (const float*)(tab + (size_t)tb.n_i16 * 2) + tb.mult2;
  const int nitems = __builtin_amdgcn_readfirstlane(ilist[2]);
  for (int li = blockIdx.x; li < nitems; li += gridDim.x) {
    const size_t item = (size_t)__builtin_amdgcn_readfirstlane(ilist[4 + ntotal - 1 - li]);
    const int b = (int)(item / Po), po = (int)(item - (size_t)b * Po);
    int a, a2;
    mm6_decode_pair_o(po, L, a, a2);
    const float bound2 = mm_collapse_bound2(amaxc[item], zmax2[a2]);
    for (int idx = tid; idx < 2 * sy[4]; idx += 256) {
      const int side = idx >= sy[4], c = idx - side * sy[4];
      nq[idx] = mom56[(item * 2 + side) * N56p + c];       // (the degree-4 block: the first columns of mom56)
    }
    if (tid < 64) {
      const int i = tid >> 3, j = tid & 7;
      const double* pm = pairmat + ((size_t)b * P + (L + po)) * (d * d + 1);
      const float g = (i < d && j < d) ? (float)pm[i * d + j] : 0.0f;
      Gl[tid] = g;                                         // row-major: the Y side's
      Gl[64 + j * 8 + i] = g;                              // transposed: the X side's
    }
    double s2r = 0.0, s2c = 0.0;
    {
      const double* hr = whR + item * Mp;
      const double* hc = whC + item * Mp;
      const unsigned char* gf = gflag + item * (size_t)(Mp / MM_GROUP_ROWS);
      for (int m = tid; m < Mp; m += 256) {
        const double x = gf[m >> 6] ? hr[m] : 0.0, y = hc[m];
        s2r = fma(x, x, s2r); s2c = fma(y, y, s2c);
      }
    }
    __syncthreads();
    float G[64];
    auto load_G = [&](int transposed) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 g4 = *reinterpret_cast<const float4*>(Gl + 64 * transposed + 4 * q);
        G[4 * q] = g4.x; G[4 * q + 1] = g4.y; G[4 * q + 2] = g4.z; G[4 * q + 3] = g4.w;
      }
    };
    float acc4 = 0.0f, dummy = 0.0f;
    const float* Q4 = nq + sy[4];
    const float* N4 = nq;
    load_G(1);
    mm6_step<0, false, 0, MM6_NC, NW>(Q4, sy[4], 1, Bf4, sy[3], tabi + tb.ins[3], d, G, nullptr, 0, dummy, wave, lane);
    __syncthreads();
    mm6_step<1, false, 1, 1, NW>(Bf4, sy[3], sy[1], X4, sy[2], tabi + tb.ins[2], d, G, mult2, xs4, dummy, wave, lane);
    __syncthreads();
    load_G(0);
    mm6_step<0, true, 0, MM6_NC, NW>(N4, sy[4], 1, Bf4, sy[3], tabi + tb.ins[3], d, G, nullptr, 0, dummy, wave, lane);
    __syncthreads();
    mm6_step<1, true, 2, 1, NW>(Bf4, sy[3], sy[1], X4, sy[2], tabi + tb.ins[2], d, G, nullptr, xs4, acc4, wave, lane);
    double tot = (double)MM_C6_C1 * (double)acc4;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      tot += __shfl_down(tot, off, 64); s2r += __shfl_down(s2r, off, 64); s2c += __shfl_down(s2c, off, 64);
    }
    if (lane == 0) { redd[wave * 3 + 0] = tot; redd[wave * 3 + 1] = s2r; redd[wave * 3 + 2] = s2c; }
    __syncthreads();
    if (tid == 0) {
      double t = 0.0, r2 = 0.0, c2 = 0.0;
      for (int w = 0; w < NW; ++w) { t += redd[w * 3]; r2 += redd[w * 3 + 1]; c2 += redd[w * 3 + 2]; }
      s56[item] = t;
      estS[item] = mm6_estS(bound2, false, false, r2, c2);
    }
    __syncthreads();
  }
}

// Degree-5/6 moments + contraction for the collapsed items of the last q stage (after k_wmom_perm on the same stream).
// allow == 0 (forced worst tier): nothing is collapsed; s56 / estS are zeroed.
int mm_launch_moments56(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl,
                        int B, int L, int d, int allow, hipStream_t stream) {
  const int N56p = mm_moment56_cols(d);
  if (N56p <= 0 || wl.Po <= 0) return 0;
  if (allow) {
    const int nrb = ((L - 1) * B + MM6_TB - 1) / MM6_TB, ncb = N56p / MM6_TB;
    const long long nwork_ll = (long long)L * ncb * nrb;
    if (nwork_ll <= 0 || nwork_ll > 0x7fffffffLL) return MM_E_DIM;
    // (offsets inside wsp / Zm56 are 32-bit element counts)
    if ((size_t)B * wl.Po * 4 * wl.Mp >= 0xffffffffull || (size_t)L * 2 * N56p * wl.Mp >= 0xffffffffull) return MM_E_DIM;
    const size_t shm = 2 * MM6_STAGE_BYTES + MM6_TB * sizeof(int);
    // (raise the dynamic-LDS limit once per device: the call is slow)
    static std::atomic<unsigned long long> attr_done{0ull};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 64;
    const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    if (!bit || !(attr_done.load() & bit)) {
      const hipError_t ea = hipFuncSetAttribute((const void*)k_wmom56_gemm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      if (ea != hipSuccess) return (int)ea;
      attr_done.fetch_or(bit);
    }
    hipLaunchKernelGGL(k_wmom56_gemm, dim3((int)nwork_ll), dim3(256), shm, stream, (const unsigned short*)(ws + wl.wsp),
                       (const unsigned short*)(packed + ml.Zm56), N56p, L, wl.Mp, B, wl.Po, nrb, ncb, (int)nwork_ll,
                       mm_moment56_off5(d) / MM6_TB, mm_moment56_off6(d) / MM6_TB,
                       (const int*)(ws + wl.gperm), (float*)(ws + wl.mom56));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  // work lists by collapse degree (k_item_classes; the items without a collapsed group get their zeros there)
  const int nitem = B * wl.Po;
  int* ilist = (int*)(ws + wl.ilist);
  {
    const hipError_t em = hipMemsetAsync(ilist, 0, 16, stream);
    if (em != hipSuccess) return (int)em;
  }
  hipLaunchKernelGGL(k_item_classes, dim3((nitem + 255) / 256), dim3(256), 0, stream, (const unsigned int*)(ws + wl.amaxc),
                     (const double*)(packed + ml.zmax2), L, wl.Po, nitem, allow, ilist, (double*)(ws + wl.s56), (float*)(ws + wl.estS));
  {
    const hipError_t ec = hipGetLastError();
    if (ec != hipSuccess) return (int)ec;
  }
  if (!allow) return 0;
  int sy[7];
  for (int k = 0; k < 7; ++k) sy[k] = mm_mono_count(k, d);
  size_t nfl = (size_t)2 * (sy[4] + sy[5] + sy[6]) + (size_t)sy[3] * (sy[3] + 1) + (size_t)sy[2] * sy[4] + (size_t)sy[1] * sy[5] + 6 * MM6_WAVES + 128 + 16
               + (size_t)sy[1] * sy[3] + (size_t)sy[2] * (sy[2] + 1);
  nfl = (nfl + 1) & ~(size_t)1;                           // (the f64 reduction scratch behind it stays 8-byte aligned)
  const size_t shm = nfl * sizeof(float);
  static std::atomic<unsigned long long> attr2_done{0ull};
  static std::atomic<int> ncu_cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 64;
  const unsigned long long bit2 = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
  int ncu = bit2 ? ncu_cached[dev].load() : 0;
  if (!bit2 || !(attr2_done.load() & bit2)) {
    const hipError_t ea = hipFuncSetAttribute((const void*)k_spoly56, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (ea != hipSuccess) return (int)ea;
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev < 64 ? dev : 0) != hipSuccess || v <= 0) v = 256;
    ncu = v;
    if (bit2) { ncu_cached[dev].store(v); attr2_done.fetch_or(bit2); }
  }
  if (ncu <= 0) ncu = 256;
  // persistent grids over the lists: one 160 KB workgroup per CU for the degree-6/5 items, eight small ones for the degree-4 items
  const int g56 = nitem < ncu ? nitem : ncu, g4 = nitem < 8 * ncu ? nitem : 8 * ncu;
  hipLaunchKernelGGL(k_spoly56, dim3(g56), dim3(MM6_THREADS), shm, stream, (const float*)(ws + wl.mom56), N56p,
                     (const double*)(ws + wl.pairmat), (const double*)(packed + ml.zmax2), (const unsigned int*)(ws + wl.amaxc),
                     (const unsigned char*)(ws + wl.gflag), (const double*)(ws + wl.whR), (const double*)(ws + wl.whC),
                     packed + ml.tab56, mm_tab56(d), L, d, wl.P, wl.Mp, (const int*)ilist, (double*)(ws + wl.s56), (float*)(ws + wl.estS));
  {
    const hipError_t e5 = hipGetLastError();
    if (e5 != hipSuccess) return (int)e5;
  }
  const size_t shm4 = (size_t)(((2 * sy[4] + sy[1] * sy[3] + sy[2] * (sy[2] + 1) + 3) & ~3) + 128 + 2 * 3 * 4 + 8) * sizeof(float);
  hipLaunchKernelGGL(k_spoly4, dim3(g4), dim3(256), shm4, stream, (const float*)(ws + wl.mom56), N56p,
                     (const double*)(ws + wl.pairmat), (const double*)(packed + ml.zmax2), (const unsigned int*)(ws + wl.amaxc),
                     (const unsigned char*)(ws + wl.gflag), (const double*)(ws + wl.whR), (const double*)(ws + wl.whC),
                     packed + ml.tab56, mm_tab56(d), L, d, wl.P, wl.Mp, nitem, (const int*)ilist, (double*)(ws + wl.s56),
                     (float*)(ws + wl.estS));
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
