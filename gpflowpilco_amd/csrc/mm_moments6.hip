// Degree-5 and degree-6 weight moments of the collapsed f32 off-diagonal pairs and their contraction (gfx950, d <= 8).
//
// mm_common.h ("THE MOMENT COLLAPSE"): a collapsed (b, pair) takes p6(x) = x^3 (C0 + C1 x + C2 x^2 + C3 x^3) ~ r(x) on |x| <= 1/4
// from weight moments.  Degrees <= 4 are f64 (mm_moments.hip).  Degrees 5 and 6 contribute <= ~3e-4 of the batch element's
// covariance scale on the items that are collapsed, so f32 accuracy is ample -- and an f64 GEMM over their 792 + 1716 columns
// (d = 8) would cost 5x the whole f64 moment GEMM.  Here:
//
//   k_pack_zm56   : the monomials of zc of degree 5 and 6, bf16 2-way split (h, m), monomial-major [L][2][N56p][Mp], and the
//                   contraction's index tables (MMTab56) -- pack time;
//   k_wmom56_gemm : mom56[(b, pair, side)][c] = sum_m what_m zc_m^alpha(c) over the COLLAPSED rows of every latent's GEMM
//                   (k_wmom_perm puts them first): the three products hh + hm + mh on v_mfma_f32_32x32x16_bf16 with f32
//                   accumulation (tools/collapse6_study.py: rounding <= 1.1e-8 of the covariance scale; hh alone would be 1e-5).
//                   128 x 128 output tile per workgroup (4 waves x 64 x 64), K blocks of 32 through a double-buffered LDS image
//                   filled by global_load_lds_dwordx4 (XOR-swizzled on the source side: conflict-free ds_read_b128 fragments);
//                   both operands are read 8 consecutive m per lane (what is [row][m], the table [monomial][m]): no transposes;
//   k_spoly56     : s56[b][po] = C2 <N_5, G^{(x)5} Q_5> + C3 <N_6, G^{(x)6} Q_6> from the PACKED symmetric moments, G applied one
//                   index at a time on tensors symmetric in the transformed and in the untransformed indices separately
//                   (tools/spoly56_proto.py: 0.54 M FMA per item at d = 8 against 15 M for full tensors), f32, one 512-thread
//                   workgroup per collapsed (b, pair); also estS (mm_common.h: MM_C6_SYS2).
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"
#include "mm_mono.h"
#include "mm_f32_tile.h"

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
  const int n5 = mm_binom_i(d + 4, 5), n6 = mm_binom_i(d + 5, 6);
  const double* zbar = (const double*)(packed + lay.zbar) + (size_t)a * d;
  unsigned short* oh = (unsigned short*)(packed + lay.Zm56) + (((size_t)a * 2 + 0) * N56p + c) * lay.Mp;
  unsigned short* om = (unsigned short*)(packed + lay.Zm56) + (((size_t)a * 2 + 1) * N56p + c) * lay.Mp;
  int n = 0, k[6] = {0, 0, 0, 0, 0, 0};
  if (c < n5) { n = 5; mm_mono_unrank(c, 5, k); }
  else if (c < n5 + n6) { n = 6; mm_mono_unrank(c - n5, 6, k); }
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
  for (int n = 5; n <= 6; ++n) {
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
      f32[(n == 5 ? t.mult5 : t.mult6) + I] = (float)mult;
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
                                                        int N56p, int L, int Mp, int B, int Po, int nrb, int ncb, int nwork,
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
  const int ncoll = gperm[(size_t)L * R + a];          // collapsed rows of this latent's GEMM (k_wmom_perm: they come first)
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
// k_spoly56: grid (Po, B), 512 threads.  LDS: two tensor buffers (even / odd stages), sized for n = 6.
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ float mm6_contract(const float* __restrict__ nq, int offn, int d, const int (&sy)[7],
                                              const short* __restrict__ tabi, const float* __restrict__ mult, const MMTab56& tb,
                                              const float (&G)[64], float* bufE, float* bufO, int tid) {
  // T_0 = Q_N (column side: nq[1]), packed
  const float* Nn = nq + offn;
  const float* Qn = nq + sy[5] + sy[6] + offn;           // (the kernel's LDS copy: [row side 5 | 6][column side 5 | 6])
  for (int idx = tid; idx < sy[N]; idx += 512) bufE[idx] = Qn[idx];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float* Tin = (k & 1) ? bufO : bufE;
    float* Tout = (k & 1) ? bufE : bufO;
    const int nI = sy[k], nJ = sy[N - k - 1], nJin = sy[N - k];
    const short* ins = tabi + tb.ins[N - k - 1];
    const short* lastk = tabi + tb.last[k];
    const int npair = nI * nJ;
    for (int q0 = 0; q0 < npair; q0 += 512) {
      const int q = q0 + tid;
      const bool in = q < npair;
      const int qq = in ? q : npair - 1;
      const int I = qq / nJ, J = qq - I * nJ;
      const short4 ia = *reinterpret_cast<const short4*>(ins + (size_t)J * 8);
      const short4 ib = *reinterpret_cast<const short4*>(ins + (size_t)J * 8 + 4);
      const float* tin = Tin + (size_t)I * nJin;
      const float v0 = tin[ia.x], v1 = tin[ia.y], v2 = tin[ia.z], v3 = tin[ia.w];
      const float v4 = tin[ib.x], v5 = tin[ib.y], v6 = tin[ib.z], v7 = tin[ib.w];
      const int lastI = lastk[I];
      // colex rank: the largest index is monotone in the rank, so the first lane of a wave has the smallest `last`
      const int lo = __builtin_amdgcn_readfirstlane(lastI);
      int app = I * nJ + J;                              // + C(i + k, k + 1) nJ for the appended index i
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i >= lo && i < d) {                          // wave-uniform
          float s = G[i * 8 + 0] * v0;
          s = fmaf(G[i * 8 + 1], v1, s); s = fmaf(G[i * 8 + 2], v2, s); s = fmaf(G[i * 8 + 3], v3, s);
          s = fmaf(G[i * 8 + 4], v4, s); s = fmaf(G[i * 8 + 5], v5, s); s = fmaf(G[i * 8 + 6], v6, s);
          s = fmaf(G[i * 8 + 7], v7, s);
          if (in && i >= lastI) Tout[app + mm_binom_i(i + k, k + 1) * nJ] = s;
        }
      }
    }
    __syncthreads();
  }
  const float* Tn = (N & 1) ? bufO : bufE;
  float part = 0.0f;
  for (int idx = tid; idx < sy[N]; idx += 512) part = fmaf(mult[idx] * Nn[idx], Tn[idx], part);
  __syncthreads();                                       // (the next degree overwrites the buffers)
  return part;
}

__global__ __launch_bounds__(512) void k_spoly56(const float* __restrict__ mom56, int N56p, const double* __restrict__ pairmat,
                                                 const double* __restrict__ zmax2, const unsigned int* __restrict__ amax,
                                                 const double* __restrict__ whR, const double* __restrict__ whC,
                                                 const char* __restrict__ tab, MMTab56 tb, int L, int d, int P, int Mp, int allow,
                                                 double* __restrict__ s56, float* __restrict__ estS) {
  extern __shared__ __align__(16) float sm6[];
  const int po = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int Po = P - L;
  int a, a2;
  mm6_decode_pair_o(po, L, a, a2);
  const size_t item = (size_t)b * Po + po;
  const bool coll = allow && mm_collapse_bound2(amax[item], zmax2[a2]) <= MM_COLLAPSE_BOUND2;
  if (!coll) {
    if (tid == 0) { s56[item] = 0.0; estS[item] = 0.0f; }
    return;
  }
  int sy[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) sy[k] = mm_binom_i(d + k - 1, k);
  // LDS: nq [2][sy5 + sy6] | bufE | bufO | red
  const int n56 = sy[5] + sy[6];
  const int szE = sy[2] * sy[4] > sy[6] ? sy[2] * sy[4] : sy[6];      // even stages of n = 6 (T_0, T_2, T_4, T_6) and of n = 5
  const int szO = sy[3] * sy[3];                                       // odd stages (T_1, T_3, T_5)
  float* nq = sm6;
  float* bufE = nq + 2 * n56;
  float* bufO = bufE + szE;
  float* red = bufO + szO;                               // [16]
  for (int idx = tid; idx < 2 * n56; idx += 512) {
    const int side = idx >= n56, c = idx - side * n56;
    nq[idx] = mom56[(item * 2 + side) * N56p + c];
  }
  float G[64];
  {
    const double* pm = pairmat + ((size_t)b * P + (L + po)) * (d * d + 1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = (i < d && j < d) ? (float)pm[(i < d ? i : 0) * d + (j < d ? j : 0)] : 0.0f;
        G[i * 8 + j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, g)));   // uniform: scalar registers
      }
  }
  // what the skipped tiles leave out, in the units of the sweep's error estimate (mm_common.h: MM_C6_SYS2)
  double s2r = 0.0, s2c = 0.0;
  {
    const double* hr = whR + item * Mp;
    const double* hc = whC + item * Mp;
    for (int m = tid; m < Mp; m += 512) { const double x = hr[m], y = hc[m]; s2r = fma(x, x, s2r); s2c = fma(y, y, s2c); }
  }
  __syncthreads();
  const short* tabi = (const short*)tab;
  const float* mult = (const float*)(tab + (size_t)tb.n_i16 * 2);
  const float p5 = mm6_contract<5>(nq, 0, d, sy, tabi, mult + tb.mult5, tb, G, bufE, bufO, tid);
  const float p6 = mm6_contract<6>(nq, sy[5], d, sy, tabi, mult + tb.mult6, tb, G, bufE, bufO, tid);
  double tot = (double)MM_C6_C2 * (double)p5 + (double)MM_C6_C3 * (double)p6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    tot += __shfl_down(tot, off, 64); s2r += __shfl_down(s2r, off, 64); s2c += __shfl_down(s2c, off, 64);
  }
  double* redd = reinterpret_cast<double*>(red);         // [8][3]
  if ((tid & 63) == 0) { redd[(tid >> 6) * 3 + 0] = tot; redd[(tid >> 6) * 3 + 1] = s2r; redd[(tid >> 6) * 3 + 2] = s2c; }
  __syncthreads();
  if (tid == 0) {
    double t = 0.0, r2 = 0.0, c2 = 0.0;
    for (int w = 0; w < 8; ++w) { t += redd[w * 3]; r2 += redd[w * 3 + 1]; c2 += redd[w * 3 + 2]; }
    s56[item] = t;
    estS[item] = (float)fmin((double)MM_C6_SYS2 * r2 * c2, 3.0e38);
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
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr_done[dev]) {
      if (hipFuncSetAttribute((const void*)k_wmom56_gemm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return MM_E_ARG;
      attr_done[dev] = true;
    }
    hipLaunchKernelGGL(k_wmom56_gemm, dim3((int)nwork_ll), dim3(256), shm, stream, (const unsigned short*)(ws + wl.wsp),
                       (const unsigned short*)(packed + ml.Zm56), N56p, L, wl.Mp, B, wl.Po, nrb, ncb, (int)nwork_ll,
                       (const int*)(ws + wl.gperm), (float*)(ws + wl.mom56));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  int sy[7];
  for (int k = 0; k < 7; ++k) sy[k] = mm_mono_count(k, d);
  const int szE = sy[2] * sy[4] > sy[6] ? sy[2] * sy[4] : sy[6], szO = sy[3] * sy[3];
  const size_t shm = (size_t)(2 * (sy[5] + sy[6]) + szE + szO + 64) * sizeof(float);
  static bool attr2_done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !attr2_done[dev]) {
    if (hipFuncSetAttribute((const void*)k_spoly56, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MM_E_ARG;
    attr2_done[dev] = true;
  }
  hipLaunchKernelGGL(k_spoly56, dim3(wl.Po, B), dim3(512), shm, stream, (const float*)(ws + wl.mom56), N56p,
                     (const double*)(ws + wl.pairmat), (const double*)(packed + ml.zmax2), (const unsigned int*)(ws + wl.amax),
                     (const double*)(ws + wl.whR), (const double*)(ws + wl.whC), packed + ml.tab56, mm_tab56(d), L, d, wl.P, wl.Mp,
                     allow, (double*)(ws + wl.s56), (float*)(ws + wl.estS));
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
