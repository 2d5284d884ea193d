// Backward of the fused reduce for the OFF-DIAGONAL pairs of an f32 model (d <= 8) on gfx950 -- SURVEY.md row f-1.
//
// The reference differentiates the rollout with tf.GradientTape (gpflow_pilco/utils/optimizers.py:51-56 through
// moment_matching/models.py:200-299).  For a frozen model the M^2-sized part of d(f1, Sff, cross)/d(mu, Sigma) of a pair
// (a, a') is, in moment form (csrc/mm_adjoint.h: mma_gp_item_bwd), a function of the AGGREGATES
//     T[alpha, beta] = sum_ij Omega_ij zeta_i^alpha zc'_j^beta,   |alpha| + |beta| <= 2,   Omega_ij = what_i what'_j e^{b_ij}
// (1 + 2 d + 3 d^2 numbers per (b, pair)); nothing needs the M-sized row / column sums the f64 sweeps (mm_backward.hip)
// produce.  As in the forward (mm_mfma.hip) the polynomial part 1 + b + b^2/2 of e^b is exact from f64 moments of the two
// weight vectors (mma_pair_poly on the tables of k_wmom_gemm, to degree 4), and the REMAINDER r(b) = e^b - 1 - b - b^2/2 is
// reduced here in f32:
//
//   k_bwd_rem_f32: the forward's tile sweep with the operand slots of the bilinear product exchanged, so that the
//     accumulator of v_mfma_f32_32x32x16_bf16 has lane = ROW i (stationary: a wave owns 64 rows) and registers = the 32
//     columns j of the streamed tile.  The weighted remainder tile V_ij = what'_j r(b_ij), split into bf16 (hi, lo), is
//     then directly the A operand of a second product
//         B_i[slot] += sum_j V_ij psi_slot(zc'_j),      psi = (1 | zc'_l | zc'_l zc'_l', l <= l')   (45 of 64 slots at d = 8)
//     against the model-constant pre-split monomial images MMModelLayout::Zq2 (3 bf16 products hi.hi + lo.hi + hi.lo:
//     2^-16 relative), accumulated over the whole column sweep in four MFMA accumulators -- no cross-lane sums, no
//     per-tile flush.  Epilogue once per sweep (f64): T_rem[alpha, slot] = sum_i what_i zeta_i^alpha B_i[slot].
//   k_pair_agg: per (b, pair) the polynomial part (mma_pair_poly), + the panels' remainder slabs, re-centred at mu
//     (mma_pair_convert) -> pagg [B][Po][mma_pair_agg_len(d)], what k_gp_bwd_items consumes.
// The L diagonal pairs (the only ones with the C term) stay on the f64 column sweep of mm_backward.hip in both modes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <atomic>
#include "mm_common.h"
#include "mm_fork.h"
#include "mm_mono.h"
#include "mm_f32_tile.h"
#include "mm_adjoint.h"
#define MMR_WLDS_MAX 16384   // column weights of a pair are staged in LDS up to this many (padded) inducing points
#define MMR_BS 33            // LDS row stride (floats) of the per-wave B_i[slot] image (32 slots at a time)

__device__ __forceinline__ void mmr_decode_pair_o(int p, int L, int& a, int& a2) {
  int r = p - L, i = 0;
  while (r >= L - 1 - i) { r -= L - 1 - i; ++i; }
  a = i; a2 = i + 1 + r;
}

typedef __bf16 bf16x2r __attribute__((ext_vector_type(2)));

// (v.x, v.y) -> packed bf16 pair (low half = x), round to nearest even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned int mmr_pk_bf16(f32x2 v) {
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2r));
}
__device__ __forceinline__ f32x2 mmr_unpk_bf16(unsigned int u) {
  return (f32x2){__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}

// v = w * x^3 * R_DEG(x) per entry of one 32 x 32 block (8 register pairs per lane; wq: the 8 column-weight pairs)
template <int DEG>
__device__ __forceinline__ void mmr_rem_entries(const f32x2 (&xx)[8], const f32x2 (&wq)[8], f32x2 (&v)[8]) {
  f32x2 pp[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) pp[r] = mm_pkfma(MM_PK(MMRem<DEG>::c[DEG]), xx[r], MM_PK(MMRem<DEG>::c[DEG - 1]));
#pragma unroll
  for (int k = DEG - 2; k >= 0; --k)
#pragma unroll
    for (int r = 0; r < 8; ++r) pp[r] = mm_pkfma(pp[r], xx[r], MM_PK(MMRem<DEG>::c[k]));
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const f32x2 wx = wq[r] * xx[r];
    const f32x2 tv = (xx[r] * xx[r]) * pp[r];
    v[r] = tv * wx;
  }
}

// grid: 1-D over (b, off-diagonal pair, 256-row panel), XCD-remapped.  slab [B][Po][npanel][nT] f64 (ASSIGNED).
template <bool TWO_NB>
__global__ __launch_bounds__(256, 2) void k_bwd_rem_f32(const unsigned short* __restrict__ Zs3, const unsigned short* __restrict__ Zq2,
                                                        const double* __restrict__ Zc64, int Kz, const double* __restrict__ zbar,
                                                        const float* __restrict__ mu, int B, int L, int Mp, int d, int Po, int npanel,
                                                        int nwork, const float* __restrict__ rowO, const float* __restrict__ colO,
                                                        const double* __restrict__ whR, double* __restrict__ slab,
                                                        float* __restrict__ estO, int* __restrict__ rcount) {
  const int orig = blockIdx.x;
  if (orig == 0 && threadIdx.x == 0 && rcount) { rcount[0] = 0; rcount[2] = 0; }   // this pass's route list (k_route_decide follows)
  const int xcd = orig & 7, slotx = orig >> 3;
  const int qn = nwork >> 3, rn = nwork & 7;
  const int wi = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slotx;
  // work order: the panels of one (b, pair), then the batch elements of one pair, then the pairs SORTED BY COLUMN LATENT a':
  // the workgroups in flight on an XCD stream the same latent's split inputs and monomial images (0.6 MB), so they stay in
  // its L2 (pair-major order had up to seven latents' images, 3.5 MB, in flight per XCD)
  const int panel = wi % npanel;
  const int tq = wi / npanel;
  const int b = tq % B, si = tq / B;
  int a2 = 1;
  while (a2 * (a2 + 1) / 2 <= si) ++a2;
  const int a = si - a2 * (a2 - 1) / 2;
  const int lp = a * (L - 1) - a * (a - 1) / 2 + (a2 - a - 1);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int row0 = panel * 256 + wv * 64;
  const int nT = mma_pair_agg_len(d);

  extern __shared__ __align__(16) char smem[];
  float* Bm = reinterpret_cast<float*>(smem) + (size_t)wv * 64 * MMR_BS;                       // [64 rows][MMR_BS]
  double* zr = reinterpret_cast<double*>(smem + (size_t)4 * 64 * MMR_BS * 4) + (size_t)wv * 64 * (d + 1);   // zeta_i | what_i
  double* Tw = reinterpret_cast<double*>(smem + (size_t)4 * 64 * MMR_BS * 4 + (size_t)4 * 64 * (d + 1) * 8) + (size_t)wv * nT;
  // the pair's column weights what'_j, shared by the four waves (a tile needs its 32 right after its bilinear product: from
  // global memory that was an exposed L2 latency per tile)
  float* wlds = reinterpret_cast<float*>(smem + (size_t)4 * 64 * MMR_BS * 4 + (size_t)4 * 64 * (d + 1) * 8 + (size_t)4 * nT * 8);
  const float* wsrc = colO + ((size_t)b * Po + lp) * Mp;
  const bool wl_ok = Mp <= MMR_WLDS_MAX;           // (beyond: read from global memory, tile by tile)
  // csq[2 ct + h]: sum of what'_j^2 over the 16 columns a lane of half h holds of tile ct (the error estimate of mm_route.hip)
  float* csq = wlds + (wl_ok ? Mp : 0);
  if (wl_ok) {
    for (int i = threadIdx.x * 4; i < Mp; i += 1024) *reinterpret_cast<float4*>(wlds + i) = *reinterpret_cast<const float4*>(wsrc + i);
    __syncthreads();
    for (int e = threadIdx.x; e < (Mp >> 4); e += 256) {
      const int ct = e >> 1, hh = e & 1;
      float s2 = 0.0f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(wlds + ct * 32 + 8 * g + 4 * hh);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
      csq[e] = s2;
    }
  }
  __syncthreads();
  // running estimate of the sweep's rounding error (mm_common.h: MM_ROUTE_TOL): per lane -- its two rows, 16 columns of each
  // tile -- sum over the tiles of (max|b|^3)^2 sum what'_j^2; rows' squares and the (1 + X + X^2) factor of rho (X = the lane's
  // largest |b|) once per sweep
  float est = 0.0f, xall = 0.0f;
  // slot 0 (psi = 1) ALSO in plain f32: the row sums sum_j V_ij feed the forward VALUE when the caller takes it from this sweep
  // (mm_moment_match_with_sums); through the bf16 (hi, lo) split they carry 2^-18 per entry -- 4x the forward sweep's error on the
  // off-diagonal covariances at C3 -- so the lane keeps the exact f32 sum of its 16 columns per tile beside the MFMA product
  float rsum[2] = {0.f, 0.f};

  f32x16 acc2[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[rt][nb][r] = 0.0f;

  const bool live = row0 < Mp;                     // Mp % 128 == 0: a wave's 64 rows are all inside or all outside
  if (live) {
    const float* ra = rowO + ((size_t)b * Po + lp) * (size_t)(d + 1) * Mp;      // [d + 1][Mp]: A_i, what_i
    // ---- stationary side: the split A_i of the wave's 64 rows, as in the forward (here the B operand) --------------------
    bf16x8 a1[2], a2v[2], a3[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = row0 + rt * 32 + l31;
      unsigned int hh[8], mm[8], ll[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = ra[(size_t)(j < d ? j : d) * Mp + row];
        mm_split3(j < d ? v : 0.0f, hh[j], mm[j], ll[j]);
      }
      u32x4 ph, pm, pl;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ph[j] = hh[2 * j] | (hh[2 * j + 1] << 16);
        pm[j] = mm[2 * j] | (mm[2 * j + 1] << 16);
        pl[j] = ll[2 * j] | (ll[2 * j + 1] << 16);
      }
      a1[rt] = __builtin_bit_cast(bf16x8, pm);              // (m, m) | (m, h)
      a2v[rt] = __builtin_bit_cast(bf16x8, h ? pl : ph);    // (h, l) | (l, h)
      a3[rt] = __builtin_bit_cast(bf16x8, ph);              // (h, m) | (h, h)
    }
    // ---- streaming side ---------------------------------------------------------------------------------------------
    const char* zbase = reinterpret_cast<const char*>(Zs3 + (size_t)a2 * Mp * 24);           // [Mp/32][3][32][8] bf16
    const char* qbase = reinterpret_cast<const char*>(Zq2 + (size_t)a2 * (Mp / 32) * 4096);  // [Mp/32][8 images][64][8] bf16
    const unsigned int offA = (h ? 0u : 1u) * 512u + (unsigned int)l31 * 16u;
    const unsigned int offB = (h ? 0u : 2u) * 512u + (unsigned int)l31 * 16u;
    const int nct = Mp >> 5;
    constexpr bool two_nb = TWO_NB;                  // 1 + d + d (d + 1) / 2 > 32 monomials (d >= 7): two 32-slot blocks
    // the split inputs of the next tile are prefetched (first thing a tile needs); the monomial images and the column
    // weights of a tile are requested at its top and consumed after its bilinear product and polynomial
    auto load_z = [&](int ct, u32x4& zA, u32x4& zB) {
      zA = *reinterpret_cast<const u32x4*>(zbase + (size_t)ct * 1536 + offA);
      zB = *reinterpret_cast<const u32x4*>(zbase + (size_t)ct * 1536 + offB);
    };
    auto process = [&](int ct, const u32x4& zA, const u32x4& zB) {
      u32x4 psi[8];
      float4 wc[4];
      float mxs[2];
#pragma unroll
      for (int i = 0; i < 8; ++i) psi[i] = *reinterpret_cast<const u32x4*>(qbase + (size_t)ct * 8192 + i * 1024 + lane * 16);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        wc[g] = wl_ok ? *reinterpret_cast<const float4*>(wlds + ct * 32 + 8 * g + 4 * h)
                      : *reinterpret_cast<const float4*>(wsrc + ct * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        // b block, TRANSPOSED: A slot = the streamed columns, B slot = the stationary rows -> lane = row, registers = columns
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, zA), a3[rt], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, zA), a1[rt], acc, 0, 0, 0);
        float m4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < 8; ++r) m4[r & 3] = fmaxf(fmaxf(m4[r & 3], fabsf(acc[2 * r])), fabsf(acc[2 * r + 1]));
        const float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
        mxs[rt] = mx;
        if (__any(mx > 0.03125f))
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, zB), a2v[rt], acc, 0, 0, 0);
        f32x2 xx[8], v[8], wq[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) xx[r] = (f32x2){acc[2 * r], acc[2 * r + 1]};
#pragma unroll
        for (int g = 0; g < 4; ++g) { wq[2 * g] = (f32x2){wc[g].x, wc[g].y}; wq[2 * g + 1] = (f32x2){wc[g].z, wc[g].w}; }
        // range tier of the 32 x 32 block (wave-uniform, ballots)
        if (!__any(mx > MM_TIER1_MAX)) mmr_rem_entries<1>(xx, wq, v);
        else if (!__any(mx > 0.25f)) mmr_rem_entries<3>(xx, wq, v);
        else if (!__any(mx > 0.5f)) mmr_rem_entries<4>(xx, wq, v);
        else if (!__any(mx > 1.0f)) mmr_rem_entries<5>(xx, wq, v);
        else {
#pragma unroll
          for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
              const float x = fminf(xx[r][e2], MM_EXP_CAP_F32);   // (mm_common.h: exponent caps)
              const float xs = fminf(fmaxf(x, -1.0f), 1.0f);
              const float big = (__builtin_amdgcn_exp2f(x * 1.44269504f) - 1.0f) - fmaf(0.5f * x, x, x);
              v[r][e2] = wq[r][e2] * ((fabsf(x) <= 1.0f) ? mm_rem_p5(xs) : big);
            }
        }
        // bf16 (hi, lo) of V, packed as the A operand: K slot t of product s = register 8 s + t of the block
        u32x4 vh[2], vl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2 val = v[4 * s + q];
            rsum[rt] += val[0] + val[1];
            const unsigned int hi = mmr_pk_bf16(val);
            vh[s][q] = hi;
            vl[s][q] = mmr_pk_bf16(val - mmr_unpk_bf16(hi));
          }
        // B_i[slot] += V psi: hi.hi + lo.hi + hi.lo, alternating between the two accumulator chains of the row tile
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
              if constexpr (!two_nb) { if (nb == 1) continue; }   // d <= 6: the monomials fit one 32-slot block
              const u32x4 av = term == 1 ? vl[s] : vh[s];
              const u32x4 bv = psi[(s * 2 + nb) * 2 + (term == 2 ? 1 : 0)];
              acc2[rt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                                     acc2[rt][nb], 0, 0, 0);
            }
      }
      {
        float cs;
        if (wl_ok) cs = csq[2 * ct + h];
        else {
          cs = 0.0f;
#pragma unroll
          for (int g = 0; g < 4; ++g) cs += (wc[g].x * wc[g].x + wc[g].y * wc[g].y) + (wc[g].z * wc[g].z + wc[g].w * wc[g].w);
        }
        const float mt = fmaxf(mxs[0], mxs[1]);
        const float x3 = (mt * mt) * mt;
        est = fmaf(x3 * x3, cs, est);
        xall = fmaxf(xall, mt);
      }
    };
    u32x4 zA0, zB0, zA1, zB1;
    load_z(0, zA0, zB0);
    for (int ct = 0; ct < nct; ct += 2) {          // nct is even (Mp % 128 == 0)
      load_z(ct + 1, zA1, zB1);
      process(ct, zA0, zB0);
      load_z(ct + 2 < nct ? ct + 2 : ct, zA0, zB0);   // clamped: the last pass re-reads its own tile
      process(ct + 1, zA1, zB1);
    }
    {
      const float w0 = ra[(size_t)d * Mp + row0 + l31], w1 = ra[(size_t)d * Mp + row0 + 32 + l31];     // the lane's two rows
      const float xf = fminf(xall, 8.0f);
      const float pf = fmaxf(fmaf(xf, xf, xf) + 1.0f, __expf(xf));   // e^X <= 1 + X + X^2 only up to X = 1.79: the larger of the two (X <= 8)
      est = (est * fmaf(w0, w0, w1 * w1)) * (pf * pf);
    }
  }
  if (estO) {
    float e = est;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off, 64);
    __shared__ float redf[4];
    if (lane == 0) redf[wv] = e;
    __syncthreads();
    if (threadIdx.x == 0) estO[((size_t)b * Po + lp) * npanel + panel] = (redf[0] + redf[1]) + (redf[2] + redf[3]);
  }

  // ---- epilogue: T_rem[alpha, slot] = sum_i what_i zeta_i^alpha B_i[slot] over the wave's rows, f64 ----------------------------
  {
    const int row = row0 + lane;
    for (int k = 0; k < d; ++k)
      zr[lane * (d + 1) + k] = live ? Zc64[((size_t)a * Mp + row) * Kz + k] - ((double)mu[(size_t)b * d + k] - zbar[a * d + k]) : 0.0;
    zr[lane * (d + 1) + d] = live ? whR[((size_t)b * Po + lp) * Mp + row] : 0.0;
  }
  const int nslot = 1 + d + d * (d + 1) / 2, n1 = d * (d + 1), ncomb = nslot + n1 + d * d;
  const int oR1 = 1, oR2 = 1 + d, oK1 = 1 + d + d * d, oK2 = 1 + 2 * d + d * d, oXC = 1 + 2 * d + 2 * d * d;
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {              // 32 slots at a time through the wave's LDS image
    if (nb == 1 && nslot <= 32) break;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) Bm[(32 * rt + 8 * (r >> 2) + 4 * h + (r & 3)) * MMR_BS + l31] = acc2[rt][nb][r];
    wave_sync();
    if (nb == 0) {                              // slot 0 from the plain f32 row sums (the lane's half of the columns + the other half's)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        double t = (double)rsum[rt];
        t += __shfl_xor(t, 32, 64);
        if (h == 0) Bm[(32 * rt + l31) * MMR_BS] = (float)t;
      }
      wave_sync();
    }
    for (int c = lane; c < ncomb; c += 64) {
      int kind, k = 0, k2 = 0, slot;
      if (c < nslot) { kind = 0; slot = c; }
      else if (c < nslot + n1) { kind = 1; k = (c - nslot) / (d + 1); slot = (c - nslot) - k * (d + 1); }
      else { kind = 2; k = (c - nslot - n1) / d; k2 = (c - nslot - n1) - k * d; slot = 0; }
      if ((slot >> 5) != nb) continue;
      double acc = 0.0;
      for (int row = 0; row < 64; ++row) {
        const double* zi = zr + row * (d + 1);
        double wz = zi[d];
        if (kind >= 1) wz *= zi[k];
        if (kind == 2) wz *= zi[k2];
        acc = fma(wz, (double)Bm[row * MMR_BS + (slot & 31)], acc);
      }
      if (kind == 0) {
        if (slot == 0) Tw[0] = acc;
        else if (slot <= d) Tw[oK1 + slot - 1] = acc;
        else {
          int qq = slot - 1 - d, l0 = 0;
          while (qq >= d - l0) { qq -= d - l0; ++l0; }
          const int l1 = l0 + qq;
          Tw[oK2 + l0 * d + l1] = acc;
          Tw[oK2 + l1 * d + l0] = acc;
        }
      } else if (kind == 1) {
        if (slot == 0) Tw[oR1 + k] = acc; else Tw[oXC + k * d + slot - 1] = acc;
      } else {
        Tw[oR2 + k * d + k2] = acc;
      }
    }
    wave_sync();
  }
  __syncthreads();
  const double* Tall = reinterpret_cast<const double*>(smem + (size_t)4 * 64 * MMR_BS * 4 + (size_t)4 * 64 * (d + 1) * 8);
  double* o = slab + (((size_t)b * Po + lp) * npanel + panel) * nT;
  for (int idx = threadIdx.x; idx < nT; idx += 256) o[idx] = (Tall[idx] + Tall[nT + idx]) + (Tall[2 * nT + idx] + Tall[3 * nT + idx]);
}

// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device): `done` is a bitmap over the device ordinals (thread
// safe; devices >= 64 set it on every call)
static hipError_t mmr_set_max_lds_once(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
  if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && bit) done.fetch_or(bit, std::memory_order_release);
  return e;
}

static size_t mmr_rem_lds_bytes(int d, int Mp) {
  return (size_t)4 * 64 * MMR_BS * 4 + (size_t)4 * 64 * (d + 1) * 8 + (size_t)4 * mma_pair_agg_len(d) * 8 +
         (size_t)(Mp <= MMR_WLDS_MAX ? Mp + (Mp >> 4) : 0) * 4;
}

// grid (Po, B), 512 threads: polynomial part + remainder slabs, re-centred at mu -> pagg [B][Po][nT]
__global__ __launch_bounds__(512) void k_pair_agg(const double* __restrict__ mom, int KMp, const double* __restrict__ pairmat,
                                                  const double* __restrict__ zbar, const float* __restrict__ mu, int L, int d,
                                                  int P, int npanel, const double* __restrict__ slab,
                                                  const short* __restrict__ rtab, const unsigned int* __restrict__ amax,
                                                  double* __restrict__ pagg) {
  extern __shared__ double smd[];
  const int po = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, Po = P - L, p = L + po;
  int a, a2;
  mmr_decode_pair_o(p, L, a, a2);
  const int nT = mma_pair_agg_len(d);
  double* nh = smd;                 // [KMp] row-side moments
  double* qh = nh + KMp;            // [KMp] column-side moments
  double* G = qh + KMp;             // [d][d]
  double* dmu = G + d * d;          // [d] mu - zbar_a
  double* dmu2 = dmu + d;           // [d] mu - zbar_a'
  double* dmuP = dmu2 + d;          // [d] the shift inside b_ij: zero where k_pairvec recentred the rows (mm_mono.h), else dmu
  double* T = dmuP + d;             // [nT]
  double* scr = T + nT;             // mma_pair_poly_scratch(d)
  const bool rcen = mm_rows_recentred(amax[(size_t)b * Po + po]);
  {
    const double* nm = mom + (((size_t)b * Po + po) * 2 + 0) * MM_MOM_SPLIT * KMp;
    const double* qm = mom + (((size_t)b * Po + po) * 2 + 1) * MM_MOM_SPLIT * KMp;
    for (int k = tid; k < KMp; k += blockDim.x) {
      double sn = 0.0, sq = 0.0;
#pragma unroll
      for (int t = 0; t < MM_MOM_SPLIT; ++t) { sn += nm[t * KMp + k]; sq += qm[t * KMp + k]; }
      nh[k] = sn; qh[k] = sq;
    }
    const double* pm = pairmat + ((size_t)b * P + p) * (d * d + 1);
    for (int idx = tid; idx < d * d; idx += blockDim.x) G[idx] = pm[idx];
    if (tid < d) {
      const double m = (double)mu[(size_t)b * d + tid];
      dmu[tid] = m - zbar[a * d + tid];
      dmu2[tid] = m - zbar[a2 * d + tid];
      dmuP[tid] = rcen ? 0.0 : dmu[tid];
    }
  }
  __syncthreads();
  mma_pair_poly(MMADevCtx(), d, G, dmuP, nh, qh, T, scr, rtab);
  if (rcen) {
    // the polynomial part came out with the row monomials of zc_i: to those of zeta_i = zc_i - dmu, as the remainder slabs have them
    __syncthreads();
    mma_pair_convert_rows(MMADevCtx(), d, dmu, T);
    __syncthreads();
  }
  const double* sl = slab + ((size_t)b * Po + po) * npanel * nT;
  for (int idx = tid; idx < nT; idx += blockDim.x) {
    double s = 0.0;
    for (int pn = 0; pn < npanel; ++pn) s += sl[(size_t)pn * nT + idx];
    T[idx] += s;
  }
  __syncthreads();
  mma_pair_convert(MMADevCtx(), d, dmu2, T);
  __syncthreads();
  double* o = pagg + ((size_t)b * Po + po) * nT;
  for (int idx = tid; idx < nT; idx += blockDim.x) o[idx] = T[idx];
}

__global__ void k_cast_f32_f64(const float* __restrict__ x, double* __restrict__ y, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (double)x[i];
}

int mm_launch_wmom_full(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl,
                        int B, int L, int d, hipStream_t stream);

extern "C" int mm_bwd_f32_supported(int d) { return d >= 1 && d <= 8; }

size_t mm_bwd_f32_slab_bytes(int B, int Po, int Mp, int d) {
  return (size_t)B * Po * ((Mp + 255) / 256) * mma_pair_agg_len(d) * sizeof(double);
}

// Off-diagonal aggregates of an f32 pack whose q stage is current on `ws` (mm_q_forward, MM_F32).  mu: [B][d] f32.
// stages (bench.py times them apart): MM_STAGE_OFFDIAG = the remainder sweep k_bwd_rem_f32 alone; MM_STAGE_FINALIZE = the full
// moment GEMM and k_pair_agg
int mm_launch_route(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl, int B, int L, int M,
                    int d, int flags, int agg, double* out, int32_t* status, hipStream_t stream);

int mm_launch_bwd_offdiag_f32(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl,
                              int B, int L, int M, int d, int flags, const float* mu, double* slab, double* pagg, int32_t* status,
                              hipStream_t stream, int stages = MM_STAGE_OFFDIAG | MM_STAGE_FINALIZE, int agg_threads = 512) {
  // agg_threads: workgroup size of k_pair_agg -- 512 on its own (256 / 1024 measured 13 % / 30 % slower), 256 when it runs on a side
  // stream BESIDE the diagonal sweep (mm_compose_bwd.hip): one wave per SIMD and 80 KB of LDS fit next to that kernel's two waves
  if (wl.Po <= 0) return 0;
  if (!mm_bwd_f32_supported(d)) return MM_E_DIM;
  if (const int rj = mm_fork_join_wait(stream)) return rj;   // the off-diagonal operands / moment chain may still be on the q stage's side stream
  const int npanel = (wl.Mp + 255) / 256;
  hipError_t e = hipSuccess;
  if (stages & MM_STAGE_OFFDIAG) {
  const long long nwork_ll = (long long)npanel * wl.Po * B;
  if (nwork_ll <= 0 || nwork_ll > 0x7fffffffLL) return MM_E_DIM;
  const size_t shm = mmr_rem_lds_bytes(d, wl.Mp);
  if (shm > 160 * 1024) return MM_E_DIM;
  // (the attribute is set once per variant AND DEVICE -- HIP applies it per device --, to the most the kernel may ask for:
  // not inside a later stream capture)
#define MMR_LAUNCH(TWO_)                                                                                                       \
  do {                                                                                                                        \
    static std::atomic<unsigned long long> attr_set{0ull};                                                                    \
    e = mmr_set_max_lds_once((const void*)k_bwd_rem_f32<TWO_>, (int)mmr_rem_lds_bytes(8, MMR_WLDS_MAX), attr_set);            \
    if (e != hipSuccess) return (int)e;                                                                                       \
    hipLaunchKernelGGL(k_bwd_rem_f32<TWO_>, dim3((int)nwork_ll), dim3(256), shm, stream, (const unsigned short*)(packed + ml.Zs3), \
                       (const unsigned short*)(packed + ml.Zq2), (const double*)(packed + ml.Zc64), ml.Kz,                    \
                       (const double*)(packed + ml.zbar), mu, B, L, wl.Mp, d, wl.Po, npanel, (int)nwork_ll,                   \
                       (const float*)(ws + wl.rowO), (const float*)(ws + wl.colO), (const double*)(ws + wl.whR), slab,        \
                       (float*)(ws + wl.estO), (int*)(ws + wl.rcount));                                                       \
  } while (0)
  if (1 + d + d * (d + 1) / 2 > 32) MMR_LAUNCH(true); else MMR_LAUNCH(false);
#undef MMR_LAUNCH
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  }
  // items whose estimated f32 / bf16 rounding error is beyond MM_ROUTE_TOL of the block's scale: remainder aggregates in f64
  if (((stages & MM_STAGE_OFFDIAG) && !(stages & MM_ISTAGE_NO_ROUTE)) || (stages & MM_ISTAGE_ROUTE)) {
    const int rcr = mm_launch_route(packed, ml, ws, wl, B, L, M, d, flags, 1, slab, status, stream);
    if (rcr) return rcr;
  }
  if (!(stages & MM_STAGE_FINALIZE)) return 0;
  if (const int rj = mm_fork_join_wait(stream)) return rj;   // the q stage's k_spoly (side stream) still reads the table the full GEMM overwrites
  const int rc = mm_launch_wmom_full(packed, ml, ws, wl, B, L, d, stream);
  if (rc) return rc;
  const int nT = mma_pair_agg_len(d);
  const size_t shm2 = (size_t)(2 * ml.KMp + d * d + 3 * d + nT + mma_pair_poly_scratch(d)) * sizeof(double);
  {
    static std::atomic<unsigned long long> agg_attr_set{0ull};
    const size_t shm_max = (size_t)(2 * mm_moment_cols(8) + 64 + 24 + mma_pair_agg_len(8) + mma_pair_poly_scratch(8)) * sizeof(double);
    e = mmr_set_max_lds_once((const void*)k_pair_agg, (int)shm_max, agg_attr_set);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k_pair_agg, dim3(wl.Po, B), dim3(agg_threads), shm2, stream, (const double*)(ws + wl.mom), ml.KMp,
                     (const double*)(ws + wl.pairmat), (const double*)(packed + ml.zbar), mu, L, d, wl.P, npanel,
                     (const double*)slab, (const short*)(packed + ml.rtab), (const unsigned int*)(ws + wl.amax), pagg);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

int mm_launch_cast_f32_f64(const float* x, double* y, size_t n, hipStream_t stream) {
  hipLaunchKernelGGL(k_cast_f32_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, n);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// The aggregates alone (tests, diagnostics): q stage of (mu, Sigma) on `workspace`, then pagg [B][Po][1 + 2d + 3d^2] f64 =
// sum_ij Omega_ij (1 | zeta_i | zeta_i zeta_i^T | zeta'_j | zeta'_j zeta'_j^T | zeta_i zeta'_j^T) per off-diagonal pair.
// scratch: mm_backward_pair_aggregates_bytes.
extern "C" size_t mm_backward_pair_aggregates_bytes(int B, int L, int M, int d, int flags) {
  if (B <= 0 || L <= 0 || M <= 0 || !mm_bwd_f32_supported(d)) return 0;
  return mm_bwd_f32_slab_bytes(B, mm_num_pairs(L, flags) - L, mm_round_up_int(M, MM_M_ALIGN), d) + 256;
}

extern "C" int mm_backward_pair_aggregates(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                                           const void* mu, const void* Sigma, int flags, void* workspace, size_t workspace_bytes,
                                           void* scratch, size_t scratch_bytes, void* pagg, size_t pagg_bytes,
                                           int32_t* status, void* stream) {
  if (!packed || !mu || !Sigma || !workspace || !scratch || !pagg) return MM_E_ARG;
  if (L <= 0 || M <= 0 || d <= 0 || B <= 0) return MM_E_ARG;
  if (d > MM_DMAX) return MM_E_DIM;
  if (dtype != MM_F32 || !mm_bwd_f32_supported(d)) return MM_E_DTYPE;
  const MMModelLayout ml = mm_model_layout(L, M, d, dtype, 1);
  const MMWorkspaceLayout wl = mm_workspace_layout(B, L, M, d, dtype, flags);
  if (packed_bytes < ml.Cm || workspace_bytes < wl.total) return MM_E_WORKSPACE;
  if (scratch_bytes < mm_backward_pair_aggregates_bytes(B, L, M, d, flags)) return MM_E_WORKSPACE;
  if (pagg_bytes < (size_t)B * wl.Po * mma_pair_agg_len(d) * sizeof(double)) return MM_E_WORKSPACE;
  char* ws = (char*)workspace;
  int rc = mm_q_forward(packed, packed_bytes, L, M, d, dtype, B, mu, Sigma, flags | MM_ISTAGE_NO_M56, ws + wl.f1s, ws + wl.crs, nullptr,
                        workspace, workspace_bytes, status, stream);
  if (rc) return rc;
  return mm_launch_bwd_offdiag_f32((const char*)packed, ml, ws, wl, B, L, M, d, flags, (const float*)mu, (double*)scratch,
                                   (double*)pagg, status, (hipStream_t)stream);
}
