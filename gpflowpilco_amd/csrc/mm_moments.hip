// Weight moments of the f32 off-diagonal pairs and their per-(b, pair) contraction (gfx950).
//
// For an off-diagonal pair (a, a') of an f32 model the tile kernel (mm_mfma.hip) reduces
//     S = sum_ij what_i what'_j E(b_ij),     b_ij = (zc_i - dmu)^T G zc'_j,
// with zc, zc' the model's inducing inputs centred at their per-latent centroids; dmu = 0 where k_pairvec recentred the rows at the
// centroid (round 5, mm_mono.h: the usual case), mu_b - zbar_a where it left them centred at mu.  Any POLYNOMIAL part
// P(x) = sum_n a_n x^n of E collapses to moments of the two weight vectors against model-constant monomial
// tables -- O(M C(d+n, n)) instead of O(M^2) per (b, pair), and exact in f64:
//     sum_ij what_i what'_j b_ij^n = < M_n , G^{(x)n} Q_n >,
//     Q_n = sum_j what'_j zc'_j^{(x)n},   M_n = sum_i what_i (zc_i - dmu)^{(x)n}.
// Orders 0..2 (1 + b + b^2/2) are ALWAYS taken this way: in f32 it is the rounding of the linear term that
// costs the digits of S (DESIGN.md "fp32 error budget").  For d <= 8 a (b, pair) whose Cauchy-Schwarz bound
// max_i |A_i| max_j |zc'_j| is <= 1/2 (MM_COLLAPSE_BOUND2) is COLLAPSED (mm_common.h): the polynomial p6(x) = x^3 (C0 + .. + C3 x^3) --
// the tile kernel's own degree-3 tier of the remainder, |p6 - r| <= 5.8e-10 on |x| <= 1/4 -- comes from moments too: its cubic
// term from the f64 tables of THIS file (the forward forms the columns to degree 3; the table goes to degree 4, 495 columns at
// d = 8, for the backward's aggregates), its degree-4, -5 and -6 terms from bf16 split tables on the matrix pipe (mm_moments6.hip).  The tile kernel skips every wave tile with max|b| <= 1/4 after a one-MFMA
// screening product and reduces only the correction r(x) - p6(x) on the others.
//
//   k_wmom_perm : per latent the rows of its GEMMs ordered by collapse class (degree 6 | 5 | 4 | not collapsed): every column block of
//                 either GEMM is formed for the rows that read it only;
//   k_wmom_gemm : mom[(b, pair, side)][:] = sum_m what_m Zm[latent(side)][m][:]   -- an f64 GEMM
//                 [rows = B per (pair, side)] x [K = Mp] x [N = KMp columns] on v_mfma_f64_16x16x4_f64:
//                 64 x 128 output tile per workgroup (4 waves x 64 x 32), K-blocks of 32 staged through LDS
//                 (bank-conflict-free strides), next block's global loads in flight during the MFMAs,
//                 split-K over MM_MOM_SPLIT slices, XCD-aware 1-D grid (tiles sharing a table slice share an L2);
//   k_spoly     : per (b, pair) the d^n contractions above for n = 0..2 (or 0..4), one workgroup each (n = 5, 6: k_spoly56).
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"
#include "mm_mono.h"

typedef double f64x4m __attribute__((ext_vector_type(4)));

#define MM_GEMM_RB 64        // batch rows per workgroup
#define MM_GEMM_NB 128       // table columns per workgroup (4 waves x 32)
#define MM_GEMM_KB 32        // K block staged through LDS
#define MM_GEMM_AS 34        // LDS row stride (doubles) of the A block: 16 rows x 2 k land on 32 distinct 8-byte banks
#define MM_GEMM_BS 144       // LDS row stride of the B block: rows k and k + 1 are 16 banks apart

__device__ __forceinline__ void mm_decode_pair_m(int p, int L, int& a, int& a2) {
  int r = p - L, i = 0;
  while (r >= L - 1 - i) { r -= L - 1 - i; ++i; }
  a = i; a2 = i + 1 + r;
}

// (54 KB of LDS admit two workgroups per CU = two waves per SIMD: pin the register budget to that occupancy, else the
// compiler aims at three waves and spills the A prefetch registers inside the K loop)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_wmom_gemm(const double* __restrict__ whR, const double* __restrict__ whC,
                                                      const double* __restrict__ Zm, int KMp, int L, int Mp, int B, int Po,
                                                      int nrb, int ncb, int nwork, int col_deg3,
                                                      const unsigned int* __restrict__ amaxc, const double* __restrict__ zmax2,
                                                      const int* __restrict__ gperm, double* __restrict__ mom) {
  // The GEMM of latent a: rows = every weight vector taken against a's table -- (L - 1) B of them: for partner
  // a' > a the ROW side of pair (a, a'), for a' < a the COLUMN side of pair (a', a) -- so a 64-row block is full
  // whatever B is (rows used to be the B elements of ONE (pair, side): half-empty blocks at the C4 shard's B = 32).
  // work item -> (latent, column block, k slice, row block); consecutive items (one XCD after the remap) share the
  // table slice [k slice][column block] of one latent
  const int orig = blockIdx.x;
  const int xcd = orig & 7, slot = orig >> 3;
  const int qn = nwork >> 3, rn = nwork & 7;
  int wi = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  const int rb = wi % nrb; wi /= nrb;
  const int ks = wi % MM_MOM_SPLIT; wi /= MM_MOM_SPLIT;
  const int cb = wi % ncb; wi /= ncb;
  const int a = wi;
  const int R = (L - 1) * B;                                // rows of this latent's GEMM
  // row r -> (batch element, off-diagonal pair index, side, column latent of that pair)
  auto row_item = [&](int r, int& b, int& po, int& side, int& acol) {
    const int which = r / B;
    b = r - which * B;
    const int ap = which < a ? which : which + 1;           // partner latent
    const int lo = ap < a ? ap : a, hi = ap < a ? a : ap;
    po = lo * (L - 1) - lo * (lo - 1) / 2 + (hi - lo - 1);  // pairs (lo, hi), lo < hi, in mm_decode_pair order
    side = ap < a ? 1 : 0;
    acol = hi;
  };
  const double* tab = Zm + (size_t)a * Mp * KMp;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  // a column block made of cubic / quartic monomials only is needed by collapsed (b, pair) items alone: the
  // workgroup leaves when none of its 64 rows belongs to one (k_spoly then never reads those columns)
  // gperm (k_wmom_perm): this latent's rows with the collapsed items first -- the cubic / quartic blocks then run over the
  // first ncoll rows only (in natural row order collapsed and other items share most 64-row blocks, and the early exit below
  // saved a third of what it could: BASELINE recipe, 18 % of the items collapsed)
  const int* perm = nullptr;
  if (cb * MM_GEMM_NB >= col_deg3) {
    if (gperm != nullptr) {
      if (rb * MM_GEMM_RB >= gperm[(size_t)L * R + a]) return;
      perm = gperm + (size_t)a * R;
    } else {
      const int r = rb * MM_GEMM_RB + lane;
      bool c = false;
      if (amaxc != nullptr && r < R) {
        int b, po, side, acol;
        row_item(r, b, po, side, acol);
        c = mm_item_collapsed(amaxc[(size_t)b * Po + po]);
      }
      if (!__any(c)) return;                                // every wave evaluates the same 64 rows: uniform exit
    }
  }
  const int kslice = Mp / MM_MOM_SPLIT;                     // Mp % 128 == 0: a multiple of MM_GEMM_KB
  const int k_begin = ks * kslice, nkb = kslice / MM_GEMM_KB;

  __shared__ double As[MM_GEMM_RB * MM_GEMM_AS];
  __shared__ double Bs[MM_GEMM_KB * MM_GEMM_BS];

  // global -> register staging: A: row ar, 8 consecutive k;  B: table row bk, 16 consecutive columns
  const int ar = tid >> 2, ak = (tid & 3) * 8;
  int arow = rb * MM_GEMM_RB + ar;
  arow = arow < R ? arow : R - 1;                           // rows past the end recompute the last one (not stored)
  int ab, apo, aside, aacol;
  row_item(perm ? perm[arow] : arow, ab, apo, aside, aacol);
  const double* aptr = (aside ? whC : whR) + ((size_t)ab * Po + apo) * Mp + k_begin + ak;

  const int bk = tid >> 3, bc = (tid & 7) * 16;
  const int col0 = cb * MM_GEMM_NB + bc;
  const bool bvalid = col0 < KMp;                           // KMp % 16 == 0: a 16-column chunk is all in or all out
  const double* bptr = tab + (size_t)(k_begin + bk) * KMp + (bvalid ? col0 : 0);
  double2 ra[4], rbv[8];
  auto load_regs = [&](int kb) {
    const double* ap = aptr + kb * MM_GEMM_KB;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const double2*>(ap + 2 * i);
    const double* bp = bptr + (size_t)kb * MM_GEMM_KB * KMp;
#pragma unroll
    for (int i = 0; i < 8; ++i) rbv[i] = *reinterpret_cast<const double2*>(bp + 2 * i);
  };
  f64x4m acc[4][2];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = (f64x4m){0.0, 0.0, 0.0, 0.0};

  load_regs(0);
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();                                        // the previous block's MFMA operands have been read
    // 16-byte stores.  The 8 threads of a table row write chunks 128 bytes (= all 32 banks) apart: stored in place
    // they would all hit the same banks.  Pair i of chunk c therefore goes to slot (i + c) & 7 of the chunk (a rotation
    // inside the 16 doubles); the B-operand read below undoes it.
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<double2*>(&As[ar * MM_GEMM_AS + ak + 2 * i]) = ra[i];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      *reinterpret_cast<double2*>(&Bs[bk * MM_GEMM_BS + bc + 2 * ((i + (tid & 7)) & 7)]) = bvalid ? rbv[i] : make_double2(0.0, 0.0);
    __syncthreads();
    load_regs(kb + 1 < nkb ? kb + 1 : kb);                  // unconditional (clamped): in flight during the MFMAs
#pragma unroll
    for (int s = 0; s < MM_GEMM_KB / 4; ++s) {
      double av[4], bv[2];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) av[rt] = As[(rt * 16 + l15) * MM_GEMM_AS + 4 * s + kq];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)     // column wv*32 + ct*16 + l15: chunk c = 2 wv + ct, pair l15 >> 1 sits in slot (pair + c) & 7
        bv[ct] = Bs[(4 * s + kq) * MM_GEMM_BS + wv * 32 + ct * 16 + 2 * (((l15 >> 1) + 2 * wv + ct) & 7) + (l15 & 1)];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rt], bv[ct], acc[rt][ct], 0, 0, 0);
    }
  }
  // accumulator layout: lane (l15 = column, kq), register r  <->  row kq + 4 r of the 16 x 16 tile
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rb * MM_GEMM_RB + rt * 16 + kq + 4 * r;
      if (row >= R) continue;
      int b, po, side, acol;
      row_item(perm ? perm[row] : row, b, po, side, acol);
      double* o = mom + ((((size_t)b * Po + po) * 2 + side) * MM_MOM_SPLIT + ks) * KMp;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int col = cb * MM_GEMM_NB + wv * 32 + ct * 16 + l15;
        if (col < KMp) o[col] = acc[rt][ct][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------
// k_wmom_perm: grid L, 1024 threads.  gperm[a][0 .. R) = the rows r of latent a's GEMM (R = (L - 1) B; row -> (partner, b) as in
// k_wmom_gemm) ordered by collapse class (mm_common.h: MM_C6_X5_2), the collapsed classes first in their natural order, the others
// behind them from the back; three counts behind the permutations.  Two passes (class totals, placement), 1024 rows at a time.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_wmom_perm(const unsigned int* __restrict__ amaxc, const double* __restrict__ zmax2, int L, int B,
                                                    int Po, int* __restrict__ gperm) {
  // classes by the item's bound X^2 (mm_common.h): 0: collapsed, X > 1/16 (degrees 3..6); 1: 1/32 < X <= 1/16 (3..5); 2: X <= 1/32
  // (3, 4); 3: not collapsed.  Classes 0..2 from the front in this order (stable inside a class), class 3 from the back.
  const int a = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int R = (L - 1) * B;
  __shared__ int wsum[3][16];
  __shared__ int tot[3];
  int* out = gperm + (size_t)a * R;
  // pass 1: class totals (the front classes' start offsets)
  int cnt[3] = {0, 0, 0};
  auto cls_of = [&](int r) {
    const int which = r / B, b = r - which * B;
    const int ap = which < a ? which : which + 1;
    const int lo = ap < a ? ap : a, hi = ap < a ? a : ap;
    const int po = lo * (L - 1) - lo * (lo - 1) / 2 + (hi - lo - 1);
    // (amaxc: the bound of the item's COLLAPSED row groups, mm_mono.h; none: class 3)
    const unsigned int ac = amaxc[(size_t)b * Po + po];
    const float x2 = mm_collapse_bound2(ac, zmax2[hi]);
    return !mm_item_collapsed(ac) ? 3 : (x2 > MM_C6_X5_2 ? 0 : (x2 > MM_C6_X4_2 ? 1 : 2));
  };
  for (int r = tid; r < R; r += 1024) { const int c = cls_of(r); if (c < 3) ++cnt[c]; }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt[c] += __shfl_down(cnt[c], off, 64);
    if (lane == 0) wsum[c][wv] = cnt[c];
  }
  __syncthreads();
  if (tid < 3) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[tid][w]; tot[tid] = t; }
  __syncthreads();
  int ofs[3] = {0, tot[0], tot[0] + tot[1]};                          // next free slot of each front class
  int ofs_n = 0;                                                      // rows of class 3 so far (filled from the back)
  // pass 2: stable placement, 1024 rows at a time
  for (int r0 = 0; r0 < R; r0 += 1024) {
    const int r = r0 + tid;
    const bool in = r < R;
    const int c = in ? cls_of(r) : 4;
    int before[3], ctot[3];
    __syncthreads();                                                   // (the previous chunk's wsum has been read)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned long long bal = __ballot(c == k);
      before[k] = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wsum[k][wv] = __popcll(bal);
    }
    __syncthreads();
    int nfront_before = 0, nfront_tot = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int wb = 0, ct = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) { if (w < wv) wb += wsum[k][w]; ct += wsum[k][w]; }
      before[k] += wb; ctot[k] = ct;
      nfront_before += before[k]; nfront_tot += ct;
    }
    if (in) {
      if (c < 3) out[ofs[c] + before[c]] = r;
      else out[R - 1 - (ofs_n + (tid - nfront_before))] = r;
    }
    const int nin = R - r0 < 1024 ? R - r0 : 1024;
#pragma unroll
    for (int k = 0; k < 3; ++k) ofs[k] += ctot[k];
    ofs_n += nin - nfront_tot;
  }
  if (tid == 0) {
    gperm[(size_t)L * R + a] = tot[0] + tot[1] + tot[2];               // collapsed rows
    gperm[(size_t)L * R + L + a] = tot[0] + tot[1];                    // rows that need the degree-5 moments
    gperm[(size_t)L * R + 2 * L + a] = tot[0];                         // rows that need the degree-6 moments
  }
}

// ---------------------------------------------------------------------------------------------
// k_spoly: s12[b][po] = sum_{n=0}^{N} a_n < M_n, G^{(x)n} Q_n >,  a = (1, 1, 1/2, c0, c1); N = 4 for a collapsed
// (b, pair), else 2.  One 256-thread workgroup per (b, pair); tensors are held in full (d^n entries) in LDS:
//   T <- Q_n expanded from the packed column-side moments; G applied along every index in place (each thread
//   owns whole fibres); then, with < (zc - dmu)^{(x)n}, T > = sum_k C(n,k) (-1)^{n-k} < zc^{(x)k} (x) dmu^{(x)(n-k)}, T >
//   (T symmetric), k runs from n down to 0: dot of the leading-k-index tensor with the packed row moments of
//   degree k, then the last index is contracted with dmu.
// ---------------------------------------------------------------------------------------------
// Tensors are held with the power-of-two extent DK per index (d <= DK; entries with an index >= d are zero); the
// rank of an index tuple inside its degree block comes from a table written at pack time (MMModelLayout::rtab).
__device__ __forceinline__ double mm_block_sum256m(double v, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// DK = 8 (d <= 8, degree <= 4) or 32 (d <= 32, degree 2); LB = log2 DK; offs[n] = first column of the degree-n block
template <int DK, int LB>
__global__ __launch_bounds__(256) void k_spoly(const double* __restrict__ mom, int KMp, const double* __restrict__ pairmat,
                                               const double* __restrict__ zbar, const double* __restrict__ zmax2,
                                               const unsigned int* __restrict__ amax, const unsigned int* __restrict__ amaxc,
                                               const float* __restrict__ mu,
                                               int L, int d, int P, int deg, int allow_collapse,
                                               int off1, int off2, int off3, int off4, const short* __restrict__ rtab,
                                               double c0, double c1, double* __restrict__ s12) {
  extern __shared__ double sm[];
  const int po = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int Po = P - L, p = L + po;
  int a, a2;
  mm_decode_pair_m(p, L, a, a2);
  const int dn = 1 << (LB * deg);                          // DK^deg
  // T / U are stored with one pad double per 32 (position i lives at i + (i >> 5)): the fibre passes read
  // DK entries 8^t doubles apart from every lane, which unpadded lands 8 lanes on one LDS bank
#define MM_PADI(i_) ((i_) + ((i_) >> 5))
  double* T = sm;                      // [DK^deg], padded
  double* U = T + MM_PADI(dn);         // [DK^(deg-1)], padded
  double* nh = U + MM_PADI(dn >> LB);  // [KMp] row-side moments
  double* qh = nh + KMp;               // [KMp] column-side moments
  double* Gs = qh + KMp;               // [DK][DK], zero padded
  double* dmu = Gs + DK * DK;          // [DK], zero padded
  __shared__ double red[4];
  {
    const double* nm = mom + (((size_t)b * Po + po) * 2 + 0) * MM_MOM_SPLIT * KMp;
    const double* qm = mom + (((size_t)b * Po + po) * 2 + 1) * MM_MOM_SPLIT * KMp;
    for (int k = tid; k < KMp; k += 256) {                 // fixed summation order of the split-K partials
      double sn = 0.0, sq = 0.0;
#pragma unroll
      for (int t = 0; t < MM_MOM_SPLIT; ++t) { sn += nm[t * KMp + k]; sq += qm[t * KMp + k]; }
      nh[k] = sn; qh[k] = sq;
    }
    const double* pm = pairmat + ((size_t)b * P + p) * (d * d + 1);
    for (int idx = tid; idx < DK * DK; idx += 256) {
      const int i = idx >> LB, j = idx & (DK - 1);
      Gs[idx] = (i < d && j < d) ? pm[i * d + j] : 0.0;
    }
    // rows recentred at zbar_a (k_pairvec, mm_mono.h): the row moments are the table's own, no shift; an item that kept its rows
    // centred at mu (marked in amax) takes the binomial shift by dmu = mu - zbar_a below
    const bool rcen = mm_rows_recentred(amax[(size_t)b * Po + po]);
    if (tid < DK) dmu[tid] = (tid < d && !rcen) ? (double)mu[(size_t)b * d + tid] - zbar[a * d + tid] : 0.0;
  }
  __syncthreads();
  // (an item with at least one collapsed row group, mm_mono.h: the CUBIC term C0 b^3 then comes from these f64 moments for EVERY
  // row -- the identity is exact for any b -- and the sweep's dense row groups reduce r(b) - C0 b^3; orders 4, 5, 6, from f32
  // moments, cover the collapsed groups alone: k_pairvec_reg writes zero bf16 row weights for the others)
  const bool coll = allow_collapse && deg >= 3 && mm_item_collapsed(amaxc[(size_t)b * Po + po]);
  const int nmax = coll ? deg : 2;                         // (deg = 3: the quartic term is mm_moments6.hip's)
  double Gr[DK == 8 ? 64 : 1];
  if constexpr (DK == 8) {
#pragma unroll
    for (int i = 0; i < 64; ++i) Gr[i] = Gs[i];
  }
  double total = 0.0;                                      // per-thread partial of the final sum
  for (int n = nmax; n >= 1; --n) {
    const double an = n == 1 ? 1.0 : n == 2 ? 0.5 : n == 3 ? c0 : c1;
    const int dsz = 1 << (LB * n);
    const int offn = n == 1 ? off1 : n == 2 ? off2 : n == 3 ? off3 : off4;
    // rank-table blocks: degree k starts at sum_{j<k} DK^j - 1 ... = (DK^k - DK) / (DK - 1)
    const short* rtn = rtab + (((1 << (LB * n)) - DK) / (DK - 1));
    for (int idx = tid; idx < dsz; idx += 256) {
      const int r = rtn[idx];
      T[MM_PADI(idx)] = r >= 0 ? qh[offn + r] : 0.0;
    }
    __syncthreads();
    // T <- G applied along every index: axis t has stride DK^t; a fibre = the DK entries along that axis,
    // owned by one thread (registers), so the transform is in place.  DK = 8: G itself sits in registers
    // (one LDS read per FMA would make the passes LDS-issue bound)
    for (int t = 0; t < n; ++t) {
      const int sh = LB * t, nfib = dsz >> LB;
      for (int f = tid; f < nfib; f += 256) {
        const int lo = f & ((1 << sh) - 1), hi = f >> sh;
        const int base = (hi << (sh + LB)) + lo;
        double v[DK];
#pragma unroll
        for (int l = 0; l < DK; ++l) v[l] = T[MM_PADI(base + (l << sh))];
        if constexpr (DK == 8) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < 8; ++l) s = fma(Gr[k * 8 + l], v[l], s);
            T[MM_PADI(base + (k << sh))] = s;
          }
        } else {
#pragma unroll 2
          for (int k = 0; k < DK; ++k) {
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < DK; ++l) s = fma(Gs[k * DK + l], v[l], s);
            T[MM_PADI(base + (k << sh))] = s;
          }
        }
      }
      __syncthreads();
    }
    // k = n .. 0: dot with the row moments of degree k, then contract the last (highest-stride) index with dmu
    double* cur = T;
    double* oth = U;
    int ksz = dsz;
    double sign_binom = 1.0;                               // C(n, k) (-1)^(n-k), starting at k = n
    for (int k = n; k >= 0; --k) {
      const int offk = k == 0 ? 0 : k == 1 ? off1 : k == 2 ? off2 : k == 3 ? off3 : off4;
      double part = 0.0;
      const short* rtk = rtab + (k ? (((1 << (LB * k)) - DK) / (DK - 1)) : 0);
      for (int idx = tid; idx < ksz; idx += 256) {
        const int r = k ? rtk[idx] : 0;
        if (r >= 0) part = fma(cur[MM_PADI(idx)], nh[offk + r], part);
      }
      total = fma(an * sign_binom, part, total);
      if (k == 0) break;
      const int nsz = ksz >> LB;                           // contract the highest index (stride nsz) with dmu
      for (int idx = tid; idx < nsz; idx += 256) {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < DK; ++l) s = fma(cur[MM_PADI(l * nsz + idx)], dmu[l], s);
        oth[MM_PADI(idx)] = s;
      }
      __syncthreads();
      double* tsw = cur; cur = oth; oth = tsw;
      ksz = nsz;
      sign_binom = -sign_binom * (double)k / (double)(n - k + 1);
    }
    __syncthreads();
  }
  const double tot = mm_block_sum256m(total, red);
  if (tid == 0) s12[(size_t)b * Po + po] = tot + nh[0] * qh[0];
#undef MM_PADI
}

// the moment GEMM alone with EVERY column block for every (b, pair) (no early exit for the cubic / quartic blocks): the
// backward's aggregates need the moments to degree 4 whether or not the forward collapsed the item (mm_bwd_f32.hip)
int mm_launch_wmom_full(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl,
                        int B, int L, int d, hipStream_t stream) {
  const int nrb = ((L - 1) * B + MM_GEMM_RB - 1) / MM_GEMM_RB, ncb = (ml.KMp + MM_GEMM_NB - 1) / MM_GEMM_NB;
  const long long nwork_ll = (long long)L * ncb * MM_MOM_SPLIT * nrb;
  if (nwork_ll <= 0 || nwork_ll > 0x7fffffffLL) return MM_E_DIM;
  hipLaunchKernelGGL(k_wmom_gemm, dim3((int)nwork_ll), dim3(256), 0, stream, (const double*)(ws + wl.whR),
                     (const double*)(ws + wl.whC), (const double*)(packed + ml.Zm), ml.KMp, L, wl.Mp, B, wl.Po, nrb, ncb,
                     (int)nwork_ll, 0x7fffffff /* no block is optional */, (const unsigned int*)nullptr, (const double*)(packed + ml.zmax2),
                     (const int*)nullptr, (double*)(ws + wl.mom));
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

int mm_launch_moments56(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl,
                        int B, int L, int d, int allow, hipStream_t stream);

int mm_launch_moments(const char* packed, const MMModelLayout& ml, char* ws, const MMWorkspaceLayout& wl,
                      int B, int L, int d, const void* mu_f32, int flags, hipStream_t stream) {
  const double* Zm = (const double*)(packed + ml.Zm);
  double* mom = (double*)(ws + wl.mom);
  // per latent a GEMM with (L - 1) B rows (k_wmom_gemm): row blocks, column blocks, k slices
  // (the forward reads the f64 moments to degree 3 only: the quartic term of a collapsed item comes from the bf16 GEMM with the
  // degree-5/6 ones, mm_moments6.hip -- emulated on the BASELINE recipe it costs <= 4.7e-7 of the covariance scale there, where the
  // CUBIC term would cost 1e-5; the backward's full GEMM, mm_launch_wmom_full, still forms every column)
  const int deg = mm_moment_deg(d) >= 4 ? 3 : mm_moment_deg(d);
  const int ncol_fwd = mm_mono_offset(deg + 1, d);
  const int nrb = ((L - 1) * B + MM_GEMM_RB - 1) / MM_GEMM_RB, ncb = (ncol_fwd + MM_GEMM_NB - 1) / MM_GEMM_NB;
  const long long nwork_ll = (long long)L * ncb * MM_MOM_SPLIT * nrb;
  if (nwork_ll <= 0 || nwork_ll > 0x7fffffffLL) return MM_E_DIM;
  const int nwork = (int)nwork_ll;
  // collapse: not with the forced worst tier (bench.py --recipe worst times the dense path)
  // ... nor where no forward reduce follows (MM_ISTAGE_NO_M56: s12 is then only the scale of the backward's route decision, for
  // which orders 0..2 are ample)
  const int allow = (flags & (MM_FORCE_WORST_TIER | MM_ISTAGE_NO_M56)) ? 0 : 1;
  const int col_deg3 = mm_mono_offset(3, d);                // first column of a cubic monomial
  // rows of every latent's GEMM ordered with the collapsed items first (none collapse: no cubic / quartic block at all)
  const bool some = allow && mm_moment_deg(d) >= 4;
  if (some) {
    hipLaunchKernelGGL(k_wmom_perm, dim3(L), dim3(1024), 0, stream, (const unsigned int*)(ws + wl.amaxc),
                       (const double*)(packed + ml.zmax2), L, B, wl.Po, (int*)(ws + wl.gperm));
    hipError_t ep = hipGetLastError();
    if (ep != hipSuccess) return (int)ep;
  }
  hipLaunchKernelGGL(k_wmom_gemm, dim3(nwork), dim3(256), 0, stream, (const double*)(ws + wl.whR),
                     (const double*)(ws + wl.whC), Zm, ml.KMp, L, wl.Mp, B, wl.Po, nrb, ncb, nwork, col_deg3,
                     some ? (const unsigned int*)(ws + wl.amaxc) : (const unsigned int*)nullptr,
                     (const double*)(packed + ml.zmax2), some ? (const int*)(ws + wl.gperm) : (const int*)nullptr, mom);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  const int off1 = mm_mono_offset(1, d), off2 = mm_mono_offset(2, d), off3 = mm_mono_offset(3, d), off4 = mm_mono_offset(4, d);
#define MM_SPOLY(DK_, LB_)                                                                                                       \
  do {                                                                                                                          \
    const int dn_ = 1 << (LB_ * deg);                                                                                           \
    const int up_ = dn_ >> LB_;                                                                                                 \
    const size_t shm_ = (size_t)(dn_ + (dn_ >> 5) + up_ + (up_ >> 5) + 2 * ml.KMp + DK_ * DK_ + DK_ + 8) * sizeof(double);      \
    hipLaunchKernelGGL((k_spoly<DK_, LB_>), dim3(wl.Po, B), dim3(256), shm_, stream, (const double*)mom, ml.KMp,                \
                       (const double*)(ws + wl.pairmat), (const double*)(packed + ml.zbar), (const double*)(packed + ml.zmax2), \
                       (const unsigned int*)(ws + wl.amax), (const unsigned int*)(ws + wl.amaxc), (const float*)mu_f32, L, d, wl.P, deg, allow, off1, off2, off3, off4, \
                       (const short*)(packed + ml.rtab),                                                                        \
                       (double)MM_C6_C0, (double)MM_C6_C1, (double*)(ws + wl.s12));                                             \
  } while (0)
  if (d <= 8) MM_SPOLY(8, 3); else MM_SPOLY(32, 5);
#undef MM_SPOLY
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (flags & MM_ISTAGE_NO_M56) {
    // (mm_common.h: no forward reduce follows on this workspace) -- poison instead of compute: all-ones doubles are NaN
    if (mm_moment56_cols(d) <= 0 || wl.Po <= 0) return 0;
    const hipError_t em = hipMemsetAsync(ws + wl.s56, 0xFF, (size_t)B * wl.Po * sizeof(double), stream);
    return em == hipSuccess ? 0 : (int)em;
  }
  // orders 5 and 6 of the collapsed items (f32 moments on the bf16 matrix pipe: mm_moments6.hip) -> s56, estS
  return mm_launch_moments56(packed, ml, ws, wl, B, L, d, some ? 1 : 0, stream);
}

// ---------------------------------------------------------------------------------------------
// mm_offdiag_stats: how many (b, off-diagonal pair) items of the last mm_q_forward are collapsed (bench.py reports
// it with the timing: the reduce kernels' time depends on the regime).  out: device int32[6] = {collapsed in every row group
// (overall bound <= 1/2), total, wholly inside the collapsed range (no tile work at all), routed, PARTLY collapsed (some row
// groups: mm_mono.h), collapsed 64-row groups over all items}.
// ---------------------------------------------------------------------------------------------
__global__ void k_offdiag_stats(const unsigned int* __restrict__ amax, const double* __restrict__ zmax2,
                                const unsigned char* __restrict__ gflag, int ng, int L, int Po, int n,
                                int32_t* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  bool c = false, in = false, part = false;
  int groups = 0;
  if (idx < n) {
    int a, a2;
    mm_decode_pair_m(L + idx % Po, L, a, a2);
    const float b2 = mm_collapse_bound2(amax[idx], zmax2[a2]);
    c = b2 <= MM_COLLAPSE_BOUND2;
    in = b2 <= MM_INSIDE_BOUND2;
    part = !c && mm_item_collapsed(amax[n + idx]);          // (amaxc right behind amax: at least one collapsed row group)
    for (int g = 0; g < ng; ++g) groups += gflag[(size_t)idx * ng + g] ? 1 : 0;
  }
  const unsigned long long m = __ballot(c), mi = __ballot(in), mp = __ballot(part);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (int)__popcll(m));
  if ((threadIdx.x & 63) == 0 && mi) atomicAdd(out + 2, (int)__popcll(mi));
  if ((threadIdx.x & 63) == 0 && mp) atomicAdd(out + 4, (int)__popcll(mp));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) groups += __shfl_down(groups, off, 64);
  if ((threadIdx.x & 63) == 0 && groups) atomicAdd(out + 5, groups);
  if (idx == 0) out[1] = n;
}

extern "C" int mm_offdiag_stats(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B, int flags,
                                const void* workspace, size_t workspace_bytes, int32_t* out, void* stream) {
  if (!packed || !workspace || !out || L <= 0 || M <= 0 || d <= 0 || d > MM_DMAX || B <= 0) return MM_E_ARG;
  const MMModelLayout ml = mm_model_layout(L, M, d, dtype, 1);
  const MMWorkspaceLayout wl = mm_workspace_layout(B, L, M, d, dtype, flags);
  if (packed_bytes < ml.Cm || workspace_bytes < wl.total) return MM_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, 6 * sizeof(int32_t), s);
  if (e != hipSuccess) return (int)e;
  if (dtype == MM_F32 && wl.Po > 0) {       // out[3]: items the last forward on this workspace re-reduced in f64 (mm_route.hip)
    e = hipMemcpyAsync(out + 3, (const char*)workspace + wl.rcount + 4, sizeof(int32_t), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
  }
  if (dtype != MM_F32 || wl.Po == 0 || mm_moment_deg(d) < 4 || (flags & MM_FORCE_WORST_TIER)) return 0;   // nothing collapses
  const int n = B * wl.Po;
  hipLaunchKernelGGL(k_offdiag_stats, dim3((n + 255) / 256), dim3(256), 0, s, (const unsigned int*)((const char*)workspace + wl.amax),
                     (const double*)((const char*)packed + ml.zmax2), (const unsigned char*)((const char*)workspace + wl.gflag),
                     wl.Mp / MM_GROUP_ROWS, L, wl.Po, n, out);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
