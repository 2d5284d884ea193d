// Forward arithmetic of the composed rollout step for SMALL models, written for an execution context (mm_adjoint.h):
// ONE workgroup runs a whole step -- encoder, policy match, NormalCDF head, drift match, forward_sde bookkeeping, Euler,
// cost -- with the state and every intermediate in LDS (k_rollout_small in mm_rollout_small.hip; SURVEY.md section 7 step 6:
// "at B = 1 the GPU is latency-bound -- needs graph capture / persistent kernel").  The multi-launch path
// (mm_rollout_composed_t) pays nine kernel boundaries and ~25 dependent global round trips per step at cartpole sizes.
//   mms_encode_fwd  moment_matching/components.py:19-57 + maths.py:143-176
//   mms_gp_fwd      moment_matching/models.py:200-299 (+ utils/kernel_expectation.py:72-247), the centred fused form of
//                   DESIGN.md section 2.1, any L (policy: L = 1 without model uncertainty; drift: L = nx with C)
//   mms_head_fwd    moment_matching/bijectors.py:39-69 + gaussian.py:53-83
//   mms_step_fwd    dynamics/forward_sde.py:105-131 + solvers.py:110-135
//   mms_cost_fwd    components.py:26-37
// All f64.  The same functions compile for one host thread (tests/hostcheck) and are checked there against the oracle.
#pragma once
#include "mm_adjoint.h"

// 48-point Gauss-Legendre nodes / weights on [-1, 1] (the quadrature of Owen's T in gpflowpilco_amd/special.py)
__host__ __device__ inline double mms_gl48_x(int k) {
  constexpr double X[48] = {-9.98771007252426068e-01, -9.93530172266350764e-01, -9.84124583722826851e-01, -9.70591592546247273e-01, -9.52987703160430910e-01, -9.31386690706554332e-01, -9.05879136715569633e-01, -8.76572020274247854e-01, -8.43588261624393487e-01, -8.07066204029442624e-01, -7.67159032515740358e-01, -7.24034130923814634e-01, -6.77872379632663891e-01, -6.28867396776513599e-01, -5.77224726083972683e-01, -5.23160974722232996e-01, -4.66902904750958414e-01, -4.08686481990716721e-01, -3.48755886292160755e-01, -2.87362487355455554e-01, -2.24763790394689050e-01, -1.61222356068891709e-01, -9.70046992094626970e-02, -3.23801709628693674e-02, 3.23801709628693674e-02, 9.70046992094626970e-02, 1.61222356068891709e-01, 2.24763790394689050e-01, 2.87362487355455554e-01, 3.48755886292160755e-01, 4.08686481990716721e-01, 4.66902904750958414e-01, 5.23160974722232996e-01, 5.77224726083972683e-01, 6.28867396776513599e-01, 6.77872379632663891e-01, 7.24034130923814634e-01, 7.67159032515740358e-01, 8.07066204029442624e-01, 8.43588261624393487e-01, 8.76572020274247854e-01, 9.05879136715569633e-01, 9.31386690706554332e-01, 9.52987703160430910e-01, 9.70591592546247273e-01, 9.84124583722826851e-01, 9.93530172266350764e-01, 9.98771007252426068e-01};
  return X[k];
}
__host__ __device__ inline double mms_gl48_w(int k) {
  constexpr double W[48] = {3.15334605230917957e-03, 7.32755390127649234e-03, 1.14772345792349736e-02, 1.55793157229429276e-02, 1.96161604573552965e-02, 2.35707608393240925e-02, 2.74265097083568818e-02, 3.11672278327983394e-02, 3.47772225647706573e-02, 3.82413510658306741e-02, 4.15450829434645535e-02, 4.46745608566940997e-02, 4.76166584924902839e-02, 5.03590355538542783e-02, 5.28901894851934867e-02, 5.51995036999840538e-02, 5.72772921004029295e-02, 5.91148396983954827e-02, 6.07044391658935825e-02, 6.20394231598924636e-02, 6.31141922862537841e-02, 6.39242385846479494e-02, 6.44661644359498381e-02, 6.47376968126836816e-02, 6.47376968126836816e-02, 6.44661644359498381e-02, 6.39242385846479494e-02, 6.31141922862537841e-02, 6.20394231598924636e-02, 6.07044391658935825e-02, 5.91148396983954827e-02, 5.72772921004029295e-02, 5.51995036999840538e-02, 5.28901894851934867e-02, 5.03590355538542783e-02, 4.76166584924902839e-02, 4.46745608566940997e-02, 4.15450829434645535e-02, 3.82413510658306741e-02, 3.47772225647706573e-02, 3.11672278327983394e-02, 2.74265097083568818e-02, 2.35707608393240925e-02, 1.96161604573552965e-02, 1.55793157229429276e-02, 1.14772345792349736e-02, 7.32755390127649234e-03, 3.15334605230917957e-03};
  return W[k];
}

// ---- encoder ------------------------------------------------------------------------------------------------------------
// (m [nx], S [nx, nx]) -> me [ne], See [ne, ne], Sxe [nx, ne].   sm: 2 na + 4 na^2 + 2 nx na doubles.
__host__ __device__ inline int mms_encode_scratch(int nx, int na) { return 2 * na + 4 * na * na + 2 * nx * na; }

MMA_FN void mms_encode_fwd(Ctx c, const MMComposeDims& D, const double* m, const double* S, double* me, double* See,
                           double* Sxe, double* sm) {
  const int lane = c.lane(), nl = c.nl();
  const int nx = D.nx, na = D.na, ne = D.ne, n2 = 2 * na;
  double* s1 = sm; double* c1 = s1 + na; double* Syy = c1 + na; double* Sxy = Syy + n2 * n2;
  for (int i = lane; i < na; i += nl) {
    const int r = D.active[i];
    const double ev = exp(-0.5 * S[r * nx + r]);
    s1[i] = ev * sin(m[r]); c1[i] = ev * cos(m[r]);
  }
  c.sync();
  for (int idx = lane; idx < na * na; idx += nl) {
    const int i = idx / na, j = idx - i * na;
    const int ri = D.active[i], rj = D.active[j];
    const double ai = m[ri], aj = m[rj], vi = S[ri * nx + ri], vj = S[rj * nx + rj];
    const double sij = 0.5 * (S[ri * nx + rj] + S[rj * nx + ri]);
    const double A = exp(-0.5 * (vi + vj) - sij), Bm = exp(-0.5 * (vi + vj) + sij);
    const double Acos = A * cos(ai + aj), Bcos = Bm * cos(ai - aj);
    const double sc = 0.5 * (sin(ai) * cos(aj) * (Bm + A) - sin(aj) * cos(ai) * (Bm - A));
    Syy[i * n2 + j] = 0.5 * (Bcos - Acos) - s1[i] * s1[j];
    Syy[(na + i) * n2 + na + j] = 0.5 * (Bcos + Acos) - c1[i] * c1[j];
    Syy[i * n2 + na + j] = sc - s1[i] * c1[j];
    Syy[(na + j) * n2 + i] = sc - s1[i] * c1[j];
  }
  for (int idx = lane; idx < nx * na; idx += nl) {
    const int r = idx / na, j = idx - r * na;
    const double sra = S[r * nx + D.active[j]];
    Sxy[r * n2 + j] = sra * c1[j];
    Sxy[r * n2 + na + j] = -sra * s1[j];
  }
  c.sync();
  for (int k = lane; k < ne; k += nl) me[k] = k < na ? s1[k] : k < n2 ? c1[k - na] : m[D.inactive[k - n2]];
  for (int idx = lane; idx < ne * ne; idx += nl) {
    const int i = idx / ne, j = idx - i * ne;
    double v;
    if (i < n2 && j < n2) v = Syy[i * n2 + j];
    else if (i >= n2 && j >= n2) v = S[D.inactive[i - n2] * nx + D.inactive[j - n2]];
    else if (i >= n2) v = Sxy[D.inactive[i - n2] * n2 + j];
    else v = Sxy[D.inactive[j - n2] * n2 + i];
    See[idx] = v;
  }
  for (int idx = lane; idx < nx * ne; idx += nl) {
    const int r = idx / ne, k = idx - r * ne;
    Sxe[idx] = k < n2 ? Sxy[r * n2 + k] : S[r * nx + D.inactive[k - n2]];
  }
  c.sync();
}

// ---- one GP moment match, small model ------------------------------------------------------------------------------------
// model: Z [L][M][d], beta [L][M], ls2 [L][d], var [L], meanc [L], Cm [L][ldc][ldc] | null (no model uncertainty).
// (mu [d], Sigma [d, d], lower triangle read) -> f1 [L], Sff [L, L], cross [d, L] (= Sigma^-1 Cov(x, f)).
__host__ __device__ inline int mms_gp_scratch(int L, int M, int d, int ngroups) {
  const int P = L * (L + 1) / 2, dd = d * d;
  return dd + 2 * L * dd + 2 * L + P * dd + P + ngroups * 2 * dd + L * M * d + 3 * L * M + M * d + 2 * M + ngroups * 8 + 8;
}

template <class Ctx, int DK>
__host__ __device__ inline void mms_gp_fwd(Ctx c, int L, int M, int d, const double* Z, const double* beta, const double* ls2, const double* var,
                       const double* meanc, const double* Cm, int ldc, const double* mu, const double* Sigma,
                       double* f1, double* Sff, double* cross, double* sm, bool* ok) {
  const int lane = c.lane(), nl = c.nl(), dd = d * d, P = L * (L + 1) / 2, ng = c.ngroups(), grp = c.group();
  double* Sg = sm;                           // [d][d]
  double* Pa = Sg + dd;                      // [L][d][d]
  double* Ea = Pa + L * dd;                  // [L][d][d]
  double* lognorm = Ea + L * dd;             // [L]
  double* ldA = lognorm + L;                 // [L]
  double* Tp = ldA + L;                      // [P][d][d]
  double* cst = Tp + P * dd;                 // [P]
  double* wsc = cst + P;                     // [ngroups][2][d][d]  per-group work matrices
  double* sv = wsc + ng * 2 * dd;            // [L][M][d]  zeta / Lambda_a
  double* wv = sv + L * M * d;               // [L][M]     w = beta q
  double* qv = wv + L * M;                   // [L][M]     q
  double* r1 = qv + L * M;                   // [L][M]     zeta^T E_a zeta
  double* uv = r1 + L * M;                   // [M][d]     T s_i      (current pair, row side)
  double* rho = uv + M * d;                  // [M]
  double* gam = rho + M;                     // [M]
  double* red = gam + M;                     // [ngroups][d + 2]
  for (int idx = lane; idx < dd; idx += nl) {
    const int i = idx / d, j = idx - i * d;
    Sg[idx] = i >= j ? Sigma[i * d + j] : Sigma[j * d + i];
  }
  c.sync();
  c.stamp(0);
  // ---- d x d items: the L latents, then the P pairs; one group (wave) per item -------------------------------------------
  auto sub = c.sub();
  const int sl = sub.lane(), snl = sub.nl();
  double* Aw = wsc + (size_t)grp * 2 * dd; double* Yw = Aw + dd;
  for (int a = grp; a < L; a += ng) {
    const double* la = ls2 + a * d;
    for (int idx = sl; idx < dd; idx += snl) { const int i = idx / d, j = idx - i * d; Aw[idx] = Sg[idx] + (i == j ? la[i] : 0.0); }
    sub.sync();
    const double ld = mma_spd_inverse(sub, Aw, Yw, d, d, ok);
    double slog = 0.0;
    for (int k = 0; k < d; ++k) slog += log(la[k]);
    if (sl == 0) { lognorm[a] = log(var[a]) + 0.5 * slog - 0.5 * ld; ldA[a] = ld; }
    for (int idx = sl; idx < dd; idx += snl) {
      const int i = idx / d, j = idx - i * d;
      double s1 = 0.0, t1 = 0.0;
      for (int k = 0; k < d; ++k) { s1 += Sg[i * d + k] * Aw[k * d + j]; t1 += Sg[j * d + k] * Aw[k * d + i]; }
      Pa[a * dd + idx] = Aw[idx];
      Ea[a * dd + idx] = 0.5 * (s1 / la[i] + t1 / la[j]);
    }
    sub.sync();
  }
  c.sync();
  for (int p = grp; p < P; p += ng) {
    int a, a2; mma_decode_pair(p, L, a, a2);
    const double* la = ls2 + a * d; const double* lb = ls2 + a2 * d;
    for (int idx = sl; idx < dd; idx += snl) {
      const int i = idx / d, j = idx - i * d;
      Aw[idx] = Sg[idx] + (i == j ? la[i] * lb[i] / (la[i] + lb[i]) : 0.0);
    }
    sub.sync();
    const double ldS = mma_spd_inverse(sub, Aw, Yw, d, d, ok);
    for (int idx = sl; idx < dd; idx += snl) {             // Yw = V S0 Sigma
      const int i = idx / d, j = idx - i * d;
      double s = 0.0;
      for (int k = 0; k < d; ++k) s += Aw[i * d + k] * Sg[k * d + j];
      Yw[idx] = (la[i] * lb[i] / (la[i] + lb[i])) * s;
    }
    sub.sync();
    for (int idx = sl; idx < dd; idx += snl) { const int i = idx / d, j = idx - i * d; Tp[p * dd + idx] = 0.5 * (Yw[i * d + j] + Yw[j * d + i]); }
    double lsum = 0.0;
    for (int k = 0; k < d; ++k) lsum += log(la[k] + lb[k]);
    if (sl == 0) cst[p] = -0.5 * ldS - 0.5 * lsum + 0.5 * ldA[a] + 0.5 * ldA[a2];
    sub.sync();
  }
  c.sync();
  c.stamp(1);
  // ---- per (latent, centre): q, w, the scaled offsets, f1 and the cross term ---------------------------------------------
  for (int a = 0; a < L; ++a) {
    const double* la = ls2 + a * d; const double* Pm = Pa + a * dd; const double* Em = Ea + a * dd;
    // (DK: compile-time bound of d, so that z and the partial sums stay in registers)
    double accz[DK], accw = 0.0;
#pragma unroll
    for (int k = 0; k < DK; ++k) accz[k] = 0.0;
    for (int m = lane; m < M; m += nl) {
      double z[DK];
#pragma unroll
      for (int k = 0; k < DK; ++k) z[k] = k < d ? Z[((size_t)a * M + m) * d + k] - mu[k] : 0.0;
      double maha = 0.0, rq = 0.0;
#pragma unroll
      for (int i = 0; i < DK; ++i) {
        if (i < d) {
          double tp = 0.0, te = 0.0;
#pragma unroll
          for (int k = 0; k < DK; ++k) if (k < d) { tp = fma(Pm[i * d + k], z[k], tp); te = fma(Em[i * d + k], z[k], te); }
          maha = fma(z[i], tp, maha); rq = fma(z[i], te, rq);
          sv[((size_t)a * M + m) * d + i] = z[i] / la[i];
        }
      }
      const double q = exp(lognorm[a] - 0.5 * maha), w = beta[(size_t)a * M + m] * q;
      qv[a * M + m] = q; wv[a * M + m] = w; r1[a * M + m] = rq;
#pragma unroll
      for (int k = 0; k < DK; ++k) accz[k] = fma(w, z[k], accz[k]);
      accw += w;
    }
    static_assert(DK <= 9, "two reductions of five values cover d + 1 <= 10");
    double v0[5], v1[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      v0[k] = k < DK ? accz[k] : 0.0;
      v1[k] = (5 + k < DK) ? accz[5 + k < DK ? 5 + k : 0] : 0.0;
    }
    v1[4] = accw;                          // (DK <= 9: slot 9 is free)
    c.reduce(v0, red);
    c.reduce(v1, red);
    double acc[DK + 1];
#pragma unroll
    for (int k = 0; k < DK; ++k) acc[k] = k < 5 ? v0[k] : v1[k - 5];
    acc[DK] = v1[4];
    if (lane == 0) f1[a] = acc[DK] + (meanc ? meanc[a] : 0.0);
    for (int i = lane; i < d; i += nl) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < DK; ++k) if (k < d) s = fma(Pm[i * d + k], acc[k], s);
      cross[i * L + a] = s;
    }
  }
  c.sync();
  c.stamp(2);
  // ---- pairs: Sff_aa' = sum_ij w_i expm1(delta_ij) w'_j  (+ [a = a'] var_a + sum_ij C_ij q_i e^{delta_ij} q_j) -------------
  for (int p = 0; p < P; ++p) {
    int a, a2; mma_decode_pair(p, L, a, a2);
    const double* T = Tp + p * dd;
    const double* sa = sv + (size_t)a * M * d; const double* sb = sv + (size_t)a2 * M * d;
    for (int t = lane; t < 2 * M; t += nl) {               // rows of latent a, then (off-diagonal pairs) columns of a'
      const int m = t < M ? t : t - M;
      if (t >= M && a2 == a) continue;
      const double* s = (t < M ? sa : sb) + (size_t)m * d;
      double quad = 0.0;
#pragma unroll
      for (int i = 0; i < DK; ++i) {
        if (i < d) {
          double u = 0.0;
#pragma unroll
          for (int k = 0; k < DK; ++k) if (k < d) u = fma(T[i * d + k], s[k], u);
          quad = fma(s[i], u, quad);
          if (t < M) uv[m * d + i] = u;
        }
      }
      if (t < M) { rho[m] = -0.5 * (r1[a * M + m] - quad); if (a2 == a) gam[m] = rho[m]; }
      else gam[m] = -0.5 * (r1[a2 * M + m] - quad);
    }
    c.sync();
    c.stamp(3);
    const bool withC = (a == a2) && (Cm != nullptr);
    const double* wa = wv + a * M; const double* wb = wv + a2 * M; const double* qa = qv + a * M;
    const double* Ca = withC ? Cm + (size_t)a * ldc * ldc : nullptr;
    double accB = 0.0, accC = 0.0;
    const double cp_ = cst[p];
    {
      // entry (i, j) of this lane advances by nl per iteration: carried, no integer division per entry
      const int di = nl / M, dj = nl - di * M;
      int i = lane / M, j = lane - i * M;
      for (int idx = lane; idx < M * M; idx += nl) {
        double delta = cp_ + rho[i] + gam[j];
#pragma unroll
        for (int k = 0; k < DK; ++k) if (k < d) delta = fma(uv[i * d + k], sb[(size_t)j * d + k], delta);
        const double E = expm1(fmin(delta, MM_EXP_CAP_F64));                 // (mm_common.h: exponent caps)
        accB = fma(wa[i] * E, wb[j], accB);
        if (withC) accC = fma(Ca[(size_t)i * ldc + j] * qa[i] * (E + 1.0), qa[j], accC);
        i += di; j += dj;
        if (j >= M) { j -= M; ++i; }
      }
    }
    c.stamp(4);
    double v2[2] = {accB, accC};
    c.reduce(v2, red);
    if (lane == 0) {
      const double val = v2[0] + (withC ? var[a] + v2[1] : 0.0);
      Sff[a * L + a2] = val; Sff[a2 * L + a] = val;
    }
    c.sync();
    c.stamp(5);
  }
}

// ---- NormalCDF head + joint ---------------------------------------------------------------------------------------------
// (me, See, pf1, pSff, pcross) -> md [nd], Sdd [nd, nd], cp [ne].   sm: 64 + ne doubles.
MMA_FN void mms_head_fwd(Ctx c, int ne, double scale, double shift, const double* me, const double* See, double pf1, double pSff,
                         const double* pcross, double* md, double* Sdd, double* cp, double* sm) {
  const int lane = c.lane(), nl = c.nl(), nd = ne + 1;
  double* gl = sm; double* Seu = sm + 64;
  const double vx = pSff > 0.0 ? pSff : 0.0;
  const double isq = 1.0 / sqrt(vx + 1.0), z = isq * pf1;
  const double aa = 1.0 / sqrt(1.0 + 2.0 * vx);
  // Owen's T(z, aa): 48-point Gauss-Legendre on [0, aa] (the quadrature gpflowpilco_amd/special.py and k_compose_policy use)
  for (int k = lane; k < 48; k += nl) {
    const double t = 0.5 * aa * (mms_gl48_x(k) + 1.0);
    gl[k] = mms_gl48_w(k) * exp(-0.5 * z * z * (1.0 + t * t)) / (1.0 + t * t);
  }
  c.sync();
  double part = 0.0;
  for (int k = 0; k < 48; ++k) part += gl[k];
  const double owen = 0.5 * aa * part * MMA_INV_2PI;
  const double y1 = 0.5 * erfc(-z * 0.70710678118654752440);
  const double y2 = y1 - 2.0 * owen;
  const double head_pre = isq * MMA_INV_SQRT_2PI * exp(-0.5 * z * z) * scale;
  for (int k = lane; k < ne; k += nl) cp[k] = pcross[k] * head_pre;
  c.sync();
  for (int k = lane; k < ne; k += nl) {
    double s = 0.0;
    for (int l = 0; l < ne; ++l) s = fma(See[k * ne + l], cp[l], s);
    Seu[k] = s;
  }
  c.sync();
  for (int k = lane; k < nd; k += nl) md[k] = k < ne ? me[k] : scale * (y1 + shift);
  for (int idx = lane; idx < nd * nd; idx += nl) {
    const int i = idx / nd, j = idx - i * nd;
    Sdd[idx] = (i < ne && j < ne) ? See[i * ne + j] : i < ne ? Seu[i] : j < ne ? Seu[j] : scale * scale * (y2 - y1 * y1);
  }
  c.sync();
}

// ---- forward_sde bookkeeping + Euler --------------------------------------------------------------------------------------
// m, S updated in place.   sm: nx (nd + nx) doubles.
MMA_FN void mms_step_fwd(Ctx c, const MMComposeDims& D, double dt, const double* Sxe, const double* cp, const double* Sdd,
                         const double* df1, const double* dSff, const double* dcross, double* m, double* S, double* sm) {
  const int lane = c.lane(), nl = c.nl();
  const int nx = D.nx, na = D.na, ne = D.ne, nd = D.nd, n2 = 2 * na;
  double* Sxd = sm; double* Sxf = Sxd + nx * nd;
  for (int idx = lane; idx < nx * nd; idx += nl) {
    const int r = idx / nd, k = idx - r * nd;
    const int sl = D.slot[r];
    double v;
    if (sl < na) {
      if (k < ne) v = Sxe[r * ne + k];
      else { double s = 0.0; for (int l = 0; l < ne; ++l) s = fma(Sxe[r * ne + l], cp[l], s); v = s; }
    } else {
      v = Sdd[(n2 + (sl - na)) * nd + k];
    }
    Sxd[idx] = v;
  }
  c.sync();
  for (int idx = lane; idx < nx * nx; idx += nl) {
    const int r = idx / nx, cc = idx - r * nx;
    double s = 0.0;
    for (int k = 0; k < nd; ++k) s = fma(Sxd[r * nd + k], dcross[k * nx + cc], s);
    Sxf[idx] = s;
  }
  c.sync();
  for (int idx = lane; idx < nx * nx; idx += nl) {
    const int r = idx / nx, cc = idx - r * nx;
    S[idx] = S[idx] + dt * (Sxf[r * nx + cc] + Sxf[cc * nx + r]) + dt * dt * dSff[idx];
  }
  for (int i = lane; i < nx; i += nl) m[i] += dt * df1[i];
  c.sync();
}

// ---- expected saturating cost ----------------------------------------------------------------------------------------------
// sm: n (n + 3) doubles.  Returns the cost (every lane).
MMA_FN double mms_cost_fwd(Ctx c, int n, const double* mean, const double* cov, const double* target, const double* W, double* sm) {
  const int lane = c.lane(), nl = c.nl(), ld = n + 1;
  double* Aug = sm; double* e = Aug + n * ld; double* We = e + n;
  for (int i = lane; i < n; i += nl) e[i] = mean[i] - target[i];
  for (int idx = lane; idx < n * n; idx += nl) {           // I + S W
    const int i = idx / n, j = idx - i * n;
    double s = i == j ? 1.0 : 0.0;
    for (int k = 0; k < n; ++k) s = fma(cov[i * n + k], W[k * n + j], s);
    Aug[i * ld + j] = s;
  }
  c.sync();
  for (int i = lane; i < n; i += nl) Aug[i * ld + n] = e[i];
  const double det = mma_solve_pivot(c, Aug, n, 1, ld);    // y = (I + S W)^-1 e
  for (int i = lane; i < n; i += nl) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s = fma(W[i * n + j], Aug[j * ld + n], s);
    We[i] = s;
  }
  c.sync();
  double dist2 = 0.0;
  for (int i = 0; i < n; ++i) dist2 = fma(e[i], We[i], dist2);
  const double cost = -exp(-0.5 * dist2) / sqrt(det);
  c.sync();
  return cost;
}

// ---- the whole H-step rollout of one batch element ------------------------------------------------------------------------
struct MMSmallModel {
  int L, M, d, ldc;
  const double *Z, *beta, *ls2, *var, *meanc, *Cm;      // Cm null: evaluated without model uncertainty
};

__host__ __device__ inline int mms_max_int(int a, int b) { return a > b ? a : b; }
// doubles of scratch mms_rollout needs (ngroups: waves of the workgroup; 1 on the host)
__host__ __device__ inline int mms_rollout_scratch(int nx, int na, int Md, int Mp_, int ngroups) {
  const int ne = nx + na, nd = ne + 1;
  int wk = mms_gp_scratch(nx, Md, nd, ngroups);
  wk = mms_max_int(wk, mms_gp_scratch(1, Mp_, ne, ngroups));
  wk = mms_max_int(wk, mms_encode_scratch(nx, na));
  wk = mms_max_int(wk, nx * (nd + nx));
  wk = mms_max_int(wk, ne * (ne + 3));
  wk = mms_max_int(wk, 64 + ne);
  return nx + nx * nx + ne + ne * ne + nx * ne + 2 + 2 * ne + nd + nd * nd + nx + nx * nx + nd * nx + ne * ne + ne + wk + 16;
}

// mx [nx], Sxx [nx, nx]: this element's state (updated in place); cost: this element's column of cost [H][B] (stride
// cost_stride) or null; traj_mu / traj_S likewise (strides B nx, B nx nx) or null.
template <class Ctx, typename T, int DK>
__host__ __device__ inline void mms_rollout(Ctx c, const MMComposeDims& D, int H, double dt, double scale, double shift,
                                            const MMSmallModel& drift, const MMSmallModel& pol, const T* target, const T* precis,
                                            T* mx, T* Sxx, T* cost, size_t cost_stride, T* traj_mu, size_t tm_stride,
                                            T* traj_S, size_t tS_stride, double* sm, bool* ok) {
  const int lane = c.lane(), nl = c.nl();
  const int nx = D.nx, ne = D.ne, nd = D.nd;
  double* m = sm; double* S = m + nx; double* me = S + nx * nx; double* See = me + ne; double* Sxe = See + ne * ne;
  double* pout = Sxe + nx * ne;              // pf1, pSff
  double* pcross = pout + 2; double* cp = pcross + ne; double* md = cp + ne; double* Sdd = md + nd;
  double* df1 = Sdd + nd * nd; double* dSff = df1 + nx; double* dcross = dSff + nx * nx;
  double* W = dcross + nd * nx; double* tgt = W + ne * ne; double* wk = tgt + ne;
  for (int i = lane; i < nx; i += nl) m[i] = (double)mx[i];
  for (int i = lane; i < nx * nx; i += nl) S[i] = (double)Sxx[i];
  if (cost) {
    for (int i = lane; i < ne * ne; i += nl) W[i] = (double)precis[i];
    for (int i = lane; i < ne; i += nl) tgt[i] = (double)target[i];
  }
  c.sync();
  mms_encode_fwd(c, D, m, S, me, See, Sxe, wk);
  for (int h = 0; h < H; ++h) {
    mms_gp_fwd<Ctx, DK>(c, 1, pol.M, ne, pol.Z, pol.beta, pol.ls2, pol.var, pol.meanc, nullptr, 0, me, See, pout, pout + 1, pcross, wk, ok);
    c.sync();
    mms_head_fwd(c, ne, scale, shift, me, See, pout[0], pout[1], pcross, md, Sdd, cp, wk);
    c.stamp(6);
    mms_gp_fwd<Ctx, DK>(c, drift.L, drift.M, nd, drift.Z, drift.beta, drift.ls2, drift.var, drift.meanc, drift.Cm, drift.ldc, md, Sdd,
                        df1, dSff, dcross, wk, ok);
    c.sync();
    mms_step_fwd(c, D, dt, Sxe, cp, Sdd, df1, dSff, dcross, m, S, wk);
    c.stamp(7);
    if (traj_mu) for (int i = lane; i < nx; i += nl) traj_mu[(size_t)h * tm_stride + i] = (T)m[i];
    if (traj_S) for (int i = lane; i < nx * nx; i += nl) traj_S[(size_t)h * tS_stride + i] = (T)S[i];
    mms_encode_fwd(c, D, m, S, me, See, Sxe, wk);
    c.stamp(8);
    if (cost) {
      const double cv = mms_cost_fwd(c, ne, me, See, tgt, W, wk);
      if (lane == 0) cost[(size_t)h * cost_stride] = (T)cv;
    }
    c.stamp(9);
  }
  for (int i = lane; i < nx; i += nl) mx[i] = (T)m[i];
  for (int i = lane; i < nx * nx; i += nl) Sxx[i] = (T)S[i];
}
