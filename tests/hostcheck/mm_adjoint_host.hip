// CPU build of the adjoint arithmetic of csrc/mm_adjoint.h (MMAHostCtx: one host thread, no barriers) -- TEST
// INFRASTRUCTURE ONLY.  tests/test_adjoint_host.py loads it to check the hand-derived adjoints against autograd and
// finite differences without a GPU; nothing under gpflowpilco_amd/ loads or links it, and the product library holds
// no host path.
#include <algorithm>
#include <vector>
#include "../../gpflowpilco_amd/csrc/mm_adjoint.h"

static MMComposeDims dims(int nx, int na, const int32_t* act) { MMComposeDims D; mm_compose_dims(nx, na, act, D); return D; }

extern "C" void hc_encode_bwd(int nx, int na, const int32_t* act, const double* m, const double* S, const double* gme,
                              const double* gSee, const double* gSxe, double* gm, double* gS) {
  MMComposeDims D = dims(nx, na, act);
  std::vector<double> sm(mma_encode_bwd_scratch(nx, na) + 8);
  mma_encode_bwd(MMAHostCtx(), D, m, S, gme, gSee, gSxe, gm, gS, sm.data());
}

extern "C" double hc_cost_bwd(int n, const double* mean, const double* cov, const double* target, const double* W, double gc,
                              double* gmean, double* gcov) {
  std::vector<double> sm(mma_cost_bwd_scratch(n) + 8);
  return mma_cost_bwd(MMAHostCtx(), n, mean, cov, target, W, gc, gmean, gcov, sm.data());
}

extern "C" void hc_step_bwd(int nx, int na, const int32_t* act, double dt, const double* Sxe, const double* cp, const double* Sdd,
                            const double* dcross, const double* gm1, const double* gS1, double* gSxe, double* gcp, double* gSdd,
                            double* gdf1, double* gdSff, double* gdcross) {
  MMComposeDims D = dims(nx, na, act);
  std::vector<double> sm(mma_step_bwd_scratch(nx, D.nd) + 8);
  mma_step_bwd(MMAHostCtx(), D, dt, Sxe, cp, Sdd, dcross, gm1, gS1, gSxe, gcp, gSdd, gdf1, gdSff, gdcross, sm.data());
}

extern "C" void hc_head_bwd(int ne, double scale, double shift, double pf1, double pSff, const double* pcross, const double* See,
                            const double* gmd, const double* gSdd, const double* gcp, double* gme, double* gSee, double* gpcross,
                            double* gp2) {
  std::vector<double> sm(ne + 8);
  mma_head_bwd(MMAHostCtx(), ne, scale, shift, pf1, pSff, pcross, See, gmd, gSdd, gcp, gme, gSee, gpcross, sm.data());
  gp2[0] = sm[ne]; gp2[1] = sm[ne + 1];
}

extern "C" int hc_policy_small_bwd(int M, int d, const double* Z, const double* beta, const double* ls2, double var, const double* mu,
                                   const double* Sigma, double gf1, double gSff, const double* gcross, double* gmu, double* gSig,
                                   double* gpar) {
  std::vector<double> sm(mma_policy_small_bwd_scratch(M, d, 1) + 8);
  bool ok = true;
  mma_policy_small_bwd<MMAHostCtx, 8>(MMAHostCtx(), M, d, Z, beta, ls2, var, mu, Sigma, gf1, gSff, gcross, gmu, gSig, gpar, sm.data(), &ok);
  return ok ? 0 : 1;
}

// all P + L items of one batch element, summed and symmetrised (what k_gp_bwd_items + k_gp_bwd_sum do on the device).
// pagg != nullptr: the off-diagonal pairs as aggregates (col then holds the L diagonal pairs only).
// chunk > 0: the moment sums of every item first as partials over chunks of that many centres (k_gp_bwd_moments).
static int gp_bwd_all(int L, int M, int Mp, int d, int full_cov, int with_unc, const double* Z, const double* ls2, const double* mu,
                      const double* Sigma, const double* latmat, const double* w, const double* q, const double* col,
                      const double* row, const double* pagg, const double* f1raw, const double* g_f1, const double* g_Sff,
                      const double* g_cross, int chunk, double* gmu, double* gS) {
  const int P = full_cov ? L * (L + 1) / 2 : L, nc = mma_gp_ncol(d);
  std::vector<double> sm(mma_gp_item_scratch(d, 1) + 8), cbuf(M), gSi(d * d), gmi(d), accS(d * d, 0.0), accm(d, 0.0);
  bool ok = true;
  for (int item = 0; item < L + P; ++item) {
    std::vector<double> pre;
    int nchunk = 0;
    const bool needs = item < L || !(pagg && item - L >= L);
    if (chunk > 0 && needs) {
      nchunk = (M + chunk - 1) / chunk;
      pre.resize((size_t)nchunk * 3 * nc);
      std::vector<double> ms(mma_gp_moments_scratch(d, 1, chunk) + 8);
      for (int ch = 0; ch < nchunk; ++ch)
        mma_gp_item_moments(MMAHostCtx(), item, ch * chunk, std::min(M, (ch + 1) * chunk), L, M, Mp, d, P, with_unc != 0, Z, mu, latmat,
                            w, q, col, row, g_f1, g_Sff, full_cov, g_cross, pagg != nullptr, pre.data() + (size_t)ch * 3 * nc, ms.data());
    }
    mma_gp_item_bwd(MMAHostCtx(), item, L, M, Mp, d, P, with_unc != 0, Z, ls2, mu, Sigma, latmat, w, q, col, row, g_f1, g_Sff,
                    full_cov, g_cross, gSi.data(), gmi.data(), cbuf.data(), sm.data(), &ok, pagg, f1raw,
                    nchunk ? pre.data() : nullptr, nchunk);
    for (int i = 0; i < d * d; ++i) accS[i] += gSi[i];
    for (int i = 0; i < d; ++i) accm[i] += gmi[i];
  }
  for (int i = 0; i < d; ++i) { gmu[i] = accm[i]; for (int j = 0; j < d; ++j) gS[i * d + j] = 0.5 * (accS[i * d + j] + accS[j * d + i]); }
  return ok ? 0 : 1;
}

extern "C" int hc_gp_bwd(int L, int M, int Mp, int d, int full_cov, int with_unc, const double* Z, const double* ls2, const double* mu,
                         const double* Sigma, const double* latmat, const double* w, const double* q, const double* col,
                         const double* row, const double* g_f1, const double* g_Sff, const double* g_cross, double* gmu, double* gS) {
  return gp_bwd_all(L, M, Mp, d, full_cov, with_unc, Z, ls2, mu, Sigma, latmat, w, q, col, row, nullptr, nullptr, g_f1, g_Sff, g_cross,
                    0, gmu, gS);
}
extern "C" int hc_gp_bwd_chunked(int L, int M, int Mp, int d, int full_cov, int with_unc, const double* Z, const double* ls2,
                                 const double* mu, const double* Sigma, const double* latmat, const double* w, const double* q,
                                 const double* col, const double* row, const double* g_f1, const double* g_Sff,
                                 const double* g_cross, int chunk, double* gmu, double* gS) {
  return gp_bwd_all(L, M, Mp, d, full_cov, with_unc, Z, ls2, mu, Sigma, latmat, w, q, col, row, nullptr, nullptr, g_f1, g_Sff, g_cross,
                    chunk, gmu, gS);
}
extern "C" int hc_gp_bwd_agg(int L, int M, int Mp, int d, int full_cov, int with_unc, const double* Z, const double* ls2, const double* mu,
                             const double* Sigma, const double* latmat, const double* w, const double* q, const double* col,
                             const double* pagg, const double* f1raw, const double* g_f1, const double* g_Sff, const double* g_cross,
                             int chunk, double* gmu, double* gS) {
  return gp_bwd_all(L, M, Mp, d, full_cov, with_unc, Z, ls2, mu, Sigma, latmat, w, q, col, nullptr, pagg, f1raw, g_f1, g_Sff, g_cross,
                    chunk, gmu, gS);
}

// polynomial part of one off-diagonal pair's aggregates from the packed weight moments, then the re-centring
extern "C" void hc_pair_poly(int d, const double* G, const double* dmu, const double* mR, const double* mC, double* T) {
  std::vector<double> sm(mma_pair_poly_scratch(d));
  mma_pair_poly(MMAHostCtx(), d, G, dmu, mR, mC, T, sm.data());
}
extern "C" void hc_pair_convert(int d, const double* dmu2, double* T) { mma_pair_convert(MMAHostCtx(), d, dmu2, T); }
extern "C" void hc_pair_convert_rows(int d, const double* dmu, double* T) { mma_pair_convert_rows(MMAHostCtx(), d, dmu, T); }
extern "C" int hc_mono_off(int n, int d) { return mm_mono_off(n, d); }
extern "C" int hc_mono_rank(const int* k, int n) { return mm_mono_rank_unsorted(n > 0 ? k[0] : 0, n > 1 ? k[1] : 0, n > 2 ? k[2] : 0, n > 3 ? k[3] : 0, n); }
