// Rollout COMPOSITION around the two GP moment matches of a PILCO step, as device code (gfx950).
//
// One step of MomentMatchingPILCO's policy-loss rollout (gpflow_pilco/loops/pilco.py:192-220) with encoder and
// policy present (gpflow_pilco/dynamics/forward_sde.py:95-137):
//     x (nx)  --TrigonometricEncoder on `active` dims-->  e = [sin a, cos a, x_inactive]            (ne = 2 na + nb)
//     e       --policy: SVGP mean (KernelRegressor) -> Chain[Scale, Shift, NormalCDF]-->  u         (nu = 1)
//     d = joint(e, u) (nd = ne + 1)  --drift SVGP-->  f (nx);  Cov(x, f) ~ Cov(x, d) Cov(d, d)^-1 Cov(d, f)
//     Euler moment update (dynamics/solvers.py:110-135), expected Gaussian cost of the encoded new state.
// The GP matches are the library's own kernels (mm_moment_match); everything between them is <= 25 x 25 algebra
// per batch element and runs here, one wave per element, f64 in LDS:
//   k_compose_encode : moment_matching/components.py:19-57 + maths.py:143-176 (sincos moments, Cov(x, e))
//   k_compose_policy : moment_matching/bijectors.py:39-69 (NormalCDF: Owen's T by 48-point Gauss-Legendre, the
//                      quadrature gpflowpilco_amd/special.py uses), Shift, Scale, the chain rule
//                      (gaussian.py:66-83) and GaussianMatch.joint (gaussian.py:53-63)
//   k_compose_tail   : forward_sde.py:105-131 bookkeeping + solvers.py:110-135, then the new state's encoding and the
//                      expected cost, one launch;  k_policy_match_small: the policy's whole moment match in one launch
// mm_rollout_composed enqueues the whole H-step rollout (per step: policy match, k_compose_policy, drift match, k_compose_tail).
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"
#include "mm_cost.h"
#include "mm_dev.h"
#include "mm_compose.h"

// Stage profile of block 0 (cycles between consecutive MM_STAMP calls, accumulated per id): only in a -DMM_STAGE_PROFILE
// build (tools/profile_c1_stages.py); expands to nothing in the shipped library.
#ifdef MM_STAGE_PROFILE
__device__ long long* mm_stage_prof = nullptr;
extern "C" void mm_stage_profile_set(void* device_buffer) {
  hipMemcpyToSymbol(HIP_SYMBOL(mm_stage_prof), &device_buffer, sizeof(void*));
}
#define MM_STAMP_INIT() long long mm_t0_ = clock64()
#define MM_STAMP(k_) do { if (mm_stage_prof && threadIdx.x == 0 && blockIdx.x == 0) { const long long t_ = clock64(); \
                          mm_stage_prof[k_] += t_ - mm_t0_; mm_t0_ = t_; } } while (0)
#else
#define MM_STAMP_INIT() do {} while (0)
#define MM_STAMP(k_) do {} while (0)
#endif

static __device__ const double MM_GL48_X[48] = {-9.98771007252426068e-01, -9.93530172266350764e-01, -9.84124583722826851e-01, -9.70591592546247273e-01, -9.52987703160430910e-01, -9.31386690706554332e-01, -9.05879136715569633e-01, -8.76572020274247854e-01, -8.43588261624393487e-01, -8.07066204029442624e-01, -7.67159032515740358e-01, -7.24034130923814634e-01, -6.77872379632663891e-01, -6.28867396776513599e-01, -5.77224726083972683e-01, -5.23160974722232996e-01, -4.66902904750958414e-01, -4.08686481990716721e-01, -3.48755886292160755e-01, -2.87362487355455554e-01, -2.24763790394689050e-01, -1.61222356068891709e-01, -9.70046992094626970e-02, -3.23801709628693674e-02, 3.23801709628693674e-02, 9.70046992094626970e-02, 1.61222356068891709e-01, 2.24763790394689050e-01, 2.87362487355455554e-01, 3.48755886292160755e-01, 4.08686481990716721e-01, 4.66902904750958414e-01, 5.23160974722232996e-01, 5.77224726083972683e-01, 6.28867396776513599e-01, 6.77872379632663891e-01, 7.24034130923814634e-01, 7.67159032515740358e-01, 8.07066204029442624e-01, 8.43588261624393487e-01, 8.76572020274247854e-01, 9.05879136715569633e-01, 9.31386690706554332e-01, 9.52987703160430910e-01, 9.70591592546247273e-01, 9.84124583722826851e-01, 9.93530172266350764e-01, 9.98771007252426068e-01};
static __device__ const double MM_GL48_W[48] = {3.15334605230917957e-03, 7.32755390127649234e-03, 1.14772345792349736e-02, 1.55793157229429276e-02, 1.96161604573552965e-02, 2.35707608393240925e-02, 2.74265097083568818e-02, 3.11672278327983394e-02, 3.47772225647706573e-02, 3.82413510658306741e-02, 4.15450829434645535e-02, 4.46745608566940997e-02, 4.76166584924902839e-02, 5.03590355538542783e-02, 5.28901894851934867e-02, 5.51995036999840538e-02, 5.72772921004029295e-02, 5.91148396983954827e-02, 6.07044391658935825e-02, 6.20394231598924636e-02, 6.31141922862537841e-02, 6.39242385846479494e-02, 6.44661644359498381e-02, 6.47376968126836816e-02, 6.47376968126836816e-02, 6.44661644359498381e-02, 6.39242385846479494e-02, 6.31141922862537841e-02, 6.20394231598924636e-02, 6.07044391658935825e-02, 5.91148396983954827e-02, 5.72772921004029295e-02, 5.51995036999840538e-02, 5.28901894851934867e-02, 5.03590355538542783e-02, 4.76166584924902839e-02, 4.46745608566940997e-02, 4.15450829434645535e-02, 3.82413510658306741e-02, 3.47772225647706573e-02, 3.11672278327983394e-02, 2.74265097083568818e-02, 2.35707608393240925e-02, 1.96161604573552965e-02, 1.55793157229429276e-02, 1.14772345792349736e-02, 7.32755390127649234e-03, 3.15334605230917957e-03};

// ---------------------------------------------------------------------------------------------
// k_compose_encode: (mx, Sxx) -> moments of e = [sin a, cos a, x_inactive] and Cov(x, e)
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void mmc_encode_body(const MMComposeDims& D, const T* mx, const T* Sxx, T* me, T* See, double* Sxe,
                                                int b, int lane) {
  const int nx = D.nx, na = D.na, nb = D.nb, ne = D.ne, n2 = 2 * na;
  __shared__ double m[MMC_NX], S[MMC_NX * MMC_NX];
  __shared__ double s1[MMC_NA], c1[MMC_NA];
  __shared__ double Syy[4 * MMC_NA * MMC_NA];      // centred covariance of [sin a, cos a]
  __shared__ double Sxy[MMC_NX * 2 * MMC_NA];      // Cov(x, [sin a, cos a])
  for (int i = lane; i < nx; i += 64) m[i] = (double)mx[(size_t)b * nx + i];
  for (int i = lane; i < nx * nx; i += 64) S[i] = (double)Sxx[(size_t)b * nx * nx + i];
  __syncthreads();
  __shared__ double sa[MMC_NA], ca[MMC_NA];        // sin a_i, cos a_i: ONE sincos per angle, the pair terms by angle addition
  if (lane < na) {                                 // maths.py:143-176: first moments
    const int r = D.active[lane];
    const double a = m[r], ev = exp(-0.5 * S[r * nx + r]);
    double sv, cv;
    sincos(a, &sv, &cv);
    sa[lane] = sv; ca[lane] = cv;
    s1[lane] = ev * sv; c1[lane] = ev * cv;
  }
  __syncthreads();
  for (int idx = lane; idx < na * na; idx += 64) { // second moments (uncentred), then centred
    const int i = idx / na, j = idx - i * na;
    const int ri = D.active[i], rj = D.active[j];
    const double vi = S[ri * nx + ri], vj = S[rj * nx + rj];
    const double sij = 0.5 * (S[ri * nx + rj] + S[rj * nx + ri]);       // (Sxx + Sxx^T) / 2
    const double A = exp(-0.5 * (vi + vj) - sij), Bm = exp(-0.5 * (vi + vj) + sij);
    const double cc = ca[i] * ca[j], ss = sa[i] * sa[j];
    const double Acos = A * (cc - ss), Bcos = Bm * (cc + ss);           // cos(a_i + a_j), cos(a_i - a_j)
    const double s2 = 0.5 * (Bcos - Acos), c2 = 0.5 * (Bcos + Acos);
    const double sc = 0.5 * (sa[i] * ca[j] * (Bm + A) - sa[j] * ca[i] * (Bm - A));   // E[sin a_i cos a_j]
    Syy[i * n2 + j] = s2 - s1[i] * s1[j];
    Syy[(na + i) * n2 + na + j] = c2 - c1[i] * c1[j];
    Syy[i * n2 + na + j] = sc - s1[i] * c1[j];
    Syy[(na + j) * n2 + i] = sc - s1[i] * c1[j];
  }
  // Cov(x, y) = Sxa [diag(c1), diag(-s1)]   (pre-inverted cross of sincos, components.py:35-37)
  for (int idx = lane; idx < nx * na; idx += 64) {
    const int r = idx / na, j = idx - r * na;
    const double sra = S[r * nx + D.active[j]];
    Sxy[r * n2 + j] = sra * c1[j];
    Sxy[r * n2 + na + j] = -sra * s1[j];
  }
  __syncthreads();
  T* meb = me + (size_t)b * ne;
  T* Seb = See + (size_t)b * ne * ne;
  double* Sxeb = Sxe + (size_t)b * nx * ne;
  for (int k = lane; k < ne; k += 64)
    meb[k] = (T)(k < na ? s1[k] : k < n2 ? c1[k - na] : m[D.inactive[k - n2]]);
  for (int idx = lane; idx < ne * ne; idx += 64) {           // components.py:41-53
    const int i = idx / ne, j = idx - i * ne;
    double v;
    if (i < n2 && j < n2) v = Syy[i * n2 + j];
    else if (i >= n2 && j >= n2) v = S[D.inactive[i - n2] * nx + D.inactive[j - n2]];
    else if (i >= n2) v = Sxy[D.inactive[i - n2] * n2 + j];  // Sby
    else v = Sxy[D.inactive[j - n2] * n2 + i];               // Sby^T
    Seb[idx] = (T)v;
  }
  for (int idx = lane; idx < nx * ne; idx += 64) {
    const int r = idx / ne, k = idx - r * ne;
    Sxeb[idx] = k < n2 ? Sxy[r * n2 + k] : S[r * nx + D.inactive[k - n2]];
  }
}

template <typename T>
__global__ __launch_bounds__(64) void k_compose_encode(MMComposeDims D, const T* __restrict__ mx, const T* __restrict__ Sxx,
                                                       T* __restrict__ me, T* __restrict__ See, double* __restrict__ Sxe) {
  mmc_encode_body<T>(D, mx, Sxx, me, See, Sxe, (int)blockIdx.x, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
// k_compose_policy: policy GP output (mean-only) -> u = scale (Phi(f) + shift); joint moments of d = (e, u)
// ---------------------------------------------------------------------------------------------
// Body of the head for one batch element, called by EVERY thread of the block (any block size that is a multiple of 64; barriers
// inside): mf, vx = the policy GP's output moments, pc [ne] = its pre-inverted cross term (any address space).
template <typename T, typename TC>
__device__ __forceinline__ void mmc_policy_head(const MMComposeDims& D, double scale, double shift, const T* __restrict__ me,
                                                const T* __restrict__ See, double mf, double vx, const TC* pc,
                                                T* __restrict__ md, T* __restrict__ Sdd, double* __restrict__ cpol, int b) {
  const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, ne = D.ne, nd = D.nd;
  __shared__ double cp[MMC_ND], Seu[MMC_ND];
  __shared__ double hv[4];                     // mu_u, Suu, head_pre
  vx = vx > 0.0 ? vx : 0.0;                    // variance of the regressor's mean under x ~ N: >= 0 up to rounding
  // bijectors.py:39-69, 1-D branch
  const double isq = rsqrt(vx + 1.0), z = isq * mf;
  const double y1 = 0.5 * erfc(-z * 0.70710678118654752440);
  // Owen's T(z, a), a = rsqrt(1 + 2 vx) in (0, 1]: 48-point Gauss-Legendre on [0, a] (gpflowpilco_amd/special.py)
  const double aa = rsqrt(1.0 + 2.0 * vx);
  double part = 0.0;
  if (lane < 48) {
    const double t = 0.5 * aa * (MM_GL48_X[lane] + 1.0);
    part = MM_GL48_W[lane] * exp(-0.5 * z * z * (1.0 + t * t)) / (1.0 + t * t);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  const double owen = 0.5 * aa * part * 0.15915494309189533577;        // / (2 pi)
  const double y2 = y1 - 2.0 * owen;                                   // E[Phi^2]
  const double head_pre = isq * 0.39894228040143267794 * exp(-0.5 * z * z) * scale;   // Var(f)^-1 Cov(f, u)
  if (tid == 0) { hv[0] = scale * (y1 + shift); hv[1] = scale * scale * (y2 - y1 * y1); }      // (every wave holds the same values)
  // chain rule (gaussian.py:66-83): Cov(e,e)^-1 Cov(e, u) = cross_pre(GP) * head_pre
  for (int k = tid; k < ne; k += nth) cp[k] = (double)pc[k] * head_pre;
  __syncthreads();
  for (int k = tid; k < ne; k += nth) {
    double s = 0.0;
    for (int l = 0; l < ne; ++l) s = fma((double)See[((size_t)b * ne + k) * ne + l], cp[l], s);
    Seu[k] = s;
    cpol[(size_t)b * ne + k] = cp[k];
  }
  __syncthreads();
  T* mdb = md + (size_t)b * nd;
  T* Sdb = Sdd + (size_t)b * nd * nd;
  for (int k = tid; k < nd; k += nth) mdb[k] = k < ne ? me[(size_t)b * ne + k] : (T)hv[0];
  for (int idx = tid; idx < nd * nd; idx += nth) {           // gaussian.py:53-63
    const int i = idx / nd, j = idx - i * nd;
    double v;
    if (i < ne && j < ne) v = (double)See[((size_t)b * ne + i) * ne + j];
    else if (i < ne) v = Seu[i];
    else if (j < ne) v = Seu[j];
    else v = hv[1];
    Sdb[idx] = (T)v;
  }
}

template <typename T>
__global__ __launch_bounds__(64) void k_compose_policy(MMComposeDims D, double scale, double shift,
                                                       const T* __restrict__ me, const T* __restrict__ See,
                                                       const T* __restrict__ pf1, const T* __restrict__ pSff,
                                                       const T* __restrict__ pcross,
                                                       T* __restrict__ md, T* __restrict__ Sdd, double* __restrict__ cpol) {
  const int b = blockIdx.x;
  mmc_policy_head<T, T>(D, scale, shift, me, See, (double)pf1[b], (double)pSff[b], pcross + (size_t)b * D.ne, md, Sdd, cpol, b);
}

// ---------------------------------------------------------------------------------------------
// mmc_step_body: Cov(x, f) bookkeeping of forward_sde.py:105-131 and the Euler moment update (first stage of k_compose_tail)
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void mmc_step_body(const MMComposeDims& D, double dt, const double* Sxe, const double* cpol, const T* Sdd,
                                              const T* df1, const T* dSff, const T* dcross, T* mx, T* Sxx,
                                              T* traj_mu, T* traj_S, int b, int lane) {
  const int nx = D.nx, na = D.na, ne = D.ne, nd = D.nd, n2 = 2 * na;
  __shared__ double Sxd[MMC_NX * MMC_ND], Sxf[MMC_NX * MMC_NX];
  const double* Sxeb = Sxe + (size_t)b * nx * ne;
  const double* cp = cpol + (size_t)b * ne;
  const T* Sdb = Sdd + (size_t)b * nd * nd;
  // Cov(x, d): rows of the encoded dims = [Sae, Sae cpol] (Cov(a, e) and its image under the policy), rows of the
  // other dims = the corresponding rows of Cov(d, d)
  for (int idx = lane; idx < nx * nd; idx += 64) {
    const int r = idx / nd, k = idx - r * nd;
    const int sl = D.slot[r];
    double v;
    if (sl < na) {
      if (k < ne) v = Sxeb[r * ne + k];
      else {
        double s = 0.0;
        for (int l = 0; l < ne; ++l) s = fma(Sxeb[r * ne + l], cp[l], s);
        v = s;
      }
    } else {
      v = (double)Sdb[(n2 + (sl - na)) * nd + k];
    }
    Sxd[idx] = v;
  }
  __syncthreads();
  const T* dc = dcross + (size_t)b * nd * nx;
  for (int idx = lane; idx < nx * nx; idx += 64) {           // Cov(x, f) = Cov(x, d) Cov(d,d)^-1 Cov(d, f)
    const int r = idx / nx, c = idx - r * nx;
    double s = 0.0;
    for (int k = 0; k < nd; ++k) s = fma(Sxd[r * nd + k], (double)dc[k * nx + c], s);
    Sxf[idx] = s;
  }
  __syncthreads();
  for (int idx = lane; idx < nx * nx; idx += 64) {           // solvers.py:110-135
    const int r = idx / nx, c = idx - r * nx;
    const double v = (double)Sxx[(size_t)b * nx * nx + idx] + dt * (Sxf[r * nx + c] + Sxf[c * nx + r])
                   + dt * dt * (double)dSff[(size_t)b * nx * nx + idx];
    Sxx[(size_t)b * nx * nx + idx] = (T)v;
    if (traj_S) traj_S[(size_t)b * nx * nx + idx] = (T)v;
  }
  if (lane < nx) {
    const double v = (double)mx[(size_t)b * nx + lane] + dt * (double)df1[(size_t)b * nx + lane];
    mx[(size_t)b * nx + lane] = (T)v;
    if (traj_mu) traj_mu[(size_t)b * nx + lane] = (T)v;
  }
}

// ---------------------------------------------------------------------------------------------
// k_compose_tail: the end of a step in ONE launch -- Euler update (mmc_step_body), the encoding of the new
// state (the next step's policy input and this step's cost argument) and the expected cost of the encoded state
// (mm_expected_cost's arithmetic).  At cartpole sizes every kernel of the chain runs for ~5 us, most of it launch and
// dependency latency: two launches fewer per step.  One wave per batch element; the stages communicate through the
// wave's own global writes (visible after the workgroup barrier).
// ---------------------------------------------------------------------------------------------
// What the tail needs to form the drift match's Sff itself from the reduce kernels' partial slabs (k_finalize's sum, mm_kernels.hip,
// for an f64 pack with full output covariance and model uncertainty: the composed rollout's drift) -- one launch less per step.
struct MMTailFinalize {
  const double *partB, *partC, *var, *f1raw;
  int P, NS, ns_diag, ns_off, nsC, enabled;
};

template <typename T>
__global__ __launch_bounds__(64) void k_compose_tail(MMComposeDims D, double dt, const double* cpol, const T* Sdd,
                                                     const T* df1, T* dSff, const T* dcross, const double* Sxe_in, T* mx, T* Sxx,
                                                     T* traj_mu, T* traj_S, T* me, T* See, double* Sxe,
                                                     const T* target, const T* precis, T* cost, MMTailFinalize fin) {
  // (me, See, Sxe: the NEW state's encoding; Sxe_in: Cov(x, e) of the state the step started from -- the same buffer
  // in an ordinary rollout, consecutive tape slots in a taped one)
  extern __shared__ double csm[];
  const int b = blockIdx.x, lane = threadIdx.x;
  MM_STAMP_INIT();
  if (fin.enabled) {
    // Sff_aa' = sum of the pair's slab (fixed order) [- (sum w)^2 + var_a on the diagonal: factored f64 reduce], symmetric fill
    const int L = D.nx;
    for (int p = lane; p < fin.P; p += 64) {
      int a = p, a2 = p;
      if (p >= L) { int r = p - L, i = 0; while (r >= L - 1 - i) { r -= L - 1 - i; ++i; } a = i; a2 = i + 1 + r; }
      const double* pb = fin.partB + ((size_t)b * fin.P + p) * fin.NS;
      double s = 0.0;
      const int ns = a == a2 ? fin.ns_diag : fin.ns_off;
      for (int k = 0; k < ns; ++k) s += pb[k];
      if (a == a2) {
        const double* pc = fin.partC + ((size_t)b * L + a) * fin.NS;
        for (int k = 0; k < fin.nsC; ++k) s += pc[k];
        const double f = fin.f1raw[(size_t)b * L + a];
        s = s - f * f + fin.var[a];
        dSff[((size_t)b * L + a) * L + a] = (T)s;
      } else {
        dSff[((size_t)b * L + a) * L + a2] = (T)s;
        dSff[((size_t)b * L + a2) * L + a] = (T)s;
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  mmc_step_body<T>(D, dt, Sxe_in, cpol, Sdd, df1, dSff, dcross, mx, Sxx, traj_mu, traj_S, b, lane);
  __syncthreads();
  MM_STAMP(0);
  mmc_encode_body<T>(D, mx, Sxx, me, See, Sxe, b, lane);
  MM_STAMP(1);
  if (cost) {
    __syncthreads();
    mm_expected_cost_body<T>(D.ne, me, See, target, precis, cost, b, lane, csm);
  }
  MM_STAMP(2);
}

// ---------------------------------------------------------------------------------------------
// k_policy_match_small: the policy's moment match in ONE launch.  The policy of the composed rollout is a one-latent,
// mean-only regressor (KernelRegressor: model_uncertainty = False, moment_matching/models.py:34-41) with a few tens of
// kernel centres; through the general path its match is five dependent launches of a few microseconds each.  Here one
// 256-thread workgroup per batch element does all of it in f64 (the general path reduces diagonal pairs in f64 too):
//   (Sigma + Lambda)^-1 and (Sigma + Lambda / 2)^-1 by two waves at once (mm_spd_inverse), E = sym(Lambda^-1 Sigma P),
//   T = V S^-1 Sigma, G = Lambda^-1 T Lambda^-1, const;  per centre m (thread m): zeta, q, w = beta q, r = -(zeta^T E zeta
//   - zeta^T G zeta) / 2, g = G zeta;  f1 = sum w + c,  cross = P sum w zeta;
//   Sff = sum_ij w_i expm1(r_i + r_j + const + zeta_i . g_j) w_j   (256 / M threads per column j).
// Requires L == 1, M <= 128, d <= 8 (else the general mm_moment_match is used).  A non-PD Sigma + V sets the status
// word exactly as k_prep does.
// ---------------------------------------------------------------------------------------------
// HEAD: the workgroup goes on with the policy head (mmc_policy_head: NormalCDF / Scale / Shift, joint moments of d = (e, u)) on the
// moments it has just formed -- the rollout's policy step is ONE launch (it was this kernel + k_compose_policy: a launch is
// ~ 5 us of the 51 us a cartpole-sized step takes).
int mm_f64_num_slots(int Mp, int diag);      // mm_f64.hip
#define MMS_MMAX 128
template <typename T, bool HEAD>
__global__ __launch_bounds__(256) void k_policy_match_small(const double* __restrict__ Z64, const double* __restrict__ beta64,
                                                            const double* __restrict__ ls2, const double* __restrict__ var,
                                                            const double* __restrict__ meanc, int M, int d,
                                                            const T* __restrict__ mu, const T* __restrict__ Sigma, double jitter,
                                                            T* __restrict__ f1, T* __restrict__ Sff, T* __restrict__ cross,
                                                            int32_t* status, MMComposeDims D, double scale, double shift,
                                                            T* __restrict__ md, T* __restrict__ Sdd, double* __restrict__ cpol) {
  constexpr int DK = 8, DP = DK + 1;
  const int b = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  __shared__ double Sg[DK * DP], Am[4][DK * DP], Ym[4][DK * DP], Em[DK * DK], Gm[DK * DK];
  __shared__ double zs[MMS_MMAX][DK], gs[MMS_MMAX][DK];
  __shared__ double ws[MMS_MMAX], rs[MMS_MMAX];
  __shared__ double red[4][DK + 2], sv[DK + 2], lds4[4];
  __shared__ int okw[4];
  const T* Sb = Sigma + (size_t)b * d * d;
  for (int idx = tid; idx < d * d; idx += 256) {
    const int i = idx / d, j = idx - i * d;
    Sg[i * DP + j] = (double)(i >= j ? Sb[i * d + j] : Sb[j * d + i]);     // symmetrised from the lower triangle (k_prep)
  }
  if (tid < 4) okw[tid] = 1;
  MM_STAMP_INIT();
  __syncthreads();
  MM_STAMP(4);
  // wave 0: Sigma + Lambda; wave 1: Sigma + Lambda / 2 (V of the pair (a, a), kernel_expectation.py:119); waves 2, 3: identity
  for (int idx = lane; idx < d * d; idx += 64) {
    const int i = idx / d, j = idx - i * d;
    const double add = wv == 0 ? ls2[i] : wv == 1 ? 0.5 * ls2[i] : 1.0;
    Am[wv][i * DP + j] = (wv < 2 ? Sg[i * DP + j] : 0.0) + (i == j ? add : 0.0);
  }
  bool ok = true;
  const double ldw = mm_spd_inverse(Am[wv], Ym[wv], d, DP, &ok);
  if (!ok && wv < 2) okw[wv] = 0;
  if (lane == 0) lds4[wv] = ldw;
  __syncthreads();
  MM_STAMP(5);
  ok = okw[0] && okw[1];
  const double ldA = lds4[0], ldS = lds4[1];
  const double* Pm = Am[0];            // (Sigma + Lambda)^-1
  const double* S0 = Am[1];            // (Sigma + V)^-1
  // E = sym(Lambda^-1 Sigma P)  (k_prep);  T = V S0 Sigma (into Ym[0]), symmetrised;  G = Lambda^-1 T Lambda^-1
  for (int idx = tid; idx < d * d; idx += 256) {
    const int i = idx / d, j = idx - i * d;
    double s1 = 0.0, t1 = 0.0, tt = 0.0;
    for (int k = 0; k < d; ++k) {
      s1 += Sg[i * DP + k] * Pm[k * DP + j];
      t1 += Sg[j * DP + k] * Pm[k * DP + i];
      tt += S0[i * DP + k] * Sg[k * DP + j];
    }
    Em[i * DK + j] = 0.5 * (s1 / ls2[i] + t1 / ls2[j]);
    Ym[0][i * DP + j] = 0.5 * ls2[i] * tt;
  }
  __syncthreads();
  for (int idx = tid; idx < d * d; idx += 256) {
    const int i = idx / d, j = idx - i * d;
    Gm[i * DK + j] = 0.5 * (Ym[0][i * DP + j] + Ym[0][j * DP + i]) / (ls2[i] * ls2[j]);
  }
  // log-normaliser of q and the pair constant (k_prep: -0.5 ldS - 0.5 sum log(2 Lambda_k) + ldA)
  double sl = (tid < d) ? log(ls2[tid]) : 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sl += __shfl_xor(sl, off, 64);
  sl = __shfl(sl, 0, 64);
  __syncthreads();
  // (every wave needs the wave-0 sum: through LDS)
  if (tid == 0) lds4[2] = sl;
  __syncthreads();
  sl = lds4[2];
  const double lognorm = log(var[0]) + 0.5 * sl - 0.5 * ldA;
  const double cst = -0.5 * ldS - 0.5 * (sl + d * 0.6931471805599453) + ldA;
  MM_STAMP(6);
  // ---- per centre -------------------------------------------------------------------------------------------------
  double wv_m = 0.0, acc[DK];
#pragma unroll
  for (int k = 0; k < DK; ++k) acc[k] = 0.0;
  if (tid < M) {
    double z[DK];
#pragma unroll
    for (int k = 0; k < DK; ++k) z[k] = k < d ? Z64[(size_t)tid * d + k] - (double)mu[(size_t)b * d + k] : 0.0;
    double maha = 0.0, r1 = 0.0, tq = 0.0;
#pragma unroll
    for (int i = 0; i < DK; ++i) {
      double tp = 0.0, te = 0.0, tg = 0.0;
#pragma unroll
      for (int k = 0; k < DK; ++k) {
        const bool in = i < d && k < d;
        tp = fma(in ? Pm[i * DP + k] : 0.0, z[k], tp);
        te = fma(in ? Em[i * DK + k] : 0.0, z[k], te);
        tg = fma(in ? Gm[i * DK + k] : 0.0, z[k], tg);
      }
      maha = fma(z[i], tp, maha);
      r1 = fma(z[i], te, r1);
      tq = fma(z[i], tg, tq);
      gs[tid][i] = tg;
      zs[tid][i] = z[i];
    }
    const double qv = exp(lognorm - 0.5 * maha);
    wv_m = beta64[tid] * qv;
    ws[tid] = wv_m;
    rs[tid] = -0.5 * (r1 - tq);
#pragma unroll
    for (int k = 0; k < DK; ++k) acc[k] = wv_m * z[k];
  }
  __syncthreads();
  MM_STAMP(7);
  // ---- the M x M sum: 256 / M threads share a column (rows interleaved); every partial goes into the workgroup sum ----
  double colsum = 0.0;
  {
    const int nsub = 256 / M;                               // >= 2 (M <= 128)
    const int j = tid % M, i0 = tid / M;
    if (i0 < nsub) {
      double g[DK];
#pragma unroll
      for (int k = 0; k < DK; ++k) g[k] = gs[j][k];
      const double base = rs[j] + cst;
      for (int i = i0; i < M; i += nsub) {
        double delta = rs[i] + base;
#pragma unroll
        for (int k = 0; k < DK; ++k) delta = fma(zs[i][k], g[k], delta);
        colsum = fma(ws[i], expm1(fmin(delta, MM_EXP_CAP_F64)), colsum);      // (mm_common.h: exponent caps)
      }
      colsum *= ws[j];
    }
  }
  MM_STAMP(8);
  // ---- the d + 2 workgroup sums (f1, sum w zeta, Sff) together -----------------------------------------------------
  {
    double v[DK + 2];
#pragma unroll
    for (int k = 0; k < DK; ++k) v[k] = acc[k];
    v[DK] = wv_m; v[DK + 1] = colsum;
#pragma unroll
    for (int k = 0; k < DK + 2; ++k) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < DK + 2; ++k) red[wv][k] = v[k];
    }
  }
  __syncthreads();
  if (tid < DK + 2) sv[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  __syncthreads();
  if (tid == 0) {
    f1[b] = (T)(sv[DK] + meanc[0]);
    Sff[b] = (T)(sv[DK + 1] + jitter);
  }
  __shared__ double hcross[DK];
  if (tid < d) {
    double s = 0.0;
    for (int k = 0; k < d; ++k) s += Pm[tid * DP + k] * sv[k];     // Sigma^-1 Cov(x, f) = P sum_i w_i zeta_i  (models.py:263-277)
    cross[(size_t)b * d + tid] = (T)s;
    if (HEAD) hcross[tid] = (double)(T)s;                           // (what the two-launch form reads back)
  }
  if (!ok && tid == 0 && status) {
    atomicMax(status, (int)gridDim.x - b);                          // B - b: the host decodes the smallest failing b
    status[1] = 0;
  }
  MM_STAMP(9);
  if constexpr (HEAD) {
    __syncthreads();
    mmc_policy_head<T, double>(D, scale, shift, mu, Sigma, (double)(T)(sv[DK] + meanc[0]), (double)(T)(sv[DK + 1] + jitter), hcross,
                               md, Sdd, cpol, b);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
extern "C" size_t mm_compose_workspace_bytes(int B, int nx, int na, int dtype) {
  if (B <= 0 || nx <= 0 || nx > MMC_NX || na <= 0 || na > MMC_NA || na > nx) return 0;
  if (2 * na + (nx - na) + 1 > MMC_ND) return 0;
  return mm_compose_layout(B, nx, na, dtype).total;
}

// one compose-workspace slot as typed pointers
template <typename T>
struct MMCSlot {
  T *me, *See, *pf1, *pSff, *pcross, *md, *Sdd, *df1, *dSff, *dcross;
  double *Sxe, *cpol;
  MMCSlot(char* w, const MMComposeLayout& cl)
      : me((T*)(w + cl.me)), See((T*)(w + cl.See)), pf1((T*)(w + cl.pf1)), pSff((T*)(w + cl.pSff)), pcross((T*)(w + cl.pcross)),
        md((T*)(w + cl.md)), Sdd((T*)(w + cl.Sdd)), df1((T*)(w + cl.df1)), dSff((T*)(w + cl.dSff)), dcross((T*)(w + cl.dcross)),
        Sxe((double*)(w + cl.Sxe)), cpol((double*)(w + cl.cpol)) {}
};

template <typename T>
static int mm_rollout_composed_t(const void* drift, size_t drift_bytes, int Md, const void* policy, size_t policy_bytes, int Mpol,
                                 int dtype, int B, int H, double dt, const MMComposeDims& D, double scale, double shift,
                                 const T* target, const T* precis, T* mx, T* Sxx, T* cost, T* traj_mu, T* traj_S,
                                 void* ws_drift, size_t ws_drift_bytes, void* ws_policy, size_t ws_policy_bytes,
                                 char* wsc, const MMComposeLayout& cl, char* tape, int32_t* status, hipStream_t s) {
  const int nx = D.nx, ne = D.ne, nd = D.nd;
  // taped: step h works in tape slot h and the tail writes the next encoding into slot h + 1; the states x_0 .. x_H
  // go to the tape's state block.  Untaped: every step reuses the one workspace.
  const MMTapeLayout tl = mm_tape_layout(B, H, nx, D.na, Md, dtype);
  auto slot = [&](int h) { return MMCSlot<T>(tape ? tape + (size_t)h * tl.slot_bytes : wsc, cl); };
  T* xm = tape ? (T*)(tape + tl.xm) : nullptr;
  T* xS = tape ? (T*)(tape + tl.xS) : nullptr;
#define MMC_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
  if (tape) {
    hipError_t e1 = hipMemcpyAsync(xm, mx, (size_t)B * nx * sizeof(T), hipMemcpyDeviceToDevice, s);
    hipError_t e2 = hipMemcpyAsync(xS, Sxx, (size_t)B * nx * nx * sizeof(T), hipMemcpyDeviceToDevice, s);
    if (e1 != hipSuccess) return (int)e1;
    if (e2 != hipSuccess) return (int)e2;
  }
  {
    MMCSlot<T> c0 = slot(0);
    hipLaunchKernelGGL((k_compose_encode<T>), dim3(B), dim3(64), 0, s, D, (const T*)mx, (const T*)Sxx, c0.me, c0.See, c0.Sxe);
    MMC_CHECK();
  }
  // the drift's layouts (the tail may finalize its Sff itself): f64 packs whose two reduces fit the one-launch form (mm_f64.hip)
  const MMModelLayout dml = mm_model_layout(nx, Md, nd, dtype, 1);
  const MMWorkspaceLayout dwl = mm_workspace_layout(B, nx, Md, nd, dtype, MM_FULL_OUTPUT_COV | MM_MODEL_UNCERTAINTY);
  const bool tail_finalizes = sizeof(T) == 8 && dwl.Po > 0 && nd <= 16 && drift_bytes >= dml.total &&
                              (long long)B * (dwl.P + nx) * mm_f64_num_slots(dwl.Mp, 0) <= 4096;
  for (int h = 0; h < H; ++h) {
    MMCSlot<T> c = slot(h), n = slot(h + 1);
    // policy: mean-only regressor (models.py:34-41: model_uncertainty = False), one latent
    int rc = 0;
    if (Mpol <= MMS_MMAX && ne <= 8) {
      const MMModelLayout pl = mm_model_layout(1, Mpol, ne, dtype, 1);
      if (policy_bytes < pl.Cm) return MM_E_WORKSPACE;         // the packed buffer must hold everything the kernel reads
      const char* pp = (const char*)policy;
      // (policy match AND head in one launch: the head's inputs (me, See) are this kernel's (mu, Sigma))
      hipLaunchKernelGGL((k_policy_match_small<T, true>), dim3(B), dim3(256), 0, s, (const double*)(pp + pl.Z64),
                         (const double*)(pp + pl.beta64), (const double*)(pp + pl.ls2), (const double*)(pp + pl.var),
                         (const double*)(pp + pl.meanc), Mpol, ne, (const T*)c.me, (const T*)c.See, 0.0, c.pf1, c.pSff, c.pcross, status,
                         D, scale, shift, c.md, c.Sdd, c.cpol);
      MMC_CHECK();
    } else {
      rc = mm_moment_match(policy, policy_bytes, 1, Mpol, ne, dtype, B, c.me, c.See, MM_FULL_OUTPUT_COV, 0.0,
                           c.pf1, c.pSff, c.pcross, ws_policy, ws_policy_bytes, status, (void*)s);
      if (rc) return rc;
      hipLaunchKernelGGL((k_compose_policy<T>), dim3(B), dim3(64), 0, s, D, scale, shift, (const T*)c.me, (const T*)c.See,
                         (const T*)c.pf1, (const T*)c.pSff, (const T*)c.pcross, c.md, c.Sdd, c.cpol);
      MMC_CHECK();
    }
    // (taped, small enough: the match runs in the tape's own workspace slot of this step, which the reverse sweep reads)
    void* wsd = (tape && tl.ws_stride) ? (void*)(tape + tl.ws + (size_t)h * tl.ws_stride) : ws_drift;
    const size_t wsd_bytes = (tape && tl.ws_stride) ? tl.ws_stride : ws_drift_bytes;
    if (tape && tl.gp_stride)     // the sums of the backward's sweeps stay on the tape and give this step's value too (mm_compose.h)
      rc = mm_moment_match_with_sums_impl(drift, drift_bytes, nx, Md, nd, dtype, B, c.md, c.Sdd,
                                          MM_FULL_OUTPUT_COV | MM_MODEL_UNCERTAINTY, 0.0, c.df1, c.dSff, c.dcross, wsd, wsd_bytes,
                                          tape + tl.gp + (size_t)h * tl.gp_stride, tl.gp_stride, status, (void*)s, false);
    else
      // (small f64 drift: the tail kernel sums the reduce kernels' slabs itself -- no k_finalize launch)
      rc = mm_moment_match(drift, drift_bytes, nx, Md, nd, dtype, B, c.md, c.Sdd,
                           MM_FULL_OUTPUT_COV | MM_MODEL_UNCERTAINTY | (tail_finalizes ? (MM_STAGE_DIAG | MM_STAGE_OFFDIAG) : 0), 0.0,
                           c.df1, c.dSff, c.dcross, wsd, wsd_bytes, status, (void*)s);
    if (rc) return rc;
    MMTailFinalize fin = {};
    if (tail_finalizes && !(tape && tl.gp_stride)) {
      const char* wq = (const char*)wsd;
      fin.partB = (const double*)(wq + dwl.partB); fin.partC = (const double*)(wq + dwl.partC);
      fin.var = (const double*)((const char*)drift + dml.var); fin.f1raw = (const double*)(wq + dwl.f1raw);
      fin.P = dwl.P; fin.NS = dwl.NS; fin.ns_diag = mm_f64_num_slots(dwl.Mp, 1); fin.ns_off = mm_f64_num_slots(dwl.Mp, 0);
      fin.nsC = fin.ns_diag; fin.enabled = 1;
    }
    // Euler update, the new state's encoding (the cost statistic of this step, pilco.py:199-205, and the next step's
    // policy input) and the expected cost: one launch
    T* tm = tape ? xm + (size_t)(h + 1) * B * nx : (traj_mu ? traj_mu + (size_t)h * B * nx : (T*)nullptr);
    T* tS = tape ? xS + (size_t)(h + 1) * B * nx * nx : (traj_S ? traj_S + (size_t)h * B * nx * nx : (T*)nullptr);
    hipLaunchKernelGGL((k_compose_tail<T>), dim3(B), dim3(64), mm_cost_lds_bytes(ne), s, D, dt, (const double*)c.cpol,
                       (const T*)c.Sdd, (const T*)c.df1, c.dSff, (const T*)c.dcross, (const double*)c.Sxe, mx, Sxx,
                       tm, tS, n.me, n.See, n.Sxe, target, precis,
                       cost ? cost + (size_t)h * B : (T*)nullptr, fin);
    MMC_CHECK();
    if (tape && traj_mu) {
      hipError_t e1 = hipMemcpyAsync(traj_mu + (size_t)h * B * nx, tm, (size_t)B * nx * sizeof(T), hipMemcpyDeviceToDevice, s);
      if (e1 != hipSuccess) return (int)e1;
    }
    if (tape && traj_S) {
      hipError_t e2 = hipMemcpyAsync(traj_S + (size_t)h * B * nx * nx, tS, (size_t)B * nx * nx * sizeof(T), hipMemcpyDeviceToDevice, s);
      if (e2 != hipSuccess) return (int)e2;
    }
  }
#undef MMC_CHECK
  return 0;
}

// (A ONE-launch kernel for small models -- the whole H-step rollout in one 512-thread workgroup per batch element, everything in
// LDS -- was built in round 3, parity-green, and measured 271 us per step against 64 for this multi-launch path at cartpole sizes
// (one compute unit does what 14-36 workgroups do here): removed from the library in round 4; DESIGN.md section 8 keeps its
// stage profile.)
static int mm_rollout_composed_impl(const void* drift_packed, size_t drift_bytes, int drift_L, int drift_M, int drift_d,
                                    const void* policy_packed, size_t policy_bytes, int policy_M, int policy_d,
                                    int dtype, int B, int H, double dt, int nx, int na, const int32_t* active_dims,
                                    double head_scale, double head_shift, const void* target, const void* precis,
                                    void* mx, void* Sxx, void* cost, void* traj_mu, void* traj_Sigma,
                                    void* ws_drift, size_t ws_drift_bytes, void* ws_policy, size_t ws_policy_bytes,
                                    void* ws_compose, size_t ws_compose_bytes, void* tape, size_t tape_bytes,
                                    int32_t* status, void* stream) {
  if (!drift_packed || !policy_packed || !mx || !Sxx) return MM_E_ARG;
  if (!ws_drift || !ws_policy || (!ws_compose && !tape)) return MM_E_ARG;
  if (B <= 0 || H <= 0 || drift_M <= 0 || policy_M <= 0) return MM_E_ARG;
  if (dtype != MM_F32 && dtype != MM_F64) return MM_E_DTYPE;
  if (cost && (!target || !precis)) return MM_E_ARG;
  MMComposeDims D;
  int rc = mm_compose_dims(nx, na, active_dims, D);
  if (rc) return rc;
  if (drift_L != nx || drift_d != D.nd || policy_d != D.ne) return MM_E_STATE;
  const MMComposeLayout cl = mm_compose_layout(B, nx, na, dtype);
  if (!tape && ws_compose_bytes < cl.total) return MM_E_WORKSPACE;
  if (tape && tape_bytes < mm_tape_layout(B, H, nx, na, drift_M, dtype).total) return MM_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MM_F64)
    return mm_rollout_composed_t<double>(drift_packed, drift_bytes, drift_M, policy_packed, policy_bytes, policy_M, dtype, B, H, dt,
                                         D, head_scale, head_shift, (const double*)target, (const double*)precis, (double*)mx,
                                         (double*)Sxx, (double*)cost, (double*)traj_mu, (double*)traj_Sigma, ws_drift, ws_drift_bytes,
                                         ws_policy, ws_policy_bytes, (char*)ws_compose, cl, (char*)tape, status, s);
  return mm_rollout_composed_t<float>(drift_packed, drift_bytes, drift_M, policy_packed, policy_bytes, policy_M, dtype, B, H, dt,
                                      D, head_scale, head_shift, (const float*)target, (const float*)precis, (float*)mx,
                                      (float*)Sxx, (float*)cost, (float*)traj_mu, (float*)traj_Sigma, ws_drift, ws_drift_bytes,
                                      ws_policy, ws_policy_bytes, (char*)ws_compose, cl, (char*)tape, status, s);
}

extern "C" int mm_rollout_composed(const void* drift_packed, size_t drift_bytes, int drift_L, int drift_M, int drift_d,
                                   const void* policy_packed, size_t policy_bytes, int policy_M, int policy_d,
                                   int dtype, int B, int H, double dt, int nx, int na, const int32_t* active_dims,
                                   double head_scale, double head_shift, const void* target, const void* precis,
                                   void* mx, void* Sxx, void* cost, void* traj_mu, void* traj_Sigma,
                                   void* ws_drift, size_t ws_drift_bytes, void* ws_policy, size_t ws_policy_bytes,
                                   void* ws_compose, size_t ws_compose_bytes, int32_t* status, void* stream) {
  return mm_rollout_composed_impl(drift_packed, drift_bytes, drift_L, drift_M, drift_d, policy_packed, policy_bytes, policy_M, policy_d,
                                  dtype, B, H, dt, nx, na, active_dims, head_scale, head_shift, target, precis, mx, Sxx, cost,
                                  traj_mu, traj_Sigma, ws_drift, ws_drift_bytes, ws_policy, ws_policy_bytes, ws_compose,
                                  ws_compose_bytes, nullptr, 0, status, stream);
}

extern "C" size_t mm_compose_tape_bytes(int B, int H, int nx, int na, int drift_M, int dtype) {
  if (B <= 0 || H <= 0 || nx <= 0 || nx > MMC_NX || na <= 0 || na > MMC_NA || na > nx || drift_M <= 0) return 0;
  if (2 * na + (nx - na) + 1 > MMC_ND) return 0;
  return mm_tape_layout(B, H, nx, na, drift_M, dtype).total;
}

// The same rollout, recorded: every per-step intermediate and the states x_0 .. x_H go to `tape`
// (mm_compose_tape_bytes), which mm_rollout_composed_backward reads (csrc/mm_compose_bwd.hip).
extern "C" int mm_rollout_composed_taped(const void* drift_packed, size_t drift_bytes, int drift_L, int drift_M, int drift_d,
                                         const void* policy_packed, size_t policy_bytes, int policy_M, int policy_d,
                                         int dtype, int B, int H, double dt, int nx, int na, const int32_t* active_dims,
                                         double head_scale, double head_shift, const void* target, const void* precis,
                                         void* mx, void* Sxx, void* cost, void* ws_drift, size_t ws_drift_bytes,
                                         void* ws_policy, size_t ws_policy_bytes, void* tape, size_t tape_bytes,
                                         int32_t* status, void* stream) {
  if (!tape) return MM_E_ARG;
  return mm_rollout_composed_impl(drift_packed, drift_bytes, drift_L, drift_M, drift_d, policy_packed, policy_bytes, policy_M, policy_d,
                                  dtype, B, H, dt, nx, na, active_dims, head_scale, head_shift, target, precis, mx, Sxx, cost,
                                  nullptr, nullptr, ws_drift, ws_drift_bytes, ws_policy, ws_policy_bytes, nullptr, 0, tape, tape_bytes,
                                  status, stream);
}
