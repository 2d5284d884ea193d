/*
 * gpflowpilco_mm.h -- C ABI of the MI355X (gfx950) moment-matched GP propagation.
 *
 * Drop-in boundary for the per-step hot path of j-wilson/GPflowPILCO:
 *   moment_matching(GaussianMoments, SVGP|GPR)            gpflow_pilco/moment_matching/models.py:44-299
 *   kernel_expectation  <K_xZ>, <K_Zx K_xZ'>              gpflow_pilco/utils/kernel_expectation.py:72-288
 *   MomentMatchingEuler.step                              gpflow_pilco/dynamics/solvers.py:108-135
 *
 * The reference has no FFI of its own (pure Python/TF); these entry points are what a
 * maintainer would bind from the handlers above.  Conventions:
 *   - every pointer is a DEVICE pointer unless it says "host"; row-major contiguous; the batch
 *     axis B is the leading axis, so a B-shard is a pointer offset;
 *   - the caller owns every buffer including the workspace (size from *_bytes queries);
 *   - calls only enqueue work (no allocation, no synchronisation; graph-capturable) in the order of `stream` (a hipStream_t
 *     passed as void*): when `stream` reaches the end of a call's work, everything the call enqueued is complete.  Some calls
 *     (mm_moment_match, the rollouts, mm_moment_match_with_sums / _backward) run independent latency-bound kernel chains on a
 *     library-owned side stream beside a long sweep and join it to `stream` before they return (fork / join by events; under
 *     stream capture the side stream joins the capture).  That side stream and its two events belong to the CALLER'S stream: one
 *     triple per (device, caller stream), created on first use and kept for the process -- calls on different streams share
 *     nothing and may run concurrently, one of them under HIP-graph capture (ABI users: one enqueuing thread per stream at a time,
 *     as for any stream-ordered API).  The stage API (mm_q_forward, mm_Q_reduce_forward) never leaves work on a side stream;
 *   - return value: 0 ok, <0 bad argument (MM_E_*), >0 a hipError_t;
 *   - `status` (device int32[4], zeroed by the caller, may be NULL): [0] = B - b for the smallest
 *     batch index b whose (Sigma + V) Cholesky was not positive definite (0 = all fine; outputs
 *     of that b are NaN), [1] = an item code.  The reference raises InvalidArgumentError there
 *     (kernel_expectation.py:125-126, models.py:271).  [2] / [3] (MM_F32 packs; ABI version 2): running counts of the
 *     (batch element, off-diagonal pair) items the forward / the backward re-reduced in f64 because the f32 sweep's own
 *     rounding-error estimate exceeded MM_ROUTE_TOL (3e-4) of the batch element's off-diagonal covariance scale
 *     (csrc/mm_route.hip): the f32 pack's accuracy contract -- what stays in f32 is within ~4e-5 of
 *     that scale (rounding; plus <= ~1e-5 systematic), the rest has f64 accuracy.  The reference computes these terms in float64 throughout
 *     (kernel_expectation.py:158-165).
 *   - `packed` / `packed_bytes`: the buffer filled by mm_pack_model and its size (whether C is
 *     present is inferred from the size).
 *
 * dtype selects the element type T of the state (mu, Sigma), outputs and the streaming
 * workspace.  All d x d algebra, log-normalisers, first moments and cross terms are done in
 * f64 regardless; T only governs the M x M inner reduction and storage.
 */
#ifndef GPFLOWPILCO_MM_H
#define GPFLOWPILCO_MM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_F32 0
#define MM_F64 1

#define MM_DMAX 32          /* largest supported GP input dimension d            */
#define MM_M_ALIGN 128      /* inducing-point axis is zero-padded to this multiple */

/* flags */
#define MM_FULL_OUTPUT_COV   1   /* models.py full_output_cov=True: Sff is [B,L,L]; else [B,L]     */
#define MM_MODEL_UNCERTAINTY 2   /* models.py model_uncertainty=True: adds E[Var f] (needs C)      */
#define MM_FORCE_GENERIC     4   /* use the portable VALU reduce kernel even where an MFMA one exists */
/* mm_Q_reduce_forward only: run a subset of its launches (none of the three bits = all three);
 * lets a caller bracket each kernel with its own events (bench.py does). */
#define MM_STAGE_DIAG        8   /* f64 reduce of the diagonal pairs a == a' (incl. the C-weighted term) */
#define MM_STAGE_OFFDIAG    16   /* T reduce of the off-diagonal pairs a < a'                          */
#define MM_STAGE_FINALIZE   32   /* partial slabs -> Sff                                               */
/* measurement aid: every tile of the reduce kernels takes its most expensive range tier (f32 reduce: all three
 * split-product stages + the exp2 branch; f64 reduce: the k ln2 + r form) whatever the data.  The tiers are all
 * valid on the whole range, so results are unchanged up to rounding; bench.py --recipe worst times it. */
#define MM_FORCE_WORST_TIER 64
/* mm_moment_match_backward only: `workspace` still holds the q stage of exactly this (mu, Sigma, flags) -- the forward of the
 * same match was the last call on it, on the same stream -- so the q stage is not run again.  The caller's promise; the
 * device checks the part it can see (the q stage stamps the mean it read into the workspace) and reports a stale workspace
 * as status = {B - b, -1}. */
#define MM_WORKSPACE_CURRENT 128
/* MM_F32 packs, test / measurement aids of the accuracy contract (csrc/mm_route.hip): MM_FORCE_ROUTE sends EVERY off-diagonal
 * (b, pair) item through the f64 re-reduce whatever its error estimate says (forward and backward); MM_NO_ROUTE none (the f32
 * sweep's result as it is: what rounds 1-3 returned). */
#define MM_FORCE_ROUTE 256
#define MM_NO_ROUTE 512
/* mm_moment_match_backward only: `bwd_ws` holds the sums mm_moment_match_with_sums left for exactly this (mu, Sigma, flags):
 * the M x M sweeps are not run again, the call is the chain rule alone.  Combine with MM_WORKSPACE_CURRENT when `workspace`
 * is untouched as well; without it the q stage (only) is re-run.  The sums are stamped with the mean they were swept for and
 * the device verifies it: sums of another state are reported as status = {B - b, -2}. */
#define MM_SUMS_CURRENT 1024

/* error codes */
#define MM_E_ARG      (-1)  /* NULL pointer / non-positive size                  */
#define MM_E_DIM      (-2)  /* d > MM_DMAX or unsupported shape                  */
#define MM_E_DTYPE    (-3)
#define MM_E_WORKSPACE (-4) /* workspace too small                               */
#define MM_E_NO_C     (-5)  /* MM_MODEL_UNCERTAINTY asked for but model packed without C */
#define MM_E_STATE    (-6)  /* euler/rollout needs d == L                         */

int mm_abi_version(void);

/* ---- model packing (once per model; replaces the per-call Kuu Cholesky + triangular
 *      solves of models.py:216-235 by the precomputed beta = Kuu^-1 u and
 *      C = Kuu^-1 S Kuu^-1 - Kuu^-1) ------------------------------------ */
size_t mm_packed_model_bytes(int L, int M, int d, int dtype, int with_C);

int mm_pack_model(void* packed, size_t packed_bytes,
                  int L, int M, int d, int dtype,
                  const double* Z,            /* [L,M,d] inducing inputs (kernel-sliced)   */
                  const double* lengthscales, /* [L,d]                                     */
                  const double* variance,     /* [L]                                       */
                  const double* beta,         /* [L,M]   Kuu^-1 u                          */
                  const double* C,            /* [L,M,M] or NULL                           */
                  const double* mean_c,       /* [L] Constant mean, or NULL (Zero)         */
                  void* stream);

/* Order of the inducing points inside the pack.  The outputs of the path are sums over a latent's inducing points, so the pack
 * is free to store them in an order of its own: packs of M > 256 points are sorted per latent by |(z - mean z) / lengthscale|
 * (tiles of the M x M reduces then hold points of similar norm: lower per-tile range tiers); smaller packs -- in particular
 * every policy pack of the composed rollout, whose g_policy is per packed centre -- keep the caller's order.
 * perm [L][M] (device, int32): the caller's index of the point at packed position m.  q_out (mm_q_forward) is written in the
 * CALLER's order; the per-point sums of mm_backward_sums are in PACKED order. */
int mm_pack_perm(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int32_t* perm, void* stream);

/* ---- one moment match: (mu, Sigma) -> (f1, Sff, Sigma^-1 Cov(x,f)) --------------------
 * Replaces _mm_gauss_svgp_mo / _so / _mm_gauss_gpr up to (not including) the
 * LinearCoregionalization mixing (models.py:279-286), which stays on the host. */
size_t mm_workspace_bytes(int B, int L, int M, int d, int dtype, int flags);

int mm_moment_match(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype,
                    int B,
                    const void* mu,        /* [B,d]   T */
                    const void* Sigma,     /* [B,d,d] T (symmetric; lower triangle is read) */
                    int flags, double jitter,
                    void* f1,              /* [B,L]   T  (includes the Constant mean)       */
                    void* Sff,             /* [B,L,L] T, or [B,L] without MM_FULL_OUTPUT_COV */
                    void* cross_pre,       /* [B,d,L] T  Sigma^-1 Cov(x,f) (preinv=True)    */
                    void* workspace, size_t workspace_bytes,
                    int32_t* status, void* stream);

/* ---- stage-level entry points; mm_moment_match = q then Q ------- */
/* <K_xZ> terms (kernel_expectation.py:200-214 + gpflow <k(x,Z)>): fills the workspace and
 * writes f1, cross_pre; q_out (optional, [B,L,M] T) receives eKfu[b,m,a] transposed. */
int mm_q_forward(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                 const void* mu, const void* Sigma, int flags,
                 void* f1, void* cross_pre, void* q_out,
                 void* workspace, size_t workspace_bytes, int32_t* status, void* stream);

/* fused <K_Zx K_xZ'> reduce (kernel_expectation.py:72-247 + models.py:219-261) -> Sff.
 * Must follow mm_q_forward on the same workspace/stream. */
int mm_Q_reduce_forward(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                        int flags, double jitter, void* Sff,
                        void* workspace, size_t workspace_bytes, void* stream);

/* MomentMatchingEuler.step (solvers.py:110-135), requires d == L:
 *   mu' = mu + dt f1;  Sigma' = Sigma + dt (Sxf + Sxf^T) + dt^2 Sff,  Sxf = Sigma cross_pre */
int mm_euler_update(int B, int d, int dtype, double dt,
                    const void* mu, const void* Sigma,
                    const void* f1, const void* Sff, const void* cross_pre,
                    void* mu_out, void* Sigma_out, void* stream);

/* Drift-only closed rollout (Euler.__call__ fold, solvers.py:67-105, with
 * forward_sde(x, drift, None, None, None), forward_sde.py:34-46): H steps enqueued
 * back-to-back.  mu/Sigma are updated in place; traj_mu [H,B,d] / traj_Sigma [H,B,d,d]
 * (optional) receive the state after every step. */
int mm_rollout_closed(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B, int H,
                      double dt, int flags, double jitter,
                      void* mu, void* Sigma, void* traj_mu, void* traj_Sigma,
                      void* workspace, size_t workspace_bytes,
                      int32_t* status, void* stream);

/* Closed-form expected saturating cost E[-exp(-0.5 (x - x*)^T W (x - x*))], x ~ N(mean, cov)
 * (GaussianObjective.__call__ on GaussianMoments, gpflow_pilco/components.py:26-37): the
 * per-step statistic the rollout accumulates (loops/pilco.py:199-205).
 * mean [N,d], cov [N,d,d], target [d], precis [d,d] -> cost [N]  (all T). */
int mm_expected_cost(int N, int d, int dtype, const void* mean, const void* cov,
                     const void* target, const void* precis, void* cost, void* stream);

/* Diagnostic: after mm_q_forward (f32 model, d <= 8), how many (batch element, off-diagonal pair) items take the
 * moment collapse of csrc/mm_moments.hip / mm_moments6.hip (the degree-3..6 polynomial p6 of the remainder from weight
 * moments, wave tiles with max |b| <= 1/4 skipped: Cauchy-Schwarz bound <= 1/2), and for how many of those the bound alone puts
 * every |b| <= 1/4 (no tile work at all).  The collapse is decided per GROUP OF 64 ROWS of an item (the pack's norm order puts the
 * rows of large |A_i| last): out: device int32[6] = {collapsed in every row group, total, wholly inside, routed, partly collapsed
 * (some row groups), collapsed row groups over all items (of total * ceil(M / 128) * 2)}; zeros where the collapse does not
 * apply; routed = the items the last mm_moment_match / mm_Q_reduce_forward on this workspace re-reduced in f64
 * (csrc/mm_route.hip; meaningful only after such a call on an MM_F32 pack). */
int mm_offdiag_stats(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B, int flags,
                     const void* workspace, size_t workspace_bytes, int32_t* out, void* stream);

/* Diagnostic of the f32 pack's accuracy contract (csrc/mm_route.hip): after mm_moment_match / mm_Q_reduce_forward (or the
 * backward) on an MM_F32 pack, out [B][P-L][2] f64 = per (batch element, off-diagonal pair) {the sweep's estimate of its own
 * rounding error, the scale it is compared with}: an item is re-reduced in f64 when est > MM_ROUTE_TOL (3e-4) x scale. */
int mm_route_estimates(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B, int flags,
                       const void* workspace, size_t workspace_bytes, double* out, void* stream);

/* ---- composed policy rollout: SURVEY.md row f-2 -------------------------------------------------------
 * The rollout harness of MomentMatchingPILCO (gpflow_pilco/loops/pilco.py:192-220) for the cartpole-shaped system
 *   x (nx) -> TrigonometricEncoder on `active_dims` (components.py:73-75, moment_matching/components.py:19-57,
 *             maths.py:143-176) -> e = [sin a, cos a, x_inactive]  (ne = nx + na)
 *          -> policy = InverseLinkWrapper(KernelRegressor(SVGP, one latent), Chain[Scale, Shift, NormalCDF])
 *             (models/core.py:60-71, moment_matching/models.py:27-41, bijectors.py:21-69): u = scale (Phi(f(e)) + shift)
 *          -> drift SVGP on d = joint(e, u) (nd = ne + 1) with the cross-covariance bookkeeping of
 *             dynamics/forward_sde.py:95-137 -> MomentMatchingEuler.step (solvers.py:110-135)
 *          -> expected Gaussian cost of the encoded new state (components.py:26-37), per step.
 * H steps are enqueued back to back: per step two mm_moment_match calls and four small kernels
 * (csrc/mm_compose.hip).  drift: packed with C (model uncertainty on), drift_L == nx, drift_d == nd;
 * policy: one latent, policy_d == ne, evaluated without model uncertainty (a KernelRegressor has none).
 * mx [B,nx] / Sxx [B,nx,nx] are updated in place; cost [H,B] (optional, needs target [ne], precis [ne,ne]) receives
 * the per-step statistic (the loss of pilco.py:199-205 is its sum over H); traj_* [H,B,..] optional.
 * active_dims: HOST array of na distinct state indices.  Only the 1-D action (Owen's T branch, bijectors.py:57-58)
 * is supported; the n-D branch needs the Genz BVN (utils/bvn.py), out of scope. */
size_t mm_compose_workspace_bytes(int B, int nx, int na, int dtype);
int mm_rollout_composed(const void* drift_packed, size_t drift_bytes, int drift_L, int drift_M, int drift_d,
                        const void* policy_packed, size_t policy_bytes, int policy_M, int policy_d,
                        int dtype, int B, int H, double dt, int nx, int na, const int32_t* active_dims,
                        double head_scale, double head_shift, const void* target, const void* precis,
                        void* mx, void* Sxx, void* cost, void* traj_mu, void* traj_Sigma,
                        void* ws_drift, size_t ws_drift_bytes, void* ws_policy, size_t ws_policy_bytes,
                        void* ws_compose, size_t ws_compose_bytes, int32_t* status, void* stream);

/* The same rollout, RECORDED for differentiation: every per-step intermediate and the states x_0 .. x_H are written into
 * `tape` (mm_compose_tape_bytes) instead of a reused workspace; mx / Sxx / cost as above.  Where H copies of the drift's
 * workspace fit in 512 MB the tape also keeps the drift match's q stage of every step (the reverse sweep then does not
 * re-run it). */
size_t mm_compose_tape_bytes(int B, int H, int nx, int na, int drift_M, int dtype);
int mm_rollout_composed_taped(const void* drift_packed, size_t drift_bytes, int drift_L, int drift_M, int drift_d,
                              const void* policy_packed, size_t policy_bytes, int policy_M, int policy_d,
                              int dtype, int B, int H, double dt, int nx, int na, const int32_t* active_dims,
                              double head_scale, double head_shift, const void* target, const void* precis,
                              void* mx, void* Sxx, void* cost, void* ws_drift, size_t ws_drift_bytes,
                              void* ws_policy, size_t ws_policy_bytes, void* tape, size_t tape_bytes,
                              int32_t* status, void* stream);

/* ---- gradient of the composed rollout: SURVEY.md rows f-1 x f-2 -----------------------------------------------
 * What the only real caller needs: update_policy differentiates the whole closure with tf.GradientTape
 * (gpflow_pilco/utils/optimizers.py:51-56, examples/cartpole_swingup/train_utils.py:91-105, loops/pilco.py:192-220).
 * mm_rollout_composed_backward is the reverse sweep over the tape (f64 only; policy M <= 256 centres on ne <= 8 encoded dims,
 * else MM_E_DIM; B >= 1):
 *   g_cost   [H][B]  (in)  d loss / d cost[h][b]  (all ones for the loss of pilco.py:199-205)
 *   g_policy [B][M d + M + d + 2] (out, overwritten): per batch element the gradient w.r.t. the PACKED policy --
 *            Z [M][d], beta = Kuu^-1 u [M], ls2 = lengthscales^2 [d], variance, mean_c; sum over B and chain through
 *            beta(q_mu, Z, lengthscales, variance) on the host (a 30 x 30 precompute)
 *   g_mx0 [B][nx], g_Sxx0 [B][nx][nx] (out, both or neither): gradient w.r.t. the initial state (symmetric)
 * ws_bwd: mm_compose_backward_workspace_bytes; ws_drift: the drift's mm_workspace_bytes (full covariance + uncertainty). */
size_t mm_compose_backward_workspace_bytes(int B, int nx, int na, int drift_M);
size_t mm_policy_grad_bytes(int B, int policy_M, int policy_d);
int mm_rollout_composed_backward(const void* drift_packed, size_t drift_bytes, int drift_L, int drift_M, int drift_d,
                                 const void* policy_packed, size_t policy_bytes, int policy_M, int policy_d,
                                 int dtype, int B, int H, double dt, int nx, int na, const int32_t* active_dims,
                                 double head_scale, double head_shift, const void* target, const void* precis,
                                 const void* tape, size_t tape_bytes, const void* g_cost,
                                 void* g_policy, void* g_mx0, void* g_Sxx0,
                                 void* ws_drift, size_t ws_drift_bytes, void* ws_bwd, size_t ws_bwd_bytes,
                                 int32_t* status, void* stream);

/* ---- backward w.r.t. the input moments, stage A (f64 mode) ---------------------
 * The M x M part of d(f1, Sff, cross)/d(mu, Sigma) reduced to M-sized sums (see csrc/mm_backward.hip);
 * gpflowpilco_amd/autodiff.py finishes the chain rule.  Must follow mm_moment_match / mm_q_forward +
 * mm_Q_reduce_forward with the same (mu, Sigma, flags) on the same workspace.
 * out: [B][P][3+d][Mp] column sums (Ksum, csum, cC, Usum[d]) then [B][P-L][2][Mp] row sums (Rsum, rsum), the inducing points
 * in PACKED order (mm_pack_perm). */
size_t mm_backward_bytes(int B, int L, int M, int d, int flags);
int mm_backward_sums(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                     const void* mu, int flags, const void* workspace, size_t workspace_bytes,
                     void* out, size_t out_bytes, void* stream);

/* ---- backward w.r.t. the input moments, complete: the vector-Jacobian product of one moment match
 *   (g_f1 [B,L], g_Sff [B,L,L] | [B,L], g_cross [B,d,L]) -> g_mu [B,d], g_Sigma [B,d,d] (symmetric; += if accumulate_Sigma)
 * for a frozen model (the reference: tf.GradientTape through moment_matching/models.py:200-299).  Gradients are f64 in and
 * out; (mu, Sigma) have the pack's type.  Re-runs the q stage for (mu, Sigma) on `workspace`, then
 *   MM_F64 packs: the M x M sweeps of mm_backward_sums for every pair,
 *   MM_F32 packs with d <= 8 (mm_bwd_f32_supported): the f64 sweep for the L diagonal pairs only; the off-diagonal pairs
 *     are reduced to 1 + 2d + 3d^2 aggregates each -- polynomial part from the degree-4 weight moments in f64, remainder
 *     by a bf16-MFMA tile sweep (csrc/mm_bwd_f32.hip); other f32 packs return MM_E_DTYPE (use an f64 pack of the model),
 * then the M-sized moments and the d x d chain rule per (latent | pair) item and their sum (csrc/mm_compose_bwd.hip,
 * mm_adjoint.h).  bwd_ws: mm_moment_match_backward_bytes (enough for either pack type).
 * flags: the forward's, optionally | MM_WORKSPACE_CURRENT, optionally | MM_STAGE_* to run only part of the backward on what
 * earlier calls left in `workspace` / `bwd_ws` (measurement): MM_STAGE_DIAG = the f64 sweep (f64 packs: of every pair),
 * MM_STAGE_OFFDIAG = the f32 remainder sweep, MM_STAGE_FINALIZE = moment GEMM, aggregates, item moments, items and their sum. */
int mm_bwd_f32_supported(int d);
size_t mm_moment_match_backward_bytes(int B, int L, int M, int d, int flags);
size_t mm_moment_match_backward_bytes_dtype(int B, int L, int M, int d, int dtype, int flags);   /* exactly what that pack type needs */
/* the off-diagonal aggregates of an MM_F32 pack alone (tests, diagnostics): runs the q stage of (mu, Sigma) on `workspace`;
 * pagg [B][P-L][1 + 2d + 3d^2] f64 = sum_ij Omega_ij (1 | zeta_i | zeta_i zeta_i^T | zeta'_j | zeta'_j zeta'_j^T | zeta_i zeta'_j^T),
 * zeta = z - mu, Omega_ij = w_i w'_j e^{delta_ij} (pairs a < a' row by row) */
size_t mm_backward_pair_aggregates_bytes(int B, int L, int M, int d, int flags);
int mm_backward_pair_aggregates(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                                const void* mu, const void* Sigma, int flags, void* workspace, size_t workspace_bytes,
                                void* scratch, size_t scratch_bytes, void* pagg, size_t pagg_bytes,
                                int32_t* status, void* stream);
int mm_moment_match_backward(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                             const void* mu, const void* Sigma, int flags,
                             const void* g_f1, const void* g_Sff, const void* g_cross,
                             void* g_mu, void* g_Sigma, int accumulate_Sigma,
                             void* workspace, size_t workspace_bytes, void* bwd_ws, size_t bwd_ws_bytes,
                             int32_t* status, void* stream);
/* Value AND gradient without sweeping twice (what tf.GradientTape + tape.gradient cost the reference together,
 * utils/optimizers.py:51-56).  Nothing the backward's M x M sweeps compute depends on the incoming gradient, and their sums
 * contain the forward's: Sff_aa = sum_j (w_j c_j + q_j cC_j) + var + jitter, Sff_aa' = pagg[0] - (sum w)(sum w').
 * mm_moment_match_with_sums = mm_moment_match (same outputs, same types; Sff to the backward sweeps' accuracy, which is the
 * forward's or better) computed from the q stage + the BACKWARD's sweeps, which stay on `bwd_ws`
 * (mm_moment_match_backward_bytes_dtype); the matching mm_moment_match_backward(flags | MM_SUMS_CURRENT [| MM_WORKSPACE_CURRENT])
 * on the same bwd_ws is then the chain rule alone.  MM_F32 packs need d <= 8 (MM_E_DTYPE otherwise). */
int mm_moment_match_with_sums(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                              const void* mu, const void* Sigma, int flags, double jitter,
                              void* f1, void* Sff, void* cross_pre,
                              void* workspace, size_t workspace_bytes, void* bwd_ws, size_t bwd_ws_bytes,
                              int32_t* status, void* stream);

/* ---- pathwise (decoupled-sampling) rollout: SURVEY.md row f-3, BASELINE.json configs[4] ---------
 * f[s,a] = scale_a sum_k w[s,a,k] cos(2 pi (omega_t[a,:,k].x_s + phase[a,k]))
 *        + var_a   sum_m v[s,a,m] 2^(zs_t[a,:,m].(x_s * x_scale[a]) - hz[a,m] - hx) + mean_a
 * i.e. PathwiseSVGP.predict_f_samples with one input per sample path (gpflow_pilco/models/svgp.py:124-130,
 * loops/pilco.py:281-291; arithmetic in the un-vendored gpflow-sampling package -- parity unpinned).
 * Shared operands are pre-scaled and k-major: omega_t [L,d,K] = omega^T / 2pi, phase [L,K] = b / 2pi,
 * x_scale [L,d] = sqrt(log2 e) / lengthscales (f64), zs_t [L,d,M] = (Z * x_scale)^T, hz [L,M] = |zs|^2 / 2,
 * hx = |x * x_scale|^2 / 2 (computed in the kernel).
 * The per-sample weights arrive as ONE blocked stream wb[G][L][NB][4][BT]: G = ceil(S/4) sample groups,
 * BT = 256 (f32) / 128 (f64) terms per block, NB = K/BT prior blocks followed by M/BT update blocks;
 * element [g][a][tb][sl][t] is w[4g+sl][a][tb*BT+t] (or v[..][(tb-K/BT)*BT+t]).  K and M must be multiples
 * of BT and S is padded to a multiple of 4 (zero weights).  Each wave then streams one contiguous region.
 * x [S,d], f_out [S,L]: T. */
int mm_pathwise_eval(int S, int L, int M, int K, int d, int dtype,
                     const void* x, const void* omega_t, const void* phase, const void* zs_t, const void* hz,
                     const double* x_scale, const double* prior_scale, const double* variance,
                     const double* mean_c, const void* wb, void* f_out, void* stream);

/* The same evaluation that also emits, in the same pass, abs_out [S,L] T = scale_a sum_k |w cos(.)| + var_a sum_m |v 2^(.)|: the sum
 * of the ABSOLUTE terms of f[s,a].  A T-typed weight stream and T-typed basis values leave an error of at most ~2 eps_T abs_out in
 * f[s,a] (eps_f32 = 6e-8: v = Kuu^-1 (u - Phi w) cancels 1e5 .. 1e7-fold in sum_m v_m k(x, z_m) at M = 2000, so an f32 sample's
 * value can have lost its digits -- the caller sees it here instead of not at all; the f64 mode is the accurate path). */
int mm_pathwise_eval_bound(int S, int L, int M, int K, int d, int dtype,
                           const void* x, const void* omega_t, const void* phase, const void* zs_t, const void* hz,
                           const double* x_scale, const double* prior_scale, const double* variance,
                           const double* mean_c, const void* wb, void* f_out, void* abs_out, void* stream);

/* Euler.step folded H times on the sample paths (dynamics/solvers.py:50-65, no diffusion; d == L):
 * x <- x + dt f(x).  x is updated in place (x_tmp: scratch of the same size); traj [H,S,d] optional. */
int mm_pathwise_rollout(int S, int L, int M, int K, int d, int dtype, int H, double dt,
                        void* x, void* x_tmp, const void* omega_t, const void* phase, const void* zs_t,
                        const void* hz, const double* x_scale, const double* prior_scale,
                        const double* variance, const double* mean_c, const void* wb,
                        void* traj, void* stream);

/* The same evaluation that also emits the per-sample Jacobian d f[s,a] / d x[s,:]  (jac_out [S,L,d] T; d <= 8): in the SAME pass
 * over the weight stream (d more FMAs and, in the prior blocks, one more transcendental per term and sample). */
int mm_pathwise_eval_jac(int S, int L, int M, int K, int d, int dtype,
                         const void* x, const void* omega_t, const void* phase, const void* zs_t, const void* hz,
                         const double* x_scale, const double* prior_scale, const double* variance,
                         const double* mean_c, const void* wb, void* f_out, void* jac_out, void* stream);

/* ---- pathwise POLICY rollout and its gradient: row f-3 completed ---------------------------------------------------
 * PathwisePILCO._policy_loss_closure (gpflow_pilco/loops/pilco.py:263-298) for the cartpole-shaped system, per sample path s:
 *   e = TrigonometricEncoder(x) (components.py:44-75) -> u = scale (Phi(f_pol(e)) + shift), f_pol = the policy SVGP's predictive
 *   mean (models/core.py:60-71) -> x' = x + dt f_s([e, u]) (tensor branch of forward_sde, dynamics/forward_sde.py:23-31; Euler.step,
 *   solvers.py:50-65) -> cost[h][s] = -exp(-(enc(x') - target)^T precis (enc(x') - target) / 2) (components.py:39-41).
 * The drift paths are the operands of mm_pathwise_eval with L = nx latents on nd = nx + na + 1 inputs (nd <= 8); the policy is a
 * one-latent pack (mm_pack_model, M <= 256 centres on ne = nx + na inputs; only its f64 blocks are read).  x0 [S,nx], target [ne],
 * precis [ne,ne], cost [H,S]: T.  tape (mm_pathwise_tape_bytes): states x_0..x_H, the drift inputs of every step and, with
 * with_jacobians != 0, d f_s / d (e, u) of every step -- what mm_pathwise_policy_rollout_backward reads (the reference
 * differentiates the mean sample loss with a gradient tape: examples/cartpole_swingup/train_utils.py:108-135):
 *   g_cost [H][S] f64 (in; 1/S everywhere for the mean loss) -> g_policy [M ne + M + ne + 2] f64 (out: dZ, dbeta, d ls^2, dvar,
 *   dmean of the PACKED policy, summed over the samples; chain through beta(q_mu, Z, ls, var) on the host), g_x0 [S][nx] f64
 *   (optional).  The reverse sweep is ONE kernel (samples are independent) and does not touch the weight stream again. */
size_t mm_pathwise_tape_bytes(int S, int H, int nx, int na, int dtype, int with_jacobians);
int mm_pathwise_policy_rollout(int S, int M, int K, int dtype, int H, double dt, int nx, int na, const int32_t* active_dims,
                               const void* omega_t, const void* phase, const void* zs_t, const void* hz,
                               const double* x_scale, const double* prior_scale, const double* variance,
                               const double* mean_c, const void* wb,
                               const void* policy_packed, size_t policy_bytes, int policy_M, double head_scale, double head_shift,
                               const void* target, const void* precis, const void* x0, void* cost,
                               void* tape, size_t tape_bytes, int with_jacobians, void* stream);
size_t mm_pathwise_backward_scratch_bytes(int S, int policy_M, int ne);
int mm_pathwise_policy_rollout_backward(int S, int dtype, int H, double dt, int nx, int na, const int32_t* active_dims,
                                        const void* policy_packed, size_t policy_bytes, int policy_M,
                                        double head_scale, double head_shift, const void* target, const void* precis,
                                        const void* tape, size_t tape_bytes, const void* g_cost, void* g_policy, void* g_x0,
                                        void* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPFLOWPILCO_MM_H */
