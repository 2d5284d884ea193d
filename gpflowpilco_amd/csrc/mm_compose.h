// Shapes, index maps and workspace / tape layouts of the composed rollout (mm_compose.hip, mm_compose_bwd.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "mm_common.h"

#define MMC_NX 16          // largest state dimension
#define MMC_NA 8           // largest number of encoded (angle) dimensions
#define MMC_ND 32          // largest drift input dimension (= MM_DMAX)

struct MMComposeDims {
  int nx, na, nb, ne, nd;
  int active[MMC_NA];
  int inactive[MMC_NX];
  int slot[MMC_NX];        // state dim r -> ia (< na) if active, else na + ib
};

// ---- workspace of the composition (caller-owned; sizes from mm_compose_workspace_bytes) ------------------------
struct MMComposeLayout {
  size_t me, See;            // [B][ne], [B][ne][ne] T   policy GP input
  size_t pf1, pSff, pcross;  // [B][1], [B][1][1], [B][ne][1] T   policy GP output
  size_t md, Sdd;            // [B][nd], [B][nd][nd] T   drift GP input
  size_t df1, dSff, dcross;  // [B][nx], [B][nx][nx], [B][nd][nx] T   drift GP output
  size_t Sxe, cpol;          // [B][nx][ne], [B][ne] f64   Cov(x, e);  Cov(e,e)^-1 Cov(e, u)
  size_t total;
};

static inline MMComposeLayout mm_compose_layout(int B, int nx, int na, int dtype) {
  MMComposeLayout o;
  const size_t es = mm_elem_size(dtype), A = 256;
  const int nb = nx - na, ne = 2 * na + nb, nd = ne + 1;
  size_t off = 0;
  o.me = off;     off = mm_align_up(off + (size_t)B * ne * es, A);
  o.See = off;    off = mm_align_up(off + (size_t)B * ne * ne * es, A);
  o.pf1 = off;    off = mm_align_up(off + (size_t)B * es, A);
  o.pSff = off;   off = mm_align_up(off + (size_t)B * es, A);
  o.pcross = off; off = mm_align_up(off + (size_t)B * ne * es, A);
  o.md = off;     off = mm_align_up(off + (size_t)B * nd * es, A);
  o.Sdd = off;    off = mm_align_up(off + (size_t)B * nd * nd * es, A);
  o.df1 = off;    off = mm_align_up(off + (size_t)B * nx * es, A);
  o.dSff = off;   off = mm_align_up(off + (size_t)B * nx * nx * es, A);
  o.dcross = off; off = mm_align_up(off + (size_t)B * nd * nx * es, A);
  o.Sxe = off;    off = mm_align_up(off + (size_t)B * nx * ne * 8, A);
  o.cpol = off;   off = mm_align_up(off + (size_t)B * ne * 8, A);
  o.total = off;
  return o;
}

static inline int mm_compose_dims(int nx, int na, const int32_t* active_dims, MMComposeDims& D) {
  if (nx <= 0 || nx > MMC_NX || na <= 0 || na > MMC_NA || na > nx || !active_dims) return MM_E_DIM;
  D.nx = nx; D.na = na; D.nb = nx - na; D.ne = 2 * na + D.nb; D.nd = D.ne + 1;
  if (D.nd > MMC_ND) return MM_E_DIM;
  bool used[MMC_NX] = {false};
  for (int i = 0; i < na; ++i) {
    const int r = active_dims[i];
    if (r < 0 || r >= nx || used[r]) return MM_E_ARG;
    used[r] = true; D.active[i] = r; D.slot[r] = i;
  }
  int ib = 0;
  for (int r = 0; r < nx; ++r) if (!used[r]) { D.inactive[ib] = r; D.slot[r] = na + ib; ++ib; }   // sorted (components.py:66)
  return 0;
}


// ---- tape of a rollout that will be differentiated (mm_rollout_composed_taped -> mm_rollout_composed_backward) ----
// H + 1 slots with the compose-workspace layout: slot h holds everything step h produced from x_h (me, See, Sxe of
// the encoding of x_h; the policy match, cpol, md, Sdd; the drift match), slot H only the encoding of x_H; then the
// states x_0 .. x_H.  The forward kernels write straight into the slots (no copies).
// mm_compose_bwd.hip: mm_moment_match_with_sums with the choice of stamping the sums with their state (the public entry does)
int mm_moment_match_with_sums_impl(const void* packed, size_t packed_bytes, int L, int M, int d, int dtype, int B,
                                   const void* mu, const void* Sigma, int flags, double jitter,
                                   void* f1, void* Sff, void* cross_pre,
                                   void* workspace, size_t workspace_bytes, void* bwd_ws, size_t bwd_ws_bytes,
                                   int32_t* status, void* stream, bool stamp);

struct MMTapeLayout {
  size_t slot_bytes;         // = mm_compose_layout(...).total
  size_t xm, xS;             // [H+1][B][nx], [H+1][B][nx][nx] T
  size_t ws, ws_stride;      // [H][ws_stride]: the drift match's workspace of every step (ws_stride = 0: not kept)
  size_t gp, gp_stride;      // [H][gp_stride]: the drift match's backward buffer of every step with the SUMS of its sweeps on it
                             // (mm_moment_match_with_sums; gp_stride = 0: not kept, the reverse sweep runs the sweeps itself)
  size_t total;
};

// The reverse sweep needs the drift's q-stage workspace of every step (mm_backward_sums reads w, q and the streamed
// operands).  Where H copies fit in MM_TAPE_WS_LIMIT the taped forward keeps them (its drift matches run IN the tape) and the
// reverse sweep skips re-running the q stage: three launches less per reverse step at cartpole sizes (110 KB per step and
// element).  Larger models re-run the q stage from the taped (md, Sdd).
#define MM_TAPE_WS_LIMIT ((size_t)512 << 20)

static inline MMTapeLayout mm_tape_layout(int B, int H, int nx, int na, int drift_M, int dtype) {
  MMTapeLayout o;
  const size_t es = mm_elem_size(dtype), A = 256;
  o.slot_bytes = mm_compose_layout(B, nx, na, dtype).total;
  size_t off = (size_t)(H + 1) * o.slot_bytes;
  o.xm = off; off = mm_align_up(off + (size_t)(H + 1) * B * nx * es, A);
  o.xS = off; off = mm_align_up(off + (size_t)(H + 1) * B * nx * nx * es, A);
  const size_t wsb = mm_align_up(mm_workspace_layout(B, nx, drift_M, nx + na + 1, dtype, MM_FULL_OUTPUT_COV | MM_MODEL_UNCERTAINTY).total, A);
  o.ws = off; o.ws_stride = 0;
  if (wsb * (size_t)H <= MM_TAPE_WS_LIMIT) { o.ws_stride = wsb; off += wsb * (size_t)H; }
  // ... and, where they fit too, the sums of the drift match's backward sweeps: nothing in them depends on the incoming
  // gradient and they contain the forward's sums, so the taped forward runs THEM instead of the forward's two reduces
  // (mm_moment_match_with_sums) and the reverse step is the chain rule alone -- two launches less per step and direction
  o.gp = off; o.gp_stride = 0;
  if (o.ws_stride) {
    const size_t gpb = mm_align_up(mm_moment_match_backward_bytes_dtype(B, nx, drift_M, nx + na + 1, dtype,
                                                                        MM_FULL_OUTPUT_COV | MM_MODEL_UNCERTAINTY), A);
    if (gpb && (wsb + gpb) * (size_t)H <= MM_TAPE_WS_LIMIT) { o.gp_stride = gpb; off += gpb * (size_t)H; }
  }
  o.total = off;
  return o;
}

// gradient slab of the packed one-latent policy, per batch element: dZ [M][d], dbeta [M], dls2 [d], dvar, dmean_c
static inline size_t mm_policy_grad_len(int M, int d) { return (size_t)M * d + M + d + 2; }
