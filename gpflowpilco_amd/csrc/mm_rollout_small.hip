// One-launch composed rollout for SMALL models (gfx950): the whole H-step policy rollout of mm_rollout_composed in ONE
// kernel, one 512-thread workgroup per batch element, state and every intermediate in LDS (csrc/mm_small.h).
//
// At cartpole sizes (BASELINE configs[0]: drift M = 100, policy M = 30, B = 1) the multi-launch rollout is nine dependent
// kernels per step, each a few microseconds of serial d x d algebra and global round trips on one or two compute units
// (profiles/r02_bench_c1_kernel_stats.csv: 13.8 + 13.5 + 9.9 + 5.5 + 5.3 + 5.2 + 4.9 + 4.7 + 4.7 us = the 63 us step).  Here
// the stages are separated by workgroup barriers, the (latent | pair) factorisations run one per wave, and the only global
// traffic of a step is the model itself (Z, beta, C: L2-resident).  The M x M sweeps run on ONE compute unit -- which is what
// bounds this kernel.
#include <hip/hip_runtime.h>
#include <math.h>
#include "mm_common.h"
#include "mm_compose.h"
#include "mm_small.h"

#define MMS_THREADS 512
#define MMS_DK 8

template <typename T>
__global__ __launch_bounds__(MMS_THREADS) void k_rollout_small(MMComposeDims D, int H, double dt, double scale, double shift,
                                                               MMSmallModel drift, MMSmallModel pol, int B,
                                                               const T* __restrict__ target, const T* __restrict__ precis,
                                                               T* __restrict__ mx, T* __restrict__ Sxx, T* __restrict__ cost,
                                                               T* __restrict__ traj_mu, T* __restrict__ traj_S, int32_t* status,
                                                               long long* prof) {
  extern __shared__ double sm[];
  const int b = blockIdx.x, nx = D.nx;
  bool ok = true;
  MMADevCtx ctx;
  long long last = clock64();
  ctx.prof = prof; ctx.last = &last;
  mms_rollout<MMADevCtx, T, MMS_DK>(ctx, D, H, dt, scale, shift, drift, pol, target, precis, mx + (size_t)b * nx,
                                    Sxx + (size_t)b * nx * nx, cost ? cost + b : (T*)nullptr, (size_t)B,
                                    traj_mu ? traj_mu + (size_t)b * nx : (T*)nullptr, (size_t)B * nx,
                                    traj_S ? traj_S + (size_t)b * nx * nx : (T*)nullptr, (size_t)B * nx * nx, sm, &ok);
  if (!ok && threadIdx.x == 0 && status) { atomicMax(status, (int)gridDim.x - b); status[1] = 0; }
}

// stage profile of block 0 (device long long[16], accumulated cycles per stage id): only in a -DMMS_PROFILE build
// (tools/profile_small.py); the shipped library has no global state
#ifdef MMS_PROFILE
static long long* mm_small_prof = nullptr;
extern "C" void mm_rollout_small_set_profile(void* device_buffer) { mm_small_prof = (long long*)device_buffer; }
#else
static long long* const mm_small_prof = nullptr;
#endif

// shapes the one-launch kernel takes: ne, nd <= 8, policy M <= 128, and the LDS image must fit
static size_t mm_rollout_small_lds(int nx, int na, int Md, int Mpol) {
  return (size_t)mms_rollout_scratch(nx, na, Md, Mpol, MMS_THREADS / 64) * sizeof(double);
}

extern "C" int mm_rollout_small_supported(int nx, int na, int drift_M, int policy_M) {
  if (nx <= 0 || na <= 0 || na > nx || nx > MMC_NX || na > MMC_NA) return 0;
  const int ne = nx + na, nd = ne + 1;
  if (nd > MMS_DK || policy_M > 128 || policy_M <= 0 || drift_M <= 0) return 0;
  return mm_rollout_small_lds(nx, na, drift_M, policy_M) <= 152 * 1024 ? 1 : 0;
}

int mm_rollout_small_launch(const void* drift_packed, size_t drift_bytes, int drift_M, const void* policy_packed, size_t policy_bytes,
                            int policy_M, int dtype, int B, int H, double dt, const MMComposeDims& D, double scale, double shift,
                            const void* target, const void* precis, void* mx, void* Sxx, void* cost, void* traj_mu, void* traj_S,
                            int32_t* status, hipStream_t s) {
  const int nx = D.nx, ne = D.ne, nd = D.nd;
  // the f64 blocks of both packs do not depend on the element type (they precede the T blocks): read through the f64 layout
  const MMModelLayout dl = mm_model_layout(nx, drift_M, nd, dtype, 1);
  const MMModelLayout pl = mm_model_layout(1, policy_M, ne, dtype, 1);
  if (drift_bytes < dl.total) return MM_E_NO_C;
  if (policy_bytes < pl.Cm) return MM_E_WORKSPACE;
  const char* dp = (const char*)drift_packed; const char* pp = (const char*)policy_packed;
  MMSmallModel drift{nx, drift_M, nd, dl.Mp, (const double*)(dp + dl.Z64), (const double*)(dp + dl.beta64), (const double*)(dp + dl.ls2),
                     (const double*)(dp + dl.var), (const double*)(dp + dl.meanc), (const double*)(dp + dl.Cm)};
  MMSmallModel pol{1, policy_M, ne, 0, (const double*)(pp + pl.Z64), (const double*)(pp + pl.beta64), (const double*)(pp + pl.ls2),
                   (const double*)(pp + pl.var), (const double*)(pp + pl.meanc), nullptr};
  const size_t lds = mm_rollout_small_lds(nx, D.na, drift_M, policy_M);
#define MMS_LAUNCH(T_)                                                                                                            \
  do {                                                                                                                            \
    if (lds > 64 * 1024) {                                                                                                        \
      hipError_t e = hipFuncSetAttribute((const void*)k_rollout_small<T_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      if (e != hipSuccess) return (int)e;                                                                                         \
    }                                                                                                                             \
    hipLaunchKernelGGL((k_rollout_small<T_>), dim3(B), dim3(MMS_THREADS), lds, s, D, H, dt, scale, shift, drift, pol, B,          \
                       (const T_*)target, (const T_*)precis, (T_*)mx, (T_*)Sxx, (T_*)cost, (T_*)traj_mu, (T_*)traj_S, status,    \
                       mm_small_prof);                                                                                          \
  } while (0)
  if (dtype == MM_F64) MMS_LAUNCH(double); else MMS_LAUNCH(float);
#undef MMS_LAUNCH
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
