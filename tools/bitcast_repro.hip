// Compiler pitfall found in round 1 (ROCm 7.2 hipcc, gfx950): the high dword of a vector ELEMENT taken with
// __builtin_bit_cast(unsigned long long, c[r]) >> 32 in an unrolled loop compiles to a test of element 0 only.
//   hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only tools/bitcast_repro.hip -o - | grep -c 0x7ff
// prints 1 for k_bad (one v_and on v1) and 4 for k_good (__double2hiint).  The f64 reduce and the backward
// kernel chose their expm1 tier from such a max: they sampled a quarter of the entries until this was found.
#include <hip/hip_runtime.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <bool GOOD>
__device__ __forceinline__ void body(const double* a, const double* b, double* out) {
  f64x4 c = {a[threadIdx.x], a[threadIdx.x + 64], a[threadIdx.x + 128], a[threadIdx.x + 192]};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[threadIdx.x + 256], b[threadIdx.x], c, 0, 0, 0);
  unsigned int mxh = 0u;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    unsigned int hi;
    if (GOOD) { const double v = c[r]; hi = (unsigned int)__double2hiint(v); }
    else hi = (unsigned int)(__builtin_bit_cast(unsigned long long, c[r]) >> 32);
    const unsigned int ah = hi & 0x7fffffffu;
    mxh = ah > mxh ? ah : mxh;
  }
  if (!__any(mxh >= 0x3f900000u)) out[threadIdx.x] = c[0] + c[1] + c[2] + c[3];
  else out[threadIdx.x] = c[0] * c[1] * c[2] * c[3];
}
__global__ void k_bad(const double* a, const double* b, double* out) { body<false>(a, b, out); }
__global__ void k_good(const double* a, const double* b, double* out) { body<true>(a, b, out); }
