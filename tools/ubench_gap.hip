// Micro-benchmark: how many VALU instructions hide inside the issue gap of a matrix instruction on ONE SIMD
// of gfx950, at 1 / 2 / 3 waves per SIMD?  (Settles the question tools/ubench_overlap.hip left open: at 16
// fillers per MFMA every model predicts "sum".)
//
// Per loop iteration and wave: 8 matrix instructions on two accumulator chains, each followed by K independent
// filler VALU instructions of one kind (inline asm, order pinned with sched_barrier: 1 MFMA, K VALU, ...).
// Reported per (filler, K, waves/SIMD): shader cycles per MFMA gap (s_memtime around the loop, median over
// waves) and ns per gap per SIMD from HIP events (all 1024 SIMDs busy).  MI355X_MICROARCH.md predicts
//   gap = max(32, 8 + sum of filler issue costs)    for v_mfma_f32_32x32x16_bf16 and one wave per SIMD,
// the serial model of round 1 predicts gap = 32 + sum.
// Build: hipcc -O3 -w --offload-arch=gfx950 tools/ubench_gap.hip -o ubench_gap
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define N_ITER 2000
#define NMF 8

enum { OP_FMA, OP_PKFMA, OP_MAX3, OP_FMA64, OP_PKMUL };
enum { MF_BF16, MF_F64 };

// fillers are inline asm (exactly one instruction each; the SLP vectoriser would otherwise merge scalar FMAs
// into v_pk_fma_f32 and the unroller regroup them)
template <int OP>
__device__ __forceinline__ void filler(int i, f32x2 (&p)[16], float (&q)[16], double (&dd)[16]) {
  i &= 15;
  const float k0 = 0.999f, k1 = 0.001f;
  const f32x2 k0p = {0.999f, 0.999f}, k1p = {0.001f, 0.001f};
  const double k0d = 0.999, k1d = 0.001;
  if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(k0), "v"(k1));
  else if (OP == OP_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(k0p), "v"(k1p));
  else if (OP == OP_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(k0p));
  else if (OP == OP_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(q[(i + 5) & 15]), "v"(q[(i + 9) & 15]));
  else if (OP == OP_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dd[i]) : "v"(k0d), "v"(k1d));
}

template <int MF, int OP, int K>
__global__ __launch_bounds__(256) void kern(float* out, long long* cyc, float seed) {
  float x = seed + threadIdx.x * 1e-6f;
  f32x2 p[16]; float q[16]; double dd[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { p[i] = (f32x2){x + i, x - i}; q[i] = x + i; dd[i] = x + i; }
  f32x16 c0 = {0}, c1 = {0};
  f64x4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
  bf16x8 av, bv;
#pragma unroll
  for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(x + i); bv[i] = (__bf16)(x - i); }
  double ad = x, bd = x * 0.5;
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int m = 0; m < NMF; ++m) {
      if (MF == MF_BF16) {
        if (m & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c1, 0, 0, 0);
        else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c0, 0, 0, 0);
      } else {
        if (m & 1) e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, bd, e1, 0, 0, 0);
        else e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, bd, e0, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K; ++k) filler<OP>(m * K + k, p, q, dd);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += p[i][0] + p[i][1] + q[i] + (float)dd[i] + c0[i] + c1[i];
  s += (float)(e0[0] + e0[1] + e0[2] + e0[3] + e1[0] + e1[1] + e1[2] + e1[3]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

static float* d_out; static long long* d_cyc;

template <int MF, int OP, int K>
void run(const char* mf, const char* name, int wps) {
  const int nblk = 256 * wps, nwave = nblk * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<MF, OP, K><<<nblk, 256>>>(d_out, d_cyc, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<MF, OP, K><<<nblk, 256>>>(d_out, d_cyc, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(nwave);
  hipMemcpy(h.data(), d_cyc, nwave * sizeof(long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double cyc_gap = (double)h[nwave / 2] / N_ITER / NMF / wps;      // per gap per SIMD (wps waves share it)
  const double ns_gap = ms * 1e6 / N_ITER / NMF / wps;
  printf("%-6s %-13s K=%2d waves/SIMD=%d  %7.1f cyc/gap/SIMD  %7.2f ns/gap/SIMD  (eff clock %.2f GHz)\n", mf, name, K, wps,
         cyc_gap, ns_gap, cyc_gap / ns_gap);
}

template <int MF, int OP>
void sweep(const char* mf, const char* name) {
  for (int wps : {1, 2, 3}) {
    run<MF, OP, 0>(mf, name, wps);
    run<MF, OP, 1>(mf, name, wps);
    run<MF, OP, 2>(mf, name, wps);
    run<MF, OP, 3>(mf, name, wps);
    run<MF, OP, 4>(mf, name, wps);
    run<MF, OP, 5>(mf, name, wps);
    run<MF, OP, 6>(mf, name, wps);
    run<MF, OP, 8>(mf, name, wps);
    run<MF, OP, 12>(mf, name, wps);
    run<MF, OP, 16>(mf, name, wps);
  }
}

int main() {
  hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
  hipMalloc(&d_cyc, 256 * 8 * 4 * sizeof(long long));
  sweep<MF_BF16, OP_FMA>("bf16", "v_fma_f32");
  sweep<MF_BF16, OP_PKFMA>("bf16", "v_pk_fma_f32");
  sweep<MF_BF16, OP_PKMUL>("bf16", "v_pk_mul_f32");
  sweep<MF_BF16, OP_MAX3>("bf16", "v_max3_f32");
  sweep<MF_BF16, OP_FMA64>("bf16", "v_fma_f64");
  sweep<MF_F64, OP_FMA64>("f64", "v_fma_f64");
  sweep<MF_F64, OP_FMA>("f64", "v_fma_f32");
  return 0;
}
