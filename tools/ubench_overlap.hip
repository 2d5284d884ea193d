// Micro-benchmark: which VALU instructions overlap with v_mfma_f32_32x32x16_bf16 on one SIMD of gfx950?
// Per iteration and wave: NM MFMAs (two accumulator chains) + 96 VALU instructions of one kind
// (16 independent chains x 6), the shape of one k_qred_f32_mfma wave tile; MFMAs and VALU are interleaved
// (1 MFMA : 16 VALU) with sched_group_barrier.  Reported: ns per wave-iteration per SIMD for VALU only,
// MFMA only, and both; "both ~ max" means overlap, "both ~ sum" means the two serialise.
// Build: hipcc -O3 -w --offload-arch=gfx950 tools/ubench_overlap.hip -o ubench_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define N_ITER 4000

enum { OP_FMA, OP_PKFMA, OP_ADD, OP_MUL, OP_PKMUL, OP_PKADD, OP_MAX3, OP_EXP, OP_FMA64, OP_NONE };

template <int OP>
__device__ __forceinline__ void valu(f32x2 (&p)[16], float (&q)[16], double (&dd)[16]) {
  const f32x2 k0 = {0.999f, 0.999f}, k1 = {0.001f, 0.001f};
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (OP == OP_FMA) q[i] = fmaf(q[i], 0.999f, 0.001f);
      else if (OP == OP_PKFMA) p[i] = __builtin_elementwise_fma(p[i], k0, k1);
      else if (OP == OP_ADD) q[i] = q[i] + 0.001f;
      else if (OP == OP_MUL) q[i] = q[i] * 0.999f;
      else if (OP == OP_PKMUL) p[i] = p[i] * k0;
      else if (OP == OP_PKADD) p[i] = p[i] + k1;
      else if (OP == OP_MAX3) q[i] = fmaxf(fmaxf(q[i], q[(i + 1) & 15]), q[(i + 2) & 15] );
      else if (OP == OP_EXP) q[i] = __builtin_amdgcn_exp2f(q[i]);
      else if (OP == OP_FMA64) dd[i] = fma(dd[i], 0.999, 0.001);
    }
}

template <int OP, int NM>
__global__ __launch_bounds__(256) void kern(float* out, float seed) {
  float x = seed + threadIdx.x * 1e-6f;
  f32x2 p[16]; float q[16]; double dd[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { p[i] = (f32x2){x + i, x - i}; q[i] = x + i; dd[i] = x + i; }
  f32x16 c0 = {0}, c1 = {0};
  bf16x8 av, bv;
#pragma unroll
  for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(x + i); bv[i] = (__bf16)(x - i); }
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int m = 0; m < NM / 2; ++m) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c1, 0, 0, 0);
    }
    valu<OP>(p, q, dd);
    if constexpr (NM > 0 && OP != OP_NONE) {
#pragma unroll
      for (int g = 0; g < NM; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 96 / NM, 0);
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += p[i][0] + p[i][1] + q[i] + (float)dd[i] + c0[i] + c1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int NM>
double run(int wps, float* d_out) {
  int nblk = 256 * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<OP, NM><<<nblk, 256>>>(d_out, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<OP, NM><<<nblk, 256>>>(d_out, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / N_ITER / wps;
}

template <int OP>
void line(const char* name, int wps, float* d_out) {
  double v = run<OP, 0>(wps, d_out), m = run<OP_NONE, 6>(wps, d_out), b = run<OP, 6>(wps, d_out);
  printf("%-14s waves/SIMD=%d  96 VALU %.1f ns | 6 MFMA %.1f ns | both %.1f ns | sum %.1f  max %.1f  -> overlap %.0f%%\n",
         name, wps, v, m, b, v + m, v > m ? v : m, 100.0 * (v + m - b) / (v < m ? v : m));
}

int main() {
  float* d_out; hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
  for (int wps : {1, 2, 4}) {
    line<OP_FMA>("v_fma_f32", wps, d_out);
    line<OP_PKFMA>("v_pk_fma_f32", wps, d_out);
    line<OP_ADD>("v_add_f32", wps, d_out);
    line<OP_MUL>("v_mul_f32", wps, d_out);
    line<OP_PKMUL>("v_pk_mul_f32", wps, d_out);
    line<OP_PKADD>("v_pk_add_f32", wps, d_out);
    line<OP_MAX3>("v_max3_f32", wps, d_out);
    line<OP_EXP>("v_exp_f32", wps, d_out);
    line<OP_FMA64>("v_fma_f64", wps, d_out);
  }
  return 0;
}
