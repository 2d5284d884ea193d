// Micro-benchmark: VALU issue rates on gfx950 (plain vs packed f32 FMA, f64 FMA, v_exp, with/without bf16 MFMA).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define N_ITER 2000
#define NCH 16

template <int MODE>
__global__ __launch_bounds__(256) void kern(float* out, float seed) {
  float x = seed + threadIdx.x * 1e-6f;
  float acc[NCH];
  f32x2 acc2[NCH];
  double accd[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) { acc[i] = x + i; acc2[i] = (f32x2){x + i, x - i}; accd[i] = x + i; }
  f32x16 c = {0}; f32x16 c2 = {0};
  bf16x8 av, bv;
#pragma unroll
  for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(x + i); bv[i] = (__bf16)(x - i); }
  for (int it = 0; it < N_ITER; ++it) {
    if (MODE == 0) {          // plain v_fma_f32: NCH independent chains
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i) acc[i] = fmaf(acc[i], 0.999f, 0.001f);
    } else if (MODE == 1) {   // packed
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i) acc2[i] = __builtin_elementwise_fma(acc2[i], (f32x2){0.999f, 0.999f}, (f32x2){0.001f, 0.001f});
    } else if (MODE == 2) {   // f64 fma
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i) accd[i] = fma(accd[i], 0.999, 0.001);
    } else if (MODE == 3) {   // packed + bf16 MFMA interleaved (64 pk_fma + 2 mfma per iter)
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c2, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i) acc2[i] = __builtin_elementwise_fma(acc2[i], (f32x2){0.999f, 0.999f}, (f32x2){0.001f, 0.001f});
    } else if (MODE == 4) {   // bf16 MFMA only: 8 per iter on 2 accumulators
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c2, 0, 0, 0);
      }
    } else if (MODE == 5) {   // v_exp_f32
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_exp2f(acc[i]);
    } else if (MODE == 6) {   // plain fma + bf16 MFMA interleaved (64 fma + 2 mfma)
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c2, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i) acc[i] = fmaf(acc[i], 0.999f, 0.001f);
    } else if (MODE == 7) {   // f32 MFMA 32x32x2 + plain fma
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, c, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, c2, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i) acc[i] = fmaf(acc[i], 0.999f, 0.001f);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += acc[i] + acc2[i][0] + acc2[i][1] + (float)accd[i];
  for (int i = 0; i < 16; ++i) s += c[i] + c2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, float* d_out, double insts_per_iter, double flops_per_inst) {
  int nblk = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<MODE><<<nblk, 256>>>(d_out, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<MODE><<<nblk, 256>>>(d_out, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double waves = (double)nblk * 4;
  double winst = waves * N_ITER * insts_per_iter;
  double per_simd_per_s = winst / (ms * 1e-3) / 1024.0;
  printf("%-34s blocks/CU=%d  %.3f ms  wave-inst/SIMD/s = %.3e  -> cycles/inst @2.4GHz = %.2f  (%.1f TFLOP/s)\n",
         name, blocks_per_cu, ms, per_simd_per_s, 2.4e9 / per_simd_per_s, winst * 64 * flops_per_inst / (ms * 1e-3) / 1e12);
}

int main() {
  float* d_out; hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
  for (int bpc : {1, 2, 4}) {
    run<0>("v_fma_f32", bpc, d_out, 64, 2);
    run<1>("v_pk_fma_f32", bpc, d_out, 64, 4);
    run<2>("v_fma_f64", bpc, d_out, 64, 2);
    run<5>("v_exp_f32", bpc, d_out, 64, 1);
    run<4>("mfma bf16 32x32x16 only", bpc, d_out, 8, 32768.0 / 64);
    run<3>("64 pk_fma + 2 mfma_bf16 (per VALU)", bpc, d_out, 64, 4);
    run<6>("64 fma + 2 mfma_bf16 (per VALU)", bpc, d_out, 64, 2);
    run<7>("64 fma + 2 mfma_f32x2 (per VALU)", bpc, d_out, 64, 2);
  }
  return 0;
}
