// Read-bandwidth ceilings for the access pattern of the pathwise kernel, no arithmetic:
//   A  chip-wide contiguous: consecutive waves read consecutive 1 KB pieces (grid-stride)
//   B  one private sequential stream per wave, 4 x 1 KB per step (the pathwise kernel's pattern: 2048 waves,
//      each walking its own contiguous region), NPRE steps in flight
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_stream.hip -o /tmp/ubench_stream && /tmp/ubench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_contig(const f4* __restrict__ p, size_t n4, float* out) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  size_t i = tid;
  for (; i + 3 * nth < n4; i += 4 * nth) {
    const f4 a = p[i], b = p[i + nth], c = p[i + 2 * nth], d = p[i + 3 * nth];
    acc += a + b + c + d;
  }
  for (; i < n4; i += nth) acc += p[i];
  if (acc.x + acc.y + acc.z + acc.w == 1.2345f) out[tid] = acc.x;
}

template <int NPRE>
__global__ __launch_bounds__(512) void k_private(const f4* __restrict__ p, size_t per_wave4, float* out) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const f4* q = p + (size_t)wave * per_wave4 + lane;            // this wave's region; a step = 4 x 64 f4 = 4 KB
  const size_t steps = per_wave4 / 256;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  f4 buf[NPRE][4];
#pragma unroll
  for (int r = 0; r < NPRE; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) buf[r][j] = q[(size_t)(r < (int)steps ? r : 0) * 256 + j * 64];
  for (size_t s = 0; s < steps; s += NPRE) {
#pragma unroll
    for (int r = 0; r < NPRE; ++r) {
      const f4 v = buf[r][0] + buf[r][1] + buf[r][2] + buf[r][3];
      const size_t nx = s + NPRE + r < steps ? s + NPRE + r : steps - 1;   // clamped, unconditional
#pragma unroll
      for (int j = 0; j < 4; ++j) buf[r][j] = q[nx * 256 + j * 64];
      acc += v;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345f) out[wave] = acc.x;
}

int main() {
  const size_t bytes = 792723456ull;                             // the C5 shard's weights per step
  const int nwaves = 2048;
  const size_t per_wave4 = bytes / 16 / nwaves / 256 * 256;      // f4 elements per wave, whole 4 KB steps
  const size_t n4 = per_wave4 * nwaves;
  f4* p; float* out;
  hipMalloc(&p, n4 * 16); hipMalloc(&out, 1 << 22);
  hipMemset(p, 0, n4 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto launch, const char* name) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-34s %7.1f us  %6.2f TB/s\n", name, ms * 1e3, n4 * 16 / (ms * 1e-3) / 1e12);
  };
  time([&] { hipLaunchKernelGGL(k_contig, dim3(256 * 4), dim3(512), 0, 0, p, n4, out); }, "A contiguous, 1024 x 512 threads");
  time([&] { hipLaunchKernelGGL(k_contig, dim3(256), dim3(512), 0, 0, p, n4, out); }, "A contiguous, 256 x 512 threads");
  time([&] { hipLaunchKernelGGL((k_private<2>), dim3(256), dim3(512), 0, 0, p, per_wave4, out); }, "B private streams, 2 steps ahead");
  time([&] { hipLaunchKernelGGL((k_private<3>), dim3(256), dim3(512), 0, 0, p, per_wave4, out); }, "B private streams, 3 steps ahead");
  time([&] { hipLaunchKernelGGL((k_private<4>), dim3(256), dim3(512), 0, 0, p, per_wave4, out); }, "B private streams, 4 steps ahead");
  return 0;
}
