// A host program on the C ABI alone (include/gpflowpilco_mm.h + the HIP runtime; no Python, no torch):
// what a compiled-language binding of the reference's moment_matching handler would do.
//
//   hipcc -O2 --offload-arch=gfx950 -Iinclude examples/abi_host.cpp -Lgpflowpilco_amd -lgpflowpilco_mm \
//         -Wl,-rpath,$PWD/gpflowpilco_amd -o /tmp/abi_host
//   /tmp/abi_host in.bin out.bin
//
// in.bin : int32 L, M, d, B, dtype (0 f32 / 1 f64), flags; then float64 Z[L,M,d], ls[L,d], var[L], beta[L,M],
//          C[L,M,M], mean_c[L], mu[B,d], Sigma[B,d,d]
// out.bin: float64 f1[B,L], Sff[B,L,L], cross_pre[B,d,L]; int32 status
// (tests/test_gpu_parity.py::test_c_abi_from_a_compiled_host writes in.bin and checks out.bin.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gpflowpilco_mm.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <typename T>
static std::vector<T> cast(const std::vector<double>& v) { return std::vector<T>(v.begin(), v.end()); }

template <typename T>
static int run(int L, int M, int d, int B, int dtype, int flags, const std::vector<double>& Z, const std::vector<double>& ls,
               const std::vector<double>& var, const std::vector<double>& beta, const std::vector<double>& C,
               const std::vector<double>& mc, const std::vector<double>& mu, const std::vector<double>& Sg, FILE* fo) {
  auto up = [](const void* h, size_t n, void** dptr) {
    if (hipMalloc(dptr, n) != hipSuccess) return 1;
    return hipMemcpy(*dptr, h, n, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
  };
  void *dZ, *dls, *dvar, *dbeta, *dC, *dmc, *dmu, *dS;
  if (up(Z.data(), Z.size() * 8, &dZ) || up(ls.data(), ls.size() * 8, &dls) || up(var.data(), var.size() * 8, &dvar) ||
      up(beta.data(), beta.size() * 8, &dbeta) || up(C.data(), C.size() * 8, &dC) || up(mc.data(), mc.size() * 8, &dmc)) return 2;
  const std::vector<T> muT = cast<T>(mu), SgT = cast<T>(Sg);
  if (up(muT.data(), muT.size() * sizeof(T), &dmu) || up(SgT.data(), SgT.size() * sizeof(T), &dS)) return 2;

  const size_t pbytes = mm_packed_model_bytes(L, M, d, dtype, 1);
  if (pbytes == 0) { fprintf(stderr, "unsupported shape\n"); return 3; }
  void* packed; HIP_OK(hipMalloc(&packed, pbytes));
  int rc = mm_pack_model(packed, pbytes, L, M, d, dtype, (const double*)dZ, (const double*)dls, (const double*)dvar,
                         (const double*)dbeta, (const double*)dC, (const double*)dmc, nullptr);
  if (rc) { fprintf(stderr, "mm_pack_model: %d\n", rc); return 3; }
  const size_t wbytes = mm_workspace_bytes(B, L, M, d, dtype, flags);
  void *ws, *f1, *Sff, *cross; int32_t* status;
  HIP_OK(hipMalloc(&ws, wbytes));
  HIP_OK(hipMalloc(&f1, (size_t)B * L * sizeof(T))); HIP_OK(hipMalloc(&Sff, (size_t)B * L * L * sizeof(T)));
  HIP_OK(hipMalloc(&cross, (size_t)B * d * L * sizeof(T)));
  HIP_OK(hipMalloc((void**)&status, 4 * sizeof(int32_t))); HIP_OK(hipMemset(status, 0, 4 * sizeof(int32_t)));
  rc = mm_moment_match(packed, pbytes, L, M, d, dtype, B, dmu, dS, flags, 0.0, f1, Sff, cross, ws, wbytes, status, nullptr);
  if (rc) { fprintf(stderr, "mm_moment_match: %d\n", rc); return 3; }
  HIP_OK(hipDeviceSynchronize());
  auto down = [&](const void* dptr, size_t n) {
    std::vector<T> h(n);
    if (hipMemcpy(h.data(), dptr, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    std::vector<double> o(h.begin(), h.end());
    return fwrite(o.data(), 8, n, fo) == n ? 0 : 1;
  };
  if (down(f1, (size_t)B * L) || down(Sff, (size_t)B * L * L) || down(cross, (size_t)B * d * L)) return 2;
  int32_t st[4]; HIP_OK(hipMemcpy(st, status, sizeof(st), hipMemcpyDeviceToHost));
  fwrite(st, sizeof(int32_t), 1, fo);
  printf("abi %d: L=%d M=%d d=%d B=%d dtype=%d packed %zu B workspace %zu B status %d\n", mm_abi_version(), L, M, d, B, dtype,
         pbytes, wbytes, st[0]);
  return 0;
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
  FILE* fi = fopen(argv[1], "rb"); if (!fi) { perror(argv[1]); return 1; }
  int32_t hdr[6];
  if (fread(hdr, 4, 6, fi) != 6) return 1;
  const int L = hdr[0], M = hdr[1], d = hdr[2], B = hdr[3], dtype = hdr[4], flags = hdr[5];
  auto rd = [&](size_t n) { std::vector<double> v(n); if (fread(v.data(), 8, n, fi) != n) { fprintf(stderr, "short read\n"); exit(1); } return v; };
  const auto Z = rd((size_t)L * M * d), ls = rd((size_t)L * d), var = rd(L), beta = rd((size_t)L * M), C = rd((size_t)L * M * M),
             mc = rd(L), mu = rd((size_t)B * d), Sg = rd((size_t)B * d * d);
  fclose(fi);
  FILE* fo = fopen(argv[2], "wb"); if (!fo) { perror(argv[2]); return 1; }
  const int rc = dtype == MM_F64 ? run<double>(L, M, d, B, dtype, flags, Z, ls, var, beta, C, mc, mu, Sg, fo)
                                 : run<float>(L, M, d, B, dtype, flags, Z, ls, var, beta, C, mc, mu, Sg, fo);
  fclose(fo);
  return rc;
}
