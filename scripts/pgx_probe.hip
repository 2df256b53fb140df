// Stand-alone timing probe of the flat exact Polya-Gamma kernel (csrc/btf_pg_exact.h) at the C4 shape:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I functionalmf_amd/csrc scripts/pgx_probe.hip -o build/pgx_probe
//   build/pgx_probe [N MT b]
// Prints the average launch time of each (waves, cells-per-lane) shape and the mean / variance of omega against
// the closed forms (a sanity check, not the statistical test: tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "btf_pg_exact.h"

using namespace btf;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NW, int CPL>
static void run(const char* name, const double* Bv, double* Cv, double* CwT, const double* W, const double* V, int N, int MT,
                int ldv, int ldw, int reps) {
  constexpr int K = 5;
  auto kern = pgx_tile_kernel<K, NW, CPL>;
  const size_t lds = pgx_tile_lds(NW, CPL);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((MT + 63) / 64, (N + NW * CPL - 1) / (NW * CPL));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, 0, Bv, Cv, CwT, W, V, N, MT, ldv, ldw, 1000ULL + i, 1, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, 0, Bv, Cv, CwT, W, V, N, MT, ldv, ldw, 2000ULL + i, 1, 1);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-14s lds %6zu B  grid %4u x %3u  %8.1f us per launch\n", name, lds, grid.x, grid.y, 1000.0 * ms / reps);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 512, MT = argc > 2 ? atoi(argv[2]) : 256 * 64, K = 5;
  const double b = argc > 3 ? atof(argv[3]) : 4.0;
  const int ldv = (MT + 127) / 128 * 128, ldw = (N + 127) / 128 * 128;
  std::mt19937_64 rng(1);
  std::normal_distribution<double> nrm;
  std::vector<double> W((size_t)N * K), V((size_t)MT * K), B((size_t)N * ldv, 0.0);
  for (auto& w : W) w = nrm(rng);
  const int T = 64;
  for (int j = 0; j < MT / T; ++j) {
    double acc[8] = {0};
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < K; ++k) { acc[k] += 0.1 * nrm(rng); V[((size_t)j * T + t) * K + k] = acc[k]; }
  }
  for (int i = 0; i < N; ++i) for (int jt = 0; jt < MT; ++jt) B[(size_t)i * ldv + jt] = b;
  double *dW, *dV, *dB, *dCv, *dCw;
  CK(hipMalloc(&dW, W.size() * 8)); CK(hipMalloc(&dV, V.size() * 8)); CK(hipMalloc(&dB, B.size() * 8));
  CK(hipMalloc(&dCv, B.size() * 8)); CK(hipMalloc(&dCw, (size_t)MT * ldw * 8));
  CK(hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dV, V.data(), V.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dCv, 0, B.size() * 8)); CK(hipMemset(dCw, 0, (size_t)MT * ldw * 8));
  const int reps = 20;
  run<4, 8>("NW4 CPL8", dB, dCv, dCw, dW, dV, N, MT, ldv, ldw, reps);
  run<4, 4>("NW4 CPL4", dB, dCv, dCw, dW, dV, N, MT, ldv, ldw, reps);
  run<4, 16>("NW4 CPL16", dB, dCv, dCw, dW, dV, N, MT, ldv, ldw, reps);
  run<8, 8>("NW8 CPL8", dB, dCv, dCw, dW, dV, N, MT, ldv, ldw, reps);
  run<8, 4>("NW8 CPL4", dB, dCv, dCw, dW, dV, N, MT, ldv, ldw, reps);
  run<4, 2>("NW4 CPL2", dB, dCv, dCw, dW, dV, N, MT, ldv, ldw, reps);
  run<4, 1>("NW4 CPL1", dB, dCv, dCw, dW, dV, N, MT, ldv, ldw, reps);
  // moments of the last launch against E = b tanh(z)/(4z)... (z = psi/2), per cell z-scores pooled
  std::vector<double> Cv(B.size()), Cw((size_t)MT * ldw);
  CK(hipMemcpy(Cv.data(), dCv, Cv.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(Cw.data(), dCw, Cw.size() * 8, hipMemcpyDeviceToHost));
  double s1 = 0, s2 = 0, mism = 0;
  long long n = 0;
  for (int i = 0; i < N; ++i)
    for (int jt = 0; jt < MT; ++jt) {
      double psi = 0;
      for (int k = 0; k < K; ++k) psi += W[(size_t)i * K + k] * V[(size_t)jt * K + k];
      const double a = std::fabs(psi);
      const double m = a > 1e-6 ? b / (2 * a) * std::tanh(a / 2) : b / 4;
      const double ch = std::cosh(a / 2);
      const double v = a > 1e-3 ? b / (4 * a * a * a) * (std::sinh(a) - a) / (ch * ch) : b / 24;
      const double om = Cv[(size_t)i * ldv + jt];
      const double zs = (om - m) / std::sqrt(v);
      s1 += zs; s2 += zs * zs; ++n;
      if (om != Cw[(size_t)jt * ldw + i]) mism += 1;
    }
  printf("z-scores over %lld cells: mean %.5f (sd of the mean %.5f)  variance %.5f   layout mismatches %.0f\n", n, s1 / n,
         1.0 / std::sqrt((double)n), s2 / n - (s1 / n) * (s1 / n), mism);
  return 0;
}
