// Practical read ceiling of one MI355X: a pure streaming read (16-byte loads, a few in flight per lane, the sum
// kept so that nothing is optimised away) over a buffer far beyond the 256 MB of Infinity Cache, for several
// launch shapes.    hipcc --offload-arch=gfx950 -O3 -o gpurun_out/read_ceiling scripts/read_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int UNR>
__global__ __launch_bounds__(1024) void read_kernel(const double2* __restrict__ x, size_t n2, double* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0;
  for (; i + (UNR - 1) * stride < n2; i += UNR * stride) {
    double2 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = x[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNR; ++u) s += v[u].x + v[u].y;
  }
  for (; i < n2; i += stride) s += x[i].x + x[i].y;
  if (s == 1.2345e300) out[0] = s;
}

template <int UNR>
float run(const double2* x, size_t n2, double* out, int blocks, int threads, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  read_kernel<UNR><<<blocks, threads>>>(x, n2, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) read_kernel<UNR><<<blocks, threads>>>(x, n2, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? atof(argv[1]) : 2.0) * (1ull << 30);
  const size_t n2 = bytes / 16;
  double2* x; double* out;
  hipMalloc(&x, bytes); hipMalloc(&out, 8);
  hipMemset(x, 0, bytes);
  printf("buffer %.2f GiB\n", bytes / double(1ull << 30));
  for (int threads : {256, 512, 1024})
    for (int per_cu : {1, 2, 4, 8}) {
      const int blocks = 256 * per_cu * (1024 / threads);
      const float m1 = run<1>(x, n2, out, blocks, threads, 10);
      const float m2 = run<2>(x, n2, out, blocks, threads, 10);
      const float m4 = run<4>(x, n2, out, blocks, threads, 10);
      const float m8 = run<8>(x, n2, out, blocks, threads, 10);
      printf("threads %4d blocks %5d : unr1 %6.0f  unr2 %6.0f  unr4 %6.0f  unr8 %6.0f GB/s\n", threads, blocks,
             bytes / m1 * 1e-6, bytes / m2 * 1e-6, bytes / m4 * 1e-6, bytes / m8 * 1e-6);
    }
  return 0;
}
