// Read bandwidth of the accumulation's access pattern against the row length of the statistic:
//   workgroup (tile, chunk) of 16 waves; wave w reads rows chunk*rpb + w, + 16, ...; a row is one 1-KiB segment
//   (16 bytes per lane) at  base + (row * ld + tile * 128) * 8   ("rows": the layout of A_wT / A_v)
//   or at                   base + ((tile * Rdim + row) * 128) * 8   ("tiled": tile-major, every workgroup streams
//   one contiguous range).     hipcc --offload-arch=gfx950 -O3 -o /tmp/stride_probe scripts/stride_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int UNR, bool TILED>
__global__ __launch_bounds__(1024) void probe(const double2* __restrict__ x, int Rdim, int ld, int rpb, double* __restrict__ out) {
  const int ntiles = ld / 128;
  const int b = blockIdx.x;
  const int chunk = b / ntiles, tile = b - chunk * ntiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = chunk * rpb, r1 = min(r0 + rpb, Rdim);
  double s = 0.0;
  for (int r = r0 + wave; r < r1; r += 16 * UNR) {
    double2 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int rr = min(r + 16 * u, r1 - 1);
      const size_t idx = TILED ? ((size_t)tile * Rdim + rr) * 64 + lane : ((size_t)rr * ld + tile * 128) / 2 + lane;
      v[u] = x[idx];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) s += v[u].x + v[u].y;
  }
  if (s == 1.2345e300) out[0] = s;
}

template <bool TILED>
float run(const double2* x, int Rdim, int ld, int rpb, double* out, int reps) {
  const int grid = (ld / 128) * ((Rdim + rpb - 1) / rpb);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  probe<2, TILED><<<grid, 1024>>>(x, Rdim, ld, rpb, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) probe<2, TILED><<<grid, 1024>>>(x, Rdim, ld, rpb, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const size_t maxbytes = 3ull << 30;
  double2* x; double* out;
  hipMalloc(&x, maxbytes); hipMalloc(&out, 8);
  hipMemset(x, 0, maxbytes);
  struct Case { const char* name; int Rdim, ld, rpb; };
  const Case cases[] = {{"C3 W (16384 x 512)", 16384, 512, 512},       {"C3 V (512 x 16384)", 512, 16384, 512},
                        {"C5/8 W slab (65536 x 512)", 65536, 512, 1024}, {"C5/8 V slab (4096 x 8192)", 4096, 8192, 512},
                        {"C5 W (65536 x 4096)", 65536, 4096, 2048},    {"C5 V (4096 x 65536)", 4096, 65536, 2048},
                        {"W slab, rows of 1024", 32768, 1024, 1024},   {"W slab, rows of 2048", 16384, 2048, 1024}};
  for (const Case& c : cases) {
    const double bytes = (double)c.Rdim * c.ld * 8;
    // a second buffer region between repetitions would defeat the Infinity Cache; here each launch re-reads the same
    // range: sizes below 256 MB are cache-resident (as the Gibbs sweep's statistic is at C3)
    const float a = run<false>(x, c.Rdim, c.ld, c.rpb, out, 20), t = run<true>(x, c.Rdim, c.ld, c.rpb, out, 20);
    printf("%-30s %7.1f MB  rows %7.1f us = %5.0f GB/s   tiled %7.1f us = %5.0f GB/s\n", c.name, bytes / 1e6, 1e3 * a, bytes / a * 1e-6,
           1e3 * t, bytes / t * 1e-6);
  }
  return 0;
}
