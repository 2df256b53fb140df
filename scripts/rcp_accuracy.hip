// How accurate are v_rcp_f64 / v_rsq_f64 on gfx950?  (decides how many Newton steps the pivot chains need)
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/rcp_accuracy scripts/rcp_accuracy.hip && ./gpurun_out/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* q0, double* q1, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = x[i];
  double r = __builtin_amdgcn_rcp(d);
  r0[i] = r;
  double e = fma(-d, r, 1.0);
  r = fma(e, r, r);
  r1[i] = r;
  e = fma(-d, r, 1.0);
  r2[i] = fma(e, r, r);
  double s = __builtin_amdgcn_rsq(d);
  q0[i] = s;
  const double h = 0.5 * s, t = fma(-d * s, h, 0.5);        // one Newton step for 1/sqrt
  q1[i] = fma(s, t, s);
}
int main() {
  const int n = 1 << 20;
  std::vector<double> h(n), a(n), b(n), c(n), d(n), e(n);
  unsigned long long st = 88172645463325252ULL;
  for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = std::ldexp(1.0 + (st >> 11) * (1.0 / 9007199254740992.0), (int)(st % 61) - 30); }
  double *dx, *d0, *d1, *d2, *d3, *d4;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8); hipMalloc(&d4, n * 8);
  hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d0, d1, d2, d3, d4, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(d.data(), d3, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(e.data(), d4, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0, s0 = 0, s1 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / (long double)h[i], u = 1.0L / sqrtl((long double)h[i]);
    m0 = fmax(m0, (double)fabsl((a[i] - t) / t)); m1 = fmax(m1, (double)fabsl((b[i] - t) / t)); m2 = fmax(m2, (double)fabsl((c[i] - t) / t));
    s0 = fmax(s0, (double)fabsl((d[i] - u) / u)); s1 = fmax(s1, (double)fabsl((e[i] - u) / u));
  }
  printf("max relative error: v_rcp_f64 %.3e (2^%.1f), +1 Newton %.3e, +2 Newton %.3e;  v_rsq_f64 %.3e (2^%.1f), +1 Newton %.3e\n",
         m0, log2(m0), m1, m2, s0, log2(s0), s1);
  return 0;
}
