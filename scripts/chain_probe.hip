// Stand-alone cycle probe of the spectral sampler's pivot chains (csrc/btf_spectral.h) on ONE wave, alone on its CU:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I functionalmf_amd/csrc -I include scripts/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
// T = 64, S = 3, K = 5: 2K chains in lanes 0..9 (lane k ascends system k, lane K + k descends it), as in the kernels.
// Prints shader cycles (s_memtime) per variant and checks every variant's records against the combined chain's.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "btf_spectral.h"

using namespace btf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int T = 64, S = 3, K = 5, D1 = S + 1, RS = S + 2, Tp = T + S + 1;
constexpr int NV = 8;      // variants

#ifndef PF
#define PF 4
#endif

// ---- variant: factor with the band rows of PF pivots fetched ahead into registers, records written per batch ----
template <int SS, int B>
__device__ __forceinline__ bool factor_batched(const double* __restrict__ Pv, double* __restrict__ rec, int n_elim, double gk, SpecWinC<SS>& w) {
#pragma unroll
  for (int b = 0; b < SS; ++b) {
#pragma unroll
    for (int d = 0; d <= SS; ++d) w.c[b][d] = Pv[b * (SS + 1) + d];
    w.c[b][0] += gk;
  }
  bool bad = false;
  double nx[B][SS + 1];
#pragma unroll
  for (int u = 0; u < B; ++u)
#pragma unroll
    for (int d = 0; d <= SS; ++d) nx[u][d] = Pv[(u + SS) * (SS + 1) + d];
  int i = 0;
  for (; i + B <= n_elim; i += B) {
    double cur[B][SS + 1];
#pragma unroll
    for (int u = 0; u < B; ++u)
#pragma unroll
      for (int d = 0; d <= SS; ++d) cur[u][d] = nx[u][d];
    // the next batch's rows go out now (zero rows pad the band: reading past the last pivot is harmless)
#pragma unroll
    for (int u = 0; u < B; ++u)
#pragma unroll
      for (int d = 0; d <= SS; ++d) nx[u][d] = Pv[(i + B + u + SS) * (SS + 1) + d];
    double out[B][SS + 1];
#pragma unroll
    for (int u = 0; u < B; ++u) {
#pragma unroll
      for (int d = 0; d <= SS; ++d) w.c[SS][d] = cur[u][d];
      w.c[SS][0] += gk;
      const double d0 = w.c[0][0];
      bad |= !(d0 > 0.0);
      const double inv = rcp_cubic(d0);
      double l[SS + 1];
#pragma unroll
      for (int d = 1; d <= SS; ++d) l[d] = w.c[0][d] * inv;
#pragma unroll
      for (int b = 1; b <= SS; ++b)
#pragma unroll
        for (int a = b; a <= SS; ++a) w.c[b][a - b] = fma(-l[a], w.c[0][b], w.c[b][a - b]);
#pragma unroll
      for (int d = 1; d <= SS; ++d) out[u][d - 1] = l[d];
      out[u][SS] = inv;
#pragma unroll
      for (int b = 0; b < SS; ++b)
#pragma unroll
        for (int d = 0; d <= SS; ++d) w.c[b][d] = w.c[b + 1][d];
    }
#pragma unroll
    for (int u = 0; u < B; ++u)
#pragma unroll
      for (int d = 0; d <= SS; ++d) rec[(i + u) * (SS + 2) + d] = out[u][d];
  }
  for (; i < n_elim; ++i) spectral_factor_pivot<SS>(Pv, rec, i, gk, w, bad);
  return !bad;
}

// ---- variant: forward substitution with the records of B pivots fetched ahead ----
template <int SS, int B>
__device__ __forceinline__ void forward_batched(const double* __restrict__ rv, const double* __restrict__ zs, double* __restrict__ rec,
                                                int n_elim, double (&r)[SS + 1]) {
#pragma unroll
  for (int b = 0; b < SS; ++b) r[b] = rv[b];
  double nl_[B][SS + 1], nr_[B], nz_[B];
  auto fetch = [&](int i0) {
#pragma unroll
    for (int u = 0; u < B; ++u) {
#pragma unroll
      for (int d = 0; d <= SS; ++d) nl_[u][d] = rec[(i0 + u) * (SS + 2) + d];
      nr_[u] = rv[i0 + u + SS];
      nz_[u] = zs[i0 + u];
    }
  };
  fetch(0);
  int i = 0;
  for (; i + B <= n_elim; i += B) {
    double cl[B][SS + 1], cr[B], cz[B], wout[B];
#pragma unroll
    for (int u = 0; u < B; ++u) {
#pragma unroll
      for (int d = 0; d <= SS; ++d) cl[u][d] = nl_[u][d];
      cr[u] = nr_[u]; cz[u] = nz_[u];
    }
    fetch(i + B);            // (the records and right-hand sides are padded: reading past the chain is harmless)
#pragma unroll
    for (int u = 0; u < B; ++u) {
      r[SS] = cr[u];
      const double uu = r[0];
#pragma unroll
      for (int d = 1; d <= SS; ++d) r[d] = fma(-cl[u][d - 1], uu, r[d]);
      wout[u] = fma(uu, cl[u][SS], cz[u]);
#pragma unroll
      for (int b = 0; b < SS; ++b) r[b] = r[b + 1];
    }
#pragma unroll
    for (int u = 0; u < B; ++u) rec[(i + u) * (SS + 2) + SS + 1] = wout[u];
  }
  for (; i < n_elim; ++i) {
    r[SS] = rv[i + SS];
    const double uu = r[0];
#pragma unroll
    for (int d = 1; d <= SS; ++d) r[d] = fma(-rec[i * (SS + 2) + d - 1], uu, r[d]);
    rec[i * (SS + 2) + SS + 1] = fma(uu, rec[i * (SS + 2) + SS], zs[i]);
#pragma unroll
    for (int b = 0; b < SS; ++b) r[b] = r[b + 1];
  }
}

// ---- variant: back-substitution with the records of B pivots fetched ahead ----
template <int SS, int B>
__device__ __forceinline__ void backward_batched(double* __restrict__ rec, int n_elim, double (&x)[SS + 1]) {
  int i = n_elim - 1;
  double nl_[B][SS + 1];
  auto fetch = [&](int ihi) {
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const int ii = ihi - u;
#pragma unroll
      for (int d = 0; d < SS; ++d) nl_[u][d] = ii >= 0 ? rec[ii * (SS + 2) + d] : 0.0;
      nl_[u][SS] = ii >= 0 ? rec[ii * (SS + 2) + SS + 1] : 0.0;
    }
  };
  fetch(i);
  for (; i - B + 1 >= 0; i -= B) {
    double cl[B][SS + 1], xo[B];
#pragma unroll
    for (int u = 0; u < B; ++u)
#pragma unroll
      for (int d = 0; d <= SS; ++d) cl[u][d] = nl_[u][d];
    fetch(i - B);
#pragma unroll
    for (int u = 0; u < B; ++u) {
      double acc = cl[u][SS];
#pragma unroll
      for (int d = SS; d >= 1; --d) acc = fma(-cl[u][d - 1], x[d], acc);
#pragma unroll
      for (int d = SS; d >= 2; --d) x[d] = x[d - 1];
      x[1] = acc;
      xo[u] = acc;
    }
#pragma unroll
    for (int u = 0; u < B; ++u) rec[(i - u) * (SS + 2) + SS + 1] = xo[u];
  }
  for (; i >= 0; --i) {
    double acc = rec[i * (SS + 2) + SS + 1];
#pragma unroll
    for (int d = SS; d >= 1; --d) acc = fma(-rec[i * (SS + 2) + d - 1], x[d], acc);
#pragma unroll
    for (int d = SS; d >= 2; --d) x[d] = x[d - 1];
    x[1] = acc;
    rec[i * (SS + 2) + SS + 1] = acc;
  }
}


// ---- variants with 16-byte aligned records of 6 doubles [l1 l2 | l3 inv | u/w/x -]: two b128 + one b64 per pivot ----
constexpr int RA = 6;
__device__ __forceinline__ bool forward_aligned(const double* __restrict__ Pv, const double* __restrict__ rv, double* __restrict__ rec,
                                                int n_elim, int n_common, double gk, SpecWin<3>& w) {
  constexpr int SS = 3;
#pragma unroll
  for (int b = 0; b < SS; ++b) {
    const double2 p0 = *reinterpret_cast<const double2*>(Pv + b * 4), p1 = *reinterpret_cast<const double2*>(Pv + b * 4 + 2);
    w.c[b][0] = p0.x + gk; w.c[b][1] = p0.y; w.c[b][2] = p1.x; w.c[b][3] = p1.y;
    w.r[b] = rv[b];
  }
  bool bad = false;
  auto pivot = [&](int i) {
    const double2 p0 = *reinterpret_cast<const double2*>(Pv + (i + SS) * 4), p1 = *reinterpret_cast<const double2*>(Pv + (i + SS) * 4 + 2);
    w.c[SS][0] = p0.x + gk; w.c[SS][1] = p0.y; w.c[SS][2] = p1.x; w.c[SS][3] = p1.y;
    w.r[SS] = rv[i + SS];
    const double d0 = w.c[0][0];
    bad |= !(d0 > 0.0);
    const double inv = rcp_cubic(d0);
    const double u = w.r[0];
    double l[SS + 1];
#pragma unroll
    for (int d = 1; d <= SS; ++d) l[d] = w.c[0][d] * inv;
#pragma unroll
    for (int b = 1; b <= SS; ++b)
#pragma unroll
      for (int a = b; a <= SS; ++a) w.c[b][a - b] = fma(-l[a], w.c[0][b], w.c[b][a - b]);
#pragma unroll
    for (int d = 1; d <= SS; ++d) w.r[d] = fma(-l[d], u, w.r[d]);
    *reinterpret_cast<double2*>(rec + i * RA) = make_double2(l[1], l[2]);
    *reinterpret_cast<double2*>(rec + i * RA + 2) = make_double2(l[3], inv);
    rec[i * RA + 4] = u;
#pragma unroll
    for (int b = 0; b < SS; ++b) {
#pragma unroll
      for (int d = 0; d <= SS; ++d) w.c[b][d] = w.c[b + 1][d];
      w.r[b] = w.r[b + 1];
    }
  };
#pragma unroll 4
  for (int i = 0; i < n_common; ++i) pivot(i);
  if (n_elim > n_common) pivot(n_common);
  return !bad;
}
__device__ __forceinline__ void backward_aligned(double* __restrict__ rec, int n_elim, double (&x)[4]) {
#pragma unroll 4
  for (int i = n_elim - 1; i >= 0; --i) {
    const double2 a0 = *reinterpret_cast<const double2*>(rec + i * RA), a1 = *reinterpret_cast<const double2*>(rec + i * RA + 2);
    double acc = rec[i * RA + 4];
    acc = fma(-a1.x, x[3], acc);
    acc = fma(-a0.y, x[2], acc);
    acc = fma(-a0.x, x[1], acc);
    x[3] = x[2]; x[2] = x[1]; x[1] = acc;
    rec[i * RA + 4] = acc;
  }
}

__global__ __launch_bounds__(64) void probe(const double* band, const double* rhs, const double* g, const double* z, long long* cyc, double* out) {
  __shared__ double P[(T + 2 * S + 8) * D1], Pm[(T + 2 * S + 8) * D1];
  __shared__ double mt[K * (Tp + 8)], mtm[K * (Tp + 8)];
  __shared__ double rec[NV][(T + 8) * K * RS];
  __shared__ double zs[T * K + 16];
  __shared__ __attribute__((aligned(16))) double reca[(T + 8) * K * RA];
  const int lane = threadIdx.x;
  int nl, nr, ns;
  spectral_split(T, S, nl, nr, ns);
  for (int i = lane; i < (T + 2 * S + 8) * D1; i += 64) { P[i] = 0.0; Pm[i] = 0.0; }
  for (int i = lane; i < K * (Tp + 8); i += 64) { mt[i] = 0.0; mtm[i] = 0.0; }
  for (int v = 0; v < NV; ++v) for (int i = lane; i < (T + 8) * K * RS; i += 64) rec[v][i] = 0.0;
  __syncthreads();
  for (int i = lane; i < T * D1; i += 64) {
    const int t = i / D1, d = i - t * D1;
    P[i] = band[i];
    if (t + d < T) Pm[(T - 1 - t - d) * D1 + d] = band[i];
  }
  for (int i = lane; i < K * T; i += 64) {
    const int k = i / T, t = i - k * T;
    mt[k * (Tp + 8) + t] = rhs[i];
    mtm[k * (Tp + 8) + T - 1 - t] = rhs[i];
    zs[i] = z[i];
  }
  __syncthreads();
  const bool chain = lane < 2 * K;
  const int side = lane >= K ? 1 : 0, k = chain ? lane - side * K : 0;
  const double* Pv = side ? Pm : P;
  const double* rv = (side ? mtm : mt) + k * (Tp + 8);
  const int n_elim = side ? nr : nl, n_common = nl < nr ? nl : nr;
  const double gk = g[k];
  const int off = (k * (T + 8) + (side ? nl : 0)) * RS;
  const double* zc = zs + k * T + (side ? nl : 0);
  long long t0, t1;
  double sink = 0.0;
#define TIME(slot, ...)                                     \
  __builtin_amdgcn_s_waitcnt(0);                            \
  t0 = __builtin_amdgcn_s_memtime();                        \
  if (chain) { __VA_ARGS__ }                                \
  __builtin_amdgcn_s_waitcnt(0);                            \
  t1 = __builtin_amdgcn_s_memtime();                        \
  if (lane == 0) cyc[slot] = t1 - t0;
  // 0: the combined chain of the kernels (factor + rhs), then the w pass's arithmetic is NOT included
  TIME(0, { SpecWin<S> w; bool ok = spectral_forward<S>(Pv, rv, rec[0] + off, n_elim, n_common, gk, w); sink += w.r[0] + (ok ? 1.0 : 0.0); })
  // 1: factor only (as the dataflow tail runs it)
  TIME(1, { SpecWinC<S> w; bool ok = spectral_factor<S>(Pv, rec[1] + off, n_elim, n_common, gk, w); sink += w.c[0][0] + (ok ? 1.0 : 0.0); })
  // 2: forward substitution from records (as the dataflow tail runs it)
  TIME(2, { double r[S + 1]; spectral_forward_rhs<S>(rv, zc, rec[1] + off, n_elim, n_common, r); sink += r[0]; })
  // 3: back-substitution (the kernels')
  TIME(3, { double x[S + 1]; for (int q = 0; q <= S; ++q) x[q] = 0.0; spectral_backward<S>(rec[1] + off, n_elim, x); sink += x[1]; })
  // 4: factor, band rows fetched a batch ahead
  TIME(4, { SpecWinC<S> w; bool ok = factor_batched<S, PF>(Pv, rec[4] + off, n_elim, gk, w); sink += w.c[0][0] + (ok ? 1.0 : 0.0); })
  // 5: forward from records, a batch ahead
  TIME(5, { double r[S + 1]; forward_batched<S, PF>(rv, zc, rec[4] + off, n_elim, r); sink += r[0]; })
  // 6: backward, a batch ahead
  TIME(6, { double x[S + 1]; for (int q = 0; q <= S; ++q) x[q] = 0.0; backward_batched<S, PF>(rec[4] + off, n_elim, x); sink += x[1]; })
  // 7 / 8: the combined chain and the back-substitution on aligned records
  const int offa = (k * (T + 8) + (side ? nl : 0)) * RA;
  TIME(7, { SpecWin<S> w; bool ok = forward_aligned(Pv, rv, reca + offa, n_elim, n_common, gk, w); sink += w.r[0] + (ok ? 1.0 : 0.0); })
  TIME(8, { double x[S + 1]; for (int q = 0; q <= S; ++q) x[q] = 0.0; backward_aligned(reca + offa, n_elim, x); sink += x[1]; })
  __syncthreads();
  if (lane == 0) out[0] = sink;
  // records of variant 1 (after its forward / backward) and 4 must agree bit for bit
  double diff = 0.0;
  for (int i = lane; i < (T + 8) * K * RS; i += 64) diff = fmax(diff, fabs(rec[1][i] - rec[4][i]));
  for (int o = 32; o > 0; o >>= 1) diff = fmax(diff, __shfl_xor(diff, o));
  if (lane == 0) out[1] = diff;
  // ... and the aligned records against the combined chain's (rec[0]: factor entries and u, before any w / x pass) - compare l, inv
  double diff2 = 0.0;
  for (int c = lane; c < 2 * K; c += 64) {
    const int sd = c >= K ? 1 : 0, kk = c - sd * K, ne = sd ? nr : nl;
    for (int i = 0; i < ne; ++i)
      for (int d = 0; d < 4; ++d)
        diff2 = fmax(diff2, fabs(rec[0][(kk * (T + 8) + (sd ? nl : 0) + i) * RS + d] - reca[(kk * (T + 8) + (sd ? nl : 0) + i) * RA + d]));
  }
  for (int o = 32; o > 0; o >>= 1) diff2 = fmax(diff2, __shfl_xor(diff2, o));
  if (lane == 0) out[2] = diff2;
}

int main() {
  std::vector<double> band(T * D1, 0.0), rhs(K * T), g(K), z(K * T);
  srand(1);
  // a third-difference penalty band, scaled by random positive weights: SPD with the shifts g_k
  std::vector<double> A((size_t)T * T, 0.0);
  for (int r = 0; r + 3 < T; ++r) {
    const double cw = 0.5 + (rand() % 1000) / 500.0;
    const double c[4] = {1, -3, 3, -1};
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) A[(size_t)(r + a) * T + r + b] += cw * c[a] * c[b];
  }
  for (int t = 0; t < T; ++t) for (int d = 0; d <= S && t + d < T; ++d) band[t * D1 + d] = A[(size_t)(t + d) * T + t];
  for (int k = 0; k < K; ++k) g[k] = 0.5 + k;
  for (int i = 0; i < K * T; ++i) { rhs[i] = (rand() % 2000) / 1000.0 - 1.0; z[i] = (rand() % 2000) / 1000.0 - 1.0; }
  double *db, *dr, *dg, *dz, *dout;
  long long* dc;
  CK(hipMalloc(&db, band.size() * 8)); CK(hipMalloc(&dr, rhs.size() * 8)); CK(hipMalloc(&dg, g.size() * 8)); CK(hipMalloc(&dz, z.size() * 8));
  CK(hipMalloc(&dc, 16 * 8)); CK(hipMalloc(&dout, 16 * 8));
  CK(hipMemcpy(db, band.data(), band.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dr, rhs.data(), rhs.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, g.data(), g.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dz, z.data(), z.size() * 8, hipMemcpyHostToDevice));
  long long best[16];
  for (int i = 0; i < 16; ++i) best[i] = 1LL << 60;
  double out[16] = {0};
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(dc, 0, 16 * 8));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, db, dr, dg, dz, dc, dout);
    CK(hipDeviceSynchronize());
    long long c[16];
    CK(hipMemcpy(c, dc, 16 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out, dout, 16 * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i) if (c[i] > 0 && c[i] < best[i]) best[i] = c[i];
  }
  const char* names[] = {"combined forward (kernels)", "factor only", "forward from records", "backward (kernels)",
                         "factor, batch ahead", "forward, batch ahead", "backward, batch ahead", "combined, aligned records", "backward, aligned records"};
  printf("PF = %d; s_memtime ticks (best of 5), 31 / 30 pivots per chain\n", PF);
  for (int i = 0; i < 9; ++i) printf("  %-28s %8lld\n", names[i], best[i]);
  printf("  records of the batched variants vs the plain ones: max |diff| = %g (must be 0); aligned vs combined factor entries: %g (must be 0); sink %g\n", out[1], out[2], out[0]);
  return 0;
}
