// Eigen-decomposition of a K x K symmetric (Gram) matrix by ONE wave (K <= 10): cyclic Jacobi with a
// round-robin parallel ordering (all disjoint pairs of a round rotate together).  Used by the spectral V
// sampler (btf_spectral.h) for the shared likelihood block W'W; runs as a side task inside the V half-sweep's
// streaming accumulation launch (accum_kernel's `side` argument), so it costs neither a launch nor time.
//
// out[0..K-1]      eigenvalues, ascending
// out[K + r*K + c] component r of the c-th eigenvector, each vector's largest-magnitude entry positive
// out[K + K*K]     number of sweeps used (diagnostic)
// out[K + K*K + 1] warm-start counter (see below); 0 = `out` holds no previous solution
//
// Warm start: between two Gibbs sweeps the Gram moves little, so the previous eigenvectors nearly diagonalise
// the new matrix: the iteration starts from A0 = U_prev' G U_prev, U0 = U_prev (two K^3 products) and needs two
// or three sweeps instead of six.  Rounding lets U0 drift from orthogonality by ~1e-16 per call, so every 16th
// call starts cold from U0 = I.  The result is the eigen-system of G to rounding either way.
//
// A lone wave issues one f64 instruction per ~8 cycles, so a round is written for instruction count: every
// index computes its own rotation (no serial section), the angle comes from the hardware rsq / rcp
// approximations (it only steers the convergence) while the cosine that keeps the rotation orthogonal is
// refined to full precision, a round is two LDS phases (parameters; J'AJ and UJ applied at once into the other
// buffer) separated by wave-level fences (no s_barrier: the rest of the workgroup may be doing something else).
// Convergence is quadratic: a sweep that met no pair with |a_pq| > 1e-7 sqrt(a_pp a_qq) is the last one.
#pragma once
#include "btf_device.h"

namespace btf {

constexpr int EIG_MAXK = 10;
constexpr int EIG_LDS_DOUBLES = 5 * EIG_MAXK * EIG_MAXK + 3 * EIG_MAXK + 8;   // scratch the caller provides

__device__ __forceinline__ double rsq_nr(double x) {       // 1/sqrt(x) to full precision
  double r = __builtin_amdgcn_rsq(x);
  double e = fma(-x * r, r, 1.0);
  r = fma(0.5 * e, r, r);
  e = fma(-x * r, r, 1.0);
  return fma(0.5 * e, r, r);
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// gsrc: `ngp` packed-lower partial Grams [ngp][K(K+1)/2] summed in block order.  lane = threadIdx.x & 63 of the
// calling wave (all 64 lanes must call).
__device__ inline void gram_eig_wave(const double* __restrict__ gsrc, int ngp, int K, double* __restrict__ out,
                                     double* __restrict__ scratch, bool warm_ok = true) {
  __builtin_amdgcn_s_setprio(3);       // beside a streaming kernel: win the issue arbitration, the stream waits on memory anyway
  const int lane = threadIdx.x & 63;
  const int KK = tri(K), K2 = K * K;
  double* Ab0 = scratch;
  double* Ab1 = scratch + EIG_MAXK * EIG_MAXK;
  double* Ub0 = scratch + 2 * EIG_MAXK * EIG_MAXK;
  double* Ub1 = scratch + 3 * EIG_MAXK * EIG_MAXK;
  double2* csg = reinterpret_cast<double2*>(scratch + 4 * EIG_MAXK * EIG_MAXK);      // (cos, signed sin) per index
  int* ptab = reinterpret_cast<int*>(scratch + 4 * EIG_MAXK * EIG_MAXK + 2 * EIG_MAXK + 2);   // partner[round][index]
  if (lane < KK) {
    double s = 0.0;
    int b = 0;
    for (; b + 4 <= ngp; b += 4) {
      const double x0 = gsrc[(size_t)b * KK + lane], x1 = gsrc[(size_t)(b + 1) * KK + lane];
      const double x2 = gsrc[(size_t)(b + 2) * KK + lane], x3 = gsrc[(size_t)(b + 3) * KK + lane];
      s += x0; s += x1; s += x2; s += x3;
    }
    for (; b < ngp; ++b) s += gsrc[(size_t)b * KK + lane];
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= lane) ++r;
    const int c = lane - r * (r + 1) / 2;
    Ab0[r * K + c] = s;
    Ab0[c * K + r] = s;
  }
  // this lane's (up to two) matrix elements
  const int e0 = lane, e1 = lane + WAVE;
  const bool h0 = e0 < K2, h1 = e1 < K2;
  const int r0 = h0 ? e0 / K : 0, c0 = h0 ? e0 - r0 * K : 0;
  const int r1 = h1 ? e1 / K : 0, c1 = h1 ? e1 - r1 * K : 0;
  const double wcount = out[K + K2 + 1];
  const bool warm = warm_ok && wcount >= 1.0 && wcount < 16.0;
  if (h0) Ub0[e0] = warm ? out[K + e0] : (r0 == c0 ? 1.0 : 0.0);
  if (h1) Ub0[e1] = warm ? out[K + e1] : (r1 == c1 ? 1.0 : 0.0);
  wave_lds_sync();
  if (warm) {                           // A0 = U0' G U0 through Ab1 = G U0
    for (int h = 0; h < 2; ++h) {
      const bool on = h ? h1 : h0;
      const int e = h ? e1 : e0, r = h ? r1 : r0, c = h ? c1 : c0;
      if (on) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s = fma(Ab0[r * K + k], Ub0[k * K + c], s);
        Ab1[e] = s;
      }
    }
    wave_lds_sync();
    double t0 = 0.0, t1 = 0.0;
    if (h0) for (int k = 0; k < K; ++k) t0 = fma(Ub0[k * K + r0], Ab1[k * K + c0], t0);
    if (h1) for (int k = 0; k < K; ++k) t1 = fma(Ub0[k * K + r1], Ab1[k * K + c1], t1);
    wave_lds_sync();
    if (h0) Ab0[e0] = t0;
    if (h1) Ab0[e1] = t1;
    wave_lds_sync();
    // (A0 is symmetric up to rounding; the rotations read its upper triangle for a pair (p < q) and both triangles
    //  in the update, exactly as for a cold start)
  }
  const int Ke = K + (K & 1);            // players of the round-robin tournament (a phantom one if K is odd)
  const int Km = Ke - 1;
  // circle method (player Km fixed, the others rotate): partner of every index in every round, once
  for (int e = lane; e < Km * K; e += WAVE) {
    const int round = e / K, i = e - round * K;
    int x = 2 * round - i;
    x += x < 0 ? Km : 0;
    x -= x >= Km ? Km : 0;
    const int pt = i == Km ? round : (i == round ? Km : x);
    ptab[e] = pt < K ? pt : i;           // paired with the phantom: sits the round out
  }
  wave_lds_sync();
  int cur = 0, sweeps = 0;
  for (; sweeps < 20 && K > 1; ++sweeps) {
    bool big = false;
    for (int round = 0; round < Km; ++round) {
      const double* A = cur ? Ab1 : Ab0;
      const double* U = cur ? Ub1 : Ub0;
      double* An = cur ? Ab0 : Ab1;
      double* Un = cur ? Ub0 : Ub1;
      // partners of this round (independent of phase 1: in flight under it)
      const int* pt_r = ptab + round * K;
      const int pr0 = pt_r[r0], pc0 = pt_r[c0];
      const int pr1 = h1 ? pt_r[r1] : 0, pc1 = h1 ? pt_r[c1] : 0;
      // ---- phase 1: index i = lane finds the rotation of its pair
      bool isbig = false;
      if (lane < K) {
        const int i = lane, pt = pt_r[i];
        double c = 1.0, sgn = 0.0;
        if (pt != i) {
          const int p = i < pt ? i : pt, q = i < pt ? pt : i;
          const double app = A[p * K + p], aqq = A[q * K + q], apq = A[p * K + q];
          const double a2 = apq * apq, dd = fabs(app * aqq);
          isbig = a2 > 1e-14 * dd;
          if (a2 > 1e-34 * dd) {
            // NR convention: t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (aqq - app) / (2 apq)
            //              = apq / (delta + sgn(delta) hypot(delta, apq)),    delta = (aqq - app) / 2
            const double delta = 0.5 * (aqq - app);
            const double h2 = fma(delta, delta, a2);
            const double h = h2 * __builtin_amdgcn_rsq(h2);
            const double t = apq * __builtin_amdgcn_rcp(delta >= 0.0 ? delta + h : delta - h);
            c = rsq_nr(fma(t, t, 1.0));
            const double sn = t * c;
            sgn = i == p ? -sn : sn;
          }
        }
        csg[i] = make_double2(c, sgn);
      }
      big |= __ballot(isbig) != 0ULL;
      wave_lds_sync();
      // ---- phase 2: A' = J' A J, U' = U J into the other buffer
      //      (X J)[r][i] = c_i X[r][i] + sg_i X[r][partner_i];   (J' X)[i][c] likewise on rows
      if (h0) {
        const int pr = pr0, pc = pc0;
        const double2 gr = csg[r0], gc = csg[c0];
        const double cr = gr.x, sr = gr.y, cc = gc.x, sc = gc.y;
        const double a00 = A[r0 * K + c0], a01 = A[r0 * K + pc], a10 = A[pr * K + c0], a11 = A[pr * K + pc];
        const double u0 = U[r0 * K + c0], u1 = U[r0 * K + pc];
        const double top = fma(cc, a00, sc * a01), bot = fma(cc, a10, sc * a11);
        An[e0] = fma(cr, top, sr * bot);
        Un[e0] = fma(cc, u0, sc * u1);
      }
      if (h1) {
        const int pr = pr1, pc = pc1;
        const double2 gr = csg[r1], gc = csg[c1];
        const double cr = gr.x, sr = gr.y, cc = gc.x, sc = gc.y;
        const double a00 = A[r1 * K + c1], a01 = A[r1 * K + pc], a10 = A[pr * K + c1], a11 = A[pr * K + pc];
        const double u0 = U[r1 * K + c1], u1 = U[r1 * K + pc];
        const double top = fma(cc, a00, sc * a01), bot = fma(cc, a10, sc * a11);
        An[e1] = fma(cr, top, sr * bot);
        Un[e1] = fma(cc, u0, sc * u1);
      }
      cur ^= 1;
      wave_lds_sync();
    }
    if (!big) { ++sweeps; break; }
  }
  // ---- sort ascending, fix the signs, write
  const double* A = cur ? Ab1 : Ab0;
  const double* U = cur ? Ub1 : Ub0;
  if (lane < K) {
    const double lam = A[lane * K + lane];
    int rank = 0;
    for (int i = 0; i < K; ++i) {
      const double li = A[i * K + i];
      if (li < lam || (li == lam && i < lane)) ++rank;
    }
    int bigr = 0;
    double bv = 0.0;
    for (int r = 0; r < K; ++r) {
      const double v = fabs(U[r * K + lane]);
      if (v > bv) { bv = v; bigr = r; }
    }
    const double sgn = U[bigr * K + lane] < 0.0 ? -1.0 : 1.0;
    out[rank] = lam;
    for (int r = 0; r < K; ++r) out[K + r * K + rank] = sgn * U[r * K + lane];
  }
  if (lane == 0) {
    out[K + K * K] = (double)sweeps;
    out[K + K * K + 1] = warm ? wcount + 1.0 : 1.0;
  }
  __builtin_amdgcn_s_setprio(0);
}

// side task of a streaming launch (accum_kernel): out == nullptr: none
struct EigSide { const double* gpart; int ngp; int K; double* out; };

__global__ __launch_bounds__(WAVE) void gram_eig_kernel(const double* __restrict__ gpart, int ngp, int K,
                                                        double* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) double scratch[EIG_LDS_DOUBLES];
  gram_eig_wave(gpart, ngp, K, out, scratch, false);
}

}  // namespace btf
