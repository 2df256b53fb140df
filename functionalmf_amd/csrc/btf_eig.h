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
// Warm start: between two Gibbs sweeps the Gram moves little, so the previous eigenvectors (left in `out`) are
// refined instead of recomputed (Ogita-Aishima iteration, below): two or three iterations of four K^3 products.
// First call, or no convergence: cold cyclic Jacobi.  The result is the eigen-system of G to rounding either way.
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

// 1/d to full precision from v_rcp_f64 (rel. error < 2^-23) and a cubic correction (btf_spectral.h's rcp_cubic): four
// instructions where an IEEE division is ~30 - the refinement below divides once per matrix element and iteration
__device__ __forceinline__ double eig_rcp(double d) {
  const double r = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, r, 1.0);
  const double p = fma(e, e, e);
  return fma(p, r, r);
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// k = 0 .. K-1: unrolled when K is a compile-time constant (KC > 0), the plain loop otherwise
template <int KC, class F>
__device__ __forceinline__ void eig_for(int K, F f) {
  if constexpr (KC > 0) {
#pragma unroll
    for (int k = 0; k < KC; ++k) f(k);
  } else {
    for (int k = 0; k < K; ++k) f(k);
  }
}
// gsrc: `ngp` packed-lower partial Grams [ngp][K(K+1)/2] summed in block order.  lane = threadIdx.x & 63 of the
// calling wave (all 64 lanes must call).
// KC > 0: K known at compile time (the callers inside kernels templated on it) - the K-long inner products unroll and
// their LDS reads go out together; a lone wave otherwise pays one LDS round trip per term (K = 10, warm path: ~20 us).
// pub != nullptr: the eigenvalues and vectors (K + K K doubles) are also stored write-through (sc1) there, for consumers
// inside the same launch (btf_fused.h); the caller drains the stores and raises the flag
// What a caller may fetch AHEAD of the call (beside its own loads of the Gram partials: one global round trip for both
// instead of two dependent ones in front of the first iteration): the warm-start counter and this lane's (up to two)
// entries of the previous eigenvectors.
struct EigWarm { double count, x0, x1; bool have; };
template <int KC>
__device__ __forceinline__ EigWarm eig_warm_fetch(const double* __restrict__ out, int Krt) {
  const int K = KC > 0 ? KC : Krt, K2 = K * K, lane = threadIdx.x & 63;
  EigWarm w;
  w.count = out[K + K2 + 1];
  w.x0 = lane < K2 ? out[K + lane] : 0.0;
  w.x1 = lane + WAVE < K2 ? out[K + lane + WAVE] : 0.0;
  w.have = true;
  return w;
}
// gran / epoch: the eigenvalues also go out as self-validating 8-byte granules {half of lambda, epoch} - FIRST, straight
// from the registers that hold them: a consumer that needs nothing but the eigenvalues (the chain waves of the dataflow
// tails, btf_fused.h) has them before the eigenvectors' stores have even been issued
// EXT: the prefetched warm start and the granules are compiled in (the side task of the fused V launch at nembeds <= 8); the
// plain form keeps the register footprint the run-time-K instances (nembeds 9, 10 on 16 waves) were tuned to
template <int KC = 0, bool EXT = false>
__device__ inline void gram_eig_wave(const double* __restrict__ gsrc, int ngp, int Krt, double* __restrict__ out,
                                     double* __restrict__ scratch, bool warm_ok = true, double* pub = nullptr,
                                     EigWarm pre = EigWarm{0.0, 0.0, 0.0, false}, unsigned long long* gran = nullptr, unsigned epoch = 0u) {
  const int K = KC > 0 ? KC : Krt;
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
  // ---- warm path: refine the previous eigenvectors (Ogita & Aishima 2018, "Iterative refinement for symmetric
  //      eigenvalue decomposition"): with X ~ eigenvectors,  R = I - X'X,  S = X'GX,  lambda_i = s_ii / (1 - r_ii),
  //      e_ij = (s_ij + lambda_j r_ij) / (lambda_j - lambda_i)  (r_ij / 2 on the diagonal and inside a cluster),
  //      X <- X + X E.  Four K^3 products per iteration, quadratic convergence, and the R term pulls X back to
  //      orthogonality, so nothing drifts from call to call.  Between two Gibbs sweeps the Gram moves by ~1e-3, so
  //      two or three iterations reach 1e-15.  Anything else (first call, no convergence) takes the Jacobi path.
  const double wcount = (EXT && pre.have) ? pre.count : out[K + K2 + 1];
  const bool warm = warm_ok && wcount >= 1.0 && K > 1;
  double* lamv = reinterpret_cast<double*>(csg);                   // K eigenvalue estimates (csg is idle here)
  bool refined = false;
  int xcur = 0;
  if (warm) {
    if (h0) Ub0[e0] = (EXT && pre.have) ? pre.x0 : out[K + e0];
    if (h1) Ub0[e1] = (EXT && pre.have) ? pre.x1 : out[K + e1];
    wave_lds_sync();
    float prev_err = 3.0e38f;
    for (int it = 0; it < 8; ++it) {
      const double* X = xcur ? Ub1 : Ub0;
      double* Xn = xcur ? Ub0 : Ub1;
      // T = G X (-> Ab1), R = I - X'X (registers)
      double R0 = 0.0, R1 = 0.0;
      {
        double t0 = 0.0, t1 = 0.0, q0 = 0.0, q1 = 0.0;
        if (h0) {
          eig_for<KC>(K, [&](int k) { const double xc = X[k * K + c0]; t0 = fma(Ab0[r0 * K + k], xc, t0); q0 = fma(X[k * K + r0], xc, q0); });
        }
        if (h1) {
          eig_for<KC>(K, [&](int k) { const double xc = X[k * K + c1]; t1 = fma(Ab0[r1 * K + k], xc, t1); q1 = fma(X[k * K + r1], xc, q1); });
        }
        if (h0) { Ab1[e0] = t0; R0 = (r0 == c0 ? 1.0 : 0.0) - q0; }
        if (h1) { Ab1[e1] = t1; R1 = (r1 == c1 ? 1.0 : 0.0) - q1; }
      }
      wave_lds_sync();
      // S = X'T (registers); eigenvalue estimates from the diagonal
      double S0 = 0.0, S1 = 0.0;
      if (h0) {
        eig_for<KC>(K, [&](int k) { S0 = fma(X[k * K + r0], Ab1[k * K + c0], S0); });
      }
      if (h1) {
        eig_for<KC>(K, [&](int k) { S1 = fma(X[k * K + r1], Ab1[k * K + c1], S1); });
      }
      if (h0 && r0 == c0) lamv[r0] = S0 * eig_rcp(1.0 - R0);
      if (h1 && r1 == c1) lamv[r1] = S1 * eig_rcp(1.0 - R1);
      wave_lds_sync();
      double lmax = 0.0;
      eig_for<KC>(K, [&](int k) { lmax = fmax(lmax, fabs(lamv[k])); });
      // error of this iterate: off-diagonal of S and all of R, relative to the largest eigenvalue
      float err = 0.0f;
      if (h0) err = fmaxf(err, (float)fmax(r0 == c0 ? 0.0 : fabs(S0), lmax * fabs(R0)));
      if (h1) err = fmaxf(err, (float)fmax(r1 == c1 ? 0.0 : fabs(S1), lmax * fabs(R1)));
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) err = fmaxf(err, __shfl_xor(err, off, WAVE));
      const float rel = err / (float)lmax;
      if (rel <= 2e-15f) { refined = true; break; }
      if (!(rel < 0.05f) || !(err < 0.5f * prev_err) || it == 7) break;      // too far, or not contracting: Jacobi
      prev_err = err;
      const bool last = rel <= 1e-8f;      // quadratic convergence: this update lands at ~rel^2, no further check needed
      const double delta = 2.0 * K * (double)err;                            // cluster threshold (bounds ||S - D|| + ||G|| ||R||)
      if (h0) {
        const double gap = lamv[c0] - lamv[r0];
        Ab1[e0] = (r0 == c0 || !(fabs(gap) > delta)) ? 0.5 * R0 : (S0 + lamv[c0] * R0) * eig_rcp(gap);
      }
      if (h1) {
        const double gap = lamv[c1] - lamv[r1];
        Ab1[e1] = (r1 == c1 || !(fabs(gap) > delta)) ? 0.5 * R1 : (S1 + lamv[c1] * R1) * eig_rcp(gap);
      }
      wave_lds_sync();
      // X <- X + X E
      if (h0) {
        double x = X[e0];
        eig_for<KC>(K, [&](int k) { x = fma(X[r0 * K + k], Ab1[k * K + c0], x); });
        Xn[e0] = x;
      }
      if (h1) {
        double x = X[e1];
        eig_for<KC>(K, [&](int k) { x = fma(X[r1 * K + k], Ab1[k * K + c1], x); });
        Xn[e1] = x;
      }
      xcur ^= 1;
      wave_lds_sync();
      if (last) { refined = true; break; }
    }
  }
  if (!refined) {
    if (h0) Ub0[e0] = r0 == c0 ? 1.0 : 0.0;
    if (h1) Ub0[e1] = r1 == c1 ? 1.0 : 0.0;
    wave_lds_sync();
  }
  const int Ke = K + (K & 1);            // players of the round-robin tournament (a phantom one if K is odd)
  const int Km = Ke - 1;
  // circle method (player Km fixed, the others rotate): partner of every index in every round, once (Jacobi path only)
  for (int e = lane; !refined && e < Km * K; e += WAVE) {
    const int round = e / K, i = e - round * K;
    int x = 2 * round - i;
    x += x < 0 ? Km : 0;
    x -= x >= Km ? Km : 0;
    const int pt = i == Km ? round : (i == round ? Km : x);
    ptab[e] = pt < K ? pt : i;           // paired with the phantom: sits the round out
  }
  wave_lds_sync();
  int cur = 0, sweeps = 0;
  for (; !refined && sweeps < 20 && K > 1; ++sweeps) {
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
  const double* U = refined ? (xcur ? Ub1 : Ub0) : (cur ? Ub1 : Ub0);
  if (!refined) {
    if (lane < K) lamv[lane] = A[lane * K + lane];
    wave_lds_sync();
  }
  if (lane < K) {
    const double lam = lamv[lane];
    int rank = 0;
    eig_for<KC>(K, [&](int i) {
      const double li = lamv[i];
      if (li < lam || (li == lam && i < lane)) ++rank;
    });
    int bigr = 0;
    double bv = 0.0;
    eig_for<KC>(K, [&](int r) {
      const double v = fabs(U[r * K + lane]);
      if (v > bv) { bv = v; bigr = r; }
    });
    const double sgn = U[bigr * K + lane] < 0.0 ? -1.0 : 1.0;
    if (EXT && gran) {
      const unsigned long long gb = (unsigned long long)__double_as_longlong(lam);
      __hip_atomic_store(gran + 2 * rank, ((gb >> 32) << 32) | (unsigned long long)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gran + 2 * rank + 1, (gb << 32) | (unsigned long long)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[rank] = lam;
    if (pub) store_sc1(pub + rank, lam);
    for (int r = 0; r < K; ++r) {
      const double v = sgn * U[r * K + lane];
      out[K + r * K + rank] = v;
      if (pub) store_sc1(pub + K + r * K + rank, v);
    }
  }
  if (lane == 0) {
    out[K + K * K] = (double)sweeps;           // Jacobi sweeps (0: the refinement path converged)
    out[K + K * K + 1] = 1.0;                  // `out` now holds a solution to start the next call from
  }
  __builtin_amdgcn_s_setprio(0);
}

// side task of a streaming launch (accum_kernel): out == nullptr: none
// (Usrc != nullptr: no partials exist - the side workgroup forms the Gram of Usrc's nrows rows itself)
// (pub / flag / epoch: the fused V launch's tails read the eigen-system inside the launch - btf_fused.h)
// (gran: the eigenvalues once more as self-validating 8-byte granules {half of g_k, epoch}, two per eigenvalue - the chain
//  waves of the dataflow tails get value and "it is this launch's" in ONE round trip instead of flag, then payload)
struct EigSide { const double* gpart; int ngp; int K; double* out; const double* Usrc; int nrows; double* pub; unsigned* flag; unsigned epoch;
                 unsigned long long* gran; };

static __global__ __launch_bounds__(WAVE) void gram_eig_kernel(const double* __restrict__ gpart, int ngp, int K,
                                                        double* __restrict__ out, int warm) {
  __shared__ __attribute__((aligned(16))) double scratch[EIG_LDS_DOUBLES];
  gram_eig_wave(gpart, ngp, K, out, scratch, warm != 0);
}

}  // namespace btf
