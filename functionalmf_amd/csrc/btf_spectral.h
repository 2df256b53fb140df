// Spectral V half-sweep sampler for complete Gaussian data (BTF_K_V_BANDED, sampler "spectral").
//
// With complete data and a scalar noise variance the likelihood block of every depth and every
// column is the same K x K matrix G = (R/nu2) W'W (factor.py:396-398 with constant weights), so in
// the reference's own k-major ordering x[k*T + t] = V[j,t,k] (factor.py:409) the conditional precision
// of column j is a Kronecker sum
//     Q_j = G (x) I_T  +  I_K (x) P_j ,        P_j = Delta' diag(1/(lam2 Tau2_j)) Delta   (factor.py:404-405)
// With G = U diag(g) U' (one K x K symmetric eigenproblem per half-sweep, shared by all columns:
// btf_eig.h) the rotation (U' (x) I_T) block-diagonalises Q_j into K independent T x T banded
// systems  A_k = g_k I + P_j  of half-bandwidth tf+1 - 64 pivots on 4 band entries per system instead of
// one 320-pivot factorisation on 16 band entries per column at (T=64, K=5, tf=2).
//
// The draw is  x = Q^-1 mu + S z  with the square root  S = (U (x) Pi') blockdiag_k(L_k^-T D_k^-1/2),
// Pi A_k Pi' = L_k D_k L_k' in the twisted depth order Pi of spectral_split below,  S S' = Q^-1:  the same
// distribution and the same mean term as fast_mvn.py:35-47 (which draws  P' L^-T z  for CHOLMOD's ordering
// P - DESIGN.md, "parity unpinned").  z[j][k*T + i] multiplies pivot i of system k.
// The jitter schedule of fast_mvn.py:62-68 shifts the diagonal of Q, i.e. every g_k, by eps.
//
// One 4-wave workgroup per column; each system is eliminated from both ends at once (2K chains in the lanes
// of wave 0, T/2 pivots each, meeting in an S x S separator system); the other waves load, rotate and draw
// the normals.  The prior band is built
// in the kernel from Tau2 (same arithmetic as prior_band_kernel), so no separate launch is needed.
#pragma once
#include "btf_kernels.h"

namespace btf {

constexpr int VS_THREADS = 256;
constexpr int VS_MAXE = 16;      // penalty rows that touch both t and t+d: at most 1+2+3+4+5 (tf_order 3)

// 1/d to full precision: v_rcp_f64 (rel. error < 2^-23) + one cubic correction r (1 + e + e^2), e = 1 - d r:
// three dependent fmas instead of the four of two Newton steps; this is the pivot chain's critical path.
__device__ __forceinline__ double rcp_cubic(double d) {
  const double r = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, r, 1.0);
  const double p = fma(e, e, e);
  return fma(p, r, r);
}

// ----------------------------------------------------------------------------------------------------
struct VSpecArgs {
  const double* part; int nch; int ld;   // accum partials [nch][K][ld], column j at offset j*T
  const double* eig;                     // gram_eig_kernel output for the unscaled Gram W'W
  double s, sR;                          // 1/nu2, nreps/nu2
  const double* Tau2; double lam2; int nD;
  const int* st_ptr; const int* st_row; const double* st_coef;   // Delta' . Delta stencil per (t,d), CSR
  const int* st_drow; const double* st_dcoef;                    // the same, VS_MAXE slots per (t,d) (one round trip)
  int T, TF, K;
  int col0, ml;
  double* V;
  const double* z; unsigned long long seed; unsigned long long stream;
  double eps0; int attempts;
  int* status; int* tries;
  double* gout;                          // [ml][KK] V_j'V_j of the fresh column, or nullptr
  const double* hyp; double Rrep; int hyp_noise;
  long long* dbg;
  // curve-structured replicate counts (btf_kernels.h, CurveLists): a column with deficient rows solves the eigen-problem
  // of ITS Gram  W'W - sum_i (1 - c_ij / R) w_i w_i'  here (warm-started from its own previous solution in eig_cols)
  CurveLists cv; const double* cv_W; const double* gpart; int ngp; double* eig_cols;
  int eig_cols_ready;                    // 1: the accumulation launch's side tasks already solved the curve columns
  // [ml] this column's part of the residual sum of squares with the W that stands and the V just drawn:
  // sum_t (R v_t' W'W v_t - 2 v_t . m_t), m the raw mean part of the partials - what nu2 | rest of the NEXT sweep needs
  // (besides the constants of the data), so that sweep spends no launch on it; nullptr: not wanted (complete data only)
  double* sse_out;
  double* rec_g;                         // [ml][T K (S+2)] HBM scratch for the pivot records (v_spectral_kernel<S, true>), else nullptr
  const double* pband;                   // [ml][T][S+1] prior band formed by prior_band_kernel (long depth axes), or nullptr: the kernel forms it
};

// Elimination order inside one system (the order the build declares for this sampler; z[j][k*T + i]
// multiplies pivot i of system k): "burn at both ends" - a separator of S = tf+1 depths [ts, ts+S),
// ts = (T-S)/2, splits the depths into two halves that do not touch; pivots 0..ts-1 are the depths 0..ts-1
// ascending, pivots ts.. the depths T-1..ts+S descending, the last S pivots the separator ascending.
// T < 2S+2: no split, natural order.
__host__ __device__ inline void spectral_split(int T, int S, int& nl, int& nr, int& ns) {
  if (T >= 2 * S + 2) { nl = (T - S) / 2; ns = S; nr = T - nl - S; }
  else { nl = T; nr = 0; ns = 0; }
}
// pivot i of a system -> depth
__host__ __device__ inline int spectral_depth_of_pivot(int i, int T, int S) {
  int nl, nr, ns;
  spectral_split(T, S, nl, nr, ns);
  if (i < nl) return i;
  if (i < nl + nr) return T - 1 - (i - nl);
  return nl + (i - nl - nr);
}

struct VsLayout {      // LDS offsets in doubles
  int U, g, itau, P, Pm, mraw, mt, mtm, zz, rec, win, gs, flag, eG, eo, esc, total;
  int Tp, RS;
};
// rec_global: the pivot records (n (S+2) doubles, the largest piece) live in HBM scratch instead - long depth axes
// (the reference's flu data: T = 370, K = 10: 148 KB of records alone) keep the spectral split that way
__host__ __device__ inline VsLayout vs_layout(int T, int K, int TF, int nD, bool rec_global = false) {
  VsLayout L;
  const int S = TF + 1, n = T * K;
  L.Tp = T + S + 1;
  L.RS = S + 2;                    // record of a pivot: S factor entries, 1/D, u -> w -> x
  int o = 0;
  L.U = o; o += K * K;
  L.g = o; o += K;
  o = (o + 1) & ~1;
  L.itau = o; o += nD;
  L.P = o; o += (T + S + 1) * (S + 1);     // prior band, P[t][d] = entry (t+d, t); zero rows behind
  L.Pm = o; o += (T + S + 1) * (S + 1);    // the same matrix seen from the far end: Pm[m][d] = entry (T-1-m, T-1-m-d)
  L.mraw = o; o += n;              // raw sums, depth-major; later the output staging
  L.mt = o; o += K * L.Tp;         // rotated right-hand sides, k-major, zero padded
  L.mtm = o; o += K * L.Tp;        // ... mirrored
  L.zz = o; o += n;
  L.rec = o; if (!rec_global) o += n * L.RS;
  L.win = o; o += 2 * K * (S * (S + 1) + S);   // the two chains' windows at the separator
  L.gs = o; o += VS_THREADS;       // Gram-share scratch: at most 16 groups of KK <= 256 doubles
  L.flag = o; o += 8;
  L.eG = o; o += (tri(K) + 1) & ~1;              // a curve column's own Gram
  L.eo = o; o += (K + K * K + 8 + 1) & ~1;       // ... its eigen-system (gram_eig_wave's output record)
  L.esc = o; o += EIG_LDS_DOUBLES;               // ... the solver's scratch
  L.total = o;
  return L;
}
__host__ __device__ inline size_t vs_lds_bytes(int T, int K, int TF, int nD, bool rec_global = false) {
  return (size_t)vs_layout(T, K, TF, nD, rec_global).total * sizeof(double);
}

// One elimination chain: LDL' of  A = g I + P  (half-bandwidth S) from one end, the forward substitution of
// rhs folded in, stopping after n_elim pivots.  `Pv` / `rv` are the band and right-hand side as seen from the
// chain's end (P / mt for the chain that ascends, the mirrored copies for the one that descends), so both
// chains of a system run the same instructions in neighbouring lanes.  c[b][d] holds A[i+b+d][i+b] as updated
// so far.  A lone wave issues one f64 instruction per ~8 cycles whatever the dependencies, so the loop is
// written for instruction count (factor entries first, then 6 + 3 fused multiply-adds, one record of S+2
// doubles per pivot).  A non-positive pivot is only recorded (no data-dependent addresses).  On return the
// window holds the Schur complement columns n_elim .. n_elim+S-1 and the reduced right-hand side.
template <int S>
struct SpecWin { double c[S + 1][S + 1]; double r[S + 1]; };

template <int S>
__device__ __forceinline__ void spectral_pivot(const double* __restrict__ Pv, const double* __restrict__ rv,
                                               double* __restrict__ rec, int i, double gk, SpecWin<S>& w, bool& bad) {
#pragma unroll
  for (int d = 0; d <= S; ++d) w.c[S][d] = Pv[(i + S) * (S + 1) + d];
  w.c[S][0] += gk;
  w.r[S] = rv[i + S];
  const double d0 = w.c[0][0];
  bad |= !(d0 > 0.0);
  const double inv = rcp_cubic(d0);
  const double u = w.r[0];
  double l[S + 1];
#pragma unroll
  for (int d = 1; d <= S; ++d) l[d] = w.c[0][d] * inv;
#pragma unroll
  for (int b = 1; b <= S; ++b)
#pragma unroll
    for (int a = b; a <= S; ++a) w.c[b][a - b] = fma(-l[a], w.c[0][b], w.c[b][a - b]);
#pragma unroll
  for (int d = 1; d <= S; ++d) {
    w.r[d] = fma(-l[d], u, w.r[d]);
    rec[i * (S + 2) + d - 1] = l[d];
  }
  rec[i * (S + 2) + S] = inv;
  rec[i * (S + 2) + S + 1] = u;
#pragma unroll
  for (int b = 0; b < S; ++b) {
#pragma unroll
    for (int d = 0; d <= S; ++d) w.c[b][d] = w.c[b + 1][d];
    w.r[b] = w.r[b + 1];
  }
}

template <int S>
__device__ __forceinline__ bool spectral_forward(const double* __restrict__ Pv, const double* __restrict__ rv,
                                                 double* __restrict__ rec, int n_elim, int n_common, double gk,
                                                 SpecWin<S>& w) {
#pragma unroll
  for (int b = 0; b < S; ++b) {
#pragma unroll
    for (int d = 0; d <= S; ++d) w.c[b][d] = Pv[b * (S + 1) + d];
    w.c[b][0] += gk;
    w.r[b] = rv[b];
  }
  bool bad = false;
#pragma unroll 4
  for (int i = 0; i < n_common; ++i) spectral_pivot<S>(Pv, rv, rec, i, gk, w, bad);
  if (n_elim > n_common) spectral_pivot<S>(Pv, rv, rec, n_common, gk, w, bad);      // (the halves differ by at most one)
  return !bad;
}

// The same chain in two passes (the dataflow tail of the fused V launch, btf_fused.h): the factorisation alone - it needs
// the band and g_k only, so it can run while the right-hand sides are still being summed - and, later, the forward
// substitution from the recorded factor entries with the draw's  w = D^-1 u + D^-1/2 z  folded in.  Every value goes
// through the operations of spectral_pivot in the same order (l = c0d * inv; r_d <- fma(-l_d, u, r_d) by ascending
// pivot; w = fma(u, inv, z sqrt(inv))): bit-identical records.
template <int S>
struct SpecWinC { double c[S + 1][S + 1]; };

template <int S>
__device__ __forceinline__ void spectral_factor_pivot(const double* __restrict__ Pv, double* __restrict__ rec, int i, double gk,
                                                      SpecWinC<S>& w, bool& bad) {
#pragma unroll
  for (int d = 0; d <= S; ++d) w.c[S][d] = Pv[(i + S) * (S + 1) + d];
  w.c[S][0] += gk;
  const double d0 = w.c[0][0];
  bad |= !(d0 > 0.0);
  const double inv = rcp_cubic(d0);
  double l[S + 1];
#pragma unroll
  for (int d = 1; d <= S; ++d) l[d] = w.c[0][d] * inv;
#pragma unroll
  for (int b = 1; b <= S; ++b)
#pragma unroll
    for (int a = b; a <= S; ++a) w.c[b][a - b] = fma(-l[a], w.c[0][b], w.c[b][a - b]);
#pragma unroll
  for (int d = 1; d <= S; ++d) rec[i * (S + 2) + d - 1] = l[d];
  rec[i * (S + 2) + S] = inv;
#pragma unroll
  for (int b = 0; b < S; ++b)
#pragma unroll
    for (int d = 0; d <= S; ++d) w.c[b][d] = w.c[b + 1][d];
}

template <int S>
__device__ __forceinline__ bool spectral_factor(const double* __restrict__ Pv, double* __restrict__ rec, int n_elim, int n_common,
                                                double gk, SpecWinC<S>& w) {
#pragma unroll
  for (int b = 0; b < S; ++b) {
#pragma unroll
    for (int d = 0; d <= S; ++d) w.c[b][d] = Pv[b * (S + 1) + d];
    w.c[b][0] += gk;
  }
  bool bad = false;
#pragma unroll 4
  for (int i = 0; i < n_common; ++i) spectral_factor_pivot<S>(Pv, rec, i, gk, w, bad);
  if (n_elim > n_common) spectral_factor_pivot<S>(Pv, rec, n_common, gk, w, bad);
  return !bad;
}

// forward substitution from the records; zs[i] = z_i sqrt(1 / D_i) (made by other waves from the recorded reciprocals);
// slot S+1 of the record receives w_i.  r[0 .. S-1] on return: the reduced right-hand side behind the last pivot.
template <int S>
__device__ __forceinline__ void spectral_forward_rhs(const double* __restrict__ rv, const double* __restrict__ zs, double* __restrict__ rec,
                                                     int n_elim, int n_common, double (&r)[S + 1]) {
#pragma unroll
  for (int b = 0; b < S; ++b) r[b] = rv[b];
  auto step = [&](int i) {
    r[S] = rv[i + S];
    const double u = r[0];
#pragma unroll
    for (int d = 1; d <= S; ++d) r[d] = fma(-rec[i * (S + 2) + d - 1], u, r[d]);
    rec[i * (S + 2) + S + 1] = fma(u, rec[i * (S + 2) + S], zs[i]);
#pragma unroll
    for (int b = 0; b < S; ++b) r[b] = r[b + 1];
  };
#pragma unroll 4
  for (int i = 0; i < n_common; ++i) step(i);
  if (n_elim > n_common) step(n_common);
}

// x = L^-T w for one chain, in place in the records (slot S+1): x_i = w_i - sum_d L[i+d, i] x_{i+d}, the term
// of x_{i+1} last (the dependent chain).  x[1..S] on entry: the unknowns behind the chain's last pivot.
template <int S>
__device__ __forceinline__ void spectral_backward(double* __restrict__ rec, int n_elim, double (&x)[S + 1]) {
#pragma unroll 4
  for (int i = n_elim - 1; i >= 0; --i) {
    double acc = rec[i * (S + 2) + S + 1];
#pragma unroll
    for (int d = S; d >= 1; --d) acc = fma(-rec[i * (S + 2) + d - 1], x[d], acc);
#pragma unroll
    for (int d = S; d >= 2; --d) x[d] = x[d - 1];
    x[1] = acc;
    rec[i * (S + 2) + S + 1] = acc;
  }
}

// KC > 0: nembeds as a compile-time constant (instances for the reference's default tf_order = 2): the K-long rotations by
// the eigenvectors unroll instead of paying an LDS round trip per term - 13.5 -> 12.2 us at C3
template <int S, bool RG = false, int KC = 0>
__global__ __launch_bounds__(VS_THREADS) void v_spectral_kernel(VSpecArgs a) {
  if (a.hyp) {
    if (a.hyp_noise) { a.s = 1.0 / a.hyp[HYP_NU2]; a.sR = a.s * a.Rrep; }
    a.lam2 = a.hyp[HYP_LAM2];
  }
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = blockIdx.x, jg = a.col0 + j;
  const int K = KC > 0 ? KC : a.K, T = a.T, n = T * K, KK = tri(K);
  constexpr int D1 = S + 1, RS = S + 2, WN = S * (S + 1) + S;
  constexpr int MAXE = VS_MAXE;
  const VsLayout L = vs_layout(T, K, a.TF, a.nD, RG);
  const int Tp = L.Tp;
  int nl, nr, ns;
  spectral_split(T, S, nl, nr, ns);
  double* Ush = lds + L.U;
  double* gsh = lds + L.g;
  double* itau = lds + L.itau;
  double* P = lds + L.P;
  double* Pm = lds + L.Pm;
  double* mraw = lds + L.mraw;
  double* mt = lds + L.mt;
  double* mtm = lds + L.mtm;
  double* zz = lds + L.zz;
  double* rec = RG ? a.rec_g + (size_t)j * n * RS : lds + L.rec;
  double* win = lds + L.win;
  double* flag = lds + L.flag;
  long long stamp[6];
  stamp[0] = __builtin_amdgcn_s_memtime();

  // ---- loads: stencil of this thread's band entry, eigen-system, Tau2 of the column, accumulation partials ----
  const int pidx = tid;                                   // band entry (t, d) = pidx / D1, pidx % D1 (first pass)
  int se0 = 0, se1 = 0;
  int srow[MAXE];
  double scf[MAXE];
  // (records in LDS and a.pband set: the prior band of the column comes precomputed in this kernel's own arithmetic - the
  //  host builds it when Tau2 / lam2 have stood still since the last V half-sweep, btf_abi.hip - and neither the stencil nor
  //  Tau2 is fetched: 49 KB less per workgroup in the cold batch of loads)
  const bool lazyb = !RG && a.pband != nullptr;
  double pbv = 0.0;
  if (lazyb && pidx < T * D1) pbv = a.pband[(size_t)j * T * D1 + pidx];
  if (pidx < T * D1 && !lazyb) {
    se0 = a.st_ptr[pidx]; se1 = a.st_ptr[pidx + 1];
#pragma unroll
    for (int u = 0; u < MAXE; u += 4) {
      const int4 r4 = *reinterpret_cast<const int4*>(a.st_drow + (size_t)pidx * MAXE + u);
      srow[u] = r4.x; srow[u + 1] = r4.y; srow[u + 2] = r4.z; srow[u + 3] = r4.w;
    }
#pragma unroll
    for (int u = 0; u < MAXE; u += 2) {
      const double2 c2 = *reinterpret_cast<const double2*>(a.st_dcoef + (size_t)pidx * MAXE + u);
      scf[u] = c2.x; scf[u + 1] = c2.y;
    }
  }
  const bool curve_col = a.cv.ptr && a.cv.ptr[jg + 1] > a.cv.ptr[jg];      // (workgroup-uniform)
  const bool own_eig = curve_col && !a.eig_cols_ready;
  if (!own_eig) {
    const double* __restrict__ es = curve_col ? a.eig_cols + (size_t)jg * (K + K * K + 8) : a.eig;
    for (int idx = tid; idx < K + K * K; idx += VS_THREADS) {
      const double v = es[idx];
      if (idx < K) gsh[idx] = v; else Ush[idx - K] = v;
    }
  }
  // 1 / (lam2 Tau2[j, r]) once per penalty row (the diagonal matrix of factor.py:404)
  if constexpr (RG) {
    for (int i0 = tid; i0 < a.nD; i0 += 8 * VS_THREADS) {   // (long depth axes: eight loads in flight; nD = 1109 at the flu shape)
      double tv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int idx = i0 + u * VS_THREADS; if (idx < a.nD) tv[u] = a.Tau2[(size_t)jg * a.nD + idx]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int idx = i0 + u * VS_THREADS; if (idx < a.nD) itau[idx] = 1.0 / (a.lam2 * tv[u]); }
    }
  } else if (!lazyb) {
    for (int idx = tid; idx < a.nD; idx += VS_THREADS) itau[idx] = 1.0 / (a.lam2 * a.Tau2[(size_t)jg * a.nD + idx]);
  }
  if (RG && a.nch == 1) {
    // long depth axes, one chunk (few rows): eight elements' loads in flight per thread instead of two
    for (int b0 = tid; b0 < n; b0 += 8 * VS_THREADS) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = b0 + u * VS_THREADS;
        if (e < n) { const int k = e / T, t = e - k * T; v[u] = a.part[(size_t)k * a.ld + (size_t)j * T + t]; }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = b0 + u * VS_THREADS;
        if (e < n) { const int k = e / T, t = e - k * T; mraw[t * K + k] = 0.0 + v[u]; }
      }
    }
  } else {
    const size_t st = (size_t)K * a.ld;
    for (int e0 = tid; e0 < n; e0 += 2 * VS_THREADS) {       // element e = k*T + t: coalesced along t
      const int e1 = e0 + VS_THREADS;
      const bool h1 = e1 < n;
      const int k0 = e0 / T, t0 = e0 - k0 * T;
      const int e1c = h1 ? e1 : e0;
      const int k1 = e1c / T, t1 = e1c - k1 * T;
      const double* p0 = a.part + (size_t)k0 * a.ld + (size_t)j * T + t0;
      const double* p1 = a.part + (size_t)k1 * a.ld + (size_t)j * T + t1;
      double s0 = 0.0, s1 = 0.0;
      int c = 0;
      for (; c + 4 <= a.nch; c += 4) {                       // fixed order, four chunks in flight per element
        const double x0 = p0[(size_t)c * st], x1 = p0[(size_t)(c + 1) * st], x2 = p0[(size_t)(c + 2) * st], x3 = p0[(size_t)(c + 3) * st];
        const double y0 = p1[(size_t)c * st], y1 = p1[(size_t)(c + 1) * st], y2 = p1[(size_t)(c + 2) * st], y3 = p1[(size_t)(c + 3) * st];
        s0 += x0; s0 += x1; s0 += x2; s0 += x3;
        s1 += y0; s1 += y1; s1 += y2; s1 += y3;
      }
      for (; c < a.nch; ++c) { s0 += p0[(size_t)c * st]; s1 += p1[(size_t)c * st]; }
      mraw[t0 * K + k0] = s0;
      if (h1) mraw[t1 * K + k1] = s1;
    }
  }
  const int scnt = se1 - se0;
  for (int idx = tid; idx < (T + S + 1) * D1; idx += VS_THREADS) Pm[idx] = 0.0;
  if (own_eig) {
    double* Gl = lds + L.eG;
    double* eo = lds + L.eo;
    const int ne = K + K * K + 8;
    double* ecol = a.eig_cols + (size_t)jg * ne;
    for (int idx = tid; idx < ne; idx += VS_THREADS) eo[idx] = ecol[idx];       // the column's previous solution
    reduce_gram(a.gpart, a.ngp, KK, 1.0, lds + L.gs, Gl);                       // W'W (ends with a barrier)
    curve_column_gram(a.cv, a.cv_W, jg, K, KK, 1.0 / a.Rrep, Gl, lds + L.esc, EIG_LDS_DOUBLES);
    __syncthreads();
    if (wave == 0) gram_eig_wave<KC>(Gl, 1, K, eo, lds + L.esc);
    __syncthreads();
    for (int idx = tid; idx < ne; idx += VS_THREADS) {
      const double v = eo[idx];
      ecol[idx] = v;
      if (idx < K) gsh[idx] = v; else if (idx < K + K * K) Ush[idx - K] = v;
    }
  }
  __syncthreads();
  stamp[1] = __builtin_amdgcn_s_memtime();
  // ---- prior band P[t][d] = sum_r Delta[r,t] Delta[r,t+d] / (lam2 Tau2_r)  (rows ascending), its mirror image,
  //      the rotated right-hand sides and their mirror image ------------------------------------------------
  for (int idx = tid; idx < (T + S + 1) * D1; idx += VS_THREADS) {
    double s = 0.0;
    if (lazyb) {
      if (idx < T * D1) s = idx == pidx ? pbv : a.pband[(size_t)j * T * D1 + idx];
    } else if (idx == pidx) {
#pragma unroll
      for (int u = 0; u < MAXE; ++u) if (u < scnt) s = fma(scf[u], itau[srow[u]], s);
    } else if (RG && idx < T * D1 && a.pband) {
      s = a.pband[(size_t)j * T * D1 + idx];               // (long depth axes: prior_band_kernel formed the band, all entries in parallel)
    } else if (!RG && idx < T * D1) {
      for (int e = a.st_ptr[idx]; e < a.st_ptr[idx + 1]; ++e) s = fma(a.st_coef[e], itau[a.st_row[e]], s);
    } else if (idx < T * D1) {
      // (long depth axes: the fixed-slot stencil again - MAXE independent loads per entry, zero coefficients in the unused
      //  slots - instead of a CSR walk whose loads depend on one another: 28 k -> cycles of this phase at T = 370)
      int rw[MAXE];
      double cf[MAXE];
#pragma unroll
      for (int u = 0; u < MAXE; u += 4) {
        const int4 r4 = *reinterpret_cast<const int4*>(a.st_drow + (size_t)idx * MAXE + u);
        rw[u] = r4.x; rw[u + 1] = r4.y; rw[u + 2] = r4.z; rw[u + 3] = r4.w;
      }
#pragma unroll
      for (int u = 0; u < MAXE; u += 2) {
        const double2 c2 = *reinterpret_cast<const double2*>(a.st_dcoef + (size_t)idx * MAXE + u);
        cf[u] = c2.x; cf[u + 1] = c2.y;
      }
      const int cnt = a.st_ptr[idx + 1] - a.st_ptr[idx];
#pragma unroll
      for (int u = 0; u < MAXE; ++u) if (u < cnt) s = fma(cf[u], itau[rw[u]], s);
    }
    P[idx] = s;
    const int t = idx / D1, d = idx - t * D1;
    if (t + d < T) Pm[(T - 1 - t - d) * D1 + d] = s;        // the same entry seen from the far end
  }
  for (int idx = tid; idx < K * Tp; idx += VS_THREADS) {
    const int k = idx / Tp, t = idx - k * Tp;
    double s = 0.0;
    if (t < T) {
      for (int kk = 0; kk < K; ++kk) s = fma(Ush[kk * K + k], mraw[t * K + kk], s);
      s *= a.s;
    }
    mt[idx] = s;
    mtm[k * Tp + (t < T ? T - 1 - t : t)] = s;
  }
  __syncthreads();
  stamp[2] = __builtin_amdgcn_s_memtime();
  // ---- the 2K chains (wave 0: lane k ascends system k, lane K+k descends it) while the other waves draw the normals ----
  if (wave == 0) {
    const bool chain = tid < (nr > 0 ? 2 * K : K);
    const int side = tid >= K ? 1 : 0;
    const int k = chain ? tid - side * K : 0;
    const double* Pv = side ? Pm : P;
    const double* rv = (side ? mtm : mt) + k * Tp;
    double* crec = rec + (size_t)(k * T + (side ? nl : 0)) * RS;
    const int n_elim = side ? nr : nl;
    const int n_common = nr > 0 ? (nl < nr ? nl : nr) : nl;
    double shift = 0.0, eps = a.eps0;
    int tried = 0;
    bool ok;
    while (true) {
      bool good = true;
      const double gk = fma(gsh[k], a.sR, shift);
      SpecWin<S> w;
      if (chain) {
        good = spectral_forward<S>(Pv, rv, crec, n_elim, n_common, gk, w);
        if (ns > 0) {                                    // park the window for the separator system
          double* wp = win + (size_t)(side * K + k) * WN;
#pragma unroll
          for (int b = 0; b < S; ++b) {
#pragma unroll
            for (int d = 0; d <= S; ++d) wp[b * (S + 1) + d] = w.c[b][d];
            wp[S * (S + 1) + b] = w.r[b];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (ns > 0 && tid < K) {
        // separator system: both halves' Schur complements carry the original entries once too often
        const double* wl = win + (size_t)k * WN;
        const double* wr = win + (size_t)(K + k) * WN;
        double Sg[S][S], us[S];
#pragma unroll
        for (int b = 0; b < S; ++b) {
#pragma unroll
          for (int aa = b; aa < S; ++aa)
            Sg[aa][b] = wl[b * (S + 1) + aa - b] + wr[(S - 1 - aa) * (S + 1) + aa - b] - (P[(nl + b) * D1 + aa - b] + (aa == b ? gk : 0.0));
          us[b] = wl[S * (S + 1) + b] + wr[S * (S + 1) + S - 1 - b] - mt[k * Tp + nl + b];
        }
        double* srec = rec + (size_t)(k * T + nl + nr) * RS;
#pragma unroll
        for (int c = 0; c < S; ++c) {
          const double d0 = Sg[c][c];
          good &= d0 > 0.0;
          const double inv = rcp_cubic(d0);
#pragma unroll
          for (int d = 1; d <= S; ++d) {
            double l = 0.0;
            if (c + d < S) {
              l = Sg[c + d][c] * inv;
              us[c + d] = fma(-l, us[c], us[c + d]);
#pragma unroll
              for (int b2 = 1; b2 <= d; ++b2) Sg[c + d][c + b2] = fma(-l, Sg[c + b2][c], Sg[c + d][c + b2]);
            }
            srec[c * RS + d - 1] = l;
          }
          srec[c * RS + S] = inv;
          srec[c * RS + S + 1] = us[c];
        }
      }
      ok = __builtin_amdgcn_readfirstlane(__ballot(!good) == 0ULL ? 1 : 0) != 0;
      if (ok || tried >= a.attempts) break;
      shift += eps;   // fast_mvn.py:64-68: cumulative eps, eps *= 10
      eps *= 10.0;
      ++tried;
    }
    if (tid == 0) { flag[0] = ok ? 1.0 : 0.0; flag[1] = (double)tried; }
  } else {
    for (int idx = tid - WAVE; idx < n; idx += VS_THREADS - WAVE)
      zz[idx] = a.z ? a.z[(size_t)jg * n + idx] : philox_normal(a.seed, a.stream, (unsigned long long)jg * n + idx);
  }
  __syncthreads();
  const bool ok = flag[0] != 0.0;
  if (tid == 0) a.tries[j] = (int)flag[1];
  if (!ok) {
    if (tid == 0 && atomicCAS(&a.status[0], 0, 1) == 0) a.status[1] = jg;
    return;
  }
  stamp[3] = __builtin_amdgcn_s_memtime();
  // ---- w = D^-1 u + D^-1/2 z  (pivot order) -----------------------------------------------------------
  {
    // (records in HBM - RG - pay a global round trip per batch: sixteen records per thread and batch)
    constexpr int WB = RG ? 16 : 1;
    for (int i0 = tid; i0 < n; i0 += WB * VS_THREADS) {
      double iv[WB], w[WB];
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        const int idx = i0 + u * VS_THREADS;
        if (idx < n) { iv[u] = rec[(size_t)idx * RS + S]; w[u] = rec[(size_t)idx * RS + S + 1]; }
      }
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        const int idx = i0 + u * VS_THREADS;
        if (idx < n) rec[(size_t)idx * RS + S + 1] = fma(w[u], iv[u], zz[idx] * sqrt(iv[u]));
      }
    }
  }
  __syncthreads();
  double x[S + 1];
#pragma unroll
  for (int d = 0; d <= S; ++d) x[d] = 0.0;
  if (wave == 0) {
    if (ns > 0 && tid < K) {                              // separator: S x S unit upper solve
      double* srec = rec + (size_t)(tid * T + nl + nr) * RS;
      if constexpr (RG) {                                  // (records in HBM: all of the separator's fetched first)
        double sl[S][S], sw[S];
#pragma unroll
        for (int c = 0; c < S; ++c) {
          sw[c] = srec[c * RS + S + 1];
#pragma unroll
          for (int d = 1; d <= S; ++d) sl[c][d - 1] = c + d < S ? srec[c * RS + d - 1] : 0.0;
        }
#pragma unroll
        for (int c = S - 1; c >= 0; --c) {
          double acc = sw[c];
#pragma unroll
          for (int d = 1; d <= S; ++d) if (c + d < S) acc = fma(-sl[c][d - 1], sw[c + d], acc);
          sw[c] = acc;
          srec[c * RS + S + 1] = acc;
        }
      } else {
#pragma unroll
        for (int c = S - 1; c >= 0; --c) {
          double acc = srec[c * RS + S + 1];
#pragma unroll
          for (int d = 1; d <= S; ++d) if (c + d < S) acc = fma(-srec[c * RS + d - 1], srec[(c + d) * RS + S + 1], acc);
          srec[c * RS + S + 1] = acc;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (tid < (nr > 0 ? 2 * K : K)) {
      const int side = tid >= K ? 1 : 0, k = tid - side * K;
      if (ns > 0) {
        const double* srec = rec + (size_t)(k * T + nl + nr) * RS;
#pragma unroll
        for (int d = 1; d <= S; ++d) x[d] = srec[(side ? S - d : d - 1) * RS + S + 1];
      }
      if constexpr (!RG) spectral_backward<S>(rec + (size_t)(k * T + (side ? nl : 0)) * RS, side ? nr : nl, x);
    }
  }
  if constexpr (RG) {
    // Records in HBM: the back-substitution would pay a global round trip per pivot (the loads do not depend on the data,
    // but only four are in flight: 185 pivots cost ~50 us at the flu shape).  The other three waves stage the records of
    // the next CHB pivots of every chain in LDS (the normals' area is free by now; chunk q+1 while wave 0 consumes chunk
    // q) - the chain reads LDS only, its results go out as plain stores.  One barrier per chunk.
    const int nchains = nr > 0 ? 2 * K : K, nmax = nl > nr ? nl : nr;
    // (staging area: the mirrored right-hand sides and the normals, contiguous in the layout and both dead by now)
    const int room = K * Tp + n;
    int CHB = (room / (2 * nchains) - 1) / RS;
    CHB = CHB > 48 ? 48 : CHB;
    if (CHB >= 2) {
      const int cst = CHB * RS + 1, tot = nchains * cst;             // (odd chain stride: the chains' lanes on different banks)
      double* stage = mtm;
      // chunk q of chain c: pivots i_hi = nc - 1 - q CHB down to i_lo = max(i_hi - CHB + 1, 0) - one contiguous span of the
      // records, copied as it lies (ascending i; the chain walks it backwards): no index arithmetic per word
      auto load_chunk = [&](int q, int t0, int nthreads) {
        double* dst = stage + (q & 1) * tot;
        for (int c0 = 0; c0 < nchains; c0 += 4) {                    // four chains' spans in flight per thread
          double v[4][2];
          int len[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = c0 + u;
            len[u] = 0;
            if (c < nchains) {
              const int side = c >= K ? 1 : 0, k = c - side * K, ihi = (side ? nr : nl) - 1 - q * CHB;
              if (ihi >= 0) {
                const int ilo = ihi - CHB + 1 > 0 ? ihi - CHB + 1 : 0;
                len[u] = (ihi - ilo + 1) * RS;
                const double* src = rec + (size_t)(k * T + (side ? nl : 0) + ilo) * RS;
#pragma unroll
                for (int w = 0; w < 2; ++w) if (t0 + w * nthreads < len[u]) v[u][w] = src[t0 + w * nthreads];
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = 0; w < 2; ++w) if (t0 + w * nthreads < len[u]) dst[(c0 + u) * cst + t0 + w * nthreads] = v[u][w];
        }
      };
      const int nq = (nmax + CHB - 1) / CHB;
      __syncthreads();                                               // (w and the separator's x are in place; zz is dead)
      load_chunk(0, tid, VS_THREADS);
      __syncthreads();
      for (int q = 0; q < nq; ++q) {
        if (wave != 0) { if (q + 1 < nq) load_chunk(q + 1, tid - WAVE, VS_THREADS - WAVE); }
        else if (tid < nchains) {
          const int side = tid >= K ? 1 : 0, k = tid - side * K, nc = side ? nr : nl;
          double* rc = rec + (size_t)(k * T + (side ? nl : 0)) * RS;
          const int ihi = nc - 1 - q * CHB, ilo = ihi - CHB + 1 > 0 ? ihi - CHB + 1 : 0;
          const double* st = stage + (q & 1) * tot + tid * cst - (size_t)ilo * RS;      // st[i RS + f]: record of pivot i
          for (int i = ihi; i >= ilo; --i) {
            double acc = st[i * RS + S + 1];
#pragma unroll
            for (int d = S; d >= 1; --d) acc = fma(-st[i * RS + d - 1], x[d], acc);
#pragma unroll
            for (int d = S; d >= 2; --d) x[d] = x[d - 1];
            x[1] = acc;
            rc[(size_t)i * RS + S + 1] = acc;
          }
        }
        __syncthreads();
      }
    } else if (wave == 0 && tid < nchains) {
      const int side = tid >= K ? 1 : 0, k = tid - side * K;
      spectral_backward<S>(rec + (size_t)(k * T + (side ? nl : 0)) * RS, side ? nr : nl, x);
    }
  }
  __syncthreads();
  stamp[4] = __builtin_amdgcn_s_memtime();
  // ---- rotate back, write V[j] (depth-major), Gram share ---------------------------------------------
  double* xout = mraw;
  double sse_acc = 0.0;
  const double inv_s2 = -2.0 / a.s;
  // the solution in the eigen-basis, x~[k T + pivot]: slot S+1 of the records - copied to LDS first when the records live
  // in HBM (sixteen loads in flight per thread; the mirrored right-hand sides' area is free), so that the rotation below
  // reads LDS either way
  const double* xs = rec + S + 1;
  constexpr int xstride = RG ? 1 : RS;
  if constexpr (RG) {
    double* xl = mtm;                                      // K Tp >= n doubles
    for (int i0 = tid; i0 < n; i0 += 16 * VS_THREADS) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int e = i0 + u * VS_THREADS; if (e < n) v[u] = rec[(size_t)e * RS + S + 1]; }
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int e = i0 + u * VS_THREADS; if (e < n) xl[e] = v[u]; }
    }
    __syncthreads();
    xs = xl;
  }
  for (int idx = tid; idx < n; idx += VS_THREADS) {
    const int t = idx / K, k = idx - t * K;
    const int pos = t < nl ? t : (t < nl + ns ? nl + nr + (t - nl) : nl + (T - 1 - t));
    double s = 0.0;
    for (int kk = 0; kk < K; ++kk) s = fma(Ush[k * K + kk], xs[(size_t)(kk * T + pos) * xstride], s);
    xout[idx] = s;
    a.V[(size_t)jg * n + idx] = s;
    if (a.sse_out) {     // in the eigen-basis: R lambda_k x~^2 - 2 x~ m~  (mt holds s U'm)
      const double xt = xs[(size_t)(k * T + pos) * xstride];
      sse_acc = fma(xt, fma(a.Rrep * gsh[k], xt, inv_s2 * mt[k * Tp + t]), sse_acc);
    }
  }
  if (a.sse_out) {       // fixed order: wave butterflies, then the four waves' sums
    const double v = wave_sum(sse_acc);
    if ((tid & 63) == 0) flag[2 + wave] = v;
    __syncthreads();
    if (tid == 0) a.sse_out[j] = (flag[2] + flag[3]) + (flag[4] + flag[5]);
  }
  if (a.gout) {
    __syncthreads();
    int ng = VS_THREADS / KK;
    if (ng > 16) ng = 16;
    if (ng < 1) ng = 1;
    const int g = tid / KK, q = tid - g * KK;
    int p = 0;
    while ((p + 1) * (p + 2) / 2 <= q) ++p;
    const int pq = q - p * (p + 1) / 2;
    double* scratch = lds + L.gs;
    if (g < ng) {
      double s = 0.0;
      for (int t = g; t < T; t += ng) s = fma(xout[t * K + p], xout[t * K + pq], s);
      scratch[g * KK + q] = s;
    }
    __syncthreads();
    if (tid < KK) {
      double s = 0.0;
      for (int b = 0; b < ng; ++b) s += scratch[b * KK + tid];
      a.gout[(size_t)j * KK + tid] = s;
    }
  }
  stamp[5] = __builtin_amdgcn_s_memtime();
  if (a.dbg && tid == 0)
    for (int i = 0; i < 6; ++i) a.dbg[(size_t)j * 6 + i] = stamp[i];
}

}  // namespace btf
