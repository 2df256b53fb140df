// Generalized analytic slice sampling (GASS) for the constrained non-conjugate model.
//
// Reference: gass() (gass.py:13-130), called per row by _resample_W_i (factor.py:665-711) and per column by
// _resample_V_j (factor.py:759-846) of ConstrainedNonconjugateBayesianTensorFiltering.  One update of a vector x
// under linear constraints  A x >= c:
//   slice height  ll(x) + log u                                                       (gass.py:21)
//   proposal v from the prior (sample_mvn)                                            (gass.py:24)
//   every constraint row with a = A x, b = A v: the angles theta on the ellipse x cos(theta) + v sin(theta) where it
//   holds are an interval or the complement of one (theta1,2 = 2 atan((b +- sqrt(a^2+b^2-c^2)) / (a+c)),
//   gass.py:38-63); the reference intersects them NUMERICALLY on linspace(-pi, pi, 10000)   (gass.py:66-80)
//   at most ngrid of the surviving grid angles, drawn without replacement              (gass.py:110-111)
//   likelihood of every candidate, those above the slice kept, one of them uniformly   (gass.py:114-126)
// The constraints of the model are  sum_t Cons[c,t] (w_i . v_jt) >= Cc[c]  for every (i, j) curve and constraint c
// (factor.py:713-727 for a row: J*ncols rows of the form (Cons_c V_j) . w_i; :848-855 for a column: J*nrows rows).
//
// Here all rows (resp. all columns) are updated by one set of launches:
//   gass_av_kernel            AV[j][c][:] = sum_t Cons[c,t] V[j,t,:]                          (rows only)
//   gass_analyse_rows/cols    a, b per constraint -> removed index ranges of the 10000-angle grid as +1/-1 marks
//                             in an LDS difference array (integer atomics: deterministic), interval bounds by a
//                             fixed-order reduction, prefix scan -> validity of every grid angle; device mode: the
//                             subsample (partial Fisher-Yates over the valid list, Philox)
//   gass_eval_rows/cols       Poisson log-likelihood of every candidate of every chain: lanes = candidates, the cells
//                             of a chain staged tile by tile in LDS (two dot products per cell, ngrid rates)
//   gass_select_kernel        candidates above the slice, one uniformly; x = x0 cos + v sin
// The likelihood is a device likelihood (Poisson, log or identity link, from the hoisted statistics) as for the
// elliptical slice sampler (btf_ess.h).
#pragma once
#include "btf_ess.h"

namespace btf {

constexpr int GASS_GRID = 10000;          // gass.py:68
constexpr int GASS_THREADS = 256;
constexpr int GASS_MAXC = 128;            // candidates per chain (ngrid <= 128: two per lane)
constexpr int GASS_PER = (GASS_GRID + GASS_THREADS - 1) / GASS_THREADS;    // grid angles per thread in the scans

// np.linspace(-pi, pi, 10000)[g]: start + g * step, the last point set to the stop value exactly
__device__ __forceinline__ double gass_grid_angle(int g, int npts) {
  const double pi = 3.14159265358979323846;
  if (g >= npts - 1) return pi;
  const double step = (2.0 * pi) / (double)(npts - 1);
  return fma((double)g, step, -pi);
}

struct GassScratch {        // LDS of an analyse workgroup
  int diff[GASS_GRID + 8];
  int tsum[GASS_THREADS];
  double rmin[GASS_THREADS], rmax[GASS_THREADS];
  int flags[GASS_THREADS];
};

// One constraint row.  any |= 1 if it restricts the ellipse, |= 2 if it is of the interval kind.
__device__ __forceinline__ void gass_constraint(double a, double b, double c, int* diff, double& tmin, double& tmax, int& any) {
  const double sq = a * a + b * b - c * c;
  if (!(sq >= 0.0) || a == -c) return;                     // the whole ellipse satisfies it (gass.py:51-55)
  const double rt = sqrt(sq), den = a + c;
  const double t1 = 2.0 * atan((b + rt) / den), t2 = 2.0 * atan((b - rt) / den);
  const double lo = fmin(t1, t2), hi = fmax(t1, t2);
  any |= 1;
  if (a * a < c * c) {                                      // convex: the open interval (lo, hi) is invalid (gass.py:71-73)
    const double pi = 3.14159265358979323846;
    const double h = (2.0 * pi) / (double)(GASS_GRID - 1);
    int gf = (int)floor((lo + pi) / h);                     // first index with angle > lo
    gf = max(0, min(GASS_GRID - 1, gf));
    while (gf < GASS_GRID && !(gass_grid_angle(gf, GASS_GRID) > lo)) ++gf;
    while (gf > 0 && gass_grid_angle(gf - 1, GASS_GRID) > lo) --gf;
    int gl = (int)floor((hi + pi) / h);                     // last index with angle < hi
    gl = max(0, min(GASS_GRID - 1, gl));
    while (gl >= 0 && !(gass_grid_angle(gl, GASS_GRID) < hi)) --gl;
    while (gl < GASS_GRID - 1 && gass_grid_angle(gl + 1, GASS_GRID) < hi) ++gl;
    if (gf <= gl) { atomicAdd(&diff[gf], 1); atomicAdd(&diff[gl + 1], -1); }
  } else {                                                  // concave: only [lo + eps, hi - eps] is valid (gass.py:76-80)
    any |= 2;
    tmin = fmax(tmin, lo);
    tmax = fmin(tmax, hi);
  }
}

__device__ __forceinline__ int any_of(const GassScratch& S) { return S.flags[0]; }     // (valid after gass_finish_grid)

// After every thread has processed its constraints: validity of the grid, written as bytes; returns (to every thread)
// the number of valid angles.  info[0] = valid count, info[1] = 1 if no constraint restricts the ellipse (the
// candidate grid is then linspace(-pi, pi, ngrid), gass.py:81-83).
// `list` (LDS, may alias S.diff): the valid indices, ascending, when want_list.
__device__ inline int gass_finish_grid(GassScratch& S, double tmin, double tmax, int any, unsigned char* __restrict__ vmask,
                                       int* __restrict__ info, bool want_list, int* list) {
  const int tid = threadIdx.x;
  S.rmin[tid] = tmin; S.rmax[tid] = tmax; S.flags[tid] = any;
  __syncthreads();
  for (int off = GASS_THREADS / 2; off > 0; off >>= 1) {
    if (tid < off) {
      S.rmin[tid] = fmax(S.rmin[tid], S.rmin[tid + off]);
      S.rmax[tid] = fmin(S.rmax[tid], S.rmax[tid + off]);
      S.flags[tid] |= S.flags[tid + off];
    }
    __syncthreads();
  }
  const int all = S.flags[0];
  const double lo = S.rmin[0] + 1e-6, hi = S.rmax[0] - 1e-6;      // eps of gass.py:47
  // prefix scan of the difference array: thread t owns angles [t*PER, (t+1)*PER)
  const int g0 = tid * GASS_PER, g1 = min(g0 + GASS_PER, GASS_GRID);
  int run = 0;
  for (int g = g0; g < g1; ++g) run += S.diff[g];
  S.tsum[tid] = run;
  __syncthreads();
  int base = 0;
  for (int t = 0; t < tid; ++t) base += S.tsum[t];
  __syncthreads();
  static_assert(GASS_PER <= 64, "validity bits of a thread's angles fit one word");
  int cnt = 0, acc = base;
  unsigned long long bits = 0ULL;
  for (int g = g0; g < g1; ++g) {
    acc += S.diff[g];
    bool ok = acc == 0;
    if (ok && (all & 2)) {
      const double th = gass_grid_angle(g, GASS_GRID);
      ok = th >= lo && th <= hi;
    }
    vmask[g] = ok ? 1 : 0;
    if (ok) { bits |= 1ULL << (g - g0); ++cnt; }
  }
  S.tsum[tid] = cnt;
  __syncthreads();                       // (every read of S.diff is behind us: `list` may alias it)
  int off0 = 0, total = 0;
  for (int t = 0; t < GASS_THREADS; ++t) { const int v = S.tsum[t]; if (t < tid) off0 += v; total += v; }
  if (want_list) {
    int w = off0;
    for (int g = g0; g < g1; ++g) if (bits >> (g - g0) & 1ULL) list[w++] = g;
  }
  if (tid == 0) { info[0] = total; info[1] = (all & 1) ? 0 : 1; }
  __syncthreads();
  return total;
}

// Device mode: the candidate angles of this chain.  No restricting constraint: linspace(-pi, pi, ngrid); at most ngrid
// valid angles: all of them, ascending; more: ngrid of them without replacement (partial Fisher-Yates over the valid
// list by thread 0, Philox keyed by (seed, chain)).  list: LDS, GASS_GRID ints (the difference array is dead by now).
__device__ inline void gass_pick(int total, int unrestricted, int ngrid, int* list,
                                 double* __restrict__ thetas, int* __restrict__ ntheta, unsigned long long seed,
                                 unsigned long long chain) {
  const int tid = threadIdx.x;
  if (unrestricted) {
    for (int g = tid; g < ngrid; g += GASS_THREADS) thetas[g] = gass_grid_angle(g, ngrid);
    if (tid == 0) *ntheta = ngrid;
    return;
  }
  if (tid == 0) {
    const int take = total < ngrid ? total : ngrid;
    if (total > ngrid) {
      for (int s = 0; s < take; ++s) {
        uint32_t r[4];
        Philox::gen(seed, chain, 0x47415353ULL + (uint64_t)s, r);
        const int pick = s + (int)(u01(r[0], r[1]) * (double)(total - s));
        const int t = list[s]; list[s] = list[min(pick, total - 1)]; list[min(pick, total - 1)] = t;
      }
    }
    for (int s = 0; s < take; ++s) thetas[s] = gass_grid_angle(list[s], GASS_GRID);
    *ntheta = take;
  }
}

// AV[j][c][k] = sum_t Cons[c][t] V[j][t][k]   (factor.py:719): one workgroup per column
static __global__ __launch_bounds__(GASS_THREADS) void gass_av_kernel(const double* __restrict__ V, const double* __restrict__ Cons,
                                                               int T, int K, int J, double* __restrict__ AV) {
  const int j = blockIdx.x;
  for (int e = threadIdx.x; e < J * K; e += GASS_THREADS) {
    const int c = e / K, k = e - c * K;
    double s = 0.0;
    for (int t = 0; t < T; ++t) s = fma(Cons[(size_t)c * T + t], V[((size_t)j * T + t) * K + k], s);
    AV[((size_t)j * J + c) * K + k] = s;
  }
}

struct GassArgs {
  const double* X0; const double* Nu;       // current state and proposal of every chain
  const double* Cons; const double* Cc; int J;      // [J][T] and [J]
  // the same matrix by its non-zeros (CSR over the constraints; nullptr: dense walk) - positivity / monotonicity rows have
  // one or two entries among T: the column analysis then does 190 instead of 8128 multiply-adds per row of W at T = 64
  const int* cs_ptr; const int* cs_idx; const double* cs_val; int cs_nnz;
  const double* AV;                          // rows: [M*J][K]
  const double* Rc; int nrc;                 // rows: fixed row constraints [nrc][K+1] or nullptr
  const double* W;                           // cols: the fixed factor
  int N, M, T, K;
  unsigned char* vmask; int* info;           // [nchains][GASS_GRID], [nchains][2]
  int pick; int ngrid; double* thetas; int* ntheta; unsigned long long seed;
};

// rows: chain i, x = W[i, :], constraints (AV[j,c,:] . x >= Cc[c]) for all (j, c), then the fixed row constraints
static __global__ __launch_bounds__(GASS_THREADS) void gass_analyse_rows_kernel(GassArgs a) {
  __shared__ GassScratch S;
  const int i = blockIdx.x, tid = threadIdx.x, K = a.K;
  for (int g = tid; g < GASS_GRID + 8; g += GASS_THREADS) S.diff[g] = 0;
  double x[EIG_MAXK], v[EIG_MAXK];
#pragma unroll
  for (int k = 0; k < EIG_MAXK; ++k) { x[k] = k < K ? a.X0[(size_t)i * K + k] : 0.0; v[k] = k < K ? a.Nu[(size_t)i * K + k] : 0.0; }
  __syncthreads();
  double tmin = -INFINITY, tmax = INFINITY;
  int any = 0;
  const int ncon = a.M * a.J;
  for (int q = tid; q < ncon + a.nrc; q += GASS_THREADS) {
    const double* __restrict__ row = q < ncon ? a.AV + (size_t)q * K : a.Rc + (size_t)(q - ncon) * (K + 1);
    double aa = 0.0, bb = 0.0;
#pragma unroll
    for (int k = 0; k < EIG_MAXK; ++k) if (k < K) { aa = fma(row[k], x[k], aa); bb = fma(row[k], v[k], bb); }
    const double cc = q < ncon ? a.Cc[q % a.J] : row[K];
    gass_constraint(aa, bb, cc, S.diff, tmin, tmax, any);
  }
  __syncthreads();
  unsigned char* vm = a.vmask + (size_t)i * GASS_GRID;
  const int total = gass_finish_grid(S, tmin, tmax, any, vm, a.info + 2 * i, a.pick != 0, S.diff);
  if (a.pick) gass_pick(total, (any_of(S) & 1) ? 0 : 1, a.ngrid, S.diff, a.thetas + (size_t)i * GASS_MAXC, a.ntheta + i, a.seed, (unsigned long long)i);
}

// columns: chain j, x = V[j] (T x K), constraints  sum_t Cons[c,t] (w_i . x_t) >= Cc[c]  for all (i, c)  (factor.py:848-855)
constexpr int GASS_RT = 32;      // rows of W per tile
static __global__ __launch_bounds__(GASS_THREADS) void gass_analyse_cols_kernel(GassArgs a) {
  __shared__ GassScratch S;
  extern __shared__ double dyn[];                   // E0[RT][T], E1[RT][T], Cons[J][T]
  const int j = blockIdx.x, tid = threadIdx.x, K = a.K, T = a.T, J = a.J;
  double* E0 = dyn;
  double* E1 = dyn + GASS_RT * T;
  double* Cn = dyn + 2 * GASS_RT * T;
  for (int g = tid; g < GASS_GRID + 8; g += GASS_THREADS) S.diff[g] = 0;
  // sparse form staged in the dense matrix's area: [nnz values][nnz column indices][J + 1 row pointers]
  const bool sparse = a.cs_ptr != nullptr;
  double* cval = Cn;
  int* cidx = reinterpret_cast<int*>(Cn + a.cs_nnz);
  int* cptr = cidx + a.cs_nnz;
  if (sparse) {
    for (int e = tid; e < a.cs_nnz; e += GASS_THREADS) { cval[e] = a.cs_val[e]; cidx[e] = a.cs_idx[e]; }
    for (int e = tid; e <= J; e += GASS_THREADS) cptr[e] = a.cs_ptr[e];
  } else {
    for (int e = tid; e < J * T; e += GASS_THREADS) Cn[e] = a.Cons[e];
  }
  const double* __restrict__ x0 = a.X0 + (size_t)j * T * K;
  const double* __restrict__ nu = a.Nu + (size_t)j * T * K;
  double tmin = -INFINITY, tmax = INFINITY;
  int any = 0;
  for (int r0 = 0; r0 < a.N; r0 += GASS_RT) {
    __syncthreads();
    for (int e = tid; e < GASS_RT * T; e += GASS_THREADS) {
      const int r = e / T, t = e - r * T;
      double s0 = 0.0, s1 = 0.0;
      if (r0 + r < a.N) {
        const double* __restrict__ w = a.W + (size_t)(r0 + r) * K;
        for (int k = 0; k < K; ++k) { s0 = fma(w[k], x0[(size_t)t * K + k], s0); s1 = fma(w[k], nu[(size_t)t * K + k], s1); }
      }
      E0[e] = s0; E1[e] = s1;
    }
    __syncthreads();
    const int nr = min(GASS_RT, a.N - r0);
    for (int q = tid; q < nr * J; q += GASS_THREADS) {
      const int r = q / J, c = q - r * J;
      double aa = 0.0, bb = 0.0;
      if (sparse) {          // the same sums without their zero terms (ascending t: bit-identical for finite predictors)
        for (int e = cptr[c]; e < cptr[c + 1]; ++e) { const int t = cidx[e]; aa = fma(cval[e], E0[r * T + t], aa); bb = fma(cval[e], E1[r * T + t], bb); }
      } else {
        for (int t = 0; t < T; ++t) { aa = fma(Cn[c * T + t], E0[r * T + t], aa); bb = fma(Cn[c * T + t], E1[r * T + t], bb); }
      }
      gass_constraint(aa, bb, a.Cc[c], S.diff, tmin, tmax, any);
    }
  }
  __syncthreads();
  unsigned char* vm = a.vmask + (size_t)j * GASS_GRID;
  const int total = gass_finish_grid(S, tmin, tmax, any, vm, a.info + 2 * j, a.pick != 0, S.diff);
  if (a.pick) gass_pick(total, (any_of(S) & 1) ? 0 : 1, a.ngrid, S.diff, a.thetas + (size_t)j * GASS_MAXC, a.ntheta + j, a.seed, (unsigned long long)j);
}

// ---- candidate likelihoods -------------------------------------------------------------------------------------
// ll[chain][g] = sum over the chain's cells of the Poisson term at eta = cos(theta_g) e0 + sin(theta_g) e1, e0 = the
// cell's linear predictor at the current state, e1 at the proposal.  Lanes = candidates (g = lane, lane + 64), the
// four waves split the cells of a tile; cells staged as (e0, e1, s1, cnt) in LDS.
constexpr int GASS_CT = 1024;     // cells per tile
struct GassEvalArgs {
  const double* X0; const double* Nu; const double* F;   // chains' state / proposal, the fixed factor (V for rows, W for cols)
  const double* A; const unsigned char* C8; const double* Cd; double Rc;    // statistics of the chains' cells, counts (bytes / f64 / constant)
  int N, M, T, K, ld;
  const double* thetas; const int* ntheta; double* ll;   // [nchains][GASS_MAXC] (nsplit == 1) or partial sums [nchains][nsplit][GASS_MAXC]
  int nsplit;                                           // workgroups per chain (blockIdx.y): tiles dealt round-robin
  LikFam lf;                                            // likelihood family and parameter (ESS_LINK_GENERIC)
};

template <int LINK, bool ROWS>
__global__ __launch_bounds__(GASS_THREADS) void gass_eval_kernel(GassEvalArgs a) {
  __shared__ double e0s[GASS_CT], e1s[GASS_CT], s1s[GASS_CT], cns[GASS_CT];
  __shared__ double red[GASS_THREADS / WAVE][GASS_MAXC];
  __shared__ double2 ltab[LOGTAB_N];
  if constexpr (LINK == ESS_LINK_IDENTITY) log_table_build(ltab); else if constexpr (LINK == ESS_LINK_LOG) exp_table_build(ltab);      // (the tile loop's first barrier publishes it)
  const int ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, K = a.K, T = a.T;
  const int nth = a.ntheta[ch];
  const double th0 = lane < nth ? a.thetas[(size_t)ch * GASS_MAXC + lane] : 0.0;
  const double th1 = lane + 64 < nth ? a.thetas[(size_t)ch * GASS_MAXC + lane + 64] : 0.0;
  double c0, s0, c1, s1;
  sincos(th0, &s0, &c0);
  sincos(th1, &s1, &c1);
  double acc0 = 0.0, acc1 = 0.0;
  const int ncell = ROWS ? a.M * T : a.N * T;
  for (int base = (int)blockIdx.y * GASS_CT; base < ncell; base += GASS_CT * a.nsplit) {
    __syncthreads();
    for (int e = tid; e < GASS_CT; e += GASS_THREADS) {
      const int cell = base + e;
      double d0 = 0.0, d1 = 0.0, sv = 0.0, cv = 0.0;
      if (cell < ncell) {
        if constexpr (ROWS) {          // chain = row i; cell = (j,t); statistics A_v[i][cell]
          const double* __restrict__ f = a.F + (size_t)cell * K;
          const double* __restrict__ x = a.X0 + (size_t)ch * K;
          const double* __restrict__ nu = a.Nu + (size_t)ch * K;
          for (int k = 0; k < K; ++k) { d0 = fma(x[k], f[k], d0); d1 = fma(nu[k], f[k], d1); }
          const size_t o = (size_t)ch * a.ld + cell;
          sv = a.A[o];
          cv = a.C8 ? (double)a.C8[o] : (a.Cd ? a.Cd[o] : a.Rc);
        } else {                       // chain = column j; cell = (t, i) with i fastest; statistics A_wT[(j,t)][i]
          const int t = cell / a.N, i = cell - t * a.N;
          const double* __restrict__ f = a.F + (size_t)i * K;
          const double* __restrict__ x = a.X0 + ((size_t)ch * T + t) * K;
          const double* __restrict__ nu = a.Nu + ((size_t)ch * T + t) * K;
          for (int k = 0; k < K; ++k) { d0 = fma(x[k], f[k], d0); d1 = fma(nu[k], f[k], d1); }
          const size_t o = ((size_t)ch * T + t) * a.ld + i;
          sv = a.A[o];
          cv = a.C8 ? (double)a.C8[o] : (a.Cd ? a.Cd[o] : a.Rc);
        }
      }
      e0s[e] = d0; e1s[e] = d1; s1s[e] = sv; cns[e] = cv;
    }
    __syncthreads();
    const int lim = min(GASS_CT, ncell - base);
    for (int e = wave; e < lim; e += GASS_THREADS / WAVE) {
      const double d0 = e0s[e], d1 = e1s[e], sv = s1s[e], cv = cns[e];
      acc0 += poisson_term<LINK>(sv, cv, fma(c0, d0, s0 * d1), ltab, a.lf);
      acc1 += poisson_term<LINK>(sv, cv, fma(c1, d0, s1 * d1), ltab, a.lf);
    }
  }
  red[wave][lane] = acc0;
  red[wave][lane + 64] = acc1;
  __syncthreads();
  if (tid < GASS_MAXC) {
    double s = 0.0;
    for (int w = 0; w < GASS_THREADS / WAVE; ++w) s += red[w][tid];
    if (a.nsplit == 1) a.ll[(size_t)ch * GASS_MAXC + tid] = tid < nth ? s : -INFINITY;
    else a.ll[((size_t)ch * a.nsplit + blockIdx.y) * GASS_MAXC + tid] = s;
  }
}

// ll[ch][q] = sum of the nsplit partial sums of gass_eval_kernel, in order; -inf beyond the chain's candidates
static __global__ void gass_ll_sum_kernel(const double* __restrict__ part, int nsplit, const int* __restrict__ ntheta, int nchains,
                                   double* __restrict__ ll) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nchains * GASS_MAXC) return;
  const int ch = idx / GASS_MAXC, q = idx - ch * GASS_MAXC;
  double s = 0.0;
  for (int u = 0; u < nsplit; ++u) s += part[((size_t)ch * nsplit + u) * GASS_MAXC + q];
  ll[idx] = q < ntheta[ch] ? s : -INFINITY;
}

// slice height of every chain: hh = ll(current) + log u   (u given, or Philox(seed, chain))
static __global__ void gass_slice_kernel(const double* __restrict__ part, int nsum, int nchains, const double* __restrict__ u,
                                  unsigned long long seed, double* __restrict__ hh, double* __restrict__ cur) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchains) return;
  double s = 0.0;
  for (int q = 0; q < nsum; ++q) s += part[(size_t)c * nsum + q];
  double uu;
  if (u) uu = u[c];
  else { uint32_t r[4]; Philox::gen(seed, (uint64_t)c, 0x534c4943ULL, r); uu = u01(r[0], r[1]); }
  cur[c] = s;
  hh[c] = s + log(uu);
}

// one of the candidates above the slice, uniformly (gass.py:121-126); none: the state stays.  x = x0 cos + v sin.
static __global__ __launch_bounds__(GASS_THREADS) void gass_select_kernel(const double* __restrict__ ll, const int* __restrict__ ntheta,
                                                                   const double* __restrict__ thetas, const double* __restrict__ hh,
                                                                   const double* __restrict__ X0, const double* __restrict__ Nu,
                                                                   double* __restrict__ X, int per, unsigned long long seed,
                                                                   int* __restrict__ naccept, double* __restrict__ newll) {
  __shared__ int pick;
  __shared__ double th;
  __shared__ unsigned char above[GASS_MAXC];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int n = ntheta[c];
  if (tid < GASS_MAXC) above[tid] = (tid < n && ll[(size_t)c * GASS_MAXC + tid] >= hh[c]) ? 1 : 0;
  __syncthreads();
  if (tid == 0) {
    int cnt = 0;
    for (int g = 0; g < n; ++g) cnt += above[g];
    pick = -1;
    if (cnt > 0) {
      uint32_t r[4];
      Philox::gen(seed, (uint64_t)c, 0x53454c45ULL, r);
      int want = min(cnt - 1, (int)(u01(r[0], r[1]) * (double)cnt));
      for (int g = 0; g < n; ++g)
        if (above[g] && want-- == 0) { pick = g; break; }
    }
    if (naccept) naccept[c] = cnt;
    if (pick >= 0) { th = thetas[(size_t)c * GASS_MAXC + pick]; if (newll) newll[c] = ll[(size_t)c * GASS_MAXC + pick]; }
  }
  __syncthreads();
  if (pick < 0) return;
  double sn, cs;
  sincos(th, &sn, &cs);
  for (int e = tid; e < per; e += GASS_THREADS) X[(size_t)c * per + e] = fma(X0[(size_t)c * per + e], cs, Nu[(size_t)c * per + e] * sn);
}

}  // namespace btf
