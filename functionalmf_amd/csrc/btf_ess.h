// Elliptical slice sampling for non-conjugate likelihoods.
//
// Reference: NonconjugateBayesianTensorFiltering._resample_W / _resample_V (factor.py:567-590) draw a prior
// sample nu (sample_mvn_from_precision on the packed prior precision, factor.py:155-195) and call
// elliptical_slice_ (elliptical_slice.py:59-124): slice height hh = ll(x) + log u, an angle theta uniform on a
// bracket that shrinks towards 0 until  ll(x cos theta + nu sin theta) >= hh.  The likelihood there is a Python
// callback over the whole tensor; here it is a device likelihood (Poisson counts with log or identity link, from
// the same hoisted statistics S1 = sum_r y, cnt the Gaussian model streams), so one evaluation is one streaming
// pass:  sum_cells  S1 log(lambda) - cnt lambda   (the term - sum lgamma(y+1) does not depend on the state).
//
// Two chain layouts:
//   joint   ONE slice over all of W (resp. all of V), as the reference does - with host-drawn uniforms the chain
//           walks the reference's path (fixture tests/golden/g9_ess_*.npz);
//   rows    one slice per row of W (resp. per column of V): given V the rows are conditionally independent, so
//           every row runs its own bracket; all brackets shrink in lockstep, one proposal per row per launch,
//           finished rows drop out.  This is the structure of the reference's constrained model
//           (ConstrainedNonconjugateBTF._resample_W_i, factor.py:665-720) with GASS replaced by the slice loop.
// The shrink loop runs on the device: evaluation and decision kernels are queued for a fixed number of rounds and
// leave at once when their chain is done; nothing is read back.
#pragma once
#include "btf_kernels.h"

namespace btf {

// Likelihood families of the slice samplers, all functions of the hoisted statistics (S1 = sum_r y, cnt = observed
// replicates) and the linear predictor eta = w.v; the state-independent normalising terms are left to the caller:
//   0 Poisson, log link       S1 eta - cnt exp(eta)
//   1 Poisson, identity link  S1 log(eta) - cnt eta                    (-inf where eta <= 0)
//   2 Bernoulli / Binomial, logit link (S1 successes of cnt trials)   S1 eta - cnt softplus(eta)
//   3 Gaussian, identity link, known variance (par = 1 / variance)    par (S1 eta - cnt eta^2 / 2)
//   4 Negative-Binomial, logit link, known rate (par = r)             S1 eta - (S1 + cnt r) softplus(eta)
// Families 0 and 1 have kernels of their own (table-based exp / log); 2..4 share the ESS_LINK_GENERIC instantiation,
// which takes the family and its parameter at run time (libm exp / log1p: one evaluation is still one pass).
enum { ESS_LINK_LOG = 0, ESS_LINK_IDENTITY = 1, ESS_LINK_GENERIC = 2 };
enum { ESS_FAM_POISSON_LOG = 0, ESS_FAM_POISSON_IDENTITY = 1, ESS_FAM_BERNOULLI_LOGIT = 2, ESS_FAM_GAUSSIAN = 3, ESS_FAM_NEGBIN_LOGIT = 4,
       ESS_FAM_COUNT = 5 };
struct LikFam { int fam; double par; };
__host__ __device__ constexpr int ess_link_of(int fam) { return fam <= ESS_FAM_POISSON_IDENTITY ? fam : ESS_LINK_GENERIC; }
__device__ __forceinline__ double softplus(double x) { return x > 0.0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
constexpr int ESS_THREADS = 256;

// prior draw of W: N(0, sigma2) on the free entries (lower triangle of the leading K rows, everything below),
// z indexed as factor.py:155-174 packs them (= the W half-sweep's normal stream)
static __global__ void ess_w_prior_kernel(double* __restrict__ nu, int N, int K, double sigma2, const double* __restrict__ hyp,
                                   const double* __restrict__ z, unsigned long long seed, unsigned long long stream) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * K) return;
  if (hyp) sigma2 = hyp[HYP_SIGMA2];
  const int i = e / K, k = e - i * K;
  const int d = i + 1 < K ? i + 1 : K;
  double v = 0.0;
  if (k < d) {
    const long long zo = w_z_offset(i, K) + k;
    v = sqrt(sigma2) * (z ? z[zo] : philox_normal(seed, stream, (unsigned long long)zo));
  }
  nu[e] = v;
}

// x = x0 cos(theta_c) + nu sin(theta_c) for the chain c of element e (elements per chain: `per`; joint: one chain).
// restore != 0: chains that never finished fall back to x0 (the current state is always on the slice).
static __global__ void ess_combine_kernel(const double* __restrict__ x0, const double* __restrict__ nu, double* __restrict__ x,
                                   long long n, int per, const double* __restrict__ theta, const int* __restrict__ done,
                                   int restore) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int c = per > 0 ? (int)(e / per) : 0;
  if (done[c]) return;
  if (restore) { x[e] = x0[e]; return; }
  double sn, cs;
  sincos(theta[c], &sn, &cs);
  x[e] = fma(x0[e], cs, nu[e] * sn);
}

// (log_tab / exp_tab: table-based double-precision log and exp, btf_device.h)
template <int LINK>
__device__ __forceinline__ double poisson_term(double s1, double cnt, double eta, const double2* __restrict__ ltab, LikFam lf = LikFam{0, 0.0}) {
  if (!(cnt > 0.0)) return 0.0;
  if constexpr (LINK == ESS_LINK_LOG) return fma(s1, eta, -cnt * exp_tab(eta, ltab));
  else if constexpr (LINK == ESS_LINK_IDENTITY) return eta > 0.0 ? fma(s1, log_tab(eta, ltab), -cnt * eta) : -INFINITY;
  else {
    if (lf.fam == ESS_FAM_GAUSSIAN) return lf.par * eta * fma(-0.5 * cnt, eta, s1);
    const double sp = softplus(eta);
    return lf.fam == ESS_FAM_BERNOULLI_LOGIT ? fma(s1, eta, -cnt * sp) : fma(s1, eta, -fma(cnt, lf.par, s1) * sp);
  }
}

// Poisson log-likelihood of the local rows, lanes along (j,t) (V layout): part[i][bx] = sum over the block's cells of
// row i.  CT: double weights, unsigned char replicate counts, or (Cx == nullptr) the constant Rc.
template <int K, int LINK, typename CT>
__global__ __launch_bounds__(ESS_THREADS) void poisson_ll_rows_kernel(
    const double* __restrict__ A, const CT* __restrict__ Cx, double Rc, const double* __restrict__ W,
    const double* __restrict__ V, int row0, int ncols, int ld, size_t col0, const int* __restrict__ done, int per_row,
    double* __restrict__ part, LikFam lf) {
  __shared__ double red[ESS_THREADS / WAVE];
  __shared__ double2 ltab[LOGTAB_N];
  const int i = blockIdx.y;
  if (per_row && done[i]) return;
  if (!per_row && done[0]) return;
  if constexpr (LINK == ESS_LINK_IDENTITY) log_table_build(ltab); else if constexpr (LINK == ESS_LINK_LOG) exp_table_build(ltab);
  __syncthreads();
  double w[K];
#pragma unroll
  for (int k = 0; k < K; ++k) w[k] = W[(size_t)(row0 + i) * K + k];
  double s = 0.0;
  for (int l = blockIdx.x * ESS_THREADS + threadIdx.x; l < ncols; l += gridDim.x * ESS_THREADS) {
    const double* __restrict__ v = V + (col0 + l) * K;
    double eta = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) eta = fma(w[k], v[k], eta);
    const double s1 = A[(size_t)i * ld + l];
    const double cnt = Cx ? (double)Cx[(size_t)i * ld + l] : Rc;
    s += poisson_term<LINK>(s1, cnt, eta, ltab, lf);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int q = 0; q < ESS_THREADS / WAVE; ++q) t += red[q];
    part[(size_t)i * gridDim.x + blockIdx.x] = t;
  }
}

// Poisson log-likelihood of the local columns, lanes along i (W layout, A_wT[jt][ldw]): part[j][bx] = sum over the
// block's rows i and all depths of column j.
template <int K, int LINK, typename CT>
__global__ __launch_bounds__(ESS_THREADS) void poisson_ll_cols_kernel(
    const double* __restrict__ A, const CT* __restrict__ Cx, double Rc, const double* __restrict__ W,
    const double* __restrict__ V, int row0, int nl, int ld, int col0, int T, const int* __restrict__ done,
    double* __restrict__ part, LikFam lf) {
  __shared__ double red[ESS_THREADS / WAVE];
  __shared__ double2 ltab[LOGTAB_N];
  const int j = blockIdx.y;
  if (done[j]) return;
  if constexpr (LINK == ESS_LINK_IDENTITY) log_table_build(ltab); else if constexpr (LINK == ESS_LINK_LOG) exp_table_build(ltab);
  __syncthreads();
  double s = 0.0;
  for (int i = blockIdx.x * ESS_THREADS + threadIdx.x; i < nl; i += gridDim.x * ESS_THREADS) {
    double w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) w[k] = W[(size_t)(row0 + i) * K + k];
    for (int t = 0; t < T; ++t) {
      const double* __restrict__ v = V + ((size_t)(col0 + j) * T + t) * K;     // wave-uniform: scalar loads
      double eta = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) eta = fma(w[k], v[k], eta);
      const size_t o = ((size_t)j * T + t) * ld + i;
      const double cnt = Cx ? (double)Cx[o] : Rc;
      s += poisson_term<LINK>(A[o], cnt, eta, ltab, lf);
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int q = 0; q < ESS_THREADS / WAVE; ++q) t += red[q];
    part[(size_t)j * gridDim.x + blockIdx.x] = t;
  }
}

// One decision per chain (elliptical_slice.py:85-122).  st[c] = {hh, lo, hi, theta, ll}.
//   round < 0: part holds ll(current state): hh = ll + log u, theta ~ U(0, 2 pi), bracket (theta - 2 pi, theta)
//   round >= 0: part holds ll(proposal): on the slice -> done; else shrink the bracket to theta and redraw.
// nsum: partials per chain (joint chains: all of them).  Uniforms: Philox (seed, chain), two per round.
static __global__ __launch_bounds__(ESS_THREADS) void ess_decide_kernel(const double* __restrict__ part, int nsum, int nchains,
                                                                 double* __restrict__ st, double* __restrict__ theta,
                                                                 int* __restrict__ done, int round,
                                                                 unsigned long long seed) {
  __shared__ double red[ESS_THREADS];
  const int c = blockIdx.x;
  if (c >= nchains) return;
  if (round >= 0 && done[c]) return;
  // fixed-order sum: thread q takes a contiguous slice, then a tree over the threads
  const int per = (nsum + ESS_THREADS - 1) / ESS_THREADS;
  double s = 0.0;
  for (int q = threadIdx.x * per; q < min(nsum, (threadIdx.x + 1) * per); ++q) s += part[(size_t)c * nsum + q];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = ESS_THREADS / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double ll = red[0];
  double* S = st + (size_t)c * 5;
  uint32_t r[4];
  Philox::gen(seed, (uint64_t)c, (uint64_t)(round + 1), r);
  const double u1 = u01(r[0], r[1]), u2 = u01(r[2], r[3]);
  const double two_pi = 6.283185307179586476925286766559;
  if (round < 0) {
    S[0] = ll + log(u1);
    const double th = u2 * two_pi;
    S[1] = th - two_pi; S[2] = th; S[3] = th; S[4] = ll;
    theta[c] = th;
    done[c] = 0;
    return;
  }
  if (ll >= S[0]) { S[4] = ll; done[c] = 1; return; }
  const double th = S[3];
  if (th > 0.0) S[2] = th;
  else if (th < 0.0) S[1] = th;
  else { done[c] = 1; return; }              // shrunk to the current state (elliptical_slice.py:112-116)
  const double nt = u1 * (S[2] - S[1]) + S[1];
  S[3] = nt;
  theta[c] = nt;
}

}  // namespace btf
