// Exact Polya-Gamma draws for integer shapes  (BTF_K_PG, the default for Binomial trial counts)
//
//   omega ~ PG(b, psi), b = 1, 2, ... : the sum of b independent J*(1, |psi|/2)/4 variates, each by Devroye's
//   alternating-series sampler (Polson, Scott & Windle 2013, sec. 4) - the algorithm pypolyagamma's
//   PyPolyaGamma.pgdrawv runs at factor.py:459.  Nothing is truncated or moment-matched.
//
// How it is laid out for a 64-wide wavefront.  A draw is a short chain of rejection steps (pick the exponential or
// the inverse-Gaussian piece; sample the truncated inverse Gaussian by rejection; accept against the alternating
// series) and a cell needs b of them, so a wave that walks "cell by cell, draw by draw" waits for its unluckiest
// lane at every step (the round-2 kernel: 2.85 ms for 33.5 M draws).  Here every lane owns a LIST of cells and
// the wave runs one flat loop of TRIPS: in a trip each lane makes ONE attempt for the draw it is currently owed
// (three words of its cell's generator, one candidate, all tests), and on success moves its own counters on - to
// the next draw of the cell, or to the next cell of its list.  No lane waits inside a rejection loop; the wave
// stops when every list is empty, and with ~32 draws per lane the trip counts of the lanes differ by ~20 %.
//
// Per-cell constants are computed once for all b draws in a first, branch-free phase (psi in f64, the mixture
// weight of the exponential piece, 1/fz, the cell's generator state) and parked in LDS; a lane that moves to its
// next cell reads two 16-byte records.
//
// Arithmetic.  Candidates and tests are evaluated in f32 with the hardware transcendentals (v_log_f32, v_exp_f32,
// v_rcp_f32, v_cos_f32: ~1 ulp) - as a SQUEEZE: every comparison carries a guard band (PGX_G, 5x the worst f32
// error), and a lane whose comparison lands inside a band repeats that trip in f64 from the same words
// (pgx_trip_f64: libm log / exp / erfc, the full alternating series), so the accept / reject decisions are those
// of the f64 algorithm.  The accepted variate itself is the f32 candidate (relative rounding ~1e-7; up to ~1e-4
// in rare draws of the z > 1/t branch, whose chi-square variate goes through v_cos_f32's absolute error).
// Uniforms: three words of a per-cell xoshiro128++ stream per trip (four for z > 1/t), seeded by one Philox block keyed
// (seed, global cell): the draw of a cell does not depend on layout, launch geometry or sharding.
#pragma once
#include "btf_device.h"

namespace btf {

constexpr double PG_T = 0.64;
constexpr double PG_PI = 3.141592653589793238462643383279502884;

__device__ inline double log_ncdf(double x) {
  if (x > -10.0) return log(0.5 * erfc(-x * 0.70710678118654752440));
  const double x2 = x * x;  // Mills-ratio asymptotics for the far lower tail
  return -0.5 * x2 - log(-x) - 0.91893853320467274178 + log1p(-1.0 / x2 + 3.0 / (x2 * x2));
}

// n-th coefficient of the alternating series for the density of J*(1, 0) at x (both expansions)
__device__ inline double pg_a(int n, double x) {
  const double Kc = (n + 0.5) * PG_PI;
  if (x > PG_T) return Kc * exp(-0.5 * Kc * Kc * x);
  const double r = 2.0 / (PG_PI * x);                       // (2/(pi x))^(3/2) without logarithms
  return Kc * r * sqrt(r) * exp(-2.0 * (n + 0.5) * (n + 0.5) / x);
}

// probability of the exponential (right) piece of the proposal
__device__ inline double pg_mass_texpon(double z) {
  const double fz = 0.125 * PG_PI * PG_PI + 0.5 * z * z;
  const double rt = 1.0 / sqrt(PG_T);
  const double b = rt * (PG_T * z - 1.0), a = -rt * (PG_T * z + 1.0);
  const double x0 = log(fz) + fz * PG_T;
  const double xb = x0 - z + log_ncdf(b), xa = x0 + z + log_ncdf(a);
  const double qdivp = 4.0 / PG_PI * (exp(xb) + exp(xa));
  return 1.0 / (1.0 + qdivp);
}

// ---------------------------------------------------------------------------------------------------------------
constexpr float PGX_T = 0.64f;
constexpr float PGX_RT = 1.5625f;              // 1/t: below, the inverse-Gaussian piece by Devroye's chi-square rejection
constexpr float PGX_G = 3e-5f;                 // relative guard band of the f32 comparisons (worst f32 error of a test: ~6e-6)
constexpr float PGX_GM = 1e-4f;                // ... of the two that go through v_cos_f32 / v_sqrt_f32
constexpr uint32_t PGX_GCOIN = 512;            // ... of the 24-bit coin, in units of 2^-24 (3e-5)
constexpr int PGX_MAX_TRIPS = 1 << 18;         // bound of the flat loop (a list of 8 cells of 199 draws takes ~2100 trips)
constexpr int PGX_MAXB = 255;                  // counts per cell the record holds (the callers stop at PG_NORMAL_B)
constexpr float PGX_LN2 = 0.69314718055994531f, PGX_LOG2E = 1.4426950408889634f;
constexpr float PGX_S24 = 1.0f / 16777216.0f;

struct Xo128 {   // xoshiro128++ (Blackman & Vigna)
  uint32_t s0, s1, s2, s3;
  __device__ __forceinline__ uint32_t next() {
    const uint32_t sum = s0 + s3;
    const uint32_t result = ((sum << 7) | (sum >> 25)) + s0;
    const uint32_t t = s1 << 9;
    s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t;
    s3 = (s3 << 11) | (s3 >> 21);
    return result;
  }
};

// erfcx(y) = exp(y^2) erfc(y), y >= 0, to 3e-7 relative: (1 + 2y) erfcx(y) is smooth and bounded in
// u = (y - 2)/(y + 2) in [-1, 1): degree-10 Chebyshev fit (scripts/pgx_fit.py), Horner in f32
__device__ __forceinline__ float pgx_erfcx(float y) {
  const float u = (y - 2.0f) * __builtin_amdgcn_rcpf(y + 2.0f);
  float p = 1.60803582e-04f;
  p = fmaf(p, u, -1.83654437e-04f);
  p = fmaf(p, u, -1.39570695e-03f);
  p = fmaf(p, u, 1.04314458e-03f);
  p = fmaf(p, u, 8.68915966e-03f);
  p = fmaf(p, u, -7.94272170e-03f);
  p = fmaf(p, u, -5.42119585e-02f);
  p = fmaf(p, u, 1.64035840e-01f);
  p = fmaf(p, u, -1.66031096e-01f);
  p = fmaf(p, u, -9.27630820e-02f);
  p = fmaf(p, u, 1.27697839e+00f);
  return p * __builtin_amdgcn_rcpf(fmaf(2.0f, y, 1.0f));
}

// mass of the exponential piece, f32.  With ya = (t z + 1)/sqrt(2t), yb = |1 - t z|/sqrt(2t) every exponent of
// pg_mass_texpon collapses to the constant pi^2 t/8 - 1/(2t):  q/p = (2 fz C0/pi) [erfcx(ya) + erfcx(yb)]  for
// z <= 1/t, and erfcx(yb) -> 2 exp(yb^2) - erfcx(yb) beyond (no overflow before p underflows; then p = 0).
__device__ __forceinline__ float pgx_mass_texpon(float z) {
  constexpr float C = 0.63661977236758134f * 1.00835305f;          // 2 C0 / pi, C0 = exp(pi^2 t/8 - 1/(2t))
  constexpr float RS = 0.88388347648318441f;                        // 1/sqrt(2t)
  const float fz = fmaf(0.5f * z, z, 1.2337005501361697f);
  const float ya = fmaf(PGX_T, z, 1.0f) * RS, yb = fabsf(fmaf(-PGX_T, z, 1.0f)) * RS;
  const float eb = pgx_erfcx(yb);
  const float S = pgx_erfcx(ya) + (z <= PGX_RT ? eb : 2.0f * __builtin_amdgcn_exp2f(yb * yb * PGX_LOG2E) - eb);
  return __builtin_amdgcn_rcpf(fmaf(C * fz, S, 1.0f));
}

// One lane's current cell.
struct PgxLane {
  Xo128 g;
  float z, rfz, mu, hz2;    // |psi|/2, ln2/fz, 1/z, -z^2 log2(e)/2
  uint32_t thr;             // 24-bit threshold of the piece coin
  int rem;                  // draws the cell is still owed
  bool left;                // the coin fell on the inverse-Gaussian piece and its sampler has not produced a value yet
  float sum;
};

// The record a cell leaves in LDS: generator state, and (z or the value to write for a cell without draws,
// threshold << 8 | count, ln2/fz, 1/z).
__device__ __forceinline__ void pgx_setup(int nb, float fillval, double psi, unsigned long long seed, unsigned long long cell,
                                          uint4& rg, uint4& rp) {
  if (nb > 0 && !(fabs(psi) < 1e30)) { nb = 0; fillval = __builtin_nanf(""); }      // NaN / inf factors: NaN out, no loop
  if (nb <= 0) {
    rg = make_uint4(0, 0, 0, 0);
    rp = make_uint4(__float_as_uint(fillval), 0, 0, 0);
    return;
  }
  uint32_t r[4];
  Philox::gen(seed, cell, 0x70677831ULL << 24, r);          // a counter no other consumer of the cell's key reaches
  rg = make_uint4(r[0], r[1], r[2], r[3] | 1u);             // never the all-zero state
  const float z = (float)(0.5 * fabs(psi));
  const float p = pgx_mass_texpon(z);
  const uint32_t thr = (uint32_t)fmaf(p, 16777216.0f, 0.5f);
  const float fz = fmaf(0.5f * z, z, 1.2337005501361697f);
  rp = make_uint4(__float_as_uint(z), (thr << 8) | (uint32_t)nb, __float_as_uint(PGX_LN2 * __builtin_amdgcn_rcpf(fz)),
                  __float_as_uint(__builtin_amdgcn_rcpf(z)));
}

__device__ __forceinline__ bool pgx_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ULL; }

// The truncated inverse Gaussian of the left piece, z <= 1/t: X = 1/N^2 with N a standard normal beyond a = 1/sqrt(t)
// (the chi-square form of Devroye's proposal), thinned by exp(-z^2 X/2).  N by Robert's (1995) translated-exponential
// rejection, N = a + Exp(1)/lambda, lambda = (a + sqrt(a^2 + 4))/2, accepted with exp(-(N - lambda)^2/2) - and both
// acceptance steps decided by ONE uniform, U <= exp(-(N - lambda)^2/2 - z^2 X/2): 0.89 of the proposals at z = 0
// (Devroye's pair of exponentials: 0.72), one uniform and one logarithm less per trip.
constexpr double PGX_A = 1.25, PGX_LAM = 1.8042476415070754;        // 1/sqrt(0.64), (a + sqrt(a^2 + 4))/2

// The words of a trip: wA, wB, wC (and wD for a cell with z > 1/t), 24 bits each for the piece coin, the exponential
// variate and the acceptance uniform of the left piece, the three low bytes for the uniform of the series test.
// The same trip in f64 from the same words: bit 0 = the piece produced a value, bit 1 = the series accepted it.
struct PgxRedo { float X; uint32_t r; };
__device__ __noinline__ PgxRedo pgx_trip_f64(double psi, int left_in, int small, uint32_t wA, uint32_t wB, uint32_t wC,
                                             uint32_t wD) {
  constexpr double S24 = 1.0 / 16777216.0;
  const double z = 0.5 * fabs(psi);
  bool left = left_in != 0;
  if (!left) left = ((double)(wA >> 8) + 0.5) * S24 >= pg_mass_texpon(z);
  const double E1 = -log(((double)(wB >> 8) + 0.5) * S24);
  double X = PG_T + E1 / (0.125 * PG_PI * PG_PI + 0.5 * z * z);
  bool ok = true;
  if (left) {
    const double u3 = ((double)(wC >> 8) + 0.5) * S24;
    if (small) {
      const double N = PGX_A + E1 / PGX_LAM;
      X = 1.0 / (N * N);
      ok = u3 <= exp(-0.5 * (N - PGX_LAM) * (N - PGX_LAM) - 0.5 * z * z * X);
    } else {
      const double mu = 1.0 / z;
      const double c = cospi(2.0 * (double)(wD >> 8) * S24);
      const double Y = 2.0 * E1 * c * c;                     // a chi-square(1) variate
      const double w = 0.5 * mu * Y;
      const double s = 1.0 + w + sqrt(w * (w + 2.0));
      X = u3 <= s / (s + 1.0) ? mu / s : mu * s;
      ok = X <= PG_T;
    }
  }
  const float Xf = (float)X;
  if (!ok) return PgxRedo{Xf, 0u};
  // alternating series on the deficits D_n = 1 - S_n/a_0 with U' = 1 - U:  odd n accept if U' >= D_n, even n reject if U' < D_n
  const double Us = ((double)(((wA & 0xffu) << 16) | ((wB & 0xffu) << 8) | (wC & 0xffu)) + 0.5) * S24;
  const double a0 = pg_a(0, X);
  double D = 0.0;
  for (int n = 1; n <= 1000; ++n) {
    const double r = pg_a(n, X) / a0;
    if (n & 1) { D += r; if (Us >= D) return PgxRedo{Xf, 3u}; }
    else { D -= r; if (Us < D) return PgxRedo{Xf, 1u}; }
  }
  return PgxRedo{Xf, 3u};
}

// One attempt of every lane.  psi_of(): the f64 psi of the lane's current cell (only evaluated in the fallback).
// ALLF64 (validation): every lane repeats every trip in f64 - the results must equal the squeezed ones up to the
// rounding of the accepted variate, decision by decision (tests/test_gpu_parity.py).
template <bool ALLF64, class PsiFn>
__device__ __forceinline__ void pgx_trip(PgxLane& L, PsiFn psi_of) {
  const uint32_t wA = L.g.next(), wB = L.g.next(), wC = L.g.next();
  const bool active = L.rem > 0;
  const bool small = L.z <= PGX_RT;
  uint32_t wD = 0;
  if (pgx_any(!small)) { if (!small) wD = L.g.next(); }    // (a property of the cell: the stream stays a function of the cell)
  const uint32_t c24 = wA >> 8;
  const bool left = L.left || c24 >= L.thr;
  bool doubt = !L.left && (c24 - L.thr + PGX_GCOIN) <= 2u * PGX_GCOIN;
  const float lB = __builtin_amdgcn_logf(fmaf((float)(wB >> 8), PGX_S24, 0.5f * PGX_S24));     // log2 of a uniform
  float X = fmaf(-lB, L.rfz, PGX_T);                        // right piece: t + Exp(1)/fz
  bool ok = true;
  if (pgx_any(active && left)) {
    const float u3 = fmaf((float)(wC >> 8), PGX_S24, 0.5f * PGX_S24);
    if (pgx_any(active && left && small)) {
      const float N = fmaf(-lB, (float)(0.69314718055994531 / PGX_LAM), (float)PGX_A);
      const float Xs = __builtin_amdgcn_rcpf(N * N);
      const float dn = N - (float)PGX_LAM;
      const float alpha = __builtin_amdgcn_exp2f(fmaf(L.hz2, Xs, -0.5f * PGX_LOG2E * dn * dn));
      if (left && small) {
        X = Xs;
        ok = u3 <= alpha;
        doubt = doubt || fabsf(u3 - alpha) <= PGX_G * alpha;
      }
    }
    if (pgx_any(active && left && !small)) {
      // inverse Gaussian(1/z, 1) by Michael, Schucany & Haas, kept if <= t;  mu/(1 + w + sqrt(w(w+2))) is the
      // cancellation-free form of mu + mu^2 Y/2 - (mu/2) sqrt(4 mu Y + mu^2 Y^2), w = mu Y/2
      const float cs = __builtin_amdgcn_cosf((float)(wD >> 8) * PGX_S24);
      const float Y = -2.0f * PGX_LN2 * lB * cs * cs;
      const float w = 0.5f * L.mu * Y;
      const float s = 1.0f + w + __builtin_amdgcn_sqrtf(w * (w + 2.0f));
      const float pk = s * __builtin_amdgcn_rcpf(s + 1.0f);
      const float Xl = u3 <= pk ? L.mu * __builtin_amdgcn_rcpf(s) : L.mu * s;
      if (left && !small) {
        X = Xl;
        ok = Xl <= PGX_T;
        doubt = doubt || fabsf(u3 - pk) <= PGX_GM * pk || fabsf(Xl - PGX_T) <= PGX_GM * PGX_T;
      }
    }
  }
  // alternating series.  a_1/a_0 = 3 exp(-pi^2 X) on the right, 3 exp(-4/X) on the left, never above 3 exp(-4/t) =
  // 0.00579: a uniform above that accepts without a look at X (99.4 % of the values)
  const float Us = fmaf((float)(((wA & 0xffu) << 16) | ((wB & 0xffu) << 8) | (wC & 0xffu)), PGX_S24, 0.5f * PGX_S24);
  bool acc = Us >= 0.0059f;
  if (pgx_any(active && ok && !acc)) {
    // first term: accept if U' >= a_1/a_0; second: reject if U' < a_1/a_0 - a_2/a_0 (a_2/a_1 < 6e-6); in between: f64
    const float arg = X > PGX_T ? -9.8696044010893586f * PGX_LOG2E * X : -4.0f * PGX_LOG2E * __builtin_amdgcn_rcpf(X);
    const float r1 = 3.0f * __builtin_amdgcn_exp2f(arg);
    const float r2 = 5.0f * __builtin_amdgcn_exp2f(3.0f * arg);
    if (!acc) {
      acc = Us >= r1 * (1.0f + PGX_G);
      if (!acc && !(Us < (r1 - r2) * (1.0f - PGX_G))) doubt = true;
    }
  }
  if (ALLF64) doubt = true;
  if (pgx_any(active && doubt)) {
    if (active && doubt) {
      const PgxRedo d = pgx_trip_f64(psi_of(), L.left ? 1 : 0, small ? 1 : 0, wA, wB, wC, wD);
      X = d.X; ok = (d.r & 1u) != 0u; acc = (d.r & 2u) != 0u;
    }
  }
  if (active) {
    L.left = !ok;
    if (ok && acc) { L.sum += X; --L.rem; }
  }
}

__device__ __forceinline__ void pgx_load(PgxLane& L, const uint4 rg, const uint4 rp) {
  L.g.s0 = rg.x; L.g.s1 = rg.y; L.g.s2 = rg.z; L.g.s3 = rg.w;
  L.z = __uint_as_float(rp.x);
  L.thr = rp.y >> 8;
  L.rem = (int)(rp.y & 0xffu);
  L.rfz = __uint_as_float(rp.z);
  L.mu = __uint_as_float(rp.w);
  L.hz2 = -0.5f * PGX_LOG2E * L.z * L.z;
  L.left = false;
  L.sum = 0.0f;
}

// The flat loop of one lane over its NC cells.  rec_g / rec_p: the lane's records, cell c at [c * rstride];
// out: where the lane's results go, cell c at [c * ostride] - omega, or the fill value of a cell without draws.
template <bool ALLF64 = false, class PsiFn>
__device__ __forceinline__ void pgx_run(const uint4* rec_g, const uint4* rec_p, int rstride, float* out, int ostride, int NC,
                                        PsiFn psi_of) {
  PgxLane L;
  int cur = 0;
  pgx_load(L, rec_g[0], rec_p[0]);
  bool empty = L.rem == 0;                                   // the current cell had no draws to begin with
  for (int it = 0; it < PGX_MAX_TRIPS && pgx_any(cur < NC); ++it) {
#ifdef PGX_PROBE_NOTRIPS
    L.rem = 0;
#endif
    if (pgx_any(L.rem > 0)) pgx_trip<ALLF64>(L, [&]() { return psi_of(cur); });
    if (cur < NC && L.rem == 0) {
      out[cur * ostride] = empty ? L.z : 0.25f * L.sum;
      ++cur;
      if (cur < NC) {
        pgx_load(L, rec_g[cur * rstride], rec_p[cur * rstride]);
        empty = L.rem == 0;
      }
    }
  }
  for (; cur < NC; ++cur) out[cur * ostride] = __builtin_nanf("");      // (only if the cap above was hit)
}

// which cells the flat exact sampler takes
constexpr int PG_NORMAL_B = 200;         // from here: moment-matched normal (pypolyagamma switches at 170)
constexpr int PG_AUTO_EXACT_MAX = 32;    // default mode: integer counts up to here are drawn exactly
enum { PG_MODE_DEFAULT = 0, PG_MODE_EXACT_ALL = 1, PG_MODE_SERIES_ALL = 2 };
enum { PG_CLASS_NONE = 0, PG_CLASS_FLAT = 1, PG_CLASS_SERIES = 2, PG_CLASS_FRAC = 3, PG_CLASS_NORMAL = 4 };
__host__ __device__ inline int pg_class_of(double b, int mode) {
  if (!(b > 0.0)) return PG_CLASS_NONE;
  if (b >= (double)PG_NORMAL_B) return PG_CLASS_NORMAL;
  if (mode == PG_MODE_SERIES_ALL) return PG_CLASS_SERIES;
  const bool integer = b == floor(b);
  if (mode == PG_MODE_EXACT_ALL) return integer ? PG_CLASS_FLAT : PG_CLASS_FRAC;
  return integer && b <= (double)PG_AUTO_EXACT_MAX ? PG_CLASS_FLAT : PG_CLASS_SERIES;
}

// ---- kernels ---------------------------------------------------------------------------------------------------
constexpr int PGX_NW = 4, PGX_CPL = 4;    // the shape the library launches: 4 waves, lists of 4 cells (37 KB of LDS: 4 workgroups / CU)
// dynamic LDS of the three kernels (bytes)
constexpr size_t pgx_tile_lds(int NW, int CPL) { return (size_t)2 * CPL * NW * 64 * 16 + (size_t)NW * CPL * 65 * 4; }
constexpr size_t pgx_rows_lds(int CPL) { return (size_t)2 * CPL * 256 * 16 + (size_t)CPL * 256 * 4; }

// Unsharded context: a workgroup of NW waves takes a tile of NW*CPL rows x 64 (column, depth) pairs; thread (wave g,
// lane c) owns rows g, g + NW, ... of column c.  The tile of results goes through LDS to both layouts
// (V layout rows, W layout transposed), as pg_tile_kernel.
template <int K, int NW, int CPL>
__global__ __launch_bounds__(NW * 64) void pgx_tile_kernel(const double* __restrict__ Bv, double* __restrict__ Cv,
                                                           double* __restrict__ CwT, const double* __restrict__ W,
                                                           const double* __restrict__ V, int N, int MT, int ldv, int ldw,
                                                           unsigned long long seed, int mode, int fill) {
  constexpr int NT = NW * 64, TI = NW * CPL;
  extern __shared__ uint4 pgx_lds[];
  uint4 (*rec_g)[NT] = reinterpret_cast<uint4 (*)[NT]>(pgx_lds);
  uint4 (*rec_p)[NT] = reinterpret_cast<uint4 (*)[NT]>(pgx_lds + CPL * NT);
  float (*tile)[65] = reinterpret_cast<float (*)[65]>(pgx_lds + 2 * CPL * NT);
  const int tid = threadIdx.x, col = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jt0 = blockIdx.x * 64, i0 = blockIdx.y * TI;
  const int jt = jt0 + col;
  const bool vc = jt < MT;
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = vc ? V[(size_t)jt * K + k] : 0.0;
  auto psi_at = [&](int c) -> double {
    const double* __restrict__ w = W + (size_t)(i0 + g + NW * c) * K;
    double psi = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) psi = fma(w[k], v[k], psi);
    return psi;
  };
#pragma unroll 2
  for (int c = 0; c < CPL; ++c) {
    const int i = i0 + g + NW * c;                          // wave-uniform
    int nb = 0;
    float fillval = -1.0f;                                  // < 0: not this launch's cell
    double psi = 0.0;
    if (i < N && vc) {
      const double b = Bv[(size_t)i * ldv + jt];
      const int cls = pg_class_of(b, mode);
      if (cls == PG_CLASS_FLAT) { nb = (int)b; psi = psi_at(c); }
      else if (cls == PG_CLASS_NONE && fill) fillval = 0.0f;
    }
    pgx_setup(nb, fillval, psi, seed, (unsigned long long)i * MT + jt, rec_g[c][tid], rec_p[c][tid]);
  }
  pgx_run(&rec_g[0][tid], &rec_p[0][tid], NT, &tile[g][col], NW * 65, CPL, psi_at);
  __syncthreads();
  for (int r = g; r < TI; r += NW) {                        // V layout: a row of the tile per wave
    const int i = i0 + r;
    const float om = tile[r][col];
    if (i < N && vc && !(om < 0.0f)) Cv[(size_t)i * ldv + jt] = (double)om;
  }
  const int ii = tid % TI, i = i0 + ii;                     // W layout: lanes along the rows
  for (int c = tid / TI; c < 64; c += NT / TI) {
    const int jj = jt0 + c;
    const float om = tile[ii][c];
    if (jj < MT && i < N && !(om < 0.0f)) CwT[(size_t)jj * ldw + i] = (double)om;
  }
}

// Sharded context, one layout per launch (as pg_kernel): out[r][l] = PG(B[r][l], L[l] . U[r]), lanes along l,
// thread l owns rows r0 .. r1 of its block in chunks of CPL.
template <int K, int CPL>
__global__ __launch_bounds__(256) void pgx_kernel(const double* __restrict__ B, double* __restrict__ out,
                                                  const double* __restrict__ Lf, const double* __restrict__ Uf, int nl,
                                                  int ld, int Rdim, int rows_per_block, unsigned long long base,
                                                  unsigned long long stride_r, unsigned long long stride_l,
                                                  unsigned long long seed, int mode, int fill) {
  extern __shared__ uint4 pgx_lds[];
  uint4 (*rec_g)[256] = reinterpret_cast<uint4 (*)[256]>(pgx_lds);
  uint4 (*rec_p)[256] = reinterpret_cast<uint4 (*)[256]>(pgx_lds + CPL * 256);
  float (*res)[256] = reinterpret_cast<float (*)[256]>(pgx_lds + 2 * CPL * 256);
  const int tid = threadIdx.x;
  const int l = blockIdx.x * 256 + tid;
  const bool lc = l < nl;
  double f[K];
#pragma unroll
  for (int k = 0; k < K; ++k) f[k] = lc ? Lf[(size_t)l * K + k] : 0.0;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, Rdim);
  for (int rb = r0; rb < r1; rb += CPL) {
    auto psi_at = [&](int c) -> double {
      const double* __restrict__ u = Uf + (size_t)(rb + c) * K;
      double psi = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) psi = fma(f[k], u[k], psi);
      return psi;
    };
#pragma unroll 2
    for (int c = 0; c < CPL; ++c) {
      const int r = rb + c;
      int nb = 0;
      float fillval = -1.0f;
      double psi = 0.0;
      if (r < r1 && lc) {
        const double b = B[(size_t)r * ld + l];
        const int cls = pg_class_of(b, mode);
        if (cls == PG_CLASS_FLAT) { nb = (int)b; psi = psi_at(c); }
        else if (cls == PG_CLASS_NONE && fill) fillval = 0.0f;
      }
      pgx_setup(nb, fillval, psi, seed, base + (unsigned long long)r * stride_r + (unsigned long long)l * stride_l,
                rec_g[c][tid], rec_p[c][tid]);
    }
    pgx_run(&rec_g[0][tid], &rec_p[0][tid], 256, &res[0][tid], 256, CPL, psi_at);
    for (int c = 0; c < CPL; ++c) {                         // (own results: no barrier)
      const int r = rb + c;
      const float om = res[c][tid];
      if (r < r1 && lc && !(om < 0.0f)) out[(size_t)r * ld + l] = (double)om;
    }
  }
}

// stand-alone batch (validation entry point): element i of (b, psi) with the stream (seed, i); thread t of a block
// owns elements blockbase + t + 256 c.  Only the elements of PG_CLASS_FLAT under `mode` are drawn (and written).
template <int CPL, bool ALLF64>
__global__ __launch_bounds__(256) void pgx_batch_kernel(const double* __restrict__ b, const double* __restrict__ psi,
                                                        double* __restrict__ out, long long n, unsigned long long seed,
                                                        int mode) {
  extern __shared__ uint4 pgx_lds[];
  uint4 (*rec_g)[256] = reinterpret_cast<uint4 (*)[256]>(pgx_lds);
  uint4 (*rec_p)[256] = reinterpret_cast<uint4 (*)[256]>(pgx_lds + CPL * 256);
  float (*res)[256] = reinterpret_cast<float (*)[256]>(pgx_lds + 2 * CPL * 256);
  const int tid = threadIdx.x;
  const long long e0 = (long long)blockIdx.x * 256 * CPL + tid;
  for (int c = 0; c < CPL; ++c) {
    const long long e = e0 + 256LL * c;
    int nb = 0;
    double ps = 0.0;
    if (e < n && pg_class_of(b[e], mode) == PG_CLASS_FLAT) { nb = (int)b[e]; ps = psi[e]; }
    pgx_setup(nb, -1.0f, ps, seed, (unsigned long long)e, rec_g[c][tid], rec_p[c][tid]);
  }
  pgx_run<ALLF64>(&rec_g[0][tid], &rec_p[0][tid], 256, &res[0][tid], 256, CPL, [&](int c) { return psi[e0 + 256LL * c]; });
  for (int c = 0; c < CPL; ++c) {
    const long long e = e0 + 256LL * c;
    const float om = res[c][tid];
    if (e < n && !(om < 0.0f)) out[e] = (double)om;
  }
}

}  // namespace btf
