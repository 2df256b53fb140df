// Device-side helpers shared by the BTF kernels (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace btf {

constexpr int WAVE = 64;

__host__ __device__ constexpr int tri(int k) { return k * (k + 1) / 2; }
// packed lower-triangular index of (a,b), a >= b
__host__ __device__ constexpr int lidx(int a, int b) { return a * (a + 1) / 2 + b; }

// ---------------------------------------------------------------- wave ops
__device__ __forceinline__ double bcast_lane(double v, int lane) {
  // wave-uniform lane index -> v_readlane_b32 x2 (no LDS traffic)
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast_first(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}


// ---------------------------------------------------------------- sc1 (write-through / L1-bypassing) accesses
// 8-byte relaxed agent-scope atomics: global_store_dwordx2 ... sc1 / global_load_dwordx2 ... sc1.  The payload of every
// in-launch hand-off between workgroups goes through these on BOTH sides (btf_fused.h states the protocol).
// (explicitly GLOBAL address space: a generic pointer lowers to flat_ instructions, which the measured hand-offs do not cover)
typedef __attribute__((address_space(1))) unsigned long long gu64_t;
typedef __attribute__((address_space(1))) unsigned gu32_t;
__device__ __forceinline__ void store_sc1(double* p, double v) {
  __hip_atomic_store((gu64_t*)reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_sc1(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const gu64_t*)reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT));
}
// every storing wave, after its sc1 stores and before the barrier in front of the flag store / ticket add
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void publish_epoch(unsigned* flag, unsigned epoch) {     // ONE lane, behind the drain (+ barrier)
  __hip_atomic_store((gu32_t*)flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------- streaming (non-temporal) loads
// global_load_dwordx4 ... nt: read-once data that does not fit the 256 MB Infinity Cache should not be allocated there
// (C5: the 2.1 GB statistic streams at 6.87 TB/s with the hint, 6.59 without); data that does fit must NOT carry it
// (C3: 67 MB per launch, both layouts resident: 10.7 us per launch, 13.4 with the hint - every read goes to HBM).
typedef double v2d_stream __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 stream_load2(const double* p) {
  const v2d_stream v = __builtin_nontemporal_load(reinterpret_cast<const v2d_stream*>(p));
  return make_double2(v.x, v.y);
}

// ---------------------------------------------------------------- Philox4x32-10
// Counter-based generator (Salmon et al. 2011): draws are a pure function of
// (seed, stream, index), so results do not depend on launch geometry or on how
// rows / columns are sharded over GPUs.
struct Philox {
  static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  static constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  __host__ __device__ static inline void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    uint64_t p0 = (uint64_t)M0 * c[0];
    uint64_t p1 = (uint64_t)M1 * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  __host__ __device__ static inline void gen(uint64_t seed, uint64_t stream, uint64_t index, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      round(c, k0, k1);
      k0 += W0; k1 += W1;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
  }
};

// uniform in (0,1) from 64 random bits (53-bit mantissa, never 0 or 1)
__host__ __device__ inline double u01(uint32_t lo, uint32_t hi) {
  uint64_t x = ((uint64_t)hi << 32) | lo;
  return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

// Standard normal number `index` of stream `stream` (Box-Muller on one Philox block;
// even/odd indices share a block).
__device__ inline double philox_normal(uint64_t seed, uint64_t stream, uint64_t index) {
  uint32_t r[4];
  Philox::gen(seed, stream, index >> 1, r);
  double u1 = u01(r[0], r[1]);
  double u2 = u01(r[2], r[3]);
  double rad = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  return (index & 1) ? rad * s : rad * c;
}

// Both normals of Philox block `pair` (indices 2 pair, 2 pair + 1 of the stream) for the price of one: the same
// operations on the same values as two philox_normal calls - bit-identical results.
__device__ inline void philox_normal_pair(uint64_t seed, uint64_t stream, uint64_t pair, double& z0, double& z1) {
  uint32_t r[4];
  Philox::gen(seed, stream, pair, r);
  double u1 = u01(r[0], r[1]);
  double u2 = u01(r[2], r[3]);
  double rad = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  z0 = rad * c;
  z1 = rad * s;
}

// offset of row i's normals in the flat W-step stream: sum_{i'<i} min(i'+1, K)
__host__ __device__ inline long long w_z_offset(int i, int K) {
  return i < K ? (long long)i * (i + 1) / 2 : (long long)K * (K + 1) / 2 + (long long)(i - K) * K;
}

// log(x) for the identity-link likelihood, evaluated once per (cell, proposal): libm's log is ~90 f64 instructions;
// here x = 2^e m, m in [1,2) is divided by the left edge c_i = 1 + i/128 of its mantissa interval through a
// 128-entry table (1/c_i rounded, and minus the log of exactly that number), leaving log1p(r) with 0 <= r < 2^-7 for a
// degree-8 polynomial (truncation r^8/9 < 2e-18 relative): ~20 instructions, absolute error of the order of one
// ulp of max(|e| ln 2, 1) - what libm gives away from x = 1.  The table is built per workgroup in LDS (one libm log
// per thread): log_table_build, then a barrier.
constexpr int LOGTAB_N = 128;
__device__ __forceinline__ void log_table_build(double2* tab) {
  for (int i = threadIdx.x; i < LOGTAB_N; i += blockDim.x) {
    const double inv = 1.0 / (1.0 + (double)i * (1.0 / LOGTAB_N));
    tab[i] = make_double2(inv, -log(inv));
  }
}
__device__ __forceinline__ double log_tab(double x, const double2* __restrict__ tab) {
  if (!(x >= 2.2250738585072014e-308 && x < INFINITY)) return log(x);        // subnormal / inf / nan: rare, exact path
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  const int e = (int)(b >> 52) - 1023;
  const double m = __longlong_as_double((long long)((b & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL));
  const double2 t = tab[(int)((b >> 45) & 127ULL)];
  const double r = fma(m, t.x, -1.0);
  double p = fma(r, -1.0 / 8.0, 1.0 / 7.0);
  p = fma(r, p, -1.0 / 6.0);
  p = fma(r, p, 1.0 / 5.0);
  p = fma(r, p, -1.0 / 4.0);
  p = fma(r, p, 1.0 / 3.0);
  p = fma(r, p, -1.0 / 2.0);
  p = fma(r * r, p, r);                                   // log1p(r) = r - r^2/2 + ... - r^8/8
  return fma((double)e, 0.6931471805599453094, t.y + p);
}

// exp(x) for the log link, the same way: x = (128 k + j) ln2/128 + r, |r| <= ln2/256, so exp(x) = 2^k 2^(j/128) e^r
// with 2^(j/128) from the table (second use of its 128 slots: .x of the exp table) and a degree-5 polynomial for e^r
// (truncation r^6/720 < 6e-19); the reduction subtracts n ln2/128 in two pieces (hi with 11 trailing zero bits: exact
// product for |n| < 2^11 * 128).  ~20 instructions instead of libm's ~55; |x| > 700 (overflow range) goes to libm.
__device__ __forceinline__ void exp_table_build(double2* tab) {
  for (int i = threadIdx.x; i < LOGTAB_N; i += blockDim.x) tab[i] = make_double2(exp2((double)i * (1.0 / LOGTAB_N)), 0.0);
}
__device__ __forceinline__ double exp_tab(double x, const double2* __restrict__ tab) {
  if (!(fabs(x) < 700.0)) return exp(x);
  const double n = rint(x * (LOGTAB_N * 1.4426950408889634074));            // x * 128 / ln 2
  const double hi = 0x1.62e42fefa3800p-8, lo = 0x1.ef35793c76730p-52;       // ln2/128 = hi + lo, hi: 42 significant bits
  const double r = fma(-n, lo, fma(-n, hi, x));
  const int ni = (int)n;
  const int j = ni & (LOGTAB_N - 1), k = (ni - j) / LOGTAB_N;               // ni = 128 k + j, 0 <= j < 128
  double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  p = fma(r, p, 1.0 / 6.0);
  p = fma(r, p, 0.5);
  p = fma(r * r, p, r);                                                      // e^r - 1
  const double t = tab[j].x;
  return ldexp(fma(t, p, t), k);
}


}  // namespace btf
