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

// offset of row i's normals in the flat W-step stream: sum_{i'<i} min(i'+1, K)
__host__ __device__ inline long long w_z_offset(int i, int K) {
  return i < K ? (long long)i * (i + 1) / 2 : (long long)K * (K + 1) / 2 + (long long)(i - K) * K;
}

}  // namespace btf
