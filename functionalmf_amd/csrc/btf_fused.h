// Latency kernels as TAILS of the streaming launches (BTF_OPT_FUSED_STEP): the kernel that follows each streaming
// accumulation - w_solve_kernel behind the W accumulation, v_spectral_kernel behind the V accumulation - runs inside the
// accumulation launch itself.  1 (the default): the V launch only; 2: both (two launches per W+V step).
//
//   V launch  accum_kernel<K, 0, 16, ..., FUSE_V>: a 128-output tile of the (column, depth) axis is 128 / T whole columns.
//             With one chunk (C3: the workgroup streams all rows of W) the column sums never leave the workgroup: they
//             go from the cross-wave reduction straight into the spectral sampler's LDS layout; with several chunks the
//             last arriver of the tile (ticket) sums the others' write-through partials.  The prior band of the columns
//             comes precomputed (prior_band_kernel, rebuilt when Tau2 / lam2 changed): one double per thread loaded at
//             kernel start, parked in registers through the stream (v_fused_band_preload / _store).  The eigen-system of
//             W'W is still solved by the side workgroup of the same launch; it is published write-through with a flag
//             the tails poll (once, by one lane, after their stream has ended).  Then the 2K elimination chains of every
//             column of the tile run as in v_spectral_kernel<S, false, K> - a four-wave "virtual workgroup" per column,
//             the chain waves of the columns on different SIMDs - with the same arithmetic in the same order:
//             bit-identical draws.  Measured at C3: 20.8 us against 10.6 + 0.8 + 12.1 us for the two launches.
//   W launch  accum_kernel<K, 0, 16, ..., FUSE_W>: the streaming workgroups of a 128-row tile write their chunk sums
//             write-through (sc1) and add to the tile's counter; dedicated OWNER workgroups (32 rows each; they stream
//             nothing and have prepared normals, Gram and scalars meanwhile) poll the counter, fetch the chunk sums, add
//             them per (row, value) pair in chunk order - w_solve_kernel's canonical order - factor and draw their rows
//             (factor.py:349-362) and leave the W'W shares of their virtual w_solve workgroups: bit-identical draws.
//             Measured at C3: 18.9 us against 10.7 + 0.8 + 4.7 us - the write-through partials of 32 workgroups have to
//             drain before the counter may move (3.6 us) and be fetched (2.8 us): an option, not the default.
//
// Hand-off protocol (MI355X_MICROARCH.md, workgroup dispatch / inter-workgroup visibility, first row of the table of
// measured sc1 hand-offs): every handed-off byte is stored sc1 (8-byte relaxed agent-scope atomic stores), every storing
// wave drains its stores (s_waitcnt vmcnt(0)) in front of a workgroup barrier, ONE lane then adds to the counter /
// stores the flag, and every load of those bytes is an sc1 load (8-byte relaxed agent-scope atomic loads) issued behind
// the returned ticket / the matched poll and a workgroup barrier.  No fences.  The counters are zeroed when they are
// allocated and reset by the last arriver (the next launch starts behind this one on the stream); the flags carry an
// epoch that the host increments per launch and never reuses.
#pragma once
#include "btf_kernels.h"
#include "btf_spectral.h"

namespace btf {

// One lane polls one word until it holds `epoch` (relaxed sc1 loads, s_sleep between them); bounded: a producer that
// never shows up (it is block 0 / 1 of the same launch and depends on nothing this workgroup does) ends the wait after
// ~0.2 s with `false`, and the caller reports BTF_EHIP through the status words instead of hanging the GPU.
__device__ __forceinline__ bool poll_flag(const unsigned* flag, unsigned epoch) {
  for (unsigned spins = 0; spins < (1u << 22); ++spins) {
    if (__hip_atomic_load((const gu32_t*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) return true;
    __builtin_amdgcn_s_sleep(8);
  }
  return false;
}

// scalars drawn by a side workgroup of THIS launch (the four-launch sweep's nu2 / sigma2 in the W launch, lam2 in the V
// launch) reach the tails through a published copy and a flag; scalars of earlier launches are read from `hyp` itself
struct HypPub { const double* pub; const unsigned* flag; unsigned epoch; int which; };   // which: bit 0 nu2, bit 1 sigma2, bit 2 lam2

// =================================================================================================================
// W launch: the tiles' owner workgroups
// =================================================================================================================
struct FuseW {
  WSolveArgs a;            // what launch_wsolve would have been handed (a.part: this launch's partials)
  unsigned* cnt;           // [tiles] arrivals of the tiles' streaming workgroups: running totals, never reset
  unsigned expected;       // ... the value a tile's counter holds when its nch workgroups of THIS launch have all arrived
  int owners;              // owner workgroups in front of the grid: tiles x (128 / rows)
  int rows;                // rows per owner: 128, 64, ... (>= rw)
  int rw;                  // rows per virtual w_solve workgroup: ws_rows_for(nl)
  HypPub hp;               // flag == nullptr: no scalar side workgroup in this launch
  const unsigned* gram_cnt; unsigned gram_expected;   // the Gram partials are made by side workgroups of this launch (GramSide.cnt), else nullptr
};
template <> __device__ __forceinline__ unsigned* fuse_tickets<FUSE_W>(const FuseW& fz) { return fz.cnt; }
template <> __device__ __forceinline__ int fuse_chunks<FUSE_W>(const FuseW& fz) { return fz.a.nch; }
template <> __device__ __forceinline__ int fuse_owners<FUSE_W>(const FuseW& fz) { return fz.owners; }

// several flags at once: lane l < 3 of the calling wave polls word l until it holds its value (nullptr: nothing to wait
// for); returns (to every lane) whether all of them showed up before the bound
__device__ __forceinline__ bool poll_flags(const unsigned* f0, unsigned e0, const unsigned* f1, unsigned e1, const unsigned* f2, unsigned e2) {
  const int lane = threadIdx.x & 63;
  const unsigned* f = lane == 0 ? f0 : (lane == 1 ? f1 : (lane == 2 ? f2 : nullptr));
  const unsigned e = lane == 0 ? e0 : (lane == 1 ? e1 : e2);
  bool done = f == nullptr;
  for (unsigned spins = 0; spins < (1u << 22); ++spins) {
    if (!done) done = __hip_atomic_load((const gu32_t*)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == e;
    if (__all(done)) return true;
    __builtin_amdgcn_s_sleep(2);
  }
  return false;
}

// LDS of an owner of R rows (doubles): [G 64][normals K x R][stage K x R][Gram scratch 256][words 8][chunk sums nch x K x R]
__host__ __device__ constexpr int w_owner_lds_doubles(int K, int R, int nch) { return 64 + 2 * K * R + 256 + 8 + nch * K * R; }
__host__ __device__ constexpr int w_tail_lds_doubles(int K) { return w_owner_lds_doubles(K, 8, 1); }
// what the FUSE_W instances have (the reduction's scratch): 16 waves x ACC_RG x 128 doubles, ACC_RG = min(6, K) but >= 4
__host__ __device__ constexpr int w_owner_lds_budget(int K) { return 16 * (K < 4 ? 4 : (K > 6 ? 6 : K)) * ACC_TILE; }

// An owner of R = fw.rows consecutive rows of a 128-row tile (128 / R owners per tile: the hand-off of the chunk sums is
// bandwidth-bound per reading workgroup - ~65 GB/s - so a tile's sums are read by several).  Everything the solve needs
// besides the chunk sums is made HERE, beside the stream and with nothing to publish - the rows' normals (device rng), the
// shared Gram summed as a 256-thread w_solve workgroup sums it, the scalars - then the owner waits for the tile's
// streaming workgroups (one counter), fetches their chunk sums (sc1 loads, all in flight), adds them per (row, value) in
// chunk order (w_solve_kernel's canonical order, chunk_sum_seq), factors and draws the rows (factor.py:349-362) and leaves
// the W'W shares of its virtual w_solve workgroups.  Bit for bit what w_solve_kernel does.
template <int K, int WAVES>
__device__ __forceinline__ void w_fused_owner(const FuseW& fw, int ob, double* lds, long long* stamps) {
  constexpr int KK = tri(K);
  constexpr int NT = WAVES * WAVE;
  constexpr int NTG = WS_ROWS * ws_split_of(K, false);  // threads of a w_solve workgroup (the Gram reduction's geometry)
  constexpr int GL = (NTG / KK) > 32 ? 32 : ((NTG / KK) < 1 ? 1 : (NTG / KK));
  const WSolveArgs& a = fw.a;
  const int RW = fw.rw;                                 // rows per virtual w_solve workgroup: 8, 16, 32 or 64
  const int R = fw.rows, OPT = ACC_TILE / R;            // rows of this owner, owners per tile
  const int tile = ob / OPT, r0 = tile * ACC_TILE + (ob - tile * OPT) * R;      // first (local) row
  double as = a.s, asR = a.sR, ainv_sigma2 = a.inv_sigma2;      // (device-resident scalars replace them below)
  double* G = lds;                                      // [64] the shared Gram, unscaled
  double* zsh = lds + 64;                               // [K][R] the rows' normals
  double* stg = zsh + K * R;                            // [K][R] the fresh rows, for the W'W shares
  double* gscr = stg + K * R;                           // [256]
  unsigned* word = reinterpret_cast<unsigned*>(gscr + 256);
  double* csum = gscr + 256 + 8;                        // [nch][K][R] the chunk sums as fetched
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (r0 >= a.nl) return;                               // (a ragged last tile: no such rows; uniform)
  // ---- prepare, while the tile's rows are being streamed elsewhere ----
  // (a) the Gram partials of the fixed factor - made by side workgroups of this launch in sharded runs: then all of
  //     them first - thread (l, q) adds the partials l, l + lanes, ... in order
  bool okp = true;
  if (fw.gram_cnt) {
    if (threadIdx.x == 0) word[0] = poll_flag(fw.gram_cnt, fw.gram_expected) ? 1u : 0u;
    __syncthreads();
    okp = word[0] != 0u;
  }
  {
    const int gl = threadIdx.x / KK, gq = threadIdx.x - gl * KK;
    if ((int)threadIdx.x < NTG && gl < GL) {
      double s = 0.0;
      for (int b0 = gl; b0 < a.ngp; b0 += 8 * GL) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int bb = b0 + u * GL;
          x[u] = bb < a.ngp ? (fw.gram_cnt ? load_sc1(a.gpart + (size_t)bb * KK + gq) : a.gpart[(size_t)bb * KK + gq]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += x[u];
      }
      gscr[gl * KK + gq] = s;
    }
  }
  // (b) the normals of the rows: (component, row) pairs over the threads
  for (int t = threadIdx.x; t < K * R; t += NT) {
    const int k = t / R, r = t - k * R;
    double zv = 0.0;
    if (r0 + r < a.nl) {
      const int i = a.row0 + r0 + r;
      const long long zoff = w_z_offset(i, K);
      const int d = i + 1 < K ? i + 1 : K;
      if (k < d) zv = a.z ? a.z[zoff + k] : philox_normal(a.seed, a.stream, (unsigned long long)(zoff + k));
    }
    zsh[t] = zv;
  }
  __syncthreads();
  if ((int)threadIdx.x < KK) {
    double t = 0.0;
    for (int bb = 0; bb < GL; ++bb) t += gscr[bb * KK + threadIdx.x];
    G[threadIdx.x] = t;                                  // (scaled by R / nu2 where it is used: w_solve_kernel's product)
  }
  // (c) scalars: earlier launches' draws, or - full sweeps - the scalar side workgroup's of this launch (its flag)
  if (a.hyp) {
    double hnu2 = a.hyp[HYP_NU2], hsg2 = a.hyp[HYP_SIGMA2];
    if (fw.hp.flag) {
      if (threadIdx.x == 0) word[0] = poll_flag(fw.hp.flag, fw.hp.epoch) ? 1u : 0u;
      __syncthreads();
      okp = okp && word[0] != 0u;
      if (fw.hp.which & 1) hnu2 = load_sc1(fw.hp.pub + HYP_NU2);
      if (fw.hp.which & 2) hsg2 = load_sc1(fw.hp.pub + HYP_SIGMA2);
    }
    if (a.hyp_noise) { as = 1.0 / hnu2; asR = as * a.Rrep; }
    ainv_sigma2 = 1.0 / hsg2;
  }
  TAIL_STAMP(stamps, 4);
  // ---- wait for the tile's streaming workgroups: one lane polls the tile's counter ----
  __syncthreads();                                       // (word[0] is read by everybody above)
  if (threadIdx.x == 0) word[0] = (okp && poll_flag(fw.cnt + tile, fw.expected)) ? 1u : 0u;
  __syncthreads();
  if (word[0] == 0u) {                                   // somebody never showed up: report, write nothing
    if (threadIdx.x == 0 && atomicCAS(&a.status[0], 0, 2) == 0) a.status[1] = -1;
    return;
  }
  TAIL_STAMP(stamps, 5);
  // ---- the chunk sums of the rows: every (chunk, value, row) a load of its own (sc1), all in flight; then the thread
  //      of a (value, row) pair adds its nch values in chunk order ----
  {
    const int KR = K * R, tot = a.nch * KR;
    const size_t cst = (size_t)K * a.ld;
    for (int e0 = threadIdx.x; e0 < tot; e0 += 8 * NT) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * NT;
        if (e < tot) {
          const int c = e / KR, rem = e - c * KR, v = rem / R, r = rem - v * R;
          x[u] = load_sc1(a.part + (size_t)c * cst + (size_t)v * a.ld + (size_t)(r0 + r));
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) if (e0 + u * NT < tot) csum[e0 + u * NT] = x[u];
    }
    __syncthreads();
    if ((int)threadIdx.x < KR) {
      double sum = 0.0;
      for (int c = 0; c < a.nch; ++c) sum += csum[c * KR + threadIdx.x];
      stg[threadIdx.x] = sum;                            // (stg [v][r] holds the sums until the fresh rows replace them)
    }
    __syncthreads();
  }
  TAIL_STAMP(stamps, 6);
  // ---- the solve: thread r < R takes row r0 + r ----
  const int il = r0 + (int)threadIdx.x;
  const bool solver = (int)threadIdx.x < R;
  const bool live = solver && il < a.nl;
  const int i = a.row0 + (live ? il : r0);
  const int d = i + 1 < K ? i + 1 : K;
  double wrow[K];
  if (solver) {
  double m[K], Q[KK];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = 0.0;
    s += stg[k * R + threadIdx.x];
    m[k] = __dmul_rn(s, as);
  }
#pragma unroll
  for (int q = 0; q < KK; ++q) Q[q] = __dmul_rn(G[q], asR);      // (w_solve_kernel rounds the scaled Gram before the shift is added)
#pragma unroll
  for (int k = 0; k < K; ++k) Q[lidx(k, k)] += ainv_sigma2;
#pragma unroll
  for (int r = 0; r < K; ++r) {
    if (r >= d) {
      m[r] = 0.0;
#pragma unroll
      for (int c = 0; c <= r; ++c) Q[lidx(r, c)] = (r == c) ? 1.0 : 0.0;
    }
  }
  bool ok = true;
  double invl[K];
#pragma unroll
  for (int c = 0; c < K; ++c) {
    double p = Q[lidx(c, c)];
#pragma unroll
    for (int q = 0; q < c; ++q) p = fma(-Q[lidx(c, q)], Q[lidx(c, q)], p);
    if (!(p > 0.0)) ok = false;
    const double inv = rsq_nr(p);
    const double l = __dmul_rn(p, inv);
    invl[c] = inv;
    Q[lidx(c, c)] = l;
#pragma unroll
    for (int r = c + 1; r < K; ++r) {
      double v = Q[lidx(r, c)];
#pragma unroll
      for (int q = 0; q < c; ++q) v = fma(-Q[lidx(r, q)], Q[lidx(c, q)], v);
      Q[lidx(r, c)] = __dmul_rn(v, inv);
    }
  }
  if (live && !ok && atomicCAS(&a.status[0], 0, 1) == 0) a.status[1] = i;
  double y[K];
#pragma unroll
  for (int r = 0; r < K; ++r) {
    double v = m[r];
#pragma unroll
    for (int c = 0; c < r; ++c) v = fma(-Q[lidx(r, c)], y[c], v);
    y[r] = __dmul_rn(v, invl[r]);
  }
#pragma unroll
  for (int r = 0; r < K; ++r)
    if (r < d) y[r] = __dadd_rn(y[r], zsh[r * R + threadIdx.x]);
#pragma unroll
  for (int r = K - 1; r >= 0; --r) {
    double v = y[r];
#pragma unroll
    for (int c = r + 1; c < K; ++c) v = fma(-Q[lidx(c, r)], y[c], v);
    y[r] = __dmul_rn(v, invl[r]);
  }
#pragma unroll
  for (int r = 0; r < K; ++r) {
    const bool fresh = live && ok && r < d;
    if (fresh) a.W[(size_t)i * K + r] = y[r];
    wrow[r] = fresh ? y[r] : 0.0;
  }
  if (live && (d < K || !ok)) {
#pragma unroll
    for (int r = 0; r < K; ++r)
      if (!(ok && r < d)) wrow[r] = a.W[(size_t)i * K + r];
  }
  }
  if (!a.gout) return;                                   // (uniform)
  __syncthreads();                                       // (every solver has read its sums: stg is free)
  if (solver) {
#pragma unroll
    for (int r = 0; r < K; ++r) stg[r * R + threadIdx.x] = wrow[r];
  }
  __syncthreads();
  {
    // W'W shares on the matrix cores, one per virtual w_solve workgroup of RW rows, as w_solve_kernel forms them: rows
    // 4 s + kk of the workgroup in step s (its steps beyond RW / 4 multiply zeros there and are skipped here: adding +0
    // products to an accumulator that started at +0 changes no bit).  Wave w takes the workgroups w, w + WAVES, ...
    typedef double v4f64_w __attribute__((ext_vector_type(4)));
    const int kk = lane >> 4, ii = lane & 15;
    const int nlt = a.nl - r0;                           // rows of this owner that exist
    for (int vbl = wave; vbl < R / RW; vbl += WAVES) {
      if (vbl * RW < nlt) {                              // (a workgroup launch_wsolve would have launched; wave-uniform)
        const double* src = stg + (ii < K ? ii : 0) * R + vbl * RW + kk;
        v4f64_w acc = {0.0, 0.0, 0.0, 0.0};
        for (int sidx = 0; sidx < RW / 4; ++sidx) {
          const double x = ii < K ? src[4 * sidx] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
        }
        const int vb = r0 / RW + vbl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * r + kk;
          if (row < K && ii <= row) a.gout[(size_t)vb * KK + lidx(row, ii)] = acc[r];
        }
      }
    }
  }
  TAIL_STAMP(stamps, 7);
}

// =================================================================================================================
// V tail
// =================================================================================================================
struct FuseV {
  VSpecArgs a;             // what launch_vspectral would have been handed (a.part: this launch's partials; a.eig unused)
  unsigned* cnt;           // [tiles] arrival tickets (several chunks), or nullptr: one chunk, the sums stay in LDS
  const double* eig_pub;   // [K + K K] eigen-system of W'W as published by the side workgroup of this launch
  const unsigned* eig_flag; unsigned epoch;
  HypPub hp;               // lam2 drawn by a side workgroup of this launch (flag == nullptr: none)
  int band_early;          // 1: the prior band is formed before the stream (v_fused_band_early)
  int dataflow;            // 1: the barrier-free tail (v_fused_df): one chunk, precomputed band, nembeds <= 6 (launch_accum decides)
  const unsigned long long* eig_gran;   // [2 K] the eigenvalues as tagged granules {half of g_k, epoch} (EigSide.gran)
  const double* pimg;      // [ml][2 PB] the LDS image [P | Pm] of every column's band (prior_band_image_kernel): dataflow tails copy it
};
template <> __device__ __forceinline__ unsigned* fuse_tickets<FUSE_V>(const FuseV& fz) { return fz.cnt; }
template <> __device__ __forceinline__ int fuse_chunks<FUSE_V>(const FuseV& fz) { return fz.a.nch; }
template <> __device__ __forceinline__ int fuse_owners<FUSE_V>(const FuseV&) { return 0; }
constexpr int VF_GROUP_WAVES = 4;
constexpr int VF_MAILBOX = 128;                          // doubles at the top of the instance's LDS: eigen-system, lam2, poll result
// what a thread of a column's virtual workgroup fetches between its stream and the cross-wave reduction - the loads fly
// while the slower waves finish their rows and the sums are reduced: the stencil of its band entry, its Tau2 values
template <> struct FusePre<FUSE_V> { typedef struct VPre type; };
template <> struct FusePre<FUSE_VDF> { typedef struct VPre type; };
static_assert(VF_MAILBOX == VF_MAILBOX_DOUBLES, "mailbox");
// BTF_VF_BANDPRE (default): the prior band of the tile's columns comes PRECOMPUTED (prior_band_kernel, whenever Tau2 / lam2
// changed: btf_abi.hip, pband_version) - one double per thread loaded at kernel start and parked in a register pair through
// the stream.  The form it replaces fetched the thread's stencil (16 rows + 16 coefficients) and Tau2 values a few row
// groups before the stream ended: 50 live VGPRs under the 128-VGPR cap of the 16-wave instance, and the stream of the
// fused launch ran 10.4 us instead of 8.5 (stamps).
#ifndef BTF_VF_BANDPRE
#define BTF_VF_BANDPRE 1
#endif
#if BTF_VF_BANDPRE
struct VPre { double pv[2]; };      // entries tid and tid + 256 of the column's band (T (S+1) <= 512: T <= 128 at tf_order 2)
#else
struct VPre { int se0, se1; int srow[VS_MAXE]; double scf[VS_MAXE]; double tau[2]; };
#endif                        // waves of a column's virtual workgroup (= VS_THREADS / 64)
// the prior band of the tile's columns, formed BEFORE the stream (v_fused_band_early): per column [1/(lam2 Tau2) nD][P][Pm]
// below the mailbox at the top of the instance's LDS
__host__ __device__ inline int vf_band_doubles(int T, int TF, int nD) { return ((nD + 1) & ~1) + 2 * (((T + TF + 2) * (TF + 2) + 1) & ~1); }
__host__ __device__ inline int vf_top_base(int T, int TF, int nD, bool unr3) {
  return vf_red_doubles(unr3) - VF_MAILBOX - (ACC_TILE / T) * vf_band_doubles(T, TF, nD);
}
// columns per 128-output tile, 0 = the fused V tail does not apply to this depth axis
__host__ __device__ constexpr int vf_cols_per_tile(int T) { return (T == 32 || T == 64 || T == 128) ? ACC_TILE / T : 0; }
__host__ __device__ inline int vf_col_stride(int T, int K, int TF, int nD) { return (vs_layout(T, K, TF, nD, false).total + 1) & ~1; }
__host__ __device__ inline bool vf_fits(int T, int K, int TF, int nD, int waves, bool unr3) {
  const int ng = vf_cols_per_tile(T);
  if (!(ng > 0 && ng * VF_GROUP_WAVES <= waves && nD <= 2 * VF_GROUP_WAVES * WAVE && K + K * K + 8 <= VF_MAILBOX)) return false;
  const int top = vf_top_base(T, TF, nD, unr3);
  return ng * vf_col_stride(T, K, TF, nD) <= top && 16 * 6 * ACC_TILE <= top;       // (the sampler's layouts and the reduction's scratch below the early bands)
}

// BEFORE the stream (every thread of the workgroup calls it; two barriers): the prior band of the tile's columns,
// P[t][d] = sum_r Delta[r,t] Delta[r,t+d] / (lam2 Tau2[j,r]) (rows ascending, as factor.py:404-405 forms the product) and
// its mirror image, into the top of the LDS - it depends on the hyper-parameters only, and the memory system is idle
// now; behind the stream the same loads queue behind everybody's.  Not in full sweeps whose lam2 is drawn by a side
// workgroup of this very launch (fv.hp.flag): returns false, and the tail forms the band itself.
#if BTF_VF_BANDPRE
// at kernel start: this thread's entry of the precomputed band (a.pband: [ml][T][S+1]), issued ahead of the stream's first
// loads (loads return in order: it is back when they are)
template <int K, int S>
__device__ __forceinline__ bool v_fused_band_preload(const FuseV& fv, int tile, VPre& pre) {
  pre.pv[0] = pre.pv[1] = 0.0;
  const VSpecArgs& a = fv.a;
  if (!a.pband) return false;                             // (uniform: the tail forms the band itself)
  const int T = a.T, NG = ACC_TILE / T;
  const int lane = threadIdx.x & 63;
  const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = pw % NG, wave = pw / NG, tid = wave * WAVE + lane;
  const int j = tile * NG + cg;
  constexpr int NT = VF_GROUP_WAVES * WAVE;
  if (wave < VF_GROUP_WAVES && j < a.ml) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (tid + u * NT < T * (S + 1)) pre.pv[u] = a.pband[(size_t)j * T * (S + 1) + tid + u * NT];
  }
  return true;
}
// behind the stream, in front of the reduction (whose barriers publish it): the band and its mirror image into the top of
// the LDS.  Thread (t, d) writes P[t][d] and its mirror entry Pm[T-1-t-d][d]; the entries of Pm no mirror image lands on
// are those with t + d >= T, written as zeros by their own (t, d) thread - no zero fill, no barrier of its own.
template <int K, int S>
__device__ __forceinline__ void v_fused_band_store(const FuseV& fv, int tile, double* lds, const VPre& pre, bool unr3) {
  const VSpecArgs& a = fv.a;
  const int T = a.T, NG = ACC_TILE / T;
  constexpr int D1 = S + 1, NT = VF_GROUP_WAVES * WAVE;
  const int lane = threadIdx.x & 63;
  const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = pw % NG, wave = pw / NG, tid = wave * WAVE + lane;
  const int j = tile * NG + cg;
  if (!(wave < VF_GROUP_WAVES && j < a.ml)) return;
  const int PB = ((T + S + 1) * D1 + 1) & ~1, nDp = (a.nD + 1) & ~1;
  double* top = lds + vf_top_base(T, a.TF, a.nD, unr3) + (size_t)cg * (nDp + 2 * PB);
  double* P = top + nDp;
  double* Pm = P + PB;
  for (int idx = tid, u = 0; idx < (T + S + 1) * D1; idx += NT, ++u) {
    const double v = (idx < T * D1 && u < 2) ? (u == 0 ? pre.pv[0] : pre.pv[1]) : 0.0;      // (T D1 <= 2 NT: vf_fits)
    const int t = idx / D1, d = idx - t * D1;
    P[idx] = v;
    if (t + d < T) Pm[(T - 1 - t - d) * D1 + d] = v; else Pm[idx] = 0.0;
  }
}
#endif
template <int K, int S>
__device__ __forceinline__ bool v_fused_band_early(const FuseV& fv, int tile, double* lds, bool unr3) {
#if BTF_VF_BANDPRE
  return false;                                           // (replaced by v_fused_band_preload / v_fused_band_store)
#endif
  if (fv.hp.flag || !fv.band_early) return false;         // (uniform)
  const VSpecArgs& a = fv.a;
  const int T = a.T;
  constexpr int D1 = S + 1, MAXE = VS_MAXE, NT = VF_GROUP_WAVES * WAVE;
  const int NG = ACC_TILE / T;
  const int lane = threadIdx.x & 63;
  const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = pw % NG, wave = pw / NG, tid = wave * WAVE + lane;
  const int j = tile * NG + cg, jg = a.col0 + j;
  const bool act = wave < VF_GROUP_WAVES && j < a.ml;
  const int PB = ((T + S + 1) * D1 + 1) & ~1, nDp = (a.nD + 1) & ~1;
  double* top = lds + vf_top_base(T, a.TF, a.nD, unr3) + (size_t)cg * (nDp + 2 * PB);
  double* itau = top;
  double* P = top + nDp;
  double* Pm = P + PB;
  const double lam2 = a.hyp ? a.hyp[HYP_LAM2] : a.lam2;
  int se0 = 0, se1 = 0;
  int srow[MAXE];
  double scf[MAXE];
  if (act) {
    if (tid < T * D1) {
      se0 = a.st_ptr[tid]; se1 = a.st_ptr[tid + 1];
#pragma unroll
      for (int u = 0; u < MAXE; u += 4) {
        const int4 r4 = *reinterpret_cast<const int4*>(a.st_drow + (size_t)tid * MAXE + u);
        srow[u] = r4.x; srow[u + 1] = r4.y; srow[u + 2] = r4.z; srow[u + 3] = r4.w;
      }
#pragma unroll
      for (int u = 0; u < MAXE; u += 2) {
        const double2 c2 = *reinterpret_cast<const double2*>(a.st_dcoef + (size_t)tid * MAXE + u);
        scf[u] = c2.x; scf[u + 1] = c2.y;
      }
    }
    for (int idx = tid; idx < a.nD; idx += NT) itau[idx] = 1.0 / (lam2 * a.Tau2[(size_t)jg * a.nD + idx]);
    for (int idx = tid; idx < (T + S + 1) * D1; idx += NT) Pm[idx] = 0.0;
  }
  __syncthreads();
  if (act) {
    const int scnt = se1 - se0;
    for (int idx = tid; idx < (T + S + 1) * D1; idx += NT) {
      double s = 0.0;
      if (idx == tid && idx < T * D1) {
#pragma unroll
        for (int u = 0; u < MAXE; ++u) if (u < scnt) s = fma(scf[u], itau[srow[u]], s);
      } else if (idx < T * D1) {
        for (int e = a.st_ptr[idx]; e < a.st_ptr[idx + 1]; ++e) s = fma(a.st_coef[e], itau[a.st_row[e]], s);
      }
      P[idx] = s;
      const int t = idx / D1, d = idx - t * D1;
      if (t + d < T) Pm[(T - 1 - t - d) * D1 + d] = s;
    }
  }
  return true;                                           // (published by the reduction's barriers)
}

// A few row groups before the stream ends (every thread calls it): the tail's own global loads - the stencil of the
// thread's band entry, its Tau2 values - so that they fly under the last rows instead of sitting in front of the
// reduction's first barrier (unless the band was formed before the stream: nothing to fetch).
template <int K, int S, class PRE>
__device__ __forceinline__ void v_fused_prefetch_loads(const FuseV& fv, int tile, PRE& pre, bool band_early) {
  const VSpecArgs& a = fv.a;
  const int T = a.T;
  constexpr int D1 = S + 1, MAXE = VS_MAXE, NT = VF_GROUP_WAVES * WAVE;
  const int NG = ACC_TILE / T;
  const int lane = threadIdx.x & 63;
  const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = pw % NG, wave = pw / NG, tid = wave * WAVE + lane;
  const int j = tile * NG + cg, jg = a.col0 + j;
  const bool act = wave < VF_GROUP_WAVES && j < a.ml;
#if BTF_VF_BANDPRE
  (void)act; (void)jg; (void)tid;
#else
  pre.se0 = pre.se1 = 0;
  pre.tau[0] = pre.tau[1] = 1.0;
  if (act && !band_early) {
    if (tid < T * D1) {
      pre.se0 = a.st_ptr[tid]; pre.se1 = a.st_ptr[tid + 1];
#pragma unroll
      for (int u = 0; u < MAXE; u += 4) {
        const int4 r4 = *reinterpret_cast<const int4*>(a.st_drow + (size_t)tid * MAXE + u);
        pre.srow[u] = r4.x; pre.srow[u + 1] = r4.y; pre.srow[u + 2] = r4.z; pre.srow[u + 3] = r4.w;
      }
#pragma unroll
      for (int u = 0; u < MAXE; u += 2) {
        const double2 c2 = *reinterpret_cast<const double2*>(a.st_dcoef + (size_t)tid * MAXE + u);
        pre.scf[u] = c2.x; pre.scf[u + 1] = c2.y;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (tid + u * NT < a.nD) pre.tau[u] = a.Tau2[(size_t)jg * a.nD + tid + u * NT];
  }
#endif
}

// Between the stream and the reduction: wave 0 waits for this launch's side workgroups (eigen-system of W'W; lam2 in
// full sweeps); their published values go to the mailbox at the top of the LDS (the reduction and the sampler's layouts
// stay below it).
template <int K, int S>
__device__ __forceinline__ void v_fused_prefetch(const FuseV& fv, int tile, double* mailbox, VPre& pre, bool band_early) {
  const int lane = threadIdx.x & 63;
  const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (pw == 0) {
    const bool okw = poll_flags(fv.eig_flag, fv.epoch, fv.hp.flag, fv.hp.epoch, nullptr, 0u);
    if (okw) {
      for (int idx = lane; idx < K + K * K; idx += WAVE) mailbox[idx] = load_sc1(fv.eig_pub + idx);
      if (fv.hp.flag && lane == 0) mailbox[VF_MAILBOX - 2] = load_sc1(fv.hp.pub + HYP_LAM2);
    }
    if (lane == 0) *reinterpret_cast<unsigned*>(mailbox + VF_MAILBOX - 1) = okw ? 1u : 0u;
  }
}

// `sums`: this thread's values of the cross-wave reduction (value g0 + tv of column tc for its rounds), used when the
// launch has ONE chunk; otherwise the last arriver of the tile adds the chunks from the partials.
// Control flow: every thread of the workgroup runs every barrier; `act` marks the threads of a live column group.
template <int K, int S, int WAVES, int NSUM>
__device__ __forceinline__ void v_fused_tail(const FuseV& fv, int tile, double* lds, const double (&sums)[NSUM], int nv_round, int tv, int tc,
                             long long* stamps, const VPre& pre, const double* mailbox, bool band_early, bool unr3) {
  TAIL_STAMP(stamps, 4);
  VSpecArgs a = fv.a;
  const int T = a.T, n = T * K, KK = tri(K);
  constexpr int D1 = S + 1, RS = S + 2, WN = S * (S + 1) + S;
  constexpr int MAXE = VS_MAXE;
  constexpr int NT = VF_GROUP_WAVES * WAVE;              // threads of a column's virtual workgroup (= VS_THREADS)
  static_assert(NT == VS_THREADS, "the virtual workgroup is v_spectral_kernel's");
  const int NG = ACC_TILE / T;                           // columns in the tile
  const VsLayout L = vs_layout(T, K, a.TF, a.nD, false);
  const int stride = (L.total + 1) & ~1;
  const int Tp = L.Tp;
  int nl, nr, ns;
  spectral_split(T, S, nl, nr, ns);
  const int lane = threadIdx.x & 63;
  const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // physical wave
  const int cg = pw % NG, wave = pw / NG;                // column group, wave of its virtual workgroup
  const int tid = wave * WAVE + lane;                    // thread of the virtual workgroup
  const int j = tile * NG + cg, jg = a.col0 + j;         // local / global column
  const bool act = wave < VF_GROUP_WAVES && j < a.ml;
  double* base = lds + (size_t)cg * stride;
  double* Ush = base + L.U;
  double* gsh = base + L.g;
  double* itau = base + L.itau;
  double* P = base + L.P;
  double* Pm = base + L.Pm;
  if (band_early) {                                        // (formed before the stream: v_fused_band_early)
    const int PB = ((T + S + 1) * (S + 1) + 1) & ~1, nDp = (a.nD + 1) & ~1;
    double* top = lds + vf_top_base(T, a.TF, a.nD, unr3) + (size_t)cg * (nDp + 2 * PB);
    itau = top; P = top + nDp; Pm = P + PB;
  }
  double* mraw = base + L.mraw;
  double* mt = base + L.mt;
  double* mtm = base + L.mtm;
  double* zz = base + L.zz;
  double* rec = base + L.rec;
  double* win = base + L.win;
  double* flag = base + L.flag;

  // ---- the column sums into the sampler's layout (one chunk: straight from the reduction's registers) ----
  if (!fv.cnt) {
    const int scg = tc / T, st = tc - scg * T;
#pragma unroll
    for (int r = 0; r < NSUM; ++r) {
      const int k = r * nv_round + tv;
      if (tv < nv_round && k < K) lds[(size_t)scg * stride + L.mraw + st * K + k] = 0.0 + sums[r];
    }
  }
  // ---- (the stencil, Tau2 and the published eigen-system / lam2 came in before the reduction: v_fused_prefetch) ----
#if !BTF_VF_BANDPRE
  const int pidx = tid;
  const int scnt = pre.se1 - pre.se0;
#endif
  if (*reinterpret_cast<const unsigned*>(mailbox + VF_MAILBOX - 1) == 0u) {      // a producer of this launch never showed up (uniform)
    if (threadIdx.x == 0 && atomicCAS(&a.status[0], 0, 2) == 0) a.status[1] = -1;
    return;
  }
  if (a.hyp) {
    if (a.hyp_noise) { a.s = 1.0 / a.hyp[HYP_NU2]; a.sR = a.s * a.Rrep; }
    a.lam2 = (fv.hp.flag && (fv.hp.which & 4)) ? mailbox[VF_MAILBOX - 2] : a.hyp[HYP_LAM2];
  }
  if (act) {
    for (int idx = tid; idx < K + K * K; idx += NT) {
      const double v = mailbox[idx];
      if (idx < K) gsh[idx] = v; else Ush[idx - K] = v;
    }
  }
  if (act) {
    if (!band_early) {
#if BTF_VF_BANDPRE
      // (no precomputed band - lam2 is drawn by a side workgroup of this very launch: Tau2 fetched here, the stencil below)
      for (int idx = tid; idx < a.nD; idx += NT) itau[idx] = 1.0 / (a.lam2 * a.Tau2[(size_t)jg * a.nD + idx]);
#else
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + u * NT < a.nD) itau[tid + u * NT] = 1.0 / (a.lam2 * pre.tau[u]);
#endif
    }
    if (fv.cnt) {
      const size_t st = (size_t)K * a.ld;
      for (int e0 = tid; e0 < n; e0 += 2 * NT) {             // element e = k*T + t: coalesced along t
        const int e1 = e0 + NT;
        const bool h1 = e1 < n;
        const int k0 = e0 / T, t0 = e0 - k0 * T;
        const int e1c = h1 ? e1 : e0;
        const int k1 = e1c / T, t1 = e1c - k1 * T;
        const double* p0 = a.part + (size_t)k0 * a.ld + (size_t)j * T + t0;
        const double* p1 = a.part + (size_t)k1 * a.ld + (size_t)j * T + t1;
        double s0 = 0.0, s1 = 0.0;
        int c = 0;
        for (; c + 4 <= a.nch; c += 4) {                       // fixed order, four chunks in flight per element
          const double x0 = load_sc1(p0 + (size_t)c * st), x1 = load_sc1(p0 + (size_t)(c + 1) * st), x2 = load_sc1(p0 + (size_t)(c + 2) * st), x3 = load_sc1(p0 + (size_t)(c + 3) * st);
          const double y0 = load_sc1(p1 + (size_t)c * st), y1 = load_sc1(p1 + (size_t)(c + 1) * st), y2 = load_sc1(p1 + (size_t)(c + 2) * st), y3 = load_sc1(p1 + (size_t)(c + 3) * st);
          s0 += x0; s0 += x1; s0 += x2; s0 += x3;
          s1 += y0; s1 += y1; s1 += y2; s1 += y3;
        }
        for (; c < a.nch; ++c) { s0 += load_sc1(p0 + (size_t)c * st); s1 += load_sc1(p1 + (size_t)c * st); }
        mraw[t0 * K + k0] = s0;
        if (h1) mraw[t1 * K + k1] = s1;
      }
    }
    if (!band_early) for (int idx = tid; idx < (T + S + 1) * D1; idx += NT) Pm[idx] = 0.0;
  }
  __syncthreads();
  TAIL_STAMP(stamps, 5);
  // ---- prior band and its mirror image (rows ascending, as factor.py:404-405 forms the product); rotated right-hand
  //      sides and their mirror image ----
  if (act) {
    if (!band_early)
    for (int idx = tid; idx < (T + S + 1) * D1; idx += NT) {
      double s = 0.0;
#if BTF_VF_BANDPRE
      if (idx < T * D1) {
#else
      if (idx == pidx) {
#pragma unroll
        for (int u = 0; u < MAXE; ++u) if (u < scnt) s = fma(pre.scf[u], itau[pre.srow[u]], s);
      } else if (idx < T * D1) {
#endif
        for (int e = a.st_ptr[idx]; e < a.st_ptr[idx + 1]; ++e) s = fma(a.st_coef[e], itau[a.st_row[e]], s);
      }
      P[idx] = s;
      const int t = idx / D1, d = idx - t * D1;
      if (t + d < T) Pm[(T - 1 - t - d) * D1 + d] = s;
    }
    for (int idx = tid; idx < K * Tp; idx += NT) {
      const int k = idx / Tp, t = idx - k * Tp;
      double s = 0.0;
      if (t < T) {
#pragma unroll
        for (int kk = 0; kk < K; ++kk) s = fma(Ush[kk * K + k], mraw[t * K + kk], s);
        s *= a.s;
      }
      mt[idx] = s;
      mtm[k * Tp + (t < T ? T - 1 - t : t)] = s;
    }
  }
  __syncthreads();
  // ---- the 2K chains (wave 0 of the group) while its other waves draw the normals ----
  if (act) {
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
          if (ns > 0) {
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
      for (int idx = tid - WAVE; idx < n; idx += NT - WAVE)
        zz[idx] = a.z ? a.z[(size_t)jg * n + idx] : philox_normal(a.seed, a.stream, (unsigned long long)jg * n + idx);
    }
  }
  __syncthreads();
  TAIL_STAMP(stamps, 6);
  const bool ok = act && flag[0] != 0.0;                   // this column's factorisation went through
  if (act && tid == 0) a.tries[j] = (int)flag[1];
  if (act && !ok) {
    if (tid == 0 && atomicCAS(&a.status[0], 0, 1) == 0) a.status[1] = jg;
  }
  // ---- w = D^-1 u + D^-1/2 z  (pivot order) ----
  if (ok) {
    for (int idx = tid; idx < n; idx += NT) {
      const double iv = rec[(size_t)idx * RS + S], w = rec[(size_t)idx * RS + S + 1];
      rec[(size_t)idx * RS + S + 1] = fma(w, iv, zz[idx] * sqrt(iv));
    }
  }
  __syncthreads();
  if (ok && wave == 0) {
    double x[S + 1];
#pragma unroll
    for (int d = 0; d <= S; ++d) x[d] = 0.0;
    if (ns > 0 && tid < K) {                              // separator: S x S unit upper solve
      double* srec = rec + (size_t)(tid * T + nl + nr) * RS;
#pragma unroll
      for (int c = S - 1; c >= 0; --c) {
        double acc = srec[c * RS + S + 1];
#pragma unroll
        for (int d = 1; d <= S; ++d) if (c + d < S) acc = fma(-srec[c * RS + d - 1], srec[(c + d) * RS + S + 1], acc);
        srec[c * RS + S + 1] = acc;
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
      spectral_backward<S>(rec + (size_t)(k * T + (side ? nl : 0)) * RS, side ? nr : nl, x);
    }
  }
  __syncthreads();
  // ---- rotate back, write V[j] (depth-major), residual part, Gram share ----
  double* xout = mraw;
  double sse_acc = 0.0;
  const double inv_s2 = -2.0 / a.s;
  const double* xs = rec + S + 1;
  if (ok) {
    for (int idx = tid; idx < n; idx += NT) {
      const int t = idx / K, k = idx - t * K;
      const int pos = t < nl ? t : (t < nl + ns ? nl + nr + (t - nl) : nl + (T - 1 - t));
      double s = 0.0;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) s = fma(Ush[k * K + kk], xs[(size_t)(kk * T + pos) * RS], s);
      xout[idx] = s;
      a.V[(size_t)jg * n + idx] = s;
      if (a.sse_out) {
        const double xt = xs[(size_t)(k * T + pos) * RS];
        sse_acc = fma(xt, fma(a.Rrep * gsh[k], xt, inv_s2 * mt[k * Tp + t]), sse_acc);
      }
    }
  }
  if (a.sse_out) {                                         // (workgroup-uniform)
    const double v = wave_sum(sse_acc);
    if (ok && lane == 0) flag[2 + wave] = v;
    __syncthreads();
    if (ok && tid == 0) a.sse_out[j] = (flag[2] + flag[3]) + (flag[4] + flag[5]);
  }
  if (a.gout) {                                            // (workgroup-uniform)
    __syncthreads();
    int ng = NT / KK;
    if (ng > 16) ng = 16;
    if (ng < 1) ng = 1;
    const int g = tid / KK, q = tid - g * KK;
    int p = 0;
    while ((p + 1) * (p + 2) / 2 <= q) ++p;
    const int pq = q - p * (p + 1) / 2;
    double* scratch = base + L.gs;
    if (ok && g < ng) {
      double s = 0.0;
      for (int t = g; t < T; t += ng) s = fma(xout[t * K + p], xout[t * K + pq], s);
      scratch[g * KK + q] = s;
    }
    __syncthreads();
    if (ok && tid < KK) {
      double s = 0.0;
      for (int b = 0; b < ng; ++b) s += scratch[b * KK + tid];
      a.gout[(size_t)j * KK + tid] = s;
    }
  }
  TAIL_STAMP(stamps, 7);
}

// =================================================================================================================
// V tail, dataflow form (FuseV.dataflow; BTF_OPT_FUSED_DATAFLOW 0 keeps the barrier form above)
// =================================================================================================================
// The barrier form serialises, behind the LAST wave's stream: reduction -> sums into the sampler's layout -> rotation ->
// the 2K elimination chains (31 pivots of ~240 cycles on ONE wave per column) -> w -> back-substitution -> rotation back.
// But the factorisation of A_k = g_k I + P_j needs only the eigenvalues (the side workgroup has them ~7 us into the
// launch) and the prior band (precomputed) - not the column sums.  Here the CHAIN wave of a column does not stream at
// all (accum_kernel deals the rows over the other 16 - NG waves, BTF_DF_UNR in flight each): it draws the column's
// normals, polls the eigenvalues' granules, factors, scales the normals by sqrt(1/D) - all while the others stream - and
// then meets the right-hand sides in a forward substitution of four fused multiply-adds per pivot.
// No workgroup barrier behind the stream: a wave at an s_barrier would wait for the chain waves.  The stages are ordered
// by counters in LDS (one relaxed add per producing wave behind a workgroup-scope release; consumers poll with s_sleep):
// every wait names producers that signal unconditionally and every wave's program is a straight line of stages, so the
// grid drains whatever the data hold (a failed factorisation or a missed flag skips the arithmetic, not the signals).
// The same operations as v_fused_tail / v_spectral_kernel on column sums that are grouped over 16 - NG waves instead of
// 16: equal to those forms to rounding, deterministic (tests/test_gpu_fused.py).
//
// What the measurements taught (scripts/chain_probe.hip, scripts/stamps_df_ab.sh): a lone wave issues one instruction
// per ~9 cycles WHATEVER it is (f64 arithmetic, an LDS read, a move) - the chains cost their instruction count, so the
// band arrives as a ready LDS image copied by another wave, the records are 32-byte quads [l1 l2 l3 | 1/D -> w -> x] and
// the right-hand sides come paired with the scaled normals [r(i+S) z(i) sqrt(1/D(i))] - three reads, four fused
// multiply-adds and one write per forward pivot, two reads, three and one per backward pivot.
//
// Roles in a tile of NG = 128 / T columns (16 waves): p < NG is the CHAIN wave of column p; every other wave streams and
// is afterwards a worker of column p % NG (one element of the column per thread): sums over the waves' partials,
// rotation into the eigen-basis, delivery where the chains read.  The first worker (p = NG) asks for the eigen-system
// inside its stream (accum_kernel: the flag's word behind one row group's loads, the payload behind the next).  The four
// virtual waves of the final stage (rotation back, store, residual part, Gram share - v_spectral_kernel's thread
// geometry) are p / NG = 0..3.  LDS (doubles): [0, 16 K 128) the waves' partial sums [wave][k][128]; behind them the
// columns' working arrays (mraw, mt, the (r, z) pairs, heads, Gram scratch); below the mailbox, per column: P | Pm, the
// record quads, the separator windows, flag words - the chain waves' working set.  Mailbox (top 128 doubles):
// eigen-system [0, K + K K), 64 counter words from double 96.
enum { DFC_PART = 0, DFC_EIG = 2, DFC_BAD = 3, DFC_GROUP0 = 8, DFC_PER_GROUP = 12 };
enum { DFG_IN = 0, DFG_ROT, DFG_X, DFG_G1, DFG_G2, DFG_SSE, DFG_BAND };
struct DfLayout { int PB, P, Pm, Q, win, flag, size; };
__host__ __device__ inline DfLayout df_layout(int T, int K, int S) {
  DfLayout D;
  D.PB = ((T + S + 1) * (S + 1) + 1) & ~1;
  int o = 0;
  D.P = o; o += D.PB;
  D.Pm = o; o += D.PB;
  D.Q = o; o += T * K * 4;
  D.win = o; o += (2 * K * (S * (S + 1) + S) + 1) & ~1;
  D.flag = o; o += 8;
  D.size = (o + 3) & ~3;                                     // (32-byte granules: the quads stay aligned)
  return D;
}
// the workers' arrays, per column, behind the streaming waves' partial sums
struct DfWork { int mraw, mt, rz, head, gs, stride; };
__host__ __device__ inline DfWork df_work(int T, int K, int S) {
  DfWork W;
  int o = 0;
  W.mraw = o; o += T * K;                                    // raw sums, depth-major; later the output staging
  W.mt = o; o += K * (T + S + 1);                            // rotated right-hand sides (the residual part reads them)
  o = (o + 1) & ~1;
  W.rz = o; o += 2 * T * K;                                  // [k][pivot][r(i+S), z sqrt(1/D)]
  W.head = o; o += 2 * K * 4;                                // [side][k][r(0), r(1), r(2), -]
  W.gs = o; o += VS_THREADS;                                 // Gram-share scratch
  W.stride = (o + 3) & ~3;
  return W;
}
// does the dataflow tail apply?  (T a whole number of columns per tile, K <= rg values in one reduction round, the chain
// region beside the partial sums and beside the overlaid working arrays)
__host__ __device__ inline bool vf_df_fits(int T, int K, int TF, int nD, int waves, int rg) {
  const int ng = vf_cols_per_tile(T);
  if (!(ng > 0 && ng * VF_GROUP_WAVES <= waves && K <= rg && K + K * K <= 96 && TF == 2)) return false;
  if (waves != 16) return false;
  if (2 * df_layout(T, K, TF + 1).PB > 2 * 8 * WAVE) return false;      // the band image: eight double2 per lane of the copying wave
  const int nwt = ((waves - ng) / ng) * WAVE;                 // worker threads per column (v_fused_df: every streaming wave): one element each
  if (T * K > nwt || T * K > 768) return false;
  const int room = vf_red_doubles(false) - VF_MAILBOX - ng * df_layout(T, K, TF + 1).size;
  return waves * K * ACC_TILE + ng * df_work(T, K, TF + 1).stride <= room;      // partial sums [w][k][128], the working arrays behind them
}
template <> __device__ __forceinline__ int fuse_chain_waves<FUSE_VDF>(const FuseV& fz) { return ACC_TILE / fz.a.T; }
template <> __device__ __forceinline__ const unsigned* fuse_eig_flag<FUSE_VDF>(const FuseV& fz) { return fz.eig_flag; }
template <> __device__ __forceinline__ const double* fuse_eig_pub<FUSE_VDF>(const FuseV& fz) { return fz.eig_pub; }
template <> __device__ __forceinline__ unsigned fuse_epoch<FUSE_VDF>(const FuseV& fz) { return fz.epoch; }
template <> __device__ __forceinline__ unsigned* fuse_tickets<FUSE_VDF>(const FuseV& fz) { return nullptr; }
template <> __device__ __forceinline__ int fuse_chunks<FUSE_VDF>(const FuseV& fz) { return 1; }
template <> __device__ __forceinline__ int fuse_owners<FUSE_VDF>(const FuseV&) { return 0; }

__device__ __forceinline__ void df_signal(unsigned* c) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// bounded (a producer is a wave of this very workgroup and signals whatever happens: the bound only turns a bug into a
// status code instead of a hung GPU)
#ifndef BTF_DF_SLEEP
#define BTF_DF_SLEEP 1
#endif
template <int SLEEP = BTF_DF_SLEEP>
__device__ __forceinline__ void df_wait(unsigned* c, unsigned target, unsigned* bad) {
  bool seen = false;
  for (unsigned spins = 0; spins < (1u << 22) / SLEEP; ++spins) {
    if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) { seen = true; break; }
    __builtin_amdgcn_s_sleep(SLEEP);
  }
  if (!seen && (threadIdx.x & 63) == 0) __hip_atomic_store(bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// kernel start (every thread): the counters to zero behind one barrier, then ONE worker wave per column asks for the
// column's band image - 2 PB / 128 double2 per lane, as it lies
template <int K, int S>
__device__ __forceinline__ void v_df_begin(const FuseV& fv, int tile, double* lds, VDfPre& pre) {
  unsigned* cw = reinterpret_cast<unsigned*>(lds + vf_red_doubles(false) - VF_MAILBOX + 96);
  if (threadIdx.x < 64) cw[threadIdx.x] = 0u;
  __syncthreads();
  v_df_band_load<K, S>(fv, tile, pre);
}
template <int K, int S>
__device__ __forceinline__ void v_df_band_load(const FuseV& fv, int tile, VDfPre& pre) {
  const VSpecArgs& a = fv.a;
  const int T = a.T, NG = ACC_TILE / T;
  const int lane = threadIdx.x & 63;
  const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = pw - NG, j = tile * NG + cg;               // waves NG .. 2 NG - 1: the columns' band copiers
#pragma unroll
  for (int u = 0; u < 8; ++u) pre.pv[u] = make_double2(0.0, 0.0);
  if (pw >= NG && pw < 2 * NG && j < a.ml) {
    const int PB = df_layout(T, K, S).PB;
    const double2* src = reinterpret_cast<const double2*>(fv.pimg + (size_t)j * 2 * PB);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (lane + 64 * u < PB) pre.pv[u] = src[lane + 64 * u];
  }
}
// ... and, once the first rows of its stream have been consumed (loads return in order: the image is back), stores it
// into the column's corner of the LDS and tells the chain wave
template <int K, int S>
__device__ __forceinline__ void v_df_band_store(const FuseV& fv, int tile, double* lds, const VDfPre& pre) {
  const VSpecArgs& a = fv.a;
  const int T = a.T, NG = ACC_TILE / T;
  const int lane = threadIdx.x & 63;
  const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = pw - NG;
  if (!(pw >= NG && pw < 2 * NG && tile * NG + cg < a.ml)) return;
  const DfLayout D = df_layout(T, K, S);
  double2* dst = reinterpret_cast<double2*>(lds + vf_red_doubles(false) - VF_MAILBOX - (size_t)(cg + 1) * D.size + D.P);
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (lane + 64 * u < D.PB) dst[lane + 64 * u] = pre.pv[u];
  unsigned* cw = reinterpret_cast<unsigned*>(lds + vf_red_doubles(false) - VF_MAILBOX + 96);
  df_signal(cw + DFC_GROUP0 + cg * DFC_PER_GROUP + DFG_BAND);
}

template <int K, int S, int WAVES, int RG, int NVV>
__device__ __forceinline__ void v_fused_df(const FuseV& fv, int tile, double* lds, const double (&acc)[NVV][2], long long* stamps,
                                           const DfEarly& early) {
  static_assert(NVV == K && K <= RG, "one reduction round");
  static_assert(S == 3 && WAVES == 16, "tf_order 2, 16 waves");
  VSpecArgs a = fv.a;
  const int T = a.T, n = T * K, KK = tri(K);
  constexpr int D1 = S + 1, WN = S * (S + 1) + S;
  constexpr int NT = VF_GROUP_WAVES * WAVE;               // threads of the final stage's virtual workgroup (= VS_THREADS)
  const int NG = ACC_TILE / T;
  const int Tp = T + S + 1;
  int nl, nr, ns;
  spectral_split(T, S, nl, nr, ns);
  const int lane = threadIdx.x & 63;
  const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lgNG = T == 128 ? 0 : (T == 64 ? 1 : 2);       // NG = 1, 2, 4: shifts instead of divisions
  const int cg = pw & (NG - 1), vw = pw >> lgNG;
  const int j = tile * NG + cg, jg = a.col0 + j;
  const bool live = j < a.ml;                              // (wave-uniform; a ragged last tile has dead column groups)
  // The chain wave of a column (p < NG) does not stream (accum_kernel): it draws the column's normals, waits for the
  // eigenvalues and factors while the other 16 - NG waves stream.  Those are all workers of the sums afterwards: wave p
  // works for column p % NG, one element of the column per thread.
  const int NSW = WAVES;                                   // every wave has partial sums (the chain waves' share of the rows): [w][k][128] from the bottom of the LDS
  const int NWK = (WAVES - NG) >> lgNG;                    // workers per column (15 / 7 / 3)
  const int wk = vw - 1;                                   // worker index of a non-chain wave: p = NG + wk NG + cg
  const int first_worker = NG;                             // it fetches the eigen-system for everybody
  const DfLayout D = df_layout(T, K, S);
  const DfWork Wk = df_work(T, K, S);
  double* mailbox = lds + vf_red_doubles(false) - VF_MAILBOX;
  unsigned* cw = reinterpret_cast<unsigned*>(mailbox + 96);
  unsigned* cgw = cw + DFC_GROUP0 + cg * DFC_PER_GROUP;
  double* top = lds + vf_red_doubles(false) - VF_MAILBOX - (size_t)(cg + 1) * D.size;
  double* P = top + D.P;
  double* Pm = top + D.Pm;
  double* Q = top + D.Q;                                   // [k][pivot][l1 l2 l3 | 1/D -> w -> x]
  double* win = top + D.win;
  double* flag = top + D.flag;
  double* base = lds + (size_t)NSW * K * ACC_TILE + (size_t)cg * Wk.stride;      // the columns' working arrays: BEHIND the partial sums
  double* mraw = base + Wk.mraw;
  double* mt = base + Wk.mt;
  double* rz = base + Wk.rz;
  double* head = base + Wk.head;
  const double* gsh = mailbox;                             // eigenvalues, eigenvectors of W'W as published
  const double* Ush = mailbox + K;
  if (a.hyp && a.hyp_noise) {
    const double nu2 = a.hyp[HYP_NU2];
    a.s = 1.0 / nu2; a.sR = a.s * a.Rrep;
  }

  if (vw == 0) {
    // =========================== the column's chain wave ===========================
    // (its share of the stream has ended: the partial sums, one count - also for a dead column group: its rows count)
#pragma unroll
    for (int v = 0; v < K; ++v)
      *reinterpret_cast<double2*>(&lds[((size_t)pw * K + v) * ACC_TILE + 2 * lane]) = make_double2(acc[v][0], acc[v][1]);
    df_signal(cw + DFC_PART);
    if (!live) return;
    const int nchain = nr > 0 ? 2 * K : K;
    const bool chain = lane < nchain;
    const int side = lane >= K ? 1 : 0;
    const int k = chain ? lane - side * K : 0;
    // the column's normals, raw, beside the places of the right-hand sides (one Philox block, one logarithm, one sincos
    // per PAIR z[2m], z[2m+1]: philox_normal_pair, the bits of two philox_normal calls; n is even or the last odd element
    // has a partner inside the column's record): the side workgroup needs ~7 us for the eigenvalues anyway
    for (int i0 = 2 * lane; i0 < n; i0 += 2 * WAVE) {
      double z0, z1;
      if (a.z) { const double2 zz = *reinterpret_cast<const double2*>(a.z + (size_t)jg * n + i0); z0 = zz.x; z1 = zz.y; }
      else philox_normal_pair(a.seed, a.stream, ((unsigned long long)jg * n + i0) >> 1, z0, z1);
      rz[(size_t)i0 * 2 + 1] = z0;
      if (i0 + 1 < n) rz[(size_t)(i0 + 1) * 2 + 1] = z1;
    }
    // the eigenvalue of the lane's system straight from the side workgroup's tagged granules {half of g_k, epoch}: value and
    // "it is this launch's" in ONE round trip
    unsigned long long ghi = 0ULL, glo = 0ULL;
    bool okw = false;
    for (unsigned spins = 0; !okw && spins < (1u << 20); ++spins) {
      ghi = __hip_atomic_load(fv.eig_gran + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      glo = __hip_atomic_load(fv.eig_gran + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      okw = __all((unsigned)ghi == fv.epoch && (unsigned)glo == fv.epoch) != 0;
      if (!okw) __builtin_amdgcn_s_sleep(4);
    }
    if (!okw && lane == 0) {
      __hip_atomic_store(cw + DFC_BAD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (atomicCAS(&a.status[0], 0, 2) == 0) a.status[1] = -1;
    }
    const double g0 = okw ? __longlong_as_double((long long)(((ghi >> 32) << 32) | (glo >> 32))) : 1.0;
    df_wait(cgw + DFG_BAND, 1u, cw + DFC_BAD);             // (the band image was stored early in the stream: long there)
    TAIL_STAMP(stamps, 3);                                 // (diagnostic builds: the eigenvalues are here, the factorisation starts)
    const double* Pv = side ? Pm : P;
    double* cq = Q + (size_t)(k * T + (side ? nl : 0)) * 4;
    const int n_elim = side ? nr : nl;
    const int n_common = nr > 0 ? (nl < nr ? nl : nr) : nl;
    double shift = 0.0, eps = a.eps0;
    int tried = 0;
    bool ok;
    double gk;
    while (true) {
      bool good = true;
      gk = fma(g0, a.sR, shift);
      {
        // factor: window c[b][d] = A[i+b+d][i+b]; per pivot the new band row in, the quad [l1 l2 l3 1/D] out
        // (the lanes outside the chains stay out of the pivots, not out of the eigen-system probe between them)
        double c[S + 1][S + 1];
#pragma unroll
        for (int b = 0; b < S; ++b) {
          const double2 p0 = *reinterpret_cast<const double2*>(Pv + b * 4), p1 = *reinterpret_cast<const double2*>(Pv + b * 4 + 2);
          c[b][0] = p0.x + gk; c[b][1] = p0.y; c[b][2] = p1.x; c[b][3] = p1.y;
        }
        bool bad = false;
        auto pivot = [&](int i) {
          const double2 p0 = *reinterpret_cast<const double2*>(Pv + (i + S) * 4), p1 = *reinterpret_cast<const double2*>(Pv + (i + S) * 4 + 2);
          c[S][0] = p0.x + gk; c[S][1] = p0.y; c[S][2] = p1.x; c[S][3] = p1.y;
          const double d0 = c[0][0];
          bad |= !(d0 > 0.0);
          const double inv = rcp_cubic(d0);
          double l[S + 1];
#pragma unroll
          for (int d = 1; d <= S; ++d) l[d] = c[0][d] * inv;
#pragma unroll
          for (int b = 1; b <= S; ++b)
#pragma unroll
            for (int aa = b; aa <= S; ++aa) c[b][aa - b] = fma(-l[aa], c[0][b], c[b][aa - b]);
          *reinterpret_cast<double2*>(cq + i * 4) = make_double2(l[1], l[2]);
          *reinterpret_cast<double2*>(cq + i * 4 + 2) = make_double2(l[3], inv);
#pragma unroll
          for (int b = 0; b < S; ++b)
#pragma unroll
            for (int d = 0; d <= S; ++d) c[b][d] = c[b + 1][d];
        };
        if (chain) {
#pragma unroll 4
          for (int i = 0; i < n_common; ++i) pivot(i);
          if (n_elim > n_common) pivot(n_common);
        }
        good = !bad;
        if (chain && ns > 0) {                             // park the window for the separator system
          double* wp = win + (size_t)(side * K + k) * WN;
#pragma unroll
          for (int b = 0; b < S; ++b)
#pragma unroll
            for (int d = 0; d <= S; ++d) wp[b * (S + 1) + d] = c[b][d];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (ns > 0 && lane < K) {
        // separator system: both halves' Schur complements carry the original entries once too often
        const double* wl = win + (size_t)k * WN;
        const double* wr = win + (size_t)(K + k) * WN;
        double Sg[S][S];
#pragma unroll
        for (int b = 0; b < S; ++b)
#pragma unroll
          for (int aa = b; aa < S; ++aa)
            Sg[aa][b] = wl[b * (S + 1) + aa - b] + wr[(S - 1 - aa) * (S + 1) + aa - b] - (P[(nl + b) * D1 + aa - b] + (aa == b ? gk : 0.0));
        double* sq = Q + (size_t)(k * T + nl + nr) * 4;
#pragma unroll
        for (int cc = 0; cc < S; ++cc) {
          const double d0 = Sg[cc][cc];
          good &= d0 > 0.0;
          const double inv = rcp_cubic(d0);
#pragma unroll
          for (int d = 1; d <= S; ++d) {
            double l = 0.0;
            if (cc + d < S) {
              l = Sg[cc + d][cc] * inv;
#pragma unroll
              for (int b2 = 1; b2 <= d; ++b2) Sg[cc + d][cc + b2] = fma(-l, Sg[cc + b2][cc], Sg[cc + d][cc + b2]);
            }
            sq[cc * 4 + d - 1] = l;
          }
          sq[cc * 4 + 3] = inv;
        }
      }
      ok = __builtin_amdgcn_readfirstlane(__ballot(!good) == 0ULL ? 1 : 0) != 0;
      if (ok || tried >= a.attempts) break;
      shift += eps;   // fast_mvn.py:64-68: cumulative eps, eps *= 10
      eps *= 10.0;
      ++tried;
    }
    ok = ok && okw;
    if (lane == 0) {
      flag[0] = ok ? 1.0 : 0.0;
      flag[1] = (double)tried;
      a.tries[j] = tried;
      if (!ok && okw && atomicCAS(&a.status[0], 0, 1) == 0) a.status[1] = jg;
    }
    TAIL_STAMP(stamps, 4);
    // z sqrt(1 / D), all lanes (pivot order: z[j][k T + i] multiplies pivot i of system k): still under the others' stream
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int idx = lane; idx < n; idx += WAVE) rz[(size_t)idx * 2 + 1] = rz[(size_t)idx * 2 + 1] * sqrt(Q[(size_t)idx * 4 + 3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- the right-hand sides meet the factors: forward substitution with w folded in, separator, back-substitution ----
    df_wait(cgw + DFG_ROT, (unsigned)NWK, cw + DFC_BAD);
    TAIL_STAMP(stamps, 5);                                 // (the workers' right-hand sides are in)
    if (ok) {
      double r[S + 1];
      if (chain) {
        // u(i) = r(i) as updated by the pivots before it (ascending, as spectral_pivot updates them); w(i) = u(i) / D(i) + z sqrt(1 / D(i))
        const double* hd = head + (size_t)(side * K + k) * 4;
        const double* pz = rz + (size_t)(k * T + (side ? nl : 0)) * 2;
        {
          const double2 h0 = *reinterpret_cast<const double2*>(hd);
          r[0] = h0.x; r[1] = h0.y; r[2] = hd[2];
        }
        auto fstep = [&](int i) {
          const double2 q0 = *reinterpret_cast<const double2*>(cq + i * 4), q1 = *reinterpret_cast<const double2*>(cq + i * 4 + 2);
          const double2 z2 = *reinterpret_cast<const double2*>(pz + i * 2);
          r[S] = z2.x;
          const double u = r[0];
          r[1] = fma(-q0.x, u, r[1]);
          r[2] = fma(-q0.y, u, r[2]);
          r[3] = fma(-q1.x, u, r[3]);
          cq[i * 4 + 3] = fma(u, q1.y, z2.y);
#pragma unroll
          for (int b = 0; b < S; ++b) r[b] = r[b + 1];
        };
#pragma unroll 4
        for (int i = 0; i < n_common; ++i) fstep(i);
        if (n_elim > n_common) fstep(n_common);
        if (ns > 0) {
          double* wp = win + (size_t)(side * K + k) * WN;
#pragma unroll
          for (int b = 0; b < S; ++b) wp[S * (S + 1) + b] = r[b];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (ns > 0 && lane < K) {
        const double* wl = win + (size_t)k * WN;
        const double* wr = win + (size_t)(K + k) * WN;
        double* sq = Q + (size_t)(k * T + nl + nr) * 4;
        const double* sz = rz + (size_t)(k * T + nl + nr) * 2;      // [rotated right-hand side at depth nl + c, scaled normal of separator pivot c]
        double us[S];
#pragma unroll
        for (int b = 0; b < S; ++b) us[b] = wl[S * (S + 1) + b] + wr[S * (S + 1) + S - 1 - b] - sz[b * 2];
#pragma unroll
        for (int cc = 0; cc < S; ++cc) {
#pragma unroll
          for (int d = 1; d <= S; ++d)
            if (cc + d < S) us[cc + d] = fma(-sq[cc * 4 + d - 1], us[cc], us[cc + d]);
          sq[cc * 4 + 3] = fma(us[cc], sq[cc * 4 + 3], sz[cc * 2 + 1]);
        }
        // ... and its S x S unit upper solve
#pragma unroll
        for (int cc = S - 1; cc >= 0; --cc) {
          double acc2 = sq[cc * 4 + 3];
#pragma unroll
          for (int d = 1; d <= S; ++d) if (cc + d < S) acc2 = fma(-sq[cc * 4 + d - 1], sq[(cc + d) * 4 + 3], acc2);
          sq[cc * 4 + 3] = acc2;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (chain) {
        double x1 = 0.0, x2 = 0.0, x3 = 0.0;
        if (ns > 0) {
          const double* sq = Q + (size_t)(k * T + nl + nr) * 4;
          x1 = sq[(side ? S - 1 : 0) * 4 + 3]; x2 = sq[(side ? S - 2 : 1) * 4 + 3]; x3 = sq[(side ? S - 3 : 2) * 4 + 3];
        }
        // x(i) = w(i) - sum_d L[i+d, i] x(i+d), the term of x(i+1) last (spectral_backward's order)
#pragma unroll 4
        for (int i = n_elim - 1; i >= 0; --i) {
          const double2 q0 = *reinterpret_cast<const double2*>(cq + i * 4), q1 = *reinterpret_cast<const double2*>(cq + i * 4 + 2);
          double acc2 = q1.y;
          acc2 = fma(-q1.x, x3, acc2);
          acc2 = fma(-q0.y, x2, acc2);
          acc2 = fma(-q0.x, x1, acc2);
          x3 = x2; x2 = x1; x1 = acc2;
          cq[i * 4 + 3] = acc2;
        }
      }
    }
    df_signal(cgw + DFG_X);
    TAIL_STAMP(stamps, 6);
  } else {
    // =========================== the other waves: stream ended ===========================
    // the wave's partial sums of the tile's 128 outputs (both lanes' pairs), then one count
#pragma unroll
    for (int v = 0; v < K; ++v)
      *reinterpret_cast<double2*>(&lds[((size_t)pw * K + v) * ACC_TILE + 2 * lane]) = make_double2(acc[v][0], acc[v][1]);
    df_signal(cw + DFC_PART);
    if (pw == first_worker) {
      // the eigen-system for the rotations (K + K K doubles, one per lane; group 0 always exists): this wave asked for it
      // inside its stream - the flag's word behind one row group's loads, the payload behind a later one's once the flag
      // had come back as this launch's (accum_kernel) - so it normally costs no round trip here
      double epub = early.epub;
      bool okw = true;
      if (!early.have) {
        if (lane == 0) okw = poll_flag(fv.eig_flag, fv.epoch);
        okw = __builtin_amdgcn_readfirstlane(okw ? 1 : 0) != 0;
        if (okw && lane < K + K * K) epub = load_sc1(fv.eig_pub + lane);
      }
      if (okw) { if (lane < K + K * K) mailbox[lane] = epub; }
      else if (lane == 0) __hip_atomic_store(cw + DFC_BAD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      df_signal(cw + DFC_EIG);
    }
    // this thread's element of the column (n <= NWK 64: vf_df_fits); T is 32, 64 or 128
    const int lgT = 7 - lgNG;
    const int e = wk * WAVE + lane;
    const bool mine = live && e < n;
    const int ek = mine ? e >> lgT : 0, et = mine ? e & (T - 1) : 0;
    const double* psum = lds + (size_t)ek * ACC_TILE + cg * T + et;
    // the column sums: every wave's partials are in (the slowest streaming wave decides), fixed order over the waves
    df_wait(cw + DFC_PART, (unsigned)NSW, cw + DFC_BAD);
    if (stamps && pw == first_worker && lane == 0) stamps[1] = wall_clock64();      // (diagnostic builds: the slowest wave's stream has ended)
    if (!live) return;
    if (mine) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) s += psum[(size_t)w * K * ACC_TILE];
      mraw[et * K + ek] = 0.0 + s;
    }
    df_signal(cgw + DFG_IN);
    df_wait(cgw + DFG_IN, (unsigned)NWK, cw + DFC_BAD);
    df_wait(cw + DFC_EIG, 1u, cw + DFC_BAD);
    // rotated right-hand sides, delivered where the chains will read them: r(t) of system k is r(i + S) of the ascending
    // chain's pivot i = t - S (its first S values go to the head), of the descending chain's pivot i = T-1-t - S, and the
    // separator's right-hand side at the depths nl .. nl + S - 1
    if (mine) {
      const int k = ek, t = et;
      double s = 0.0;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) s = fma(Ush[kk * K + k], mraw[t * K + kk], s);
      s *= a.s;
      mt[k * Tp + t] = s;
      const int m = T - 1 - t;
      if (t < S) head[(size_t)k * 4 + t] = s; else if (t - S < nl) rz[(size_t)(k * T + t - S) * 2] = s;
      if (nr > 0) { if (m < S) head[(size_t)(K + k) * 4 + m] = s; else if (m - S < nr) rz[(size_t)(k * T + nl + m - S) * 2] = s; }
      if (ns > 0 && t >= nl && t < nl + S) rz[(size_t)(k * T + nl + nr + t - nl) * 2] = s;
    }
    df_signal(cgw + DFG_ROT);
    if (vw >= VF_GROUP_WAVES) return;
    df_wait<8>(cgw + DFG_X, 1u, cw + DFC_BAD);
  }
  // =========================== final stage: the column's four virtual waves ===========================
  // rotate back, write V[j] (depth-major), residual part, Gram share - v_spectral_kernel's geometry (tid = vw 64 + lane)
  const bool bad = __hip_atomic_load(cw + DFC_BAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u;
  const bool ok = flag[0] != 0.0 && !bad;
  const int tid = vw * WAVE + lane;
  double* xout = mraw;
  double sse_acc = 0.0;
  const double inv_s2 = -2.0 / a.s;
  const double* xs = Q + 3;
  if (ok) {
    for (int idx = tid; idx < n; idx += NT) {
      const int t = idx / K, k = idx - t * K;
      const int pos = t < nl ? t : (t < nl + ns ? nl + nr + (t - nl) : nl + (T - 1 - t));
      double s = 0.0;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) s = fma(Ush[k * K + kk], xs[(size_t)(kk * T + pos) * 4], s);
      xout[idx] = s;
      a.V[(size_t)jg * n + idx] = s;
      if (a.sse_out) {
        const double xt = xs[(size_t)(k * T + pos) * 4];
        sse_acc = fma(xt, fma(a.Rrep * gsh[k], xt, inv_s2 * mt[k * Tp + t]), sse_acc);
      }
    }
  }
  if (a.sse_out) {
    const double v = wave_sum(sse_acc);
    if (lane == 0) flag[2 + vw] = v;
    if (!a.gout) {                                           // (with a Gram share behind it, its first count covers the parts too)
      df_signal(cgw + DFG_SSE);
      if (vw == 0) {
        df_wait(cgw + DFG_SSE, (unsigned)VF_GROUP_WAVES, cw + DFC_BAD);
        if (ok && lane == 0) a.sse_out[j] = (flag[2] + flag[3]) + (flag[4] + flag[5]);
      }
    }
  }
  if (a.gout) {
    df_signal(cgw + DFG_G1);
    df_wait(cgw + DFG_G1, (unsigned)VF_GROUP_WAVES, cw + DFC_BAD);      // the fresh column stands in xout (and the residual parts in flag[])
    if (a.sse_out && vw == 0 && ok && lane == 0) a.sse_out[j] = (flag[2] + flag[3]) + (flag[4] + flag[5]);
    int ng = NT / KK;
    if (ng > 16) ng = 16;
    if (ng < 1) ng = 1;
    const int g = tid / KK, q = tid - g * KK;
    int p = 0;
    while ((p + 1) * (p + 2) / 2 <= q) ++p;
    const int pq = q - p * (p + 1) / 2;
    double* scratch = base + Wk.gs;
    if (ok && g < ng) {
      double s = 0.0;
      for (int t = g; t < T; t += ng) s = fma(xout[t * K + p], xout[t * K + pq], s);
      scratch[g * KK + q] = s;
    }
    df_signal(cgw + DFG_G2);
    if (tid < KK) {                                          // (virtual wave 0: KK <= 36 < 64)
      df_wait(cgw + DFG_G2, (unsigned)VF_GROUP_WAVES, cw + DFC_BAD);
      if (ok) {
        double s = 0.0;
        for (int b = 0; b < ng; ++b) s += scratch[b * KK + tid];
        a.gout[(size_t)j * KK + tid] = s;
      }
    }
  }
  if (vw == 0) TAIL_STAMP(stamps, 7);
}

}  // namespace btf
