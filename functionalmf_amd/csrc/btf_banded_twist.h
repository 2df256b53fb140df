// Twisted ("burn at both ends") V half-sweep sampler - the default (BTF_K_V_BANDED).
//
// The pivot chain of the banded LDL' is sequential, and a lone wave needs ~430 cycles per
// pivot (btf_banded_fast.h).  The precision of a column couples depth t only with t +- (tf+1),
// so a separator of S = tf+1 consecutive depths [ts, ts+S) splits the unknowns into two
// blocks that do not touch: wave 0 eliminates depths 0..ts-1 downwards, wave 1 eliminates
// depths T-1..ts+S upwards (the same kernel on the mirrored indices), concurrently and with
// no synchronisation per pivot; their Schur complements meet in the small dense separator
// system (S*K = half-bandwidth unknowns), which wave 0 factors.  Back-substitution runs the
// other way: separator first, then the two halves in parallel.  The sequential chain is
// ~n/2 + bw pivots instead of n.
//
// This is the LDL' of P Q P' for the ordering
//     [ depths 0..ts-1 ascending | depths T-1..ts+S descending (k descending) | separator ]
// and the draw is  x = Q^-1 mu + P' L^-T D^-1/2 z  with z indexed in THAT order - the
// ordering the build declares for this kernel (btf_get_V_order reports it; the oracle takes
// it as `perm`).  Same distribution and same mean term as any other ordering
// ("parity unpinned" note on CHOLMOD's own ordering).
#pragma once
#include "btf_banded_fast.h"

namespace btf {

constexpr int VT_THREADS = 256;
// An entry of the band assembly program as the kernel reads it: 8 bytes {dst | dia16 << 16, src} - dst and the diagonal's
// source are LDS word offsets below 2^16 (0xffff: no diagonal term), src is an LDS word offset or the negative code of a
// likelihood-block entry in the accumulation partials (tw_layout, weighted == 2).  (16-byte entries until round 4: the
// 256 workgroups of a C3 launch pulled 12.5 MB of them through the L2s in the cold batch of loads the kernel starts with.)
__device__ __forceinline__ void tw_fill_unpack(const int2 f, int& dst, int& src, int& dia) {
  dst = f.x & 0xffff;
  const int d16 = (int)((unsigned)f.x >> 16);
  dia = d16 == 0xffff ? -1 : d16;
  src = f.y;
}
#ifndef BTF_TWIST_BACKPAR
#define BTF_TWIST_BACKPAR 0       // 1: parallel block solves (backpar16_prepare) + four MFMAs per 16 columns on the chain (backpar16_chain_mfma).
                                  // Round 3: the chain itself shrinks to ~2 k cycles, but the 2 x 11 x 17 triangular solves of the prepare
                                  // phase cost ~18 k (LDS-latency-bound) against 14.4 k for the whole column-by-column routine; hiding
                                  // them under the forward chain (two idle waves) would need 27 k of the chain's 43 k cycles and still
                                  // leave the blocks next to the separator on the critical path - kept for A/B, not shipped
#endif

struct TwLayout {
  VbLayout L, R, S;      // left chain, right chain, separator: views for the shared routines
  int m0, zs, P, Ql, flag, itau;
  int total;
  int ts, nl, nr, ns, nL, nR;
};

__host__ __device__ inline int twist_ts(int T, int TF) { return (T - (TF + 1)) / 2; }
__host__ __device__ inline bool twist_ok(int T, int K, int TF) {
  return (TF + 1) * K >= 3 && T >= 2 * (TF + 1) + 2;
}

__host__ __device__ inline TwLayout tw_layout(int T, int K, int TF, int weighted) {
  TwLayout W;
  // band stride: 16 words for every bw <= 15 (the unused tail of a column stays zero), so that the panelised
  // MFMA factorisation - written for 16-word columns - serves all of them; bw + 1 beyond
  const int n = T * K, S = TF + 1, bw = S * K, D1 = TF + 2, KK = tri(K), R1 = bw <= 15 ? 16 : bw + 1;
  W.ts = twist_ts(T, TF);
  W.ns = bw;
  W.nl = W.ts * K;
  W.nr = n - W.nl - W.ns;
  W.nL = W.nl + W.ns;
  W.nR = W.nr + W.ns;
  int o = 0;
  auto carve = [&](VbLayout& V, int nn) {
    V.R1 = R1;
    V.npad = nn + bw + 2;
    V.FP = bw + 4;
    V.band = o; o += V.npad * R1 + 64;
    V.rhs = o + V.FP; o += V.FP + V.npad + 64;
    V.invd = o; o += V.npad;
    V.vsc = o; o += 64;
    V.vs4 = o; o += 64;
    V.m0 = V.zs = V.P = V.Ql = V.flag = 0;
    V.total = 0;
  };
  carve(W.L, W.nL);
  carve(W.R, W.nR);
  carve(W.S, W.ns);
  // one dummy region for all views: idle lanes only ever park garbage there (races between the
  // two concurrent chains are harmless), and its last word is never written (the zero source)
  W.L.dummy = W.R.dummy = W.S.dummy = o; o += 64 * 9 + 8;
  W.m0 = o; o += n;
  W.zs = o; o += n;
  W.P = o; o += T * D1;
  // weighted == 2: the per-depth likelihood blocks are NOT staged in LDS - the band assembly program fetches each of them
  // from the accumulation partials where it needs it (fill entries with a negative source) - for the shapes whose
  // T K(K+1)/2 blocks are what keeps the layout from fitting (nembeds 8 at 64 depths: 180 -> 162 KB)
  W.Ql = o; o += weighted == 1 ? T * KK : (weighted == 2 ? 0 : KK);
  W.flag = o; o += 8;
  W.itau = o; o += (TF + 2) * T;            // 1 / (lam2 Tau2_jr) per penalty row (at most (tf+2) T rows)
  W.total = o;
  return W;
}
__host__ __device__ inline size_t tw_lds_bytes(int T, int K, int TF, int weighted) {
  return (size_t)tw_layout(T, K, TF, weighted).total * sizeof(double);
}

// position i in the elimination order -> depth-major index g = t*K + k
__host__ __device__ inline int twist_order(int i, int n, int nl, int nr) {
  if (i < nl) return i;
  if (i < nl + nr) return n - 1 - (i - nl);
  return nl + (i - nl - nr);
}

// pairs-per-lane count and band stride of a (K, tf) shape (what dispatch_vbanded_twist derives at run time)
__host__ __device__ constexpr int tw_npl(int K, int TF) {
  const int bw = (TF + 1) * K, np = ((bw - 1) * (bw - 2) / 2 + WAVE - 1) / WAVE;
  return np < 1 ? 1 : np;
}
__host__ __device__ constexpr bool tw_row16(int K, int TF) { return (TF + 1) * K <= 15; }

// KC > 0: nembeds and the trend-filter order as compile-time constants (the instances of BTF_TWIST_SET for the reference's
// default tf_order = 2): the setup's address arithmetic folds - 234 -> 109 spilled SGPRs, 37.0 -> 35.2 us at C3
// TC > 0: the depth axis as a compile-time constant too (round 4: the instance of BASELINE configs 3 / 4, ndepth 64) - every
// offset of tw_layout folds into the instructions
template <int NPL, bool ROW16, int KC = 0, int TFC = 0, int TC = 0>
__global__ __launch_bounds__(VT_THREADS) void v_banded_twist_kernel(VBandArgs a, int K) {
  if constexpr (KC > 0) { K = KC; a.TF = TFC; }
  if constexpr (TC > 0) a.T = TC;
  vband_load_hyp(a);
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  if (!lds_base_is_zero(lds)) {      // ldsr / ldsw address LDS absolutely (see btf_banded_fast.h); never taken
    if (tid == 0) { a.status[0] = 1; a.status[1] = -7; }
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = blockIdx.x, jg = a.col0 + j;
  const int jq = a.gsrc ? a.gsrc[j * a.T] / a.T : j;      // (stale cached weights: the column whose Gram blocks this one reuses - VBandArgs.gsrc)
  const int KK = tri(K), T = a.T, n = T * K, D1 = a.TF + 2, bw = (a.TF + 1) * K, R1 = bw <= 15 ? 16 : bw + 1;   // = tw_layout's stride
  const int NV = a.weighted ? K + KK : K;
  const bool ql_global = a.weighted && a.ql_global;          // (see tw_layout: weighted == 2)
  const TwLayout W = tw_layout(T, K, a.TF, a.weighted ? (ql_global ? 2 : 1) : 0);
  const int nl = W.nl, nr = W.nr, ns = W.ns, nL = W.nL, nR = W.nR;
  double* m0 = lds + W.m0;
  double* zs = lds + W.zs;
  double* P = lds + W.P;
  double* Ql = lds + W.Ql;
  double* flag = lds + W.flag;
  long long stamp[6];
  stamp[0] = __builtin_amdgcn_s_memtime();

  // ---- likelihood mean part / Gram blocks / prior band ---------------------------------------
  auto chunk_sum = [&](const double* p) -> double {
    double s = 0.0;
    const size_t st = (size_t)NV * a.ld;
    int c = 0;
    for (; c + 4 <= a.nch; c += 4) {
      const double x0 = p[(size_t)c * st], x1 = p[(size_t)(c + 1) * st], x2 = p[(size_t)(c + 2) * st], x3 = p[(size_t)(c + 3) * st];
      s += x0; s += x1; s += x2; s += x3;
    }
    for (; c < a.nch; ++c) s += p[(size_t)c * st];
    return s;
  };
  // band assembly program (host-made, btf_abi.hip:make_fill_table): this thread's first FILL_REG entries are
  // fetched now, under the latency of the partial sums
  constexpr int FILL_REG = 12;
  int fdst[FILL_REG], fsrc[FILL_REG], fdia[FILL_REG];
  const int nfe = a.fill ? a.nfill / VT_THREADS : 0;          // entries per thread
  if (a.fill) {
#pragma unroll
    for (int u = 0; u < FILL_REG; ++u) {
      const int e = (u < nfe ? u : 0) * VT_THREADS + tid;
      tw_fill_unpack(reinterpret_cast<const int2*>(a.fill)[e], fdst[u], fsrc[u], fdia[u]);      // one 8-B load
    }
  }
  // prior band built here (no prior_band_kernel launch): 1 / (lam2 Tau2) per penalty row into LDS, and the fixed-slot
  // stencil of this thread's band entry (t, d) = tid into registers - all in the same round trip as the loads below
  const bool fuse_prior = a.pband == nullptr;
  double* itau = lds + W.itau;
  const bool tau_one = fuse_prior && a.nD <= VT_THREADS;
  double tau_reg = 1.0;
  int prow[PB_MAXE], pcnt = 0;
  double pcf[PB_MAXE], pacc = 0.0;
  if (fuse_prior) {
    // (one penalty row per thread: only the load is issued here - dividing and storing now would put a whole global
    //  round trip in front of the loads below; the reciprocal is formed after them)
    if (tau_one) { if (tid < a.nD) tau_reg = a.Tau2[(size_t)jg * a.nD + tid]; }
    else for (int r = tid; r < a.nD; r += VT_THREADS) itau[r] = 1.0 / (a.lam2 * a.Tau2[(size_t)jg * a.nD + r]);
    if (tid < T * D1) {
      pcnt = a.st_ptr[tid + 1] - a.st_ptr[tid];
#pragma unroll
      for (int u = 0; u < PB_MAXE; u += 4) {
        const int4 r4 = *reinterpret_cast<const int4*>(a.st_drow + (size_t)tid * PB_MAXE + u);
        prow[u] = r4.x; prow[u + 1] = r4.y; prow[u + 2] = r4.z; prow[u + 3] = r4.w;
      }
#pragma unroll
      for (int u = 0; u < PB_MAXE; u += 2) {
        const double2 c2 = *reinterpret_cast<const double2*>(a.st_dcoef + (size_t)tid * PB_MAXE + u);
        pcf[u] = c2.x; pcf[u + 1] = c2.y;
      }
    }
  }
  double gx[8];
  const bool g_early = !a.weighted && gram_early_ok(a.ngp, KK);     // Gram partials: fetched now, summed below
  if (g_early) reduce_gram_fetch(a.gpart, a.ngp, KK, gx);
  // per-depth likelihood blocks of weighted data, element e = q*T + t (see the general loop below)
  const bool ql_now = a.weighted && !ql_global && n <= 2 * VT_THREADS && T * KK <= 4 * VT_THREADS;
  const double* qp[4];
  int qdst[4];
  bool qhas[4];
  double qacc[4] = {0.0, 0.0, 0.0, 0.0};
  if (ql_now) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = u * VT_THREADS + tid;
      qhas[u] = e < T * KK;
      const int ec = qhas[u] ? e : 0;
      const int q = ec / T, t = ec - q * T;
      qp[u] = a.part + (size_t)(K + q) * a.ld + (size_t)jq * T + t;
      qdst[u] = t * KK + q;
    }
  }
  if (n <= 2 * VT_THREADS) {   // both elements of a thread at once: one round of global-load latency, not two
    // element e = k*T + t: consecutive lanes read consecutive depths t of one factor row k (coalesced 8-B
    // words of the partials; the depth-major scatter m0[t*K + k] happens on the LDS side)
    const int i0 = tid, i1 = tid + VT_THREADS;
    const bool h0 = i0 < n, h1 = i1 < n;
    const int e0 = h0 ? i0 : 0, e1 = h1 ? i1 : e0;       // clamped: idle slots re-read a valid word
    const int k0 = e0 / T, t0 = e0 - k0 * T, k1 = e1 / T, t1 = e1 - k1 * T;
    const double* p0 = a.part + (size_t)k0 * a.ld + (size_t)j * T + t0;
    const double* p1 = a.part + (size_t)k1 * a.ld + (size_t)j * T + t1;
    const size_t st = (size_t)NV * a.ld;
    // (the prior band of this column rides in the same batch of loads: issued any earlier, the register-starved
    //  compiler sinks it below the Gram reduction and a second global round trip shows up)
    double pb_reg[4];
    if (!fuse_prior) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = tid + u * VT_THREADS;
        pb_reg[u] = idx < T * D1 ? a.pband[(size_t)j * T * D1 + idx] : 0.0;
      }
    }
    double s0 = 0.0, s1 = 0.0;
    int c = 0;
    if (ql_now) {
      // weighted data whose per-depth blocks fit one pass (T*KK <= 4 per thread): their chunks ride in the same batches
      // of loads as the mean part's (one global round trip less; same sums, c ascending)
      for (; c + 2 <= a.nch; c += 2) {
        const double x0 = p0[(size_t)c * st], x1 = p0[(size_t)(c + 1) * st];
        const double y0 = p1[(size_t)c * st], y1 = p1[(size_t)(c + 1) * st];
        double q0[4], q1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { q0[u] = qp[u][(size_t)c * st]; q1[u] = qp[u][(size_t)(c + 1) * st]; }
        s0 += x0; s0 += x1;
        s1 += y0; s1 += y1;
#pragma unroll
        for (int u = 0; u < 4; ++u) { qacc[u] += q0[u]; qacc[u] += q1[u]; }
      }
      for (; c < a.nch; ++c) {
        double q0[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q0[u] = qp[u][(size_t)c * st];
        s0 += p0[(size_t)c * st]; s1 += p1[(size_t)c * st];
#pragma unroll
        for (int u = 0; u < 4; ++u) qacc[u] += q0[u];
      }
    }
    for (; c + 4 <= a.nch; c += 4) {                     // fixed order c ascending, as chunk_sum
      const double x0 = p0[(size_t)c * st], x1 = p0[(size_t)(c + 1) * st], x2 = p0[(size_t)(c + 2) * st], x3 = p0[(size_t)(c + 3) * st];
      const double y0 = p1[(size_t)c * st], y1 = p1[(size_t)(c + 1) * st], y2 = p1[(size_t)(c + 2) * st], y3 = p1[(size_t)(c + 3) * st];
      s0 += x0; s0 += x1; s0 += x2; s0 += x3;
      s1 += y0; s1 += y1; s1 += y2; s1 += y3;
    }
    for (; c < a.nch; ++c) { s0 += p0[(size_t)c * st]; s1 += p1[(size_t)c * st]; }
    if (h0) m0[t0 * K + k0] = s0 * a.s;
    if (h1) m0[t1 * K + k1] = s1 * a.s;
    if (!fuse_prior) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = tid + u * VT_THREADS;
        if (idx < T * D1) P[idx] = pb_reg[u];
      }
      for (int idx = tid + 4 * VT_THREADS; idx < T * D1; idx += VT_THREADS) P[idx] = a.pband[(size_t)j * T * D1 + idx];
    }
  } else {
    for (int idx = tid; idx < n; idx += VT_THREADS) {
      const int t = idx / K, k = idx - t * K;
      m0[idx] = chunk_sum(a.part + (size_t)k * a.ld + (size_t)j * T + t) * a.s;
    }
    if (!fuse_prior)
      for (int idx = tid; idx < T * D1; idx += VT_THREADS) P[idx] = a.pband[(size_t)j * T * D1 + idx];
  }
  if (fuse_prior) {
    // prior band P[t][d] = sum_r Delta[r,t] Delta[r,t+d] / (lam2 Tau2_jr), rows ascending (factor.py:404-405): the
    // reciprocal once per penalty row, this thread's entry from its fixed-slot stencil (fetched above, with everything else)
    if (tau_one && tid < a.nD) itau[tid] = 1.0 / (a.lam2 * tau_reg);
    __syncthreads();                                       // itau complete
#pragma unroll
    for (int u = 0; u < PB_MAXE; ++u) if (u < pcnt) pacc = fma(pcf[u], itau[prow[u]], pacc);
    if (tid < T * D1) P[tid] = pacc;
    for (int idx = tid + VT_THREADS; idx < T * D1; idx += VT_THREADS) {
      double sacc = 0.0;
      for (int e = a.st_ptr[idx]; e < a.st_ptr[idx + 1]; ++e) sacc = fma(a.st_coef[e], itau[a.st_row[e]], sacc);
      P[idx] = sacc;
    }
  }
  if (ql_now) {
#pragma unroll
    for (int u = 0; u < 4; ++u) if (qhas[u]) Ql[qdst[u]] = qacc[u] * a.s;
  } else if (ql_global) {
    // (nothing staged: the assembly reads the blocks from the partials)
  } else if (a.weighted) {
    // per-depth likelihood blocks: element e = q*T + t, so that consecutive lanes read consecutive depths of one
    // Gram entry (coalesced 8-B words; the scatter to Ql[t*KK + q] is on the LDS side), four elements' chunks in
    // flight per thread and trip (one or two global round trips for the 960 entries of C3 instead of four)
    const int tot = T * KK;
    const size_t st = (size_t)NV * a.ld;
    for (int base = 0; base < tot; base += 4 * VT_THREADS) {
      const double* pp[4];
      int dst[4];
      bool has[4];
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = base + u * VT_THREADS + tid;
        has[u] = e < tot;
        const int ec = has[u] ? e : 0;
        const int q = ec / T, t = ec - q * T;
        pp[u] = a.part + (size_t)(K + q) * a.ld + (size_t)jq * T + t;
        dst[u] = t * KK + q;
      }
      int c = 0;
      for (; c + 2 <= a.nch; c += 2) {                  // fixed order, c ascending
        double x0[4], x1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { x0[u] = pp[u][(size_t)c * st]; x1[u] = pp[u][(size_t)(c + 1) * st]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc[u] += x0[u]; acc[u] += x1[u]; }
      }
      for (; c < a.nch; ++c) {
        double x0[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x0[u] = pp[u][(size_t)c * st];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += x0[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) if (has[u]) Ql[dst[u]] = acc[u] * a.s;
    }
  } else if (g_early) {
    reduce_gram_finish(gx, a.ngp, KK, a.sR, lds + W.L.band, Ql);
    curve_column_gram(a.cv, a.cv_W, jg, K, KK, a.s, Ql, lds + W.L.band, VT_THREADS * 2);
  } else {
    reduce_gram(a.gpart, a.ngp, KK, a.sR, lds + W.L.band, Ql);
    curve_column_gram(a.cv, a.cv_W, jg, K, KK, a.s, Ql, lds + W.L.band, VT_THREADS * 2);
  }
  // (the static zero regions of the three views - pads, scratch rows, dummy words - are part of the one wide
  //  zero fill at the top of the assembly below: the views and the dummy block are contiguous in LDS)
  __syncthreads();
  stamp[1] = __builtin_amdgcn_s_memtime();

  // entry (g + aa, g) of the precision, g = t*K + k depth-major, 0 <= aa <= bw
  auto qentry = [&](int g, int aa, double shift) -> double {
    const int t = g / K, k = g - t * K;
    if (aa < K - k) {
      double v = a.weighted ? Ql[t * KK + lidx(k + aa, k)] : Ql[lidx(k + aa, k)];
      if (aa == 0) v += P[t * D1] + shift;
      return v;
    }
    const int dd = aa / K;
    if (dd * K == aa && dd < D1 && t + dd < T) return P[t * D1 + dd];
    return 0.0;
  };

  double shift = 0.0, eps = a.eps0;
  int tried = 0;
  bool ok = false;
  while (true) {
    // ---- assemble the two bands -------------------------------------------------------------
    // left view: index i = g, rows/cols < nl+ns.  right view: index m = n-1-g (mirrored), rows/cols
    // >= nl, and its separator-by-separator block left at zero (it is counted once, on the left).
    // Only K-k (same depth block) + tf+1 (prior couplings) of the bw+1 entries of a band column
    // are non-zero: clear both bands with wide stores, then write just those.  LS lanes per
    // column (lane = slot), VT_THREADS/LS columns per pass; no divisions inside the loops.
    {
      double2* zb = reinterpret_cast<double2*>(lds + W.L.band);      // views L, R, S and the dummy block are contiguous
      const int nzw = W.L.dummy + 64 * 9 + 8 - W.L.band;            // words; m0 follows directly: no overshoot
      const int nz2 = nzw / 2;
      for (int idx = tid; idx < nz2; idx += VT_THREADS) zb[idx] = make_double2(0.0, 0.0);
      if ((nzw & 1) && tid == 0) lds[W.L.band + nzw - 1] = 0.0;
    }
    __syncthreads();
    if (a.fill) {
      // table-driven: every entry is an independent LDS read -> write (the loops below did one dependent
      // read -> write per column and trip, eleven trips per view, with one wave per SIMD to hide it)
      if (!ql_global) {
      double val[FILL_REG];
#pragma unroll
      for (int u = 0; u < FILL_REG; ++u) {
        val[u] = lds[fsrc[u]];
        if (fdia[u] >= 0) val[u] += lds[fdia[u]] + shift;
      }
#pragma unroll
      for (int u = 0; u < FILL_REG; ++u) if (u < nfe) lds[fdst[u]] = val[u];
      for (int u = FILL_REG; u < nfe; ++u) {               // (larger systems: the rest on demand)
        const int e = u * VT_THREADS + tid;
        int dst, src, dia;
        tw_fill_unpack(reinterpret_cast<const int2*>(a.fill)[e], dst, src, dia);
        double v = lds[src];
        if (dia >= 0) v += lds[dia] + shift;
        lds[dst] = v;
      }
      } else {
      // likelihood blocks not staged (tw_layout, weighted == 2): a negative source -2 - (q T + t) names entry q of depth
      // t's block in the accumulation partials - summed here, chunk by chunk, where it is written (a path of its own: the
      // loads in the unrolled form above cost the common case 3.5 us of registers and code)
      for (int u = 0; u < nfe; ++u) {
        const int e = u * VT_THREADS + tid;
        int dst, src, dia;
        tw_fill_unpack(reinterpret_cast<const int2*>(a.fill)[e], dst, src, dia);
        double v;
        if (src >= 0) v = lds[src];
        else {
          const int ge = -2 - src, q = ge / T, t = ge - q * T;
          v = chunk_sum(a.part + (size_t)(K + q) * a.ld + (size_t)jq * T + t) * a.s;
        }
        if (dia >= 0) v += lds[dia] + shift;
        lds[dst] = v;
      }
      }
    } else
    {
      const int NS = K + D1 - 1;                          // slots: K same-block offsets, then d = 1..tf+1
      const int LS = NS <= 8 ? 8 : 16;
      const int CPP = VT_THREADS / LS;
      const int slot = tid % LS, grp = tid / LS;
      const bool same_blk = slot < K;
      const int dslot = slot - K + 1;                      // prior coupling distance for slot >= K
      const int aa = same_blk ? slot : dslot * K;
      const bool slot_on = slot < NS;
      const int stepk = CPP % K, stept = CPP / K;
      {  // left view: column i (global g = i), entry (g+aa, g)
        int t = grp / K, k = grp - t * K;
        for (int i = grp; i < nL; i += CPP) {
          if (slot_on && i + aa < nL) {
            double v = 0.0;
            bool put = false;
            if (same_blk) {
              if (aa < K - k) {
                v = (a.weighted ? Ql + t * KK : Ql)[lidx(k + aa, k)];
                if (aa == 0) v += P[t * D1] + shift;
                put = true;
              }
            } else if (t + dslot < T) {
              v = P[t * D1 + dslot];
              put = true;
            }
            if (put) lds[W.L.band + i * R1 + aa] = v;
          }
          k += stepk; t += stept;
          if (k >= K) { k -= K; ++t; }
        }
      }
      {  // right view: mirrored column m (global gc = n-1-m), entry (gc, gc-aa)
        const int g0 = n - 1 - grp;
        int tc = g0 >= 0 ? g0 / K : 0, kc = g0 >= 0 ? g0 - tc * K : 0;
        for (int m = grp; m < nr; m += CPP) {
          const int gc = n - 1 - m;
          if (slot_on && gc - aa >= nl) {
            if (same_blk) {
              if (aa <= kc) {                              // row (tc, kc-aa): same depth block
                double v = (a.weighted ? Ql + tc * KK : Ql)[lidx(kc, kc - aa)];
                if (aa == 0) v += P[tc * D1] + shift;
                lds[W.R.band + m * R1 + aa] = v;
              }
            } else {
              lds[W.R.band + m * R1 + aa] = P[(tc - dslot) * D1 + dslot];   // row (tc-d, kc)
            }
          }
          kc -= stepk; tc -= stept;
          if (kc < 0) { kc += K; --tc; }
        }
      }
    }
    for (int idx = tid; idx < W.L.npad; idx += VT_THREADS) lds[W.L.rhs + idx] = idx < nL ? m0[idx] : 0.0;
    for (int idx = tid; idx < W.R.npad; idx += VT_THREADS) lds[W.R.rhs + idx] = idx < nr ? m0[n - 1 - idx] : 0.0;
    __syncthreads();
    stamp[2] = __builtin_amdgcn_s_memtime();
    // ---- the two chains, concurrently -------------------------------------------------------
    if (wave == 0) {
      bool good;
      if constexpr (ROW16) good = a.panel4 ? banded_ldl_forward_p4<NPL>(lds, W.L, nL, bw, nl) : banded_ldl_forward<NPL, ROW16>(lds, W.L, nL, bw, nl);
      else good = banded_ldl_forward<NPL, ROW16>(lds, W.L, nL, bw, nl);
      if (tid == 0) flag[0] = good ? 1.0 : 0.0;
    } else if (wave == 1) {
      bool good;
      if constexpr (ROW16) good = a.panel4 ? banded_ldl_forward_p4<NPL>(lds, W.R, nR, bw, nr) : banded_ldl_forward<NPL, ROW16>(lds, W.R, nR, bw, nr);
      else good = banded_ldl_forward<NPL, ROW16>(lds, W.R, nR, bw, nr);
      if (tid == 64) flag[1] = good ? 1.0 : 0.0;
    } else if (tried == 0) {
      // the normals of this column, indexed in elimination order (drawn once, whatever the retries)
      for (int idx = tid - 2 * WAVE; idx < n; idx += VT_THREADS - 2 * WAVE)
        zs[idx] = a.z ? a.z[(size_t)jg * n + idx] : philox_normal(a.seed, a.stream, (unsigned long long)jg * n + idx);
    }
    __syncthreads();
    ok = flag[0] != 0.0 && flag[1] != 0.0;
    if (ok) {
      // ---- separator system: S = S_L + mirror(S_R), reduced right-hand side likewise ----------
      for (int idx = tid; idx < W.S.npad * R1; idx += VT_THREADS) {
        const int jj = idx / R1, aa = idx - jj * R1;
        double v = 0.0;
        if (jj < ns && jj + aa < ns)
          v = lds[W.L.band + (nl + jj) * R1 + aa] + lds[W.R.band + (nr + ns - 1 - (jj + aa)) * R1 + aa];
        lds[W.S.band + idx] = v;
      }
      for (int idx = tid; idx < W.S.npad; idx += VT_THREADS)
        lds[W.S.rhs + idx] = idx < ns ? lds[W.L.rhs + nl + idx] + lds[W.R.rhs + nr + ns - 1 - idx] : 0.0;
      __syncthreads();
      // the halves' factors have no columns for the separator unknowns
      for (int idx = tid; idx < ns * R1; idx += VT_THREADS) {
        lds[W.L.band + nl * R1 + idx] = 0.0;
        lds[W.R.band + nr * R1 + idx] = 0.0;
      }
      if (wave == 0) {
        bool good;
        if constexpr (ROW16) good = a.panel4 ? banded_ldl_forward_p4<NPL>(lds, W.S, ns, bw, ns) : banded_ldl_forward<NPL, ROW16>(lds, W.S, ns, bw);
        else good = banded_ldl_forward<NPL, ROW16>(lds, W.S, ns, bw);
        if (tid == 0) flag[0] = good ? 1.0 : 0.0;
      } else {
        // meanwhile: w = D^-1 u + D^-1/2 z for the two interiors
        for (int idx = tid - WAVE; idx < nl + nr; idx += VT_THREADS - WAVE) {
          const bool left = idx < nl;
          const int i = left ? idx : idx - nl;
          const int ro = (left ? W.L.rhs : W.R.rhs) + i;
          const double iv = lds[(left ? W.L.invd : W.R.invd) + i];
          lds[ro] = fma(lds[ro], iv, zs[idx] * sqrt(iv));
        }
      }
      __syncthreads();
      ok = flag[0] != 0.0;
    }
    if (ok || tried >= a.attempts) break;
    shift += eps;   // fast_mvn.py:64-68: cumulative eps, eps *= 10
    eps *= 10.0;
    ++tried;
    __syncthreads();
  }
  if (tid == 0) a.tries[j] = tried;
  if (!ok) {
    if (tid == 0 && atomicCAS(&a.status[0], 0, 1) == 0) a.status[1] = jg;
    return;
  }
  stamp[3] = __builtin_amdgcn_s_memtime();
  // ---- separator: w, back-substitution ---------------------------------------------------------
  if (tid < ns) {
    const double iv = lds[W.S.invd + tid];
    lds[W.S.rhs + tid] = fma(lds[W.S.rhs + tid], iv, zs[nl + nr + tid] * sqrt(iv));
  }
  __syncthreads();
  if (wave == 0) banded_unit_backward_auto<ROW16>(lds, W.S, ns, bw);   // (separator: one block of the blocked routine)
  __syncthreads();
  // known trailing unknowns of both halves
  if (tid < ns) {
    const double xs = lds[W.S.rhs + tid];
    lds[W.L.rhs + nl + tid] = xs;
    lds[W.R.rhs + nr + ns - 1 - tid] = xs;
  }
  __syncthreads();
  stamp[4] = __builtin_amdgcn_s_memtime();
  if (ROW16 && BTF_TWIST_BACKPAR) {
    // (band stride 16: the 2 x 11 blocks' triangular solves in parallel, then one mat-vec per block and chain)
    const int nbL = (nL + 15) / 16, nbR = (nR + 15) / 16;
    backpar16_prepare(lds, W.L, nbL, W.R, nbR, tid, VT_THREADS);
    if (wave == 0) backpar16_chain_mfma(lds, W.L, nbL);
    else if (wave == 1) backpar16_chain_mfma(lds, W.R, nbR);
  } else {
    if (wave == 0) banded_unit_backward_auto<ROW16>(lds, W.L, nL, bw);
    else if (wave == 1) banded_unit_backward_auto<ROW16>(lds, W.R, nR, bw);
  }
  __syncthreads();
  stamp[5] = __builtin_amdgcn_s_memtime();
  // ---- write V[j] (depth-major), Gram share --------------------------------------------------
  double* xout = m0;                                     // m0 is dead: gather x in depth-major order
  for (int g = tid; g < n; g += VT_THREADS)
    xout[g] = g < nL ? lds[W.L.rhs + g] : lds[W.R.rhs + (n - 1 - g)];
  __syncthreads();
  for (int idx = tid; idx < n; idx += VT_THREADS) a.V[(size_t)jg * n + idx] = xout[idx];
  if (a.gout) {
    int ng = VT_THREADS / KK;
    if (ng > 16) ng = 16;
    if (ng < 1) ng = 1;
    const int g = tid / KK, q = tid - g * KK;
    int p = 0;
    while ((p + 1) * (p + 2) / 2 <= q) ++p;
    const int pq = q - p * (p + 1) / 2;
    double* scratch = lds + W.L.band;                    // dead by now; >= (bw+2)*(bw+1)+64 >= 16*KK doubles
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
  if (a.dbg && tid == 0)
    for (int i = 0; i < 6; ++i) a.dbg[(size_t)j * 6 + i] = stamp[i];
}

}  // namespace btf
