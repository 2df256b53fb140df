// Wave-specialised V half-sweep sampler (BTF_K_V_BANDED, preferred path).
//
// A lone wave issues about one instruction every 4-8 cycles, and a pivot of the banded
// LDL' costs ~70 instructions when one wave does everything (btf_banded_fast.h).  Here the
// four waves of the workgroup that owns a column take different jobs and run as a software
// pipeline through LDS:
//
//   wave 0  the dependent chain only: pivot -> 1/D -> next two pivot columns (registers,
//           DPP lane shifts + readlane).  It PUBLISHES (unscaled pivot column, 1/D) into a
//           4-slot ring and never touches the band except to load column n+2.
//   wave 1  half of the trailing pairs (a,b), b >= 3: A[n+a,n+b] -= v_a v_b / D on the LDS
//           band, and the row-major copy of L (Lrow[r][p mod (bw+1)] = L[r,p]) that makes
//           the back-substitution a lane-linear read.
//   wave 2  the other half of the pairs, the forward substitution of the right-hand side
//           and 1/D.
//   wave 3  the Philox normals of the column.
//
// Hand-off: two monotone counters per direction in LDS (pub: pivots published; done1/2:
// pivots whose updates a helper has finished).  Wave 0 may load column n+2 at pivot n only
// once both helpers have finished pivot n-1.  LDS executes each wave's operations in
// order, so "data, then counter" on the producer and "counter, then data" on the consumer
// is enough inside one workgroup.  EVERY spin is bounded; on timeout or a non-positive
// pivot an abort word releases everybody, so the grid always drains.
//
// Lrow rows reuse the band storage: band column c is dead after pivot c-2, and row r of
// Lrow is first written at pivot r-bw, so Lrow[r] lives in band column r-(bw+1) (a front
// pad of bw+1 columns takes the first rows).
#pragma once
#include "btf_banded_fast.h"

namespace btf {

constexpr int VP_THREADS = 256;
constexpr int VP_SPIN_MAX = 1 << 22;

struct VpLayout {
  int lrow;           // = band - Rw*R1 : Lrow[r] at lrow + r*R1
  int band;           // npad*R1 + 64
  int rhs;            // FP zeros in front, npad, +64
  int m0, zs, invd;   // npad each
  int P, Ql;
  int ring;           // 4 x 64 published (v; lane 0 carries 1/D)
  int sync;           // 8 ints: pub, done1, done2, abort, ok
  int dummy;          // 64*10 private words + a never-written zero word
  int total;
  int FP, npad, R1;
};
__host__ __device__ inline VpLayout vp_layout(int T, int K, int TF, int weighted) {
  VpLayout L;
  const int n = T * K, bw = (TF + 1) * K, D1 = TF + 2, KK = tri(K);
  L.R1 = bw + 1;
  L.npad = n + bw + 2;
  L.FP = bw + 4;
  int o = 0;
  L.lrow = o; o += L.R1 * L.R1;
  L.band = o; o += L.npad * L.R1 + 64;
  L.rhs = o + L.FP; o += L.FP + L.npad + 64;
  L.m0 = o; o += L.npad;
  L.zs = o; o += L.npad;
  L.invd = o; o += L.npad;
  L.P = o; o += T * D1;
  L.Ql = o; o += weighted ? T * KK : KK;
  L.ring = o; o += 4 * 64;
  L.sync = o; o += 4;
  L.dummy = o; o += 64 * 10 + 8;
  L.total = o;
  return L;
}
__host__ __device__ inline size_t vp_lds_bytes(int T, int K, int TF, int weighted) {
  return (size_t)vp_layout(T, K, TF, weighted).total * sizeof(double);
}

__device__ __forceinline__ int lds_flag_read(const double* lds, int word_off_doubles, int idx) {
  const volatile int* p = reinterpret_cast<const volatile int*>(lds + word_off_doubles);
  return p[idx];
}
__device__ __forceinline__ void lds_flag_write(double* lds, int word_off_doubles, int idx, int v) {
  volatile int* p = reinterpret_cast<volatile int*>(lds + word_off_doubles);
  p[idx] = v;
}
enum { VP_PUB = 0, VP_DONE1 = 1, VP_DONE2 = 2, VP_ABORT = 3, VP_OK = 4 };

// compiler-only ordering point: LDS executes a wave's operations in issue order and is one
// coherent memory for the workgroup, so "data then counter" / "counter then data" needs no
// s_waitcnt-draining fence, only that hipcc does not reorder the accesses.
#define VP_ORDER() asm volatile("" ::: "memory")

// ---- wave 0: the chain ------------------------------------------------------------------
// Column c of the band is fetched at pivot c-1, i.e. one pivot LATER than its first
// in-register correction is due: by then both helpers only need to have finished pivot c-3
// (two pivots of slack instead of one, which is what lets the waves overlap at all), and the
// missed b=2 correction of pivot c-2 is applied from saved registers together with the b=1
// correction of pivot c-1:   v_c = band_c - shl2(v_{c-2}) y2_{c-2} - shl1(v_{c-1}) y1_{c-1}.
template <bool ROW16>
__device__ inline bool vp_chain(double* lds, const VpLayout L, int n, int bw) {
  const int lane = threadIdx.x & 63;
  const int R1B = L.R1 * 8;
  const bool in_col = lane <= bw;
  const int zero = 8 * (L.dummy + 64 * 10);
  double v = in_col ? lds[L.band + lane] : 0.0;             // column 0
  int wo = in_col ? 8 * (L.band + L.R1 + lane) : zero;       // column nn+1
  const int woinc = in_col ? R1B : 0;
  const int ro = 8 * (L.ring + lane);
  double sh2p = 0.0, y2p = 0.0;                              // shl2(v), y_2 of the previous pivot
  int d1 = 0, d2 = 0;
  __builtin_amdgcn_s_waitcnt(0xc07f);
  for (int nn = 0; nn < n; ++nn) {
    const double p = bcast_first(v);
    if (!(p > 0.0)) return false;
    // column nn+1 must carry every helper update of pivots <= nn-2
    int dmin = d1 < d2 ? d1 : d2;
    for (int spin = 0; dmin < nn - 1; ++spin) {
      if (spin > VP_SPIN_MAX || lds_flag_read(lds, L.sync, VP_ABORT)) return false;
      const int e1 = lds_flag_read(lds, L.sync, VP_DONE1), e2 = lds_flag_read(lds, L.sync, VP_DONE2);
      dmin = e1 < e2 ? e1 : e2;
    }
    VP_ORDER();
    const double wn = ldsr(lds, wo);                         // consumed after the rcp chain
    wo += woinc;
    const double inv = rcp_nr(p);
    // publish (unscaled column, 1/D in lane 0), then the counter
    ldsw(lds, ro + 512 * (nn & 3), lane == 0 ? inv : v);
    VP_ORDER();
    lds_flag_write(lds, L.sync, VP_PUB, nn + 1);
    const double y = v * inv;
    const double y1 = bcast_lane(y, 1);
    const double late = fma(-sh2p, y2p, wn);                 // pivot nn-1, b = 2
    const double vnext = fma(-shift_down1<ROW16>(v), y1, late);
    sh2p = shift_down2<ROW16>(v);
    y2p = bcast_lane(y, 2);
    d1 = lds_flag_read(lds, L.sync, VP_DONE1);               // for the next pivot: issued now, used then
    d2 = lds_flag_read(lds, L.sync, VP_DONE2);
    v = in_col ? vnext : 0.0;
  }
  return true;
}

// ---- waves 1 and 2: trailing updates, Lrow / forward substitution ------------------------
template <int NPLH>
__device__ inline void vp_helper(double* lds, const VpLayout L, int n, int bw, int h /* 1 or 2 */) {
  const int lane = threadIdx.x & 63;
  const int R1 = L.R1, R1B = L.R1 * 8, Rw = bw + 1;
  // Trailing pairs (a,b), 3 <= b <= a <= bw.  A physical band entry keeps its diagonal offset
  // d = a-b for its whole life, so the split between the two helpers is by the parity of d:
  // the same helper then owns an entry at every pivot and its read-modify-writes stay in order.
  int to[NPLH], ao[NPLH], bo[NPLH], tinc[NPLH];
#pragma unroll
  for (int s = 0; s < NPLH; ++s) {
    const int want = lane + WAVE * s;             // my s-th pair is the want-th (d,b) of my parity
    int cnt = 0, fa = -1, fb = -1;
    for (int d = h - 1; d <= bw - 3 && fa < 0; d += 2) {
      const int nb = bw - d - 2;                   // b = 3 .. bw-d
      if (want < cnt + nb) { fb = 3 + (want - cnt); fa = fb + d; }
      cnt += nb;
    }
    if (fa >= 0) {
      to[s] = 8 * (L.band + fb * R1 + (fa - fb));
      ao[s] = 8 * fa;
      bo[s] = 8 * fb;
      tinc[s] = R1B;
    } else {                                       // idle slot: private dummy word, zero operands
      to[s] = 8 * (L.dummy + 64 * (1 + (h - 1) * 4 + s) + lane);
      ao[s] = bo[s] = 8 * 63;                      // ring entry 63 is always 0
      tinc[s] = 0;
    }
  }
  const bool in_sub = lane >= 1 && lane <= bw;
  const int dmy = 8 * (L.dummy + lane);
  // wave 1: Lrow[nn+lane][nn mod Rw] = L[nn+lane, nn];  wave 2: rhs[nn+lane] -= L[nn+lane,nn] u, invd[nn] = 1/D
  int xo = h == 1 ? (in_sub ? 8 * (L.lrow + lane * R1) : dmy)
                  : (in_sub ? 8 * (L.rhs + lane) : (lane == 0 ? 8 * L.invd : dmy));
  const int xinc = h == 1 ? (in_sub ? R1B : 0) : ((in_sub || lane == 0) ? 8 : 0);
  int slotc = 0;                                            // nn mod Rw (wave 1)
  const int ringb = 8 * L.ring;
  double t[NPLH];
#pragma unroll
  for (int s = 0; s < NPLH; ++s) t[s] = ldsr(lds, to[s]);
  double rt = (h == 2) ? ldsr(lds, xo) : 0.0;
  for (int nn = 0; nn < n; ++nn) {
    int pub = lds_flag_read(lds, L.sync, VP_PUB);
    for (int spin = 0; pub <= nn; ++spin) {
      if (spin > VP_SPIN_MAX || lds_flag_read(lds, L.sync, VP_ABORT)) return;
      pub = lds_flag_read(lds, L.sync, VP_PUB);
    }
    VP_ORDER();
    const int rb = ringb + 512 * (nn & 3);
    const double inv = ldsr(lds, rb);                       // lane 0 slot: 1/D
    const double myv = ldsr(lds, rb + 8 * lane);
    double xa[NPLH], xb[NPLH];
#pragma unroll
    for (int s = 0; s < NPLH; ++s) {
      xa[s] = ldsr(lds, rb + ao[s]);
      xb[s] = ldsr(lds, rb + bo[s]);
    }
    const double u = (h == 2) ? lds[L.rhs + nn] : 0.0;
#pragma unroll
    for (int s = 0; s < NPLH; ++s) {
      ldsw(lds, to[s], fma(-(xa[s] * xb[s]), inv, t[s]));
      to[s] += tinc[s];
    }
    const double y = myv * inv;                             // L[nn+lane, nn]
    if (h == 1) {
      ldsw(lds, xo + 8 * slotc, y);
      slotc = slotc + 1 == Rw ? 0 : slotc + 1;
    } else {
      ldsw(lds, xo, lane == 0 ? inv : fma(-y, u, rt));
    }
    xo += xinc;
    VP_ORDER();
    lds_flag_write(lds, L.sync, h == 1 ? VP_DONE1 : VP_DONE2, nn + 1);
    // next step's targets (own writes above are ordered before these reads)
#pragma unroll
    for (int s = 0; s < NPLH; ++s) t[s] = ldsr(lds, to[s]);
    if (h == 2) rt = ldsr(lds, xo);
  }
}

// ---- wave 0 again: back-substitution on the row-major factor ------------------------------
__device__ inline void vp_backward(double* lds, const VpLayout L, int n, int bw) {
  const int lane = threadIdx.x & 63;
  const int R1B = L.R1 * 8, Rw = bw + 1;
  const bool act = lane < Rw;
  const int dmy = 8 * (L.dummy + lane);
  const int p0 = (n - 1) - ((n - 1 - lane) % Rw + Rw) % Rw;
  double wv = (act && p0 >= 0) ? lds[L.rhs + p0] : 0.0;
  int own = __builtin_amdgcn_readfirstlane((n - 1) % Rw);
  int la = act ? 8 * (L.lrow + (n - 1) * L.R1 + lane) : dmy;   // Lrow[r][lane], r walks down
  const int lstep = act ? R1B : 0;
  int wa = 8 * (L.rhs + n - 1 - Rw);
  auto fetch = [&](double& Lc, double& Wc) {
    Lc = ldsr(lds, la);
    Wc = ldsr(lds, wa);
    la -= lstep;
    wa -= 8;
  };
  auto step = [&](int r, double& Lc, double& Wc) {
    const double xr = bcast_lane(wv, own);
    const bool retire = lane == own;
    wv = retire ? Wc : fma(-Lc, xr, wv);
    ldsw(lds, retire ? 8 * (L.rhs + r) : dmy, xr);
    own = own == 0 ? Rw - 1 : own - 1;
    fetch(Lc, Wc);
  };
  double L0, L1, L2, W0, W1, W2;
  fetch(L0, W0);
  fetch(L1, W1);
  fetch(L2, W2);
  int r = n - 1;
  for (; r >= 2; r -= 3) {
    step(r, L0, W0);
    step(r - 1, L1, W1);
    step(r - 2, L2, W2);
  }
  if (r >= 0) step(r, L0, W0);
  if (r >= 1) step(r - 1, L1, W1);
}

template <int NPLH, bool ROW16>
__global__ __launch_bounds__(VP_THREADS) void v_banded_pipe_kernel(VBandArgs a, int K) {
  vband_load_hyp(a);
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  if (!lds_base_is_zero(lds)) {      // ldsr / ldsw address LDS absolutely (see btf_banded_fast.h); never taken
    if (tid == 0) { a.status[0] = 1; a.status[1] = -7; }
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = blockIdx.x, jg = a.col0 + j;
  const int KK = tri(K), T = a.T, n = T * K, D1 = a.TF + 2, bw = (a.TF + 1) * K, R1 = bw + 1;
  const int NV = a.weighted ? K + KK : K;
  const VpLayout L = vp_layout(T, K, a.TF, a.weighted);
  double* Bc = lds + L.band;
  double* rhs = lds + L.rhs;
  double* m0 = lds + L.m0;
  double* zs = lds + L.zs;
  double* invd = lds + L.invd;
  double* P = lds + L.P;
  double* Ql = lds + L.Ql;
  long long stamp[6];
  stamp[0] = __builtin_amdgcn_s_memtime();

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
  for (int idx = tid; idx < n; idx += VP_THREADS) {
    const int t = idx / K, k = idx - t * K;
    m0[idx] = chunk_sum(a.part + (size_t)k * a.ld + (size_t)j * T + t) * a.s;
  }
  if (a.weighted) {
    for (int idx = tid; idx < T * KK; idx += VP_THREADS) {
      const int t = idx / KK, q = idx - t * KK;
      Ql[idx] = chunk_sum(a.part + (size_t)(K + q) * a.ld + (size_t)j * T + t) * a.s;
    }
  } else {
    reduce_gram(a.gpart, a.ngp, KK, a.sR, Bc, Ql);
  }
  for (int idx = tid; idx < T * D1; idx += VP_THREADS) P[idx] = a.pband[(size_t)j * T * D1 + idx];
  const int npad = L.npad;
  for (int idx = n + tid; idx < npad; idx += VP_THREADS) m0[idx] = 0.0;
  for (int idx = tid; idx < 64; idx += VP_THREADS) {
    Bc[npad * R1 + idx] = 0.0;
    rhs[npad + idx] = 0.0;
  }
  for (int idx = tid; idx < L.FP; idx += VP_THREADS) rhs[idx - L.FP] = 0.0;
  for (int idx = tid; idx < 64 * 10 + 8; idx += VP_THREADS) lds[L.dummy + idx] = 0.0;
  for (int idx = tid; idx < 4 * 64; idx += VP_THREADS) lds[L.ring + idx] = 0.0;
  __syncthreads();
  stamp[1] = __builtin_amdgcn_s_memtime();

  double shift = 0.0, eps = a.eps0;
  int tried = 0;
  bool ok = false;
  while (true) {
    for (int nn = tid; nn < npad; nn += VP_THREADS) {
      double* colw = Bc + (size_t)nn * R1;
      if (nn >= n) {
        for (int aa = 0; aa < R1; ++aa) colw[aa] = 0.0;
        continue;
      }
      const int t = nn / K, k = nn - t * K;
      const double* q = a.weighted ? Ql + t * KK : Ql;
      int dd = 0, rem = 0;
      for (int aa = 0; aa < R1; ++aa) {
        double v = 0.0;
        if (aa < K - k) {
          v = q[lidx(k + aa, k)];
          if (aa == 0) v += P[t * D1] + shift;
        } else if (rem == 0 && dd < D1 && t + dd < T) {
          v = P[t * D1 + dd];
        }
        colw[aa] = v;
        if (++rem == K) { rem = 0; ++dd; }
      }
    }
    for (int idx = tid; idx < R1 * R1; idx += VP_THREADS) lds[L.lrow + idx] = 0.0;   // first rows of Lrow
    for (int idx = tid; idx < npad; idx += VP_THREADS) rhs[idx] = m0[idx];
    if (tid < 8) lds_flag_write(lds, L.sync, tid, 0);
    __syncthreads();
    stamp[2] = __builtin_amdgcn_s_memtime();
    if (wave == 0) {
      const bool good = vp_chain<ROW16>(lds, L, n, bw);
      if (!good) lds_flag_write(lds, L.sync, VP_ABORT, 1);
      if (tid == 0) lds_flag_write(lds, L.sync, VP_OK, good ? 1 : 0);
    } else if (wave == 1) {
      vp_helper<NPLH>(lds, L, n, bw, 1);
    } else if (wave == 2) {
      vp_helper<NPLH>(lds, L, n, bw, 2);
    } else if (tried == 0) {
      for (int idx = tid - 3 * WAVE; idx < n; idx += WAVE)
        zs[idx] = a.z ? a.z[(size_t)jg * n + idx] : philox_normal(a.seed, a.stream, (unsigned long long)jg * n + idx);
    }
    __syncthreads();
    ok = lds_flag_read(lds, L.sync, VP_OK) != 0;
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
  for (int idx = tid; idx < n; idx += VP_THREADS) {
    const double iv = invd[idx];
    rhs[idx] = fma(rhs[idx], iv, zs[idx] * sqrt(iv));
  }
  __syncthreads();
  stamp[4] = __builtin_amdgcn_s_memtime();
  if (wave == 0) vp_backward(lds, L, n, bw);
  __syncthreads();
  stamp[5] = __builtin_amdgcn_s_memtime();
  for (int idx = tid; idx < n; idx += VP_THREADS) a.V[(size_t)jg * n + idx] = rhs[idx];
  if (a.gout) {   // this column's share of V'V: two fixed-order levels; scratch = the dead factor storage
    __syncthreads();
    int ng = VP_THREADS / KK;
    if (ng > 16) ng = 16;
    if (ng * KK > 120) ng = 120 / KK;              // the smallest supported band region holds 124 doubles
    if (ng < 1) ng = 1;
    const int g = tid / KK, q = tid - g * KK;
    int p = 0;
    while ((p + 1) * (p + 2) / 2 <= q) ++p;
    const int pq = q - p * (p + 1) / 2;
    if (g < ng) {
      double s = 0.0;
      for (int t = g; t < T; t += ng) s = fma(rhs[t * K + p], rhs[t * K + pq], s);
      lds[g * KK + q] = s;
    }
    __syncthreads();
    if (tid < KK) {
      double s = 0.0;
      for (int b = 0; b < ng; ++b) s += lds[b * KK + tid];
      a.gout[(size_t)j * KK + tid] = s;
    }
  }
  if (a.dbg && tid == 0)
    for (int i = 0; i < 6; ++i) a.dbg[(size_t)j * 6 + i] = stamp[i];
}

}  // namespace btf
