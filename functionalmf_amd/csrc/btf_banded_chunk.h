// Chunked single-chain V half-sweep sampler: the block-banded LDL' of btf_banded_fast.h for columns whose band does
// not fit the 160 KB of LDS (weighted data at nembeds >= 8, long depth axes).
//
// The factorisation only ever touches bw + 1 columns at a time, so the band is never needed whole: the workgroup
// assembles CH + bw columns of it in LDS, eliminates the first CH (banded_ldl_forward with n_elim = CH leaves the Schur
// complement of the rest in the last bw columns, and the forward-substituted right-hand side beside it), parks the
// CH finished factor columns with their 1/D and rhs entries in an HBM scratch record, moves the bw Schur columns to
// the front, assembles the next CH columns behind them, and so on.  The back-substitution walks the chunks the other
// way: the last chunk is still in LDS; each earlier one is read back (one coalesced copy), with the bw unknowns that
// follow it - already solved - appended as known trailing unknowns behind zero factor columns, exactly as the twisted
// kernel appends its separator.  Same arithmetic, same order as the one-piece chain: the draw is the depth-major
// x = Q^-1 mu + L^-T D^-1/2 z of BTF_SAMPLER_CHAIN.  HBM traffic: the factor once out and once in, n (bw + 3) doubles
// per column; everything else (assembly from the accumulation partials, the chain, the solves) stays on chip.
//
// Before this kernel such columns went to the any-size kernel (one wave per column, band in HBM, ~2.7 us per pivot:
// 1.75 ms at (512,256,64) nembeds = 10 with missing data); round 3.
#pragma once
#include "btf_banded_fast.h"

namespace btf {

struct VcLayout {
  VbLayout V;                  // the view the chain routines work on (band, rhs, invd, vsc, dummy)
  int m0, zs, P, Ql, flag, xk; // whole-column vectors, prior band, likelihood blocks of the chunk's depths, flags, known tail
  int CH, VC, QT;              // columns eliminated per chunk, columns in the view, depths the Ql buffer holds
  int total;
};
__host__ __device__ inline VcLayout vc_layout(int T, int K, int TF, int weighted, int CH) {
  VcLayout C;
  const int n = T * K, bw = (TF + 1) * K, D1 = TF + 2, KK = tri(K);
  VbLayout& L = C.V;
  C.CH = CH; C.VC = CH + bw; C.QT = (C.VC + K - 1) / K + 2;
  L.R1 = bw + 1;
  L.npad = C.VC + bw + 2;
  L.FP = bw + 4;
  int o = 0;
  L.band = o; o += L.npad * L.R1 + 64;
  L.rhs = o + L.FP; o += L.FP + L.npad + 64;
  L.invd = o; o += L.npad;
  L.vsc = o; o += 64;
  L.vs4 = o;                   // (panelised factorisation only: unused here)
  L.dummy = o; o += 64 * 9 + 8;
  L.m0 = L.zs = L.P = L.Ql = L.flag = 0; L.total = 0;
  C.m0 = o; o += n;
  C.zs = o; o += n;
  C.P = o; o += T * D1;
  C.Ql = o; o += weighted ? C.QT * KK : KK;
  C.flag = o; o += 8;
  C.xk = o; o += 64;
  C.total = o;
  return C;
}
__host__ __device__ inline size_t vc_lds_bytes(int T, int K, int TF, int weighted, int CH) {
  return (size_t)vc_layout(T, K, TF, weighted, CH).total * sizeof(double);
}
// columns per chunk for an LDS budget (0: does not fit - the fixed vectors alone are too long, or no room for a chunk
// worth the copy: at least 2 bw + 2 columns)
__host__ inline int vc_pick_chunk(int T, int K, int TF, int weighted, size_t budget) {
  const int bw = (TF + 1) * K, n = T * K;
  const size_t base = vc_lds_bytes(T, K, TF, weighted, 0);
  if (base >= budget) return 0;
  // per extra column of the view: a band column, rhs, 1/D and (weighted) its share of the per-depth blocks
  const size_t per = (size_t)(bw + 1 + 2) * sizeof(double) + (weighted ? (size_t)(tri(K) + K - 1) / K * sizeof(double) : 0);
  int ch = (int)((budget - base) / per);
  while (ch > 0 && vc_lds_bytes(T, K, TF, weighted, ch) > budget) --ch;
  if (ch > n) ch = n;
  ch &= ~1;
  return ch >= 2 * bw + 2 ? ch : 0;
}
__host__ __device__ inline size_t vc_scratch_stride(int T, int K, int TF) { return (size_t)T * K * ((TF + 1) * K + 1 + 2); }

template <int NPL>
__global__ __launch_bounds__(VB_THREADS) void v_banded_chunk_kernel(VBandArgs a, int K, int CH) {
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
  const int KK = tri(K), T = a.T, n = T * K, D1 = a.TF + 2, bw = (a.TF + 1) * K, R1 = bw + 1;
  const int NV = a.weighted ? K + KK : K;
  const VcLayout C = vc_layout(T, K, a.TF, a.weighted, CH);
  const VbLayout L = C.V;
  const int VC = C.VC, npad = L.npad;
  double* Bc = lds + L.band;
  double* rhs = lds + L.rhs;
  double* invd = lds + L.invd;
  double* m0 = lds + C.m0;
  double* zs = lds + C.zs;
  double* P = lds + C.P;
  double* Ql = lds + C.Ql;
  double* flag = lds + C.flag;
  double* xk = lds + C.xk;
  // this column's factor record in HBM: [n][R1] unit-lower factor columns, [n] 1/D, [n] forward-substituted rhs
  double* gL = a.gband + (size_t)j * a.gband_stride;
  double* gI = gL + (size_t)n * R1;
  double* gU = gI + n;

  auto chunk_sum = [&](const double* p) -> double {      // fixed order; loads issued 4 at a time
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
  // ---- whole-column vectors: likelihood mean part (element e = k T + t: coalesced along t), prior band; zero pads ----
  for (int e = tid; e < n; e += VB_THREADS) {
    const int k = e / T, t = e - k * T;
    m0[t * K + k] = chunk_sum(a.part + (size_t)k * a.ld + (size_t)j * T + t) * a.s;
  }
  if (!a.weighted) reduce_gram(a.gpart, a.ngp, KK, a.sR, Bc, Ql);   // Bc is free scratch until the assembly (ends with a barrier)
  for (int idx = tid; idx < T * D1; idx += VB_THREADS) P[idx] = a.pband[(size_t)j * T * D1 + idx];
  for (int idx = tid; idx < 64; idx += VB_THREADS) {
    Bc[npad * R1 + idx] = 0.0;
    rhs[npad + idx] = 0.0;
    lds[L.vsc + idx] = 0.0;
  }
  for (int idx = tid; idx < L.FP; idx += VB_THREADS) rhs[idx - L.FP] = 0.0;
  for (int idx = tid; idx < 64 * 9 + 8; idx += VB_THREADS) lds[L.dummy + idx] = 0.0;
  __syncthreads();

  double shift = 0.0, eps = a.eps0;
  int tried = 0;
  bool ok = false;
  int c0 = 0, ne = 0;                                      // the chunk in LDS: columns c0 .. c0 + ne - 1 eliminated
  while (true) {
    // ================= forward: chunk after chunk =================
    c0 = 0;
    ok = true;
    bool first = true;
    while (true) {
      const int rem = n - c0;
      ne = rem <= VC ? rem : CH;                           // the last chunk takes everything that is left
      const int nview = rem < VC ? rem : VC;               // matrix columns in the view
      const int f0 = first ? 0 : bw;                       // fresh columns: view index f0 .. nview - 1 (the first bw are the carried Schur columns)
      // per-depth likelihood blocks of the fresh columns' depths (element e = q * cnt + tt: coalesced along the depth)
      int t0 = 0;
      if (a.weighted) {
        t0 = (c0 + f0) / K;
        const int t1 = (c0 + nview - 1) / K, cnt = t1 - t0 + 1;
        for (int e = tid; e < cnt * KK; e += VB_THREADS) {
          const int q = e / cnt, tt = e - q * cnt;
          Ql[tt * KK + q] = chunk_sum(a.part + (size_t)(K + q) * a.ld + (size_t)jq * T + t0 + tt) * a.s;
        }
        __syncthreads();
      }
      for (int i = f0 + tid; i < npad; i += VB_THREADS) {  // one band column per thread, no divisions inside the entry loop
        double* colw = Bc + (size_t)i * R1;
        if (i >= nview) {
          for (int aa = 0; aa < R1; ++aa) colw[aa] = 0.0;
          rhs[i] = 0.0;
          continue;
        }
        const int g = c0 + i, t = g / K, k = g - t * K;
        const double* q = a.weighted ? Ql + (t - t0) * KK : Ql;
        int dd = 0, rm = 0;                                // aa = dd*K + rm
        for (int aa = 0; aa < R1; ++aa) {
          double v = 0.0;
          if (aa < K - k) {
            v = q[lidx(k + aa, k)];
            if (aa == 0) v += P[t * D1] + shift;
          } else if (rm == 0 && dd < D1 && t + dd < T) {
            v = P[t * D1 + dd];
          }
          colw[aa] = v;
          if (++rm == K) { rm = 0; ++dd; }
        }
        rhs[i] = m0[g];
      }
      __syncthreads();
      if (wave == 0) {
        const bool good = banded_ldl_forward<NPL, false>(lds, L, nview, bw, ne);
        if (tid == 0) flag[0] = good ? 1.0 : 0.0;
      } else if (tried == 0 && first) {
        // the normals of this column, depth-major index (drawn once, whatever the retries)
        for (int idx = tid - WAVE; idx < n; idx += VB_THREADS - WAVE)
          zs[idx] = a.z ? a.z[(size_t)jg * n + idx] : philox_normal(a.seed, a.stream, (unsigned long long)jg * n + idx);
      }
      __syncthreads();
      if (flag[0] == 0.0) { ok = false; break; }
      if (ne == rem) break;                                // the last chunk stays in LDS for the back-substitution
      // park the finished columns, then move the Schur columns (and their rhs entries) to the front
      for (int idx = tid; idx < ne * R1; idx += VB_THREADS) gL[(size_t)c0 * R1 + idx] = Bc[idx];
      for (int idx = tid; idx < ne; idx += VB_THREADS) { gI[c0 + idx] = invd[idx]; gU[c0 + idx] = rhs[idx]; }
      __syncthreads();                                     // (ne >= bw: source and destination do not overlap)
      for (int idx = tid; idx < bw * R1; idx += VB_THREADS) Bc[idx] = Bc[(size_t)ne * R1 + idx];
      for (int idx = tid; idx < bw; idx += VB_THREADS) rhs[idx] = rhs[ne + idx];
      __syncthreads();
      c0 += ne;
      first = false;
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
  // ================= backward: the chunk in LDS first, then the parked ones, last to first =================
  double* xout = m0;                                       // m0 is dead: x in depth-major order
  bool have_tail = false;
  while (true) {
    // w = D^-1 u + D^-1/2 z
    for (int idx = tid; idx < ne; idx += VB_THREADS) {
      const double iv = invd[idx];
      rhs[idx] = fma(rhs[idx], iv, zs[c0 + idx] * sqrt(iv));
    }
    if (have_tail) {                                       // the bw unknowns behind this chunk are known: zero factor columns, x as rhs
      // (everything behind the chunk: the blocked routine's top block reaches up to bw columns past the last unknown,
      //  and the later chunk's columns are still lying there)
      for (int idx = ne * R1 + tid; idx < npad * R1; idx += VB_THREADS) Bc[idx] = 0.0;
      for (int idx = tid; idx < bw; idx += VB_THREADS) rhs[ne + idx] = xk[idx];
      for (int idx = ne + bw + tid; idx < npad; idx += VB_THREADS) rhs[idx] = 0.0;
    }
    __syncthreads();
    if (wave == 0) banded_unit_backward_auto<false>(lds, L, have_tail ? ne + bw : ne, bw);
    __syncthreads();
    for (int idx = tid; idx < ne; idx += VB_THREADS) xout[c0 + idx] = rhs[idx];
    for (int idx = tid; idx < bw; idx += VB_THREADS) xk[idx] = idx < ne ? rhs[idx] : 0.0;
    if (c0 == 0) break;
    __syncthreads();
    c0 -= CH; ne = CH; have_tail = true;
    for (int idx = tid; idx < ne * R1; idx += VB_THREADS) Bc[idx] = gL[(size_t)c0 * R1 + idx];
    for (int idx = tid; idx < ne; idx += VB_THREADS) { invd[idx] = gI[c0 + idx]; rhs[idx] = gU[c0 + idx]; }
    __syncthreads();
  }
  __syncthreads();
  for (int idx = tid; idx < n; idx += VB_THREADS) a.V[(size_t)jg * n + idx] = xout[idx];
  if (a.gout) {   // this column's share of V'V (rows (j,t), t = 0..T-1): two fixed-order levels
    const int ng = VB_THREADS / KK, g = tid / KK, q = tid - g * KK;
    int p = 0;
    while ((p + 1) * (p + 2) / 2 <= q) ++p;
    const int pq = q - p * (p + 1) / 2;
    if (g < ng) {
      double s = 0.0;
      for (int t = g; t < T; t += ng) s = fma(xout[t * K + p], xout[t * K + pq], s);
      Bc[g * KK + q] = s;                  // the band is dead by now: scratch
    }
    __syncthreads();
    if (tid < KK) {
      double s = 0.0;
      for (int b = 0; b < ng; ++b) s += Bc[b * KK + tid];
      a.gout[(size_t)j * KK + tid] = s;
    }
  }
}

}  // namespace btf
