// Fast path of the V half-sweep sampler (BTF_K_V_BANDED): band resident in LDS,
// one workgroup of 4 waves per column.
//
//   * LDL' instead of LL': the sequential pivot chain carries one v_rcp_f64 (+2 Newton
//     steps) instead of a square root and a division; the square roots are taken
//     afterwards, in parallel, where the noise term needs them
//     (x = L^-T (D^-1 L^-1 mu + D^-1/2 z): the same vector as Q^-1 mu + L~^-T z with
//     the Cholesky factor L~ = L D^1/2 of fast_mvn.py:38-47).
//   * the NEXT pivot column never leaves registers: lane l of wave 0 keeps A[n+1+l, n+1];
//     its rank-1 correction needs only a one-lane shift and one readlane, so the
//     dependent chain per pivot is readlane -> rcp -> mul -> fma.  All other trailing
//     updates are read-modify-writes on the LDS band, off the chain.
//   * the band and rhs are zero-padded by bw+1 columns: no tail special-casing.
//   * waves 1..3 assemble, then draw the Philox normals while wave 0 factors.
//   * back-substitution keeps the bw+1 live unknowns in registers (lane = column mod
//     (bw+1)); factor entries and fresh right-hand sides are prefetched two steps ahead.
#pragma once
#include "btf_kernels.h"

namespace btf {

constexpr int VB_THREADS = 256;

__device__ __forceinline__ double rcp_nr(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(e, r, r);
  e = fma(-d, r, 1.0);
  return fma(e, r, r);
}

// lane l <- lane l+1 (lanes past the end read 0)
template <bool ROW16>
__device__ __forceinline__ double shift_down1(double v) {
  if constexpr (ROW16) {  // DPP row_shl:1 inside a 16-lane row, out-of-row reads give 0
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x101, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x101, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  } else {
    const int lane = threadIdx.x & 63;
    double r = __shfl_down(v, 1, WAVE);
    return lane == 63 ? 0.0 : r;
  }
}

// Everything below addresses LDS as `lds[int offset]` (offsets in doubles from the
// dynamic-LDS base): pointer selects make hipcc fall back to flat_load/flat_store,
// and exec-masked regions make it drain lgkmcnt(0); so the two sequential loops are
// written branch-free - idle lanes are steered to private dummy words instead of being
// masked off - and every load of a step is issued before its first store.
struct VbLayout {     // offsets (doubles) inside the dynamic LDS block
  int band;           // npad*R1 + 64   band, [c][a] = A[c+a,c]; zero padded
  int rhs;            // FP + npad + 64 right-hand side, FP zeros in front
  int m0, zs, invd;   // npad each
  int P, Ql, flag, vsc, dummy;
  int vs4;            // 64     unscaled pivot columns of the current panel (panelised factorisation)
  int total;
  int FP, npad, R1;
};
__host__ __device__ inline VbLayout vb_layout(int T, int K, int TF, int weighted) {
  VbLayout L;
  const int n = T * K, bw = (TF + 1) * K, D1 = TF + 2, KK = tri(K);
  L.R1 = bw + 1;
  L.npad = n + bw + 2;
  L.FP = bw + 4;
  int o = 0;
  L.band = o; o += L.npad * L.R1 + 64;
  L.rhs = o + L.FP; o += L.FP + L.npad + 64;
  L.m0 = o; o += L.npad;
  L.zs = o; o += L.npad;
  L.invd = o; o += L.npad;
  L.P = o; o += T * D1;
  L.Ql = o; o += weighted ? T * KK : KK;
  L.flag = o; o += 8;
  L.vsc = o; o += 64;
  L.vs4 = o; o += 64;
  L.dummy = o; o += 64 * 9 + 8;
  L.total = o;
  return L;
}
__host__ __device__ inline size_t vb_fast_lds_bytes(int T, int K, int TF, int weighted) {
  return (size_t)vb_layout(T, K, TF, weighted).total * sizeof(double);
}

// lane l <- lane l+2 (lanes past the end read 0)
template <bool ROW16>
__device__ __forceinline__ double shift_down2(double v) {
  if constexpr (ROW16) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x102, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x102, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  } else {
    const int lane = threadIdx.x & 63;
    double r = __shfl_down(v, 2, WAVE);
    return lane >= 62 ? 0.0 : r;
  }
}

// LDS access by BYTE offset from the dynamic-LDS base (one ds_read/ds_write, no shift, no add).
// The kernels that use these declare no static __shared__ data, so the dynamic block starts at LDS
// address 0 and the offset IS the address: going through `(char*)lds + boff` cost one
// `v_add_u32 x, 0, off` per access in the ISA.  `lds_base_is_zero` is checked once per kernel.
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ bool lds_base_is_zero(const double* lds) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) double*)lds == 0u;
}
__device__ __forceinline__ double ldsr(const double* lds, int boff) {
  (void)lds;
  return *(const lds_f64*)(size_t)(unsigned)boff;
}
__device__ __forceinline__ void ldsw(double* lds, int boff, double x) {
  (void)lds;
  *(lds_f64*)(size_t)(unsigned)boff = x;
}

// LDL' of the band with the forward substitution of rhs folded in.  Wave 0 only.
// On return: band = unit-lower factor columns, invd[c] = 1/D_c, rhs = L^-1 rhs.
//
// The next TWO pivot columns live in registers (v: column nn, w: column nn+1, lane l =
// row offset l); their rank-1 corrections need a lane shift and a readlane only, so the
// dependent chain per pivot is readfirstlane -> rcp -> mul -> readlane -> fma.  The other
// trailing pairs (a,b), 3 <= b <= a <= bw, are read-modify-writes on the LDS band whose
// loads are issued at the END of the previous step (software pipelining): their operands
// are the UNSCALED pivot column (A[n+a,n] A[n+b,n] / D_n), parked in a scratch row as
// soon as it is known, so no LDS load is consumed in the step that issues it.
// A lone wave issues roughly one instruction per 4-8 cycles, so the loop is written for
// instruction count: byte offsets, and lanes outside the band are steered to private
// dummy / zero words through their (precomputed) addresses and strides instead of
// being masked with selects.
// n_elim < n: eliminate only the first n_elim unknowns; the band columns >= n_elim then hold
// the Schur complement of the rest (the two register-resident columns are flushed) and
// rhs[>= n_elim] the correspondingly reduced right-hand side (twisted factorisation).
template <int NPL, bool ROW16>
__device__ inline bool banded_ldl_forward(double* lds, const VbLayout L, int n, int bw, int n_elim = -1) {
  const int lane = threadIdx.x & 63;
  const int R1 = L.R1, R1B = L.R1 * 8;
  if (n_elim < 0) n_elim = n;
  const int npairs = (bw - 1) * (bw - 2) / 2;       // pairs with b >= 3
  int to[NPL], ao[NPL], bo[NPL], tinc[NPL];          // byte offsets
#pragma unroll
  for (int s = 0; s < NPL; ++s) {
    const int q = lane + WAVE * s;
    if (q < npairs) {   // q = (a-3)(a-2)/2 + (b-3)   (row-major; a column-major order measured 8 % slower)
      int a = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5) + 3;
      while ((a - 3) * (a - 2) / 2 > q) --a;
      while ((a - 2) * (a - 1) / 2 <= q) ++a;
      const int b = q - (a - 3) * (a - 2) / 2 + 3;
      to[s] = 8 * (L.band + b * R1 + (a - b));
      ao[s] = 8 * (L.vsc + a);
      bo[s] = 8 * (L.vsc + b);
      tinc[s] = R1B;
    } else {                       // idle slot: read-modify-write of a private dummy word
      to[s] = 8 * (L.dummy + 64 * (1 + s) + lane);
      ao[s] = bo[s] = 8 * (L.vsc + 63);  // entry 63 of the scratch row is always 0 (bw <= 62)
      tinc[s] = 0;
    }
  }
  const bool in_col = lane <= bw;
  const bool in_sub = in_col && lane >= 1;
  const int dmy = 8 * (L.dummy + lane);
  const int zero = 8 * (L.dummy + 64 * 9);                   // a word that is never written
  double v = in_col ? lds[L.band + lane] : 0.0;             // column 0
  double w = in_col ? lds[L.band + R1 + lane] : 0.0;        // column 1 (uncorrected)
  // per-lane addresses with per-lane strides: lanes outside the band stay on dummy / zero words
  int co = in_col ? 8 * (L.band + lane) : dmy;               // store of L[nn+lane, nn]
  const int coinc = in_col ? R1B : 0;
  int wo = in_col ? 8 * (L.band + 2 * R1 + lane) : zero;     // load of column nn+2
  int ro = in_sub ? 8 * (L.rhs + lane) : dmy;                // rhs[nn+lane] read-modify-write
  const int roinc = in_sub ? 8 : 0;
  int io = lane == 0 ? 8 * L.invd : dmy;                     // invd[nn] (lane 0)
  const int ioinc = lane == 0 ? 8 : 0;
  const int vo = 8 * (L.vsc + lane);
  // prologue of the software pipeline: operands / targets of step 0
  ldsw(lds, vo, v);
  double t[NPL], xa[NPL], xb[NPL];
#pragma unroll
  for (int s = 0; s < NPL; ++s) {
    t[s] = ldsr(lds, to[s]);
    xa[s] = ldsr(lds, ao[s]);
    xb[s] = ldsr(lds, bo[s]);
  }
  double rt = ldsr(lds, ro);
  double wn = ldsr(lds, wo);
  double u = lds[L.rhs];                                     // rhs[0]
  __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): enter the loop with nothing pending
  // a non-positive (or NaN) pivot is only RECORDED: the loop has no data-dependent addresses, so it
  // runs to the end on garbage and the caller retries with jitter (rare) - no compare-and-branch per pivot
  bool bad = false;
  for (int nn = 0; nn < n_elim; ++nn) {
    const double p = bcast_first(v);
    bad |= !(p > 0.0);
    // ---- dependent chain
    const double inv = rcp_nr(p);
    const double y = v * inv;                                // L[nn+lane, nn]  (lane 0: 1)
    const double y1 = bcast_lane(y, 1);
    const double y2 = bcast_lane(y, 2);
    const double vnext = fma(-shift_down1<ROW16>(v), y1, w);   // column nn+1, final
    const double wnext = fma(-shift_down2<ROW16>(v), y2, wn);  // column nn+2 after this step
    const double rnew = fma(-y, u, rt);                      // rhs[nn+lane] - L[nn+lane,nn] u
    // ---- trailing updates  A[nn+a, nn+b] -= A[nn+a,nn] A[nn+b,nn] / D   (b >= 3), kept in registers for now
    // (scaled by inv FIRST: consuming the LDS-loaded operands only after the rcp chain keeps their
    //  s_waitcnt off the top of the loop)
    double tn[NPL];
#pragma unroll
    for (int s = 0; s < NPL; ++s) tn[s] = fma(-(xa[s] * inv), xb[s], t[s]);
    // ---- the next pivot column is known: park it and fetch the operands of step nn+1 now
    ldsw(lds, vo, vnext);
#pragma unroll
    for (int s = 0; s < NPL; ++s) {
      xa[s] = ldsr(lds, ao[s]);
      xb[s] = ldsr(lds, bo[s]);
    }
    // ---- stores of this step, then the loads that must see them (in-order LDS)
#pragma unroll
    for (int s = 0; s < NPL; ++s) {
      ldsw(lds, to[s], tn[s]);
      to[s] += tinc[s];
    }
    ldsw(lds, co, y);
    ldsw(lds, ro, rnew);
    ldsw(lds, io, inv);
    co += coinc;
    ro += roinc;
    io += ioinc;
    wo += coinc;
    v = vnext;                                               // lanes outside the band: 0 by construction
    w = wnext;
    u = bcast_lane(rnew, 1);                                 // rhs[nn+1], final after this step
#pragma unroll
    for (int s = 0; s < NPL; ++s) t[s] = ldsr(lds, to[s]);
    rt = ldsr(lds, ro);
    wn = ldsr(lds, wo);
  }
  if (n_elim < n && in_col) {          // flush the two register-resident columns of the Schur complement
    lds[L.band + n_elim * R1 + lane] = v;
    lds[L.band + (n_elim + 1) * R1 + lane] = w;
  }
  return !bad;
}

// Panelised variant of banded_ldl_forward for bw <= 15 stored with a band stride of 16 doubles: four pivots per panel,
// the whole trailing update of a panel done by ONE v_mfma_f64_16x16x4.
//   * the panel's four columns (16 entries each, replicated in all four lane groups) live in registers;
//     pivot k updates the later panel columns with DPP row_shl lane shifts and a readlane multiplier,
//     exactly like the two-column chain of banded_ldl_forward;
//   * the 16 x 16 trailing window, rows / columns c+4 .. c+19, is the MFMA accumulator: element (i,j) sits
//     in register i/4, lane 16 (i%4) + j, so  D = C - Y V'  with  A[i][k] = L[c+4+i, c+k]  (lane (k,i)) and
//     B[k][j] = unscaled column entry (lane (k,j)) - both formed from the panel's registers by one in-row DPP shift per
//     lane group (p4_operand; round 3: they used to be stored and re-read in that lane order, an LDS round trip on the chain);
//   * column 15 of the window (matrix column c+19) is never touched by a panel at c (reach c+18), so it
//     carries the right-hand side instead: lanes with j == 15 address rhs[c+4+i], and B[k][15] = u_k (the
//     panel's finished rhs entries, a 4 x 4 forward substitution on wave-uniform values): the forward
//     substitution of all later rows rides along in the same MFMA.
// Only the lower triangle of the window is loaded / stored (other lanes park garbage in private dummies).
// n_elim % 4 leftover pivots go through banded_ldl_forward on a shifted view.  ~29 instructions per pivot
// instead of 52.
typedef double v4f64_t __attribute__((ext_vector_type(4)));

template <int S>
__device__ __forceinline__ double row_shl_zero(double v) {       // lane i <- lane i+S inside its 16-lane row, else 0
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + S, 0xF, 0xF, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + S, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// rows (16-lane groups) named by the mask RM <- lane i+S of v inside the row (0 beyond it); the other rows keep `old`
template <int S, int RM>
__device__ __forceinline__ double row_shl_into(double old, double v) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x100 + S, RM, 0xF, true);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x100 + S, RM, 0xF, true);
  return __hiloint2double(hi, lo);
}
// the MFMA operand of lane (k = g, j): column k of the panel, entry 4 + j - k (zero past entry 15) - the four panel
// columns are replicated in every lane group, so this is one in-row shift per group, no LDS round trip
__device__ __forceinline__ double p4_operand(double c0, double c1, double c2, double c3) {
  double x = row_shl_zero<4>(c0);
  x = row_shl_into<3, 0x2>(x, c1);
  x = row_shl_into<2, 0x4>(x, c2);
  return row_shl_into<1, 0x8>(x, c3);
}

template <int NPL>
__device__ inline bool banded_ldl_forward_p4(double* lds, const VbLayout L, int n, int bw, int n_elim) {
  constexpr int R1 = 16;             // band stride (L.R1 == 16): columns of bw + 1 <= 16 words, zero beyond
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, j = lane & 15;
  if (n_elim < 0) n_elim = n;
  const int npan = n_elim / 4;
  // uniform-stride addresses (all lanes valid; the four lane groups hold / store identical copies)
  int cl = 8 * (L.band + j);                                   // column c+q, entry j       at +128 q
  int ia = 8 * L.invd;                                         // invd[c+q]                 at +8 q
  int ra = 8 * L.rhs;                                          // rhs[c+q]                  at +8 q
  // window (accumulator) addresses: register r, lane (g,j) <-> element (4r+g, j)
  int ca[4], cs[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + g;
    if (j == 15) { ca[r] = 8 * (L.rhs + 4 + i); cs[r] = 32; }                                   // rhs[c+4+i]
    else if (i >= j) { ca[r] = 8 * (L.band + (4 + j) * R1 + (i - j)); cs[r] = 4 * R1 * 8; }     // A[c+4+i, c+4+j]
    else { ca[r] = 8 * (L.dummy + 64 * (1 + r) + lane); cs[r] = 0; }                            // upper triangle: unused
  }
  // MFMA operands: lane (k = g, i|j = j), built from the panel's registers (p4_operand); column 15 of B carries the
  // panel's finished rhs entries u_k: one-hot lane masks (lane 16 k + 15) blend them in
  const double m15_0 = lane == 15 ? 1.0 : 0.0, m15_1 = lane == 31 ? 1.0 : 0.0, m15_2 = lane == 47 ? 1.0 : 0.0, m15_3 = lane == 63 ? 1.0 : 0.0;
  bool bad = false;
  for (int pnl = 0; pnl < npan; ++pnl) {
    // ---- loads: panel columns first (the pivot chain waits for nothing else), then the panel's rhs entries
    //      and the window, which are not needed before the chain is through
    double C0 = ldsr(lds, cl), C1 = ldsr(lds, cl + 128), C2 = ldsr(lds, cl + 256), C3 = ldsr(lds, cl + 384);
    const double r0 = ldsr(lds, ra), r1 = ldsr(lds, ra + 8), r2 = ldsr(lds, ra + 16), r3 = ldsr(lds, ra + 24);
    __builtin_amdgcn_sched_barrier(0);       // keep the window loads BEHIND the column loads (the scheduler hoists them)
    v4f64_t W;
    W[0] = ldsr(lds, ca[0]); W[1] = ldsr(lds, ca[1]); W[2] = ldsr(lds, ca[2]); W[3] = ldsr(lds, ca[3]);
    // ---- pivot 0
    const double p0 = bcast_first(C0);
    const double i0 = rcp_nr(p0);
    const double Y0 = C0 * i0;
    const double y01 = bcast_lane(Y0, 1), y02 = bcast_lane(Y0, 2), y03 = bcast_lane(Y0, 3);
    C1 = fma(-row_shl_zero<1>(C0), y01, C1);
    C2 = fma(-row_shl_zero<2>(C0), y02, C2);
    C3 = fma(-row_shl_zero<3>(C0), y03, C3);
    // ---- pivot 1
    const double p1 = bcast_first(C1);
    const double i1 = rcp_nr(p1);
    const double Y1 = C1 * i1;
    const double y11 = bcast_lane(Y1, 1), y12 = bcast_lane(Y1, 2);
    C2 = fma(-row_shl_zero<1>(C1), y11, C2);
    C3 = fma(-row_shl_zero<2>(C1), y12, C3);
    // ---- pivot 2
    const double p2 = bcast_first(C2);
    const double i2 = rcp_nr(p2);
    const double Y2 = C2 * i2;
    const double y21 = bcast_lane(Y2, 1);
    C3 = fma(-row_shl_zero<1>(C2), y21, C3);
    // ---- pivot 3
    const double p3 = bcast_first(C3);
    const double i3 = rcp_nr(p3);
    const double Y3 = C3 * i3;
    // ---- the panel's rhs entries (wave-uniform 4 x 4 forward substitution)
    const double u0 = r0;
    const double u1 = fma(-y01, u0, r1);
    const double u2 = fma(-y11, u1, fma(-y02, u0, r2));
    const double u3 = fma(-y21, u2, fma(-y12, u1, fma(-y03, u0, r3)));
    // ---- stores: L columns (in place), unscaled columns, 1/D, finished rhs
    ldsw(lds, cl, Y0); ldsw(lds, cl + 128, Y1); ldsw(lds, cl + 256, Y2); ldsw(lds, cl + 384, Y3);
    ldsw(lds, ia, i0); ldsw(lds, ia + 8, i1); ldsw(lds, ia + 16, i2); ldsw(lds, ia + 24, i3);
    ldsw(lds, ra, u0); ldsw(lds, ra + 8, u1); ldsw(lds, ra + 16, u2); ldsw(lds, ra + 24, u3);
    // ---- trailing window: W -= Y V'  (column 15: rhs[c+4+i] -= sum_k L[c+4+i, c+k] u_k)
    const double A = p4_operand(Y0, Y1, Y2, Y3);
    const double B = fma(m15_3, u3, fma(m15_2, u2, fma(m15_1, u1, fma(m15_0, u0, p4_operand(C0, C1, C2, C3)))));
    W = __builtin_amdgcn_mfma_f64_16x16x4f64(-A, B, W, 0, 0, 0);
    ldsw(lds, ca[0], W[0]); ldsw(lds, ca[1], W[1]); ldsw(lds, ca[2], W[2]); ldsw(lds, ca[3], W[3]);
    cl += 4 * R1 * 8; ia += 32; ra += 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) ca[r] += cs[r];
  }
  // ---- every pivot of the panels positive?  Read off their stored reciprocals, once, after the chain (a compare and an
  //      s_or per pivot on the chain otherwise): p <= 0 or NaN leaves 1/p <= 0, inf or NaN
  {
    bool bad_l = false;
    for (int e = lane; e < 4 * npan; e += WAVE) {
      const double iv = lds[L.invd + e];
      bad_l |= !(iv > 0.0 && iv < __builtin_huge_val());
    }
    bad = __ballot(bad_l) != 0ULL;
  }
  // ---- leftover pivots (and the flush of the register-resident columns) on a view shifted by 4 npan
  VbLayout L2 = L;
  const int nn0 = 4 * npan;
  L2.band += nn0 * R1; L2.rhs += nn0; L2.invd += nn0;
  const bool rest = banded_ldl_forward<NPL, true>(lds, L2, n - nn0, bw, n_elim - nn0);
  return rest && !bad;
}

// x = L^-T w for the unit-lower band factor; w = rhs in/out.  Wave 0 only; bw >= 3.
// Lane = column mod (bw+1) keeps the live unknowns in registers.  The factor entry
// L[r, p(lane)] and the fresh right-hand side of step r are fetched three steps ahead into
// registers renamed (not copied) by the 3-way unrolled loop; the per-lane load address
// just walks down the band column (-8 bytes per step) and jumps to the lane's next column
// when the look-ahead stream passes a retirement, so a step is ~25 instructions.
__device__ inline void banded_unit_backward(double* lds, const VbLayout L, int n, int bw) {
  const int lane = threadIdx.x & 63;
  const int R1 = L.R1, Rw = bw + 1;
  const bool act = lane < Rw;
  const int dmy = 8 * (L.dummy + lane);
  // lane owns the column p == lane (mod Rw) of the sliding window (r-Rw, r]
  const int p0 = (n - 1) - ((n - 1 - lane) % Rw + Rw) % Rw;   // first (largest) column of this lane
  double wv = (act && p0 >= 0) ? lds[L.rhs + p0] : 0.0;
  int own = __builtin_amdgcn_readfirstlane((n - 1) % Rw);       // lane that retires at step r
  // look-ahead stream (target step r-3 .. ): byte address of L[rt, p] = 8*(band + p*(R1-1) + rt)
  const int jump = 8 * Rw * (R1 - 1);
  int la = act ? 8 * (L.band + p0 * (R1 - 1) + (n - 1)) : dmy;
  const int lstep = act ? 8 : 0, ljump = act ? jump : 0;
  int lown = own;                                               // lane retiring at the look-ahead target
  int wa = 8 * (L.rhs + n - 1 - Rw);                            // &w[rt - Rw]  (front-padded with zeros)
  auto fetch = [&](double& Lc, double& Wc) {                    // issue the loads for the next target
    Lc = ldsr(lds, la);      // at rt == p (the retirement itself) this is the diagonal: unused
    Wc = ldsr(lds, wa);
    la -= lstep + (lane == lown ? ljump : 0);
    wa -= 8;
    lown = lown == 0 ? Rw - 1 : lown - 1;
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

// Blocked back-substitution: x = L^-T w, B = bw+1 columns per block, lane i of the block owns column c+i
// for the whole block (no retirement bookkeeping).  Per block
//   rect part, d = 1..bw:  wv_i -= L[c+i+d, c+i] * x[c+i+d] for the x of the PREVIOUS block, which sit in
//                          lanes 0..B-1 of prevX: lane i needs lane i+d-B, i.e. a lane shift by B-d with
//                          zero fill (DPP row_shr inside a 16-lane row, ds_bpermute otherwise) - lanes
//                          whose x belongs to the current block get 0 by construction, no masks;
//   tri part,  q = B-1..0: x_q = wv_q (readlane), kept by v_writelane; lanes i < q subtract
//                          L[c+q, c+i] x_q.  Lanes i >= q read in-bounds garbage into accumulators
//                          that are already retired.
// All 2*bw+1 LDS loads of a block have per-lane base + immediate addresses and do not depend on the
// chain; one load of w and one store of x per block.  ~10 instructions per column against ~25 for the
// sliding-window routine above.  Columns n..c0+B-1 of the top block are the zero padding.
template <int S, bool ROW16>
__device__ __forceinline__ double lane_shr_zero(double v, int bidx) {      // lane i <- lane i-S, 0 if i < S
  if constexpr (S == 0) {
    return v;
  } else if constexpr (ROW16) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + S, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + S, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  } else {   // lanes >= B of v are 0 and B <= 32: a wrapped source lane (i - S + 64 >= 32) reads 0
    int lo = __builtin_amdgcn_ds_bpermute(bidx - 4 * S, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(bidx - 4 * S, __double2hiint(v));
    return __hiloint2double(hi, lo);
  }
}

template <int B, bool ROW16, int D>
struct BackRect {   // d = D..B-1
  static __device__ __forceinline__ void load(const double* lds, int cb, double (&Lr)[B]) {
    if constexpr (D < B) { Lr[D] = ldsr(lds, cb + 8 * D); BackRect<B, ROW16, D + 1>::load(lds, cb, Lr); }
  }
  static __device__ __forceinline__ void apply(const double (&Lr)[B], double prevX, int bidx, double& wv) {
    if constexpr (D < B) {
      wv = fma(-Lr[D], lane_shr_zero<B - D, ROW16>(prevX, bidx), wv);
      BackRect<B, ROW16, D + 1>::apply(Lr, prevX, bidx, wv);
    }
  }
};
template <int B, int Q>
struct BackTri {    // q = Q..0
  static __device__ __forceinline__ void load(const double* lds, int tb, double (&Lt)[B]) {
    if constexpr (Q >= 1) { Lt[Q] = ldsr(lds, tb + 8 * Q); BackTri<B, Q - 1>::load(lds, tb, Lt); }
  }
  static __device__ __forceinline__ void apply(const double (&Lt)[B], double& wv, int& xlo, int& xhi) {
    if constexpr (Q >= 0) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(wv), Q);
      const int hi = __builtin_amdgcn_readlane(__double2hiint(wv), Q);
      asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(xlo) : "s"(lo), "n"(Q));
      asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(xhi) : "s"(hi), "n"(Q));
      if constexpr (Q >= 1) {
        wv = fma(-Lt[Q], __hiloint2double(hi, lo), wv);
        BackTri<B, Q - 1>::apply(Lt, wv, xlo, xhi);
      }
    }
  }
};

template <int B, bool ROW16>
__device__ inline void banded_unit_backward_blk(double* lds, const VbLayout L, int n) {
  static_assert(B >= 2 && B <= 32 && (!ROW16 || B <= 16), "block = bw+1 lanes");
  const int lane = threadIdx.x & 63;
  const int R1 = L.R1;
  const bool act = lane < B;
  const int c0 = ((n - 1) / B) * B;
  // byte addresses; lanes >= B read in-bounds words (results unused) and write to private dummies
  int cb = act ? 8 * (L.band + (c0 + lane) * R1) : 8 * L.band;          // &L[c+lane+d, c+lane] at +8d
  int tb = act ? cb - 8 * lane : 8 * L.band;                            // &L[c+q, c+lane]       at +8q
  int wa = act ? 8 * (L.rhs + c0 + lane) : 8 * (L.dummy + lane);        // w in / x out
  const int bstep = act ? 8 * B * R1 : 0, wstep = act ? 8 * B : 0;
  const int bidx = 4 * lane;
  double prevX = 0.0;
  for (int c = c0; c >= 0; c -= B) {
    double Lr[B], Lt[B];
    double wv = ldsr(lds, wa);
    BackRect<B, ROW16, 1>::load(lds, cb, Lr);
    BackTri<B, B - 1>::load(lds, tb, Lt);
    if (!act) wv = 0.0;
    BackRect<B, ROW16, 1>::apply(Lr, prevX, bidx, wv);
    int xlo = 0, xhi = 0;                                               // lanes >= B stay 0
    BackTri<B, B - 1>::apply(Lt, wv, xlo, xhi);
    prevX = __hiloint2double(xhi, xlo);
    ldsw(lds, wa, prevX);
    cb -= bstep; tb -= bstep; wa -= wstep;
  }
}

// ---- back-substitution with the sequential part reduced to one mat-vec per 16 columns (band stride 16, bw <= 15) ----
// x = L^-T w in blocks of 16 columns: block b at column c = 16 b couples only to block b+1 (bw < 16), with
//   T_b[i][j] = L[c+i, c+j]  (unit lower triangular)   and   N_b[i][j] = L[c+16+i, c+j]  (non-zero for i < j), so
//   x_b = T_b^-T (w_b - N_b' x_{b+1}) = y_b - M_b x_{b+1},     y_b = T_b^-T w_b,   M_b = T_b^-T N_b'.
// y_b and the 16 columns of M_b are 17 independent 16-step triangular solves per block that ALL blocks of both views do
// at once (backpar16_prepare: every thread of the workgroup takes one or two of them, results back in place - M_b over the
// block's own band storage, y_b over w_b), which leaves one dependent 16 x 16 mat-vec per block for the chain
// (backpar16_chain: ~60 instructions per 16 columns against ~10 per column for banded_unit_backward_blk).
// T_b^-T is formed implicitly, solve by solve; its conditioning is that of a 16 x 16 piece of the factor (<= sqrt cond Q).
__device__ inline void backpar16_prepare(double* lds, const VbLayout LA, int nbA, const VbLayout LB, int nbB, int tid, int nthreads) {
  constexpr int MAXT = 2;
  const int ntask = (nbA + nbB) * 17;
  double s[MAXT][16];
  int dst[MAXT], dstep[MAXT], dmask[MAXT];
#pragma unroll
  for (int u = 0; u < MAXT; ++u) {
    const int t = tid + u * nthreads;
    const bool has = t < ntask;
    const int tt = has ? t : 0;
    const int blk = tt / 17, cl = tt - blk * 17;
    const bool inA = blk < nbA;
    const int c = 16 * (inA ? blk : blk - nbA);
    const int band0 = (inA ? LA.band : LB.band) + c * 16, rhs0 = (inA ? LA.rhs : LB.rhs) + c;
    double r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      // column cl of N_b' (entry j: N_b[cl][j], non-zero for j > cl), or the right-hand side w_b for cl == 16
      const int a = cl == 16 ? rhs0 + j : band0 + j * 16 + (j > cl ? 16 + cl - j : 0);
      const double v = lds[a];
      r[j] = (cl == 16 || j > cl) ? v : 0.0;
    }
#pragma unroll
    for (int j = 15; j >= 0; --j) {        // T_b' s = r, unit upper triangular: s_j = r_j - sum_{i > j} T_b[i][j] s_i
      double acc = r[j];
#pragma unroll
      for (int i = j + 1; i < 16; ++i) acc = fma(-lds[band0 + j * 16 + (i - j)], s[u][i], acc);
      s[u][j] = acc;
    }
    // where the results go: y_b over w_b; column cl of M_b into the block's own storage, entry M_b[j][cl] at word
    // cl * 16 + ((j + cl) & 15) - the chain's lane j then reads M_b[j][i], i fixed, at consecutive words (no bank
    // conflict; rows of 16 doubles at stride 16 would put all sixteen lanes on one bank), and so do these writes.
    // Tasks beyond the list park theirs in the thread's private dummy word.
    dst[u] = !has ? LA.dummy + 64 + (tid & 63) : (cl == 16 ? rhs0 : band0 + cl * 16);
    dstep[u] = !has || cl == 16 ? 0 : cl;                  // word (j + dstep) & dmask of the destination
    dmask[u] = !has ? 0 : (cl == 16 ? 31 : 15);
  }
  __syncthreads();                         // every solve has read its block before any block is overwritten
#pragma unroll
  for (int u = 0; u < MAXT; ++u)
#pragma unroll
    for (int j = 0; j < 16; ++j) lds[dst[u] + ((j + dstep[u]) & dmask[u])] = s[u][j];
  __syncthreads();
}
// the chain of one view (one wave; lanes 16.. repeat lanes 0..15): x_b = y_b - M_b x_{b+1}, top block first
__device__ inline void backpar16_chain(double* lds, const VbLayout L, int nb) {
  const int j = threadIdx.x & 15;
  double xprev = 0.0;
  for (int b = nb - 1; b >= 0; --b) {
    const int c = 16 * b;
    const double* Mb = lds + L.band + c * 16;
    double m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = Mb[i * 16 + ((j + i) & 15)];      // M_b[j][i] (the swizzle of backpar16_prepare)
    double acc = lds[L.rhs + c + j];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = fma(-m[i], bcast_lane(xprev, i), acc);
    xprev = acc;
    lds[L.rhs + c + j] = acc;
  }
}

// the same chain on the matrix cores: the mat-vec M_b x_{b+1} as four v_mfma_f64_16x16x4 (A = four columns of M_b, B = four
// entries of x_{b+1} repeated along the 16 result columns).  The accumulator of block b - element r of lane (g, .) =
// x_b[4 r + g] - IS the B operand of block b-1 (lane (k, .) of sub-step m wants x[4 m + k]): no lane movement at all
// between blocks, four dependent MFMAs per 16 columns.
__device__ inline void backpar16_chain_mfma(double* lds, const VbLayout L, int nb) {
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  v4f64_t x = {0.0, 0.0, 0.0, 0.0};
  for (int b = nb - 1; b >= 0; --b) {
    const int c = 16 * b;
    const double* Mb = lds + L.band + c * 16;
    double A[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) { const int i = 4 * m + g; A[m] = Mb[i * 16 + ((j + i) & 15)]; }      // M_b[j][i] (backpar16_prepare's swizzle)
    v4f64_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = lds[L.rhs + c + 4 * r + g];                                      // y_b
#pragma unroll
    for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-A[m], x[m], acc, 0, 0, 0);
    x = acc;
    if (j == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[L.rhs + c + 4 * r + g] = acc[r];
    }
  }
}

// dispatch on bw (wave-uniform); false if no blocked variant exists for this band width
template <bool ROW16>
__device__ inline bool banded_unit_backward_auto(double* lds, const VbLayout L, int n, int bw) {
#define BTF_BACK_CASE(BB) case BB - 1: banded_unit_backward_blk<BB, ROW16>(lds, L, n); return true;
  if constexpr (ROW16) {
    switch (bw) {
      BTF_BACK_CASE(4) BTF_BACK_CASE(5) BTF_BACK_CASE(6) BTF_BACK_CASE(7) BTF_BACK_CASE(8) BTF_BACK_CASE(9)
      BTF_BACK_CASE(10) BTF_BACK_CASE(11) BTF_BACK_CASE(12) BTF_BACK_CASE(13) BTF_BACK_CASE(14) BTF_BACK_CASE(15)
      BTF_BACK_CASE(16)
      default: break;
    }
  } else {
    switch (bw) {
      BTF_BACK_CASE(17) BTF_BACK_CASE(19) BTF_BACK_CASE(21) BTF_BACK_CASE(22) BTF_BACK_CASE(25) BTF_BACK_CASE(28)
      BTF_BACK_CASE(29) BTF_BACK_CASE(31)
      default: break;
    }
  }
#undef BTF_BACK_CASE
  banded_unit_backward(lds, L, n, bw);
  return false;
}

template <int NPL, bool ROW16>
__global__ __launch_bounds__(VB_THREADS) void v_banded_fast_kernel(VBandArgs a, int K) {
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

  const VbLayout L = vb_layout(T, K, a.TF, a.weighted);
  double* Bc = lds + L.band;
  double* rhs = lds + L.rhs;
  double* m0 = lds + L.m0;
  double* zs = lds + L.zs;
  double* invd = lds + L.invd;
  double* P = lds + L.P;
  double* Ql = lds + L.Ql;
  double* flag = lds + L.flag;
  long long stamp[6];
  stamp[0] = __builtin_amdgcn_s_memtime();

  // ---- likelihood mean part / Gram blocks / prior band (all threads) ----------------
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
  for (int idx = tid; idx < n; idx += VB_THREADS) {
    const int t = idx / K, k = idx - t * K;
    m0[idx] = chunk_sum(a.part + (size_t)k * a.ld + (size_t)j * T + t) * a.s;
  }
  if (a.weighted) {
    for (int idx = tid; idx < T * KK; idx += VB_THREADS) {
      const int t = idx / KK, q = idx - t * KK;
      Ql[idx] = chunk_sum(a.part + (size_t)(K + q) * a.ld + (size_t)jq * T + t) * a.s;
    }
  } else {
    reduce_gram(a.gpart, a.ngp, KK, a.sR, Bc, Ql);   // Bc is free scratch until the assembly
  }
  for (int idx = tid; idx < T * D1; idx += VB_THREADS) P[idx] = a.pband[(size_t)j * T * D1 + idx];
  const int npad = L.npad;
  for (int idx = n + tid; idx < npad; idx += VB_THREADS) m0[idx] = 0.0;
  // zero pads: band tail (+64), rhs front pad and tail, scratch row, dummy words
  for (int idx = tid; idx < 64; idx += VB_THREADS) {
    Bc[npad * R1 + idx] = 0.0;
    rhs[npad + idx] = 0.0;
    lds[L.vsc + idx] = 0.0;
  }
  for (int idx = tid; idx < L.FP; idx += VB_THREADS) rhs[idx - L.FP] = 0.0;
  for (int idx = tid; idx < 64 * 9 + 8; idx += VB_THREADS) lds[L.dummy + idx] = 0.0;
  __syncthreads();
  stamp[1] = __builtin_amdgcn_s_memtime();

  double shift = 0.0, eps = a.eps0;
  int tried = 0;
  bool ok = false;
  while (true) {
    for (int nn = tid; nn < npad; nn += VB_THREADS) {      // one band column per thread, no divisions inside
      double* colw = Bc + (size_t)nn * R1;
      if (nn >= n) {
        for (int aa = 0; aa < R1; ++aa) colw[aa] = 0.0;
        continue;
      }
      const int t = nn / K, k = nn - t * K;
      const double* q = a.weighted ? Ql + t * KK : Ql;
      int dd = 0, rem = 0;                                   // aa = dd*K + rem
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
    for (int idx = tid; idx < npad; idx += VB_THREADS) rhs[idx] = m0[idx];
    __syncthreads();
    stamp[2] = __builtin_amdgcn_s_memtime();
    if (wave == 0) {
      bool good;
      if constexpr (ROW16) good = (a.panel4 && bw == 15) ? banded_ldl_forward_p4<NPL>(lds, L, n, bw, n) : banded_ldl_forward<NPL, ROW16>(lds, L, n, bw);
      else good = banded_ldl_forward<NPL, ROW16>(lds, L, n, bw);
      if (tid == 0) flag[0] = good ? 1.0 : 0.0;
    } else if (tried == 0) {
      // the normals of this column, depth-major index (drawn once, whatever the retries)
      for (int idx = tid - WAVE; idx < n; idx += VB_THREADS - WAVE)
        zs[idx] = a.z ? a.z[(size_t)jg * n + idx] : philox_normal(a.seed, a.stream, (unsigned long long)jg * n + idx);
    }
    __syncthreads();
    ok = flag[0] != 0.0;
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
  // w = D^-1 u + D^-1/2 z
  for (int idx = tid; idx < n; idx += VB_THREADS) {
    const double iv = invd[idx];
    rhs[idx] = fma(rhs[idx], iv, zs[idx] * sqrt(iv));
  }
  __syncthreads();
  stamp[4] = __builtin_amdgcn_s_memtime();
  if (wave == 0) banded_unit_backward_auto<ROW16>(lds, L, n, bw);
  __syncthreads();
  stamp[5] = __builtin_amdgcn_s_memtime();
  for (int idx = tid; idx < n; idx += VB_THREADS) a.V[(size_t)jg * n + idx] = rhs[idx];
  if (a.gout) {   // this column's share of V'V (rows (j,t), t = 0..T-1): two fixed-order levels
    __syncthreads();
    const int ng = VB_THREADS / KK, g = tid / KK, q = tid - g * KK;
    int p = 0;
    while ((p + 1) * (p + 2) / 2 <= q) ++p;
    const int pq = q - p * (p + 1) / 2;
    if (g < ng) {
      double s = 0.0;
      for (int t = g; t < T; t += ng) s = fma(rhs[t * K + p], rhs[t * K + pq], s);
      Bc[g * KK + q] = s;                  // the band is dead by now: scratch
    }
    __syncthreads();
    if (tid < KK) {
      double s = 0.0;
      for (int b = 0; b < ng; ++b) s += Bc[b * KK + tid];
      a.gout[(size_t)j * KK + tid] = s;
    }
  }
  if (a.dbg && tid == 0)
    for (int i = 0; i < 6; ++i) a.dbg[(size_t)j * 6 + i] = stamp[i];
}

}  // namespace btf
