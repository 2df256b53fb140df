// Hand-written CDNA4 kernels of the BTF Gibbs core (fp64 throughout).
//
// Data layout in HBM (DESIGN.md "Layout"):
//   A_v  [N][ldv]   linear statistic, lanes run along (j,t)   -> V half-sweep, SSE
//   A_wT [MT][ldw]  the same numbers transposed, lanes run along i -> W half-sweep
//   C_v / C_wT      per-cell precision weights (replicate counts or PG omegas);
//                   absent on the complete-data Gaussian path
// Both half-sweeps are the same streaming contraction
//   out[l][k] = sum_r X[r][l] * U[r][k]          (+ sum_r C[r][l] * U[r][k]U[r][k'])
// with the reduction index r wave-uniform: U[r][:] is fetched with scalar loads
// and used as the SGPR operand of v_fma_f64, every lane owns two adjacent outputs
// l (one 16-byte global_load per row), and no cross-lane reduction is needed.
#pragma once
#include <type_traits>
#include "btf_device.h"
#include "btf_eig.h"
#include "btf_pg_exact.h"

namespace btf {

// ============================================================================
// streaming accumulation  (BTF_K_W_ACCUM / BTF_K_V_ACCUM)
// ============================================================================
#ifndef BTF_ACC_WAVES
#define BTF_ACC_WAVES 16
#endif
#ifndef BTF_DF_UNR
#define BTF_DF_UNR 4
#endif
#ifndef BTF_DF_PREFETCH
#define BTF_DF_PREFETCH 1
#endif
#ifndef BTF_DF_SHARE
#define BTF_DF_SHARE 0    // percent of a workgroup's rows the chain waves of the dataflow V tail stream too (0: none)
#endif
#ifndef BTF_ACC_UNR
#define BTF_ACC_UNR 2
#endif
#ifndef BTF_ACC_PF_WT
#define BTF_ACC_PF_WT 0   // weighted modes: software-pipelined accumulation loop (A/B aid; measured slower, see below)
#endif
#ifndef BTF_ACC_UNR_WT
#define BTF_ACC_UNR_WT 2     // weighted modes: rows in flight per wave (and per buffer of the pipelined loop)
#endif
#ifndef BTF_ACC_WAVES_WT
#define BTF_ACC_WAVES_WT 12   // weighted modes, K <= 5: waves per workgroup
#endif
#ifndef BTF_ACC_WAVES_K10
#define BTF_ACC_WAVES_K10 8   // complete data, K = 10: waves per workgroup
#endif
#ifndef BTF_ACC_UNR_K9
#define BTF_ACC_UNR_K9 2      // complete data, K >= 9: rows in flight per wave
#endif
#ifndef BTF_ACC_ULDS_MINK
#define BTF_ACC_ULDS_MINK 8   // complete data: from this K on the factor rows are staged in LDS (accum_kernel, ULDS)
#endif
constexpr int ACC_WAVES = BTF_ACC_WAVES;   // waves per workgroup
constexpr int ACC_THREADS = ACC_WAVES * WAVE;
constexpr int ACC_TILE = 2 * WAVE;      // outputs per workgroup along the lane axis
constexpr int ACC_UNR = BTF_ACC_UNR;       // rows in flight per wave
constexpr int ACC_RG = (ACC_THREADS / ACC_TILE) < 4 ? (ACC_THREADS / ACC_TILE) : 4;   // values reduced per LDS round

__device__ inline void reduce_gram(const double* __restrict__ gpart, int ngp, int KK, double scale, double* scratch, double* G);

// curve-structured replicate counts (see the section above colgram_kernel): CSR lists of deficient partners
struct CurveLists {
  const int* ptr;        // [n + 1]
  const int* idx;        // partner index (column for a row list, row for a column list)
  const double* def;     // R - c_ij > 0
};
// the per-column Gram of curve-structured counts: Ql (= sR W'W, LDS) -= scale sum_{i in D(j)} (R - c_ij) w_i w_i'.
// Workgroup-collective (barriers inside; every thread of the workgroup calls it, ptr == nullptr returns at once):
// the listed rows of W are staged in LDS, blockDim/K of them per pass - two dependent global round trips per
// pass instead of two per row - and threads 0..KK-1 add the rank-one terms in list order.  scratch: at least
// (blockDim/K)(K+1) doubles, or fewer rows go into a pass.  The caller's next barrier publishes Ql.
__device__ __forceinline__ void curve_column_gram(const CurveLists& cv, const double* __restrict__ W, int jg, int K, int KK,
                                         double scale, double* Ql, double* scratch, int scratch_doubles) {
  if (!cv.ptr) return;
  const int e0 = cv.ptr[jg], e1 = cv.ptr[jg + 1];
  if (e0 == e1) return;                                  // (uniform: jg is the workgroup's column)
  const int q = threadIdx.x;
  int p = 0;
  while ((p + 1) * (p + 2) / 2 <= q) ++p;
  const int pq = q - p * (p + 1) / 2;
  double corr = 0.0;
  if (e1 - e0 <= 6) {                                    // a handful of rows: straight from memory, no barriers
    if (q < KK) {
      for (int e = e0; e < e1; ++e) {
        const double* __restrict__ w = W + (size_t)cv.idx[e] * K;
        corr = fma(cv.def[e] * w[p], w[pq], corr);
      }
      Ql[q] = fma(-scale, corr, Ql[q]);
    }
    return;
  }
  int per = (int)blockDim.x / K;
  if (per * (K + 1) > scratch_doubles) per = scratch_doubles / (K + 1);
  const int el = threadIdx.x / K, k = threadIdx.x - el * K;
  double* rows = scratch;                                // [per][K]
  double* defs = scratch + per * K;                      // [per]
  for (int eb = e0; eb < e1; eb += per) {
    const int e = eb + el;
    if (el < per && e < e1) {
      const int i = cv.idx[e];
      rows[el * K + k] = W[(size_t)i * K + k];
      if (k == 0) defs[el] = cv.def[e];
    }
    __syncthreads();
    if (q < KK) {
      const int cnt = min(per, e1 - eb);
      for (int u = 0; u < cnt; ++u) corr = fma(defs[u] * rows[u * K + p], rows[u * K + pq], corr);
    }
    __syncthreads();
  }
  if (q < KK) Ql[q] = fma(-scale, corr, Ql[q]);
}

// Packed lower-triangular Gram accumulators of a side task live in registers next to the streaming kernel's budget
// (128 VGPRs under a 1024-thread launch bound): K(K+1)/2 doubles fit for K <= 9; K = 10 (55 entries, 110 VGPRs) is
// summed in two passes over the rows of the packed triangle - the factor rows are read twice (they are tiny).
template <int K>
struct GramPasses {
  static constexpr int NP = tri(K) > 45 ? 2 : 1;
  static constexpr int SPLIT = NP == 2 ? (K * 7 + 9) / 10 : K;          // rows [0, SPLIT) then [SPLIT, K): 28 + 27 entries at K = 10
  __host__ __device__ static constexpr int p0(int pass) { return pass == 0 ? 0 : SPLIT; }
  __host__ __device__ static constexpr int p1(int pass) { return NP == 1 || pass == 1 ? K : SPLIT; }
};
// acc[lidx(p,q) - lidx(P0,0)] += d * u[p] * u[q] for P0 <= p < P1
template <int K, int P0, int P1>
__device__ __forceinline__ void gram_rank1(const double (&u)[K], double d, double (&acc)[tri(P1) - tri(P0)]) {
#pragma unroll
  for (int p = P0; p < P1; ++p)
#pragma unroll
    for (int q = 0; q <= p; ++q) acc[lidx(p, q) - tri(P0)] = fma(d * u[p], u[q], acc[lidx(p, q) - tri(P0)]);
}
// rows r = rbeg, rbeg + rstep, ... < rend of U (K doubles each): this wave's sums of u u' (times dfn(r) if given) to
// out[q], q = 0..KK-1 (lane 0 writes)
template <int K, int PASS, class RowFn>
__device__ __forceinline__ void gram_pass_wave(RowFn row_of, int ebeg, int eend, int estep, double* out) {
  constexpr int P0 = GramPasses<K>::p0(PASS), P1 = GramPasses<K>::p1(PASS), NQ = tri(P1) - tri(P0);
  const int lane = threadIdx.x & 63;
  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
  for (int e = ebeg; e < eend; e += estep) {
    double u[K], d;
    row_of(e, u, d);
    gram_rank1<K, P0, P1>(u, d, acc);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double t = wave_sum(acc[q]);
    if (lane == 0) out[tri(P0) + q] = t;
  }
  if constexpr (PASS + 1 < GramPasses<K>::NP) gram_pass_wave<K, PASS + 1>(row_of, ebeg, eend, estep, out);
}

// The Gram of a tall factor on the matrix cores (side tasks, K <= 16): one v_mfma_f64_16x16x4_f64 per four rows with
// a == b - lane l holds U[4 g + (l >> 4)][l & 15] (zero beyond K or the last row), so every row group costs one load
// and one MFMA per wave and the 16 x 16 result comes out summed over the wave's rows: no cross-lane reduction at all
// (the register form above ends in K(K+1)/2 six-step shuffles per wave - 25 us for a side workgroup of 16 waves at
// K = 8, which kept its CU from the stream for half the launch on a rank's slab, round 3).  acc: row (l >> 4) + 4 i,
// column l & 15 in element i.  Groups g0 .. g1-1, eight loads in flight.
typedef double gram_f64x4 __attribute__((ext_vector_type(4)));
template <int K>
__device__ __forceinline__ void gram_mfma_groups(const double* __restrict__ U, int Rdim, int g0, int g1, gram_f64x4& acc) {
  static_assert(K <= 16, "one 16 x 16 tile");
  const int lane = threadIdx.x & 63, sub = lane >> 4, k = lane & 15;
  constexpr int UN = K >= 9 ? 4 : 8;                        // (K >= 9: the 128-VGPR kernels have no room for eight)
  for (int g = g0; g < g1; g += UN) {
    double x[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int r = 4 * (g + u) + sub;
      x[u] = (k < K && g + u < g1 && r < Rdim) ? U[(size_t)r * K + k] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u], x[u], acc, 0, 0, 0);
  }
}
// the whole workgroup: NW waves split the row groups [0, ceil(Rdim / 4)) of this workgroup's share (block b of nb) into
// contiguous runs, then the NW tiles are added in a fixed order; out[lidx(p, q)], q <= p < K (packed lower triangle).
// scr: NW * 256 doubles of LDS.  No barrier after the last write.
// (sc1: the partial is stored write-through - a tail of the same launch reads it, btf_fused.h)
template <int K, int NW>
__device__ __forceinline__ void gram_mfma_block(const double* __restrict__ U, int Rdim, int b, int nb, double* scr, double* out, bool sc1 = false) {
  static_assert(NW >= 4, "256 threads write the tile out");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int G = (Rdim + 3) >> 2;
  const int per = (G + nb * NW - 1) / (nb * NW);
  const int g0 = min(G, (b * NW + wave) * per), g1 = min(G, g0 + per);
  gram_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  gram_mfma_groups<K>(U, Rdim, g0, g1, acc);
#pragma unroll
  for (int i = 0; i < 4; ++i) scr[wave * 256 + ((lane >> 4) + 4 * i) * 16 + (lane & 15)] = acc[i];
  __syncthreads();
  if ((int)threadIdx.x < 256) {
    const int p = threadIdx.x >> 4, q = threadIdx.x & 15;
    if (p < K && q <= p) {
      double t = 0.0;
      for (int w = 0; w < NW; ++w) t += scr[w * 256 + threadIdx.x];
      if (sc1) store_sc1(&out[lidx(p, q)], t); else out[lidx(p, q)] = t;
    }
  }
}

// one-wave form (side tasks: a wave per column): lane l takes entries e0 + l, + 64, ...;
// out[q] = sum_{i in D(col)} (R - c_i,col) w_i[p] w_i[p'] (the caller scales and subtracts)
template <int K>
__device__ __forceinline__ void curve_column_sum_wave(const CurveLists& cv, const double* __restrict__ W, int col, double* out) {
  const int lane = threadIdx.x & 63;
  const int e0 = cv.ptr[col], e1 = cv.ptr[col + 1];
  gram_pass_wave<K, 0>([&](int e, double (&wr)[K], double& d) {
    const double* __restrict__ w = W + (size_t)cv.idx[e] * K;
    d = cv.def[e];
#pragma unroll
    for (int k = 0; k < K; ++k) wr[k] = w[k];
  }, e0 + lane, e1, WAVE, out);
}

// side tasks of the V accumulation launch (spectral sampler): workgroup 0 solves the eigen-problem of the shared
// Gram W'W, workgroups 1..ncols those of the listed curve columns' own Grams (their eigen-systems in eig_cols)
// (EIG_SIDE_TPW columns per workgroup, a wave each)
struct EigSideCols { const int* cols; int ncols; CurveLists cv; const double* W; double inv_R; double* eig_cols; };
__host__ __device__ constexpr int eig_side_tpw(int waves) { return waves >= 4 ? 4 : waves; }      // one per SIMD
__host__ __device__ constexpr int eig_side_groups(int ncols, int waves) { return (ncols + eig_side_tpw(waves) - 1) / eig_side_tpw(waves); }

// the horseshoe+ chain update (tau2_kernel below) as side workgroups of the W accumulation launch: it depends on V
// only and is transcendental-bound, the stream is memory-bound and leaves half the CUs idle at C3 - every 256-thread
// slice of a side workgroup takes one column
struct TauSide {
  const double* V; int T, nD, M;
  const int* dr_ptr; const int* dr_col; const double* dr_val;
  double lam2, lo, hi;
  double* Tau2; double* Ta; double* Tb; double* Tc; double* lsum;
  unsigned long long seed; const double* hyp;
};
__device__ void tau2_column(const TauSide& t, int K, int j, int tid256, double* red4);
// two columns per side workgroup: its waves land one per SIMD per column, and the gamma draws are f64-issue-bound
// (four columns: 14.4 us for the launch at C3, two: hidden under the 11 us stream); the other waves leave at once
constexpr int TAU_SIDE_CPW = 2;

// the Gram of the fixed factor (what gram_kernel computes) as side workgroups of the accumulation launch that the
// solve kernel FOLLOWS: nblocks partial Grams of U's rows, one per side workgroup, consumed across the kernel boundary
// (cnt != nullptr: the partials are consumed inside this launch - stored write-through, every block adds one to *cnt)
struct GramSide {
  const double* U; int Rdim; double* gpart; int nblocks; unsigned* cnt;
  // ONE more side workgroup: the sum of `sum_n` partial Grams [sum_n][KK] the previous V sampler left (one per column) into
  // sum_out[KK], in the order w_solve_kernel's reduce_gram would form it with `sum_threads` threads - so that every
  // workgroup of the W solve that follows reads KK doubles instead of all the partials (30 KB each at C3: a third of its cold
  // batch of loads) and lands on the same bits.  sum_src == nullptr: none.
  const double* sum_src; int sum_n; double* sum_out; int sum_threads;
};

// Which rows of the reduction axis a launch covers, and where its partial sums go: logical chunk l of the launch takes
// rows r0 .. min(r0 + rows_per_block, row_end) - 1 with r0 = row_base + l rows_per_block, moved up by `skip_rows` from
// `skip_at` on (a hole), and writes slot slot_base + l of the partials.  Sharded runs (BTF_OPT_SPLIT_ACCUM): the rows of
// the rank's OWN block of the fixed factor are accumulated first - that block needs no exchange - in chunks sized for
// the whole chip, and the rest in a second launch behind the all-gather, around the hole; the consumers just add all
// slots.  {0, 0, INT_MAX, 0, Rdim}: every row, slot = chunk.
struct ChunkMap {
  int slot_base, row_base, skip_at, skip_rows, row_end; int nside;     // nside: side workgroups in front (set by launch_accum)
  int nt;                  // 1: the statistic is streamed with non-temporal loads (slabs beyond the Infinity Cache: launch_accum)
#ifdef BTF_ACC_STAMPS
  long long* stamps;       // diagnostic builds: [8192][8] wall-clock stamps of the streaming workgroups (4..7: the fused tails)
#endif
};

// nu2 | rest and sigma2 | rest as ONE side workgroup of the W accumulation launch (full sweeps, rng="device", complete
// Gaussian data): the residual sum of squares comes from the per-column parts the spectral V sampler left behind at the
// end of the previous sweep (sse_cols: sum_t R v'W'Wv - 2 v.m with the W and V that still stand) plus the constants of
// the data, sum W^2 from W itself; w_solve - the next kernel - reads the two draws.  Same Philox streams as
// scalars_kernel.  hyp == nullptr: no such workgroup.
// (pub / flag / epoch: the fused W launch's tails read the draws inside the launch - btf_fused.h)
struct ScalarSide {
  const double* sse_cols; int M; double sconst, nobs;
  const double* W; int NK; double nfree;
  double nu2_a, nu2_b, sig_a, sig_b; int which; unsigned long long seed; double* hyp;
  double* pub; unsigned* flag; unsigned epoch;
};
// lam2 | rest (and lam2_a) as a side workgroup of the V accumulation launch: it needs the column sums the Tau2 chain left
// in the W accumulation launch; the V sampler - the next kernel - reads the draw.  hyp == nullptr: none.
struct LamSide { const double* lsum; int M; double shape; int exact; unsigned long long seed; double* hyp; double* pub; unsigned* flag; unsigned epoch; };
struct SweepSide { ScalarSide sc; LamSide lam; };
// The prior band of every local column (P_j = Delta' diag(1 / (lam2 Tau2_j)) Delta, half-bandwidth tf_order + 1: what
// prior_band_cols_kernel computes) as MORE workgroups of the w_solve launch of a full sweep, one per column: Tau2 was
// redrawn by the accumulation launch in front, lam2 by the lam2 workgroup of this very launch (LamSide: the band
// workgroups fetch their Tau2 row and stencil bounds, then wait for its flag) - and the V launch behind finds its band and
// the band's LDS image ready without a launch of their own.  pband == nullptr: none.
struct BandSide {
  const double* Tau2; int nD;
  const int* st_ptr; const int* st_row; const double* st_coef;
  int TD1, col0, ml;
  double* pband; double* pimg; int img_T, img_D1, img_PB;
  const double* hyp; double lam2;                 // lam2 when no workgroup of this launch draws it: hyp[HYP_LAM2] or the host's value
  const double* lam_pub; const unsigned* lam_flag; unsigned epoch;
  int* status;
};
__device__ inline void band_side(const BandSide& bs, int j, double* itau, int nthreads);

// MODE 0: X only (complete data)
// MODE 1: X, C and the outer products UU
// MODE 2: as 1, but the precision weight of output l is read from output srcmap[l] of the
//         same row and the linear statistic is rescaled by C_src/C_own - the reference's
//         stale cached weights (factor.py:349 and :394-401)
// waves per workgroup: 16 wherever the accumulators fit 128 VGPRs (1024-thread launch bound);
// the weighted modes of K >= 6 would spill there and run with 8 waves (256-VGPR budget).
// Where the weighted launches stand (round 3, C3, PMC passes in profiles/README.md): complete data issues 0.94 M VALU
// instructions per launch and takes 10.9 us (6.1 TB/s of the 67.1 MB statistic); byte counts (MODE 1) issue 4.28 M -
// 65 per wave and row of 128 cells: 2 x (20 FMA + 5 products) + conversions - which is 7 us of f64 issue spread over
// all SIMDs, beside 12.3 us of HBM time for the 75.5 MB at the complete-data rate; the launch takes 15.3 us.  A wave
// alternates "loads in flight" and "130 dependent-free FMAs", and at 121 VGPRs (the accumulator pairs alone are 80)
// three waves per SIMD overlap the two only partly.  Tried, A/B in one gpurun call each: software-pipelined rows
// (BTF_ACC_PF_WT=1: spills), 8 / 16 waves, four rows in flight (round 2: all equal or slower), one output per lane at
// K = 5 for twice the waves (BTF_ACC_OPL1_MINK=5: byte counts 14.5 / 14.9 us, but f64 weights 25.7 / 23.4 against
// 19.0 / 19.2 - not shipped).  MODE 2 adds two scalar gathers of the stale source's weight per lane and row.
// (complete data at K = 10: the stream itself needs 40 VGPRs, but the eigen side task compiled into the same kernel
//  spilled 7 under the 128-VGPR budget of 16 waves - 8 waves there, two workgroups per CU)
__host__ __device__ constexpr int acc_waves(int K, int MODE) {
  return MODE >= 1 ? (K >= 6 ? 8 : BTF_ACC_WAVES_WT) : (K >= 10 && ACC_WAVES > BTF_ACC_WAVES_K10 ? BTF_ACC_WAVES_K10 : ACC_WAVES);
}
// (A/B aid) occupancy hint of the K = 10 complete-data kernel: BTF_ACC_K10_EU waves per SIMD (0: none)
#ifndef BTF_ACC_K10_EU
#define BTF_ACC_K10_EU 0
#endif
// the lean instance (FUSE_LEAN) may be asked to fit two 16-wave workgroups per CU (8 waves per SIMD: at most 64 VGPRs):
// -DBTF_ACC_LEAN_EU=8 (A/B aid; 0: whatever the compiler picks)
#ifndef BTF_ACC_LEAN_EU
#define BTF_ACC_LEAN_EU 0
#endif
__host__ __device__ constexpr int acc_eu_min(int K, int MODE, int FUSE) { return (FUSE == 4 && BTF_ACC_LEAN_EU > 0) ? BTF_ACC_LEAN_EU : ((K >= 10 && MODE == 0 && BTF_ACC_K10_EU > 0) ? BTF_ACC_K10_EU : 1); }
__host__ __device__ constexpr int acc_eu_max(int K, int MODE, int FUSE) { return (FUSE == 4 && BTF_ACC_LEAN_EU > 0) ? BTF_ACC_LEAN_EU : ((K >= 10 && MODE == 0 && BTF_ACC_K10_EU > 0) ? BTF_ACC_K10_EU : 8); }
#if BTF_ACC_K10_EU > 0 || BTF_ACC_LEAN_EU > 0
#define BTF_ACC_EU_ATTR(K, MODE, WAVES, FUSE) __attribute__((amdgpu_waves_per_eu(acc_eu_min(K, MODE, FUSE), acc_eu_max(K, MODE, FUSE))))
#else
#define BTF_ACC_EU_ATTR(K, MODE, WAVES, FUSE)
#endif
// outputs per lane: two adjacent ones (one 16-byte load per row and lane) wherever the K + K(K+1)/2 accumulator pairs
// fit the register file; the weighted modes of K >= 9 (54 / 65 values: 216 / 260 VGPRs for the pairs alone) keep ONE
// output per lane - the waves of a workgroup pair up over the two halves of the 128-column tile - and do not spill
#ifndef BTF_ACC_OPL1_MINK
#define BTF_ACC_OPL1_MINK 9
#endif
__host__ __device__ constexpr int acc_opl(int K, int MODE) { return MODE >= 1 && K >= BTF_ACC_OPL1_MINK ? 1 : 2; }
template <int NW> __device__ void sweep_scalar_side(const ScalarSide& sc, double* red);
template <int NW> __device__ void sweep_lam_side(const LamSide& lm, double* red);

// FUSE: the latency kernel that follows the accumulation runs as the tail of this launch (btf_fused.h): FUSE_W - the
// workgroup that finishes a 128-row tile last sums the chunks and draws the rows (w_solve_kernel's work); FUSE_V - the
// columns of a 128-output tile are sampled where their sums are (v_spectral_kernel's work).  The extra kernel argument
// is empty for FUSE_NONE.
enum { FUSE_NONE = 0, FUSE_W = 1, FUSE_V = 2, FUSE_VDF = 3, FUSE_LEAN = 4 };      // FUSE_LEAN: FUSE_NONE without the gamma-drawing side tasks (Tau2 chain, scalars, lam2) - the plain W+V step's W launch      // FUSE_VDF: FUSE_V with the barrier-free (dataflow) tail only - an instance of its own
struct FuseNone {};
struct FuseW;
struct FuseV;
template <int FUSE> struct FuseSel { typedef FuseNone type; };
template <> struct FuseSel<FUSE_W> { typedef FuseW type; };
template <> struct FuseSel<FUSE_V> { typedef FuseV type; };
template <> struct FuseSel<FUSE_VDF> { typedef FuseV type; };
template <> struct FuseSel<FUSE_LEAN> { typedef FuseNone type; };
template <int K, int WAVES> __device__ __forceinline__ void w_fused_owner(const FuseW& fw, int tile, double* lds, long long* stamps);
template <int FUSE> __device__ __forceinline__ int fuse_owners(const typename FuseSel<FUSE>::type& fz);
struct VPre;
template <int K, int S, int WAVES, int NSUM>
__device__ __forceinline__ void v_fused_tail(const FuseV& fv, int tile, double* lds, const double (&sums)[NSUM], int nv_round, int tv, int tc, long long* stamps,
                                             const VPre& pre, const double* mailbox, bool band_early, bool unr3);
template <int K, int S> __device__ __forceinline__ void v_fused_prefetch(const FuseV& fv, int tile, double* mailbox, VPre& pre, bool band_early);
template <int K, int S, class PRE> __device__ __forceinline__ void v_fused_prefetch_loads(const FuseV& fv, int tile, PRE& pre, bool band_early);
constexpr int VF_PRE = 4;                  // row groups of the stream behind the fused V tail's prefetch
template <int K, int S> __device__ __forceinline__ bool v_fused_band_early(const FuseV& fv, int tile, double* lds, bool unr3);
#ifndef BTF_VF_BANDPRE
#define BTF_VF_BANDPRE 1
#endif
#if BTF_VF_BANDPRE
template <int K, int S> __device__ __forceinline__ bool v_fused_band_preload(const FuseV& fv, int tile, VPre& pre);
template <int K, int S> __device__ __forceinline__ void v_fused_band_store(const FuseV& fv, int tile, double* lds, const VPre& pre, bool unr3);
#endif
template <int FUSE> struct FusePre { struct type {}; };
constexpr int VF_MAILBOX_DOUBLES = 128;
// the dataflow form of the fused V tail (btf_fused.h, v_fused_df): no workgroup barrier behind the stream - the chain waves
// factor while the other waves are still streaming / reducing; LDS counters order the stages
struct VDfPre { double2 pv[8]; };
template <int FUSE> __device__ __forceinline__ int fuse_chain_waves(const typename FuseSel<FUSE>::type& fz) { return 0; }
// what a chain wave fetched BEHIND its last rows' loads (in order: back right after them, no wait of its own): the tagged
// eigenvalue granules of its lane's system, the device-resident nu2
struct DfEarly { double epub; bool have; };      // the eigen-system entry of this lane as the first worker wave prefetched it inside its stream
template <int FUSE> __device__ __forceinline__ const unsigned* fuse_eig_flag(const typename FuseSel<FUSE>::type& fz) { return nullptr; }
template <int FUSE> __device__ __forceinline__ const double* fuse_eig_pub(const typename FuseSel<FUSE>::type& fz) { return nullptr; }
template <int FUSE> __device__ __forceinline__ unsigned fuse_epoch(const typename FuseSel<FUSE>::type& fz) { return 0u; }
template <int K, int S> __device__ __forceinline__ void v_df_begin(const FuseV& fv, int tile, double* lds, VDfPre& pre);
template <int K, int S> __device__ __forceinline__ void v_df_band_load(const FuseV& fv, int tile, VDfPre& pre);
template <int K, int S> __device__ __forceinline__ void v_df_band_store(const FuseV& fv, int tile, double* lds, const VDfPre& pre);
template <int K, int S, int WAVES, int RG, int NVV>
__device__ __forceinline__ void v_fused_df(const FuseV& fv, int tile, double* lds, const double (&acc)[NVV][2], long long* stamps, const DfEarly& early);
template <int FUSE> __device__ __forceinline__ unsigned* fuse_tickets(const typename FuseSel<FUSE>::type& fz);
template <int FUSE> __device__ __forceinline__ int fuse_chunks(const typename FuseSel<FUSE>::type& fz);
// a tile's arrival ticket: call after drain_stores() + __syncthreads(); tells every thread whether this workgroup came
// last.  `word`: an LDS word free across the two barriers inside.  The counter is reset for the next launch.
__device__ __forceinline__ bool last_arriver_of(unsigned* cnt, unsigned expected, unsigned* word) {
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add((gu32_t*)cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t + 1u == expected;
    if (last) __hip_atomic_store((gu32_t*)cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *word = last ? 1u : 0u;
  }
  __syncthreads();
  const bool last = *word != 0u;
  __syncthreads();
  return last;
}
// LDS doubles of the FUSE_V instances: 144 KiB; 120 KiB beside the staged factor rows (ULDS) of the three-rows-in-flight form
__host__ __device__ constexpr int vf_red_doubles(bool unr3) { return unr3 ? 15360 : 18432; }
__host__ __device__ constexpr int w_tail_lds_doubles(int K);
__host__ __device__ constexpr int w_owner_lds_budget(int K);

// CT: storage type of the weights C: double (Binomial: the Polya-Gamma draws) or unsigned char (Gaussian data with
// missing replicates: C is the replicate count 0..R, 1 byte instead of 8 per cell: 9 instead of 16 B/cell)
// XT: storage type of the linear statistic X: double, or signed char for Binomial pseudo-data kappa = Y - N/2 with
// integer counts (2 kappa in -127..127 stored, 1 byte instead of 8 per cell: 9 instead of 16 B/cell with f64 weights)
// UNRV: rows in flight per wave; 0 = the build's default (2).  Long row ranges per workgroup (C5-sized slabs) stream
// 1-2 % faster with 3 (359 / 346 us against 361 / 353 us per launch at C5), short ones (C3: 32 rows per wave) slower.
#ifdef BTF_ACC_STAMPS     // diagnostic builds (scripts/acc_stamps.sh): wall-clock stamps (100 MHz) of every streaming workgroup
#define ACC_STAMP(i) do { if (threadIdx.x == 0 && b < 4096 && cm.stamps) cm.stamps[b * 8 + (i)] = wall_clock64(); } while (0)
#define ACC_SIDE_STAMP(i) do { if (threadIdx.x == 0 && cm.stamps) cm.stamps[(4096 + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#define ACC_TAIL_STAMPS (cm.stamps && b < 4096 ? cm.stamps + b * 8 : nullptr)
#define ACC_OWNER_STAMPS (cm.stamps ? cm.stamps + (size_t)(4096 + blockIdx.x) * 8 : nullptr)
#else
#define ACC_STAMP(i) do { } while (0)
#define ACC_SIDE_STAMP(i) do { } while (0)
#define ACC_TAIL_STAMPS nullptr
#define ACC_OWNER_STAMPS nullptr
#endif
// (fused tails: thread 0 of the workgroup stamps slot i of its record - diagnostic builds only, else st is nullptr)
#define TAIL_STAMP(st, i) do { if ((st) && threadIdx.x == 0) (st)[i] = wall_clock64(); } while (0)
template <int K, int MODE, int WAVES = acc_waves(K, MODE), typename CT = double, typename XT = double, int UNRV = 0,
          int OPL = acc_opl(K, MODE), int FUSE = FUSE_NONE>
__global__ __launch_bounds__(WAVES * WAVE) BTF_ACC_EU_ATTR(K, MODE, WAVES, FUSE) void accum_kernel(
    const XT* __restrict__ X, const CT* __restrict__ Cx, const double* __restrict__ U,
    const int* __restrict__ srcmap, double* __restrict__ part, int Rdim, int ld,
    int rows_per_block, EigSide side, EigSideCols sidec, TauSide tau, GramSide gram, ChunkMap cm, SweepSide sw,
    typename FuseSel<FUSE>::type fz) {
  static_assert(FUSE == FUSE_NONE || (MODE == 0 && OPL == 2 && WAVES == 16), "the fused tails follow the complete-data stream");
  constexpr bool PLAIN = FUSE == FUSE_NONE || FUSE == FUSE_LEAN;      // no tail: the partial sums go to HBM
  constexpr int KK = tri(K);
  constexpr int NV = MODE == 0 ? K : K + KK;
  constexpr int ACC_WAVES = WAVES;                       // shadows the namespace defaults inside this kernel
  // (the dataflow instance: its chain waves do not stream - the others keep BTF_DF_UNR rows in flight each)
  constexpr int ACC_UNR = FUSE == FUSE_VDF ? BTF_DF_UNR : (UNRV > 0 ? UNRV : (MODE >= 1 ? BTF_ACC_UNR_WT : (K >= 9 ? BTF_ACC_UNR_K9 : BTF_ACC_UNR)));
  // values per round of the cross-wave reduction: as many as the workgroup has 128-thread slots for and 96 KB of LDS hold
  // (16 waves: 6, 12 waves: 6, 8 waves: 4), never more than there are - K = 5 complete data (5 values) reduces in one round
  // instead of two, the weighted modes (20 values, 12 waves) in four instead of five; at least the old four (side-task scratch)
  constexpr int ACC_RGT = WAVES * WAVE / ACC_TILE, ACC_RGL = (96 * 1024) / (WAVES * ACC_TILE * 8);
  constexpr int ACC_RG0 = ACC_RGT < 4 ? ACC_RGT : 4;
  constexpr int ACC_RGW = ACC_RGT < ACC_RGL ? (ACC_RGT < NV ? ACC_RGT : NV) : (ACC_RGL < NV ? ACC_RGL : NV);
  constexpr int ACC_RG = ACC_RGW > ACC_RG0 ? ACC_RGW : ACC_RG0;
  // (FUSE_V: the same array also holds the sampler's layout of the tile's columns - btf_fused.h - and is sized for it)
  constexpr int RED_SLOT = ACC_RG * ACC_TILE;
  constexpr int RED_SLOTS = (FUSE == FUSE_V || FUSE == FUSE_VDF) ? (vf_red_doubles(UNRV == 3) + RED_SLOT - 1) / RED_SLOT : ACC_WAVES;
  static_assert(RED_SLOTS >= ACC_WAVES, "reduction scratch");
  __shared__ double red[RED_SLOTS][ACC_RG][ACC_TILE];
  static_assert(FUSE != FUSE_W || (w_tail_lds_doubles(K) <= WAVES * ACC_RG * ACC_TILE && w_owner_lds_budget(K) == WAVES * ACC_RG * ACC_TILE), "W owner scratch");
  // ULDS: the factor rows U[r][:] of the workgroup's row range come in through LDS (one coalesced copy per block of
  // ACC_UROWS rows, then a broadcast ds_read per row and wave) instead of scalar loads - for the long row ranges of
  // C5-sized slabs at K >= 8 (the three-rows-in-flight instance): 353 / 346 -> 342 / 337 us per launch at C5.  Short
  // ranges (C3-sized, 512 rows per workgroup) gain nothing (K = 8: 12.6 -> 13.1 us; K = 10: 18.7 us either way).
  constexpr bool ULDS = MODE == 0 && K >= BTF_ACC_ULDS_MINK && UNRV == 3;
  constexpr int ACC_UROWS_MAX = 512;
  __shared__ double ush[ULDS ? ACC_UROWS_MAX * K : 1];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // grid: one workgroup per (tile, chunk), linear; with a side task one more in front of them
  int b = blockIdx.x;
  // (one scalar test keeps the streaming workgroups clear of the side tasks' arguments)
  if (b < cm.nside) {
  if constexpr (FUSE == FUSE_W) {
    // the tiles' OWNER workgroups (btf_fused.h): they stream nothing - each prepares what the solve of its 128 rows needs
    // besides the chunk sums (normals, the shared Gram, the scalars), waits for the tile's streaming workgroups and
    // draws the rows
    const int nown = fuse_owners<FUSE>(fz);
    if (b < nown) {
      ACC_SIDE_STAMP(0);
      w_fused_owner<K, WAVES>(fz, b, &red[0][0][0], ACC_OWNER_STAMPS);
      ACC_SIDE_STAMP(3);
      return;
    }
    b -= nown;
  }
  if constexpr (FUSE != FUSE_LEAN) {
  if (sw.sc.hyp) {
    if (b == 0) { sweep_scalar_side<WAVES>(sw.sc, &red[0][0][0]); return; }
    b -= 1;
  }
  if (sw.lam.hyp) {
    if (b == 0) { sweep_lam_side<WAVES>(sw.lam, &red[0][0][0]); return; }
    b -= 1;
  }
  }
  if (side.out) {
    // side task (spectral V sampler): the first workgroup is dispatched first; one wave of it solves the K x K
    // eigenproblem of the Gram beside the stream, its other waves leave at once
    constexpr int TPW = eig_side_tpw(WAVES);
    const int nside = 1 + eig_side_groups(sidec.ncols, WAVES);
    if (b < nside) {
#ifndef BTF_EIG_NOP      // (timing aid: BTF_EIG_NOP builds skip the side task - results are wrong, only the clock is read)
      // the whole workgroup sums the Gram partials (one round of loads, whatever their number); then workgroup 0's
      // first wave solves the shared Gram, and in the others wave w takes curve column (b-1) TPW + w: its own Gram,
      // its own (warm-started) eigen-system
      // LDS: [gsum 64][gown TPW x 64][sc: the eigen-solvers' scratch; before they start, the Gram reduction's]
      double* gsum = &red[0][0][0];                              // [64]
      double* gown = gsum + 64;                                  // [TPW][64]
      double* sc = gown + TPW * 64;                              // [TPW][EIG_LDS_DOUBLES]
      double* rsc = sc;                                          // reduce_gram: <= blockDim doubles; gram_mfma_block: 256 per wave
      static_assert(64 + TPW * 64 + TPW * EIG_LDS_DOUBLES <= ACC_WAVES * ACC_RG * ACC_TILE, "side task scratch");
      static_assert(64 + TPW * 64 + ACC_WAVES * 256 <= ACC_WAVES * ACC_RG * ACC_TILE, "side task Gram scratch");
      static_assert(tri(EIG_MAXK) <= 64, "side task Gram slots");
      // (the unrolled eigen-solver needs registers: two matrix elements per lane from K = 9 on, which the 128-VGPR
      //  budget of a 16-wave workgroup does not have beside the stream's - there K stays a run-time value)
      constexpr int EIG_KC = (K <= 8 || WAVES <= 8) ? K : 0;
      ACC_SIDE_STAMP(0);
      const int t = (b - 1) * TPW + wave;
      const bool task = b > 0 && wave < TPW && t < sidec.ncols;
      const int col = task ? sidec.cols[t] : 0;
      if (task) curve_column_sum_wave<K>(sidec.cv, sidec.W, col, gown + wave * 64);    // (its loads fly with reduce_gram's)
      // (workgroup 0's first wave: the previous eigen-system, the warm start of the solve - fetched with the Gram partials)
      EigWarm eig_pre{0.0, 0.0, 0.0, false};
      // (not where the solver runs with a run-time K - nembeds 9, 10 on 16 waves: no register to park it in)
      if constexpr (EIG_KC > 0) { if (b == 0 && wave == 0) eig_pre = eig_warm_fetch<EIG_KC>(side.out, K); }
      if (side.Usrc) {         // (sharded runs: W has just been all-gathered, nobody summed its Gram)
        gram_mfma_block<K, WAVES>(side.Usrc, side.nrows, 0, 1, rsc, gsum);
        __syncthreads();
      } else {
        reduce_gram(side.gpart, side.ngp, KK, 1.0, rsc, gsum);
      }
      if (b == 0) {
        ACC_SIDE_STAMP(1);
        if (wave == 0) {
          // (the run-time-K form of the solver - nembeds 9, 10 on 16 waves - has no register to spare for the prefetched warm
          //  start or the granules: nobody reads granules there, the dataflow tails stop at nembeds 6)
          if constexpr (EIG_KC > 0) gram_eig_wave<EIG_KC, true>(gsum, 1, K, side.out, sc, true, side.pub, eig_pre, side.flag ? side.gran : nullptr, side.epoch);
          else gram_eig_wave<EIG_KC>(gsum, 1, K, side.out, sc, true, side.pub);
          if (side.flag) {                     // the tails of this launch wait for it (btf_fused.h): one storing wave
            drain_stores();
            if (lane == 0) publish_epoch(side.flag, side.epoch);
          }
          ACC_SIDE_STAMP(3);
        }
      } else if (task) {
        if (lane < KK) gown[wave * 64 + lane] = fma(-sidec.inv_R, gown[wave * 64 + lane], gsum[lane]);
        wave_lds_sync();
        gram_eig_wave<EIG_KC>(gown + wave * 64, 1, K, sidec.eig_cols + (size_t)col * (K + K * K + 8), sc + wave * EIG_LDS_DOUBLES);
      }
#endif
      return;
    }
    b -= nside;
  }
  if constexpr (FUSE != FUSE_LEAN)
  if (tau.Tau2) {
    constexpr int CPW = TAU_SIDE_CPW;                       // columns per side workgroup
    const int ntw = (tau.M + CPW - 1) / CPW;
    if (b < ntw) {
      const int slot = threadIdx.x >> 8;
      if (slot < CPW) tau2_column(tau, K, b * CPW + slot, threadIdx.x & 255, &red[0][0][0] + 4 * slot);
      else __syncthreads();                                 // (12-wave workgroups have no such threads; kept for safety)
      return;
    }
    b -= ntw;
  }
  if (gram.gpart) {
    if (b < gram.nblocks) {
      // rows b, b + nblocks*threads, ... (strided like gram_kernel), K(K+1)/2 sums per thread, fixed-order reduction
      // this block's share of the rows on the matrix cores (gram_mfma_block), fixed-order reduction over the waves
      ACC_SIDE_STAMP(0);
      static_assert(ACC_WAVES * 256 <= ACC_WAVES * ACC_RG * ACC_TILE, "Gram side scratch");
      gram_mfma_block<K, WAVES>(gram.U, gram.Rdim, b, gram.nblocks, &red[0][0][0], gram.gpart + (size_t)b * KK, gram.cnt != nullptr);
      if (gram.cnt) {
        drain_stores();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add((gu32_t*)gram.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      ACC_SIDE_STAMP(3);
      return;
    }
    b -= gram.nblocks;
  }
  if (gram.sum_src) {
    if (b == 0) {
      // thread (l, q) of the first sum_threads adds the partials l, l + lanes, ... in batches of eight (reduce_gram's order for
      // a sum_threads-wide workgroup), then KK threads add the lanes' sums in order
      double* scratch = &red[0][0][0];
      const int lanes = gram.sum_threads / KK > 32 ? 32 : (gram.sum_threads / KK < 1 ? 1 : gram.sum_threads / KK);
      const int l = threadIdx.x / KK, q = threadIdx.x - l * KK;
      if ((int)threadIdx.x < gram.sum_threads && l < lanes) {
        double sg = 0.0;
        for (int b0 = l; b0 < gram.sum_n; b0 += 8 * lanes) {
          double x[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) x[u] = b0 + u * lanes < gram.sum_n ? gram.sum_src[(size_t)(b0 + u * lanes) * KK + q] : 0.0;
#pragma unroll
          for (int u = 0; u < 8; ++u) sg += x[u];
        }
        scratch[l * KK + q] = sg;
      }
      __syncthreads();
      if ((int)threadIdx.x < KK) {
        double t = 0.0;
        for (int bb = 0; bb < lanes; ++bb) t += scratch[bb * KK + threadIdx.x];
        gram.sum_out[threadIdx.x] = t;
      }
      return;
    }
    b -= 1;
  }
  return;                                                    // (not reached: cm.nside counts exactly the side workgroups)
  }
  b -= cm.nside;
  ACC_STAMP(0);
  const int ntiles = ld / ACC_TILE;
  const int lchunk = b / ntiles, tile = b - lchunk * ntiles;
  const int chunk = cm.slot_base + lchunk;                  // slot of the partials
  // OPL == 2: every wave covers the tile's 128 columns (two per lane) and takes every WAVES-th row.
  // OPL == 1: wave w covers the half (w & 1) of the tile, one column per lane, and takes every (WAVES/2)-th row.
  static_assert(OPL == 2 || (OPL == 1 && WAVES % 2 == 0), "outputs per lane");
  constexpr int NWR = OPL == 2 ? WAVES : WAVES / 2;             // waves along the rows
  const int half = OPL == 2 ? 0 : (wave & 1);
  const int wv = OPL == 2 ? wave : (wave >> 1);
  const size_t col = (size_t)tile * ACC_TILE + (OPL == 2 ? 2 * lane : 64 * half + lane);
  int r0 = cm.row_base + lchunk * rows_per_block;
  if (r0 >= cm.skip_at) r0 += cm.skip_rows;
  const int r1 = min(r0 + rows_per_block, cm.row_end);

  // (fused V launch: the prior band of the tile's columns while the memory system is still idle - btf_fused.h)
  // (dataflow tail: counters zeroed behind one barrier at kernel start; the chain waves fetch the whole band of their column)
  constexpr bool dataflow = FUSE == FUSE_VDF;
  static_assert(!dataflow || (UNRV != 3 && NV <= ACC_RG && OPL == 2), "the dataflow tail: one reduction round, two rows in flight");
  bool df_band_done = false;
  DfEarly df_early{0.0, false};
  VDfPre dfpre;
  if constexpr (dataflow) v_df_begin<K, 3>(fz, tile, &red[0][0][0], dfpre);
  // (dataflow tail: the columns' chain waves - waves 0 .. ncw-1 - do NOT stream: they wait for the eigenvalues and factor
  //  while the other waves, which share the rows among themselves, stream; btf_fused.h)
  int df_ncw = 0;
  if constexpr (dataflow) df_ncw = fuse_chain_waves<FUSE>(fz);
  int nwr_rt = NWR;                                             // waves along the rows of the span being streamed (dataflow only)
  bool band_early = false;
  if constexpr (FUSE == FUSE_V) band_early = v_fused_band_early<K, 3>(fz, tile, &red[0][0][0], UNRV == 3);
  typename FusePre<FUSE>::type vpre;
#if BTF_VF_BANDPRE
  if constexpr (FUSE == FUSE_V) band_early = v_fused_band_preload<K, 3>(fz, tile, vpre);
#endif
  double acc[NV][OPL];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int o = 0; o < OPL; ++o) acc[v][o] = 0.0;
#ifdef BTF_NO_NT
  constexpr bool nt = false;                              // (A/B builds: the loop without the second form of the load)
#else
  const bool nt = __builtin_amdgcn_readfirstlane(cm.nt) != 0;      // (uniform: the loop is compiled for both forms of the load)
#endif
  int s0 = 0, s1 = 0;
  // MODE 2 (the stale cached weights of compat="reference"): every output that shares a source shares the source's Gram -
  // factor.py:349-357 reuses Xt / Lt for the rows >= K, :394-400 Q_likelihood until the NaN pattern changes - so the
  // K(K+1)/2 outer-product sums are accumulated only where an output IS a source (its own weights); a tile without any
  // (all but the first at C4: every wave of a workgroup covers the tile's 128 outputs, so this is workgroup-uniform) runs
  // the K-sum stream with the rescaled statistic, reduces one round instead of four and writes a quarter of the partials.
  // The consumers read a dependent output's Gram at its source's column (WSolveArgs.gsrc / VBandArgs.gsrc): the same
  // sums in the same order, accumulated once.
  bool gram_on = true;
  if constexpr (MODE == 2) {
    s0 = srcmap[col];
    if constexpr (OPL == 2) s1 = srcmap[col + 1];
    const bool own = s0 == (int)col || (OPL == 2 && s1 == (int)col + 1);
    gram_on = __builtin_amdgcn_readfirstlane(__any(own) ? 1 : 0) != 0;
  }
  // ... and how the source's weight of a row is fetched (wave-uniform choice, made once): 1 - every output of the wave has
  // the SAME source (the W half-sweep: row K-1 for all later rows): one scalar load per row; 2 - the lane's two outputs have
  // neighbouring sources (the V half-sweep: (j, t), (j, t+1) -> (src j, t), (src j, t+1)): one 16-byte load like the lane's
  // own weights; 0 - two gathers per lane and row (what every case used to pay: the vector-memory issue rate, not the bytes,
  // bounded this mode - 19.6 us for 75.5 MB at C4)
  int cs_form = 0, cs_src = 0;
  if constexpr (MODE == 2 && OPL == 2) {
    cs_src = __builtin_amdgcn_readfirstlane(s0);
    if (__all(s0 == cs_src && s1 == cs_src)) cs_form = 1;
    else if (__all(s1 == s0 + 1 && (s0 & 1) == 0)) cs_form = 2;
    cs_form = __builtin_amdgcn_readfirstlane(cs_form);
  }

  int ublk0 = r0;                                             // first row of the block of U staged in `ush` (ULDS)
  // (uk: the wave-uniform factor row - scalar loads issued WITH the vector loads, one wait for all of them)
  // FULL: every row of the group exists - no guards, one basic block (the guarded form is the tail's)
  // (OPL == 1: only the .x halves of the pairs are used)
  struct Rows { double2 x[ACC_UNR]; double2 c[MODE >= 1 ? ACC_UNR : 1]; double2 cs[MODE == 2 ? ACC_UNR : 1]; double uk[ACC_UNR][K]; };
  auto load_rows = [&](int rb, Rows& R, auto full, auto ntc) {
    constexpr bool FULL = decltype(full)::value;
    constexpr bool NT = decltype(ntc)::value;       // (the statistic by non-temporal loads: see ChunkMap.nt)
#pragma unroll
    for (int u = 0; u < ACC_UNR; ++u) {
      const int r = rb + u * (dataflow ? nwr_rt : NWR);  // wave-uniform
      if constexpr (ULDS) {
        const double* up = ush + (size_t)((FULL ? r : min(r, r1 - 1)) - ublk0) * K;         // wave-uniform: broadcast reads
#pragma unroll
        for (int k = 0; k < K; ++k) R.uk[u][k] = up[k];
      } else {
        const double* __restrict__ up = U + (size_t)(FULL ? r : min(r, r1 - 1)) * K;      // clamp: x/c are zero beyond r1
#pragma unroll
        for (int k = 0; k < K; ++k) R.uk[u][k] = up[k];
      }
      if (FULL || r < r1) {
        if constexpr (OPL == 1) {
          if constexpr (sizeof(XT) == 1) R.x[u] = make_double2(0.5 * (double)X[(size_t)r * ld + col], 0.0);
          else R.x[u] = make_double2((double)X[(size_t)r * ld + col], 0.0);
          if constexpr (MODE >= 1) R.c[u] = make_double2((double)Cx[(size_t)r * ld + col], 0.0);
          if constexpr (MODE == 2) R.cs[u] = make_double2((double)Cx[(size_t)r * ld + s0], 0.0);
        } else {
          if constexpr (sizeof(XT) == 1) {
            const char2 xx = *reinterpret_cast<const char2*>(X + (size_t)r * ld + col);
            R.x[u] = make_double2(0.5 * (double)xx.x, 0.5 * (double)xx.y);
          } else {
            if constexpr (NT) R.x[u] = stream_load2(reinterpret_cast<const double*>(X) + (size_t)r * ld + col);
            else R.x[u] = *reinterpret_cast<const double2*>(X + (size_t)r * ld + col);
          }
          if constexpr (MODE >= 1) {
            if constexpr (sizeof(CT) == 1) {
              const uchar2 cc = *reinterpret_cast<const uchar2*>(Cx + (size_t)r * ld + col);
              R.c[u] = make_double2((double)cc.x, (double)cc.y);
            } else {
              if constexpr (NT) R.c[u] = stream_load2(reinterpret_cast<const double*>(Cx) + (size_t)r * ld + col);
              else R.c[u] = *reinterpret_cast<const double2*>(Cx + (size_t)r * ld + col);
            }
          }
          if constexpr (MODE == 2) {
            if (cs_form == 1) { const double w1 = (double)Cx[(size_t)r * ld + cs_src]; R.cs[u] = make_double2(w1, w1); }      // (uniform address: a scalar load)
            else if (cs_form == 2 && sizeof(CT) == 8) R.cs[u] = *reinterpret_cast<const double2*>(reinterpret_cast<const double*>(Cx) + (size_t)r * ld + s0);
            else R.cs[u] = make_double2((double)Cx[(size_t)r * ld + s0], (double)Cx[(size_t)r * ld + s1]);
          }
        }
      } else {
        R.x[u] = make_double2(0.0, 0.0);
        if constexpr (MODE >= 1) R.c[u] = make_double2(0.0, 0.0);
        if constexpr (MODE == 2) R.cs[u] = make_double2(0.0, 0.0);
      }
    }
  };
  auto compute = [&](int rb, Rows& R) {
#pragma unroll
    for (int u = 0; u < ACC_UNR; ++u) {
      if constexpr (MODE == 2) {           // stale cached weights: the source output's weight, the statistic rescaled
        // (reciprocal by v_rcp_f64 + two Newton steps: 0.5 ulp, a fifth of the instructions of an IEEE division)
        auto rcp2 = [](double d) { double r = __builtin_amdgcn_rcp(d); r = fma(fma(-d, r, 1.0), r, r); return fma(fma(-d, r, 1.0), r, r); };
        if (s0 != (int)col) R.x[u].x = R.c[u].x != 0.0 ? R.x[u].x * R.cs[u].x * rcp2(R.c[u].x) : 0.0;
        if constexpr (OPL == 2) {
          if (s1 != (int)col + 1) R.x[u].y = R.c[u].y != 0.0 ? R.x[u].y * R.cs[u].y * rcp2(R.c[u].y) : 0.0;
        }
        R.c[u] = R.cs[u];
      }
      const double* up = R.uk[u];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const double uk = up[k];
        acc[k][0] = fma(R.x[u].x, uk, acc[k][0]);
        if constexpr (OPL == 2) acc[k][1] = fma(R.x[u].y, uk, acc[k][1]);
      }
      if (MODE >= 1 && (MODE != 2 || gram_on)) {
        // outer products on the fly: (c u_p) u_q, K more multiplies per output instead of KK more scalar operands
        // per row - the scalar loads of a precomputed table (KK doubles per row, more than the SGPR file holds for
        // the rows in flight) were issued just in time and stalled every wave four times per row
#pragma unroll
        for (int p = 0; p < K; ++p) {
          const double cx = R.c[u].x * up[p], cy = OPL == 2 ? R.c[u].y * up[p] : 0.0;
#pragma unroll
          for (int q = 0; q <= p; ++q) {
            acc[K + lidx(p, q)][0] = fma(cx, up[q], acc[K + lidx(p, q)][0]);
            if constexpr (OPL == 2) acc[K + lidx(p, q)][1] = fma(cy, up[q], acc[K + lidx(p, q)][1]);
          }
        }
      }
    }
  };
  constexpr int STEP = NWR * ACC_UNR;
  constexpr bool PIPELINED = MODE >= 1 && BTF_ACC_PF_WT;      // (complete data: 5 FMAs per load, nothing to hide)
  const int full_end = r1 - (ACC_UNR - 1) * NWR;              // groups starting below it have all their rows
  int rb = r0 + wv;
  // (the whole stream once per form of the load: no branch inside the loops)
  auto run_stream = [&](auto ntc) {
  if constexpr (ULDS) {
    // blocks of ACC_UROWS rows: copy the block's factor rows (contiguous: rows x K doubles, 16-byte aligned since the
    // block starts at a multiple of 64 rows), then the waves stream its rows as below; STEP divides ACC_UROWS, so a
    // wave's row sequence simply continues from block to block
    constexpr int ACC_UROWS = (ACC_UROWS_MAX / STEP) * STEP;   // whole groups of rows; even (STEP is)
    for (ublk0 = r0; ublk0 < r1; ublk0 += ACC_UROWS) {
      const int bend = min(ublk0 + ACC_UROWS, r1);
      if (ublk0 > r0) __syncthreads();                        // everybody is done with the previous block
      {
        const double2* __restrict__ src = reinterpret_cast<const double2*>(U + (size_t)ublk0 * K);
        double2* dst = reinterpret_cast<double2*>(ush);
        const int n2 = (bend - ublk0) * K / 2;                // (an odd last double - K odd, odd row count - goes separately)
        for (int i = threadIdx.x; i < n2; i += WAVES * WAVE) dst[i] = src[i];
        if (((bend - ublk0) * K & 1) && threadIdx.x == 0) ush[(bend - ublk0) * K - 1] = U[(size_t)bend * K - 1];
      }
      __syncthreads();
      if (ublk0 == r0) ACC_STAMP(1);
      const int bfull = min(full_end, bend - (ACC_UNR - 1) * NWR);
      for (; rb < bfull; rb += STEP) {
        Rows A;
        load_rows(rb, A, std::true_type{}, ntc);
        compute(rb, A);
      }
      if (rb < bend) {                                         // the block's (= the range's: blocks before the last are whole) tail
        Rows A;
        load_rows(rb, A, std::false_type{}, ntc);
        compute(rb, A);
        rb += STEP;
      }
    }
  } else if constexpr (PIPELINED) {
    Rows A, B;
    bool more = rb < full_end;
    if (more) load_rows(rb, A, std::true_type{}, ntc);
    while (more) {
      const bool nextB = rb + STEP < full_end;
      if (nextB) load_rows(rb + STEP, B, std::true_type{}, ntc);
      compute(rb, A);
      rb += STEP;
      if (!nextB) break;
      more = rb + STEP < full_end;
      if (more) load_rows(rb + STEP, A, std::true_type{}, ntc);
      compute(rb, B);
      rb += STEP;
    }
  } else {
    if constexpr (FUSE == FUSE_V) {
      // fused V launch: the tail's own global loads (stencil, Tau2) go out VF_PRE row groups before the stream ends - under
      // load they take ~2 us, and behind the stream they would sit in front of the reduction's first barrier
      const int pre_end = full_end - VF_PRE * STEP;
      for (; rb < pre_end; rb += STEP) {
        Rows A;
        load_rows(rb, A, std::true_type{}, ntc);
        compute(rb, A);
      }
      v_fused_prefetch_loads<K, 3>(fz, tile, vpre, band_early);
    }
    if constexpr (dataflow) {
      // dataflow tail (btf_fused.h): the first BTF_DF_SHARE percent of the rows are dealt over all the waves - the chain
      // waves have nothing else to do until the eigenvalues arrive - the rest over the WAVES - ncw streaming waves.  The band
      // image the columns' copying waves asked for at kernel start came back with the first rows' loads (in order): into
      // the LDS behind the first row group, nothing parked through the stream
      const int g1 = NWR * ACC_UNR;
      const int rsplit = r0 + ((r1 - r0) * BTF_DF_SHARE / 100) / g1 * g1;      // whole row groups of all the waves
      nwr_rt = NWR;
      for (rb = r0 + wv; rb < rsplit; rb += g1) {
        Rows A;
        load_rows(rb, A, std::true_type{}, ntc);
        compute(rb, A);
        if (!df_band_done) { v_df_band_store<K, 3>(fz, tile, &red[0][0][0], dfpre); df_band_done = true; }
      }
      if (wave >= df_ncw) {
        nwr_rt = NWR - df_ncw;
        const int step = nwr_rt * ACC_UNR, fend = r1 - (ACC_UNR - 1) * nwr_rt;
        // (the first of them also prefetches the eigen-system the workers' rotations need - btf_fused.h: from 5/8 of its
        //  rows on it looks at the side workgroup's flag behind a row group's loads, and once the word has come back as
        //  this launch's it asks for the K + K K doubles, one per lane, behind the next group's: no round trip at stream end)
        const bool eigw = BTF_DF_PREFETCH && wave == df_ncw;
        const int rprobe = r0 + (r1 - r0) * 5 / 8;
        unsigned eflag = 0u;
        int estate = 0;
        for (rb = rsplit + wv - df_ncw; rb < fend; rb += step) {
          Rows A;
          load_rows(rb, A, std::true_type{}, ntc);
          if (eigw && estate < 2 && rb >= rprobe) {
            if (estate == 1 && eflag == fuse_epoch<FUSE>(fz)) {
              if (lane < K + K * K) df_early.epub = load_sc1(fuse_eig_pub<FUSE>(fz) + lane);
              df_early.have = true;
              estate = 2;
            } else {
              eflag = __hip_atomic_load(fuse_eig_flag<FUSE>(fz), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              estate = 1;
            }
          }
          compute(rb, A);
          if (!df_band_done) { v_df_band_store<K, 3>(fz, tile, &red[0][0][0], dfpre); df_band_done = true; }
        }
        if (rb < r1) {
          Rows A;
          load_rows(rb, A, std::false_type{}, ntc);
          compute(rb, A);
        }
      }
      rb = r1;                                                // (the generic loops below: nothing left)
    }
    for (; rb < full_end; rb += STEP) {
      Rows A;
      load_rows(rb, A, std::true_type{}, ntc);
      compute(rb, A);
    }
  }
  if constexpr (!ULDS) {
    if (rb < r1) {
      Rows A;
      load_rows(rb, A, std::false_type{}, ntc);
      compute(rb, A);
    }
  }

  };
  if (nt) run_stream(std::true_type{}); else run_stream(std::false_type{});
  ACC_STAMP(2);
  if constexpr (dataflow) {
    if (!df_band_done) {                                    // (a row range too short for the in-stream store: fetched again, now)
      v_df_band_load<K, 3>(fz, tile, dfpre);
      v_df_band_store<K, 3>(fz, tile, &red[0][0][0], dfpre);
    }
    v_fused_df<K, 3, WAVES, ACC_RG, NV>(fz, tile, &red[0][0][0], acc, ACC_TAIL_STAMPS, df_early);
    return;
  }
  // (fused V launch: the tail's global loads and its wait for the side workgroups go out here, under the wave skew and
  //  the reduction - btf_fused.h)
  if constexpr (FUSE == FUSE_V) {
    if constexpr (ULDS) v_fused_prefetch_loads<K, 3>(fz, tile, vpre, band_early);      // (the staged-factor form of long row ranges: here)
#if BTF_VF_BANDPRE
    if (band_early) v_fused_band_store<K, 3>(fz, tile, &red[0][0][0], vpre, UNRV == 3);
#endif
    v_fused_prefetch<K, 3>(fz, tile, &red[0][0][0] + RED_SLOTS * RED_SLOT - VF_MAILBOX_DOUBLES, vpre, band_early);
  }
  // cross-wave reduction through LDS, ACC_RG values per round, fixed order
  const int tv = threadIdx.x >> 7;   // value slot 0..3
  const int tc = threadIdx.x & 127;  // column inside the tile
  constexpr int NFS = PLAIN ? 1 : (NV + ACC_RG - 1) / ACC_RG;
  double fsum[NFS];                  // (fused tails: this thread's sums, one per round)
#pragma unroll
  for (int r = 0; r < NFS; ++r) fsum[r] = 0.0;
#pragma unroll
  for (int g = 0; g < NV; g += ACC_RG) {
    // (MODE 2 without a source output in the tile: the Gram slots were never accumulated and nobody reads them here - rounds
    //  that hold nothing but Gram values are skipped, uniformly)
    if (MODE == 2 && !gram_on && g >= K) break;
#pragma unroll
    for (int v = 0; v < ACC_RG; ++v) {
      if (g + v < NV) {
        if constexpr (OPL == 2) *reinterpret_cast<double2*>(&red[wave][v][2 * lane]) = make_double2(acc[g + v][0], acc[g + v][1]);
        else red[wv][v][64 * half + lane] = acc[g + v][0];
      }
    }
    __syncthreads();
    if constexpr (!ULDS) { if (g == 0) ACC_STAMP(1); }         // (diagnostic builds: every wave has finished its rows)
    if (tv < ACC_RG && g + tv < NV && !(MODE == 2 && !gram_on && g + tv >= K)) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < NWR; ++w) s += red[w][tv][tc];
      if constexpr (PLAIN) {
#ifdef BTF_PART_SC1      // (A/B aid: the partial sums written through - does the next kernel's cold read of them get shorter?)
        store_sc1(&part[((size_t)chunk * NV + (g + tv)) * ld + (size_t)tile * ACC_TILE + tc], s);
#else
        part[((size_t)chunk * NV + (g + tv)) * ld + (size_t)tile * ACC_TILE + tc] = s;
#endif
      } else {
        // the tail of another workgroup of this launch reads them: write-through (one chunk, FUSE_V: they stay here)
        if (FUSE == FUSE_W || fuse_tickets<FUSE>(fz)) store_sc1(&part[((size_t)chunk * NV + (g + tv)) * ld + (size_t)tile * ACC_TILE + tc], s);
        if constexpr (FUSE == FUSE_V) fsum[g / ACC_RG] = s;
      }
    }
    __syncthreads();
  }
  ACC_STAMP(3);
  if constexpr (FUSE == FUSE_W) {
    // every wave's chunk sums are on their way (write-through): drained, then ONE add to the tile's counter - its owner
    // workgroup is waiting for the tile's last one
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add((gu32_t*)(fuse_tickets<FUSE>(fz) + tile), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (FUSE == FUSE_V) {
    double* lds = &red[0][0][0];
    unsigned* tickets = fuse_tickets<FUSE>(fz);
    if (tickets) {
      // every wave's chunk sums are on their way: drained, then ONE ticket for the workgroup; whoever draws the tile's
      // last ticket goes on alone
      drain_stores();
      __syncthreads();
      if (!last_arriver_of(tickets + tile, (unsigned)fuse_chunks<FUSE>(fz), reinterpret_cast<unsigned*>(lds))) return;
    }
    v_fused_tail<K, 3, WAVES, NFS>(fz, tile, lds, fsum, ACC_RG, tv, tc, ACC_TAIL_STAMPS, vpre, lds + RED_SLOTS * RED_SLOT - VF_MAILBOX_DOUBLES, band_early, UNRV == 3);
  }
}

// ============================================================================
// small helpers: Gram of the fixed factor, per-row outer products
// ============================================================================
constexpr int GRAM_BLOCKS = 64;          // at most (gpart is sized for it); gram_blocks() picks the launch
__host__ inline int gram_blocks(int Rdim) { int b = Rdim / 1024; return b < 16 ? 16 : (b > GRAM_BLOCKS ? GRAM_BLOCKS : b); }
constexpr int GRAM_THREADS = 256;

template <int K>
__global__ __launch_bounds__(GRAM_THREADS) void gram_kernel(const double* __restrict__ U, int Rdim,
                                                            double* __restrict__ gpart) {
  constexpr int KK = tri(K);
  __shared__ double red[GRAM_THREADS / WAVE][KK];
  double acc[KK];
#pragma unroll
  for (int q = 0; q < KK; ++q) acc[q] = 0.0;
  for (int r = blockIdx.x * GRAM_THREADS + threadIdx.x; r < Rdim; r += gridDim.x * GRAM_THREADS) {
    double u[K];
#pragma unroll
    for (int k = 0; k < K; ++k) u[k] = U[(size_t)r * K + k];
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) acc[lidx(a, b)] = fma(u[a], u[b], acc[lidx(a, b)]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < KK; ++q) {
    double s = wave_sum(acc[q]);
    if (lane == 0) red[wave][q] = s;
  }
  __syncthreads();
  if (threadIdx.x < KK) {
    double s = 0.0;
    for (int w = 0; w < GRAM_THREADS / WAVE; ++w) s += red[w][threadIdx.x];
    gpart[blockIdx.x * KK + threadIdx.x] = s;
  }
}

// Sum the ngp partial Grams into G[KK] (LDS), fixed order, one or two global-load latencies: thread
// (l = tid / KK, q = tid % KK), l < lanes, adds the partials b = l, l + lanes, ... of entry q (eight loads in
// flight), then KK threads add the `lanes` sums.  `scratch`: lanes*KK <= blockDim.x doubles of LDS, whatever ngp
// is - the partials are never staged.  Ends with a barrier.
__device__ __forceinline__ int gram_lanes(int KK) {
  const int l = (int)blockDim.x / KK;
  return l > 32 ? 32 : (l < 1 ? 1 : l);
}
__device__ __forceinline__ void reduce_gram_tail(double s, int KK, double scale, double* scratch, double* G) {
  const int lanes = gram_lanes(KK);
  const int l = threadIdx.x / KK, q = threadIdx.x - l * KK;
  if (l < lanes) scratch[l * KK + q] = s;
  __syncthreads();
  if ((int)threadIdx.x < KK) {
    double t = 0.0;
    for (int b = 0; b < lanes; ++b) t += scratch[b * KK + threadIdx.x];
    G[threadIdx.x] = t * scale;
  }
  __syncthreads();
}
__device__ inline void reduce_gram(const double* __restrict__ gpart, int ngp, int KK, double scale,
                                   double* scratch, double* G) {
  const int lanes = gram_lanes(KK);
  const int l = threadIdx.x / KK, q = threadIdx.x - l * KK;
  double s = 0.0;
  if (l < lanes) {
    for (int b0 = l; b0 < ngp; b0 += 8 * lanes) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = b0 + u * lanes < ngp ? gpart[(size_t)(b0 + u * lanes) * KK + q] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += x[u];
    }
  }
  reduce_gram_tail(s, KK, scale, scratch, G);
}

// reduce_gram in two halves for callers that want the global loads in flight early (gram_early_ok: at most eight
// partials per thread): reduce_gram_fetch issues them, reduce_gram_finish adds them up later.
__device__ __forceinline__ bool gram_early_ok(int ngp, int KK) { return ngp <= 8 * gram_lanes(KK); }
__device__ __forceinline__ void reduce_gram_fetch(const double* __restrict__ gpart, int ngp, int KK, double (&x)[8]) {
  const int lanes = gram_lanes(KK);
  const int l = threadIdx.x / KK, q = threadIdx.x - l * KK;
#pragma unroll
  for (int u = 0; u < 8; ++u) x[u] = (l < lanes && l + u * lanes < ngp) ? gpart[(size_t)(l + u * lanes) * KK + q] : 0.0;
}
__device__ inline void reduce_gram_finish(const double (&x)[8], int ngp, int KK, double scale, double* scratch, double* G) {
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < 8; ++u) s += x[u];
  reduce_gram_tail(s, KK, scale, scratch, G);
}

// ============================================================================
// Curve-structured replicate counts (Gaussian data whose counts c_ij do not vary along the depth axis: whole
// curves held out or thinned, the pattern of the reference's examples - Y[:3, :3] = NaN in
// examples/gaussian_tensor_filtering.py:16-18).  The weighted systems of factor.py:343-346 / :388-391 then are the
// complete-data ones minus a few rank-one / per-column terms,
//     row i:     sum_(j,t) c_ij v v' = R V'V - sum_{j in D(i)} (R - c_ij) V_j'V_j          (V_j'V_j: K x K per column)
//     column j:  sum_i c_ij w w'     = R W'W - sum_{i in D(j)} (R - c_ij) w_i w_i'         (the same at every depth)
// with D(.) the deficient partners (CSR lists, built once at upload): the streaming accumulation stays the
// complete-data one (K values per cell, no counts read), and a column keeps the Kronecker structure the spectral
// sampler needs - with its own Gram.
// ============================================================================
// per-column Gram of V: out[j][KK] = sum_t v_jt v_jt'   (one wave per column; the samplers emit the same blocks
// for a column they have just drawn)
template <int K>
__global__ __launch_bounds__(WAVE) void colgram_kernel(const double* __restrict__ V, int T, int ncols, double* __restrict__ out) {
  constexpr int KK = tri(K);
  const int j = blockIdx.x, lane = threadIdx.x;
  if (j >= ncols) return;
  double acc[KK];
#pragma unroll
  for (int q = 0; q < KK; ++q) acc[q] = 0.0;
  for (int t = lane; t < T; t += WAVE) {
    double v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = V[((size_t)j * T + t) * K + k];
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) acc[lidx(a, b)] = fma(v[a], v[b], acc[lidx(a, b)]);
  }
#pragma unroll
  for (int q = 0; q < KK; ++q) {
    const double t = wave_sum(acc[q]);
    if (lane == 0) out[(size_t)j * KK + q] = t;
  }
}

// ============================================================================
// W half-sweep: batched K x K Cholesky draw, one lane per row  (BTF_K_W_SOLVE)
//   factor.py:349-362:  Q = X'CX + I/sigma2 ; Lt = chol(Q)' ;
//                       W[i,:d] = cho_solve(Lt, m) + Lt^-1 z
// ============================================================================
struct WSolveArgs {
  const double* part; int nch; int ld;
  // compat="reference", stale cached weights (MODE 2 accumulation): the Gram sums of row il stand in the partials' column
  // gsrc[il] - the row whose weights it shares (factor.py:349-357) - and were accumulated there ONCE; nullptr: its own
  const int* gsrc;
  const double* gpart; int ngp;
  int weighted;
  double s;        // 1/nu2 (Gaussian) or 1 (Binomial)
  double sR;       // s * nreps : scale of the shared Gram on the complete-data path
  double inv_sigma2;
  double* W; int row0; int nl;
  const double* z; unsigned long long seed; unsigned long long stream;
  int* status;     // [0] = failure flag, [1] = first failing row
  double* gout;    // [gridDim.x][KK] Gram partial of the rows this workgroup wrote (W'W for the V half-sweep)
  const double* hyp;   // device-resident scalars (HYP_*) or nullptr: when set they override s, sR, inv_sigma2
  double Rrep;         // nreps (sR = s * Rrep)
  int hyp_noise;       // 1: the noise scale s comes from hyp[HYP_NU2] (scalar-nu2 models only)
  CurveLists cv;       // curve-structured counts: deficient columns of every row (global row index), or ptr == nullptr
  const double* cv_blocks;   // [M][KK] per-column Grams V_j'V_j
  // full sweeps: lam2 | rest drawn by ONE MORE workgroup of this launch (the last block index; hyp == nullptr: none) - it
  // needs the column sums the Tau2 chain left in the W accumulation launch in front of this one and nothing of the solve;
  // the V half-sweep behind it then finds lam2 drawn and can use the precomputed prior band (btf_abi.hip)
  int nside;           // workgroups behind the solves' (0: none): the two side tasks below
  LamSide lam;
  BandSide band;       // ... and behind it one workgroup per local column for the prior band of the V half-sweep (pband == nullptr: none)
#ifdef BTF_WS_STAMPS
  long long* dbg;      // diagnostic builds: [gridDim.x][6] shader-clock stamps of wave 0 (scripts/ws_stamps.py)
#endif
};
#ifdef BTF_WS_STAMPS
#define WS_STAMP(n) do { if (threadIdx.x == 0 && a.dbg) a.dbg[(size_t)blockIdx.x * 6 + (n)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define WS_STAMP(n) do {} while (0)
#endif

// device-resident scalar hyper-parameters (rng="device": drawn by scalars_kernel / lam2_kernel,
// read by the half-sweep kernels, so that a full sweep needs no host round trip)
enum { HYP_NU2 = 0, HYP_SIGMA2 = 1, HYP_LAM2 = 2, HYP_LAM2A = 3, HYP_SSE = 4, HYP_WSQ = 5, HYP_COUNT = 8 };
__device__ inline void band_side(const BandSide& bs, int j, double* itau, int nthreads) {
  const double* tau = bs.Tau2 + (size_t)(bs.col0 + j) * bs.nD;
  // independent of lam2: this thread's Tau2 values (two: nD <= 2 nthreads) and the CSR bounds of its entries (two)
  double tv[2] = {1.0, 1.0};
  int p0[2] = {0, 0}, p1[2] = {0, 0};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = threadIdx.x + u * nthreads, e0 = threadIdx.x + u * nthreads;
    if (r < bs.nD) tv[u] = tau[r];
    if (e0 < bs.TD1) { p0[u] = bs.st_ptr[e0]; p1[u] = bs.st_ptr[e0 + 1]; }
  }
  double lam2 = bs.hyp ? bs.hyp[HYP_LAM2] : bs.lam2;
  if (bs.lam_flag) {
    // (bounded: the lam2 workgroup has a lower block index - dispatched first - and publishes whatever happens)
    __shared__ int okf;
    if (threadIdx.x == 0) {
      bool seen = false;
      for (unsigned spins = 0; spins < (1u << 22); ++spins) {
        if (__hip_atomic_load((const gu32_t*)bs.lam_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == bs.epoch) { seen = true; break; }
        __builtin_amdgcn_s_sleep(8);
      }
      okf = seen ? 1 : 0;
      if (!seen && atomicCAS(&bs.status[0], 0, 2) == 0) bs.status[1] = -2;
    }
    __syncthreads();
    if (!okf) return;
    lam2 = load_sc1(bs.lam_pub + HYP_LAM2);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = threadIdx.x + u * nthreads;
    if (r < bs.nD) itau[r] = 1.0 / __dmul_rn(lam2, tv[u]);
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e0 = threadIdx.x + u * nthreads;
    if (e0 < bs.TD1) {
      double s = 0.0;
      for (int e = p0[u]; e < p1[u]; ++e) s = fma(bs.st_coef[e], itau[bs.st_row[e]], s);
      bs.pband[(size_t)j * bs.TD1 + e0] = s;
      if (bs.pimg) {
        const int t = e0 / bs.img_D1, d = e0 - t * bs.img_D1;
        double* im = bs.pimg + (size_t)j * 2 * bs.img_PB;
        im[e0] = s;
        if (t + d < bs.img_T) im[bs.img_PB + (bs.img_T - 1 - t - d) * bs.img_D1 + d] = s;
      }
    }
  }
}


constexpr int WS_ROWS = 64;   // rows per workgroup (one per lane)
// waves per workgroup: the chunk partials are summed WS_SPLIT-way in parallel (LDS-bounded)
__host__ __device__ constexpr int ws_split(int K) { return K <= 6 ? 8 : (K <= 8 ? 4 : 2); }
// (complete data, 16 waves so that the 64 chunks of a C3 row come in with one batch of loads per wave: measured
//  10.2 us against 8.2 us for 8 waves - the 1024-thread workgroup costs more than the second batch)
// (with 8 rows per workgroup - 8 chunk subgroups per wave - four waves cover the 32 chunks of a C3 row with one chunk per
//  (wave, subgroup) pair: 6.1 us against 6.35 us for eight waves; sixteen do not fit LDS)
#ifndef BTF_WS_SPLIT_U
#define BTF_WS_SPLIT_U 4
#endif
__host__ __device__ constexpr int ws_split_of(int K, bool weighted) { return weighted ? ws_split(K) : (K <= 6 ? BTF_WS_SPLIT_U : ws_split(K)); }
// bound on the doubles of fused Gram partials a V half-sweep hands to w_solve (historically its LDS staging area)
__host__ __device__ constexpr size_t ws_gram_stage(int K, bool weighted) {
  return (size_t)ws_split_of(K, weighted) * (K + tri(K)) * WS_ROWS;
}

// RW rows per workgroup (8..64): lane = (row rr = lane % RW, chunk subgroup sub = lane / RW).  A factor of N rows
// gives only N/64 workgroups with a lane per row - 8 at C3, every one summing 64 chunks of partials in a few
// dependent batches of loads; with RW = 8 there are 64 workgroups and the 8 x 64/RW (wave, subgroup) pairs of a
// row take ONE chunk each.  The solve itself is a per-lane latency chain whatever the number of live lanes.
__host__ __device__ constexpr int ws_rows_for(int nl) { return nl > 2048 ? 64 : (nl > 1024 ? 32 : (nl > 512 ? 16 : 8)); }

// sum of one (row, value) pair over the chunks of the accumulation partials, in chunk order (the canonical order of the
// complete-data W step: w_solve_kernel and the fused tail both call this).  p: the pair's slot in chunk 0, cst: chunk
// stride.  WS_CHB loads in flight.  SC1: the partials were written by other workgroups of the SAME launch (sc1 loads).
constexpr int WS_CHB = 32;
template <bool SC1 = false>
__device__ __forceinline__ double chunk_sum_seq(const double* __restrict__ p, size_t cst, int nch) {
  double s = 0.0;
  int c = 0;
  for (; c + WS_CHB <= nch; c += WS_CHB) {
    double x[WS_CHB];
#pragma unroll
    for (int u = 0; u < WS_CHB; ++u) x[u] = SC1 ? load_sc1(p + (size_t)(c + u) * cst) : p[(size_t)(c + u) * cst];
#pragma unroll
    for (int u = 0; u < WS_CHB; ++u) s += x[u];
  }
  if (c < nch) {                                          // the last, partial batch: the same loads under a guard
    double x[WS_CHB];
#pragma unroll
    for (int u = 0; u < WS_CHB; ++u) x[u] = c + u < nch ? (SC1 ? load_sc1(p + (size_t)(c + u) * cst) : p[(size_t)(c + u) * cst]) : 0.0;
#pragma unroll
    for (int u = 0; u < WS_CHB; ++u) if (c + u < nch) s += x[u];
  }
  return s;
}

template <int K, bool WEIGHTED, int RW = WS_ROWS>
__global__ __launch_bounds__(WS_ROWS * ws_split_of(K, WEIGHTED)) void w_solve_kernel(WSolveArgs a) {
  constexpr int KK = tri(K);
  constexpr int SUB = WS_ROWS / RW;
  constexpr int WS_SPLIT = ws_split_of(K, WEIGHTED);
  constexpr int NVMAX = WEIGHTED ? K + KK : K;
  constexpr int NV = NVMAX;
  WS_STAMP(0);
  // chunks whose loads are in flight together (weighted rows of K >= 6 carry 27+ values per chunk: two chunks' worth of
  // them beside the running sums did not fit the 256 VGPRs of the 8-wave workgroup - 156 spilled at K = 6)
  constexpr int UNR = WEIGHTED ? (K + KK > 24 ? 1 : 2) : 4;
  __shared__ double G[KK];
  __shared__ double red[WS_SPLIT][K + KK][WS_ROWS];   // also the staging area of the Gram partials
  __shared__ double zsh[K][WS_ROWS];
  __shared__ double cvq[WEIGHTED ? 1 : RW][WEIGHTED ? 1 : KK];   // curve counts: sum_{j in D(i)} (R - c_ij) V_j'V_j per row
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  if (a.nside) {
    // block indices behind the solves': the lam2 workgroup (if any), then the band's (if any) - all uniform per workgroup
    const int nsolve = (int)gridDim.x - a.nside;
    if ((int)blockIdx.x >= nsolve) {
      int b = (int)blockIdx.x - nsolve;
      if (a.lam.hyp) {
        if (b == 0) { sweep_lam_side<WS_ROWS * WS_SPLIT / WAVE>(a.lam, &red[0][0][0]); return; }
        --b;
      }
      if (a.band.pband && b < a.band.ml) band_side(a.band, b, &red[0][0][0], WS_ROWS * WS_SPLIT);
      return;
    }
  }
  if constexpr (!WEIGHTED) {
    if (a.cv.ptr) {
      // The rows of a workgroup are consecutive, so their lists are ONE range of the CSR arrays: thread (g, q) takes
      // entries g, g + G, ... and parks def * block[q] in LDS (`red` is free until the chunk sums land there); thread
      // (row, q) then adds its row's entries in list order.  Two dependent round trips per pass, whatever the lists.
      constexpr int NT = WS_ROWS * WS_SPLIT, G = NT / KK, CAP = (WS_SPLIT * (K + KK) * WS_ROWS) / KK;
      const int i0 = a.row0 + blockIdx.x * RW, i1 = min(i0 + RW, a.row0 + a.nl);
      const int e0 = a.cv.ptr[i0], e1 = a.cv.ptr[i1];
      double* stage = &red[0][0][0];
      const int g = threadIdx.x / KK, q = threadIdx.x - g * KK;
      if (g < G)
        for (int r = g; r < RW; r += G) cvq[r][q] = 0.0;
      for (int eb = e0; eb < e1; eb += CAP) {
        const int ee = min(eb + CAP, e1);
        if (g < G) {
          for (int e = eb + g; e < ee; e += G)
            stage[(size_t)(e - eb) * KK + q] = a.cv.def[e] * a.cv_blocks[(size_t)a.cv.idx[e] * KK + q];
        }
        __syncthreads();
        if (g < G) {
          for (int r = g; r < RW && i0 + r < i1; r += G) {          // thread (g, q) owns rows g, g + G, ... of entry q
            double acc = cvq[r][q];
            for (int e = max(a.cv.ptr[i0 + r], eb); e < min(a.cv.ptr[i0 + r + 1], ee); ++e) acc += stage[(size_t)(e - eb) * KK + q];
            cvq[r][q] = acc;
          }
        }
        __syncthreads();
      }
      // (published by the barrier that follows the chunk sums)
    }
  }
  if (a.hyp) {
    if (a.hyp_noise) { a.s = 1.0 / a.hyp[HYP_NU2]; a.sR = a.s * a.Rrep; }
    a.inv_sigma2 = 1.0 / a.hyp[HYP_SIGMA2];
  }
  // Gram partials of the fixed factor: fetched NOW, together with the first batch of chunk loads below (one
  // global round trip for both), summed after the chunk sums have left for LDS
  double gx[8];
  bool g_early = false;
  if constexpr (!WEIGHTED) {
    g_early = gram_early_ok(a.ngp, KK);
    if (g_early) reduce_gram_fetch(a.gpart, a.ngp, KK, gx);
    else reduce_gram(a.gpart, a.ngp, KK, a.sR, &red[0][0][0], G);
  }
  const int rr = lane % RW, sub = lane / RW;
  const int il = blockIdx.x * RW + rr;
  constexpr int CS = WS_SPLIT * SUB;                     // chunk stride of one (wave, subgroup) pair
  if constexpr (!WEIGHTED) {
    // Complete data (and curve counts): the chunk sums of a (row, value) pair are added by ONE thread in chunk order
    // 0, 1, 2, ... - an order that depends on nothing but the number of chunks, so that the tail of the fused W launch
    // (btf_fused.h: w_tail_rows) lands on the same bits whatever its own geometry.  Thread t < RW K takes the pair
    // (value t / RW, row t % RW), WS_CHB chunks' loads in flight; the normals are drawn meanwhile by the other threads.
    constexpr int NT = WS_ROWS * WS_SPLIT;
    const size_t cst = (size_t)NV * a.ld;
    for (int t = threadIdx.x; t < RW * K; t += NT) {
      const int v = t / RW, r = t - v * RW;
      const int ilr = blockIdx.x * RW + r;
      red[0][v][r] = ilr < a.nl ? chunk_sum_seq(a.part + (size_t)v * a.ld + ilr, cst, a.nch) : 0.0;
    }
    // (threads from the far end take the normals: at RW = 8, K = 5 the first 40 threads sum chunks, the last 40 draw)
    for (int t = NT - 1 - (int)threadIdx.x; t < RW * K; t += NT) {
      const int k = t / RW, r = t - k * RW;
      const int ilr = blockIdx.x * RW + r;
      double zv = 0.0;
      if (ilr < a.nl) {
        const int i = a.row0 + ilr;
        const long long zoff = w_z_offset(i, K);
        const int d = i + 1 < K ? i + 1 : K;
        if (k < d) zv = a.z ? a.z[zoff + k] : philox_normal(a.seed, a.stream, (unsigned long long)(zoff + k));
      }
      zsh[k][r] = zv;
    }
    if (g_early) reduce_gram_finish(gx, a.ngp, KK, a.sR, &red[1][0][0], G);     // ends with a barrier (red[1]: scratch; red[0] holds the sums)
  } else {
  // stage 1: pair (grp, sub) sums chunks grp*SUB+sub, +CS, ... (fixed order => deterministic).  The first
  // batch of loads is issued BEFORE the Philox normals of the row are computed (component k by wave
  // k % WS_SPLIT), so the transcendental work hides under the memory latency.
  {
    double part[NVMAX];
#pragma unroll
    for (int v = 0; v < NVMAX; ++v) part[v] = 0.0;
    const size_t cst = (size_t)NV * a.ld;               // chunk stride
    const bool live1 = il < a.nl;
    // (stale cached weights: the Gram sums of this row stand at its source's column of the partials)
    const ptrdiff_t goff = (a.gsrc && live1) ? (ptrdiff_t)a.gsrc[il] - (ptrdiff_t)il : 0;
    int c = grp * SUB + sub;
    // (slot u of the first batch exists if chunk c + u CS does: at C3 the 64 (wave, subgroup) pairs of a row own ONE chunk
    //  each, and a batch that asked for all UNR of them never left before the normals - 1.6 us of Philox in front of the
    //  only round of loads)
    double x0[UNR][NV];
    const bool first = live1 && c < a.nch;
    int nfirst = 0;
    if (first) {
      const double* p = a.part + (size_t)c * cst + il;
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const bool has = c + u * CS < a.nch;
        nfirst += has ? 1 : 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) x0[u][v] = has ? p[(ptrdiff_t)((size_t)u * CS * cst + (size_t)v * a.ld) + (v >= K ? goff : 0)] : 0.0;
      }
    }
    if (live1) {
      const int i = a.row0 + il;
      const long long zoff = w_z_offset(i, K);
      const int d = i + 1 < K ? i + 1 : K;
      // (component k by the pair CS - 1 - k, i.e. by the LAST wave: wave 0 - which finishes the rows alone - goes from its
      //  loads straight to the butterfly)
      for (int k = CS - 1 - (grp * SUB + sub); k < K; k += CS)
        zsh[k][rr] = k < d ? (a.z ? a.z[zoff + k] : philox_normal(a.seed, a.stream, (unsigned long long)(zoff + k))) : 0.0;
    }
    if (first) {
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (u < nfirst) {
#pragma unroll
          for (int v = 0; v < NV; ++v) part[v] += x0[u][v];
        }
      c += nfirst * CS;
    }
    if (live1) {
      for (; c + (UNR - 1) * CS < a.nch; c += UNR * CS) {   // UNR chunks' loads in flight, added in order
        const double* p = a.part + (size_t)c * cst + il;
        double x[UNR][NV];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
          for (int v = 0; v < NV; ++v) x[u][v] = p[(ptrdiff_t)((size_t)u * CS * cst + (size_t)v * a.ld) + (v >= K ? goff : 0)];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
          for (int v = 0; v < NV; ++v) part[v] += x[u][v];
      }
      for (; c < a.nch; c += CS) {
        const double* p = a.part + (size_t)c * cst + il;
#pragma unroll
        for (int v = 0; v < NV; ++v) part[v] += p[(ptrdiff_t)((size_t)v * a.ld) + (v >= K ? goff : 0)];
      }
    }
    // (the staging area of the Gram partials is `red` itself: this wave's sums wait in registers meanwhile)
    if constexpr (!WEIGHTED) {
      if (g_early) reduce_gram_finish(gx, a.ngp, KK, a.sR, &red[0][0][0], G);     // ends with a barrier
    }
    // the SUB subgroups of a wave hold shares of the same rows: butterfly over the lane bits above RW (fixed
    // order), so that one value per (wave, row) goes to LDS
#pragma unroll
    for (int st = RW; st < WS_ROWS; st <<= 1)
#pragma unroll
      for (int v = 0; v < NVMAX; ++v) part[v] += __shfl_xor(part[v], st);
#pragma unroll
    for (int v = 0; v < NVMAX; ++v) red[grp][v][lane] = part[v];
  }
  }
  WS_STAMP(1);
  __syncthreads();
  WS_STAMP(2);
  // Weighted rows of K = 6 and 8 leave WS_SPLIT (K + KK) = 216 / 176 values per row in LDS: wave 0 summing them all on
  // its own kept two hundred loaded doubles alive next to the solve's registers (148 VGPRs spilled at K = 6).  There
  // the waves first add the WS_SPLIT shares of every WS_SPLIT-th value each (same order: bit-identical sums), in place.
  // (round 4: from 128 values on, i.e. nembeds 5 too - wave 0's 160 dependent LDS reads on 8 live lanes were 0.5 us of the
  //  9.0 us launch at C3 with 5 % of the replicates missing)
#ifndef BTF_WS_TWO_LEVEL_MIN
#define BTF_WS_TWO_LEVEL_MIN 128
#endif
  constexpr bool TWO_LEVEL = WEIGHTED && WS_SPLIT * (K + KK) > BTF_WS_TWO_LEVEL_MIN;
  if constexpr (TWO_LEVEL) {
    for (int v = grp; v < NV; v += WS_SPLIT) {
      double s2 = 0.0;
#pragma unroll
      for (int w = 0; w < WS_SPLIT; ++w) s2 += red[w][v][lane];
      red[0][v][lane] = s2;
    }
    __syncthreads();
  }
  constexpr int NSH = (TWO_LEVEL || !WEIGHTED) ? 1 : WS_SPLIT;          // shares wave 0 still has to add (complete data: the pair's sum is whole)
  if (grp != 0) return;                      // wave 0 finishes: one lane per row
  WS_STAMP(3);
  const bool live = lane < RW && il < a.nl;
  const int i = a.row0 + (live ? il : 0);
  const int d = i + 1 < K ? i + 1 : K;
  double m[K], Q[KK];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NSH; ++w) s += red[w][k][rr];
    m[k] = __dmul_rn(s, a.s);
  }
#pragma unroll
  for (int q = 0; q < KK; ++q) {
    if constexpr (WEIGHTED) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < NSH; ++w) s += red[w][K + q][rr];
      Q[q] = __dmul_rn(s, a.s);
    } else {
      Q[q] = G[q];
    }
  }
  if constexpr (!WEIGHTED) {
    if (a.cv.ptr && live) {          // curve-structured counts: take the missing replicates' share out again
#pragma unroll
      for (int q = 0; q < KK; ++q) Q[q] = fma(-a.s, cvq[rr][q], Q[q]);
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) Q[lidx(k, k)] += a.inv_sigma2;
  // rows/cols >= d are frozen (W is lower triangular in its first K rows): identity there
#pragma unroll
  for (int r = 0; r < K; ++r) {
    if (r >= d) {
      m[r] = 0.0;
#pragma unroll
      for (int c = 0; c <= r; ++c) Q[lidx(r, c)] = (r == c) ? 1.0 : 0.0;
    }
  }
  // in-register Cholesky (lower, packed); 1/L_cc kept so that the solves multiply instead of divide.  (Every product
  // that an addition follows is a __dmul_rn / __dadd_rn: the fused tail of btf_fused.h repeats this arithmetic in another
  // code shape, and -ffp-contract=fast must not be allowed to fuse differently there)
  bool ok = true;
  double invl[K];
#pragma unroll
  for (int c = 0; c < K; ++c) {
    double p = Q[lidx(c, c)];
#pragma unroll
    for (int q = 0; q < c; ++q) p = fma(-Q[lidx(c, q)], Q[lidx(c, q)], p);
    if (!(p > 0.0)) ok = false;
    // 1/sqrt(p) by v_rsq_f64 + two Newton steps (full precision), l = p / sqrt(p): ~12 instructions on the lane's
    // dependent chain instead of the ~55 of an IEEE square root and division
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
  WS_STAMP(4);
  // y = L^-1 m ; x = L^-T (y + z)
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
    if (r < d) y[r] = __dadd_rn(y[r], zsh[r][rr]);
#pragma unroll
  for (int r = K - 1; r >= 0; --r) {
    double v = y[r];
#pragma unroll
    for (int c = r + 1; c < K; ++c) v = fma(-Q[lidx(c, r)], y[c], v);
    y[r] = __dmul_rn(v, invl[r]);
  }
  // the row as it now stands (entries >= d keep their stored value), written back and
  // folded into this workgroup's share of W'W for the next V half-sweep
  double wrow[K];
#pragma unroll
  for (int r = 0; r < K; ++r) {
    const bool fresh = live && ok && r < d;
    if (fresh) a.W[(size_t)i * K + r] = y[r];
    wrow[r] = fresh ? y[r] : 0.0;
  }
  if (live && (d < K || !ok)) {   // only the first K rows keep stored entries (their frozen upper triangle)
#pragma unroll
    for (int r = 0; r < K; ++r)
      if (!(ok && r < d)) wrow[r] = a.W[(size_t)i * K + r];
  }
  if (a.gout) {
    // W'W of this workgroup's 64 rows as a 16 x 16 x 64 f64 MFMA product (K <= 10 of the 16 rows / columns
    // used): the rows are staged in LDS as [k][row]; step s takes rows 4s..4s+3, lane (kk = lane>>4,
    // i = lane&15) holds W[4s+kk][i] and is both the A and the B operand.  (Fifteen 6-stage wave
    // reductions cost 3.6 of this kernel's 10.9 us.)
    typedef double v4f64_w __attribute__((ext_vector_type(4)));
    double* stg = &red[0][0][0];                     // free again: [K][64]
#pragma unroll
    for (int r = 0; r < K; ++r) stg[r * WS_ROWS + lane] = wrow[r];
    const int kk = lane >> 4, ii = lane & 15;
    const double* src = stg + (ii < K ? ii : 0) * WS_ROWS + kk;
    v4f64_w acc = {0.0, 0.0, 0.0, 0.0};
    // (only the first RW staged rows are live - the others hold exact zeros: (RW + 3) / 4 steps give the same bits as 16)
#pragma unroll
    for (int sidx = 0; sidx < (RW + 3) / 4; ++sidx) {
      const double x = ii < K ? src[4 * sidx] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
    }
    // element (row 4r + kk, col ii) sits in acc[r] of this lane; keep the lower triangle of the K x K corner
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * r + kk;
      if (row < K && ii <= row) a.gout[(size_t)blockIdx.x * KK + lidx(row, ii)] = acc[r];
    }
  }
  WS_STAMP(5);
}

// ============================================================================
// V half-sweep: block-banded Cholesky sampler, one wave per column (BTF_K_V_BANDED)
//   the fast_mvn equivalent (fast_mvn.py:35-47 with the jitter loop :62-68)
//   unknowns in depth-major order n = t*K + k, half-bandwidth bw = (tf+1)*K
//   band storage (LDS or HBM scratch): Bc[n][a] = Q[n+a, n], a = 0..bw
// ============================================================================
struct VBandArgs {
  // likelihood part
  const double* part; int nch; int ld;   // accum partials [nch][NV][ld], column j at offset j*T
  // compat="reference", stale cached weights (MODE 2 accumulation): the Gram blocks of local column j stand at the partials'
  // column gsrc[j * T] / T - the column whose Q_likelihood it reuses (factor.py:394-400), accumulated there ONCE; nullptr: its own
  const int* gsrc;
  const double* gpart; int ngp;          // Gram partials (complete-data path)
  int weighted;
  double s, sR;
  // prior part
  const double* Tau2; double lam2; int nD;
  const int* st_ptr; const int* st_row; const double* st_coef;  // Delta'.Delta stencil per (t,d)
  int T, TF;
  int col0, ml;
  double* V;
  const double* z; unsigned long long seed; unsigned long long stream;
  double eps0; int attempts;
  int* status;   // [0] flag, [1] failing column
  int* tries;    // [ml]
  double* gband; // HBM scratch for the band when it does not fit LDS, else nullptr
  size_t gband_stride;
  long long* dbg; // diagnostic phase stamps [ml][6] (nullptr in normal runs)
  double* gout;   // [ml][KK] V_j'V_j of the freshly drawn column (V'V partials for the next W half-sweep) or nullptr
  const double* pband; // [ml][T][TF+2] prior band Delta' diag(1/(lam2 Tau2_j)) Delta, entry (t+d,t) (fast kernel)
  const double* hyp;   // device-resident scalars or nullptr (see WSolveArgs)
  double Rrep;
  int hyp_noise;
  int panel4;          // 1: panelised MFMA factorisation where it applies (bw == 15)
  const int* fill;     // band assembly program of the twisted kernel: [nfill][4] = {dst, src, diag-src or -1, 0} (LDS word offsets)
  int nfill;           // multiple of the workgroup size (padded with writes to a dummy word)
  int ql_global;       // twisted kernel, weighted data: 1 = likelihood blocks fetched from the partials by the assembly program, not staged in LDS
  CurveLists cv;       // curve-structured counts: deficient rows of every column (global column index), or ptr == nullptr
  const double* cv_W;  // the factor W (rows of the rank-one terms)
  const int* st_drow; const double* st_dcoef;   // the stencil with VS_MAXE fixed slots per (t,d): the twisted kernel builds
                                                // its prior band itself when pband == nullptr (no prior_band_kernel launch)
};
constexpr int PB_MAXE = 16;                     // (= VS_MAXE of btf_spectral.h)
__device__ __forceinline__ void vband_load_hyp(VBandArgs& a) {
  if (a.hyp) {
    if (a.hyp_noise) { a.s = 1.0 / a.hyp[HYP_NU2]; a.sR = a.s * a.Rrep; }
    a.lam2 = a.hyp[HYP_LAM2];
  }
}

// Prior band of every local column at once (depends on the hyper-parameters only, so it is
// recomputed when they change, not per half-sweep): one thread per (column, t, d), Delta
// rows ascending as in the sparse product of factor.py:404-405.
static __global__ void prior_band_kernel(const double* __restrict__ Tau2, double lam2, int nD,
                                  const int* __restrict__ st_ptr, const int* __restrict__ st_row,
                                  const double* __restrict__ st_coef, int TD1, int col0, int ml,
                                  double* __restrict__ pband, const double* __restrict__ hyp, int spectral_form,
                                  double* __restrict__ pimg, int img_T, int img_D1, int img_PB) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ml * TD1) return;
  if (hyp) lam2 = hyp[HYP_LAM2];
  const int j = idx / TD1, e0 = idx - j * TD1;
  const double* tau = Tau2 + (size_t)(col0 + j) * nD;
  double s = 0.0;
  if (spectral_form) {
    // the arithmetic of v_spectral_kernel's own band (reciprocal of the rounded product, then one fma per penalty row): the
    // fused V tails (btf_fused.h) load this band and must land on the bits of the four-launch path
    for (int e = st_ptr[e0]; e < st_ptr[e0 + 1]; ++e) {
      const double it = 1.0 / __dmul_rn(lam2, tau[st_row[e]]);
      s = fma(st_coef[e], it, s);
    }
  } else {
    for (int e = st_ptr[e0]; e < st_ptr[e0 + 1]; ++e) s += st_coef[e] / (lam2 * tau[st_row[e]]);
  }
  pband[idx] = s;
  // the same band as the LDS image [P | Pm] of the column (dataflow tails of the fused V launch, btf_fused.h: a wave copies
  // it as it lies): entry (t, d) and its mirror image Pm[T-1-t-d][d]; the zero rows were zeroed once, at allocation
  if (pimg) {
    const int t = e0 / img_D1, d = e0 - t * img_D1;
    double* im = pimg + (size_t)j * 2 * img_PB;
    im[e0] = s;
    if (t + d < img_T) im[img_PB + (img_T - 1 - t - d) * img_D1 + d] = s;
  }
}

// The spectral form of the same band, one WORKGROUP per column (T <= 128: the fused V launch's shapes): the reciprocal
// 1 / (lam2 Tau2[j][r]) once per penalty row into LDS - one round trip, beside the threads' CSR bounds - then every band
// entry from its stencil's entries.  Two dependent round trips and nD divisions per column where the thread-per-entry
// kernel above takes four and a division per term (4.4 us -> see profiles/README.md); the same bits (the reciprocal of the
// rounded product, one fma per penalty row in row order).  Full sweeps rebuild the band every sweep.
static __global__ __launch_bounds__(256) void prior_band_cols_kernel(const double* __restrict__ Tau2, double lam2, int nD,
                                  const int* __restrict__ st_ptr, const int* __restrict__ st_row,
                                  const double* __restrict__ st_coef, int TD1, int col0,
                                  double* __restrict__ pband, const double* __restrict__ hyp,
                                  double* __restrict__ pimg, int img_T, int img_D1, int img_PB) {
  __shared__ double itau[256];
  const int j = blockIdx.x;
  if (hyp) lam2 = hyp[HYP_LAM2];
  const double* tau = Tau2 + (size_t)(col0 + j) * nD;
  int p0[2], p1[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e0 = threadIdx.x + u * 256;
    p0[u] = e0 < TD1 ? st_ptr[e0] : 0;
    p1[u] = e0 < TD1 ? st_ptr[e0 + 1] : 0;
  }
  for (int r = threadIdx.x; r < nD; r += 256) itau[r] = 1.0 / __dmul_rn(lam2, tau[r]);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e0 = threadIdx.x + u * 256;
    if (e0 < TD1) {
      double s = 0.0;
      for (int e = p0[u]; e < p1[u]; ++e) s = fma(st_coef[e], itau[st_row[e]], s);
      pband[(size_t)j * TD1 + e0] = s;
      if (pimg) {
        const int t = e0 / img_D1, d = e0 - t * img_D1;
        double* im = pimg + (size_t)j * 2 * img_PB;
        im[e0] = s;
        if (t + d < img_T) im[img_PB + (img_T - 1 - t - d) * img_D1 + d] = s;
      }
    }
  }
}

// generic banded factor + solves on a band at `Bc` (LDS or global), single wave.
// returns false if a pivot is not positive.
// pair table: entry q = (a << 8) | b enumerates 1 <= b <= a <= bw (filled once per kernel)
__device__ inline void fill_pair_table(unsigned short* ptab, int bw) {
  const int npairs = bw * (bw + 1) / 2;
  for (int q = threadIdx.x; q < npairs; q += WAVE) {
    int a = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5) + 1;
    while ((a - 1) * a / 2 > q) --a;
    while (a * (a + 1) / 2 <= q) ++a;
    const int b = q - (a - 1) * a / 2 + 1;
    ptab[q] = (unsigned short)((a << 8) | b);
  }
}

__device__ inline bool banded_factor_forward(double* Bc, double* rhs, double* invd, const unsigned short* ptab,
                                             int n, int bw) {
  const int lane = threadIdx.x;
  const int R1 = bw + 1;
  const int npairs = bw * (bw + 1) / 2;
  for (int nn = 0; nn < n; ++nn) {
    const int len = min(bw, n - 1 - nn);
    double* col = Bc + (size_t)nn * R1;
    double v = (lane <= len) ? col[lane] : 0.0;
    const double p = bcast_first(v);
    if (!(p > 0.0)) return false;
    const double l = sqrt(p);
    const double inv = 1.0 / l;
    double x = v * inv;
    if (lane == 0) {
      x = l;
      invd[nn] = inv;
    }
    if (lane <= len) col[lane] = x;
    // forward substitution folded in: y_nn = rhs[nn]/l ; rhs[nn+a] -= L[nn+a,nn] y_nn
    const double yn = rhs[nn] * inv;
    if (lane == 0) rhs[nn] = yn;
    else if (lane <= len) rhs[nn + lane] -= x * yn;
    // trailing update: A[nn+a, nn+b] -= L[nn+a,nn] L[nn+b,nn], 1 <= b <= a <= len
    for (int q = lane; q < npairs; q += WAVE) {
      const int ab = ptab[q];
      const int a = ab >> 8, b = ab & 255;
      if (a <= len) {
        const double xa = col[a];
        const double xb = col[b];
        Bc[(size_t)(nn + b) * R1 + (a - b)] -= xa * xb;
      }
    }
  }
  return true;
}

__device__ inline void banded_backward(const double* Bc, double* rhs, const double* invd, int n, int bw) {
  const int lane = threadIdx.x;
  const int R1 = bw + 1;
  for (int nn = n - 1; nn >= 0; --nn) {
    const double xn = rhs[nn] * invd[nn];
    if (lane == 0) rhs[nn] = xn;
    else if (lane <= bw && lane <= nn) rhs[nn - lane] -= Bc[(size_t)(nn - lane) * R1 + lane] * xn;
  }
}

template <int K>
__global__ __launch_bounds__(WAVE) void v_banded_kernel(VBandArgs a) {
  vband_load_hyp(a);
  constexpr int KK = tri(K);
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int j = blockIdx.x;
  const int jg = a.col0 + j;
  const int T = a.T, n = T * K, D1 = a.TF + 2, bw = (a.TF + 1) * K, R1 = bw + 1;
  const int NV = a.weighted ? K + KK : K;

  // LDS carve-up
  double* rhs = lds;                // n
  double* m0 = rhs + n;             // n
  double* invd = m0 + n;            // n
  double* P = invd + n;             // T*D1 prior band  P[t][d] = (Delta' Lambda Delta)[t+d, t]
  // likelihood blocks: weighted T*KK, else KK.  With the band in HBM scratch (long depth axes) the per-depth blocks
  // of the weighted case go there too, behind the band: only the vectors stay on chip
  const bool ql_hbm = a.gband != nullptr && a.weighted;
  double* Ql = ql_hbm ? a.gband + (size_t)j * a.gband_stride + (size_t)n * R1 : P + T * D1;
  double* Qe = P + T * D1 + (ql_hbm ? 0 : (a.weighted ? T * KK : KK));
  unsigned short* ptab = reinterpret_cast<unsigned short*>(Qe);   // bw(bw+1)/2 entries, padded to doubles
  double* Bl = Qe + (bw * (bw + 1) / 2 + 3) / 4;
  double* Bc = a.gband ? a.gband + (size_t)j * a.gband_stride : Bl;
  fill_pair_table(ptab, bw);

  // likelihood mean part and Gram blocks (fixed summation order over chunks)
  for (int idx = lane; idx < n; idx += WAVE) {
    const int t = idx / K, k = idx - t * K;
    const double* p = a.part + (size_t)k * a.ld + (size_t)j * T + t;
    double s = 0.0;
    for (int c = 0; c < a.nch; ++c) s += p[(size_t)c * NV * a.ld];
    m0[idx] = s * a.s;
  }
  if (a.weighted) {
    for (int idx = lane; idx < T * KK; idx += WAVE) {
      const int t = idx / KK, q = idx - t * KK;
      const double* p = a.part + (size_t)(K + q) * a.ld + (size_t)(a.gsrc ? a.gsrc[j * T] / T : j) * T + t;
      double s = 0.0;
      for (int c = 0; c < a.nch; ++c) s += p[(size_t)c * NV * a.ld];
      Ql[idx] = s * a.s;
    }
  } else {
    reduce_gram(a.gpart, a.ngp, KK, a.sR, Bc, Ql);   // Bc is free scratch until the assembly
  }
  // prior band from the Delta stencil, rows ascending (the order of the sparse product)
  for (int idx = lane; idx < T * D1; idx += WAVE) {
    double s = 0.0;
    for (int e = a.st_ptr[idx]; e < a.st_ptr[idx + 1]; ++e)
      s += a.st_coef[e] / (a.lam2 * a.Tau2[(size_t)jg * a.nD + a.st_row[e]]);
    P[idx] = s;
  }
  __syncthreads();

  double shift = 0.0, eps = a.eps0;
  int tried = 0;
  bool ok = false;
  while (true) {
    // assemble the band
    for (int idx = lane; idx < n * R1; idx += WAVE) {
      const int nn = idx / R1, aa = idx - nn * R1;
      const int t = nn / K, k = nn - t * K;
      double v = 0.0;
      if (aa < K - k) {
        v = a.weighted ? Ql[t * KK + lidx(k + aa, k)] : Ql[lidx(k + aa, k)];
        if (aa == 0) v += P[t * D1] + shift;
      } else {
        const int dd = aa / K;
        if (dd * K == aa && dd < D1 && t + dd < T) v = P[t * D1 + dd];
      }
      Bc[idx] = v;
    }
    for (int idx = lane; idx < n; idx += WAVE) rhs[idx] = m0[idx];
    __syncthreads();
    ok = banded_factor_forward(Bc, rhs, invd, ptab, n, bw);
    if (ok || tried >= a.attempts) break;
    shift += eps;       // fast_mvn.py:64-68 : cumulative eps, eps *= 10
    eps *= 10.0;
    ++tried;
    __syncthreads();
  }
  if (lane == 0) a.tries[j] = tried;
  if (!ok) {
    if (lane == 0 && atomicCAS(&a.status[0], 0, 1) == 0) a.status[1] = jg;
    return;
  }
  // rhs holds y = L^-1 mu ; add z (depth-major index), back-substitute
  for (int idx = lane; idx < n; idx += WAVE) {
    const double zz = a.z ? a.z[(size_t)jg * n + idx]
                          : philox_normal(a.seed, a.stream, (unsigned long long)jg * n + idx);
    rhs[idx] += zz;
  }
  __syncthreads();
  banded_backward(Bc, rhs, invd, n, bw);
  __syncthreads();
  for (int idx = lane; idx < n; idx += WAVE) a.V[(size_t)jg * n + idx] = rhs[idx];
}

// stand-alone banded sampler on a caller-supplied band (btf_mvn_banded)
struct MvnArgs {
  const double* band; const double* mu; const double* z; double* x; double* work;  // work: batch*(n*(bw+1)+2n)
  int n, bw; unsigned long long seed; double eps0; int attempts; int* tries; int* status;
};

static __global__ __launch_bounds__(WAVE) void mvn_banded_kernel(MvnArgs a) {
  extern __shared__ double lds[];
  unsigned short* ptab = reinterpret_cast<unsigned short*>(lds);
  const int lane = threadIdx.x, b = blockIdx.x;
  const int n = a.n, bw = a.bw, R1 = bw + 1;
  fill_pair_table(ptab, bw);
  __syncthreads();
  double* Bc = a.work + (size_t)b * ((size_t)n * R1 + 2 * n);
  double* rhs = Bc + (size_t)n * R1;
  double* invd = rhs + n;
  const double* src = a.band + (size_t)b * n * R1;
  double shift = 0.0, eps = a.eps0;
  int tried = 0;
  bool ok = false;
  while (true) {
    for (int idx = lane; idx < n * R1; idx += WAVE) Bc[idx] = src[idx] + ((idx % R1) == 0 ? shift : 0.0);
    for (int idx = lane; idx < n; idx += WAVE) rhs[idx] = a.mu ? a.mu[(size_t)b * n + idx] : 0.0;
    __syncthreads();
    ok = banded_factor_forward(Bc, rhs, invd, ptab, n, bw);
    if (ok || tried >= a.attempts) break;
    shift += eps;
    eps *= 10.0;
    ++tried;
    __syncthreads();
  }
  if (lane == 0) a.tries[b] = tried;
  if (!ok) {
    if (lane == 0 && atomicCAS(&a.status[0], 0, 1) == 0) a.status[1] = b;
    return;
  }
  for (int idx = lane; idx < n; idx += WAVE)
    rhs[idx] += a.z ? a.z[(size_t)b * n + idx] : philox_normal(a.seed, 7, (unsigned long long)b * n + idx);
  __syncthreads();
  banded_backward(Bc, rhs, invd, n, bw);
  __syncthreads();
  for (int idx = lane; idx < n; idx += WAVE) a.x[(size_t)b * n + idx] = rhs[idx];
}

// ============================================================================
// Dense Gaussian sampler: the sparse=False branches of fast_mvn (sample_mvn_from_precision fast_mvn.py:49-60,
// sample_mvn_from_covariance :126-142; dispatcher sample_mvn :145-179).  One workgroup per system, the n x n
// matrix in an HBM work area (small n: it lives in L2), vectors in LDS.
//   precision:  L L' = Q (or L given);  x = L^-T (z + L^-1 mu_part)  |  L^-T z + mu
//   covariance: L L' = S (or L given);  x = L (z + L' mu_part)       |  L z + mu         (S mu_part = L L' mu_part)
// Not positive definite: eps added to the diagonal cumulatively, x10 per retry (fast_mvn.py:62-68), at most
// `attempts` times, then the failure is reported (np.linalg.cholesky raises LinAlgError there).
// ============================================================================
constexpr int MVD_THREADS = 256;
enum { MVD_PRECISION = 1, MVD_FACTOR = 2 };
struct MvnDenseArgs {
  const double* A; const double* mu; const double* mu_part; const double* z; double* x; double* work;
  int n; int form; unsigned long long seed; double eps0; int attempts; int* tries; int* status;
};
static __global__ __launch_bounds__(MVD_THREADS) void mvn_dense_kernel(MvnDenseArgs a) {
  extern __shared__ double lds[];
  const int b = blockIdx.x, tid = threadIdx.x, n = a.n;
  const double* __restrict__ A = a.A + (size_t)b * n * n;
  double* __restrict__ L = a.work + (size_t)b * n * n;
  double* v = lds;              // [n] working vector
  double* u = lds + n;          // [n] second vector
  const bool prec = a.form & MVD_PRECISION, fact = a.form & MVD_FACTOR;
  double shift = 0.0, eps = a.eps0;
  int tried = 0;
  bool ok = true;
  while (true) {
    for (int idx = tid; idx < n * n; idx += MVD_THREADS) {
      const int r = idx / n, c = idx - r * n;
      L[idx] = c <= r ? A[idx] + ((r == c && !fact) ? shift : 0.0) : 0.0;
    }
    ok = true;
    if (!fact) {
      for (int c = 0; c < n; ++c) {           // right-looking Cholesky, lower
        __syncthreads();
        const double d = L[(size_t)c * n + c];
        if (!(d > 0.0)) { ok = false; break; }         // (uniform: everybody reads the same word after the barrier)
        const double l = sqrt(d), inv = 1.0 / l;
        __syncthreads();
        if (tid == 0) L[(size_t)c * n + c] = l;
        for (int r = c + 1 + tid; r < n; r += MVD_THREADS) L[(size_t)r * n + c] *= inv;
        __syncthreads();
        for (int r = c + 1 + (tid >> 5); r < n; r += MVD_THREADS / 32) {
          const double lr = L[(size_t)r * n + c];
          for (int q = c + 1 + (tid & 31); q <= r; q += 32) L[(size_t)r * n + q] = fma(-lr, L[(size_t)q * n + c], L[(size_t)r * n + q]);
        }
      }
    }
    __syncthreads();
    if (ok || tried >= a.attempts) break;
    shift += eps;
    eps *= 10.0;
    ++tried;
  }
  if (tid == 0 && a.tries) a.tries[b] = tried;
  if (!ok) {
    if (tid == 0 && atomicCAS(&a.status[0], 0, 1) == 0) a.status[1] = b;
    return;
  }
  // v = z (given, or Philox keyed by (seed, system, coordinate)); u = mu_part
  for (int i = tid; i < n; i += MVD_THREADS) {
    v[i] = a.z ? a.z[(size_t)b * n + i] : philox_normal(a.seed, 0x4d564eULL, (unsigned long long)b * n + i);
    u[i] = a.mu_part ? a.mu_part[(size_t)b * n + i] : 0.0;
  }
  __syncthreads();
  if (prec) {
    if (a.mu_part) {            // u <- L^-1 mu_part (forward substitution), v += u
      for (int c = 0; c < n; ++c) {
        const double yc = u[c] / L[(size_t)c * n + c];
        __syncthreads();
        if (tid == 0) u[c] = yc;
        for (int r = c + 1 + tid; r < n; r += MVD_THREADS) u[r] = fma(-L[(size_t)r * n + c], yc, u[r]);
        __syncthreads();
      }
      for (int i = tid; i < n; i += MVD_THREADS) v[i] += u[i];
      __syncthreads();
    }
    for (int c = n - 1; c >= 0; --c) {       // v <- L^-T v (back substitution with the columns of L')
      const double xc = v[c] / L[(size_t)c * n + c];
      __syncthreads();
      if (tid == 0) v[c] = xc;
      for (int r = tid; r < c; r += MVD_THREADS) v[r] = fma(-L[(size_t)c * n + r], xc, v[r]);
      __syncthreads();
    }
  } else {
    if (a.mu_part) {            // v += L' mu_part
      for (int i = tid; i < n; i += MVD_THREADS) {
        double s = 0.0;
        for (int r = i; r < n; ++r) s = fma(L[(size_t)r * n + i], u[r], s);
        v[i] += s;
      }
      __syncthreads();
    }
    for (int i = tid; i < n; i += MVD_THREADS) {       // u <- L v
      double s = 0.0;
      for (int c = 0; c <= i; ++c) s = fma(L[(size_t)i * n + c], v[c], s);
      u[i] = s;
    }
    __syncthreads();
    for (int i = tid; i < n; i += MVD_THREADS) v[i] = u[i];
    __syncthreads();
  }
  for (int i = tid; i < n; i += MVD_THREADS)
    a.x[(size_t)b * n + i] = v[i] + ((a.mu && !a.mu_part) ? a.mu[(size_t)b * n + i] : 0.0);
}

// ============================================================================
// one-time sufficient statistics  (BTF_K_STATS)
//   replaces the per-half-sweep nanmean / count of factor.py:329-330, :374-375
// ============================================================================
// Y slab [rows][cols][R] -> out[(transposed ? col*ld + row : row*ld + col)]
//   Gaussian : A = sum_r y (observed),  C = count            (A = S1, ybar = A/C)
//   Binomial : A = succ - trials/2,     C = trials  (0 where missing)
// block partial sums: ssw (within-cell sum of squares about the cell mean), nobs,
// and a flag whether every cell has all R replicates.
struct StatsArgs {
  const double* Y; const double* Y2;  // Y2 = trials (binomial) or nullptr
  int rows, cols, R; int ld; int transposed;
  double* A; double* C;               // C may be nullptr (not kept)
  double* bsum;                       // [gridDim][3] : ssw, nobs, sum S1^2/cnt   (may be nullptr)
  int* incomplete;                    // set to 1 if any cell has cnt != R (Gaussian) / is missing (binomial)
  int sum_rows;                       // rows >= sum_rows (a halo source row, btf_set_shard_halo) stay out of the block sums
};

static __global__ __launch_bounds__(256) void stats_kernel(StatsArgs a) {
  __shared__ double red[4][3];
  const size_t cells = (size_t)a.rows * a.cols;
  double ssw = 0.0, nobs = 0.0, sa2 = 0.0;
  bool inc = false;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < cells; idx += (size_t)gridDim.x * blockDim.x) {
    size_t row, col;
    if (a.transposed) {  // consecutive lanes -> consecutive rows (coalesced writes)
      col = idx / a.rows;
      row = idx - col * a.rows;
    } else {
      row = idx / a.cols;
      col = idx - row * a.cols;
    }
    const size_t cell = row * a.cols + col;
    double A, C;
    if (a.Y2 == nullptr) {
      double s1 = 0.0;
      int cnt = 0;
      for (int r = 0; r < a.R; ++r) {
        const double y = a.Y[cell * a.R + r];
        if (y == y) { s1 += y; ++cnt; }
      }
      const bool summed = (int)row < a.sum_rows;
      if (cnt > 0 && summed) {
        const double mean = s1 / cnt;
        for (int r = 0; r < a.R; ++r) {
          const double y = a.Y[cell * a.R + r];
          if (y == y) ssw = fma(y - mean, y - mean, ssw);
        }
      }
      if (summed) nobs += cnt;
      if (cnt > 0 && summed) sa2 = fma(s1, s1 / cnt, sa2);
      inc |= (cnt != a.R);
      A = s1;
      C = (double)cnt;
    } else {
      const double y = a.Y[cell], nt = a.Y2[cell];
      const bool miss = !(y == y) || !(nt == nt);
      A = miss ? 0.0 : y - 0.5 * nt;
      C = miss ? 0.0 : nt;
      inc |= miss;
      nobs += miss || (int)row >= a.sum_rows ? 0.0 : 1.0;
    }
    const size_t o = a.transposed ? col * a.ld + row : row * a.ld + col;
    a.A[o] = A;
    if (a.C) a.C[o] = C;
  }
  if (inc) *a.incomplete = 1;
  if (a.bsum) {
    ssw = wave_sum(ssw);
    nobs = wave_sum(nobs);
    sa2 = wave_sum(sa2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = ssw; red[wave][1] = nobs; red[wave][2] = sa2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      a.bsum[3 * blockIdx.x] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
      a.bsum[3 * blockIdx.x + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
      a.bsum[3 * blockIdx.x + 2] = red[0][2] + red[1][2] + red[2][2] + red[3][2];
    }
  }
}

// plain / transposing copy of a host-layout slab [rows][cols] into a padded device layout
static __global__ void relayout_kernel(const double* src, int rows, int cols, double* dst, int ld, int transposed, int zero_nan) {
  const size_t cells = (size_t)rows * cols;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < cells; idx += (size_t)gridDim.x * blockDim.x) {
    size_t row, col;
    if (transposed) { col = idx / rows; row = idx - col * rows; }
    else { row = idx / cols; col = idx - row * cols; }
    double v = src[row * cols + col];
    if (zero_nan && !(v == v)) v = 0.0;
    dst[transposed ? col * ld + row : row * ld + col] = v;
  }
}

// Binomial pseudo-data kappa = Y - N/2 as bytes: dst = 2 kappa when every value is an integer in -127..127
// (*bad is set otherwise and the f64 array stays in use)
static __global__ void kappa_to_i8_kernel(const double* __restrict__ src, signed char* __restrict__ dst, size_t n, int* __restrict__ bad) {
  bool b = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double h = 2.0 * src[i];
    b |= !(h == floor(h)) || h > 127.0 || h < -127.0;
    dst[i] = (signed char)h;
  }
  if (b) *bad = 1;
}

static __global__ void f64_to_u8_kernel(const double* __restrict__ src, unsigned char* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = (unsigned char)src[i];
}

static __global__ void mask_kernel(double* C, const double* B, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (!(B[i] > 0.0)) C[i] = 0.0;
}

// ============================================================================
// residual sum of squares for the nu2 update  (BTF_K_SSE)
//   sum_{cells} sum_r (y - mu)^2 = SSW + sum_cells (S1 - cnt*mu)^2 / cnt
//   uses the V-layout slab: lanes along (j,t), W[i][:] wave-uniform
// ============================================================================
constexpr int SSE_THREADS = 256;

template <int K, typename CT = double>
__global__ __launch_bounds__(SSE_THREADS) void sse_kernel(const double* __restrict__ A, const CT* __restrict__ C,
                                                          double Rconst, const double* __restrict__ W,
                                                          const double* __restrict__ V, int N, int ncell_cols,
                                                          int ld, int rows_per_block, size_t vcol0,
                                                          double* __restrict__ bsum) {
  __shared__ double red[SSE_THREADS / WAVE];
  const int col = blockIdx.x * SSE_THREADS + threadIdx.x;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, N);
  double acc = 0.0;
  if (col < ncell_cols) {
    double v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = V[(vcol0 + col) * K + k];
    for (int i = r0; i < r1; ++i) {
      const double* __restrict__ w = W + (size_t)i * K;
      double mu = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) mu = fma(w[k], v[k], mu);
      const double x = A[(size_t)i * ld + col];
      const double c = C ? (double)C[(size_t)i * ld + col] : Rconst;
      const double e = x - c * mu;
      if (c > 0.0) acc += e * e / c;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < SSE_THREADS / WAVE; ++w) s += red[w];
    bsum[blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
}

// ============================================================================
// Polya-Gamma draws  (BTF_K_PG)   omega ~ PG(b, psi),  psi = w_i . v_jt
//   replaces the pypolyagamma call of factor.py:459.  Integer b: sum of b exact
//   PG(1,psi) draws by Devroye's alternating-series method (Polson, Scott &
//   Windle 2013, sec. 4); fractional remainder by the truncated sum-of-gammas
//   representation; b >= PG_NORMAL_B by moment-matched normal.
//   (integer b: the flat exact kernels of btf_pg_exact.h, the default for Binomial trial counts)
//   RNG: Philox keyed by (seed, global cell index) - the draw of a cell does not
//   depend on layout, launch geometry or sharding.
// ============================================================================
constexpr int PG_SERIES_NT = 16;     // terms of the sum-of-gammas series drawn for b < 3, + 2|psi|/(2 pi); the rest enters through a normal
constexpr int PG_SERIES_NT_BIG = 2;  // ... for b >= 3 (+ 2|psi|/(2 pi)): the rest enters through a moment-matched gamma (pg_draw_series)
constexpr int PG_SERIES_NT_MAX = 96;
constexpr int PG_PRODUCT_B = 8;      // integer shapes up to here draw their Gamma(b) terms as -ln(U_1...U_b) in the series sampler

struct CellRng {
  uint64_t seed, cell, ctr;
  uint32_t b0, b1, b2, b3;   // one Philox block as four named words (an indexed array would live in scratch)
  int have;            // unread 64-bit halves of the current Philox block
  uint32_t w32;        // leftover 32-bit word of a half split by uniform32()
  bool has32;
  double spare;        // second Box-Muller variate
  bool has_spare;
  float sparef = 0.0f; // ... of the single-precision pair
  bool has_sparef = false;
  __device__ CellRng(uint64_t s, uint64_t c)
      : seed(s), cell(c), ctr(0), b0(0), b1(0), b2(0), b3(0), have(0), w32(0), has32(false), spare(0.0), has_spare(false) {}
  __device__ __forceinline__ void half(uint32_t& lo, uint32_t& hi) {
    if (have == 0) {
      uint32_t r[4];
      Philox::gen(seed, cell, ctr++, r);
      b0 = r[0]; b1 = r[1]; b2 = r[2]; b3 = r[3];
      have = 2;
    }
    --have;
    lo = have ? b2 : b0;
    hi = have ? b3 : b1;
  }
  __device__ __forceinline__ double uniform() {                 // 53-bit, never 0 or 1
    uint32_t lo, hi;
    half(lo, hi);
    return u01(lo, hi);
  }
  __device__ __forceinline__ double uniform32() {               // 32-bit resolution: accept/reject decisions only
    uint32_t w;
    if (has32) { w = w32; has32 = false; }
    else { uint32_t hi; half(w, hi); w32 = hi; has32 = true; }
    return ((double)w + 0.5) * (1.0 / 4294967296.0);
  }
  __device__ __forceinline__ double expo() { return -log(uniform()); }
  // Box-Muller evaluated with the f32 hardware transcendentals (v_log_f32 / v_sin_f32 / v_cos_f32 /
  // v_sqrt_f32): a standard normal VARIATE good to ~1e-6 relative, |z| <= 6.7 - for rejection
  // samplers whose output is a random draw anyway (the Polya-Gamma series); ~15 instructions
  // instead of ~300 for the f64 log / sqrt / sincospi.  One 64-bit half gives two variates.
  __device__ __forceinline__ double normal32() {
    if (has_spare) { has_spare = false; return spare; }
    uint32_t lo, hi;
    half(lo, hi);
    const float u1 = ((float)(lo >> 1) + 0.5f) * (1.0f / 2147483648.0f);      // (0,1), exact for the small values
    const float u2 = (float)(hi >> 8) * (1.0f / 16777216.0f);                 // [0,1) revolutions
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
    spare = (double)(r * __builtin_amdgcn_sinf(u2));
    has_spare = true;
    return (double)(r * __builtin_amdgcn_cosf(u2));
  }
  // the same pair kept in single precision (the flat Gamma loop of the Polya-Gamma series works in f32 throughout)
  __device__ __forceinline__ float normal32f() {
    if (has_sparef) { has_sparef = false; return sparef; }
    uint32_t lo, hi;
    half(lo, hi);
    const float u1 = ((float)(lo >> 1) + 0.5f) * (1.0f / 2147483648.0f);
    const float u2 = (float)(hi >> 8) * (1.0f / 16777216.0f);
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    sparef = r * __builtin_amdgcn_sinf(u2);
    has_sparef = true;
    return r * __builtin_amdgcn_cosf(u2);
  }
  __device__ __forceinline__ float uniform32f() {               // (0,1), 24-bit resolution: accept / reject only
    uint32_t w;
    if (has32) { w = w32; has32 = false; }
    else { uint32_t hi; half(w, hi); w32 = hi; has32 = true; }
    return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f);
  }
  __device__ __forceinline__ double normal() {                  // Box-Muller, both variates used
    if (has_spare) { has_spare = false; return spare; }
    const double u1 = uniform(), u2 = uniform();
    const double r = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    spare = r * sn;
    has_spare = true;
    return r * cs;
  }
};

// inverse-Gaussian(1/z, 1) truncated to (0, PG_T)
__device__ __forceinline__ double pg_rtigauss(double z, CellRng& g) {
  double X = PG_T + 1.0;
  if (1.0 / z > PG_T) {  // mu > t
    double alpha = 0.0;
    while (g.uniform() > alpha) {
      double E1 = g.expo(), E2 = g.expo();
      while (E1 * E1 > 2.0 * E2 / PG_T) { E1 = g.expo(); E2 = g.expo(); }
      X = 1.0 + E1 * PG_T;
      X = PG_T / (X * X);
      alpha = exp(-0.5 * z * z * X);
    }
  } else {
    const double mu = 1.0 / z;
    while (X > PG_T) {
      double Y = g.normal();
      Y *= Y;
      const double half_mu = 0.5 * mu, mu_Y = mu * Y;
      X = mu + half_mu * mu_Y - half_mu * sqrt(4.0 * mu_Y + mu_Y * mu_Y);
      if (g.uniform() > mu / (mu + X)) X = mu * mu / X;
    }
  }
  return X;
}

__device__ __forceinline__ double pg_one(double z /* = |psi|/2 */, double p_exp, CellRng& g) {
  const double fz = 0.125 * PG_PI * PG_PI + 0.5 * z * z;
  while (true) {
    const double X = (g.uniform32() < p_exp) ? PG_T + g.expo() / fz : pg_rtigauss(z, g);
    double S = pg_a(0, X);
    const double Y = g.uniform() * S;
    int n = 0;
    bool go = true;
    while (go) {
      ++n;
      if (n & 1) {
        S -= pg_a(n, X);
        if (Y <= S) return 0.25 * X;
      } else {
        S += pg_a(n, X);
        if (Y > S) go = false;
      }
      if (n > 1000) return 0.25 * X;  // unreachable in exact arithmetic; bounds the loop
    }
  }
}

__device__ inline double pg_mean_dev(double b, double c) {
  const double a = fabs(c);
  return a > 1e-6 ? b / (2.0 * a) * tanh(0.5 * a) : b * 0.25 * (1.0 - a * a / 12.0);
}
__device__ inline double pg_var_dev(double b, double c) {
  const double a = fabs(c);
  if (a < 1e-3) return b / 24.0 * (1.0 - a * a / 5.0);
  const double ch = cosh(0.5 * a);
  return a > 40.0 ? b / (2.0 * a * a * a) * (1.0 - 2.0 * a * exp(-a))
                  : b / (4.0 * a * a * a) * (sinh(a) - a) / (ch * ch);
}

// Gamma(shape,1), Marsaglia & Tsang (2000); shape < 1 by the U^(1/shape) boost.
// FAST: normals from CellRng::normal32 (f32 hardware transcendentals; see there).
template <bool FAST = false>
__device__ __forceinline__ double gamma_mt(double shape, CellRng& g) {
  double boost = 1.0;
  if (shape < 1.0) { boost = pow(g.uniform(), 1.0 / shape); shape += 1.0; }
  const double d = shape - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * d);
  while (true) {
    double x, v;
    do { x = FAST ? g.normal32() : g.normal(); v = 1.0 + cc * x; } while (v <= 0.0);
    v = v * v * v;
    const double u = g.uniform32();
    if (u < 1.0 - 0.0331 * x * x * x * x) return boost * d * v;
    if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return boost * d * v;
  }
}

// ---- the two samplers -------------------------------------------------------------------------------------
// PATH_SERIES (fast, approximate): sum-of-gammas series with a normal remainder, any 0 < b < PG_NORMAL_B.
// PATH_EXACT:  Devroye's alternating-series sampler (Polson, Scott & Windle 2013, the algorithm of
//              pypolyagamma's PyPolyaGamma.pgdrawv for integer b: factor.py:459) summed floor(b) times - exact;
//              a fractional part of b is added as a 128-term series of exact f64 Gamma(frac) draws with the mean of
//              the remainder (truncation: < 1e-6 of that part's variance).
// b >= PG_NORMAL_B: moment-matched normal in both (pypolyagamma switches to a normal approximation at b > 170).
enum { PG_PATH_SERIES = 0, PG_PATH_EXACT = 1 };

__device__ __forceinline__ double pg_draw_normal(double b, double psi, CellRng& g) {
  const double m = pg_mean_dev(b, psi), sd = sqrt(pg_var_dev(b, psi));
  double x;
  do { x = m + sd * g.normal(); } while (x <= 0.0);
  return x;
}

__device__ __forceinline__ double pg_draw_exact(double b, double psi, CellRng& g) {
  const double fl = floor(b);
  const double z = 0.5 * fabs(psi);
  const double p_exp = pg_mass_texpon(z);
  double sum = 0.0;
  for (int i = 0; i < (int)fl; ++i) sum += pg_one(z, p_exp, g);
  const double fr = b - fl;
  if (fr > 0.0) {
    const double c2 = psi * psi / (4.0 * PG_PI * PG_PI);
    const double sc = sqrt(c2);
    const int NT = 128 + (int)(2.0 * sc);
    double sfr = 0.0;
    for (int k = 1; k <= NT; ++k) sfr += gamma_mt<false>(fr, g) / ((k - 0.5) * (k - 0.5) + c2);
    const double tmean = sc > 1e-4 * NT ? atan(sc / NT) / sc : 1.0 / NT;      // int_NT^inf dx / (x^2 + c2)
    sum += (sfr + fr * tmean) / (2.0 * PG_PI * PG_PI);
  }
  return sum;
}

// The candidate loop of the series sampler draws ~10 words per cell; Philox4x32-10 costs twenty 32 x 32 -> 64
// multiplies (quarter rate) per four of them - two thirds of that loop.  xoshiro128++ (Blackman & Vigna), seeded
// per cell from ONE Philox block of the same (seed, cell) key, gives a word in a dozen full-rate integer
// instructions; the stream is still a function of (seed, global cell) alone, so layouts and shardings see the same draws.
struct FastRng {
  uint32_t s0, s1, s2, s3;
  float sparef;
  bool has_spare;
  __device__ FastRng(uint64_t seed, uint64_t cell) : sparef(0.0f), has_spare(false) {
    uint32_t r[4];
    Philox::gen(seed, cell, 0x78736f7368ULL << 16, r);      // a counter no CellRng reaches
    s0 = r[0]; s1 = r[1]; s2 = r[2]; s3 = r[3] | 1u;         // never the all-zero state
  }
  __device__ __forceinline__ uint32_t next() {
    const uint32_t sum = s0 + s3;
    const uint32_t result = ((sum << 7) | (sum >> 25)) + s0;
    const uint32_t t = s1 << 9;
    s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t;
    s3 = (s3 << 11) | (s3 >> 21);
    return result;
  }
  __device__ __forceinline__ float uniform32f() { return ((float)(next() >> 8) + 0.5f) * (1.0f / 16777216.0f); }
  __device__ __forceinline__ float normal32f() {            // f32 Box-Muller as CellRng::normal32f
    if (has_spare) { has_spare = false; return sparef; }
    const uint32_t lo = next(), hi = next();
    const float u1 = ((float)(lo >> 1) + 0.5f) * (1.0f / 2147483648.0f);
    const float u2 = (float)(hi >> 8) * (1.0f / 16777216.0f);
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    sparef = r * __builtin_amdgcn_sinf(u2);
    has_spare = true;
    return r * __builtin_amdgcn_cosf(u2);
  }
  __device__ __forceinline__ float normal32f_single() {     // one variate, nothing kept (the last draw of a cell)
    if (has_spare) { has_spare = false; return sparef; }
    const uint32_t lo = next(), hi = next();
    const float u1 = ((float)(lo >> 1) + 0.5f) * (1.0f / 2147483648.0f);
    const float u2 = (float)(hi >> 8) * (1.0f / 16777216.0f);
    return __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
  }
};

__device__ __forceinline__ double pg_draw_series(double b, double psi, CellRng& g) {
  // any other b: PG(b, psi) = 1/(2 pi^2) sum_k g_k / ((k-1/2)^2 + c2), g_k ~ Gamma(b, 1), c2 = psi^2/(4 pi^2).
  // The first NT terms are drawn; the remainder - a sum of many comparably small independent terms -
  // enters through a normal with its exact mean  b int_NT^inf dx/(x^2+c2)  and variance
  // b int_NT^inf dx/(x^2+c2)^2  (midpoint rule; share of the total variance 5e-6 at psi = 0, 6 % at |psi| = 60)
  const double sc = fabs(psi) * (1.0 / (2.0 * PG_PI));      // = sqrt(c2): no square root needed
  const double c2 = sc * sc;
  // the weights are flat up to k ~ sc: start the tail only where they decay (k^-2)
  const int NT = min(PG_SERIES_NT_MAX, (b >= 3.0 ? PG_SERIES_NT_BIG : PG_SERIES_NT) + (int)(2.0 * sc));
  double s = 0.0;
  FastRng f(g.seed, g.cell);         // (the candidate loop and the remainder's normal)
  if (b < 1.0) {
    for (int k = 1; k <= NT; ++k) s += gamma_mt<true>(b, g) / ((k - 0.5) * (k - 0.5) + c2);
  } else {
    // NT Marsaglia-Tsang Gamma(b) draws as ONE flat loop: every trip each lane tries one candidate and,
    // if it is accepted, adds it to its current term - a wave runs ~NT + 3 trips instead of NT rejection
    // loops that each last as long as the unluckiest lane.  Candidate normal and the acceptance test use
    // the f32 hardware log (threshold good to ~1e-6: the bias is far below the Monte-Carlo noise).
    // The candidate lives in f32 end to end (variates good to ~1e-7 relative); only the accepted term is widened
    // for the f64 sum.
    const double d = b - 1.0 / 3.0;
    const float df = (float)d, ccf = __builtin_amdgcn_rsqf(9.0f * df);
    // weights 1/((k-1/2)^2 + c2) in single precision too
    const float c2f = (float)c2;
    auto weight1 = [&](int kk) -> float {
      const float q = fmaf((float)kk - 0.5f, (float)kk - 0.5f, c2f);
      return __builtin_amdgcn_rcpf(q);                      // (v_rcp_f32: 1 ulp, as good as the variates)
    };
    // Integer shapes up to PG_PRODUCT_B (the Binomial case: b = number of trials): Gamma(b) = -ln(U_1 ... U_b), no
    // rejection, so every lane finishes a term per trip - b uniforms and one hardware log per four of them instead
    // of ~1.6 Marsaglia-Tsang trips of a normal, a uniform and two logs (and no trips spent waiting for the
    // unluckiest lane).  24-bit uniforms: a product of four stays a normal f32 (>= 2^-100).
    const bool prod = b <= (double)PG_PRODUCT_B && b == floor(b);
    if (prod) {
      const int bi = (int)b;
      float sf = 0.0f;                                      // (<= 96 positive f32 terms: ~1e-7 relative, as each term)
      for (int kk = 1; kk <= NT; ++kk) {
        // four uniforms per log, branch-free: the ones past b are replaced by 1 (their words are still consumed)
        auto group = [&](int i0) -> float {
          // (four 24-bit uniforms from three words: the fourth takes the low bytes the other three leave unused)
          const uint32_t w0 = f.next(), w1 = f.next(), w2 = f.next();
          const uint32_t w3 = ((w0 & 0xffu) << 16) | ((w1 & 0xffu) << 8) | (w2 & 0xffu);
          constexpr float S24 = 1.0f / 16777216.0f;
          const float u0 = ((float)(w0 >> 8) + 0.5f) * S24, u1 = ((float)(w1 >> 8) + 0.5f) * S24;
          const float u2 = ((float)(w2 >> 8) + 0.5f) * S24, u3 = ((float)w3 + 0.5f) * S24;
          float p = u0;
          p *= i0 + 1 < bi ? u1 : 1.0f;
          p *= i0 + 2 < bi ? u2 : 1.0f;
          p *= i0 + 3 < bi ? u3 : 1.0f;
          return __builtin_amdgcn_logf(p);
        };
        float l2 = group(0);
        if (bi > 4) l2 += group(4);
        sf = fmaf(-0.69314718f * l2, weight1(kk), sf);
      }
      s = (double)sf;
    }
    int k = prod ? NT + 1 : 1;
    float wk = df * weight1(1);                             // (d folded in: the term is d v^3 w_k)
    while (k <= NT) {
      const float xf = f.normal32f();
      const float v1 = fmaf(ccf, xf, 1.0f);
      const float vf = v1 * v1 * v1;
      const float lnu = 0.69314718f * __builtin_amdgcn_logf(f.uniform32f());
      const float rhs = 0.5f * xf * xf + df * (1.0f - vf + 0.69314718f * __builtin_amdgcn_logf(vf));
      if (v1 > 0.0f && lnu < rhs) {
        s += (double)(vf * wk);
        ++k;
        wk = df * weight1(k);
      }
    }
  }
  // Remainder in single precision: it carries a few per cent of the mean and 3e-4 of the variance, so f32's 1e-7
  // is far inside the Monte-Carlo noise - and the f64 atan / sin / Box-Muller it replaces were ~700 of the ~1000
  // instructions of a draw.  u = sc / N0: int_N0^inf dx/(x^2+c2) = atan(u)/sc, int dx/(x^2+c2)^2 =
  // (atan u - u/(1+u^2)) / (2 sc^3); small u by series (the closed forms cancel there).
  // b >= 3 draws only PG_SERIES_NT_BIG (+ 2|psi|/2pi) terms: the weights of the terms between NT and N0 = 4 are summed
  // explicitly (from there the integrals + first Euler-Maclaurin term give the tail mean to 3e-5 and its variance to
  // 1.5e-3 of themselves: 5e-6 and 3e-7 of the whole sum's).
  const float scf = (float)sc, c2r = scf * scf;
  const bool big = b >= 3.0;
  const int N0 = big ? max(NT, 4) : NT;
  float e1 = 0.0f, e2 = 0.0f;
  for (int kk = NT + 1; kk <= N0; ++kk) {
    const float w = __builtin_amdgcn_rcpf(fmaf((float)kk - 0.5f, (float)kk - 0.5f, c2r));
    e1 += w;
    e2 = fmaf(w, w, e2);
  }
  // (reciprocals by v_rcp_f32, 1 ulp: the IEEE f32 divisions here were ~70 instructions of the draw)
  const float NTf = (float)N0, rN = __builtin_amdgcn_rcpf(NTf), u = scf * rN, u2 = u * u;
  float tmean, tvar;
  if (u < 0.3f) {
    tmean = (1.0f - u2 * (1.0f / 3.0f - u2 * (0.2f - u2 * (1.0f / 7.0f)))) * rN;
    tvar = (1.0f - u2 * (1.2f - u2 * (9.0f / 7.0f - u2 * (4.0f / 3.0f)))) * (1.0f / 3.0f) * (rN * rN * rN);
  } else {
    const float phi = atanf(u), rs = __builtin_amdgcn_rcpf(scf);
    tmean = phi * rs;
    tvar = (phi - u * __builtin_amdgcn_rcpf(1.0f + u2)) * 0.5f * (rs * rs * rs);
  }
  {   // first Euler-Maclaurin term of the midpoint sums: sum_{k>N0} f(k-1/2) = int_N0^inf f dx + f'(N0)/24 + ...
    const float rq = __builtin_amdgcn_rcpf(fmaf(NTf, NTf, c2r)), g = NTf * rq * rq;
    tmean -= g * (1.0f / 12.0f);
    tvar -= g * rq * (1.0f / 6.0f);
  }
  tmean += e1;
  tvar += e2;
  const float bf = (float)b;
  float rem;
  if (big) {
    // sum_{k>NT} w_k g_k, g_k ~ Gamma(b): a gamma variate with its mean b S1 and variance b S2 (shape alpha =
    // b S1^2/S2 >= 20, scale S2/S1), drawn by the Wilson-Hilferty cube d (1 + x/sqrt(9d))^3, d = alpha - 1/3 (the
    // Marsaglia-Tsang proposal without the rejection step: mean exact, variance +0.7 %, skewness as the gamma's).
    // At NT = 2 the tail holds 1.6e-3 of the variance and 8e-5 of the third cumulant of the whole sum.
    const float th = tvar * __builtin_amdgcn_rcpf(tmean), d = bf * tmean * tmean * __builtin_amdgcn_rcpf(tvar) - (1.0f / 3.0f);
    const float v1 = fmaxf(fmaf(f.normal32f_single(), __builtin_amdgcn_rsqf(9.0f * d), 1.0f), 0.0f);
    rem = th * d * (v1 * v1 * v1);
  } else {
    rem = bf * tmean + __builtin_amdgcn_sqrtf(bf * tvar) * f.normal32f_single();
  }
  const double x = s + (double)rem;
  return fmax(x, 1e-300) * (1.0 / (2.0 * PG_PI * PG_PI));      // (a multiplication: the IEEE f64 division is a dozen instructions)
}

// Which sampler a cell goes through: pg_class_of (btf_pg_exact.h).  Integer counts go to the flat exact kernels
// (pgx_*; every integer count up to PG_AUTO_EXACT_MAX by default, every integer count in the exact mode); the kernels
// below take the rest: PG_PATH_SERIES the sum-of-gammas series (non-integer counts - Negative-Binomial pseudo-trial
// counts sum(y) + n r - and, by default, integer counts above PG_AUTO_EXACT_MAX; every cell in PG_MODE_SERIES_ALL),
// PG_PATH_EXACT the f64 Devroye sampler with a series for the fractional part (non-integer counts of the exact mode).
__device__ __forceinline__ int pg_path_of_class(int cls) {
  return cls == PG_CLASS_SERIES ? PG_PATH_SERIES : cls == PG_CLASS_FRAC ? PG_PATH_EXACT : -1;
}

// one-call form (validation entry point, small problems): per-lane choice of the path.
// mode: PG_MODE_* ; PG_MODE_REF_F64 (3): the f64 Devroye sampler for everything below the normal range (the
// round-2 exact kernel, kept as the reference the flat f32-squeeze sampler is tested against).
constexpr int PG_MODE_REF_F64 = 3;
__device__ __forceinline__ double pg_draw(double b, double psi, CellRng& g, int mode = 0) {
  if (!(b > 0.0)) return 0.0;
  if (b >= PG_NORMAL_B) return pg_draw_normal(b, psi, g);
  if (mode == PG_MODE_REF_F64) return pg_draw_exact(b, psi, g);
  const int cls = pg_class_of(b, mode);
  return cls == PG_CLASS_SERIES ? pg_draw_series(b, psi, g) : pg_draw_exact(b, psi, g);
}


constexpr int PG_THREADS = 256;

// out[r][l] = PG(B[r][l], L[l0+l] . U[r]);  global cell id = base + r*stride_r + l*stride_l
// A launch handles the cells of ONE path (PATH) - the other path's cells are left to its own launch (the host
// skips a launch no cell needs): each kernel carries one sampler's registers, not both.  `fill`: this launch also
// writes the zeros of the cells without an observation and the normal-range draws (b >= PG_NORMAL_B).
template <int PATH>
__device__ __forceinline__ bool pg_cell(double b, double psi, int exact_mode, bool fill, CellRng& g, double& om) {
  if (!(b > 0.0)) { om = 0.0; return fill; }
  if (b >= PG_NORMAL_B) { if (fill) om = pg_draw_normal(b, psi, g); return fill; }
  if (pg_path_of_class(pg_class_of(b, exact_mode)) != PATH) return false;
  om = PATH == PG_PATH_EXACT ? pg_draw_exact(b, psi, g) : pg_draw_series(b, psi, g);
  return true;
}

template <int K, int PATH>
__global__ __launch_bounds__(PG_THREADS, 2) void pg_kernel(const double* __restrict__ B, double* __restrict__ out,
                                                        const double* __restrict__ Lf, const double* __restrict__ Uf,
                                                        int nl, int ld, int Rdim, int rows_per_block,
                                                        unsigned long long base, unsigned long long stride_r,
                                                        unsigned long long stride_l, unsigned long long seed,
                                                        int exact_mode, int fill) {
  const int l = blockIdx.x * PG_THREADS + threadIdx.x;
  if (l >= nl) return;
  double f[K];
#pragma unroll
  for (int k = 0; k < K; ++k) f[k] = Lf[(size_t)l * K + k];
  const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, Rdim);
  for (int r = r0; r < r1; ++r) {
    const double* __restrict__ u = Uf + (size_t)r * K;
    double psi = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) psi = fma(f[k], u[k], psi);
    const double b = B[(size_t)r * ld + l];
    CellRng g(seed, base + (unsigned long long)r * stride_r + (unsigned long long)l * stride_l);
    double om;
    if (pg_cell<PATH>(b, psi, exact_mode, fill != 0, g, om)) out[(size_t)r * ld + l] = om;
  }
}

// One-pass variant for an unsharded context: every cell is drawn once (lanes along (j,t)),
// written to the V-layout array directly and to the W-layout array through an LDS tile
// transpose.  Same (seed, cell) streams as pg_kernel, hence identical draws.
template <int K, int PATH>
__global__ __launch_bounds__(256, 2) void pg_tile_kernel(const double* __restrict__ Bv, double* __restrict__ Cv,
                                                     double* __restrict__ CwT, const double* __restrict__ W,
                                                     const double* __restrict__ V, int N, int MT, int ldv, int ldw,
                                                     unsigned long long seed, int exact_mode, int fill) {
  __shared__ double tile[64][65];
  const int col = threadIdx.x & 63;
  const int rgrp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int jt0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
  const int jt = jt0 + col;
  const bool vc = jt < MT;
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = vc ? V[(size_t)jt * K + k] : 0.0;
  for (int q = 0; q < 16; ++q) {
    const int r = rgrp + 4 * q, i = i0 + r;               // wave-uniform
    double om = -1.0;                                      // < 0: not this launch's cell
    if (i < N && vc) {
      const double* __restrict__ w = W + (size_t)i * K;
      double psi = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) psi = fma(w[k], v[k], psi);
      const double b = Bv[(size_t)i * ldv + jt];
      CellRng g(seed, (unsigned long long)i * MT + jt);
      double x;
      if (pg_cell<PATH>(b, psi, exact_mode, fill != 0, g, x)) {
        om = x;
        Cv[(size_t)i * ldv + jt] = om;
      }
    }
    tile[r][col] = om;
  }
  __syncthreads();
  const int i = i0 + col;                                  // now lanes run along the rows
  for (int q = 0; q < 16; ++q) {
    const int c = rgrp + 4 * q, jj = jt0 + c;
    const double om = tile[col][c];
    if (jj < MT && i < N && om >= 0.0) CwT[(size_t)jj * ldw + i] = om;
  }
}

// stand-alone batch of PG draws (validation entry point)
// (skip_flat: the elements of PG_CLASS_FLAT are left to pgx_batch_kernel)
static __global__ void pg_batch_kernel(const double* b, const double* psi, double* out, long long n, unsigned long long seed,
                                int mode, int skip_flat) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (skip_flat && pg_class_of(b[i], mode) == PG_CLASS_FLAT) return;
  CellRng g(seed, (unsigned long long)i);
  out[i] = pg_draw(b[i], psi[i], g, mode);
}

// ============================================================================
// Horseshoe+ local scales on the device (rng="device")
//   factor.py:134-141 for every (column j, penalty row r), elementwise:
//     rate = sum_k (Delta V_j)[r,k]^2 / (2 lam2) + 1/clip(c);  Tau2 = 1/Gamma((K+1)/2, 1/clip(rate))
//     c = 1/Gamma(1, 1/clip(1/Tau2 + 1/b)); b = 1/Gamma(1, 1/clip(1/c + 1/a)); a = 1/Gamma(1, 1/clip(1/b + 1))
//   plus the per-column sums  lsum[j] = sum_r dsq[j,r] / Tau2_new[j,r]  that the lam2 update needs
//   (factor.py:148-150).  One workgroup per column; Philox keyed by (seed, j*nD + r).
// ============================================================================
// one column: 256 threads (tid256), red4: four doubles of LDS of this column's slot; every thread of the WORKGROUP
// reaches the barrier inside (columns past the end do no work)
__device__ void tau2_column(const TauSide& t, int K, int j, int tid256, double* red4) {
  double acc = 0.0;
  if (j < t.M) {
    const double lam2 = t.hyp ? t.hyp[HYP_LAM2] : t.lam2;
    const double lo = t.lo, hi = t.hi;
    const int nD = t.nD;
    const double* Vj = t.V + (size_t)j * t.T * K;
    for (int r = tid256; r < nD; r += 256) {
      const size_t o = (size_t)j * nD + r;
      const double tc0 = t.Tc[o], tb0 = t.Tb[o], ta0 = t.Ta[o];          // (in flight with the stencil gathers)
      // (Delta V_j)[r, :]: the stencil row once (<= tf+2 entries), then its rows of V_j - two dependent round trips,
      // not two per embedding dimension
      const int e0 = t.dr_ptr[r], e1 = t.dr_ptr[r + 1];
      double dv[EIG_MAXK];
#pragma unroll
      for (int k = 0; k < EIG_MAXK; ++k) dv[k] = 0.0;
      for (int e = e0; e < e1; ++e) {
        const double cv = t.dr_val[e];
        const double* __restrict__ vr = Vj + (size_t)t.dr_col[e] * K;
#pragma unroll
        for (int k = 0; k < EIG_MAXK; ++k) if (k < K) dv[k] = fma(cv, vr[k], dv[k]);
      }
      double dsq = 0.0;
#pragma unroll
      for (int k = 0; k < EIG_MAXK; ++k) dsq = fma(dv[k], dv[k], dsq);
      CellRng g(t.seed, (unsigned long long)o);
      auto clip = [&](double x) { return fmin(fmax(x, lo), hi); };
      const double rate = dsq / (2.0 * lam2) + 1.0 / clip(tc0);
      const double tau = clip(rate) / gamma_mt(0.5 * (K + 1), g);
      const double c = clip(1.0 / tau + 1.0 / tb0) / g.expo();
      const double b = clip(1.0 / c + 1.0 / ta0) / g.expo();
      const double a = clip(1.0 / b + 1.0) / g.expo();
      t.Tau2[o] = tau; t.Tc[o] = c; t.Tb[o] = b; t.Ta[o] = a;
      acc += dsq / tau;
    }
  }
  acc = wave_sum(acc);
  if ((tid256 & 63) == 0) red4[tid256 >> 6] = acc;
  __syncthreads();
  if (tid256 == 0 && j < t.M) t.lsum[j] = red4[0] + red4[1] + red4[2] + red4[3];
}
static __global__ __launch_bounds__(256) void tau2_kernel(TauSide t, int K) {
  __shared__ double red[4];
  tau2_column(t, K, blockIdx.x, threadIdx.x, red);
}

// ============================================================================
// Scalar hyper-parameters on the device (rng="device")
//   nu2    | rest : 1/Gamma(a + n/2, scale 1/(b + SSE/2))               factor.py:411-416, genlasso.py:160-164
//   sigma2 | rest : 1/Gamma(a + nW/2, scale 1/(b + sum W_free^2/2))      factor.py:130-132
//   lam2   | rest : max(1e-5, 1/Gamma(shape/2, scale 1/rate)), then lam2_a = 1/Gamma(1, 1/(1/lam2 + 1))
//                   rate = last column's term (compat="reference", quirk Q3) or 1/lam2_a + sum   factor.py:143-153
//   One workgroup; the reductions are fixed-order (thread-strided partial sums, then a serial
//   sum over the 256 lanes' partials), the draws are made by thread 0 from Philox streams
//   keyed (seed, HYP_* id).  Results land in hyp[] where the half-sweep kernels read them.
// ============================================================================
__device__ inline double block_sum_fixed(double x, double* red) {   // all 256 threads; deterministic
  x = wave_sum(x);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  const double s = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return s;
}

__device__ inline void lam2_draw(const double* __restrict__ lsum, int M, double shape, int exact,
                                 unsigned long long seed, double* __restrict__ hyp, double* red);

// which: bit 0 = nu2 (needs the SSE block partials), bit 1 = sigma2
// phase 0: reduce and draw (one GPU).  Sharded runs split the nu2 part around an all-reduce of the rank-local sum:
// phase 1 = reduce only, hyp[HYP_SSE] <- this rank's residual sum of squares; phase 2 = draw from hyp[HYP_SSE]
// as it stands (the collective has summed it over the ranks in between; every rank then draws the same value).
static __global__ __launch_bounds__(256) void scalars_kernel(const double* __restrict__ bsum, int nb, double ssw, double nobs,
                                                      const double* __restrict__ W, int N, int K, double nfree,
                                                      double nu2_a, double nu2_b, double sig_a, double sig_b, int which,
                                                      unsigned long long seed, double* __restrict__ hyp, int phase,
                                                      const double* __restrict__ lsum, int M, double lam_shape, int lam_exact,
                                                      unsigned long long lam_seed) {
  __shared__ double red[4];
  __shared__ double stat[2];
  if (blockIdx.x == 1) {       // a queued lam2 | rest draw rides in this launch (btf_queue_lam2): lam2_kernel's body
    lam2_draw(lsum, M, lam_shape, lam_exact, lam_seed, hyp, red);
    return;
  }
  // both reductions first (all threads), then the two draws side by side: thread 0 draws nu2 while thread 64
  // (another wave) draws sigma2 - a Gamma draw with shape ~1e7 is ~2 us of dependent f64 work on one lane
  if ((which & 1) && phase != 2) {
    double acc = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) acc += bsum[b];
    const double sse = block_sum_fixed(acc, red) + ssw;
    if (threadIdx.x == 0) stat[0] = sse;
  }
  if (phase == 1) {
    __syncthreads();
    if ((which & 1) && threadIdx.x == 0) hyp[HYP_SSE] = stat[0];
    return;
  }
  if ((which & 1) && phase == 2 && threadIdx.x == 0) stat[0] = hyp[HYP_SSE];
  if (which & 2) {
    double acc = 0.0;
    for (int e = threadIdx.x; e < N * K; e += 256) { const double w = W[e]; acc = fma(w, w, acc); }   // the structural zeros add 0
    const double wsq = block_sum_fixed(acc, red);
    if (threadIdx.x == 0) stat[1] = wsq;
  }
  __syncthreads();
  if ((which & 1) && threadIdx.x == 0) {
    const double sse = stat[0];
    CellRng g(seed, (unsigned long long)HYP_NU2);
    hyp[HYP_SSE] = sse;
    hyp[HYP_NU2] = (nu2_b + 0.5 * sse) / gamma_mt(nu2_a + 0.5 * nobs, g);
  }
  if ((which & 2) && threadIdx.x == 64) {
    const double wsq = stat[1];
    CellRng g(seed, (unsigned long long)HYP_SIGMA2);
    hyp[HYP_WSQ] = wsq;
    hyp[HYP_SIGMA2] = (sig_b + 0.5 * wsq) / gamma_mt(sig_a + 0.5 * nfree, g);
  }
}

__device__ inline void lam2_draw(const double* __restrict__ lsum, int M, double shape, int exact,
                                 unsigned long long seed, double* __restrict__ hyp, double* red) {
  double rate;
  if (exact) {
    double acc = 0.0;
    for (int j = threadIdx.x; j < M; j += 256) acc += lsum[j];
    rate = 1.0 / hyp[HYP_LAM2A] + 0.5 * block_sum_fixed(acc, red);
  } else {
    rate = 0.5 * lsum[M - 1];
  }
  if (threadIdx.x == 0) {
    CellRng g(seed, (unsigned long long)HYP_LAM2);
    const double lam2 = fmax(1e-5, rate / gamma_mt(0.5 * shape, g));
    hyp[HYP_LAM2] = lam2;
    hyp[HYP_LAM2A] = (1.0 / lam2 + 1.0) / g.expo();
  }
}
static __global__ __launch_bounds__(256) void lam2_kernel(const double* __restrict__ lsum, int M, double shape, int exact,
                                                   unsigned long long seed, double* __restrict__ hyp) {
  __shared__ double red[4];
  lam2_draw(lsum, M, shape, exact, seed, hyp, red);
}

// the same draws as side workgroups of the accumulation launches (NW waves; fixed-order reductions: lane-strided partial
// sums, wave butterflies, the waves' sums in order)
template <int NW>
__device__ inline double side_block_sum(double x, double* red) {
  x = wave_sum(x);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w) s += red[w];
  __syncthreads();
  return s;
}
template <int NW>
__device__ void sweep_scalar_side(const ScalarSide& sc, double* red) {
  double sse = 0.0, wsq = 0.0;
  if (sc.which & 1) {
    double acc = 0.0;
    for (int j = threadIdx.x; j < sc.M; j += NW * WAVE) acc += sc.sse_cols[j];
    sse = side_block_sum<NW>(acc, red) + sc.sconst;
  }
  if (sc.which & 2) {
    double acc = 0.0;
    for (int e = threadIdx.x; e < sc.NK; e += NW * WAVE) { const double w = sc.W[e]; acc = fma(w, w, acc); }
    wsq = side_block_sum<NW>(acc, red);
  }
  if ((sc.which & 1) && threadIdx.x == 0) {
    CellRng g(sc.seed, (unsigned long long)HYP_NU2);
    sc.hyp[HYP_SSE] = sse;
    const double nu2 = (sc.nu2_b + 0.5 * sse) / gamma_mt(sc.nu2_a + 0.5 * sc.nobs, g);
    sc.hyp[HYP_NU2] = nu2;
    if (sc.flag) store_sc1(sc.pub + HYP_NU2, nu2);
  }
  if ((sc.which & 2) && threadIdx.x == 64) {
    CellRng g(sc.seed, (unsigned long long)HYP_SIGMA2);
    sc.hyp[HYP_WSQ] = wsq;
    const double sg2 = (sc.sig_b + 0.5 * wsq) / gamma_mt(sc.sig_a + 0.5 * sc.nfree, g);
    sc.hyp[HYP_SIGMA2] = sg2;
    if (sc.flag) store_sc1(sc.pub + HYP_SIGMA2, sg2);
  }
  if (sc.flag) {                         // two storing waves: drained, barrier, one flag store
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) publish_epoch(sc.flag, sc.epoch);
  }
}
template <int NW>
__device__ void sweep_lam_side(const LamSide& lm, double* red) {
  double rate;
  if (lm.exact) {
    double acc = 0.0;
    for (int j = threadIdx.x; j < lm.M; j += NW * WAVE) acc += lm.lsum[j];
    rate = 1.0 / lm.hyp[HYP_LAM2A] + 0.5 * side_block_sum<NW>(acc, red);
  } else {
    rate = 0.5 * lm.lsum[lm.M - 1];
  }
  if (threadIdx.x == 0) {
    CellRng g(lm.seed, (unsigned long long)HYP_LAM2);
    const double lam2 = fmax(1e-5, rate / gamma_mt(0.5 * lm.shape, g));
    lm.hyp[HYP_LAM2] = lam2;
    lm.hyp[HYP_LAM2A] = (1.0 / lam2 + 1.0) / g.expo();
    if (lm.flag) {                       // one storing lane: drained, then the flag
      store_sc1(lm.pub + HYP_LAM2, lam2);
      drain_stores();
      publish_epoch(lm.flag, lm.epoch);
    }
  }
}

// ============================================================================
// Negative-Binomial rate update  (factor.py:513-554)
//   counts y_r ~ NB(R, p), p = ilogit(clip(w_i.v_jt, -10, 10)).  One random-walk MH step needs, for
//   every R element, the log-likelihood ratio candidate/current summed over the replicates
//   and over the dims R is shared across:
//     sum_r [ lgamma(y+c) - lgamma(c) - lgamma(y+R) + lgamma(R) + (c-R) log(1-p) ]   (NaN y: term dropped)
//   nb_loglik_kernel: lanes along (j,t), one row i per blockIdx.y; small integer counts use
//     lgamma(y+c) - lgamma(c) = sum_{k<y} log(c+k) as ONE log of a ratio of products.
//     shared_jt != 0: fixed-order block sum -> out[i][blockIdx.x]; else per cell out[i][jt].
//   nb_reduce_kernel: one workgroup per R element sums its slice of a (d0,d1,d2) array in a
//     fixed order (deterministic).
//   nb_trials_kernel: Binomial pseudo-data of the augmented model in both device layouts:
//     trials N = S + cnt R, A = S - N/2 (S = sum of observed counts, cnt = number observed).
// ============================================================================
constexpr int NB_FAST_MAX = 32;    // integer counts up to here: product form when R varies inside a block
constexpr int NB_TAB = 1024;       // integer counts below this: LDS table when R is constant inside a block

// lgamma(x) for x >= NB_STIRLING_MIN by Stirling's series (truncation error < 1/(1188 x^9) = 4e-17 at x = 33)
constexpr double NB_STIRLING_MIN = 33.0;
__device__ __forceinline__ double lgamma_big(double x) {
  const double ix = 1.0 / x, ix2 = ix * ix;
  const double ser = ix * (8.3333333333333333e-2 + ix2 * (-2.7777777777777778e-3 + ix2 * (7.9365079365079365e-4 + ix2 * -5.9523809523809524e-4)));
  return (x - 0.5) * log(x) - x + 0.91893853320467274178 + ser;
}

// lgamma(a) - lgamma(b) for a, b > 0 without libm's lgamma (which costs ~200 registers): both arguments
// are shifted up by the same n until Stirling applies,
//   lgamma(a) - lgamma(b) = lgamma_big(a+n) - lgamma_big(b+n) - log( prod_{k<n}(a+k) / prod_{k<n}(b+k) ).
__device__ inline double lgamma_diff(double a, double b) {
  const double lo = fmin(a, b);
  if (lo >= NB_STIRLING_MIN) return lgamma_big(a) - lgamma_big(b);
  const int n = (int)ceil(NB_STIRLING_MIN - lo);                                   // <= 33
  double pa = 1.0, pb = 1.0;
  for (int k = 0; k < n; ++k) { pa *= a + k; pb *= b + k; }                        // < 66^33: no overflow
  return lgamma_big(a + n) - lgamma_big(b + n) - log(pa / pb);
}

// count part of one replicate's term: lgamma(y+c) - lgamma(c) - lgamma(y+r) + lgamma(r);
// base = lgamma(r) - lgamma(c) (NaN: not computed yet - filled on first use)
__device__ inline double nb_count_part(double y, double r, double c, double& base) {
  if (y >= 0.0 && y <= (double)NB_FAST_MAX && y == floor(y)) {   // = log prod_{k<y} (c+k)/(r+k)
    double num = 1.0, den = 1.0;
    const int n = (int)y;
    for (int k = 0; k < n; ++k) { num *= c + k; den *= r + k; }
    return log(num / den);
  }
  if (!(base == base)) base = lgamma_diff(r, c);
  return lgamma_diff(y + c, y + r) + base;
}

// table of sum_{k<y} log((c+k)/(r+k)), y = 0..NB_TAB-1, by all 256 threads of a block (4 entries each)
// (ymax: entries above ymax + 1 are not needed by the caller - their logarithms are skipped)
__device__ inline void nb_build_table(double r, double c, double* tab, double* wsum, int ymax = NB_TAB) {
  static_assert(NB_TAB == 1024, "4 table entries per thread of a 256-thread block");
  const int k0 = 4 * threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double l[4], run = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) { if (k0 + q < ymax) run += log((c + (k0 + q)) / (r + (k0 + q))); l[q] = run; }
  double incl = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double up = __shfl_up(incl, d, WAVE);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  double off = incl - run;                       // exclusive prefix inside the wave
  for (int u = 0; u < wv; ++u) off += wsum[u];
  if (threadIdx.x == 0) tab[0] = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) if (k0 + q + 1 < NB_TAB) tab[k0 + q + 1] = off + l[q];
  __syncthreads();
}

// `row_rate` != 0: R does not vary along (j,t) (sr1 = sr2 = 0), so the count part of the term is a
// function of y alone inside a block: tabulated once in LDS for integer y < NB_TAB (a prefix sum of
// log((c+k)/(r+k))); everything else goes through nb_count_part.
// RR > 0: number of replicates known at compile time (all loads of a cell in flight together).
template <int K, int RR>
__global__ __launch_bounds__(256) void nb_loglik_kernel(const double* __restrict__ data, int Rr,
                                                       const double* __restrict__ W, const double* __restrict__ V,
                                                       int MT, int T, const double* __restrict__ Rv,
                                                       const double* __restrict__ Cv, long long sr0, long long sr1,
                                                       long long sr2, int shared_jt, double* __restrict__ out) {
  __shared__ double red[4];
  __shared__ double tab[NB_TAB + 1];     // tab[y] = sum_{k<y} log((c+k)/(r+k)), y < NB_TAB; tab[NB_TAB] = lgamma(r) - lgamma(c)
  __shared__ double wsum[4];
  const int i = blockIdx.y;
  const bool row_rate = sr1 == 0 && sr2 == 0;
  if (row_rate) {
    const double r = Rv[(long long)i * sr0], c = Cv[(long long)i * sr0];
    if (threadIdx.x == 0) tab[NB_TAB] = lgamma_diff(r, c);
    nb_build_table(r, c, tab, wsum);
  }
  // grid-stride over the (j,t) cells of row i: the table is built once per block, not once per cell
  double bterm = 0.0;
  const double* __restrict__ w = W + (size_t)i * K;
  const int nrep = RR > 0 ? RR : Rr;
  for (int jt = blockIdx.x * 256 + threadIdx.x; jt < MT; jt += gridDim.x * 256) {
    const double* __restrict__ y = data + ((size_t)i * MT + jt) * nrep;
    double yv[RR > 0 ? RR : 1];
    if constexpr (RR > 0) {
#pragma unroll
      for (int q = 0; q < RR; ++q) yv[q] = y[q];
    }
    const int j = jt / T, t = jt - j * T;
    const long long ri = (long long)i * sr0 + (long long)j * sr1 + (long long)t * sr2;
    const double r = Rv[ri], c = Cv[ri];
    double psi = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) psi = fma(w[k], V[(size_t)jt * K + k], psi);
    psi = fmin(fmax(psi, -10.0), 10.0);
    const double dl = (c - r) * -log1p(exp(psi));          // (c - R) log(1 - p), once per observed replicate
    double base = row_rate ? tab[NB_TAB] : __builtin_nan("");
    double term = 0.0;
    for (int q = 0; q < nrep; ++q) {
      double yy;
      if constexpr (RR > 0) yy = yv[q]; else yy = y[q];
      if (yy == yy) {
        const bool tabbed = row_rate && yy >= 0.0 && yy < (double)NB_TAB && yy == floor(yy);
        term += dl + (tabbed ? tab[(int)yy] : nb_count_part(yy, r, c, base));
      }
    }
    if (shared_jt) bterm += term;
    else out[(size_t)i * MT + jt] = term;
  }
  if (shared_jt) {
    bterm = wave_sum(bterm);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bterm;
    __syncthreads();
    if (threadIdx.x == 0) out[(size_t)i * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

// out[o] = sum over the shared dims of src[(a0,a1,a2)], src C-contiguous (d0,d1,d2); sh* flag the shared dims
static __global__ __launch_bounds__(256) void nb_reduce_kernel(const double* __restrict__ src, int d0, int d1, int d2, int sh0,
                                                       int sh1, int sh2, double* __restrict__ out) {
  __shared__ double red[4];
  const int u0 = sh0 ? 1 : d0, u1 = sh1 ? 1 : d1, u2 = sh2 ? 1 : d2;   // unshared extents
  const int s0 = sh0 ? d0 : 1, s1 = sh1 ? d1 : 1, s2 = sh2 ? d2 : 1;   // shared extents
  int o = blockIdx.x;
  const int b2 = o % u2; o /= u2;
  const int b1 = o % u1; o /= u1;
  const int b0 = o;
  const long long ns = (long long)s0 * s1 * s2;
  double acc = 0.0;
  for (long long q = threadIdx.x; q < ns; q += 256) {
    long long x = q;
    const int c2 = (int)(x % s2); x /= s2;
    const int c1 = (int)(x % s1); x /= s1;
    const int c0 = (int)x;
    const int a0 = sh0 ? c0 : b0, a1 = sh1 ? c1 : b1, a2 = sh2 ? c2 : b2;
    acc += src[((size_t)a0 * d1 + a1) * d2 + a2];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// S[cell] = sum of observed counts, cnt[cell] = number of observed replicates
static __global__ void nb_stats_kernel(const double* __restrict__ data, int Rr, size_t cells, double* __restrict__ S,
                                double* __restrict__ cnt) {
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (size_t)gridDim.x * blockDim.x) {
    double s = 0.0, n = 0.0;
    for (int q = 0; q < Rr; ++q) {
      const double y = data[c * Rr + q];
      if (y == y) { s += y; n += 1.0; }
    }
    S[c] = s; cnt[c] = n;
  }
}

// 64 x 64 tiles of the (N, MT) cell grid: V layout written directly, W layout through an LDS transpose
// (sharded contexts hold the WHOLE count tensor - the rate update is a function of all of it and is computed by every rank,
//  like the other hyper-parameters - and only their two slabs of the augmented model: rows row0 .. row0+nl-1 go to the
//  transposed layout, columns jt_lo .. jt_lo+jt_n-1 of the (j,t) axis to the other)
static __global__ __launch_bounds__(256) void nb_trials_kernel(const double* __restrict__ S, const double* __restrict__ cnt,
                                                       const double* __restrict__ Rv, long long sr0, long long sr1,
                                                       long long sr2, int N, int MT, int T, int ldv, int ldw,
                                                       double* __restrict__ Av, double* __restrict__ Bv,
                                                       double* __restrict__ AwT, double* __restrict__ BwT,
                                                       int row0, int nl, int jt_lo, int jt_n, int hrow, int hjt) {
  // (hrow / hjt: the halo source row / first (j,t) of the halo source column, -1: none - local index nl / jt_n .. jt_n+T-1)
  __shared__ double ta[64][65], tb[64][65];
  const int col = threadIdx.x & 63;
  const int rgrp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int jt0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
  const int jt = jt0 + col;
  const int j = jt / T, t = jt - j * T;
  for (int q = 0; q < 16; ++q) {
    const int r = rgrp + 4 * q, i = i0 + r;
    double a = 0.0, b = 0.0;
    if (i < N && jt < MT) {
      const size_t cell = (size_t)i * MT + jt;
      const double n = cnt[cell], s = S[cell];
      if (n > 0.0) {
        b = s + n * Rv[(long long)i * sr0 + (long long)j * sr1 + (long long)t * sr2];
        a = s - 0.5 * b;
      }
      if (jt >= jt_lo && jt < jt_lo + jt_n) {
        Av[(size_t)i * ldv + (jt - jt_lo)] = a;
        Bv[(size_t)i * ldv + (jt - jt_lo)] = b;
      } else if (hjt >= 0 && jt >= hjt && jt < hjt + T) {
        Av[(size_t)i * ldv + (jt_n + jt - hjt)] = a;
        Bv[(size_t)i * ldv + (jt_n + jt - hjt)] = b;
      }
    }
    ta[r][col] = a; tb[r][col] = b;
  }
  __syncthreads();
  const int i = i0 + col;
  for (int q = 0; q < 16; ++q) {
    const int c = rgrp + 4 * q, jj = jt0 + c;
    if (jj < MT && i >= row0 && i < row0 + nl) {
      AwT[(size_t)jj * ldw + (i - row0)] = ta[col][c];
      if (BwT) BwT[(size_t)jj * ldw + (i - row0)] = tb[col][c];     // (nullptr: nobody reads the trial counts in this layout)
    } else if (jj < MT && i == hrow) {
      AwT[(size_t)jj * ldw + nl] = ta[col][c];
      if (BwT) BwT[(size_t)jj * ldw + nl] = tb[col][c];
    }
  }
}

// one kept state of the chain into its slot of the sample store: W, V, Tau2 and the device-resident scalars in ONE launch
// (four hipMemcpyAsync device-to-device copies cost ~70 us per kept sample at C3 - more than the sweep that produced it)
static __global__ __launch_bounds__(256) void collect_kernel(const double* __restrict__ W, size_t nW, const double* __restrict__ V,
                                                     size_t nV, const double* __restrict__ Tau2, size_t nT,
                                                     const double* __restrict__ hyp, int nh, double* __restrict__ sW,
                                                     double* __restrict__ sV, double* __restrict__ sT, double* __restrict__ ss) {
  const size_t tot = nW + nV + nT + (size_t)nh;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
    if (i < nW) sW[i] = W[i];
    else if (i < nW + nV) sV[i - nW] = V[i - nW];
    else if (i < nW + nV + nT) sT[i - nW - nV] = Tau2[i - nW - nV];
    else ss[i - nW - nV - nT] = hyp[i - nW - nV - nT];
  }
}

// ---- count histograms: when R is shared along (j,t) the MH log-likelihood ratio of row i is
//        sum_y H[i][y] * tab_y(R_i, c_i) + (c_i - R_i) * L[i],
//      H[i][y] = number of observed replicates in row i equal to y (data only: built once at upload),
//      L[i]    = sum_(j,t) cnt * log(1 - p_ijt)                  (W, V only: built once per sweep),
//      so the 30 MH steps of a sweep read 8 KB per row instead of the whole count tensor.
// H via integer atomics (exact, order-free).  Observed values that are not integers in [0, NB_TAB)
// ("outliers": large counts, fractional pseudo-counts) are counted per row, then compacted IN DATA
// ORDER into per-row lists (nb_outlier_fill_kernel) and summed term by term in that fixed order.
static __global__ __launch_bounds__(256) void nb_hist_kernel(const double* __restrict__ data, int Rr, int MT,
                                                     unsigned int* __restrict__ H, int* __restrict__ nout) {
  __shared__ unsigned int h[NB_TAB];
  const int i = blockIdx.y;
  for (int y = threadIdx.x; y < NB_TAB; y += 256) h[y] = 0u;
  __syncthreads();
  const size_t n = (size_t)MT * Rr;
  const double* __restrict__ row = data + (size_t)i * n;
  int b = 0;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const double y = row[e];
    if (y == y) {
      if (y >= 0.0 && y < (double)NB_TAB && y == floor(y)) atomicAdd(&h[(int)y], 1u);
      else ++b;
    }
  }
  if (b) atomicAdd(&nout[i], b);
  __syncthreads();
  for (int y = threadIdx.x; y < NB_TAB; y += 256)
    if (h[y]) atomicAdd(&H[(size_t)i * NB_TAB + y], h[y]);
}

// in-order compaction of row i's outliers into val[ptr[i] ..): one block per row, chunks of 256 in order
static __global__ __launch_bounds__(256) void nb_outlier_fill_kernel(const double* __restrict__ data, int Rr, int MT,
                                                             const int* __restrict__ ptr, double* __restrict__ val) {
  __shared__ int wcnt[4];
  __shared__ int base_s;
  const int i = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t n = (size_t)MT * Rr;
  const double* __restrict__ row = data + (size_t)i * n;
  if (threadIdx.x == 0) base_s = ptr[i];
  __syncthreads();
  for (size_t e0 = 0; e0 < n; e0 += 256) {
    const size_t e = e0 + threadIdx.x;
    const double y = e < n ? row[e] : 0.0;
    const bool out = e < n && y == y && !(y >= 0.0 && y < (double)NB_TAB && y == floor(y));
    const unsigned long long m = __ballot(out);
    if (lane == 0) wcnt[wv] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int u = 0; u < wv; ++u) off += wcnt[u];
    if (out) val[off + __popcll(m & ((1ull << lane) - 1ull))] = y;
    __syncthreads();
    if (threadIdx.x == 0) base_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
}

static __global__ void u32_to_f64_kernel(const unsigned int* __restrict__ src, double* __restrict__ dst, size_t n) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) dst[e] = (double)src[e];
}

// part[i][blockIdx.x] = sum over this block's (j,t) of cnt * log(1 - ilogit(clip(w_i.v_jt)))
template <int K>
__global__ __launch_bounds__(256) void nb_l1p_kernel(const double* __restrict__ cnt, const double* __restrict__ W,
                                                    const double* __restrict__ V, int MT, double* __restrict__ part) {
  __shared__ double red[4];
  __shared__ double2 ltab[LOGTAB_N], etab[LOGTAB_N];
  log_table_build(ltab);
  exp_table_build(etab);
  __syncthreads();
  const int i = blockIdx.y;
  const double* __restrict__ w = W + (size_t)i * K;
  double acc = 0.0;
  for (int jt = blockIdx.x * 256 + threadIdx.x; jt < MT; jt += gridDim.x * 256) {
    const double n = cnt[(size_t)i * MT + jt];
    double psi = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) psi = fma(w[k], V[(size_t)jt * K + k], psi);
    psi = fmin(fmax(psi, -10.0), 10.0);
    // log(1 + e^psi) = max(psi, 0) + log(1 + e^-|psi|): table exp and log (btf_device.h), 1 + e in (1, 2]
    if (n > 0.0) acc = fma(n, -(fmax(psi, 0.0) + log_tab(1.0 + exp_tab(-fabs(psi), etab), ltab)), acc);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)i * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// one block per row; rate_stride = 1: R per row, 0: one R for all rows (the caller then sums ll over rows):
//   ll[i] = sum_y H[i][y] tab_y(R, c) + (c - R) L[i] + sum over row i's outliers
static __global__ __launch_bounds__(256) void nb_hist_loglik_kernel(const double* __restrict__ Hd, const double* __restrict__ L,
                                                            const double* __restrict__ Rv, const double* __restrict__ Cv,
                                                            const int* __restrict__ optr, const double* __restrict__ oval,
                                                            int rate_stride, double* __restrict__ ll) {
  __shared__ double tab[NB_TAB];
  __shared__ double wsum[4];
  __shared__ double red[4];
  const int b = blockIdx.x;
  const double r = Rv[b * rate_stride], c = Cv[b * rate_stride];
  nb_build_table(r, c, tab, wsum);
  double acc = 0.0;
  for (int y = threadIdx.x; y < NB_TAB; y += 256) acc = fma(Hd[(size_t)b * NB_TAB + y], tab[y], acc);
  if (optr) {
    const int e0 = optr[b], e1 = optr[b + 1];
    if (e1 > e0) {
      const double base = lgamma_diff(r, c);
      for (int e = e0 + threadIdx.x; e < e1; e += 256) acc += lgamma_diff(oval[e] + c, oval[e] + r) + base;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ll[b] = red[0] + red[1] + red[2] + red[3] + (c - r) * L[b];
}

// One random-walk MH step on log R for every rate element, entirely on the device (rng="device";
// factor.py:523-549): accept the pending candidate C[e] with probability exp(clip(prior + ll, -10, 1))
// (and only if C[e] > 1, the reference's floor), then propose the next one,
//   C[e] = exp(log R[e] + rpropstdev * z).
// llrow: per-row log-likelihood ratios; scalar != 0: one rate, ll = fixed-order sum of the nrow entries.
// step < 0: propose only (start of the loop).  Philox streams keyed (seed, element), counters advance
// with the step so that every step sees fresh numbers.  One block; elements strided over its threads.
static __global__ __launch_bounds__(256) void nb_mh_step_kernel(const double* __restrict__ llrow, int nrow, int nelem,
                                                        int scalar, double* __restrict__ Rv, double* __restrict__ Cv,
                                                        double rpropstdev, double rstdev, int step,
                                                        unsigned long long seed) {
  __shared__ double red[4];
  double lls = 0.0;
  if (scalar && step >= 0) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nrow; i += 256) acc += llrow[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    lls = red[0] + red[1] + red[2] + red[3];
  }
  for (int e = threadIdx.x; e < nelem; e += 256) {
    CellRng g(seed, (unsigned long long)e);
    g.ctr = 2ull * (unsigned long long)(step + 1);           // two Philox blocks per step and element
    double r = Rv[e];
    if (step >= 0) {
      const double c = Cv[e];
      const double lr = log(r), lc = log(c);
      const double prior = (lr * lr - lc * lc) / (2.0 * rstdev * rstdev);
      const double ll = scalar ? lls : llrow[e];
      const double prob = exp(fmin(fmax(prior + ll, -10.0), 1.0));
      if (g.uniform() <= prob && c > 1.0) { r = c; Rv[e] = c; }
    }
    Cv[e] = exp(log(r) + rpropstdev * g.normal());
  }
}

// The whole random-walk MH loop for one rate PER ROW (rdims = (1,2), the sharing pattern of the reference's example), in one
// launch: the chains of different rows do not interact, so workgroup b runs row b's nsteps steps by itself - its histogram
// row in registers, the table of sum_{k<y} log((c+k)/(r+k)) rebuilt per step up to the largest count of the data, the
// accept / propose arithmetic of nb_mh_step_kernel on thread 0 (same Philox stream: element b, two blocks per step) -
// instead of two launches per step.  Same sums in the same order as nb_hist_loglik_kernel + nb_mh_step_kernel: the
// chains are bit-identical (BTF_NB_MH_STEPWISE=1 keeps the stepwise form for the test).
static __global__ __launch_bounds__(256) void nb_mh_rows_kernel(const double* __restrict__ Hd, const double* __restrict__ L,
                                                        const int* __restrict__ optr, const double* __restrict__ oval,
                                                        double* __restrict__ Rv, double* __restrict__ Cv, double rpropstdev,
                                                        double rstdev, int nsteps, unsigned long long seed, int ymax) {
  __shared__ double tab[NB_TAB];
  __shared__ double wsum[4];
  __shared__ double red[4];
  __shared__ double rc[2];
  const int b = blockIdx.x, tid = threadIdx.x;
  double h[NB_TAB / 256];
#pragma unroll
  for (int q = 0; q < NB_TAB / 256; ++q) h[q] = Hd[(size_t)b * NB_TAB + tid + 256 * q];
  const int e0 = optr ? optr[b] : 0, e1 = optr ? optr[b + 1] : 0;
  const double Lb = L[b];
  if (tid == 0) {          // step -1: propose only
    CellRng g(seed, (unsigned long long)b);
    g.ctr = 0ull;
    const double r = Rv[b];
    rc[0] = r;
    rc[1] = exp(log(r) + rpropstdev * g.normal());
  }
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const double r = rc[0], c = rc[1];
    nb_build_table(r, c, tab, wsum, ymax);               // (ends with a barrier: everybody has read rc)
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NB_TAB / 256; ++q) acc = fma(h[q], tab[tid + 256 * q], acc);
    if (e1 > e0) {
      const double base = lgamma_diff(r, c);
      for (int e = e0 + tid; e < e1; e += 256) acc += lgamma_diff(oval[e] + c, oval[e] + r) + base;
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      const double ll = red[0] + red[1] + red[2] + red[3] + (c - r) * Lb;
      CellRng g(seed, (unsigned long long)b);
      g.ctr = 2ull * (unsigned long long)(step + 1);
      const double lr = log(r), lc = log(c);
      const double prior = (lr * lr - lc * lc) / (2.0 * rstdev * rstdev);
      const double prob = exp(fmin(fmax(prior + ll, -10.0), 1.0));
      double rn = r;
      if (g.uniform() <= prob && c > 1.0) rn = c;
      rc[0] = rn;
      rc[1] = exp(log(rn) + rpropstdev * g.normal());
    }
    __syncthreads();
  }
  if (tid == 0) { Rv[b] = rc[0]; Cv[b] = rc[1]; }
}

// Single shared rate, more outliers than one workgroup should take: the per-step likelihood ratio as gridDim.x
// partial sums (workgroup g: logarithm k = 256 g + tid of the suffix-sum form, and its chunk of the outlier list;
// workgroup 0 adds (c - r) * sum L) - nb_mh_step_kernel then sums them as it sums per-row values.
static __global__ __launch_bounds__(256) void nb_scalar_part_kernel(const double* __restrict__ Gs, int ymax,
                                                            const double* __restrict__ Ltot, const double* __restrict__ oval,
                                                            int nout, const double* __restrict__ Rv,
                                                            const double* __restrict__ Cv, double* __restrict__ part) {
  __shared__ double red[4];
  const int tid = threadIdx.x, g = blockIdx.x;
  const double r = Rv[0], c = Cv[0];
  double acc = 0.0;
  const int k = g * 256 + tid;
  if (k < ymax) acc = Gs[k] * log((c + k) / (r + k));
  const int chunk = (nout + (int)gridDim.x - 1) / (int)gridDim.x;
  const int e0 = g * chunk, e1 = min(e0 + chunk, nout);
  if (e1 > e0) {
    const double base = lgamma_diff(r, c);
    for (int e = e0 + tid; e < e1; e += 256) acc += lgamma_diff(oval[e] + c, oval[e] + r) + base;
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) part[g] = red[0] + red[1] + red[2] + red[3] + (g == 0 ? (c - r) * Ltot[0] : 0.0);
}

// The same loop with ONE launch per step: launch l first takes the decision of step l-1 from the previous launch's partial
// sums (every workgroup by itself - same numbers, same result: the sum, the accept test and the proposal exactly as
// nb_mh_step_kernel forms them) and then forms its share of the ratio for the new (r, c).  The pair (r, c) is double-buffered
// (rc[l & 1] read, rc[(l + 1) & 1] written by workgroup 0) so that a late workgroup never reads a pair already replaced;
// l = 0: propose only; l = nsteps: decision only, rate and pending candidate go back to Rv / Cv.
static __global__ __launch_bounds__(256) void nb_scalar_step_kernel(const double* __restrict__ Gs, int ymax,
                                                            const double* __restrict__ Ltot, const double* __restrict__ oval,
                                                            int nout, double* __restrict__ Rv, double* __restrict__ Cv,
                                                            double* __restrict__ rcbuf, const double* __restrict__ part_prev,
                                                            double* __restrict__ part_out, double rpropstdev, double rstdev,
                                                            int l, int nsteps, unsigned long long seed) {
  __shared__ double red[4];
  __shared__ double rc[2];
  const int tid = threadIdx.x, g = blockIdx.x, gp = gridDim.x;
  const double* src = rcbuf + 2 * (l & 1);
  double lls = 0.0;
  if (l >= 1) {
    double acc = 0.0;
    for (int i = tid; i < gp; i += 256) acc += part_prev[i];
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    lls = red[0] + red[1] + red[2] + red[3];
  }
  if (tid == 0) {
    CellRng gen(seed, 0ull);
    gen.ctr = 2ull * (unsigned long long)l;                  // step = l - 1: two Philox blocks per step
    double r = l == 0 ? Rv[0] : src[0];
    if (l >= 1) {
      const double c = src[1];
      const double lr = log(r), lc = log(c);
      const double prior = (lr * lr - lc * lc) / (2.0 * rstdev * rstdev);
      const double prob = exp(fmin(fmax(prior + lls, -10.0), 1.0));
      if (gen.uniform() <= prob && c > 1.0) r = c;
    }
    rc[0] = r;
    rc[1] = exp(log(r) + rpropstdev * gen.normal());
    if (g == 0) {
      double* dst = rcbuf + 2 * ((l + 1) & 1);
      dst[0] = rc[0]; dst[1] = rc[1];
      if (l == nsteps) { Rv[0] = rc[0]; Cv[0] = rc[1]; }
    }
  }
  __syncthreads();
  if (l == nsteps) return;
  const double r = rc[0], c = rc[1];
  double acc = 0.0;
  const int k = g * 256 + tid;
  if (k < ymax) acc = Gs[k] * log((c + k) / (r + k));
  const int chunk = (nout + gp - 1) / gp;
  const int e0 = g * chunk, e1 = min(e0 + chunk, nout);
  if (e1 > e0) {
    const double base = lgamma_diff(r, c);
    for (int e = e0 + tid; e < e1; e += 256) acc += lgamma_diff(oval[e] + c, oval[e] + r) + base;
  }
  acc = wave_sum(acc);
  __syncthreads();                                           // (red is read above by every thread)
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) part_out[g] = red[0] + red[1] + red[2] + red[3] + (g == 0 ? (c - r) * Ltot[0] : 0.0);
}

// The whole random-walk MH loop for ONE rate shared by every cell (rdims = (0,1,2), the reference's default), in one
// launch by one workgroup: with a single rate the likelihood ratio needs only the histogram of all counts, as its
// suffix sums Gs[k] = #{observations > k}, k < ymax:  sum_y H[y] sum_{k<y} log((c+k)/(r+k)) = sum_k Gs[k] log((c+k)/(r+k))
// - ymax logarithms per step, no table, no scan -, the sum of the rows' L and the outlier list; then the accept /
// propose arithmetic of nb_mh_step_kernel (same Philox stream: element 0, two blocks per step), instead of two
// launches per step with every row rebuilding the same table.
static __global__ __launch_bounds__(256) void nb_mh_scalar_kernel(const double* __restrict__ Gs, int ymax,
                                                          const double* __restrict__ L, int nrow, const int* __restrict__ optr,
                                                          const double* __restrict__ oval, double* __restrict__ Rv,
                                                          double* __restrict__ Cv, double rpropstdev, double rstdev,
                                                          int nsteps, unsigned long long seed) {
  __shared__ double red[4];
  __shared__ double rc[2];
  const int tid = threadIdx.x;
  double acc = 0.0;
  for (int i = tid; i < nrow; i += 256) acc += L[i];
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  if (tid == 0) {                                            // the step "-1" of nb_mh_step_kernel: propose only
    CellRng g(seed, 0ull);
    g.ctr = 0ull;
    const double r = Rv[0];
    rc[0] = r;
    rc[1] = exp(log(r) + rpropstdev * g.normal());
  }
  __syncthreads();
  const double Ltot = red[0] + red[1] + red[2] + red[3];
  const int e0 = optr ? optr[0] : 0, e1 = optr ? optr[nrow] : 0;
  for (int step = 0; step < nsteps; ++step) {
    const double r = rc[0], c = rc[1];
    acc = 0.0;
    for (int k = tid; k < ymax; k += 256) acc = fma(Gs[k], log((c + k) / (r + k)), acc);
    if (e1 > e0) {
      const double base = lgamma_diff(r, c);
      for (int e = e0 + tid; e < e1; e += 256) acc += lgamma_diff(oval[e] + c, oval[e] + r) + base;
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      const double ll = red[0] + red[1] + red[2] + red[3] + (c - r) * Ltot;
      CellRng g(seed, 0ull);
      g.ctr = 2ull * (unsigned long long)(step + 1);
      const double lr = log(r), lc = log(c);
      const double prior = (lr * lr - lc * lc) / (2.0 * rstdev * rstdev);
      const double prob = exp(fmin(fmax(prior + ll, -10.0), 1.0));
      double rn = r;
      if (g.uniform() <= prob && c > 1.0) rn = c;
      rc[0] = rn;
      rc[1] = exp(log(rn) + rpropstdev * g.normal());
    }
    __syncthreads();
  }
  if (tid == 0) { Rv[0] = rc[0]; Cv[0] = rc[1]; }
}

// ============================================================================
// Posterior summaries over the kept samples  
//   what the reference's example scripts do on the host with
//     Mu = einsum('znk,zmtk->znmt', Ws, Vs);  Mu.mean(0);  np.percentile(Mu, q, axis=0)
//   (examples/gaussian_tensor_filtering.py:82-85) - an (S, N, M, T) tensor (67 GB at C3, S = 1000) that is
//   never materialised here: a workgroup takes `cells` consecutive (j,t) cells of one row i, computes
//   their S values f(w_s . v_s) into LDS (rows padded to a power of two with +inf), sorts every row
//   with a bitonic network, and reads the mean and the order statistics off the sorted rows.
//   Percentiles follow numpy's default ('linear'): pos = q/100 (S-1), x[lo] + frac (x[lo+1] - x[lo]).
//   transform: 0 identity, 1 ilogit (Binomial examples), 2 square.
// ============================================================================
template <int K>
__global__ __launch_bounds__(256) void posterior_summary_kernel(const double* __restrict__ Ws, const double* __restrict__ Vs,
                                                               int S, int N, int MT, int P, int cells, int transform,
                                                               const double* __restrict__ q, int nq,
                                                               double* __restrict__ mean_out, double* __restrict__ q_out) {
  extern __shared__ double srt[];                 // [cells][P]
  const int i = blockIdx.y, jt0 = blockIdx.x * cells;
  const int nc = min(cells, MT - jt0);
  // ---- values: thread -> (cell c, sample s); V[s][jt0 + c][:] is contiguous over c
  for (int e = threadIdx.x; e < cells * P; e += 256) {
    const int c = e % cells, sidx = e / cells;
    double val = __builtin_inf();
    if (sidx < S && c < nc) {
      const double* __restrict__ w = Ws + ((size_t)sidx * N + i) * K;
      const double* __restrict__ v = Vs + ((size_t)sidx * MT + jt0 + c) * K;
      double x = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) x = fma(w[k], v[k], x);
      val = transform == 1 ? 1.0 / (1.0 + exp(-x)) : (transform == 2 ? x * x : x);
    }
    srt[(size_t)c * P + sidx] = val;
  }
  __syncthreads();
  // ---- bitonic sort of every row (ascending); P/2 compare-exchanges per row and stage
  const int half = P >> 1;
  for (int kk = 2; kk <= P; kk <<= 1) {
    for (int jj = kk >> 1; jj > 0; jj >>= 1) {
      for (int e = threadIdx.x; e < cells * half; e += 256) {
        const int c = e / half, pidx = e - c * half;
        const int i1 = ((pidx / jj) * 2 * jj) + (pidx % jj), i2 = i1 + jj;
        double* row = srt + (size_t)c * P;
        const double a = row[i1], b = row[i2];
        const bool up = (i1 & kk) == 0;
        if ((a > b) == up) { row[i1] = b; row[i2] = a; }
      }
      __syncthreads();
    }
  }
  // ---- mean (fixed order: ascending values) and percentiles
  for (int c = threadIdx.x; c < nc; c += 256) {
    const double* row = srt + (size_t)c * P;
    double sum = 0.0;
    for (int sidx = 0; sidx < S; ++sidx) sum += row[sidx];
    mean_out[(size_t)i * MT + jt0 + c] = sum / S;
  }
  for (int e = threadIdx.x; e < nc * nq; e += 256) {
    const int c = e % nc, qi = e / nc;
    const double* row = srt + (size_t)c * P;
    const double pos = q[qi] * 0.01 * (S - 1);
    int lo = (int)floor(pos);
    lo = max(0, min(lo, S - 1));
    const int hi = min(lo + 1, S - 1);
    const double frac = pos - lo;
    q_out[((size_t)qi * N + i) * MT + jt0 + c] = row[lo] + frac * (row[hi] - row[lo]);
  }
}

// ============================================================================
// Residual sum of squares from the W half-sweep's accumulation partials (rng="device" full sweep):
//   sum_cells (S1 - cnt mu)^2 / cnt = sum S1^2/cnt - 2 sum_i w_i . m_i + sum_i w_i' Q_i w_i,
//   m_i = sum_(j,t) S1 v_jt and Q_i = sum_(j,t) cnt v_jt v_jt' (= R V'V on complete data) being exactly
//   what accum_kernel just produced for the W step with the CURRENT V - so nu2 | rest needs no pass of
//   its own over the data (sse_kernel: 15 us at C3).  One lane per row, chunks split over the waves as in
//   w_solve_kernel; bsum[block] = sum over the block's rows of  -2 w.m + w'Qw.
// ============================================================================
template <int K, bool WEIGHTED>
__global__ __launch_bounds__(WS_ROWS * ws_split(K)) void sse_part_kernel(const double* __restrict__ part, int nch, int ld,
                                                                        const double* __restrict__ gpart, int ngp, double Rrep,
                                                                        const double* __restrict__ W, int row0, int nl,
                                                                        double* __restrict__ bsum, CurveLists cv,
                                                                        const double* __restrict__ cv_blocks) {
  constexpr int KK = tri(K);
  constexpr int WS_SPLIT = ws_split(K);
  constexpr int NV = WEIGHTED ? K + KK : K;
  __shared__ double G[KK];
  __shared__ double red[WS_SPLIT][K + KK][WS_ROWS];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  if constexpr (!WEIGHTED) reduce_gram(gpart, ngp, KK, Rrep, &red[0][0][0], G);
  const int il = blockIdx.x * WS_ROWS + lane;
  const bool live = il < nl;
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.0;
  if (live) {
    const size_t cst = (size_t)NV * ld;
    int c = grp;
    for (; c + 3 * WS_SPLIT < nch; c += 4 * WS_SPLIT) {          // four chunks' loads in flight, added in chunk order
      const double* p = part + (size_t)c * cst + il;
      double x[4][NV];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) x[u][v] = p[(size_t)u * WS_SPLIT * cst + (size_t)v * ld];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] += x[u][v];
    }
    for (; c < nch; c += WS_SPLIT) {
      const double* p = part + (size_t)c * cst + il;
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] += p[(size_t)v * ld];
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) red[grp][v][lane] = acc[v];
  __syncthreads();
  if (grp != 0) return;
  double s = 0.0;
  if (live) {
    double w[K], m[K], Q[KK];
    const double* wr = W + (size_t)(row0 + il) * K;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      w[k] = wr[k];
      double t = 0.0;
#pragma unroll
      for (int g = 0; g < WS_SPLIT; ++g) t += red[g][k][lane];
      m[k] = t;
    }
#pragma unroll
    for (int q = 0; q < KK; ++q) {
      if constexpr (WEIGHTED) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < WS_SPLIT; ++g) t += red[g][K + q][lane];
        Q[q] = t;
      } else {
        Q[q] = G[q];
      }
    }
    if constexpr (!WEIGHTED) {
      if (cv.ptr) {
        for (int e = cv.ptr[row0 + il]; e < cv.ptr[row0 + il + 1]; ++e) {
          const double* __restrict__ blk = cv_blocks + (size_t)cv.idx[e] * KK;
          const double f = cv.def[e];
#pragma unroll
          for (int q = 0; q < KK; ++q) Q[q] = fma(-f, blk[q], Q[q]);
        }
      }
    }
    double lin = 0.0, quad = 0.0;
#pragma unroll
    for (int a = 0; a < K; ++a) {
      lin = fma(w[a], m[a], lin);
#pragma unroll
      for (int b = 0; b <= a; ++b) quad = fma((a == b ? 1.0 : 2.0) * w[a] * w[b], Q[lidx(a, b)], quad);
    }
    s = quad - 2.0 * lin;
  }
  s = wave_sum(s);
  if (lane == 0) bsum[blockIdx.x] = s;
}

}  // namespace btf
