// The kernel families that make up most of the code object (accumulation, W solve, Polya-Gamma, Negative-Binomial
// log-likelihood, twisted banded sampler: 80 % of it) are instantiated in translation units of their own
// (btf_instances.hip, compiled once per BTF_INST_PART, in parallel) and only declared here - `extern template` for
// the C-ABI unit (btf_abi.hip), which launches them through their host stubs.  One list, two expansions.
//
//   BTF_INST_PART undefined : every family as `extern template` (include after the kernel headers)
//   BTF_INST_PART = n       : explicit instantiation definitions of part n
#pragma once
#include "btf_kernels.h"
#include "btf_banded_twist.h"
#include "btf_spectral.h"
#include "btf_fused.h"

namespace btf {

#define BTF_ACC_ARGS_F(XT, CT, FZ) \
  (const XT*, const CT*, const double*, const int*, double*, int, int, int, EigSide, EigSideCols, TauSide, GramSide, ChunkMap, SweepSide, FZ)
#define BTF_ACC_ARGS(XT, CT) BTF_ACC_ARGS_F(XT, CT, FuseNone)
#define BTF_ACCUM_SET(P, K)                                                                              \
  P void accum_kernel<K, 0> BTF_ACC_ARGS(double, double);                                                \
  P void accum_kernel<K, 0, acc_waves(K, 0), double, double, 3> BTF_ACC_ARGS(double, double);            \
  P void accum_kernel<K, 1> BTF_ACC_ARGS(double, double);                                                \
  P void accum_kernel<K, 2> BTF_ACC_ARGS(double, double);                                                \
  P void accum_kernel<K, 1, acc_waves(K, 1), unsigned char> BTF_ACC_ARGS(double, unsigned char);         \
  P void accum_kernel<K, 2, acc_waves(K, 2), unsigned char> BTF_ACC_ARGS(double, unsigned char);         \
  P void accum_kernel<K, 1, acc_waves(K, 1), double, signed char> BTF_ACC_ARGS(signed char, double);     \
  P void accum_kernel<K, 2, acc_waves(K, 2), double, signed char> BTF_ACC_ARGS(signed char, double);

// nembeds 10, complete data: the W launch (no eigen side task, whose unrolled form is what needs the 8-wave register budget)
// runs 16 waves per workgroup like the smaller K: 16.7 -> 13.3 us at (512,256,64), 5.0 TB/s (not the three-rows-in-flight
// form of long row ranges: two spilled VGPRs at 16 waves - it stays with 8)
#define BTF_ACCUM_K10W_SET(P)                                                                            \
  P void accum_kernel<10, 0, ACC_WAVES> BTF_ACC_ARGS(double, double);

// the fused forms of the W+V step (btf_fused.h, BTF_OPT_FUSED_STEP): complete data, 16 waves, nembeds <= 8 (where the owner's
// batch of chunk loads / the eigen side task fit the 16-wave register budget); two and - nembeds 8 - three rows in flight
#define BTF_FUSED_W_SET(P, K)                                                                            \
  P void accum_kernel<K, 0, 16, double, double, 0, 2, FUSE_W> BTF_ACC_ARGS_F(double, double, FuseW);
#define BTF_FUSED_V_SET(P, K)                                                                            \
  P void accum_kernel<K, 0, 16, double, double, 0, 2, FUSE_V> BTF_ACC_ARGS_F(double, double, FuseV);
#define BTF_FUSED_VDF_SET(P, K)                                                                          \
  P void accum_kernel<K, 0, 16, double, double, 0, 2, FUSE_VDF> BTF_ACC_ARGS_F(double, double, FuseV);
#define BTF_LEAN_SET(P, K)                                                                               \
  P void accum_kernel<K, 0, 16, double, double, 0, 2, FUSE_LEAN> BTF_ACC_ARGS(double, double);
#define BTF_FUSED_UNR3_SET(P, K)                                                                         \
  P void accum_kernel<K, 0, 16, double, double, 3, 2, FUSE_W> BTF_ACC_ARGS_F(double, double, FuseW);     \
  P void accum_kernel<K, 0, 16, double, double, 3, 2, FUSE_V> BTF_ACC_ARGS_F(double, double, FuseV);
#define BTF_FUSED_SET(P)                                                                                 \
  BTF_FUSED_W_SET(P, 1) BTF_FUSED_W_SET(P, 2) BTF_FUSED_W_SET(P, 3) BTF_FUSED_W_SET(P, 4)                \
  BTF_FUSED_W_SET(P, 5) BTF_FUSED_W_SET(P, 6) BTF_FUSED_W_SET(P, 7) BTF_FUSED_W_SET(P, 8)                \
  BTF_FUSED_V_SET(P, 1) BTF_FUSED_V_SET(P, 2) BTF_FUSED_V_SET(P, 3) BTF_FUSED_V_SET(P, 4)                \
  BTF_FUSED_V_SET(P, 5) BTF_FUSED_V_SET(P, 6) BTF_FUSED_V_SET(P, 7) BTF_FUSED_V_SET(P, 8)                \
  BTF_FUSED_VDF_SET(P, 1) BTF_FUSED_VDF_SET(P, 2) BTF_FUSED_VDF_SET(P, 3) BTF_FUSED_VDF_SET(P, 4)        \
  BTF_FUSED_VDF_SET(P, 5) BTF_FUSED_VDF_SET(P, 6)                                                        \
  BTF_LEAN_SET(P, 1) BTF_LEAN_SET(P, 2) BTF_LEAN_SET(P, 3) BTF_LEAN_SET(P, 4)                            \
  BTF_LEAN_SET(P, 5) BTF_LEAN_SET(P, 6) BTF_LEAN_SET(P, 7) BTF_LEAN_SET(P, 8)                            \
  BTF_FUSED_UNR3_SET(P, 8)

#define BTF_WSOLVE_SET(P, K)                                                                             \
  P void w_solve_kernel<K, false, 8>(WSolveArgs);  P void w_solve_kernel<K, true, 8>(WSolveArgs);        \
  P void w_solve_kernel<K, false, 16>(WSolveArgs); P void w_solve_kernel<K, true, 16>(WSolveArgs);       \
  P void w_solve_kernel<K, false, 32>(WSolveArgs); P void w_solve_kernel<K, true, 32>(WSolveArgs);       \
  P void w_solve_kernel<K, false, 64>(WSolveArgs); P void w_solve_kernel<K, true, 64>(WSolveArgs);

#define BTF_PG_ARGS                                                                                      \
  (const double*, double*, const double*, const double*, int, int, int, int, unsigned long long, unsigned long long, \
   unsigned long long, unsigned long long, int, int)
#define BTF_PGT_ARGS \
  (const double*, double*, double*, const double*, const double*, int, int, int, int, unsigned long long, int, int)
#define BTF_PG_SET(P, K)                                                                                 \
  P void pg_kernel<K, PG_PATH_SERIES> BTF_PG_ARGS;       P void pg_kernel<K, PG_PATH_EXACT> BTF_PG_ARGS;  \
  P void pg_tile_kernel<K, PG_PATH_SERIES> BTF_PGT_ARGS; P void pg_tile_kernel<K, PG_PATH_EXACT> BTF_PGT_ARGS; \
  P void pgx_kernel<K, PGX_CPL> BTF_PG_ARGS;             P void pgx_tile_kernel<K, PGX_NW, PGX_CPL> BTF_PGT_ARGS;

#define BTF_NB_ARGS \
  (const double*, int, const double*, const double*, int, int, const double*, const double*, long long, long long, long long, int, double*)
#define BTF_NB_SET(P, K)                                                                                 \
  P void nb_loglik_kernel<K, 0> BTF_NB_ARGS; P void nb_loglik_kernel<K, 1> BTF_NB_ARGS;                  \
  P void nb_loglik_kernel<K, 2> BTF_NB_ARGS; P void nb_loglik_kernel<K, 3> BTF_NB_ARGS;                  \
  P void nb_loglik_kernel<K, 4> BTF_NB_ARGS;

#define BTF_TWIST_SET(P)                                                                                 \
  P void v_banded_twist_kernel<1, true>(VBandArgs, int);  P void v_banded_twist_kernel<2, true>(VBandArgs, int);  \
  P void v_banded_twist_kernel<2, false>(VBandArgs, int); P void v_banded_twist_kernel<3, false>(VBandArgs, int); \
  P void v_banded_twist_kernel<4, false>(VBandArgs, int); P void v_banded_twist_kernel<5, false>(VBandArgs, int); \
  P void v_banded_twist_kernel<6, false>(VBandArgs, int); P void v_banded_twist_kernel<7, false>(VBandArgs, int); \
  P void v_banded_twist_kernel<8, false>(VBandArgs, int);                                                \
  BTF_TWIST_FIXED(P, 1, 2) BTF_TWIST_FIXED(P, 2, 2) BTF_TWIST_FIXED(P, 3, 2) BTF_TWIST_FIXED(P, 4, 2) BTF_TWIST_FIXED(P, 5, 2) \
  BTF_TWIST_FIXED(P, 6, 2) BTF_TWIST_FIXED(P, 7, 2) BTF_TWIST_FIXED(P, 8, 2) BTF_TWIST_FIXED(P, 9, 2) BTF_TWIST_FIXED(P, 10, 2) \
  BTF_TWIST_FIXED_T(P, 5, 2, 64) BTF_TWIST_FIXED_T(P, 8, 2, 64)
// (nembeds, tf_order) as compile-time constants: the reference's default tf_order = 2, every supported nembeds
#define BTF_TWIST_FIXED(P, K, TF) P void v_banded_twist_kernel<tw_npl(K, TF), tw_row16(K, TF), K, TF>(VBandArgs, int);
// ... and ndepth 64 (BASELINE configs 3 / 4 / 5): nembeds 5 and 8
#define BTF_TWIST_FIXED_T(P, K, TF, T) P void v_banded_twist_kernel<tw_npl(K, TF), tw_row16(K, TF), K, TF, T>(VBandArgs, int);

#define BTF_FOR_K(SET, P) \
  SET(P, 1) SET(P, 2) SET(P, 3) SET(P, 4) SET(P, 5) SET(P, 6) SET(P, 7) SET(P, 8) SET(P, 9) SET(P, 10)

// The parts (balanced by code size: the accumulation kernels grow with K + K(K+1)/2):
//   0: accumulation K = 10, 4      1: accumulation K = 9, 5      2: accumulation K = 8, 6
//   3: accumulation K = 7, 3, 2, 1 4: W solve                    5: Polya-Gamma
//   6: Negative-Binomial log-likelihood                         7: twisted banded sampler
//   8: the fused W / V launches
constexpr int BTF_INST_PARTS = 9;

#ifndef BTF_INST_PART
#define BTF_X extern template __global__
BTF_FOR_K(BTF_ACCUM_SET, BTF_X)
BTF_ACCUM_K10W_SET(BTF_X)
BTF_FOR_K(BTF_WSOLVE_SET, BTF_X)
BTF_FOR_K(BTF_PG_SET, BTF_X)
BTF_FOR_K(BTF_NB_SET, BTF_X)
BTF_TWIST_SET(BTF_X)
BTF_FUSED_SET(BTF_X)
#undef BTF_X
#else
#define BTF_D template __global__
#if BTF_INST_PART == 0
BTF_ACCUM_SET(BTF_D, 10) BTF_ACCUM_K10W_SET(BTF_D) BTF_ACCUM_SET(BTF_D, 4)
#elif BTF_INST_PART == 1
BTF_ACCUM_SET(BTF_D, 9) BTF_ACCUM_SET(BTF_D, 5)
#elif BTF_INST_PART == 2
BTF_ACCUM_SET(BTF_D, 8) BTF_ACCUM_SET(BTF_D, 6)
#elif BTF_INST_PART == 3
BTF_ACCUM_SET(BTF_D, 7) BTF_ACCUM_SET(BTF_D, 3) BTF_ACCUM_SET(BTF_D, 2) BTF_ACCUM_SET(BTF_D, 1)
#elif BTF_INST_PART == 4
BTF_FOR_K(BTF_WSOLVE_SET, BTF_D)
#elif BTF_INST_PART == 5
BTF_FOR_K(BTF_PG_SET, BTF_D)
#elif BTF_INST_PART == 6
BTF_FOR_K(BTF_NB_SET, BTF_D)
#elif BTF_INST_PART == 7
BTF_TWIST_SET(BTF_D)
#elif BTF_INST_PART == 8
BTF_FUSED_SET(BTF_D)
#else
#error "BTF_INST_PART out of range"
#endif
#undef BTF_D
#endif

}  // namespace btf
