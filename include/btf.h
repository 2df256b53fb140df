/* btf.h - C ABI of the MI355X-native Gibbs core for Bayesian Tensor Filtering.
 *
 * This is the drop-in boundary for the hot path of tansey/functionalmf
 *   run_gibbs -> resample -> _resample_W / _resample_V / _resample_nu2
 * (reference functionalmf/factor.py:306-460, functionalmf/fast_mvn.py:10-74).
 * The reference has no FFI layer of its own: the boundary there is the Python
 * method contract of the model object.  Each entry point below
 * names the reference code whose *body* it replaces; the Python classes in
 * functionalmf_amd/factor.py keep the reference's method names and call these
 * through ctypes (see INTEGRATION.md for the binding a maintainer would add to
 * functionalmf/factor.py itself).
 *
 * Conventions
 *   - plain C types only; all arrays are float64, C-contiguous, caller-owned and
 *     borrowed for the duration of the call;
 *   - every function returns a BTF_* status; text via btf_last_error();
 *   - one ctx = one Markov chain on one GPU; calls on a ctx must be serialised by
 *     the caller; different ctxs may live on different threads / devices;
 *   - "step" functions only enqueue work on the ctx's HIP stream; results are
 *     visible after btf_sync() or any btf_get_* (which synchronise);
 *   - NaN marks a missing observation (reference convention);
 *   - dims: N=nrows, M=ncols, T=ndepth, R=nreps, K=nembeds, nD = rows of the
 *     trend-filter penalty Delta (utils.py:66-90).
 */
#ifndef BTF_H_
#define BTF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct btf_ctx btf_ctx;

enum {
  BTF_OK = 0,
  BTF_EINVAL = 1, /* bad argument / unsupported shape                          */
  BTF_EHIP = 2,   /* a HIP runtime call failed (text in btf_last_error)        */
  BTF_ENOTPD = 3, /* a conditional precision was not positive definite (after
                     the jitter retries for V; immediately for W, as
                     np.linalg.cholesky raises at factor.py:357).  Where the
                     reference would warn forever (fast_mvn.py:69-72) this
                     returns instead.  btf_fail_index() names the row/column.  */
  BTF_ESTATE = 4  /* call order: data / state not set yet                      */
};

enum { BTF_COMPAT_REFERENCE = 0, BTF_COMPAT_EXACT = 1 };

/* kernel ids for btf_kernel_times() */
enum {
  BTF_K_STATS = 0,   /* one-time sufficient statistics                      */
  BTF_K_W_ACCUM = 1, /* streaming Gram/mean accumulation, W half-sweep      */
  BTF_K_W_SOLVE = 2, /* batched KxK Cholesky + draw per row                 */
  BTF_K_V_ACCUM = 3, /* streaming Gram/mean accumulation, V half-sweep      */
  BTF_K_V_BANDED = 4,/* block-banded Cholesky sampler per column (fast_mvn) */
  BTF_K_GRAM = 5,    /* K x K Gram of the fixed factor (complete-data path) */
  BTF_K_PROD = 6,    /* small reductions (NB log-likelihood partials); the per-row outer products of the weighted
                        path are formed inside the accumulation since round 2 */
  BTF_K_SSE = 7,     /* residual sum of squares for nu2                     */
  BTF_K_PG = 8,      /* Polya-Gamma draws                                   */
  BTF_K_NB = 9,      /* Negative-Binomial rate update: MH log-likelihood ratio */
  BTF_K_PRIOR = 10,  /* prior band Delta' diag(1/(lam2 Tau2_j)) Delta of every column (factor.py:404-405) */
  BTF_K_EIG = 11,    /* eigen-system of the K x K Gram (spectral sampler)   */
  BTF_K_HYPER = 12,  /* device hyper-parameter draws: Tau2 chain, nu2/sigma2, lam2 */
  BTF_K_ESS = 13,    /* elliptical slice sampling: prior draws, proposals, likelihood passes, decisions */
  BTF_K_COUNT = 14
};

/* btf_set_option keys / values */
enum {
  BTF_OPT_SAMPLER = 0,       /* V half-sweep sampler, BTF_SAMPLER_*                                   */
  BTF_OPT_NB_HISTOGRAMS = 1, /* 1 (default): Negative-Binomial rate update from per-row count
                                histograms where they apply; 0: always the full-tensor kernel          */
  BTF_OPT_FUSE_GRAM = 2,     /* 1 (default): W'W / V'V partials come out of the preceding solve kernel */
  BTF_OPT_CURVE_COUNTS = 4,  /* 1 (default): Gaussian data whose replicate counts do not vary along the depth axis
                                (whole curves missing, as Y[:3,:3] = NaN in the reference's examples) run the
                                complete-data kernels plus per-row / per-column corrections (same conditionals as the
                                weighted form of factor.py:343-346, :388-391); 0: always the weighted form         */
  BTF_OPT_FUSED_SWEEP = 6,   /* 1 (default): full sweeps with device-resident scalars on complete Gaussian data run as FOUR launches -
                                the spectral V sampler leaves every column's part of the residual sum of squares behind, nu2 | rest
                                and sigma2 | rest of the next sweep ride in its W accumulation launch as a side workgroup
                                (btf_queue_scalars), lam2 | rest in the V accumulation launch (btf_queue_lam2) - instead of six
                                (a residual reduction and a scalar-draw launch between W accumulation and W solve).  Same
                                conditionals, same Philox streams; the sums are formed in another order (rounding).  0: six.   */
  BTF_OPT_FUSED_STEP = 7,    /* Latency kernels as parts of the streaming launches (csrc/btf_fused.h), complete Gaussian data.
                                0: a W+V step is four launches.
                                1 (default): the spectral sampler of _resample_V (factor.py:377-409, fast_mvn.py:35-47) is the TAIL
                                of the V accumulation launch - a tile of 128 (column, depth) outputs is whole columns (ndepth 32, 64
                                or 128; tf_order 2; nembeds <= 8; other shapes run the four-launch form), their sums go from the
                                reduction straight into the sampler's layout; the prior band of the columns is precomputed
                                (prior_band_kernel, only when Tau2 / lam2 changed) and loaded at kernel start.
                                2: also the batched K x K solve of _resample_W (factor.py:349-362), by "owner" workgroups of the W
                                accumulation launch that wait for their tile's streaming workgroups: two launches.
                                All three walk the same chain bit for bit (tests/test_gpu_fused.py).  Measured on MI355X at
                                (512,256,64) nembeds 5: 41.2 / 38.7 / 42.2 us per step. */
  BTF_OPT_FUSED_DATAFLOW = 8,/* 1 (default): where the fused V launch of BTF_OPT_FUSED_STEP >= 1 allows it - one chunk per tile, the
                                precomputed prior band, nembeds <= 6 - its tail is the BARRIER-FREE form (csrc/btf_fused.h,
                                v_fused_df): the chain wave of a column factors A_k = g_k I + P_j (fast_mvn.py:38) the moment its own
                                stream ends, while the other waves still stream, reduce and rotate; LDS counters order the
                                stages.  Bit-identical to the barrier tail and to the four-launch form.  Measured at
                                (512,256,64) nembeds 5: 20.4 us per V launch against 21.0.  0: the barrier tail. */
  BTF_OPT_SPLIT_ACCUM = 5,   /* sharded Gaussian contexts: 1 - the streaming accumulation of a half-sweep runs in two launches: the
                                chunks that reduce over this rank's OWN block of the fixed factor (its columns of V for the W
                                half-sweep, its rows of W for the V half-sweep) are queued right behind the kernel that drew
                                that block - they need no exchange - and the remaining chunks by the next half-sweep call, behind
                                the all-gather (btf_comm_fork / btf_comm_join order a communication stream against them).
                                Same partial sums, same results.  0 (default): one launch.                              */
  BTF_OPT_PG_EXACT = 3       /* Polya-Gamma sampler of btf_pg_draw (what pypolyagamma's pgdrawv does at factor.py:459).
                                0 (default): every integer trial count up to 32 by Devroye's exact alternating-series
                                sampler, summed b times, as pypolyagamma does (flat per-lane work-queue kernels: f32
                                squeeze, f64 decisions inside the guard bands); larger and non-integer counts by the
                                sum-of-gammas series with a moment-matched remainder (approximate; validated against the
                                exact sampler).  Count data (btf_set_data_counts: pseudo-trial counts sum(y) + n r, integers
                                only by accident): the series for every cell.
                                1: exact for every count below 200 - floor(b) Devroye draws plus a 128-term series for a
                                fractional part.
                                2: the series for every count (opt-in; the round-2 default for counts >= 3).
                                Counts >= 200: moment-matched normal in every mode.                             */
};
enum {
  BTF_SAMPLER_BANDED = 0,   /* block-banded LDL' in the declared elimination order (btf_get_V_order):
                               twisted two-chain kernel where it applies; default                     */
  BTF_SAMPLER_SPECTRAL = 1, /* complete Gaussian data only: rotate by the eigenvectors of the shared
                               K x K likelihood block, K scalar banded chains per column (falls back to
                               BANDED for weighted data); square root documented at btf_resample_V     */
  BTF_SAMPLER_CHAIN = 2,    /* single chain, depth-major order                                         */
  BTF_SAMPLER_GENERIC = 3,  /* any-size kernel (band in LDS or HBM scratch), depth-major order         */
  BTF_SAMPLER_BANDED_NOPANEL = 4 /* BANDED without the panelised MFMA factorisation (A/B aid)          */
};

/* ---- lifetime ------------------------------------------------------------
 * Replaces the array allocations of BayesianTensorFiltering.__init__
 * (factor.py:24-110) on the device side.  `stream` is a hipStream_t to run on
 * (e.g. torch.cuda.current_stream().cuda_stream) or NULL for a private one. */
int btf_create(btf_ctx** out, int nrows, int ncols, int ndepth, int nembeds,
               int tf_order, int device, void* stream);
void btf_destroy(btf_ctx* ctx);
const char* btf_last_error(const btf_ctx* ctx);
int btf_fail_index(const btf_ctx* ctx);

/* ---- sharding (multi-GPU) --------------------------------------
 * This ctx updates rows [row0,row0+nrows_local) in the W half-sweep and columns
 * [col0,col0+ncols_local) in the V half-sweep.  Must precede btf_set_data_*.
 * Default: everything.  W and V are replicated: after each half-sweep the updated block is
 * all-gathered - btf_allgather_W / btf_allgather_V below (the ctx's own RCCL communicator), or by
 * the caller on the device pointers btf_dev_W / btf_dev_V. */
int btf_set_shard(btf_ctx* ctx, int row0, int nrows_local, int col0, int ncols_local);
/* compat=REFERENCE in a sharded run (compat caveat).  The cached likelihood weights of quirks Q1/Q2 come from
 * a SOURCE row / column (factor.py:320,349: row nembeds-1 for every later row when the data hold no NaN;
 * factor.py:394-401: the last column at which the NaN pattern changed), which may lie outside the rank's blocks.  With
 * contiguous blocks at most ONE such row and ONE such column exist per rank (the row sources are nembeds-1 or the row
 * itself; the column sources never decrease and are a column's own index or its left neighbour's source).  Declare them
 * here (global indices, -1: none), after btf_set_shard and before the upload: every row slab handed to
 * btf_set_data_* / btf_set_omega then has nrows_local + 1 rows, the source row LAST, every column slab
 * ncols_local + 1 columns, the source column last (btf_set_data_counts keeps taking the whole tensor).  The halo is
 * never updated and never enters a sum (nobs, SSE); only its weights are read - by the stale-weight accumulation
 * (btf_set_stale_sources maps a source equal to the halo onto its slot) - and btf_pg_draw draws them from the streams
 * of the source's global cells, i.e. the values the owning rank draws.  A changed halo invalidates the uploaded data. */
int btf_set_shard_halo(btf_ctx* ctx, int halo_row, int halo_col);
void* btf_stream(btf_ctx* ctx); /* the hipStream_t every step function of this ctx enqueues on (the one passed
                                  to btf_create, or the private one): collectives on the ctx's buffers and event
                                  timing must be ordered against it */
/* Ordering a communication stream (the one the all-gathers of W / V are issued on) against the ctx's stream:
 * btf_comm_fork makes `comm_stream` wait for the kernel that drew this rank's block in the last half-sweep (not for work
 * queued behind it: BTF_OPT_SPLIT_ACCUM); btf_comm_join makes the ctx's stream wait for everything queued on
 * `comm_stream` so far.  hipStream_t handles; no host synchronisation. */
int btf_comm_fork(btf_ctx* ctx, void* comm_stream);
int btf_comm_join(btf_ctx* ctx, void* comm_stream);
/* ---- the ctx-owned communicator ("ctx owns device buffers, streams, RCCL communicators"; 8(e)) -----------
 * The exchange of a sharded run - given V the rows of W are independent (factor.py:333), given W the columns of V are
 * (factor.py:378), so each half-sweep ends in ONE all-gather of the freshly drawn blocks - issued by the library
 * itself: ncclAllGather / ncclAllReduce of RCCL, in place on the ctx's own W / V / scalar buffers and on the ctx's own
 * stream (stream-ordered, no host synchronisation).  RCCL is bound at run time (csrc/btf_comm.h): a process that never
 * calls these never maps it.  One process per GPU, one ctx per process and communicator.
 *
 *   rank 0:      btf_comm_unique_id(id, BTF_COMM_ID_BYTES)       (ncclGetUniqueId)
 *   the caller:  hands the 128 bytes to every rank over any channel it has (MPI, a file, a socket, torch.distributed)
 *   every rank:  btf_set_shard(blocks of btf_comm_block) ... btf_comm_init(ctx, rank, world, id, BTF_COMM_ID_BYTES)
 *   every sweep: btf_resample_W -> btf_allgather_W -> btf_resample_V -> btf_allgather_V
 *
 * btf_comm_block: the block [lo, lo + len) of an axis of n rows / columns that rank `rank` of `world` owns - equal
 * chunks ceil(n / world), the tail ranks short or empty (the only decomposition one in-place all-gather reassembles;
 * W and V are allocated with 64 spare rows / columns for it, hence world <= 64).  The all-gathers return BTF_ESTATE
 * if btf_set_shard was given anything else.
 * btf_allgather_W / _V: all-gather of the block this rank drew in the last half-sweep.  With BTF_OPT_SPLIT_ACCUM the
 * gather runs on a communication stream of the ctx, ordered by btf_comm_fork / btf_comm_join as described there.
 * btf_allreduce_sse: sums the device scalar HYP_SSE (slot 4 of btf_dev_hyp) over the ranks - the one exchange of a
 * sharded nu2 draw (btf_draw_scalars which | 8, this call, which | 16).
 * btf_allreduce_sum: sums n <= 16 host doubles over the ranks, in place (observation counts at set-up, the host-side
 * nu2 draw's residual sum of squares).  Synchronises.
 * btf_comm_rehearse: a timing aid - ONE process plays rank `rank` of `world`: a one-rank communicator, and every
 * all-gather moves the full gathered message (world x chunk doubles) from one scratch buffer to another, so the step
 * pays RCCL's call and the message's bytes but the blocks of the other ranks are never refreshed: not a sampler.
 * btf_comm_info: out[8] = {has communicator, its rank, its size, gather rank, gather world, rehearsal, RCCL version
 * code, ncclCommCount}; out[0] is 2 for the peer-window transport below.
 *
 * The peer-window transport (csrc/btf_comm.h) runs the same four collectives without RCCL: every rank maps the W / V
 * buffers and a mailbox of every other rank (hipIpc handles: processes of one node whose GPUs can map each other) and one
 * kernel per collective stores this rank's block straight into the peers' buffers, flags it and waits for theirs - one
 * launch, two flag round trips.  It is also the only device-side transport for several ranks on ONE GPU.
 *   every rank:  btf_peer_export(ctx, desc, BTF_PEER_DESC_BYTES)
 *   the caller:  all-gathers the descriptors over its channel (MPI_Allgather, torch.distributed)
 *   every rank:  btf_peer_init(ctx, rank, world, descs, world * BTF_PEER_DESC_BYTES)       (descs rank-major)
 * after which btf_allgather_W / _V, btf_allreduce_sse and btf_allreduce_sum use it (btf_comm_info out[0] == 2), under
 * the rule of every collective library: all ranks issue the same sequence of collectives.  Sums are taken in rank order
 * on every rank (identical bits everywhere).  A peer that never arrives turns, after BTF_PEER_TIMEOUT_MS (default
 * 20000), into BTF_EHIP at the next step's status check - not into a hung GPU.  btf_comm_destroy tears it down.          */
#define BTF_COMM_ID_BYTES 128
#define BTF_PEER_DESC_BYTES 256
int btf_peer_export(btf_ctx* ctx, unsigned char* desc, int nbytes);
int btf_peer_init(btf_ctx* ctx, int rank, int world, const unsigned char* descs, int nbytes);
int btf_comm_unique_id(unsigned char* id, int nbytes);
int btf_comm_block(int n, int rank, int world, int32_t* lo, int32_t* len);
int btf_comm_init(btf_ctx* ctx, int rank, int world, const unsigned char* id, int nbytes);
int btf_comm_rehearse(btf_ctx* ctx, int rank, int world);
int btf_comm_destroy(btf_ctx* ctx);
int btf_comm_info(btf_ctx* ctx, int32_t* out);
int btf_allgather_W(btf_ctx* ctx);
int btf_allgather_V(btf_ctx* ctx);
int btf_allreduce_sse(btf_ctx* ctx);
int btf_allreduce_sum(btf_ctx* ctx, double* vals, int n);
/* btf_set_W / btf_set_V for an exchange staged through the host: the caller's own block is unchanged, so chunks already
 * accumulated from it (BTF_OPT_SPLIT_ACCUM) stay valid. */
int btf_set_gathered_W(btf_ctx* ctx, const double* W);
int btf_set_gathered_V(btf_ctx* ctx, const double* V);
void* btf_dev_W(btf_ctx* ctx); /* device double[N][K]    */
void* btf_dev_V(btf_ctx* ctx); /* device double[M][T][K] */

/* ---- data -----------------------------------------------------------------
 * Hoists what factor.py:329-330 and :374-375 recompute every half-sweep
 * (replicate counts and NaN-means) into one pass.  y_rows is the row slab
 * Y[row0:row0+nrows_local] of shape (nrows_local,M,T,R); y_cols is the column
 * slab Y[:,col0:col0+ncols_local] of shape (N,ncols_local,T,R).  Unsharded:
 * pass the same pointer twice.                                               */
int btf_set_data_gaussian(btf_ctx* ctx, const double* y_rows, const double* y_cols, int nreps);
/* Binomial data tuple (Y, N) of factor.py:437-460; same slab convention, R=1. */
int btf_set_data_binomial(btf_ctx* ctx, const double* succ_rows, const double* trials_rows,
                          const double* succ_cols, const double* trials_cols);
/* Quirks Q1/Q2: which row / column the cached likelihood weights
 * come from in compat=REFERENCE.  src_row[N], src_col[M] (global indices);
 * NULL = identity.  Only consulted by the weighted kernels.  Call after
 * btf_set_data_*; a source outside this ctx's shard that is not its declared halo
 * (btf_set_shard_halo) is BTF_EINVAL.                                           */
int btf_set_stale_sources(btf_ctx* ctx, const int32_t* src_row, const int32_t* src_col);

/* ---- Negative-Binomial counts (unsharded contexts) -----------------
 * counts: (N,M,T,nreps) C-order, NaN = missing; kept on the device.  The model is the
 * Binomial one on pseudo-data Y = sum_r y_r, N = sum_r (y_r + R) (factor.py:494-511, :552):
 * btf_nb_set_rate builds those (both layouts) from the rate R; afterwards btf_pg_draw /
 * btf_resample_W / btf_resample_V run as for btf_set_data_binomial.
 * shared[d] != 0: one R value is shared along dim d of (rows, cols, depth) (the reference's
 * `rdims`); R and cand are C-contiguous over the unshared dims.
 * btf_nb_loglik replaces the data-sized part of one random-walk MH step of
 * NegativeBinomialBTF._resample_R (factor.py:533-541): ll[e] = sum over replicates and shared
 * dims of lgamma(y+cand)-lgamma(cand)-lgamma(y+R)+lgamma(R)+(cand-R) log(1-p),
 * p = ilogit(clip(w.v,-10,10)), NaN observations dropped.  Synchronises.  When R is shared
 * along (cols, depth) and every count is an integer < 1024 the sum is evaluated from per-row
 * count histograms (built once at upload) and one pass per sweep for sum cnt*log(1-p): an MH
 * step then reads 8 KB per row instead of the count tensor.
 * Sharded contexts (btf_set_shard) pass the WHOLE (nrows, ncols, ndepth, nreps) tensor as well: the rate
 * update is a function of all of it and every rank computes it (same streams, same result), while
 * the augmented Binomial model is kept for the rank's row and column slabs only.              */
int btf_set_data_counts(btf_ctx* ctx, const double* counts, int nreps);
int btf_nb_loglik(btf_ctx* ctx, const double* R, const double* cand, const int32_t* shared3, double* ll);
int btf_nb_set_rate(btf_ctx* ctx, const double* R, const int32_t* shared3);
/* rng="device": the whole MH loop of NegativeBinomialBTF._resample_R (nsteps random-walk steps,
 * proposals and accept/reject from Philox) plus the pseudo-data rebuild, without a host round
 * trip.  R_in: start value (NULL: continue from the rate on the device).  Needs the histogram
 * form (see btf_nb_loglik); BTF_ESTATE otherwise.  btf_nb_get_rate fetches R (synchronises).   */
int btf_nb_mh(btf_ctx* ctx, uint64_t seed, int nsteps, double rpropstdev, double rstdev, const int32_t* shared3,
              const double* R_in);
int btf_nb_get_rate(btf_ctx* ctx, double* R, const int32_t* shared3);

/* ---- state ---------------------------------------------------------------- */
int btf_set_W(btf_ctx* ctx, const double* W);            /* (N,K)      */
int btf_get_W(btf_ctx* ctx, double* W);
int btf_set_V(btf_ctx* ctx, const double* V);            /* (M,T,K)    */
int btf_get_V(btf_ctx* ctx, double* V);
/* Tau2 NULL: keep the device copy (only lam2 / sigma2 move).  A non-NULL Tau2, or a lam2 that differs from the last
 * one, invalidates the precomputed prior band of the V half-sweep (prior_band_kernel runs again before the next one):
 * a host loop that re-sends unchanged hyper-parameters every sweep should pass NULL.  Reading the scalars back
 * (btf_get_scalars) invalidates nothing.                                                                       */
int btf_set_hyper(btf_ctx* ctx, const double* Tau2 /* (M,nD) */, double lam2, double sigma2);
/* Horseshoe+ local scales on the device (rng="device" only - the draws
 * come from Philox, not from the legacy numpy stream).  btf_set_tau_chain uploads the three
 * auxiliary levels Tau2_a/b/c (M,nD); btf_resample_Tau2 replaces the per-column loop of
 * BTF._resample_Tau2 (factor.py:134-141) for all columns at once, updating Tau2 and the chain
 * in place on the device, and returns lsum[j] = sum_r dsq[j,r]/Tau2_new[j,r] (the per-column
 * terms of the lam2 rate, factor.py:148-150) when lsum_out != NULL (synchronises).  With
 * device-resident scalars enabled the lam2 argument is ignored.                             */
int btf_set_tau_chain(btf_ctx* ctx, const double* Tau2_a, const double* Tau2_b, const double* Tau2_c);
int btf_get_tau(btf_ctx* ctx, double* Tau2, double* Tau2_a, double* Tau2_b, double* Tau2_c); /* a,b,c may be NULL */
int btf_resample_Tau2(btf_ctx* ctx, uint64_t seed, double lam2, double stability, double* lsum_out /* (M) or NULL */);
/* Device-resident scalars only: the NEXT btf_w_accum launch also runs this horseshoe+ update, as side workgroups
 * beside the stream (Tau2 | V, lam2 does not depend on nu2 / sigma2: the same conditionals as btf_resample_Tau2 after
 * the scalar draws; same Philox streams).  The lam2-rate terms stay on the device for btf_draw_lam2. */
int btf_queue_Tau2(btf_ctx* ctx, uint64_t seed, double stability);
/* ... and the NEXT drawing btf_draw_scalars launch also draws lam2 | rest (a second workgroup of that launch; the
 * lam2-rate terms must be on the device by then: a Tau2 update queued with btf_queue_Tau2, or btf_resample_Tau2). */
int btf_queue_lam2(btf_ctx* ctx, uint64_t seed, int compat);
/* (if no btf_draw_scalars launch takes it, the V accumulation launch of the next btf_resample_V carries it as a side workgroup)
 * nu2 | rest (which & 1) and sigma2 | rest (which & 2) as a side workgroup of the NEXT W accumulation launch (btf_w_accum /
 * btf_resample_W): possible on complete Gaussian data, unsharded, when the spectral V sampler of the previous half-sweep left
 * the per-column residual parts behind (BTF_OPT_FUSED_SWEEP) and nothing has touched W or V since.  *queued = 1: queued, the
 * W solve of this sweep will read the new values; 0: not possible now - draw them with btf_draw_scalars.  Replaces the
 * residual pass of factor.py:411-416 and the two gamma draws of genlasso.py:160-164 / factor.py:130-132. */
int btf_queue_scalars(btf_ctx* ctx, uint64_t seed, int which, double nu2_a, double nu2_b, double sigma2_a, double sigma2_b,
                      int32_t* queued);

/* n whole Gibbs sweeps (nu2, sigma2, Tau2 chain, lam2, W, V: GaussianBTF.resample, factor.py:306-311 over :112-128)
 * queued by one call: scalar-noise Gaussian data, device-resident scalars, unsharded.  Seeds: sweep s uses
 * seed_base + draws0 + 5 s + k, k = 1..5 for (Tau2, lam2, nu2/sigma2, W, V) - the sequence the Python driver
 * consumes (functionalmf_amd/factor.py:_next_seed), so both drivers walk the same chain.  Nothing is read back;
 * errors of the factorisations surface at the next btf_sync. */
int btf_gibbs_sweeps(btf_ctx* ctx, int nsweeps, uint64_t seed_base, uint64_t draws0, int compat, double nu2_a, double nu2_b,
                     double sigma2_a, double sigma2_b, double stability, double eps0, int attempts);
/* n W+V updates - _resample_W then _resample_V (factor.py:313-409), device normals - queued by ONE call: what a host loop
 * of btf_resample_W(NULL, seed_base + draws0 + 2 s + 1, compat); btf_resample_V(NULL, seed_base + draws0 + 2 s + 2, ...)
 * over s = 0 .. n-1 does, without a round trip through the caller between the launches (the same launches, the same
 * seeds: the same chain).  Unsharded contexts; no host synchronisation.                                              */
int btf_wv_steps(btf_ctx* ctx, int n, uint64_t seed_base, uint64_t draws0, int compat, double eps0, int attempts);
int btf_set_nu2(btf_ctx* ctx, double nu2);               /* Gaussian scalar noise variance */
/* Device-resident scalar hyper-parameters (rng="device" only, unsharded
 * contexts).  After btf_device_scalars(ctx,1) the half-sweep, prior-band and Tau2 kernels read
 * nu2 (Gaussian data), sigma2 and lam2 from a small device array instead of the host copies, and
 *   btf_draw_scalars: nu2 | rest (which&1; runs the residual reduction of factor.py:411-416 and the
 *                     inverse-gamma draw of genlasso.py:160-164) and sigma2 | rest (which&2;
 *                     factor.py:130-132) - priors InvGamma(a,b);
 *   btf_draw_lam2:    lam2, lam2_a | rest (factor.py:143-153; compat selects quirk Q3's rate), from
 *                     the per-column sums the preceding btf_resample_Tau2 left on the device
 * draw them there from Philox streams, so that a full Gibbs sweep queues without a host round
 * trip.  btf_set_scalars uploads values, btf_get_scalars (synchronises) returns
 * {nu2, sigma2, lam2, lam2_a, SSE of the last nu2 draw, sum W^2 of the last sigma2 draw}.      */
/* Sharded contexts: the residual sum of squares of nu2 | rest is a sum over the ranks' row slabs.  btf_draw_scalars
 * with which | 8 only reduces this rank's share into the device scalar out6[4] (btf_dev_hyp() + 4 doubles); the caller
 * all-reduces that one double over the ranks (RCCL, on btf_stream) and calls btf_draw_scalars with which | 16, which
 * draws from it - every rank the same value (same seed).  btf_set_global_nobs: the observation count over all
 * ranks (a constant of the data set).  sigma2, Tau2, lam2 need no exchange: W and V are replicated.            */
void* btf_dev_hyp(btf_ctx* ctx);
int btf_set_global_nobs(btf_ctx* ctx, double nobs);
int btf_set_scalar_slot(btf_ctx* ctx, int slot, double value);  /* one entry of the btf_get_scalars array (staged exchanges) */
int btf_device_scalars(btf_ctx* ctx, int enable);
int btf_set_scalars(btf_ctx* ctx, double nu2, double sigma2, double lam2, double lam2_a);
int btf_get_scalars(btf_ctx* ctx, double* out6);
int btf_draw_scalars(btf_ctx* ctx, uint64_t seed, int which, double nu2_a, double nu2_b, double sigma2_a, double sigma2_b);
int btf_draw_lam2(btf_ctx* ctx, uint64_t seed, int compat);
int btf_set_omega(btf_ctx* ctx, const double* omega_rows, const double* omega_cols); /* Binomial: PG draws, slabs as data */
int btf_get_omega(btf_ctx* ctx, double* omega_rows);     /* (nrows_local,M,T) */

/* ---- half-sweeps ----------------------------------------------------------
 * btf_resample_W replaces the body of GaussianBTF._resample_W (factor.py:313-362)
 * [and BinomialBTF._resample_W :437-440]: for every local row i, d=min(i+1,K):
 *   Q_i = sum_{j,t} c_ijt v_jt v_jt' + I/sigma2,  m_i = sum c_ijt ybar_ijt v_jt,
 *   W[i,:d] = Q_i^-1 m_i + L_i^-T z_i.
 * z: host array of the sum_i min(i+1,K) standard normals of ALL rows in row
 * order (the legacy-RNG stream of factor.py:361), or NULL to draw them on the
 * device (Philox4x32-10 keyed by `seed`).                                     */
/* btf_w_accum queues the first phase of btf_resample_W alone (the streaming accumulation, which
 * depends on the data and on V only); the next btf_resample_W then goes straight to the solve.
 * Between the two, btf_draw_scalars(which | 4) can take the residual sum of squares of nu2 | rest
 * from the same partials instead of a pass of its own over the data.                          */
int btf_w_accum(btf_ctx* ctx, int compat);
int btf_resample_W(btf_ctx* ctx, const double* z, uint64_t seed, int compat);
/* btf_resample_V replaces GaussianBTF._resample_V (factor.py:364-409) including
 * its call into sample_mvn_from_precision (fast_mvn.py:35-47, :62-68): for every
 * local column j the (K*T)x(K*T) precision  kron(W,I)'C kron(W,I) + I_K (x)
 * Delta' diag(1/(lam2 Tau2_j)) Delta  is formed in depth-major order (t,k), factored
 * as a block-banded LDL' in the declared elimination order P (btf_get_V_order), and
 * V[j] = Q^-1 mu + P' L^-T z  is drawn.
 * z: host (M, K*T) normals, row j for column j, indexed in the factor's
 * elimination order, or NULL for device Philox.  eps0/attempts: the
 * force_psd jitter schedule (eps0*10^a added cumulatively, <= attempts).      */
int btf_resample_V(btf_ctx* ctx, const double* z, uint64_t seed, int compat,
                   double eps0, int attempts);
int btf_get_V_attempts(btf_ctx* ctx, int32_t* tries /* (ncols_local) */);
/* The elimination order P the V half-sweep of this context uses (after btf_set_data_*):
 * order[i] = depth-major index t*K+k of the i-th pivot, i = 0..K*T-1.  Identity for the
 * single-chain kernels; for the default twisted kernel: depths 0..ts-1 ascending, then depths
 * T-1..ts+tf+1 descending (k descending), then the separator depths ts..ts+tf.  z[j][i] is
 * the normal that multiplies pivot i (this is CHOLMOD's P() in fast_mvn.py:44).            */
int btf_get_V_order(btf_ctx* ctx, int32_t* order /* (K*T) */);
/* BTF_SAMPLER_SPECTRAL (complete Gaussian data: every depth of every column has the same K x K
 * likelihood block G = (R/nu2) W'W, factor.py:396-398 with constant weights).  In the reference's own
 * k-major ordering of the unknowns (factor.py:409) Q_j = G (x) I_T + I_K (x) P_j with P_j the prior
 * precision of factor.py:404-405; with G = U diag(g) U' the draw is
 *   V[j] = Q_j^-1 mu + (U (x) I_T) blockdiag_k(L_k^-T D_k^-1/2) z,   g_k I + P_j = L_k D_k L_k',
 * z[j][k*T + t] multiplying pivot t of system k - a square root of Q_j^-1 like fast_mvn.py:44's
 * P' L^-T, same mean term, same jitter schedule (the shift is added to every g_k).  U: eigenvectors in
 * ascending order of eigenvalue, each with its largest-magnitude entry positive.
 * btf_get_V_sampler reports the BTF_SAMPLER_* the next V half-sweep of this context will run.  */
/* The eigen-solver the spectral sampler uses, stand-alone (one-wave cyclic Jacobi, K <= 10): sums `nparts`
 * packed-lower K x K matrices parts[p][r(r+1)/2 + c], returns out[0..K-1] eigenvalues ascending,
 * out[K + r*K + c] component r of eigenvector c (largest-magnitude entry positive), out[K+K*K] sweeps. */
int btf_sym_eig(int device, int nembeds, int nparts, const double* parts, double* out, const double* warm_from);

/* Measurement aid: rate (GB/s) of a plain streaming read of `bytes` of device memory, averaged over `reps`
 * launches - the practical read ceiling bench.py reports beside the spec peak (roofline.read_ceiling_GBs). */
int btf_read_probe(int device, size_t bytes, int reps, double* gb_per_s);
/* Which form of the likelihood part the half-sweeps of this context run (after btf_set_data_*):
 *   BTF_LIK_COMPLETE      complete Gaussian data: K sums per cell, one shared Gram (factor.py:347-348, :392-393)
 *   BTF_LIK_WEIGHTED      per-cell weights (missing replicates, Polya-Gamma omegas): K + K(K+1)/2 sums per cell
 *   BTF_LIK_CURVE_COUNTS  replicate counts constant along the depth axis: the complete-data stream plus per-row /
 *                         per-column corrections (BTF_OPT_CURVE_COUNTS) - the same conditionals as the weighted form */
enum { BTF_LIK_COMPLETE = 0, BTF_LIK_WEIGHTED = 1, BTF_LIK_CURVE_COUNTS = 2 };
int btf_get_likelihood_form(btf_ctx* ctx, int32_t* form);
/* Checkpoint / resume (the reference keeps no state between runs - genlasso.py:57-66 returns the samples; SURVEY
 * section 5): besides the caller's seeds, the rng="device" draws of W and V are keyed by how many W / V half-sweeps
 * this context has run.  A chain continued in another context gets the same draws after handing these two over.
 * btf_set_draw_counters also forgets the spectral sampler's eigen warm start (the next V half-sweep solves the K x K
 * eigenproblem from scratch), so a chain that calls it with its own counters and one restored elsewhere continue
 * identically. */
int btf_get_draw_counters(btf_ctx* ctx, uint64_t* w_half_sweeps, uint64_t* v_half_sweeps);
int btf_set_draw_counters(btf_ctx* ctx, uint64_t w_half_sweeps, uint64_t v_half_sweeps);
/* Algorithmic bytes per cell one accumulation launch streams for the bound data: 8 (linear statistic alone: complete data,
 * curve counts), 9 (+ replicate counts as bytes; or Binomial pseudo-data kappa = Y - N/2 as bytes + f64 weights when
 * the counts are integers up to 127), 16 (f64 statistic + f64 weights).  bench.py's roofline uses it. */
int btf_get_accum_bytes_per_cell(btf_ctx* ctx, double* bytes);

/* warm_from: NULL (cyclic Jacobi from the identity) or the `out` of a nearby matrix, refined by the
 * Ogita-Aishima iteration (the path the sampler takes from the second sweep on; out[K+K*K] = 0 then). */
int btf_set_option(btf_ctx* ctx, int option, int value);
int btf_get_V_sampler(btf_ctx* ctx, int32_t* which);

/* Residual sum of squares and observation count over the LOCAL rows: the two
 * numbers GaussianBTF._resample_nu2 (factor.py:411-416, genlasso.py:157-160)
 * reduces the whole tensor for.  Synchronises.                                */
int btf_sse(btf_ctx* ctx, double* sse, double* nobs);
/* The same in two halves: btf_sse_begin enqueues the reduction (and a copy of W) without
 * waiting; btf_sse_end synchronises, returns the two numbers and, if W_out != NULL, the
 * (N,K) factor as it was when btf_sse_begin was called.  Lets the host queue a whole sweep. */
int btf_sse_begin(btf_ctx* ctx);
int btf_sse_end(btf_ctx* ctx, double* sse, double* nobs, double* W_out);
/* omega_ijt ~ PG(Ntrials_ijt, w_i . v_jt) on the device: replaces the
 * pypolyagamma call at factor.py:459 (own RNG: Philox keyed by seed).         */
int btf_pg_draw(btf_ctx* ctx, uint64_t seed);

/* Stand-alone batch of PG(b_i, psi_i) draws from the same device sampler (used to
 * validate its distribution; element i uses the Philox stream (seed, i)).      */
int btf_pg_batch(int device, int64_t n, const double* b, const double* psi, uint64_t seed, double* out);
/* the same with the sampler chosen by `mode`: 0 / 1 / 2 as BTF_OPT_PG_EXACT; 3: the f64 Devroye sampler for every
 * count below 200 (the round-2 exact kernel: the reference the flat sampler is tested against); 4: as 1 with every
 * trip of the flat sampler repeated in f64 (validation of its fallback: same decisions, same draws up to f32 rounding) */
int btf_pg_batch_mode(int device, int64_t n, const double* b, const double* psi, uint64_t seed, int mode, double* out);

int btf_sync(btf_ctx* ctx); /* waits; returns BTF_ENOTPD if a step failed since the last sync */
/* Host-only self-test of the index arithmetic behind the kernels' tables, LDS layouts, elimination orders and chunk maps
 * (no GPU, no HIP call): 0, or the source line of the first failed check.  scripts/asan_host.sh runs it on a build of the
 * host side with -fsanitize=address,undefined (the reference has no native code to sanitise; SURVEY section 5 aux). */
int btf_host_selftest(void);

/* ---- structured-Gaussian sampler (the fast_mvn equivalent, stand-alone) ----
 * Batched draw  x_b = Q_b^-1 mu_b + L_b^-T z_b,  L_b L_b' = Q_b (+ jitter), for
 * symmetric banded precisions given as lower band by columns:
 * band[b][c][a] = Q_b[c+a, c], a = 0..bw.  mu_part and/or z may be NULL (zero /
 * device Philox).  Replaces sample_mvn_from_precision(Q, mu_part=..., sparse=True)
 * (fast_mvn.py:10-74) for banded Q in the given ordering.                     */
int btf_mvn_banded(int device, int batch, int n, int bw, const double* band,
                   const double* mu_part, const double* z, uint64_t seed,
                   double eps0, int attempts, double* x_out, int32_t* tries_out);

/* Dense forms of fast_mvn (the sparse=False branches: sample_mvn_from_precision fast_mvn.py:49-60,
 * sample_mvn_from_covariance fast_mvn.py:126-142, reached through the dispatcher sample_mvn :145-179).
 * A: batch row-major n x n matrices (only the lower triangle is read), n <= 1024.
 *   form & BTF_MVN_PRECISION: A is a precision Q:  x = Lt^-1 z + Q^-1 mu_part   (or + mu),  L L' = Q
 *   otherwise A is a covariance S:                 x = L z + S mu_part          (or + mu),  L L' = S
 *   form & BTF_MVN_FACTOR: A already IS the lower Cholesky factor L (chol_factor=True)
 * mu / mu_part: [batch][n] or NULL (at most one).  z: [batch][n] standard normals (the reference draws
 * np.random.normal(size=n) after the factorisation) or NULL: Philox(seed).  Not positive definite: eps0 is added to
 * the diagonal cumulatively, x10 per retry, at most `attempts` times (fast_mvn.py:62-68, :133-139); then
 * BTF_ENOTPD with the batch index in the message (the reference's dense branches raise LinAlgError / warn forever).
 * tries_out: [batch] retries used, or NULL. */
enum { BTF_MVN_PRECISION = 1, BTF_MVN_FACTOR = 2 };
int btf_mvn_dense(int device, int batch, int n, const double* A, int form, const double* mu, const double* mu_part,
                  const double* z, uint64_t seed, double eps0, int attempts, double* x_out, int32_t* tries_out);

/* ---- elliptical slice sampling for non-conjugate likelihoods -----------------
 * Replaces NonconjugateBayesianTensorFiltering._resample_W / _resample_V (factor.py:567-590): a prior draw nu
 * (sample_mvn_from_precision on the packed prior precision of factor.py:155-195 - for V the banded sampler with
 * the likelihood switched off, same declared order as btf_resample_V with BTF_SAMPLER_BANDED) and the slice loop
 * of elliptical_slice_ (elliptical_slice.py:59-124).  The likelihood callback of the reference becomes a device
 * likelihood over the statistics btf_set_data_gaussian hoisted (counts y, NaN = missing):
 *   link 0  Poisson, log link:       sum_cells  S1 (w.v) - cnt exp(w.v)
 *   link 1  Poisson, identity link:  sum_cells  S1 log(w.v) - cnt (w.v),  -inf where w.v <= 0
 *   link 2  Bernoulli / Binomial, logit link (S1 successes of cnt trials):  S1 (w.v) - cnt log(1 + exp(w.v))
 *   link 3  Gaussian, identity link, known variance s2 (parameter 1/s2):     (S1 (w.v) - cnt (w.v)^2 / 2) / s2
 *   link 4  Negative-Binomial, logit link, known rate r (parameter r):      S1 (w.v) - (S1 + cnt r) log(1 + exp(w.v))
 * - every one a function of the hoisted statistics S1 = sum_r y, cnt = observed replicates; the state-independent terms
 * (- sum lgamma(y+1), the Gaussian's - sum y^2 / 2 s2 - n log(2 pi s2) / 2, the Negative-Binomial's log binomial
 * coefficients) are left to the caller.  Parameters: btf_set_likelihood_param.  what: 0 = W, 1 = V.  Unsharded contexts.
 *
 * Host-driven form (rng="host": the uniforms come from the caller's generator, so a seeded chain walks the
 * reference's path): btf_ess_begin saves the current state x0 and draws nu (z: the normals of
 * sample_mvn_from_precision, W: sum_i min(i+1,K), V: (M, K*T) in the declared order; NULL = device Philox);
 * btf_ess_eval(theta, current=0) sets the state to x0 cos(theta) + nu sin(theta) and returns its log-likelihood
 * (current=1: of the state as it stands); when the caller stops, the state holds the last proposal, exactly what
 * elliptical_slice_ returns.  Synchronises.
 * link = BTF_ESS_HOST_LIKELIHOOD (-1): the likelihood is the CALLER's (the reference's Python callback
 * `loglikelihood(W, V, data)`, factor.py:567-612): btf_ess_begin draws nu as above (it needs no data on the device),
 * btf_ess_eval only forms the proposal in W (resp. V) - the caller reads it back (btf_get_W / btf_get_V), evaluates its
 * function and decides; *ll = 0.
 *
 * Device-driven form: btf_ess_run = begin + at most max_rounds proposal / likelihood / decision rounds queued on the
 * stream, uniforms from Philox, nothing read back.  mode 0: one slice over all of W (resp. V), as the reference;
 * mode 1: one slice per row of W (resp. per column of V) - the rows are conditionally independent given V - all
 * brackets shrinking in lockstep, one proposal per row per round.  btf_ess_info (synchronises): chains that used up
 * max_rounds (they keep the current state) and the log-likelihood the first chain ended on.                      */
#define BTF_ESS_HOST_LIKELIHOOD (-1)
int btf_set_likelihood_param(btf_ctx* ctx, int link, double parameter);   /* links 3 (1 / variance) and 4 (rate); > 0 */
int btf_ess_begin(btf_ctx* ctx, int what, const double* z, uint64_t seed, double eps0, int attempts);
int btf_ess_eval(btf_ctx* ctx, int what, double theta, int current, int link, double* ll);
int btf_ess_run(btf_ctx* ctx, int what, int link, int mode, const double* z, uint64_t seed, int max_rounds,
                double eps0, int attempts);
int btf_ess_info(btf_ctx* ctx, int32_t* unfinished, double* ll_first);

/* ---- generalized analytic slice sampling (gass.py:13-130) for the constrained non-conjugate model
 * (ConstrainedNonconjugateBayesianTensorFiltering._resample_W / _resample_V, factor.py:665-855): every row of W (what = 0)
 * or every column of V (what = 1) is one chain, all chains advance together.  Likelihood: the device likelihoods
 * of btf_ess_* (links 0..4) on the statistics of btf_set_data_gaussian.  Unsharded contexts.
 *
 * btf_gass_set_constraints: cons [J][T+1], row q = (Cons_q, bound_q): every curve tau_ij = (w_i . v_jt)_t must satisfy
 *   Cons_q . tau_ij >= bound_q (the reference's `Constraints`, factor.py:910); row_cons [nrc][K+1]: fixed constraints on
 *   every row of W (`Row_constraints`), or NULL.
 * btf_gass_begin: x0 <- current state, v <- prior draw (W: sigma z on the free entries; V: the banded sampler with the
 *   likelihood off - z / seed as btf_ess_begin), slice heights ll(x0) + log u (u: [nchains] uniforms, or NULL: Philox),
 *   and the constraint analysis: which of the 10000 grid angles of gass.py:68 are valid for every chain.
 *   pick_ngrid > 0: the device also draws the candidate angles (<= 128): all valid ones, or pick_ngrid of them
 *   without replacement; linspace(-pi, pi, pick_ngrid) when no constraint restricts the ellipse (gass.py:81-83).
 * btf_gass_grid: info [nchains][2] = {valid angles, 1 if unrestricted}; mask [nchains][10000] bytes, slice [nchains],
 *   cur_ll [nchains] (each optional).
 * btf_gass_eval: log-likelihood of every candidate, ll_out [nchains][128] (-inf beyond a chain's count).  thetas
 *   [nchains][128] + ntheta [nchains]: the caller's candidates (host-driven: the reference's np.random.choice), or NULL:
 *   the device's own.
 * btf_gass_commit: x = x0 cos(theta_c) + v sin(theta_c) for chains with keep[c] == 0 (host-driven selection).
 * btf_gass_select: one candidate above the slice per chain, uniformly (Philox); none: the state stays (gass.py:121-128).
 * btf_gass_run = begin(pick) + eval + select, nothing read back. */
int btf_gass_set_constraints(btf_ctx* ctx, const double* cons, int J, const double* row_cons, int nrc);
int btf_gass_begin(btf_ctx* ctx, int what, int link, const double* z, const double* u, uint64_t seed, double eps0, int attempts,
                   int pick_ngrid);
int btf_gass_grid(btf_ctx* ctx, int what, int32_t* info, uint8_t* mask, double* slice, double* cur_ll);
int btf_gass_eval(btf_ctx* ctx, int what, const double* thetas, const int32_t* ntheta, double* ll_out);
int btf_gass_commit(btf_ctx* ctx, int what, const double* theta, const int32_t* keep);
int btf_gass_select(btf_ctx* ctx, int what, uint64_t seed, int32_t* naccept_out);
int btf_gass_run(btf_ctx* ctx, int what, int link, uint64_t seed, int ngrid, double eps0, int attempts);

/* ---- posterior summaries (stateless) --------------------------------
 * Mean and percentiles over the kept samples of f(w_s[i] . v_s[j,t]) for every cell: what the
 * reference's example scripts compute on the host as einsum('znk,zmtk->znmt', Ws, Vs).mean(0) /
 * np.percentile(., q, axis=0) (examples/gaussian_tensor_filtering.py:82-85), without materialising
 * the (S,N,M,T) tensor.  Ws (S,N,K), Vs (S,M,T,K) as run_gibbs returns them; transform 0 identity,
 * 1 ilogit, 2 square; q in [0,100], numpy's default linear interpolation; mean_out (N,M,T),
 * q_out (nq,N,M,T).  nsamples <= 16384.                                                      */
int btf_posterior_summary(int device, int nsamples, int nrows, int ncols, int ndepth, int nembeds,
                          const double* Ws, const double* Vs, int transform, const double* q, int nq,
                          double* mean_out, double* q_out);

/* ---- on-device sample collection (rng="device"; replaces the per-sample copies of
 * genlasso.py:51-65) ---------------------------------------------------------------------
 * btf_collect_begin allocates nsamples slots for W, V, Tau2 and the device-resident scalars;
 * btf_collect(slot) queues one copy kernel for the current state (no synchronisation);
 * btf_collect_end downloads the first nsamples slots (W (S,N,K), V (S,M,T,K), Tau2 (S,M,nD),
 * scalars (S,8): nu2, sigma2, lam2, lam2_a, ...; any may be NULL) and synchronises;
 * btf_collect_summary is btf_posterior_summary on the collected samples, without an upload.   */
int btf_collect_begin(btf_ctx* ctx, int nsamples);
int btf_collect(btf_ctx* ctx, int slot);
/* Let btf_gibbs_sweeps keep states by itself: after `countdown` more sweeps the state goes to slot first_slot, then every
 * `every`-th sweep to the next slot, until the slots of btf_collect_begin are full (genlasso.py:52-60: the samples kept
 * after burn-in, every nthin-th sweep).  every = 0 switches the schedule off.  One C call per block of sweeps instead of
 * one per kept sample.                                                                                               */
int btf_collect_schedule(btf_ctx* ctx, int every, int first_slot, int countdown);
int btf_collect_end(btf_ctx* ctx, int nsamples, double* W, double* V, double* Tau2, double* scalars8);
int btf_collect_summary(btf_ctx* ctx, int nsamples, int transform, const double* q, int nq, double* mean_out, double* q_out);

/* ---- measurement ----------------------------------------------------------
 * With profiling on, every kernel launch is bracketed by hipEvents on the ctx
 * stream; btf_kernel_times drains them: total milliseconds and launch count per
 * BTF_K_* id (arrays of BTF_K_COUNT).                                         */
int btf_set_profiling(btf_ctx* ctx, int on);
int btf_kernel_times(btf_ctx* ctx, double* ms_total, int64_t* launches);
/* Launch geometry of the streaming kernels: rows per workgroup (0 = default).   */
int btf_set_tuning(btf_ctx* ctx, int rows_per_block_w, int rows_per_block_v);

#ifdef __cplusplus
}
#endif
#endif /* BTF_H_ */
