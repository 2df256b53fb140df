// C ABI (include/btf.h) over the HIP kernels: context, device buffers, launches.
// gfx950 only.  No CPU fallback: every entry point either runs on the GPU or
// returns an error.
#include "../../include/btf.h"
#include "btf_kernels.h"
#include "btf_banded_fast.h"
#include "btf_banded_twist.h"
#include "btf_banded_chunk.h"
#include "btf_spectral.h"
#include "btf_gass.h"
#include "btf_fused.h"
#include "btf_instances.h"      // the large kernel families: extern templates, compiled in btf_instances.hip
#include "btf_comm.h"           // RCCL, bound at run time
#include <hip/hip_ext.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace btf;

namespace {

constexpr int MAX_K = 10;
constexpr int MAX_EVENTS = 8192;

struct EvPair { hipEvent_t a, b; int kid; };

}  // namespace

struct btf_ctx {
  int N = 0, M = 0, T = 0, K = 0, TF = 0, nD = 0, KK = 0;
  int dev = 0;
  int ncu = 256;                     // compute units of the device (hipDeviceAttributeMultiprocessorCount)
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int row0 = 0, nl = 0, col0 = 0, ml = 0;
  int hrow = -1, hcol = -1;  // btf_set_shard_halo: global index of the ONE stale-weight source row / column outside the blocks (-1: none);
                             // its statistics sit at local index nl / ml of the slabs (never updated, never summed - only its weights are read)
  int ldw = 0, ldv = 0;      // padded leading dimensions of A_wT / A_v
  int R = 1;
  bool have_data = false, binomial = false, weighted = false;
  double* A_wT = nullptr; double* C_wT = nullptr; double* A_v = nullptr; double* C_v = nullptr;
  double* B_wT = nullptr; double* B_v = nullptr;   // binomial: trials (0 where missing)
  unsigned char* C8_wT = nullptr; unsigned char* C8_v = nullptr;   // Gaussian data with missing replicates: counts as bytes (C_* freed)
  signed char* A8_wT = nullptr; signed char* A8_v = nullptr;       // Binomial data with integer counts: 2 (Y - N/2) as bytes
  double* W = nullptr; double* V = nullptr;
  double* Tau2 = nullptr;
  double lam2 = 1.0, sigma2 = 1.0, nu2 = 1.0;
  bool have_W = false, have_V = false, have_hyper = false;
  double* part = nullptr; size_t part_elems = 0;
  double* gpart = nullptr; int ngp_gram = 16;      // partial Grams of the last gram_kernel launch
  double* zbuf = nullptr; size_t z_elems = 0;
  double* bsum = nullptr; size_t bsum_elems = 0;
  double* gband = nullptr; size_t gband_stride = 0;
  int* status = nullptr;   // [0] flag [1] index
  int* tries = nullptr;
  int* st_ptr = nullptr; int* st_row = nullptr; double* st_coef = nullptr;
  int* st_drow = nullptr; double* st_dcoef = nullptr; bool st_dense_ok = false;   // VS_MAXE slots per (t,d) (spectral sampler)
  int* srcmap_w = nullptr; int* srcmap_v = nullptr;   // per-output source index of the cached weights
  bool stale_w = false, stale_v = false;
  int v_part_mode = 0;               // accumulation mode of the V half-sweep's partials in c->part (2: Gram blocks at the source columns)
  double ssw = 0.0, nobs = 0.0, sa2 = 0.0;      // within-cell SS, observation count, sum S1^2/cnt (Gaussian data)
  double nobs_global = -1.0;                    // sharded runs: observation count over all ranks (btf_set_global_nobs)
  bool w_part_valid = false; int w_part_mode = 0, w_part_nch = 0, w_part_rpb = 0; bool w_part_gv = false;   // W-step partials current?
  int rpb_w = 0, rpb_v = 0;
  int sampler = BTF_SAMPLER_BANDED;   // BTF_OPT_SAMPLER
  double* eig = nullptr;              // gram_eig_kernel output (spectral sampler): K eigenvalues, K*K vectors, sweeps
  // elliptical slice sampling (btf_ess_*): current state, prior draw, per-chain {hh, lo, hi, theta, ll}, partial sums
  double* essX0 = nullptr; double* essNu = nullptr; double* ess_st = nullptr; double* ess_theta = nullptr; int* ess_done = nullptr;
  double* ess_part = nullptr; size_t ess_part_elems = 0; int ess_last_chains = 0;
  // generalized analytic slice sampling (btf_gass_*): constraints, per-chain grids / candidates / likelihoods
  double* gs_cons = nullptr; double* gs_cc = nullptr; double* gs_rc = nullptr; int gs_J = 0, gs_nrc = 0;
  int* gs_cptr = nullptr; int* gs_cidx = nullptr; double* gs_cval = nullptr; int gs_cnnz = 0;     // the constraint matrix by its non-zeros
  double* gs_av = nullptr; unsigned char* gs_mask = nullptr; int* gs_info = nullptr;
  double* gs_thetas = nullptr; int* gs_ntheta = nullptr; double* gs_ll = nullptr; double* gs_llp = nullptr; size_t gs_llp_elems = 0; double* gs_hh = nullptr; double* gs_cur = nullptr;
  int* gs_nacc = nullptr; double* gs_u = nullptr;
  int gs_chains = 0, gs_what = -1, gs_link = 0;
  double lik_par[ESS_FAM_COUNT] = {0, 0, 0, 1.0, 1.0};     // parameter per likelihood family (btf_set_likelihood_param)
  long long* dbg = nullptr;
  double* vc_scratch = nullptr; size_t vc_scratch_elems = 0;     // factor records of the chunked chain sampler
#ifdef BTF_ACC_STAMPS
  long long* acc_stamps = nullptr;
#endif
  double* pband = nullptr;
  double* pimg = nullptr; unsigned long long pimg_version = 0;       // the band as LDS images [P | Pm] (dataflow tails of the fused V launch)
  // what the precomputed prior band (fused V launch, btf_fused.h) was formed from: every change of Tau2 / lam2 / the shard
  // moves prior_version on; the band is rebuilt (prior_band_kernel) when pband_version lags behind
  unsigned long long prior_version = 1, pband_version = 0, last_v_prior_version = 0;
  double* Ta = nullptr; double* Tb = nullptr; double* Tc = nullptr; double* lsum = nullptr;   // horseshoe+ chain (device mode)
  int* dr_ptr = nullptr; int* dr_col = nullptr; double* dr_val = nullptr;                   // Delta, CSR by row
  bool have_chain = false;
  double* pin = nullptr; size_t pin_elems = 0;   // pinned host staging (async SSE partials + W)
  size_t sse_nb = 0; bool sse_pending = false;
  double* pin_lsum = nullptr;
  // Negative-Binomial counts: raw replicates, per-cell sums / counts, rate buffers
  double* nb_data = nullptr; double* nb_S = nullptr; double* nb_cnt = nullptr;
  double* nb_R = nullptr; double* nb_C = nullptr; size_t nb_relems = 0;
  double* nb_tmp = nullptr; size_t nb_tmp_elems = 0;
  double* nb_out = nullptr; size_t nb_out_elems = 0;
  int nb_Rr = 0; bool counts = false; bool nb_bwt_written = false;
  unsigned int* nb_H = nullptr; double* nb_Hd = nullptr; double* nb_Hs = nullptr;   // per-row count histograms (u32, f64) and their sum over rows
  double* nb_L = nullptr;            // [N + 1]: per-row sum cnt*log(1-p), then the total
  int* fill_tab = nullptr; int fill_n = 0; int fill_key = -1;   // band assembly program of the twisted kernel
  int* nb_optr = nullptr; double* nb_oval = nullptr; int nb_nout = 0;   // per-row outlier lists (CSR)
  int nb_ymax = 0;                   // largest tabulated count present (histogram bins above it are empty)
  double* nb_G = nullptr;            // suffix sums of the histogram of all counts: nb_G[k] = #{observations > k}, k < NB_TAB
  bool nb_tabulable = false;        // every observed count is an integer in [0, NB_TAB)
  bool nb_L_valid = false;          // nb_L matches the current W, V
  bool nb_hist = true;              // BTF_OPT_NB_HISTOGRAMS
  int pg_mode = PG_MODE_DEFAULT;    // BTF_OPT_PG_EXACT: PG_MODE_DEFAULT / PG_MODE_EXACT_ALL / PG_MODE_SERIES_ALL
  // trial counts below the normal range: any integer up to PG_AUTO_EXACT_MAX / any larger integer / any non-integer
  bool pg_has_small = true, pg_has_big = true, pg_has_frac = true;
  // on-device sample collection (run_gibbs, rng="device"): [nsamp] slots of W, V, Tau2 and the scalars
  double* smp_W = nullptr; double* smp_V = nullptr; double* smp_T = nullptr; double* smp_s = nullptr; int smp_n = 0;
  int col_every = 0, col_slot = 0, col_count = 0;       // btf_collect_schedule: btf_gibbs_sweeps keeps every col_every-th state
  double* hyp = nullptr;        // device-resident scalars [HYP_COUNT] (nu2, sigma2, lam2, lam2_a, ...)
  bool dev_scalars = false;     // kernels read nu2 / sigma2 / lam2 from hyp instead of the host copies
  double* pin_hyp = nullptr;
  double* gsum_v = nullptr; bool w_part_gsum = false;   // V'V summed by a side workgroup of the W accumulation launch (GramSide.sum_*): w_solve reads KK doubles
  double* gpart_w = nullptr; int ngp_w = 0;   // W'W partials written by w_solve (valid until W changes otherwise)
  double* gpart_v = nullptr; int ngp_v = 0;   // V'V partials written by the fast banded sampler
  bool fuse_gram = true;
  // curve-structured replicate counts (btf_kernels.h, CurveLists): counts constant along the depth axis
  bool curve = false, curve_opt = true;
  std::vector<unsigned char> cv_cij;                                  // host copy of c_ij [N][M] (stale-source test)
  int* cv_cptr = nullptr; int* cv_crow = nullptr; double* cv_cdef = nullptr;   // by column: deficient rows
  int* cv_rptr = nullptr; int* cv_rcol = nullptr; double* cv_rdef = nullptr;   // by row: deficient columns
  double* eig_cols = nullptr;                                         // [M][K + K*K + 8] per-column eigen-systems
  int* cv_dcols = nullptr; int cv_ndef = 0;                           // the columns that have deficient rows
  bool w_part_curve = false;                                          // the W-step partials were made in curve mode
  // sharded runs, BTF_OPT_SPLIT_ACCUM: the chunks of the rank's own block of the fixed factor are accumulated right
  // behind the kernel that drew it (no exchange needed), the rest behind the all-gather
  bool split_accum = false;
  bool w_local_done = false, v_local_done = false;                    // own-block chunks of the next W / V accumulation are in c->part
  int w_local_rpb = 0, w_local_mode = 0, v_local_rpb = 0, v_local_mode = 0;
  hipEvent_t ev_draw = nullptr, ev_join = nullptr;                    // behind the last draw kernel / the comm stream's tail
  // the ctx-owned communicator (btf_comm_init; btf_comm.h).  comm_rank / comm_world are the communicator's; gather_rank /
  // gather_world the block decomposition the all-gathers reassemble - the same, except in a rehearsal (btf_comm_rehearse:
  // a one-rank communicator moving the messages of rank gather_rank of gather_world through scratch buffers)
  ncclComm_t comm = nullptr; int comm_rank = 0, comm_world = 1, gather_rank = 0, gather_world = 1;
  bool comm_rehearse = false;
  hipStream_t comm_stream = nullptr;                                  // the overlapped exchange runs its gathers here
  double* comm_scr = nullptr; size_t comm_scr_elems = 0;              // rehearsal: [send | recv] of the larger message
  double* comm_words = nullptr;                                       // 16 device doubles: btf_allreduce_sum's staging
  // the peer-window transport (btf_comm.h): this rank's mailbox, the table of where every rank's buffers are mapped
  // here, the mappings to close, the collective counter
  PeerMailbox* peer_box = nullptr;
  PeerTable* peer_tab = nullptr;
  unsigned* peer_counters = nullptr;
  std::vector<void*> peer_opened;
  bool peer_on = false;
  unsigned long long peer_epoch = 0;
  long long peer_timeout_ticks = 0;
  bool tau_pending = false; unsigned long long tau_seed = 0; double tau_stability = 1e-6;   // btf_queue_Tau2
  // the four-launch sweep (BTF_OPT_FUSED_SWEEP): per-column residual parts left by the spectral V sampler, and a queued
  // nu2 / sigma2 draw that the next W accumulation launch carries as a side workgroup (btf_queue_scalars)
  bool fused_sweep = true;
  double* sse_cols = nullptr; bool sse_cols_valid = false;
  double* vs_rec = nullptr; size_t vs_rec_elems = 0;       // HBM scratch of the spectral sampler's pivot records (long depth axes)
  bool nu2_drawn_since_v = false;      // a device nu2 draw happened since the last V half-sweep: the caller runs full sweeps
  bool sc_pending = false; unsigned long long sc_seed = 0; int sc_which = 0; double sc_prior[4] = {0, 0, 0, 0};
  bool lam_pending = false; unsigned long long lam_seed = 0; int lam_exact = 0;            // btf_queue_lam2
  bool band_in_wsolve = true;                                         // (A/B aid: BTF_BAND_IN_WSOLVE=0: the band's own launch)
  bool v_wants_band = false, band_img = false; int band_PB = 0;       // the last fused V launch loaded the precomputed prior band (and its LDS image)
  bool lam_in_wsolve = true;                                          // (A/B aid: BTF_LAM_IN_WSOLVE=0 leaves the draw to the V launch)
  unsigned long long sweep_w = 0, sweep_v = 0;
  // the two-launch W+V step (BTF_OPT_FUSED_STEP, btf_fused.h): tickets / flags (zeroed once; 32 words = one 128-byte line
  // per flag), the write-through copies the tails read, the epoch of the hand-offs (one per fused launch, never reused)
  int fused_dataflow = 1;            // BTF_OPT_FUSED_DATAFLOW: 1 (default) the fused V launch runs the barrier-free tail where it applies
  int fused_step = 1;                // BTF_OPT_FUSED_STEP: 0 four launches, 1 (default) the V launch carries its sampler, 2 the W launch its solve too
  unsigned* fz_words = nullptr; int fz_tiles_w = 0, fz_tiles_v = 0;
  double* fz_pub = nullptr;
  unsigned fz_epoch = 0, fz_gram_total = 0, fz_w_total = 0;
  bool profiling = false;
  std::vector<EvPair> ev_pool;
  size_t ev_used = 0;
  double ms_total[BTF_K_COUNT] = {0};
  int64_t launches[BTF_K_COUNT] = {0};
  std::string err;
  int fail_index = -1;
};

namespace {

thread_local std::string g_err;

int fail(btf_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  g_err = msg;
  return code;
}

#define HIPCHK(ctx, call)                                                                  \
  do {                                                                                     \
    hipError_t e__ = (call);                                                               \
    if (e__ != hipSuccess)                                                                 \
      return fail(ctx, BTF_EHIP, std::string(#call) + ": " + hipGetErrorString(e__));      \
  } while (0)

template <typename T>
int dev_alloc(btf_ctx* c, T** p, size_t n) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (n == 0) n = 1;
  HIPCHK(c, hipMalloc((void**)p, n * sizeof(T)));
  return BTF_OK;
}

int round_up(int x, int m) { return (x + m - 1) / m * m; }
// "this function's attribute was set on device d": hipFuncSetAttribute is per device, and `device=` is a public
// constructor keyword - a process-wide flag would skip the call for a second context on another GPU
inline bool dev_flag_is_set(const std::atomic<unsigned long long>& f, int dev) { return dev >= 0 && dev < 64 && ((f.load() >> dev) & 1ULL); }
inline void dev_flag_set(std::atomic<unsigned long long>& f, int dev) { if (dev >= 0 && dev < 64) f.fetch_or(1ULL << dev); }

// One kernel launch, counted per BTF_K_* id.  With profiling on the launch goes through
// hipExtLaunchKernelGGL so that the two events bracket exactly this dispatch (its start
// and completion timestamps), not the gaps around it.
struct Prof {
  btf_ctx* c; int kid; EvPair* ev = nullptr;
  Prof(btf_ctx* c_, int kid_) : c(c_), kid(kid_) {
    c->launches[kid]++;
    if (c->profiling && c->ev_used < c->ev_pool.size()) {
      ev = &c->ev_pool[c->ev_used++];
      ev->kid = kid;
    }
  }
  template <typename F, typename... Args>
  void launch(F kernel, dim3 grid, dim3 block, size_t lds, Args... args) {
    launch_on(c->stream, kernel, grid, block, lds, args...);
  }
  template <typename F, typename... Args>
  void launch_on(hipStream_t st, F kernel, dim3 grid, dim3 block, size_t lds, Args... args) {
    if (ev) hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, ev->a, ev->b, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, args...);
  }
};

// Delta' diag(lambda) Delta stencil: for every (t,d), d = 0..tf+1, the Delta rows r that
// touch both t and t+d with the coefficient product Delta[r,t]*Delta[r,t+d].
// Delta is rebuilt here from its definition (reference utils.py:56-98): anchor row
// e_0, then D^(0..tf) with D the first-difference operator, alternating D' and D.
void build_delta_dense(int T, int tf, std::vector<double>& Delta, int& nD) {
  std::vector<std::vector<double>> rows;
  {
    std::vector<double> a(T, 0.0);
    a[0] = 1.0;
    rows.push_back(a);
  }
  // D : (T-1) x T
  auto D = [&](int r, int c) -> double { return c == r ? -1.0 : (c == r + 1 ? 1.0 : 0.0); };
  std::vector<std::vector<double>> cur(T - 1, std::vector<double>(T, 0.0));
  for (int r = 0; r < T - 1; ++r) { cur[r][r] = -1.0; cur[r][r + 1] = 1.0; }
  for (int k = 0; k <= tf; ++k) {
    if (k > 0) {
      std::vector<std::vector<double>> nxt;
      if ((k - 1) % 2 == 0) {  // D' * cur : T x T
        nxt.assign(T, std::vector<double>(T, 0.0));
        for (int r = 0; r < T; ++r)
          for (int q = 0; q < T - 1; ++q) {
            double dq = D(q, r);
            if (dq != 0.0)
              for (int c = 0; c < T; ++c) nxt[r][c] += dq * cur[q][c];
          }
      } else {  // D * cur : (T-1) x T
        nxt.assign(T - 1, std::vector<double>(T, 0.0));
        for (int r = 0; r < T - 1; ++r)
          for (int q = 0; q < T; ++q) {
            double dq = D(r, q);
            if (dq != 0.0)
              for (int c = 0; c < T; ++c) nxt[r][c] += dq * cur[q][c];
          }
      }
      cur.swap(nxt);
    }
    for (auto& r : cur) rows.push_back(r);
  }
  nD = (int)rows.size();
  Delta.assign((size_t)nD * T, 0.0);
  for (int r = 0; r < nD; ++r)
    for (int c = 0; c < T; ++c) Delta[(size_t)r * T + c] = rows[r][c];
}

// the Delta'.Delta stencil per (t, d) as CSR over the penalty rows: pure host code (btf_host_selftest walks it under the
// host sanitizers, scripts/asan_host.sh)
void stencil_csr(int T, int tf, const std::vector<double>& Delta, int nD, std::vector<int>& ptr, std::vector<int>& row,
                 std::vector<double>& coef) {
  const int D1 = tf + 2;
  ptr.assign((size_t)T * D1 + 1, 0);
  row.clear(); coef.clear();
  for (int t = 0; t < T; ++t)
    for (int d = 0; d < D1; ++d) {
      if (t + d < T)
        for (int r = 0; r < nD; ++r) {
          double p = Delta[(size_t)r * T + t] * Delta[(size_t)r * T + t + d];
          if (p != 0.0) { row.push_back(r); coef.push_back(p); }
        }
      ptr[t * D1 + d + 1] = (int)row.size();
    }
}
int build_stencil(btf_ctx* c) {
  const int T = c->T, tf = c->TF, D1 = tf + 2;
  std::vector<double> Delta;
  int nD = 0;
  build_delta_dense(T, tf, Delta, nD);
  c->nD = nD;
  std::vector<int> ptr, row;
  std::vector<double> coef;
  stencil_csr(T, tf, Delta, nD, ptr, row, coef);
  {  // Delta itself, CSR by row (device Tau2 update)
    std::vector<int> rp(nD + 1, 0), rc;
    std::vector<double> rv;
    for (int r = 0; r < nD; ++r) {
      for (int t = 0; t < T; ++t)
        if (Delta[(size_t)r * T + t] != 0.0) { rc.push_back(t); rv.push_back(Delta[(size_t)r * T + t]); }
      rp[r + 1] = (int)rc.size();
    }
    int rc2;
    if ((rc2 = dev_alloc(c, &c->dr_ptr, rp.size()))) return rc2;
    if ((rc2 = dev_alloc(c, &c->dr_col, rc.size()))) return rc2;
    if ((rc2 = dev_alloc(c, &c->dr_val, rv.size()))) return rc2;
    HIPCHK(c, hipMemcpy(c->dr_ptr, rp.data(), rp.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->dr_col, rc.data(), rc.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->dr_val, rv.data(), rv.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  // sanity: nothing outside the band
  for (int r = 0; r < nD; ++r) {
    int lo = T, hi = -1;
    for (int t = 0; t < T; ++t)
      if (Delta[(size_t)r * T + t] != 0.0) { lo = std::min(lo, t); hi = std::max(hi, t); }
    if (hi - lo > tf + 1) return fail(c, BTF_EINVAL, "penalty row wider than the band");
  }
  int rc;
  if ((rc = dev_alloc(c, &c->st_ptr, ptr.size()))) return rc;
  if ((rc = dev_alloc(c, &c->st_row, row.size()))) return rc;
  if ((rc = dev_alloc(c, &c->st_coef, coef.size()))) return rc;
  HIPCHK(c, hipMemcpy(c->st_ptr, ptr.data(), ptr.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->st_row, row.data(), row.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->st_coef, coef.data(), coef.size() * sizeof(double), hipMemcpyHostToDevice));
  {  // the same stencil with a fixed number of slots per (t,d): the spectral sampler fetches it in one round trip
    const size_t ne = (size_t)T * D1;
    std::vector<int> drow(ne * VS_MAXE, 0);
    std::vector<double> dcoef(ne * VS_MAXE, 0.0);
    c->st_dense_ok = true;
    for (size_t e = 0; e < ne; ++e) {
      const int cnt = ptr[e + 1] - ptr[e];
      if (cnt > VS_MAXE) { c->st_dense_ok = false; break; }
      for (int u = 0; u < cnt; ++u) { drow[e * VS_MAXE + u] = row[ptr[e] + u]; dcoef[e * VS_MAXE + u] = coef[ptr[e] + u]; }
    }
    if ((rc = dev_alloc(c, &c->st_drow, drow.size()))) return rc;
    if ((rc = dev_alloc(c, &c->st_dcoef, dcoef.size()))) return rc;
    HIPCHK(c, hipMemcpy(c->st_drow, drow.data(), drow.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->st_dcoef, dcoef.data(), dcoef.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  return BTF_OK;
}

// ---- templated launch tables -------------------------------------------------
// rows per workgroup from which the complete-data stream keeps three rows in flight per wave (and, K >= 8, stages the
// factor rows in LDS); BTF_UNR3_RPB overrides it for A/B runs
inline int unr3_min_rpb() {
  static const int v = [] { const char* e = std::getenv("BTF_UNR3_RPB"); return e ? std::atoi(e) : 2048; }();
  return v;
}
inline bool lean_on() {
  static const bool v = [] { const char* e = std::getenv("BTF_ACC_LEAN"); return !e || std::atoi(e) != 0; }();      // (A/B aid)
  return v;
}
template <int K>
void launch_accum(btf_ctx* c, int kid, int mode, const double* X, const double* Cx, const unsigned char* C8, const double* U,
                  const int* srcmap, int Rdim, int ld, int rpb, int nch, EigSide side = EigSide{nullptr, 0, 0, nullptr},
                  EigSideCols sidec = EigSideCols{nullptr, 0, CurveLists{nullptr, nullptr, nullptr}, nullptr, 0.0, nullptr},
                  TauSide tau = TauSide{}, GramSide gram = GramSide{},
                  ChunkMap cm = ChunkMap{0, 0, INT_MAX, 0, 0, 0}, SweepSide sw = SweepSide{},
                  const FuseW* fw = nullptr, const FuseV* fv = nullptr) {
  // nch: the chunks THIS launch covers (all of them unless cm says otherwise)
  if (cm.row_end == 0) cm.row_end = Rdim;
  {
    // Non-temporal loads for slabs the Infinity Cache cannot keep: a half-sweep pair streams both layouts, so a layout is
    // still there one step later only if two of them (plus the weights) fit.  Threshold on the f64 statistic of ONE launch;
    // BTF_NT_MIN_MB overrides (0: always, a huge value: never).
    static const double min_mb = [] { const char* e = std::getenv("BTF_NT_MIN_MB"); return e ? std::atof(e) : 100.0; }();
    cm.nt = (double)Rdim * (double)ld * 8.0 >= min_mb * 1048576.0 ? 1 : 0;
  }
#ifdef BTF_ACC_STAMPS
  cm.stamps = c->acc_stamps;
#endif
  Prof p(c, kid);
  const int cpw = TAU_SIDE_CPW;
  cm.nside = (sw.sc.hyp ? 1 : 0) + (sw.lam.hyp ? 1 : 0) + (side.out ? 1 + eig_side_groups(sidec.ncols, acc_waves(K, mode)) : 0) +
             (tau.Tau2 ? (tau.M + cpw - 1) / cpw : 0) + (gram.gpart ? gram.nblocks : 0) + (gram.sum_src ? 1 : 0) + (fw ? fw->owners : 0);   // the side tasks' workgroups (and the fused W launch's owners), in front
  dim3 grid((ld / ACC_TILE) * nch + cm.nside);
  if (fw || fv) {              // the two-launch step: complete data, 16 waves, the tail in the same launch (btf_fused.h)
    const bool unr3 = rpb >= unr3_min_rpb();
    if constexpr (K == 8) {
      if (unr3 && fw) { p.launch(accum_kernel<K, 0, 16, double, double, 3, 2, FUSE_W>, grid, dim3(16 * WAVE), 0, X, Cx, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, *fw); return; }
      if (unr3 && fv) { p.launch(accum_kernel<K, 0, 16, double, double, 3, 2, FUSE_V>, grid, dim3(16 * WAVE), 0, X, Cx, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, *fv); return; }
    }
    if constexpr (K <= 8) {
      if (fw) p.launch(accum_kernel<K, 0, 16, double, double, 0, 2, FUSE_W>, grid, dim3(16 * WAVE), 0, X, Cx, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, *fw);
      else {
        // the barrier-free (dataflow) tail where it applies: one chunk per tile (the sums never leave the workgroup), the
        // precomputed band, no scalar drawn by a side workgroup of this launch, every value in one reduction round
        // (nembeds <= 6), its LDS footprint beside the partial sums; BTF_OPT_FUSED_DATAFLOW 0: the barrier form
        const bool df_on = c->fused_dataflow != 0;
        FuseV f2 = *fv;
        constexpr int RG = K < 4 ? 4 : (K > 6 ? 6 : K);        // accum_kernel's ACC_RG of the 16-wave complete-data instance
        f2.dataflow = (df_on && f2.a.pband && f2.pimg && !f2.cnt && !f2.hp.flag && K <= 6 &&
                       vf_df_fits(f2.a.T, K, f2.a.TF, f2.a.nD, 16, RG)) ? 1 : 0;

        if constexpr (K <= 6) {
          if (f2.dataflow) { p.launch(accum_kernel<K, 0, 16, double, double, 0, 2, FUSE_VDF>, grid, dim3(16 * WAVE), 0, X, Cx, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, f2); return; }
        }
        p.launch(accum_kernel<K, 0, 16, double, double, 0, 2, FUSE_V>, grid, dim3(16 * WAVE), 0, X, Cx, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, f2);
      }
    }
    return;
  }
  const signed char* A8 = (mode >= 1 && !C8 && X == c->A_wT) ? c->A8_wT : ((mode >= 1 && !C8 && X == c->A_v) ? c->A8_v : nullptr);
  if (A8) {                    // Binomial pseudo-data as bytes (f64 weights)
    if (mode == 2) p.launch(accum_kernel<K, 2, acc_waves(K, 2), double, signed char>, grid, dim3(acc_waves(K, 2) * WAVE), 0, A8, Cx, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, FuseNone{});
    else p.launch(accum_kernel<K, 1, acc_waves(K, 1), double, signed char>, grid, dim3(acc_waves(K, 1) * WAVE), 0, A8, Cx, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, FuseNone{});
    return;
  }
  if (mode >= 1 && C8) {       // byte weights (Gaussian replicate counts)
    if (mode == 2) p.launch(accum_kernel<K, 2, acc_waves(K, 2), unsigned char>, grid, dim3(acc_waves(K, 2) * WAVE), 0, X, C8, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, FuseNone{});
    else p.launch(accum_kernel<K, 1, acc_waves(K, 1), unsigned char>, grid, dim3(acc_waves(K, 1) * WAVE), 0, X, C8, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, FuseNone{});
  } else if (mode == 2) p.launch(accum_kernel<K, 2>, grid, dim3(acc_waves(K, 2) * WAVE), 0, X, Cx, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, FuseNone{});
  else if (mode == 1) p.launch(accum_kernel<K, 1>, grid, dim3(acc_waves(K, 1) * WAVE), 0, X, Cx, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, FuseNone{});
  else if (K <= 8 && acc_waves(K, 0) == 16 && rpb < unr3_min_rpb() && !tau.Tau2 && !sw.sc.hyp && !sw.lam.hyp && lean_on()) {
    // the plain W+V step's launches carry no gamma-drawing side task: the instance compiled without them
    if constexpr (K <= 8) p.launch(accum_kernel<K, 0, 16, double, double, 0, 2, FUSE_LEAN>, grid, dim3(16 * WAVE), 0, X, Cx, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, FuseNone{});
  }
  else if (K >= 10 && ACC_WAVES != acc_waves(10, 0) && !side.out && rpb < unr3_min_rpb()) {
    // nembeds 10: a launch without eigen side tasks (the W half-sweep's) takes the 16-wave instance
    if constexpr (K >= 10) p.launch(accum_kernel<K, 0, ACC_WAVES>, grid, dim3(ACC_WAVES * WAVE), 0, X, Cx, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, FuseNone{});
  }
  else if (rpb >= unr3_min_rpb()) p.launch(accum_kernel<K, 0, acc_waves(K, 0), double, double, 3>, grid, dim3(acc_waves(K, 0) * WAVE), 0, X, Cx, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, FuseNone{});
  else p.launch(accum_kernel<K, 0>, grid, dim3(acc_waves(K, 0) * WAVE), 0, X, Cx, U, srcmap, c->part, Rdim, ld, rpb, side, sidec, tau, gram, cm, sw, FuseNone{});
}
// Which Polya-Gamma launches a draw needs (pg_class_of): the flat exact kernel for the integer counts it takes
// under the mode, the series kernel and / or the f64 Devroye kernel with a fractional part for the rest - each only
// if the trial counts hold such a cell; the first launch also writes the zeros of the unobserved cells and the
// normal-range draws.  Count data (Negative-Binomial): the pseudo-trial counts move with the rate and are integers
// only by accident - the series takes every cell in one launch unless another mode was asked for.
struct PgPasses { bool flat, series, frac; int mode; };
PgPasses pg_passes(const btf_ctx* c) {
  PgPasses p{false, false, false, c->pg_mode};
  if (c->counts && p.mode == PG_MODE_DEFAULT) p.mode = PG_MODE_SERIES_ALL;
  if (p.mode == PG_MODE_SERIES_ALL) { p.series = true; return p; }
  if (p.mode == PG_MODE_EXACT_ALL) { p.flat = c->pg_has_small || c->pg_has_big; p.frac = c->pg_has_frac; }
  else { p.flat = c->pg_has_small; p.series = c->pg_has_big || c->pg_has_frac; }
  if (!p.flat && !p.series && !p.frac) p.series = true;      // nothing to draw: one launch for the zeros
  return p;
}
template <int K>
void launch_pg(btf_ctx* c, const double* B, double* out, const double* Lf, const double* Uf, int nl, int ld, int Rdim,
               unsigned long long base, unsigned long long stride_r, unsigned long long stride_l, unsigned long long seed) {
  const int gx = (nl + PG_THREADS - 1) / PG_THREADS;
  int nrb = std::max(1, std::min(Rdim, 4096 / std::max(1, gx)));
  const int rpb = (Rdim + nrb - 1) / nrb;
  nrb = (Rdim + rpb - 1) / rpb;
  const PgPasses ps = pg_passes(c);
  int fill = 1;
  if (ps.flat) {
    Prof p(c, BTF_K_PG);
    const int rpx = round_up(rpb, PGX_CPL);                   // whole lists
    p.launch(pgx_kernel<K, PGX_CPL>, dim3(gx, (Rdim + rpx - 1) / rpx), dim3(256), pgx_rows_lds(PGX_CPL), B, out, Lf, Uf, nl, ld, Rdim,
             rpx, base, stride_r, stride_l, seed, ps.mode, fill);
    fill = 0;
  }
  if (ps.series) {
    Prof p(c, BTF_K_PG);
    p.launch(pg_kernel<K, PG_PATH_SERIES>, dim3(gx, nrb), dim3(PG_THREADS), 0, B, out, Lf, Uf, nl, ld, Rdim, rpb, base, stride_r, stride_l, seed, ps.mode, fill);
    fill = 0;
  }
  if (ps.frac) {
    Prof p(c, BTF_K_PG);
    p.launch(pg_kernel<K, PG_PATH_EXACT>, dim3(gx, nrb), dim3(PG_THREADS), 0, B, out, Lf, Uf, nl, ld, Rdim, rpb, base, stride_r, stride_l, seed, ps.mode, fill);
  }
}
template <int K>
void launch_gram(btf_ctx* c, const double* U, int Rdim) {
  Prof p(c, BTF_K_GRAM);
  c->ngp_gram = gram_blocks(Rdim);      // 16 partial Grams, up to 64 for long factors (4 MB of V at C5: 16 us -> 5 us)
  p.launch(gram_kernel<K>, dim3(c->ngp_gram), dim3(GRAM_THREADS), 0, U, Rdim, c->gpart);
}
template <int K>
void launch_wsolve(btf_ctx* c, const WSolveArgs& a0) {
  Prof p(c, BTF_K_W_SOLVE);
  WSolveArgs a = a0;
  const int rw = ws_rows_for(a.nl);
  a.nside = (a.lam.hyp ? 1 : 0) + (a.band.pband ? a.band.ml : 0);      // (the lam2 workgroup and the band's of a full sweep)
  const dim3 grid((a.nl + rw - 1) / rw + a.nside);
#define WS_LAUNCH(WT, RWV) p.launch(w_solve_kernel<K, WT, RWV>, grid, dim3(WS_ROWS * ws_split_of(K, WT)), 0, a)
#define WS_PICK(WT) do { if (rw == 8) WS_LAUNCH(WT, 8); else if (rw == 16) WS_LAUNCH(WT, 16); else if (rw == 32) WS_LAUNCH(WT, 32); else WS_LAUNCH(WT, 64); } while (0)
  if (a.weighted) WS_PICK(true); else WS_PICK(false);
#undef WS_PICK
#undef WS_LAUNCH
}
template <int K>
hipError_t launch_vbanded(btf_ctx* c, const VBandArgs& a, size_t lds_bytes) {
  static std::atomic<unsigned long long> attr_set{0};      // one bit per device: the attribute is per device
  if (!dev_flag_is_set(attr_set, c->dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)v_banded_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    dev_flag_set(attr_set, c->dev);
  }
  Prof p(c, BTF_K_V_BANDED);
  p.launch(v_banded_kernel<K>, dim3(a.ml), dim3(WAVE), lds_bytes, a);
  return hipSuccess;
}
template <int NPL, bool ROW16>
hipError_t launch_vbanded_fast(btf_ctx* c, const VBandArgs& a, size_t lds_bytes) {
  static std::atomic<unsigned long long> attr_set{0};      // one bit per device: the attribute is per device
  if (!dev_flag_is_set(attr_set, c->dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)v_banded_fast_kernel<NPL, ROW16>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    dev_flag_set(attr_set, c->dev);
  }
  Prof p(c, BTF_K_V_BANDED);
  p.launch(v_banded_fast_kernel<NPL, ROW16>, dim3(a.ml), dim3(VB_THREADS), lds_bytes, a, c->K);
  return hipSuccess;
}
hipError_t dispatch_vbanded_fast(btf_ctx* c, const VBandArgs& a, int bw, size_t lds_bytes, bool* handled) {
  const int npairs = bw * (bw - 1) / 2;
  const int npl = std::max(1, (npairs + WAVE - 1) / WAVE);
  *handled = true;
  if (bw <= 15) {
    if (npl == 1) return launch_vbanded_fast<1, true>(c, a, lds_bytes);
    if (npl == 2) return launch_vbanded_fast<2, true>(c, a, lds_bytes);
  } else {
    switch (npl) {
      case 2: return launch_vbanded_fast<2, false>(c, a, lds_bytes);
      case 3: return launch_vbanded_fast<3, false>(c, a, lds_bytes);
      case 4: return launch_vbanded_fast<4, false>(c, a, lds_bytes);
      case 5: return launch_vbanded_fast<5, false>(c, a, lds_bytes);
      case 6: return launch_vbanded_fast<6, false>(c, a, lds_bytes);
      case 7: return launch_vbanded_fast<7, false>(c, a, lds_bytes);
      case 8: return launch_vbanded_fast<8, false>(c, a, lds_bytes);
      default: break;
    }
  }
  *handled = false;
  return hipSuccess;
}
// the chunked single chain (btf_banded_chunk.h): bands that do not fit LDS in one piece
template <int NPL>
hipError_t launch_vbanded_chunk(btf_ctx* c, const VBandArgs& a, size_t lds_bytes, int CH) {
  static std::atomic<unsigned long long> attr_set{0};      // one bit per device: the attribute is per device
  if (!dev_flag_is_set(attr_set, c->dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)v_banded_chunk_kernel<NPL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    dev_flag_set(attr_set, c->dev);
  }
  Prof p(c, BTF_K_V_BANDED);
  p.launch(v_banded_chunk_kernel<NPL>, dim3(a.ml), dim3(VB_THREADS), lds_bytes, a, c->K, CH);
  return hipSuccess;
}
constexpr size_t VC_LDS_BUDGET = 158 * 1024;
inline int vc_npl(int bw) { return std::max(1, ((bw - 1) * (bw - 2) / 2 + WAVE - 1) / WAVE); }
// columns per chunk of the chunked chain for this context, 0 if it does not apply
inline int vc_chunk_for(const btf_ctx* c, bool wt) {
  const int bw = (c->TF + 1) * c->K;
  if (bw < 3 || vc_npl(bw) > 8) return 0;
  return vc_pick_chunk(c->T, c->K, c->TF, wt ? 1 : 0, VC_LDS_BUDGET);
}
hipError_t dispatch_vbanded_chunk(btf_ctx* c, const VBandArgs& a, int bw, int CH, bool wt, bool* handled) {
  const size_t lds_bytes = vc_lds_bytes(c->T, c->K, c->TF, wt ? 1 : 0, CH);
  *handled = true;
  switch (vc_npl(bw)) {
    case 1: return launch_vbanded_chunk<1>(c, a, lds_bytes, CH);
    case 2: return launch_vbanded_chunk<2>(c, a, lds_bytes, CH);
    case 3: return launch_vbanded_chunk<3>(c, a, lds_bytes, CH);
    case 4: return launch_vbanded_chunk<4>(c, a, lds_bytes, CH);
    case 5: return launch_vbanded_chunk<5>(c, a, lds_bytes, CH);
    case 6: return launch_vbanded_chunk<6>(c, a, lds_bytes, CH);
    case 7: return launch_vbanded_chunk<7>(c, a, lds_bytes, CH);
    case 8: return launch_vbanded_chunk<8>(c, a, lds_bytes, CH);
    default: break;
  }
  *handled = false;
  return hipSuccess;
}
template <int NPL, bool ROW16, int KC = 0, int TFC = 0, int TC = 0>
hipError_t launch_vbanded_twist(btf_ctx* c, const VBandArgs& a, size_t lds_bytes) {
  static std::atomic<unsigned long long> attr_set{0};      // one bit per device: the attribute is per device
  if (!dev_flag_is_set(attr_set, c->dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)v_banded_twist_kernel<NPL, ROW16, KC, TFC, TC>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    dev_flag_set(attr_set, c->dev);
  }
  Prof p(c, BTF_K_V_BANDED);
  p.launch(v_banded_twist_kernel<NPL, ROW16, KC, TFC, TC>, dim3(a.ml), dim3(VT_THREADS), lds_bytes, a, c->K);
  return hipSuccess;
}
// Band assembly program of v_banded_twist_kernel: one entry per structurally non-zero band word of the left
// and right views, {destination, source, diagonal source or -1, 0} as LDS word offsets of tw_layout: the K-k
// same-depth-block entries come from the Gram block(s) Ql, the tf+1 prior couplings from the prior band P
// (btf_banded_twist.h, "assemble the two bands").  Padded to a multiple of the workgroup size with reads of
// P[0] written to a dummy word.
// band assembly program of the twisted kernel, [n][4] = {dst, src, diag-src or -1, 0} in LDS word offsets: pure host code
// wmode: 0 complete data (one shared block), 1 weighted (per-depth blocks staged in LDS), 2 weighted, blocks fetched from
// the partials: source -2 - (q T + t)
void fill_table_host(int T, int K, int TF, int wmode, std::vector<int>& tab) {
  const int KK = tri(K), D1 = TF + 2, n = T * K;
  const bool weighted = wmode != 0;
  const TwLayout W = tw_layout(T, K, TF, wmode);
  auto qsrc = [&](int t, int q) { return wmode == 2 ? -2 - (q * T + t) : W.Ql + (weighted ? t * KK : 0) + q; };
  const int R1 = W.L.R1, nl = W.nl, nr = W.nr, nL = W.nL;
  tab.clear();
  auto put = [&](int dst, int src, int dia) { tab.push_back(dst); tab.push_back(src); tab.push_back(dia); tab.push_back(0); };
  for (int i = 0; i < nL; ++i) {                       // left view: column i (global g = i), entry (g+aa, g)
    const int t = i / K, k = i - t * K;
    for (int aa = 0; aa < K - k; ++aa)
      if (i + aa < nL) put(W.L.band + i * R1 + aa, qsrc(t, lidx(k + aa, k)), aa == 0 ? W.P + t * D1 : -1);
    for (int d = 1; d < D1; ++d)
      if (t + d < T && i + d * K < nL) put(W.L.band + i * R1 + d * K, W.P + t * D1 + d, -1);
  }
  for (int m = 0; m < nr; ++m) {                       // right view: mirrored column m (global gc = n-1-m), entry (gc, gc-aa)
    const int gc = n - 1 - m, tc = gc / K, kc = gc - tc * K;
    for (int aa = 0; aa <= kc; ++aa)
      if (gc - aa >= nl) put(W.R.band + m * R1 + aa, qsrc(tc, lidx(kc, kc - aa)), aa == 0 ? W.P + tc * D1 : -1);
    for (int d = 1; d < D1; ++d)
      if (gc - d * K >= nl) put(W.R.band + m * R1 + d * K, W.P + (tc - d) * D1 + d, -1);
  }
  while ((tab.size() / 4) % VT_THREADS) put(W.L.dummy + 1, W.P, -1);
}
// how the twisted kernel holds the likelihood blocks of this context: 0 / 1 / 2 as fill_table_host's wmode; -1: does not fit
inline int twist_wmode(const btf_ctx* c, bool wt) {
  if (tw_lds_bytes(c->T, c->K, c->TF, wt ? 1 : 0) <= 160 * 1024) return wt ? 1 : 0;
  if (wt && tw_lds_bytes(c->T, c->K, c->TF, 2) <= 160 * 1024) return 2;
  return -1;
}
int make_fill_table(btf_ctx* c, bool weighted) {
  const int key = twist_wmode(c, weighted);
  if (c->fill_tab && c->fill_key == key) return BTF_OK;
  std::vector<int> tab;
  fill_table_host(c->T, c->K, c->TF, key, tab);
  // {dst, src, dia, 0} -> the 8-byte entries the kernel reads (tw_fill_unpack): dst | dia16 << 16, src
  std::vector<int> packed(tab.size() / 2);
  for (size_t e = 0; e + 3 < tab.size(); e += 4) {
    const int dst = tab[e], src = tab[e + 1], dia = tab[e + 2];
    if (dst < 0 || dst >= 0xffff || dia >= 0xffff) return fail(c, BTF_EINVAL, "band assembly program: LDS offset beyond 16 bits");
    packed[e / 2] = (int)((unsigned)dst | ((unsigned)(dia < 0 ? 0xffff : dia) << 16));
    packed[e / 2 + 1] = src;
  }
  int rc;
  if ((rc = dev_alloc(c, &c->fill_tab, packed.size()))) return rc;
  HIPCHK(c, hipMemcpy(c->fill_tab, packed.data(), packed.size() * sizeof(int), hipMemcpyHostToDevice));
  c->fill_n = (int)(tab.size() / 4);
  c->fill_key = key;
  return BTF_OK;
}
hipError_t dispatch_vbanded_twist(btf_ctx* c, const VBandArgs& a, int bw, size_t lds_bytes, bool* handled) {
  const int npairs = (bw - 1) * (bw - 2) / 2;
  const int npl = std::max(1, (npairs + WAVE - 1) / WAVE);
  *handled = true;
  if (c->TF == 2 && c->T == 64 && (c->K == 5 || c->K == 8)) {      // ... and the depth axis of BASELINE configs 3 / 4 / 5
    if (c->K == 5) return launch_vbanded_twist<tw_npl(5, 2), tw_row16(5, 2), 5, 2, 64>(c, a, lds_bytes);
    return launch_vbanded_twist<tw_npl(8, 2), tw_row16(8, 2), 8, 2, 64>(c, a, lds_bytes);
  }
  if (c->TF == 2 && bw == 3 * c->K) {        // the instances with (nembeds, tf_order = 2) compiled in
#define TW_FIXED(KV) case KV: return launch_vbanded_twist<tw_npl(KV, 2), tw_row16(KV, 2), KV, 2>(c, a, lds_bytes)
    switch (c->K) { TW_FIXED(1); TW_FIXED(2); TW_FIXED(3); TW_FIXED(4); TW_FIXED(5); TW_FIXED(6); TW_FIXED(7); TW_FIXED(8); TW_FIXED(9); TW_FIXED(10); default: break; }
#undef TW_FIXED
  }
  if (bw <= 15) {
    if (npl == 1) return launch_vbanded_twist<1, true>(c, a, lds_bytes);
    if (npl == 2) return launch_vbanded_twist<2, true>(c, a, lds_bytes);
  } else {
    switch (npl) {
      case 2: return launch_vbanded_twist<2, false>(c, a, lds_bytes);
      case 3: return launch_vbanded_twist<3, false>(c, a, lds_bytes);
      case 4: return launch_vbanded_twist<4, false>(c, a, lds_bytes);
      case 5: return launch_vbanded_twist<5, false>(c, a, lds_bytes);
      case 6: return launch_vbanded_twist<6, false>(c, a, lds_bytes);
      case 7: return launch_vbanded_twist<7, false>(c, a, lds_bytes);
      case 8: return launch_vbanded_twist<8, false>(c, a, lds_bytes);
      default: break;
    }
  }
  *handled = false;
  return hipSuccess;
}
// which sampler a V half-sweep of this context will use: 3 spectral, 2 twisted, 1 single chain (LDS), 4 single chain in
// chunks (band too large for LDS in one piece), -1 generic
int banded_choice_for(const btf_ctx* c, bool wt, bool allow_spectral) {
  const int bw = (c->TF + 1) * c->K;
  if (allow_spectral && c->sampler == BTF_SAMPLER_SPECTRAL && !wt && !c->binomial && c->st_dense_ok &&
      vs_lds_bytes(c->T, c->K, c->TF, c->nD, true) <= 160 * 1024) return 3;        // (pivot records in LDS, or in HBM scratch for long depth axes)
  if (c->sampler == BTF_SAMPLER_GENERIC || bw < 3) return -1;
  if (c->sampler != BTF_SAMPLER_CHAIN && twist_ok(c->T, c->K, c->TF) && twist_wmode(c, wt) >= 0) return 2;
  if (vb_fast_lds_bytes(c->T, c->K, c->TF, wt) <= 158 * 1024) return 1;
  if (c->sampler != BTF_SAMPLER_GENERIC && vc_chunk_for(c, wt) > 0) return 4;
  return -1;
}
// curve-structured counts are handled as complete data plus corrections when nothing is stale, the per-column
// Grams of V fit w_solve's budget and the sampler is one that knows the per-column Gram (spectral, twisted)
bool curve_on(const btf_ctx* c) {
  if (!c->curve || !c->curve_opt || c->binomial || !c->weighted || c->stale_w || c->stale_v) return false;
  if (c->nl != c->N || c->ml != c->M) return false;
  if ((size_t)c->ml * c->KK + 16 * c->KK > ws_gram_stage(c->K, false)) return false;
  const int ch = banded_choice_for(c, false, true);       // (spectral or twisted: the samplers that take the per-column Gram)
  return ch == 2 || ch == 3;
}
// does the likelihood part run its weighted form (per-cell weights streamed, per-depth Gram blocks)?
bool lik_weighted(const btf_ctx* c) { return c->weighted && !curve_on(c); }
int banded_choice(const btf_ctx* c, bool allow_spectral = true) { return banded_choice_for(c, lik_weighted(c), allow_spectral); }
template <int S, bool RG, int KC = 0>
hipError_t launch_vspectral(btf_ctx* c, const VSpecArgs& a, size_t lds_bytes) {
  static std::atomic<unsigned long long> attr_set{0};      // one bit per device: the attribute is per device
  if (!dev_flag_is_set(attr_set, c->dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)v_spectral_kernel<S, RG, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    dev_flag_set(attr_set, c->dev);
  }
  Prof p(c, BTF_K_V_BANDED);
  p.launch(v_spectral_kernel<S, RG, KC>, dim3(a.ml), dim3(VS_THREADS), lds_bytes, a);
  return hipSuccess;
}
template <int K>
void launch_sse(btf_ctx* c, const double* A, const double* C, double Rc, int ncols, int ld, int rpb, int nrb) {
  Prof p(c, BTF_K_SSE);
  dim3 grid((ncols + SSE_THREADS - 1) / SSE_THREADS, nrb);
  if (c->C8_v) p.launch(sse_kernel<K, unsigned char>, grid, dim3(SSE_THREADS), 0, A, (const unsigned char*)c->C8_v, Rc, (const double*)c->W,
                        (const double*)c->V, c->N, ncols, ld, rpb, (size_t)c->col0 * c->T, c->bsum);
  else p.launch(sse_kernel<K>, grid, dim3(SSE_THREADS), 0, A, C, Rc, (const double*)c->W, (const double*)c->V, c->N, ncols, ld, rpb,
                (size_t)c->col0 * c->T, c->bsum);
}

// log-likelihood partials of the current W, V: rows layout (joint chains and per-row chains) or columns layout
template <int K>
void launch_ess_ll(btf_ctx* c, int what, int mode, int link, int nbx) {
  Prof p(c, BTF_K_ESS);
  const double Rc = (double)c->R;
  const LikFam lf{link, c->lik_par[link]};                 // `link` is the likelihood family (ESS_FAM_*)
  const int tl = ess_link_of(link);                        // the kernel instantiation that evaluates it
  const unsigned char* c8v = c->C8_v; const unsigned char* c8w = c->C8_wT;
  if (what == 1 && mode == 1) {      // per-column chains: W layout
    dim3 grid(nbx, c->M);
#define ESS_COLS(LINK_)                                                                                                   \
    if (c8w) p.launch(poisson_ll_cols_kernel<K, LINK_, unsigned char>, grid, dim3(ESS_THREADS), 0, (const double*)c->A_wT, c8w, Rc, \
                      (const double*)c->W, (const double*)c->V, 0, c->N, c->ldw, 0, c->T, (const int*)c->ess_done, c->ess_part, lf);   \
    else p.launch(poisson_ll_cols_kernel<K, LINK_, double>, grid, dim3(ESS_THREADS), 0, (const double*)c->A_wT, (const double*)c->C_wT, Rc, \
                  (const double*)c->W, (const double*)c->V, 0, c->N, c->ldw, 0, c->T, (const int*)c->ess_done, c->ess_part, lf);
    if (tl == ESS_LINK_LOG) { ESS_COLS(ESS_LINK_LOG) } else if (tl == ESS_LINK_IDENTITY) { ESS_COLS(ESS_LINK_IDENTITY) } else { ESS_COLS(ESS_LINK_GENERIC) }
#undef ESS_COLS
  } else {
    dim3 grid(nbx, c->N);
    const int per_row = (what == 0 && mode == 1) ? 1 : 0;
#define ESS_ROWS(LINK_)                                                                                                   \
    if (c8v) p.launch(poisson_ll_rows_kernel<K, LINK_, unsigned char>, grid, dim3(ESS_THREADS), 0, (const double*)c->A_v, c8v, Rc,  \
                      (const double*)c->W, (const double*)c->V, 0, c->M * c->T, c->ldv, (size_t)0, (const int*)c->ess_done, per_row, c->ess_part, lf); \
    else p.launch(poisson_ll_rows_kernel<K, LINK_, double>, grid, dim3(ESS_THREADS), 0, (const double*)c->A_v, (const double*)c->C_v, Rc, \
                  (const double*)c->W, (const double*)c->V, 0, c->M * c->T, c->ldv, (size_t)0, (const int*)c->ess_done, per_row, c->ess_part, lf);
    if (tl == ESS_LINK_LOG) { ESS_ROWS(ESS_LINK_LOG) } else if (tl == ESS_LINK_IDENTITY) { ESS_ROWS(ESS_LINK_IDENTITY) } else { ESS_ROWS(ESS_LINK_GENERIC) }
#undef ESS_ROWS
  }
}

#define K_SWITCH(K, CALL)                                          \
  switch (K) {                                                     \
    case 1: { constexpr int KT = 1; CALL; } break;                 \
    case 2: { constexpr int KT = 2; CALL; } break;                 \
    case 3: { constexpr int KT = 3; CALL; } break;                 \
    case 4: { constexpr int KT = 4; CALL; } break;                 \
    case 5: { constexpr int KT = 5; CALL; } break;                 \
    case 6: { constexpr int KT = 6; CALL; } break;                 \
    case 7: { constexpr int KT = 7; CALL; } break;                 \
    case 8: { constexpr int KT = 8; CALL; } break;                 \
    case 9: { constexpr int KT = 9; CALL; } break;                 \
    case 10: { constexpr int KT = 10; CALL; } break;               \
    default: break;                                                \
  }

int check_status(btf_ctx* c) {
  int st[2] = {0, -1};
  HIPCHK(c, hipMemcpyAsync(st, c->status, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (st[0] != 0) {
    c->fail_index = st[1];
    int zero[2] = {0, -1};
    HIPCHK(c, hipMemcpy(c->status, zero, sizeof(zero), hipMemcpyHostToDevice));
    if (st[0] == 2) return fail(c, BTF_EHIP, "a hand-off inside a fused launch timed out (btf_fused.h): its producer workgroup never published");
    if (st[0] == 3) return fail(c, BTF_EHIP, "the peer-window exchange timed out waiting for rank " + std::to_string(st[1]) + " (btf_comm.h; BTF_PEER_TIMEOUT_MS)");
    return fail(c, BTF_ENOTPD, "conditional precision not positive definite at index " + std::to_string(st[1]));
  }
  return BTF_OK;
}

int ensure_part(btf_ctx* c, size_t elems) {
  if (elems > c->part_elems) {
    int rc = dev_alloc(c, &c->part, elems);
    if (rc) return rc;
    c->part_elems = elems;
  }
  return BTF_OK;
}

int ensure_z(btf_ctx* c, size_t elems) {
  if (elems > c->z_elems) {
    int rc = dev_alloc(c, &c->zbuf, elems);
    if (rc) return rc;
    c->z_elems = elems;
  }
  return BTF_OK;
}

// `slots`: workgroups of the accumulation kernel the chip holds at once (one 12- or 16-wave workgroup per CU at its
// 121-128 VGPRs; 0 = unknown), `reserve`: side workgroups the launch will put in front of the streaming ones.
int pick_rpb(int Rdim, int tiles, int user, bool weighted, int slots = 0, int reserve = 0) {
  // rows per workgroup (a multiple of 64): enough workgroups to fill the chip twice over when the problem is that
  // large, but never fewer rows than pay for a workgroup's prologue and its share of the partials - measured at
  // C3 (scripts/ab_rpb.sh): complete data 512 rows (128 workgroups: W 11.5 us, V 12.8 -> 11.2 us, and w_solve
  // sums half as many chunks), weighted data 256 rows (V 19.4 -> 15.5 us; 512 is slower there)
  if (user > 0) return std::max(64, round_up(user, 64));
  const long long want_wgs = tiles >= 32 ? 512 : 256;
  long long rpb = ((long long)Rdim * tiles + want_wgs - 1) / want_wgs;
  rpb = std::max<long long>(rpb, weighted ? 256 : 512);
  rpb = std::min<long long>(rpb, std::max(Rdim, 1));
  rpb = std::max(128, round_up((int)rpb, 64));
  // One round.  A launch of one to two rounds of workgroups (a rank's slab of a sharded run: 256 streaming workgroups
  // + 32 Gram side workgroups, or 512 + 1) ends in a round that is mostly tail: every CU holds ONE workgroup, the side
  // workgroups keep theirs for 10-25 us, and the streaming workgroups they displace start late and finish alone, at
  // the rate of a few lone CUs (scripts/acc_stamps.sh: 57 us for a 268 MB slab whose workgroups stream at 7 TB/s
  // while they all run).  So when the launch can fit the chip in one round - side workgroups included - it does.
  static const bool fit = [] { const char* e = std::getenv("BTF_FIT_ROUND"); return !e || std::atoi(e) != 0; }();
  if (fit && slots > 0) {
    // (cost model: rounds x rows per workgroup; only a strict gain moves the rule - the weighted launches at C3 are
    //  issue-bound and want all 256 CUs: 2 x 128 workgroups of 256 rows beat 128 of 512, 15.4 against 21.3 us)
    const long long nch = (Rdim + rpb - 1) / rpb, total = nch * tiles + reserve;
    const long long rounds = (total + slots - 1) / slots;
    const long long nch1 = (slots - reserve) / std::max(tiles, 1);
    if (rounds >= 2 && rounds <= 3 && nch1 >= 1) {
      const long long r1 = round_up((int)((Rdim + nch1 - 1) / nch1), 64);
      if (r1 < rounds * rpb) rpb = r1;
    }
  }
  return (int)rpb;
}
// slab extents: the rank's block plus the halo source (btf_set_shard_halo)
inline int slab_rows(const btf_ctx* c) { return c->nl + (c->hrow >= 0 ? 1 : 0); }
inline int slab_cols(const btf_ctx* c) { return c->ml + (c->hcol >= 0 ? 1 : 0); }
// (w_launch: the W half-sweep's launches carry no eigen side task and run the 16-wave instance at nembeds 10 too)
inline int acc_slots(const btf_ctx* c, int K, int mode, bool w_launch = false) {
  const int waves = (w_launch && mode == 0 && K >= 10) ? ACC_WAVES : acc_waves(K, mode);
  return c->ncu * std::max(1, 16 / waves);
}
// upper bounds of the side workgroups the two accumulation launches put in front (launch_accum counts them exactly)
inline int w_side_reserve(const btf_ctx* c, bool wt) {
  const bool whole = c->nl == c->N && c->ml == c->M;
  const bool gram_side = !wt && !(whole && c->fuse_gram && c->ngp_v > 0) && !c->weighted;
  return (gram_side ? 32 : 0) + (c->tau_pending && c->dev_scalars && c->have_chain ? (c->ml + TAU_SIDE_CPW - 1) / TAU_SIDE_CPW : 0) +
         (c->sc_pending ? 1 : 0);
}
inline int v_side_reserve(const btf_ctx* c, bool wt) {
  const bool spectral = banded_choice(c) == 3;               // the eigen side tasks ride with the spectral sampler only
  return (spectral ? 1 + (c->weighted && !wt ? (c->cv_ndef + 3) / 4 : 0) : 0) + (c->lam_pending ? 1 : 0);
}

// upload a host slab and turn it into the padded device layouts
// `keep`: device copies of the uploaded slab(s) from the previous call on the same host arrays (the unsharded
// case hands in one array as both the row and the column slab): uploaded once, freed by the second call.
struct Uploaded { const double* h = nullptr; const double* h2 = nullptr; double* d = nullptr; double* d2 = nullptr; };
int make_stats(btf_ctx* c, const double* hY, const double* hY2, int rows, int cols, int R, bool transposed,
               double** A, double** C, double** B, int ld, size_t out_rows, bool want_sums, Uploaded* keep, bool last, int sum_rows = -1) {
  const size_t cells = (size_t)rows * cols;
  double* dY = nullptr; double* dY2 = nullptr;
  int rc;
  const bool reuse = keep && keep->d && keep->h == hY && keep->h2 == hY2;
  if (reuse) { dY = keep->d; dY2 = keep->d2; }
  else {
    if ((rc = dev_alloc(c, &dY, cells * R))) return rc;
    HIPCHK(c, hipMemcpy(dY, hY, cells * R * sizeof(double), hipMemcpyHostToDevice));
    if (hY2) {
      if ((rc = dev_alloc(c, &dY2, cells))) return rc;
      HIPCHK(c, hipMemcpy(dY2, hY2, cells * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  const size_t out_elems = out_rows * (size_t)ld;
  if ((rc = dev_alloc(c, A, out_elems))) return rc;
  HIPCHK(c, hipMemsetAsync(*A, 0, out_elems * sizeof(double), c->stream));
  if ((rc = dev_alloc(c, C, out_elems))) return rc;
  HIPCHK(c, hipMemsetAsync(*C, 0, out_elems * sizeof(double), c->stream));
  int* dflag = c->status + 2;
  const int blocks = (int)std::min<size_t>(4096, (cells + 255) / 256);
  if (want_sums) {
    if ((size_t)blocks * 3 > c->bsum_elems) {
      if ((rc = dev_alloc(c, &c->bsum, (size_t)blocks * 3))) return rc;
      c->bsum_elems = (size_t)blocks * 3;
    }
  }
  StatsArgs a{dY, dY2, rows, cols, R, ld, transposed ? 1 : 0, *A, *C, want_sums ? c->bsum : nullptr, dflag, sum_rows < 0 ? rows : sum_rows};
  {
    Prof p(c, BTF_K_STATS);
    p.launch(stats_kernel, dim3(blocks), dim3(256), 0, a);
  }
  HIPCHK(c, hipGetLastError());
  if (B && hY2) {  // keep the trial counts for the PG draw: B = C at this point
    if ((rc = dev_alloc(c, B, out_elems))) return rc;
    HIPCHK(c, hipMemcpyAsync(*B, *C, out_elems * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (want_sums) {
    std::vector<double> h((size_t)blocks * 3);
    HIPCHK(c, hipMemcpy(h.data(), c->bsum, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    double ssw = 0.0, nobs = 0.0, sa2 = 0.0;
    for (int b = 0; b < blocks; ++b) { ssw += h[3 * b]; nobs += h[3 * b + 1]; sa2 += h[3 * b + 2]; }
    c->ssw = ssw;
    c->nobs = nobs;
    c->sa2 = sa2;
  }
  if (keep && !last) { keep->h = hY; keep->h2 = hY2; keep->d = dY; keep->d2 = dY2; }
  else {
    (void)hipFree(dY);
    if (dY2) (void)hipFree(dY2);
    if (keep) *keep = Uploaded{};
  }
  return BTF_OK;
}

int upload_relayout(btf_ctx* c, const double* h, int rows, int cols, double* dst, int ld, bool transposed) {
  const size_t cells = (size_t)rows * cols;
  double* d = nullptr;
  int rc;
  if ((rc = dev_alloc(c, &d, cells))) return rc;
  HIPCHK(c, hipMemcpy(d, h, cells * sizeof(double), hipMemcpyHostToDevice));
  const int blocks = (int)std::min<size_t>(4096, (cells + 255) / 256);
  hipLaunchKernelGGL(relayout_kernel, dim3(blocks), dim3(256), 0, c->stream, d, rows, cols, dst, ld, transposed ? 1 : 0, 1);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d);
  return BTF_OK;
}

}  // namespace

// =============================================================================
extern "C" {

int btf_create(btf_ctx** out, int nrows, int ncols, int ndepth, int nembeds, int tf_order, int device, void* stream) {
  if (!out) return BTF_EINVAL;
  *out = nullptr;
  if (nrows < 1 || ncols < 1 || ndepth < 2 || nembeds < 1 || nembeds > MAX_K || tf_order < 0 || tf_order > 3)
    return fail(nullptr, BTF_EINVAL, "unsupported dims (need 1<=nembeds<=10, 0<=tf_order<=3, ndepth>=2)");
  if ((tf_order + 1) * nembeds > 63) return fail(nullptr, BTF_EINVAL, "half-bandwidth (tf_order+1)*nembeds must be <= 63");
  btf_ctx* c = new btf_ctx();
  c->N = nrows; c->M = ncols; c->T = ndepth; c->K = nembeds; c->TF = tf_order; c->KK = tri(nembeds);
  c->dev = device;
  c->row0 = 0; c->nl = nrows; c->col0 = 0; c->ml = ncols;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { g_err = std::string("hipSetDevice: ") + hipGetErrorString(e); delete c; return BTF_EHIP; }
  { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) c->ncu = n; }
  { const char* e = std::getenv("BTF_FUSED_STEP"); if (e) c->fused_step = std::max(0, std::min(2, std::atoi(e))); }      // (A/B aid; BTF_OPT_FUSED_STEP is the interface)
  { const char* e = std::getenv("BTF_LAM_IN_WSOLVE"); if (e) c->lam_in_wsolve = std::atoi(e) != 0; }
  { const char* e = std::getenv("BTF_BAND_IN_WSOLVE"); if (e) c->band_in_wsolve = std::atoi(e) != 0; }
  { const char* e = std::getenv("BTF_VF_DATAFLOW"); if (e) c->fused_dataflow = std::atoi(e) != 0 ? 1 : 0; }             // (A/B aid; BTF_OPT_FUSED_DATAFLOW is the interface)
  if (stream) { c->stream = (hipStream_t)stream; }
  else {
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_err = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete c; return BTF_EHIP; }
    c->own_stream = true;
  }
  int rc = BTF_OK;
  auto A = [&](int r) { if (rc == BTF_OK) rc = r; };
  A(dev_alloc(c, &c->W, (size_t)(nrows + 64) * nembeds));        // +64: room for a padded all-gather
  A(dev_alloc(c, &c->V, (size_t)(ncols + 64) * ndepth * nembeds));
  A(dev_alloc(c, &c->gpart, (size_t)GRAM_BLOCKS * c->KK));
  A(dev_alloc(c, &c->status, 4));
  if (rc == BTF_OK) {
    int init[4] = {0, -1, 0, 0};
    if (hipMemcpy(c->status, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) rc = BTF_EHIP;
  }
  if (rc == BTF_OK) rc = build_stencil(c);
  A(dev_alloc(c, &c->Tau2, (size_t)ncols * c->nD));
  A(dev_alloc(c, &c->tries, (size_t)ncols));
  if (rc != BTF_OK) { std::string m = c->err; btf_destroy(c); g_err = m; return rc; }
  *out = c;
  return BTF_OK;
}

void btf_destroy(btf_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->dev);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  void* ptrs[] = {c->A_wT, c->C_wT, c->A_v, c->C_v, c->B_wT, c->B_v, c->W, c->V, c->Tau2, c->part,
                  c->gpart, c->zbuf, c->bsum, c->gband, c->status, c->tries, c->st_ptr, c->st_row, c->st_coef,
                  c->srcmap_w, c->srcmap_v, c->pband, c->pimg, c->dbg, c->gpart_w, c->gpart_v, c->gsum_v, c->eig, c->cv_cptr, c->cv_crow, c->cv_cdef, c->cv_rptr, c->cv_rcol, c->cv_rdef, c->eig_cols, c->cv_dcols, c->A8_wT, c->A8_v, c->gs_cons, c->gs_cc, c->gs_rc, c->gs_av, c->gs_mask, c->gs_info, c->gs_thetas, c->gs_ntheta, c->gs_ll, c->gs_llp, c->gs_hh, c->gs_cur, c->gs_nacc, c->gs_u, c->st_drow, c->st_dcoef, c->essX0, c->essNu, c->ess_st, c->ess_theta, c->ess_done, c->ess_part, c->Ta, c->Tb, c->Tc, c->lsum, c->dr_ptr, c->dr_col, c->dr_val, c->sse_cols, c->vs_rec, c->vc_scratch, c->gs_cptr, c->gs_cidx, c->gs_cval};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (c->pin) (void)hipHostFree(c->pin);
  if (c->pin_lsum) (void)hipHostFree(c->pin_lsum);
  if (c->pin_hyp) (void)hipHostFree(c->pin_hyp);
  for (void* p : {(void*)c->nb_data, (void*)c->nb_S, (void*)c->nb_cnt, (void*)c->nb_R, (void*)c->nb_C, (void*)c->nb_tmp, (void*)c->nb_out, (void*)c->nb_H, (void*)c->nb_Hd, (void*)c->nb_Hs, (void*)c->nb_G, (void*)c->nb_L, (void*)c->nb_optr, (void*)c->nb_oval, (void*)c->fill_tab, (void*)c->C8_wT, (void*)c->C8_v, (void*)c->smp_W, (void*)c->smp_V, (void*)c->smp_T, (void*)c->smp_s})
    if (p) (void)hipFree(p);
  if (c->hyp) (void)hipFree(c->hyp);
  if (c->fz_words) (void)hipFree(c->fz_words);
  if (c->fz_pub) (void)hipFree(c->fz_pub);
  for (auto& e : c->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  if (c->ev_draw) (void)hipEventDestroy(c->ev_draw);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  (void)btf_comm_destroy(c);

  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* btf_last_error(const btf_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }
int btf_fail_index(const btf_ctx* c) { return c ? c->fail_index : -1; }

int btf_set_shard(btf_ctx* c, int row0, int nrows_local, int col0, int ncols_local) {
  if (!c) return BTF_EINVAL;
  if (c->have_data) return fail(c, BTF_ESTATE, "btf_set_shard must precede btf_set_data_*");
  if (row0 < 0 || nrows_local < 0 || row0 + nrows_local > c->N || col0 < 0 || ncols_local < 0 || col0 + ncols_local > c->M)
    return fail(c, BTF_EINVAL, "shard out of range");
  c->row0 = row0; c->nl = nrows_local; c->col0 = col0; c->ml = ncols_local;
  c->hrow = c->hcol = -1;
  ++c->prior_version;
  if (c->pband) { (void)hipFree(c->pband); c->pband = nullptr; c->pband_version = 0; }      // (sized for the old column block)
  if (c->pimg) { (void)hipFree(c->pimg); c->pimg = nullptr; c->pimg_version = 0; }
  c->v_wants_band = false;        // (until a V half-sweep of the new geometry has sized the band again)
  c->nb_bwt_written = false;      // (the skip of nb_bwt_target is only valid for the shard geometry B_wT was written under)
  return BTF_OK;
}
int btf_set_shard_halo(btf_ctx* c, int halo_row, int halo_col) {
  if (!c) return BTF_EINVAL;
  if (halo_row < -1 || halo_row >= c->N || halo_col < -1 || halo_col >= c->M) return fail(c, BTF_EINVAL, "halo source out of range");
  if ((halo_row >= c->row0 && halo_row < c->row0 + c->nl) || (halo_col >= c->col0 && halo_col < c->col0 + c->ml))
    return fail(c, BTF_EINVAL, "halo source inside the shard's own block");
  if (halo_row != c->hrow || halo_col != c->hcol) {        // the slabs change shape: whatever was uploaded is gone
    c->have_data = false;
    c->nb_bwt_written = false;
  }
  c->hrow = halo_row; c->hcol = halo_col;
  return BTF_OK;
}
void* btf_stream(btf_ctx* c) { return c ? (void*)c->stream : nullptr; }
void* btf_dev_W(btf_ctx* c) { return c ? c->W : nullptr; }
void* btf_dev_V(btf_ctx* c) { return c ? c->V : nullptr; }

// Are the replicate counts constant along the depth axis (whole curves missing or thinned)?  Then keep c_ij and
// the lists of deficient partners (see CurveLists).  Unsharded contexts only.
static int detect_curve_counts(btf_ctx* c) {
  c->curve = false;
  c->cv_cij.clear();
  if (!c->C8_v || c->nl != c->N || c->ml != c->M) return BTF_OK;
  const int N = c->N, M = c->M, T = c->T;
  std::vector<unsigned char> h((size_t)N * c->ldv);
  HIPCHK(c, hipMemcpy(h.data(), c->C8_v, h.size(), hipMemcpyDeviceToHost));
  std::vector<unsigned char> cij((size_t)N * M);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) {
      const unsigned char* p = h.data() + (size_t)i * c->ldv + (size_t)j * T;
      const unsigned char v = p[0];
      for (int t = 1; t < T; ++t) if (p[t] != v) return BTF_OK;          // varies with depth: the general weighted path
      cij[(size_t)i * M + j] = v;
    }
  std::vector<int> cptr(M + 1, 0), rptr(N + 1, 0);
  size_t nnz = 0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) if (cij[(size_t)i * M + j] < c->R) { ++cptr[j + 1]; ++rptr[i + 1]; ++nnz; }
  if (nnz == 0 || nnz > (size_t)N * M / 4) return BTF_OK;                 // (many deficient curves: the corrections stop being cheap)
  for (int j = 0; j < M; ++j) cptr[j + 1] += cptr[j];
  for (int i = 0; i < N; ++i) rptr[i + 1] += rptr[i];
  std::vector<int> crow(nnz), rcol(nnz), cfill(cptr.begin(), cptr.end() - 1), rfill(rptr.begin(), rptr.end() - 1);
  std::vector<double> cdef(nnz), rdef(nnz);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) {
      const int v = cij[(size_t)i * M + j];
      if (v >= c->R) continue;
      const double d = (double)(c->R - v);
      crow[cfill[j]] = i; cdef[cfill[j]++] = d;
      rcol[rfill[i]] = j; rdef[rfill[i]++] = d;
    }
  int rc;
  if ((rc = dev_alloc(c, &c->cv_cptr, cptr.size()))) return rc;
  if ((rc = dev_alloc(c, &c->cv_crow, nnz))) return rc;
  if ((rc = dev_alloc(c, &c->cv_cdef, nnz))) return rc;
  if ((rc = dev_alloc(c, &c->cv_rptr, rptr.size()))) return rc;
  if ((rc = dev_alloc(c, &c->cv_rcol, nnz))) return rc;
  if ((rc = dev_alloc(c, &c->cv_rdef, nnz))) return rc;
  HIPCHK(c, hipMemcpy(c->cv_cptr, cptr.data(), cptr.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->cv_crow, crow.data(), nnz * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->cv_cdef, cdef.data(), nnz * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->cv_rptr, rptr.data(), rptr.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->cv_rcol, rcol.data(), nnz * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->cv_rdef, rdef.data(), nnz * sizeof(double), hipMemcpyHostToDevice));
  std::vector<int> dcols;
  for (int j = 0; j < M; ++j) if (cptr[j + 1] > cptr[j]) dcols.push_back(j);
  if ((rc = dev_alloc(c, &c->cv_dcols, dcols.size()))) return rc;
  HIPCHK(c, hipMemcpy(c->cv_dcols, dcols.data(), dcols.size() * sizeof(int), hipMemcpyHostToDevice));
  c->cv_ndef = (int)dcols.size();
  const size_t ne = (size_t)M * (c->K + c->K * c->K + 8);
  if ((rc = dev_alloc(c, &c->eig_cols, ne))) return rc;
  HIPCHK(c, hipMemset(c->eig_cols, 0, ne * sizeof(double)));               // no previous solutions
  c->cv_cij.swap(cij);
  c->curve = true;
  return BTF_OK;
}

static int finish_data(btf_ctx* c) {
  int flag = 0;
  HIPCHK(c, hipMemcpy(&flag, c->status + 2, sizeof(int), hipMemcpyDeviceToHost));
  c->weighted = c->binomial || flag != 0;
  c->curve = false; c->cv_cij.clear();
  if (c->A8_wT) { (void)hipFree(c->A8_wT); c->A8_wT = nullptr; }
  if (c->A8_v) { (void)hipFree(c->A8_v); c->A8_v = nullptr; }
  if (c->C8_wT) { (void)hipFree(c->C8_wT); c->C8_wT = nullptr; }
  if (c->C8_v) { (void)hipFree(c->C8_v); c->C8_v = nullptr; }
  if (!c->weighted) {  // complete Gaussian data: counts are the constant R, drop them
    (void)hipFree(c->C_wT); c->C_wT = nullptr;
    (void)hipFree(c->C_v); c->C_v = nullptr;
  } else if (!c->binomial && c->R <= 255) {   // replicate counts 0..R: one byte per cell instead of eight
    const size_t ew = (size_t)c->M * c->T * c->ldw, ev = (size_t)c->N * c->ldv;
    int rc;
    if ((rc = dev_alloc(c, &c->C8_wT, ew))) return rc;
    if ((rc = dev_alloc(c, &c->C8_v, ev))) return rc;
    hipLaunchKernelGGL(f64_to_u8_kernel, dim3((unsigned)std::min<size_t>(4096, (ew + 255) / 256)), dim3(256), 0, c->stream, (const double*)c->C_wT, c->C8_wT, ew);
    hipLaunchKernelGGL(f64_to_u8_kernel, dim3((unsigned)std::min<size_t>(4096, (ev + 255) / 256)), dim3(256), 0, c->stream, (const double*)c->C_v, c->C8_v, ev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->C_wT); c->C_wT = nullptr;
    (void)hipFree(c->C_v); c->C_v = nullptr;
    int rc2 = detect_curve_counts(c);
    if (rc2) return rc2;
  }
  c->have_data = true;
  c->w_part_valid = false; c->w_local_done = c->v_local_done = false; c->sse_cols_valid = false;
  return BTF_OK;
}

int btf_set_data_gaussian(btf_ctx* c, const double* y_rows, const double* y_cols, int nreps) {
  if (!c || !y_rows || !y_cols || nreps < 1) return fail(c, BTF_EINVAL, "bad data arguments");
  HIPCHK(c, hipSetDevice(c->dev));
  c->R = nreps; c->binomial = false; c->counts = false;
  const int MT = c->M * c->T;
  c->ldw = round_up(std::max(slab_rows(c), 1), ACC_TILE);
  c->ldv = round_up(std::max(slab_cols(c) * c->T, 1), ACC_TILE);
  int zero = 0;
  HIPCHK(c, hipMemcpy(c->status + 2, &zero, sizeof(int), hipMemcpyHostToDevice));
  int rc;
  // row slab -> transposed layout A_wT[MT][ldw] (W half-sweep); carries the global sums
  Uploaded up;
  const bool same = y_rows == y_cols && c->nl == c->N && c->ml == c->M;     // unsharded: one upload serves both layouts
  if ((rc = make_stats(c, y_rows, nullptr, slab_rows(c), MT, nreps, true, &c->A_wT, &c->C_wT, nullptr, c->ldw, MT, true, same ? &up : nullptr, false, c->nl))) return rc;
  // column slab -> A_v[N][ldv] (V half-sweep, SSE)
  if ((rc = make_stats(c, y_cols, nullptr, c->N, slab_cols(c) * c->T, nreps, false, &c->A_v, &c->C_v, nullptr, c->ldv, c->N, false, same ? &up : nullptr, true))) {
    if (up.d) (void)hipFree(up.d);
    return rc;
  }
  return finish_data(c);
}

int btf_set_data_binomial(btf_ctx* c, const double* succ_rows, const double* trials_rows, const double* succ_cols,
                          const double* trials_cols) {
  if (!c || !succ_rows || !trials_rows || !succ_cols || !trials_cols) return fail(c, BTF_EINVAL, "bad data arguments");
  HIPCHK(c, hipSetDevice(c->dev));
  c->R = 1; c->binomial = true; c->counts = false;
  const int MT = c->M * c->T;
  c->ldw = round_up(std::max(slab_rows(c), 1), ACC_TILE);
  c->ldv = round_up(std::max(slab_cols(c) * c->T, 1), ACC_TILE);
  int zero = 0;
  HIPCHK(c, hipMemcpy(c->status + 2, &zero, sizeof(int), hipMemcpyHostToDevice));
  int rc;
  Uploaded up;
  const bool same = succ_rows == succ_cols && trials_rows == trials_cols && c->nl == c->N && c->ml == c->M;
  if ((rc = make_stats(c, succ_rows, trials_rows, slab_rows(c), MT, 1, true, &c->A_wT, &c->C_wT, &c->B_wT, c->ldw, MT, true, same ? &up : nullptr, false, c->nl))) return rc;
  if ((rc = make_stats(c, succ_cols, trials_cols, c->N, slab_cols(c) * c->T, 1, false, &c->A_v, &c->C_v, &c->B_v, c->ldv, c->N, false, same ? &up : nullptr, true))) {
    if (up.d) (void)hipFree(up.d);
    if (up.d2) (void)hipFree(up.d2);
    return rc;
  }
  {   // which samplers the trial counts need (see pg_passes); both slabs hold every class a rank can meet
    bool small = false, big = false, frac = false;
    auto scan = [&](const double* t, size_t n) {
      for (size_t i = 0; i < n && !(small && big && frac); ++i) {
        const double b = t[i];
        if (!(b > 0.0) || b >= (double)PG_NORMAL_B) continue;
        if (b != std::floor(b)) frac = true;
        else if (b <= (double)PG_AUTO_EXACT_MAX) small = true;
        else big = true;
      }
    };
    scan(trials_cols, (size_t)c->N * slab_cols(c) * c->T);
    scan(trials_rows, (size_t)slab_rows(c) * MT);
    c->pg_has_small = small; c->pg_has_big = big; c->pg_has_frac = frac;
  }
  // until the first PG draw / set_omega the weights are zero
  HIPCHK(c, hipMemset(c->C_wT, 0, (size_t)MT * c->ldw * sizeof(double)));
  HIPCHK(c, hipMemset(c->C_v, 0, (size_t)c->N * c->ldv * sizeof(double)));
  if ((rc = finish_data(c))) return rc;
  {   // integer counts: the pseudo-data kappa = Y - N/2 as one byte per cell for the accumulation launches
    const size_t ew = (size_t)MT * c->ldw, ev = (size_t)c->N * c->ldv;
    if ((rc = dev_alloc(c, &c->A8_wT, ew))) return rc;
    if ((rc = dev_alloc(c, &c->A8_v, ev))) return rc;
    HIPCHK(c, hipMemcpy(c->status + 3, &zero, sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kappa_to_i8_kernel, dim3((unsigned)std::min<size_t>(4096, (ew + 255) / 256)), dim3(256), 0, c->stream, (const double*)c->A_wT, c->A8_wT, ew, c->status + 3);
    hipLaunchKernelGGL(kappa_to_i8_kernel, dim3((unsigned)std::min<size_t>(4096, (ev + 255) / 256)), dim3(256), 0, c->stream, (const double*)c->A_v, c->A8_v, ev, c->status + 3);
    HIPCHK(c, hipGetLastError());
    int bad = 0;
    HIPCHK(c, hipMemcpyAsync(&bad, c->status + 3, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (bad) { (void)hipFree(c->A8_wT); (void)hipFree(c->A8_v); c->A8_wT = nullptr; c->A8_v = nullptr; }
  }
  return BTF_OK;
}

int btf_set_stale_sources(btf_ctx* c, const int32_t* src_row, const int32_t* src_col) {
  if (!c) return BTF_EINVAL;
  if (!c->have_data) return fail(c, BTF_ESTATE, "btf_set_stale_sources follows btf_set_data_*");
  HIPCHK(c, hipSetDevice(c->dev));
  int rc;
  std::vector<int> mw(c->ldw), mv(c->ldv);
  for (int l = 0; l < c->ldw; ++l) mw[l] = l;
  for (int l = 0; l < c->ldv; ++l) mv[l] = l;
  c->stale_w = c->stale_v = false;
  if (!c->weighted) src_row = src_col = nullptr;   // constant weights: staleness cannot change anything
  if (src_row)
    for (int il = 0; il < c->nl; ++il) {
      int s = src_row[c->row0 + il] - c->row0;
      if (src_row[c->row0 + il] == c->hrow) s = c->nl;                    // the halo slot (btf_set_shard_halo)
      else if (s < 0 || s >= c->nl) return fail(c, BTF_EINVAL, "stale weight source row outside this shard: declare it with btf_set_shard_halo before the upload (or use compat=exact)");
      if (s != il && c->curve && !std::memcmp(&c->cv_cij[(size_t)il * c->M], &c->cv_cij[(size_t)s * c->M], (size_t)c->M)) continue;   // same counts: nothing stale
      mw[il] = s;
      c->stale_w |= (s != il);
    }
  if (src_col)
    for (int jl = 0; jl < c->ml; ++jl) {
      int s = src_col[c->col0 + jl] - c->col0;
      if (src_col[c->col0 + jl] == c->hcol) s = c->ml;
      else if (s < 0 || s >= c->ml) return fail(c, BTF_EINVAL, "stale weight source column outside this shard: declare it with btf_set_shard_halo before the upload (or use compat=exact)");
      if (s != jl && c->curve) {
        bool same = true;
        for (int i = 0; i < c->N && same; ++i) same = c->cv_cij[(size_t)i * c->M + jl] == c->cv_cij[(size_t)i * c->M + s];
        if (same) continue;                                              // same counts: nothing stale
      }
      for (int t = 0; t < c->T; ++t) mv[jl * c->T + t] = s * c->T + t;
      c->stale_v |= (s != jl);
    }
  if ((rc = dev_alloc(c, &c->srcmap_w, mw.size()))) return rc;
  if ((rc = dev_alloc(c, &c->srcmap_v, mv.size()))) return rc;
  HIPCHK(c, hipMemcpy(c->srcmap_w, mw.data(), mw.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->srcmap_v, mv.data(), mv.size() * sizeof(int), hipMemcpyHostToDevice));
  return BTF_OK;
}

int btf_set_W(btf_ctx* c, const double* W) {
  if (!c || !W) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipMemcpyAsync(c->W, W, (size_t)c->N * c->K * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_W = true;
  c->nb_L_valid = false;
  c->ngp_w = 0;
  c->v_local_done = false;
  c->sse_cols_valid = false;
  return BTF_OK;
}
int btf_set_gathered_W(btf_ctx* c, const double* W) {
  if (!c) return BTF_EINVAL;
  const bool keep = c->v_local_done;
  const int rc = btf_set_W(c, W);
  if (rc == BTF_OK) c->v_local_done = keep;       // the caller's own rows are unchanged: what was accumulated from them stands
  return rc;
}
int btf_get_W(btf_ctx* c, double* W) {
  if (!c || !W) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipMemcpyAsync(W, c->W, (size_t)c->N * c->K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  return check_status(c);
}
int btf_set_V(btf_ctx* c, const double* V) {
  if (!c || !V) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipMemcpyAsync(c->V, V, (size_t)c->M * c->T * c->K * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_V = true;
  c->nb_L_valid = false;
  c->w_part_valid = false;
  c->ngp_v = 0;
  c->w_local_done = false;
  c->sse_cols_valid = false;
  return BTF_OK;
}
int btf_set_gathered_V(btf_ctx* c, const double* V) {
  if (!c) return BTF_EINVAL;
  const bool keep = c->w_local_done;
  const int rc = btf_set_V(c, V);
  if (rc == BTF_OK) c->w_local_done = keep;       // the caller's own columns are unchanged
  return rc;
}
int btf_get_V(btf_ctx* c, double* V) {
  if (!c || !V) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipMemcpyAsync(V, c->V, (size_t)c->M * c->T * c->K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  return check_status(c);
}
int btf_set_hyper(btf_ctx* c, const double* Tau2, double lam2, double sigma2) {
  if (!c) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (Tau2) {
    HIPCHK(c, hipMemcpyAsync(c->Tau2, Tau2, (size_t)c->M * c->nD * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_hyper = true;
  }
  if (Tau2 || lam2 != c->lam2) ++c->prior_version;        // (the host-RNG path re-sends the same scalars before every half-sweep)
  c->lam2 = lam2; c->sigma2 = sigma2;
  return BTF_OK;
}
int btf_set_tau_chain(btf_ctx* c, const double* Ta, const double* Tb, const double* Tc) {
  if (!c || !Ta || !Tb || !Tc) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  const size_t n = (size_t)c->M * c->nD;
  int rc;
  if (!c->Ta) {
    if ((rc = dev_alloc(c, &c->Ta, n))) return rc;
    if ((rc = dev_alloc(c, &c->Tb, n))) return rc;
    if ((rc = dev_alloc(c, &c->Tc, n))) return rc;
    if ((rc = dev_alloc(c, &c->lsum, (size_t)c->M))) return rc;
  }
  HIPCHK(c, hipMemcpyAsync(c->Ta, Ta, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->Tb, Tb, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->Tc, Tc, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_chain = true;
  return BTF_OK;
}

int btf_get_tau(btf_ctx* c, double* Tau2, double* Ta, double* Tb, double* Tc) {
  if (!c || !Tau2) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  const size_t n = (size_t)c->M * c->nD * sizeof(double);
  HIPCHK(c, hipMemcpyAsync(Tau2, c->Tau2, n, hipMemcpyDeviceToHost, c->stream));
  if (Ta && c->Ta) {
    HIPCHK(c, hipMemcpyAsync(Ta, c->Ta, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(Tb, c->Tb, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(Tc, c->Tc, n, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return BTF_OK;
}

static TauSide tau_side_of(btf_ctx* c, uint64_t seed, double lam2, double stability) {
  TauSide t{};
  ++c->prior_version;                                    // (whoever launches this side task redraws Tau2)
  t.V = c->V; t.T = c->T; t.nD = c->nD; t.M = c->M;
  t.dr_ptr = c->dr_ptr; t.dr_col = c->dr_col; t.dr_val = c->dr_val;
  t.lam2 = lam2; t.lo = stability; t.hi = 1.0 / stability;
  t.Tau2 = c->Tau2; t.Ta = c->Ta; t.Tb = c->Tb; t.Tc = c->Tc; t.lsum = c->lsum;
  t.seed = (unsigned long long)seed; t.hyp = c->dev_scalars ? c->hyp : nullptr;
  return t;
}

int btf_queue_Tau2(btf_ctx* c, uint64_t seed, double stability) {
  if (!c || !(stability > 0.0)) return fail(c, BTF_EINVAL, "bad Tau2 update arguments");
  if (!c->dev_scalars) return fail(c, BTF_ESTATE, "btf_queue_Tau2 needs device-resident scalars");
  if (!c->have_V || !c->have_hyper || !c->have_chain) return fail(c, BTF_ESTATE, "set V, Tau2 and the horseshoe+ chain first");
  c->tau_pending = true; c->tau_seed = seed; c->tau_stability = stability;
  return BTF_OK;
}

int btf_resample_Tau2(btf_ctx* c, uint64_t seed, double lam2, double stability, double* lsum_out) {
  if (!c || !(stability > 0.0)) return fail(c, BTF_EINVAL, "bad Tau2 update arguments");
  if (c->dev_scalars) lam2 = 1.0;   // the kernel reads the device-resident value
  if (!(lam2 > 0.0)) return fail(c, BTF_EINVAL, "bad Tau2 update arguments");
  if (!c->have_V || !c->have_hyper || !c->have_chain) return fail(c, BTF_ESTATE, "set V, Tau2 and the horseshoe+ chain first");
  HIPCHK(c, hipSetDevice(c->dev));
  c->tau_pending = false;
  {
    Prof p(c, BTF_K_HYPER);
    p.launch(tau2_kernel, dim3(c->M), dim3(256), 0, tau_side_of(c, seed, lam2, stability), c->K);
  }
  HIPCHK(c, hipGetLastError());
  if (lsum_out) {   // through pinned memory: a pageable destination makes the small copy several times slower
    if (!c->pin_lsum) HIPCHK(c, hipHostMalloc((void**)&c->pin_lsum, (size_t)c->M * sizeof(double), hipHostMallocDefault));
    HIPCHK(c, hipMemcpyAsync(c->pin_lsum, c->lsum, (size_t)c->M * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(lsum_out, c->pin_lsum, (size_t)c->M * sizeof(double));
  }
  return BTF_OK;
}

int btf_set_nu2(btf_ctx* c, double nu2) {
  if (!c || !(nu2 > 0.0)) return fail(c, BTF_EINVAL, "nu2 must be positive");
  c->nu2 = nu2;
  return BTF_OK;
}
int btf_set_omega(btf_ctx* c, const double* omega_rows, const double* omega_cols) {
  if (c) { c->w_part_valid = false; c->w_local_done = c->v_local_done = false; c->sse_cols_valid = false; }   // the weights change
  if (!c || !omega_rows || !omega_cols) return BTF_EINVAL;
  if (!c->have_data || !c->binomial) return fail(c, BTF_ESTATE, "btf_set_omega needs binomial data");
  HIPCHK(c, hipSetDevice(c->dev));
  int rc;
  const int MT = c->M * c->T;
  HIPCHK(c, hipMemsetAsync(c->C_wT, 0, (size_t)MT * c->ldw * sizeof(double), c->stream));
  HIPCHK(c, hipMemsetAsync(c->C_v, 0, (size_t)c->N * c->ldv * sizeof(double), c->stream));
  if ((rc = upload_relayout(c, omega_rows, slab_rows(c), MT, c->C_wT, c->ldw, true))) return rc;
  if ((rc = upload_relayout(c, omega_cols, c->N, slab_cols(c) * c->T, c->C_v, c->ldv, false))) return rc;
  // cells without an observation (trials stored as 0) carry no weight
  hipLaunchKernelGGL(mask_kernel, dim3(1024), dim3(256), 0, c->stream, c->C_wT, c->B_wT, (size_t)MT * c->ldw);
  hipLaunchKernelGGL(mask_kernel, dim3(1024), dim3(256), 0, c->stream, c->C_v, c->B_v, (size_t)c->N * c->ldv);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return BTF_OK;
}
int btf_get_omega(btf_ctx* c, double* omega_rows) {
  if (!c || !omega_rows) return BTF_EINVAL;
  if (!c->have_data || !c->C_wT) return fail(c, BTF_ESTATE, "no weights on this context");
  HIPCHK(c, hipSetDevice(c->dev));
  const int MT = c->M * c->T;
  std::vector<double> h((size_t)MT * c->ldw);
  HIPCHK(c, hipMemcpyAsync(h.data(), c->C_wT, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < c->nl; ++i)
    for (int jt = 0; jt < MT; ++jt) omega_rows[(size_t)i * MT + jt] = h[(size_t)jt * c->ldw + i];
  return BTF_OK;
}

// ---------------------------------------------------------------------------- W
namespace {
// phase 1 of the W half-sweep: Gram / outer products of V and the streaming accumulation into c->part.
// It depends on the data and on V only, so it can be queued ahead of the hyper-parameter draws
// (btf_w_accum) and its partials also give the residual sum of squares (btf_draw_scalars, which & 4).
// Split of an accumulation over the rank's own block [lo, hi) of the reduction axis: the REST launch covers the other
// rows in chunks of rpb (slots 0 .. nch_r - 1; needs lo % rpb == 0 so that no chunk straddles the hole), the LOCAL launch
// the own rows in chunks of rpb_l - sized to spread that eighth of the work over the whole chip, not over an eighth of
// the workgroups - with slots nch_r .. nch_r + nch_l - 1.
struct SplitGeom { bool ok; int lo, hi, rpb_l, nch_l, nch_r; };
SplitGeom split_geom(int lo, int hi, int Rdim, int rpb, int tiles) {
  SplitGeom g{false, lo, hi, 0, 0, 0};
  if (hi <= lo || hi - lo >= Rdim || rpb <= 0 || lo % rpb != 0) return g;
  g.ok = true;
  g.nch_r = (Rdim - (hi - lo) + rpb - 1) / rpb;
  long long rl = ((long long)(hi - lo) * tiles + 255) / 256;
  g.rpb_l = std::min(rpb, std::max(64, round_up((int)rl, 64)));
  g.nch_l = (hi - lo + g.rpb_l - 1) / g.rpb_l;
  return g;
}
bool split_applies(const btf_ctx* c) {
  return c->split_accum && !c->binomial && !c->counts && !(c->nl == c->N && c->ml == c->M) && c->nl > 0 && c->ml > 0;
}
enum { ACC_ALL = 2, ACC_LOCAL = 0 };

// ---- the two-launch W+V step (btf_fused.h) ----------------------------------------------------------------------
// words: one 128-byte line per flag / counter, then the tiles' tickets of the W launch, then those of the V launch
enum { FZ_EIG = 0, FZ_SC = 32, FZ_LAM = 64, FZ_GRAM = 96, FZ_GSUM = 128, FZ_TICKETS = 160 };
constexpr int FZ_PUB_EIG = 0, FZ_PUB_HYP = 128, FZ_PUB_GSUM = 136, FZ_PUB_GRAN = 200, FZ_PUB_DOUBLES = 224;
int ensure_fused(btf_ctx* c) {
  const int tw = c->ldw / ACC_TILE, tv = c->ldv / ACC_TILE;
  if (c->fz_words && c->fz_tiles_w == tw && c->fz_tiles_v == tv) return BTF_OK;
  int rc;
  const size_t nwords = (size_t)((FZ_TICKETS + tw + tv + 31) / 32) * 32;
  if ((rc = dev_alloc(c, &c->fz_words, nwords))) return rc;
  HIPCHK(c, hipMemsetAsync(c->fz_words, 0, nwords * sizeof(unsigned), c->stream));      // once: the last arrivers reset their tickets
  if (!c->fz_pub) {
    if ((rc = dev_alloc(c, &c->fz_pub, (size_t)FZ_PUB_DOUBLES))) return rc;
    HIPCHK(c, hipMemsetAsync(c->fz_pub, 0, (size_t)FZ_PUB_DOUBLES * sizeof(double), c->stream));      // (the tagged granules: no stale tag may equal an epoch)
  }
  c->fz_tiles_w = tw; c->fz_tiles_v = tv;
  c->fz_gram_total = 0; c->fz_w_total = 0;
  return BTF_OK;
}
// the W launch can carry w_solve's work as its tail: complete-data stream on the 16-wave instances (two rows in flight
// for every nembeds, three - long row ranges - for nembeds 8), no curve-count corrections, no split accumulation
bool fuse_w_applies(const btf_ctx* c, bool wt, bool cv, int mode, int rpb) {
  if (c->fused_step < 2 || wt || cv || mode != 0 || c->nl <= 0 || c->K > 8) return false;
  if (split_applies(c)) return false;
  if (rpb >= unr3_min_rpb() && c->K != 8) return false;
  return true;
}
struct WFuseReq { const double* dz; uint64_t seed; bool done; };
// part: ACC_ALL - whatever is still missing (every chunk, or the rest behind an ACC_LOCAL launch); ACC_LOCAL - only the
// chunks of this rank's own columns of V, no side tasks (queued right behind the V draw, before the all-gather of V)
int w_accum_phase(btf_ctx* c, int compat, int part = ACC_ALL, WFuseReq* wf = nullptr) {
  const int K = c->K, KK = c->KK, MT = c->M * c->T;
  const bool wt = lik_weighted(c), cv = c->weighted && !wt;
  const int mode = !wt ? 0 : (compat == BTF_COMPAT_REFERENCE && c->stale_w && c->srcmap_w ? 2 : 1);
  const int NV = wt ? K + KK : K;
  const int tiles = c->ldw / ACC_TILE;
  const int rpb = pick_rpb(MT, tiles, c->rpb_w, wt, acc_slots(c, K, mode, true), w_side_reserve(c, wt));
  const int nch = (MT + rpb - 1) / rpb;
  int rc;
  // (the fused tail fetches the chunk sums in whole batches of up to 32 slots: room for the last batch - btf_fused.h)
  if ((rc = ensure_part(c, (size_t)round_up(nch, 32) * NV * c->ldw))) return rc;
  const SplitGeom sg = split_applies(c) ? split_geom(c->col0 * c->T, (c->col0 + c->ml) * c->T, MT, rpb, tiles) : SplitGeom{};
  if (sg.ok) { if ((rc = ensure_part(c, (size_t)(sg.nch_r + sg.nch_l) * NV * c->ldw))) return rc; }
  if (part == ACC_LOCAL) {
    c->w_local_done = false;
    if (!sg.ok || mode == 2) return BTF_OK;               // nothing queued ahead: the next call accumulates everything
    K_SWITCH(K, launch_accum<KT>(c, BTF_K_W_ACCUM, mode, c->A_wT, c->C_wT, c->C8_wT, c->V, c->srcmap_w, MT, c->ldw, sg.rpb_l, sg.nch_l,
                                 EigSide{nullptr, 0, 0, nullptr}, EigSideCols{nullptr, 0, CurveLists{nullptr, nullptr, nullptr}, nullptr, 0.0, nullptr},
                                 TauSide{}, GramSide{}, ChunkMap{sg.nch_r, sg.lo, INT_MAX, 0, sg.hi, 0}));
    HIPCHK(c, hipGetLastError());
    c->w_local_done = true; c->w_local_rpb = rpb; c->w_local_mode = mode;
    return BTF_OK;
  }
  const bool rest_only = c->w_local_done && sg.ok && c->w_local_rpb == rpb && c->w_local_mode == mode;
  c->w_local_done = false;
  const ChunkMap cm = rest_only ? ChunkMap{0, 0, sg.lo, sg.hi - sg.lo, MT, 0} : ChunkMap{0, 0, INT_MAX, 0, MT, 0};
  const int nch_launch = rest_only ? sg.nch_r : nch;
  const int nch_total = rest_only ? sg.nch_r + sg.nch_l : nch;
  const bool whole = c->nl == c->N && c->ml == c->M;      // fused Grams cover all rows/columns only when unsharded
  bool use_gv = !wt && whole && c->fuse_gram && c->ngp_v > 0;
  if (cv && c->nl > 0) {                      // the per-column Grams V_j'V_j: from the sampler that drew V, or computed here
    if (c->ngp_v != c->ml) {
      if (!c->gpart_v) { if ((rc = dev_alloc(c, &c->gpart_v, (size_t)c->M * KK))) return rc; }
      Prof p(c, BTF_K_GRAM);
      K_SWITCH(K, p.launch(colgram_kernel<KT>, dim3(c->ml), dim3(WAVE), 0, (const double*)c->V, c->T, c->ml, c->gpart_v));
      c->ngp_v = c->ml;
    }
    use_gv = true;
  }
  if (c->nl > 0) {
    // V'V when no sampler left its per-column blocks (sharded runs, long V): partial Grams by side workgroups of the
    // accumulation launch itself - the solve that consumes them is the next kernel (no gram_kernel launch)
    GramSide gram{nullptr, 0, nullptr, 0};
    if (!wt && !use_gv) {
      c->ngp_gram = std::min(gram_blocks(MT), 32);
      gram = GramSide{c->V, MT, c->gpart, c->ngp_gram, nullptr, nullptr, 0, nullptr, 0};
    }
    // the per-column blocks V_j'V_j of the sampler that drew V (ngp_v = ml of them): summed ONCE, by a side workgroup of this
    // launch, instead of by every workgroup of the W solve behind it (same order, same bits: GramSide.sum_*)
    bool gsum = false;
    c->w_part_gsum = false;
    if (!wt && use_gv && part == ACC_ALL && c->ngp_v > 8 * std::min(32, (WS_ROWS * ws_split_of(K, false)) / KK)) {
      if (!c->gsum_v) { if ((rc = dev_alloc(c, &c->gsum_v, (size_t)tri(MAX_K)))) return rc; }
      gram.sum_src = c->gpart_v; gram.sum_n = c->ngp_v; gram.sum_out = c->gsum_v; gram.sum_threads = WS_ROWS * ws_split_of(K, false);
      gsum = true;
    }
    c->w_part_gsum = gsum;
    TauSide tau{};
    if (c->tau_pending && c->dev_scalars && c->have_chain) tau = tau_side_of(c, c->tau_seed, 1.0, c->tau_stability);
    SweepSide sw{};
    if (c->sc_pending) {                      // queued nu2 / sigma2 draw (btf_queue_scalars checked that it can ride here)
      const int h = std::min(c->K, c->N);
      const double nfree = (double)c->N * c->K - (double)h * (h - 1) / 2.0 - (double)(c->K - h) * c->N;   // factor.py:155-174
      sw.sc = ScalarSide{c->sse_cols, c->M, c->ssw + c->sa2, c->nobs_global >= 0.0 ? c->nobs_global : c->nobs, c->W, c->N * c->K, nfree,
                         c->sc_prior[0], c->sc_prior[1], c->sc_prior[2], c->sc_prior[3], c->sc_which, c->sc_seed, c->hyp};
      c->sc_pending = false;
    }
    FuseW fw{};
    // (the owner workgroups stream nothing: only when owners, side and streaming workgroups fit the chip in one round.
    //  Rows per owner: as few as the chip has room for - the hand-off of a tile's chunk sums is bandwidth-bound per
    //  reading workgroup - but whole virtual w_solve workgroups, and the fetched sums within the instance's LDS)
    int own_rows = 0;
    if (wf && !rest_only && fuse_w_applies(c, wt, cv, mode, rpb)) {
      const long long others = (long long)tiles * nch + (tau.Tau2 ? (tau.M + TAU_SIDE_CPW - 1) / TAU_SIDE_CPW : 0) + (gram.gpart ? gram.nblocks : 0) + (sw.sc.hyp ? 1 : 0);
      static const int min_rows = [] { const char* e = std::getenv("BTF_OWNER_ROWS"); return e ? std::atoi(e) : 32; }();     // (A/B aid)
      for (int r = std::max(ws_rows_for(c->nl), min_rows); r <= ACC_TILE; r *= 2) {
        if (others + (long long)tiles * (ACC_TILE / r) <= (long long)c->ncu && w_owner_lds_doubles(K, r, nch) <= w_owner_lds_budget(K)) { own_rows = r; break; }
      }
    }
    const bool fuse = own_rows > 0;
    if (fuse) {
      // w_solve's work as the tail of this launch (btf_fused.h): its arguments, the tiles' tickets, and - when scalar
      // draws or the Gram of V are made by side workgroups of this same launch - their flags
      if ((rc = ensure_fused(c))) return rc;
      ++c->fz_epoch;
      WSolveArgs& a = fw.a;
      a.part = c->part; a.nch = nch_total; a.ld = c->ldw; a.weighted = 0;
      a.gpart = use_gv ? c->gpart_v : c->gpart; a.ngp = use_gv ? c->ngp_v : c->ngp_gram;
      if (whole && c->fuse_gram) {
        if (!c->gpart_w) { if ((rc = dev_alloc(c, &c->gpart_w, (size_t)std::max(64, (c->N + WS_ROWS - 1) / WS_ROWS) * KK))) return rc; }
        a.gout = c->gpart_w;
      }
      a.s = c->binomial ? 1.0 : 1.0 / c->nu2;
      a.sR = a.s * c->R;
      a.inv_sigma2 = 1.0 / c->sigma2;
      a.hyp = c->dev_scalars ? c->hyp : nullptr; a.Rrep = c->R; a.hyp_noise = c->binomial ? 0 : 1;
      a.W = c->W; a.row0 = c->row0; a.nl = c->nl;
#ifdef BTF_WS_STAMPS
      a.dbg = c->dbg;
#endif
      a.z = wf->dz; a.seed = wf->seed; a.stream = 2 * c->sweep_w + 0x10000ULL;
      a.status = c->status;
      fw.cnt = c->fz_words + FZ_TICKETS;
      fw.rw = ws_rows_for(c->nl);
      if (sw.sc.hyp) {
        sw.sc.pub = c->fz_pub + FZ_PUB_HYP; sw.sc.flag = c->fz_words + FZ_SC; sw.sc.epoch = c->fz_epoch;
        fw.hp = HypPub{c->fz_pub + FZ_PUB_HYP, c->fz_words + FZ_SC, c->fz_epoch, sw.sc.which & 3};
      }
      if (gram.gpart) {                        // V'V by Gram side workgroups of this launch (sharded runs): the owners wait for them
        gram.cnt = c->fz_words + FZ_GRAM;
        c->fz_gram_total += (unsigned)gram.nblocks;
        fw.gram_cnt = gram.cnt; fw.gram_expected = c->fz_gram_total;
      }
      c->fz_w_total += (unsigned)nch_total;
      fw.expected = c->fz_w_total;
      fw.rows = own_rows;
      fw.owners = tiles * (ACC_TILE / own_rows);
    }
    K_SWITCH(K, launch_accum<KT>(c, BTF_K_W_ACCUM, mode, c->A_wT, c->C_wT, c->C8_wT, c->V, c->srcmap_w, MT, c->ldw, rpb, nch_launch,
                                 EigSide{nullptr, 0, 0, nullptr}, EigSideCols{nullptr, 0, CurveLists{nullptr, nullptr, nullptr}, nullptr, 0.0, nullptr}, tau, gram, cm, sw,
                                 fuse ? &fw : nullptr));
    if (fuse) {
      wf->done = true;
      const int wrows = ws_rows_for(c->nl);
      c->ngp_w = fw.a.gout ? (c->nl + wrows - 1) / wrows : 0;
      c->ngp_v = 0;
    }
    c->tau_pending = false;
  } else if (c->tau_pending && c->dev_scalars && c->have_chain) {
    // a rank without rows (ceil chunks: N = 10 over 8 ranks leaves ranks 5-7 empty) has no accumulation launch to
    // carry the queued Tau2 chain: draw it on its own, so that this rank's Tau2 / lsum match the other ranks'
    Prof p(c, BTF_K_HYPER);
    p.launch(tau2_kernel, dim3(c->M), dim3(256), 0, tau_side_of(c, c->tau_seed, 1.0, c->tau_stability), c->K);
    c->tau_pending = false;
  }
  HIPCHK(c, hipGetLastError());
  c->w_part_valid = true; c->w_part_mode = mode; c->w_part_nch = nch_total; c->w_part_rpb = rpb; c->w_part_gv = use_gv;
  c->w_part_curve = cv;
  return BTF_OK;
}
}  // namespace

namespace {
int v_accum_local(btf_ctx* c, int compat);
int mark_draw(btf_ctx* c);
}  // namespace

int btf_w_accum(btf_ctx* c, int compat) {
  if (!c) return BTF_EINVAL;
  if (!c->have_data || !c->have_V || !c->have_W) return fail(c, BTF_ESTATE, "set data, W and V first");
  HIPCHK(c, hipSetDevice(c->dev));
  return w_accum_phase(c, compat);
}

int btf_resample_W(btf_ctx* c, const double* z, uint64_t seed, int compat) {
  if (!c) return BTF_EINVAL;
  if (!c->have_data || !c->have_V || !c->have_W) return fail(c, BTF_ESTATE, "set data, W and V first");
  HIPCHK(c, hipSetDevice(c->dev));
  const int K = c->K, KK = c->KK;
  const bool wt = lik_weighted(c), cv = c->weighted && !wt;
  const int want_mode = !wt ? 0 : (compat == BTF_COMPAT_REFERENCE && c->stale_w && c->srcmap_w ? 2 : 1);
  int rc;
  c->v_local_done = false;                                 // W is about to change
  const double* dz = nullptr;
  if (z) {
    const size_t nz = (size_t)w_z_offset(c->N, K);
    if ((rc = ensure_z(c, nz))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->zbuf, z, nz * sizeof(double), hipMemcpyHostToDevice, c->stream));
    dz = c->zbuf;
  }
  // (no current partials: the accumulation is launched here - with the solve as its tail where that applies, btf_fused.h)
  WFuseReq wf{dz, seed, false};
  if (!(c->w_part_valid && c->w_part_mode == want_mode && c->w_part_curve == cv)) { if ((rc = w_accum_phase(c, compat, ACC_ALL, &wf))) return rc; }
  if (c->sc_pending) return fail(c, BTF_ESTATE, "btf_queue_scalars must be followed by the W accumulation that carries it");
  const int nch = c->w_part_nch;
  const bool use_gv = c->w_part_gv;
  c->w_part_valid = false;                                 // consumed: W changes below
  if (c->nl > 0 && !wf.done) {
    const bool whole = c->nl == c->N && c->ml == c->M;
    WSolveArgs a{};
    a.part = c->part; a.nch = nch; a.ld = c->ldw; a.weighted = wt ? 1 : 0;
    a.gsrc = (wt && want_mode == 2) ? c->srcmap_w : nullptr;     // (stale cached weights: Gram sums at the source row's column)
    a.gpart = use_gv ? c->gpart_v : c->gpart; a.ngp = use_gv ? c->ngp_v : c->ngp_gram;
    if (use_gv && c->w_part_gsum) { a.gpart = c->gsum_v; a.ngp = 1; }      // (summed by a side workgroup of the accumulation launch)
    const int wrows = ws_rows_for(c->nl);
    const int wblocks = (c->nl + wrows - 1) / wrows;
    if (whole && c->fuse_gram) {
      if (!c->gpart_w) { if ((rc = dev_alloc(c, &c->gpart_w, (size_t)std::max(64, (c->N + WS_ROWS - 1) / WS_ROWS) * KK))) return rc; }
      a.gout = c->gpart_w;
    }
    a.s = c->binomial ? 1.0 : 1.0 / c->nu2;
    a.sR = a.s * c->R;
    a.inv_sigma2 = 1.0 / c->sigma2;
    a.hyp = c->dev_scalars ? c->hyp : nullptr; a.Rrep = c->R; a.hyp_noise = c->binomial ? 0 : 1;
    a.W = c->W; a.row0 = c->row0; a.nl = c->nl;
#ifdef BTF_WS_STAMPS
    a.dbg = c->dbg;
#endif
    a.z = dz; a.seed = seed; a.stream = 2 * c->sweep_w + 0x10000ULL;
    a.status = c->status;
    if (cv) { a.cv = CurveLists{c->cv_rptr, c->cv_rcol, c->cv_rdef}; a.cv_blocks = c->gpart_v; }
    // a queued lam2 | rest draw (full sweeps) rides here, as one more workgroup: the Tau2 chain of this sweep - whose
    // column sums it needs - ran in the accumulation launch in front, and a V half-sweep that finds lam2 already drawn
    // uses the precomputed prior band and the dataflow tail (the draw used to be a side workgroup of the V launch itself,
    // whose tails then had to form the band from Tau2: 25.4 us per V launch against 19 + 2 for the band's launch)
    if (c->lam_in_wsolve && c->lam_pending && c->dev_scalars && c->lsum && c->have_chain) {
      a.lam = LamSide{c->lsum, c->M, (double)c->nD * c->M * c->K + 1.0, c->lam_exact, c->lam_seed, c->hyp, nullptr, nullptr, 0u};
      c->lam_pending = false;
      ++c->prior_version;
      // ... and, where the last V half-sweep loaded the precomputed prior band, that band - of the Tau2 the accumulation
      // launch in front has just drawn and the lam2 the workgroup above draws - by one more workgroup per column (BandSide:
      // they wait for the lam2 workgroup's flag), instead of a launch of its own in front of the V launch
      const int TD1 = c->T * (c->TF + 2);
      if (c->band_in_wsolve && c->v_wants_band && c->pband && (!c->band_img || c->pimg) && TD1 <= 2 * WS_ROWS * ws_split_of(K, wt) &&
          c->nD <= 2 * WS_ROWS * ws_split_of(K, wt)) {
        if ((rc = ensure_fused(c))) return rc;
        ++c->fz_epoch;
        a.lam.pub = c->fz_pub + FZ_PUB_HYP; a.lam.flag = c->fz_words + FZ_LAM; a.lam.epoch = c->fz_epoch;
        a.band = BandSide{c->Tau2, c->nD, c->st_ptr, c->st_row, c->st_coef, TD1, c->col0, c->ml, c->pband,
                          c->band_img ? c->pimg : nullptr, c->T, c->TF + 2, c->band_PB, nullptr, 0.0,
                          c->fz_pub + FZ_PUB_HYP, c->fz_words + FZ_LAM, c->fz_epoch, c->status};
        c->pband_version = c->prior_version;
        c->pimg_version = c->band_img ? c->prior_version : 0;
      }
    }
    K_SWITCH(K, launch_wsolve<KT>(c, a));
    c->ngp_w = a.gout ? wblocks : 0;
    c->ngp_v = 0;   // V'V partials are consumed once; any other W/V change must recompute
  }
  c->sweep_w++;
  c->nb_L_valid = false;
  c->sse_cols_valid = false;                               // W has changed: the residual parts are those of the old W
  HIPCHK(c, hipGetLastError());
  if (split_applies(c)) {        // behind the W draw: mark it, then queue the own-rows chunks of the next V accumulation
    if ((rc = mark_draw(c))) return rc;
    return v_accum_local(c, compat);
  }
  return BTF_OK;
}

// ---------------------------------------------------------------------------- V
// The banded samplers of the V half-sweep (twisted / single chain / any-size), writing the draw of every local
// column to `out`.  prior_only: the likelihood part is switched off (no partials, no Gram) - a draw from the
// prior N(0, (I_K (x) Delta' Lambda_j Delta)^-1) in the same declared order (elliptical slice sampling, btf_ess_*).
static int v_banded_dispatch(btf_ctx* c, int choice, const double* dz, uint64_t seed, int nch, bool use_gw, double eps0,
                             int attempts, double* out, bool prior_only) {
  const int K = c->K, KK = c->KK, T = c->T, n = T * K;
  const bool wt = lik_weighted(c), cv = c->weighted && !wt && !prior_only;
  const bool whole = c->nl == c->N && c->ml == c->M;
  int rc;
  const int bw = (c->TF + 1) * K, R1 = bw + 1, D1 = c->TF + 2;
  size_t lds_fixed = (size_t)(3 * n + T * D1 + (wt ? T * KK : KK) + (bw * (bw + 1) / 2 + 3) / 4) * sizeof(double);
  size_t lds_band = std::max((size_t)n * R1, (size_t)(GRAM_BLOCKS + 16) * KK) * sizeof(double);   // (also the Gram staging area)
  size_t lds_bytes = lds_fixed + lds_band;
  VBandArgs a{};
  if (lds_bytes > 150 * 1024) {  // band (and the weighted case's per-depth likelihood blocks) in HBM scratch, only vectors on chip
    if (wt) lds_fixed -= (size_t)T * KK * sizeof(double);
    if (lds_fixed > 150 * 1024) return fail(c, BTF_EINVAL, "ndepth*nembeds too large for the on-chip vectors");
    const size_t stride = (size_t)n * R1 + (wt ? (size_t)T * KK : 0);
    if (!c->gband || c->gband_stride != stride) {
      if ((rc = dev_alloc(c, &c->gband, (size_t)c->ml * stride))) return rc;
      c->gband_stride = stride;
    }
    a.gband = c->gband; a.gband_stride = c->gband_stride;
    lds_bytes = lds_fixed;
  }
  a.part = c->part; a.nch = nch; a.ld = c->ldv; a.weighted = wt ? 1 : 0;
  a.gsrc = (wt && c->v_part_mode == 2) ? c->srcmap_v : nullptr;   // (stale cached weights: Gram blocks at the source column)
  a.gpart = use_gw ? c->gpart_w : c->gpart; a.ngp = use_gw ? c->ngp_w : c->ngp_gram;
  a.s = c->binomial ? 1.0 : 1.0 / c->nu2;
  a.sR = a.s * c->R;
  a.Tau2 = c->Tau2; a.lam2 = c->lam2; a.nD = c->nD;
  a.hyp = c->dev_scalars ? c->hyp : nullptr; a.Rrep = c->R; a.hyp_noise = c->binomial ? 0 : 1;
  a.panel4 = c->sampler == BTF_SAMPLER_BANDED_NOPANEL ? 0 : 1;
  a.st_ptr = c->st_ptr; a.st_row = c->st_row; a.st_coef = c->st_coef;
  a.T = T; a.TF = c->TF; a.col0 = c->col0; a.ml = c->ml;
  a.V = c->V; a.z = dz; a.seed = seed; a.stream = 2 * c->sweep_v + 0x10001ULL;
  a.eps0 = eps0; a.attempts = attempts; a.status = c->status; a.tries = c->tries; a.dbg = c->dbg;
  if (prior_only) { a.nch = 0; a.ngp = 0; a.s = 0.0; a.sR = 0.0; a.hyp_noise = 0; }
  if (cv) { a.cv = CurveLists{c->cv_cptr, c->cv_crow, c->cv_cdef}; a.cv_W = c->W; }
  a.V = out;
  hipError_t e = hipSuccess;
  bool handled = false;
  const bool fast = choice >= 0;
  // the twisted kernel builds the prior band itself from the fixed-slot stencil (as the spectral one does)
  const int tw_npl = std::max(1, ((bw - 1) * (bw - 2) / 2 + WAVE - 1) / WAVE);      // (dispatch_vbanded_twist's own test)
  const bool tw_handles = bw <= 15 ? tw_npl <= 2 : (tw_npl >= 2 && tw_npl <= 8);
  const bool own_prior = choice == 2 && tw_handles && c->st_dense_ok && c->st_drow && (c->TF + 2) * T >= c->nD;
  if (own_prior) {
    a.st_drow = c->st_drow; a.st_dcoef = c->st_dcoef; a.pband = nullptr;
    // ... unless Tau2 / lam2 have stood still since the last V half-sweep (loops of bare W+V steps, host-driven chains with
    // fixed hyper-parameters): then the band of every column is built ONCE, in the kernel's own arithmetic
    // (prior_band_kernel's spectral form: same bits), and the sampler workgroups load it instead of their stencil - 49 KB
    // less per workgroup in the cold batch of loads the kernel starts with.  Full sweeps redraw Tau2 every time: nothing changes.
    const int TD1 = T * D1;
    static const bool lazy_band = [] { const char* e = std::getenv("BTF_TWIST_PBAND"); return !e || std::atoi(e) != 0; }();      // (A/B aid)
    if (!lazy_band) { /* the kernel forms the band itself, every time */ }
    else if (c->pband && c->pband_version == c->prior_version) a.pband = c->pband;
    else if (c->last_v_prior_version == c->prior_version) {
      if (!c->pband) { if ((rc = dev_alloc(c, &c->pband, (size_t)c->ml * TD1))) return rc; }
      Prof p(c, BTF_K_PRIOR);
      p.launch(prior_band_kernel, dim3((c->ml * TD1 + 255) / 256), dim3(256), 0, (const double*)c->Tau2, c->lam2, c->nD,
               (const int*)c->st_ptr, (const int*)c->st_row, (const double*)c->st_coef, TD1, c->col0, c->ml, c->pband,
               (const double*)(c->dev_scalars ? c->hyp : nullptr), 1, (double*)nullptr, 0, 0, 0);
      c->pband_version = c->prior_version;
      a.pband = c->pband;
    }
    c->last_v_prior_version = c->prior_version;
  }
  if (fast && !own_prior) {
    const int TD1 = T * D1;
    if (!c->pband) { if ((rc = dev_alloc(c, &c->pband, (size_t)c->ml * TD1))) return rc; }
    {   // rebuilt on every call, as the reference rebuilds Q_prior per column (factor.py:404-405)
      Prof p(c, BTF_K_PRIOR);
      p.launch(prior_band_kernel, dim3((c->ml * TD1 + 255) / 256), dim3(256), 0, (const double*)c->Tau2, c->lam2, c->nD,
               (const int*)c->st_ptr, (const int*)c->st_row, (const double*)c->st_coef, TD1, c->col0, c->ml, c->pband,
               (const double*)(c->dev_scalars ? c->hyp : nullptr), 0, (double*)nullptr, 0, 0, 0);
    }
    c->pband_version = 0;                 // (this form of the band is not the one the version stands for)
    a.pband = c->pband;
  }
  // V'V partials for the next W half-sweep (one KK block per column; bounded: every w_solve workgroup sums all of them)
  const bool emit_gv = !prior_only && fast && whole && c->fuse_gram && !wt &&
                       (size_t)c->ml * KK + 16 * KK <= ws_gram_stage(K, false);
  if (emit_gv) {
    if (!c->gpart_v) { if ((rc = dev_alloc(c, &c->gpart_v, (size_t)c->M * KK))) return rc; }
    a.gout = c->gpart_v;
  }
  if (!prior_only) { c->ngp_v = emit_gv ? c->ml : 0; c->ngp_w = 0; }
  if (choice == 2) {
    if ((rc = make_fill_table(c, wt))) return rc;
    a.fill = c->fill_tab; a.nfill = c->fill_n;
    a.ql_global = twist_wmode(c, wt) == 2 ? 1 : 0;
    e = dispatch_vbanded_twist(c, a, bw, tw_lds_bytes(T, K, c->TF, twist_wmode(c, wt)), &handled);
  }
  HIPCHK(c, e);
  if (fast && !handled && vb_fast_lds_bytes(T, K, c->TF, wt ? 1 : 0) <= 158 * 1024)
    e = dispatch_vbanded_fast(c, a, bw, vb_fast_lds_bytes(T, K, c->TF, wt ? 1 : 0), &handled);
  HIPCHK(c, e);
  if (fast && !handled) {
    const int CH = vc_chunk_for(c, wt);
    if (CH > 0) {
      const size_t stride = vc_scratch_stride(T, K, c->TF);
      if (!c->vc_scratch || c->vc_scratch_elems < (size_t)c->ml * stride) {
        if ((rc = dev_alloc(c, &c->vc_scratch, (size_t)c->ml * stride))) return rc;
        c->vc_scratch_elems = (size_t)c->ml * stride;
      }
      VBandArgs ac = a;
      ac.gband = c->vc_scratch; ac.gband_stride = stride;     // the factor record of every column (btf_banded_chunk.h)
      e = dispatch_vbanded_chunk(c, ac, bw, CH, wt, &handled);
      HIPCHK(c, e);
    }
  }
  if (!handled) { a.gout = nullptr; if (!prior_only) c->ngp_v = 0; }
  if (!handled) { K_SWITCH(K, e = launch_vbanded<KT>(c, a, lds_bytes)); }
  HIPCHK(c, e);
  return BTF_OK;
}

namespace {
// the own-rows chunks of the next V accumulation, queued right behind the W solve (before the all-gather of W)
int v_accum_local(btf_ctx* c, int compat) {
  c->v_local_done = false;
  if (!split_applies(c)) return BTF_OK;
  const int K = c->K, KK = c->KK;
  const bool wt = lik_weighted(c);
  const int mode = !wt ? 0 : (compat == BTF_COMPAT_REFERENCE && c->stale_v && c->srcmap_v ? 2 : 1);
  if (mode == 2) return BTF_OK;
  const int NV = wt ? K + KK : K;
  const int tiles = c->ldv / ACC_TILE;
  const int rpb = pick_rpb(c->N, tiles, c->rpb_v, wt, acc_slots(c, K, mode), v_side_reserve(c, wt));
  const int nch = (c->N + rpb - 1) / rpb;
  const SplitGeom sg = split_geom(c->row0, c->row0 + c->nl, c->N, rpb, tiles);
  if (!sg.ok) return BTF_OK;
  int rc;
  if ((rc = ensure_part(c, (size_t)std::max(nch, sg.nch_r + sg.nch_l) * NV * c->ldv))) return rc;
  K_SWITCH(K, launch_accum<KT>(c, BTF_K_V_ACCUM, mode, c->A_v, c->C_v, c->C8_v, c->W, c->srcmap_v, c->N, c->ldv, sg.rpb_l, sg.nch_l,
                               EigSide{nullptr, 0, 0, nullptr}, EigSideCols{nullptr, 0, CurveLists{nullptr, nullptr, nullptr}, nullptr, 0.0, nullptr},
                               TauSide{}, GramSide{}, ChunkMap{sg.nch_r, sg.lo, INT_MAX, 0, sg.hi, 0}));
  HIPCHK(c, hipGetLastError());
  c->v_local_done = true; c->v_local_rpb = rpb; c->v_local_mode = mode;
  return BTF_OK;
}
// the event the comm stream waits for: right behind the kernel that drew this rank's block
int mark_draw(btf_ctx* c) {
  if (!c->ev_draw) HIPCHK(c, hipEventCreateWithFlags(&c->ev_draw, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->ev_draw, c->stream));
  return BTF_OK;
}
// behind the V draw: mark it, then queue the own-columns chunks of the next W accumulation
int after_v_draw(btf_ctx* c, int compat) {
  if (!split_applies(c)) return BTF_OK;
  int rc;
  if ((rc = mark_draw(c))) return rc;
  return w_accum_phase(c, compat, ACC_LOCAL);
}
}  // namespace

int btf_resample_V(btf_ctx* c, const double* z, uint64_t seed, int compat, double eps0, int attempts) {
  if (!c) return BTF_EINVAL;
  if (!c->have_data || !c->have_V || !c->have_W || !c->have_hyper) return fail(c, BTF_ESTATE, "set data, W, V and hyper-parameters first");
  if (attempts < 0) attempts = 0;
  HIPCHK(c, hipSetDevice(c->dev));
  c->w_local_done = false;                                 // V is about to change
  c->sse_cols_valid = false;
  struct ResetFlag { bool& f; ~ResetFlag() { f = false; } } reset_nu2_flag{c->nu2_drawn_since_v};
  const int K = c->K, KK = c->KK, T = c->T, n = T * K;
  const bool wt = lik_weighted(c), cv = c->weighted && !wt;
  const int mode = !wt ? 0 : (compat == BTF_COMPAT_REFERENCE && c->stale_v && c->srcmap_v ? 2 : 1);
  c->v_part_mode = mode;
  const int NV = wt ? K + KK : K;
  const int tiles = c->ldv / ACC_TILE;
  const int rpb = pick_rpb(c->N, tiles, c->rpb_v, wt, acc_slots(c, K, mode), v_side_reserve(c, wt));
  // (the own-rows chunks may already be in the partials, queued behind the W draw: BTF_OPT_SPLIT_ACCUM)
  const SplitGeom vsg = split_applies(c) ? split_geom(c->row0, c->row0 + c->nl, c->N, rpb, tiles) : SplitGeom{};
  const bool v_rest_only = c->v_local_done && vsg.ok && c->v_local_rpb == rpb && c->v_local_mode == mode;
  c->v_local_done = false;
  const int nch_all = (c->N + rpb - 1) / rpb;
  const int nch = v_rest_only ? vsg.nch_r + vsg.nch_l : nch_all;       // slots the sampler adds up
  int rc;
  if ((rc = ensure_part(c, (size_t)nch * NV * c->ldv))) return rc;
  const double* dz = nullptr;
  if (z) {
    const size_t nz = (size_t)c->M * n;
    if ((rc = ensure_z(c, nz))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->zbuf, z, nz * sizeof(double), hipMemcpyHostToDevice, c->stream));
    dz = c->zbuf;
  }
  if (c->ml > 0) {
    const bool whole = c->nl == c->N && c->ml == c->M;
    const bool use_gw = !wt && whole && c->fuse_gram && c->ngp_w > 0;
    const int choice = banded_choice(c);
    // W'W: partials of w_solve, or gram_kernel - except for the spectral sampler without partials (sharded runs), whose only
    // consumer of the Gram is the eigen side task of the accumulation launch: that workgroup then forms it from W itself
    const bool gram_aside = !wt && !use_gw && choice == 3 && c->N <= 16384;
    if (!wt && !use_gw && !gram_aside) { K_SWITCH(K, launch_gram<KT>(c, c->W, c->N)); }
    EigSide side{nullptr, 0, 0, nullptr};
    if (choice == 3) {
      // spectral sampler: the eigen-system of the Gram W'W rides along in the accumulation launch (btf_eig.h)
      if (!c->eig) {
        if ((rc = dev_alloc(c, &c->eig, (size_t)K + K * K + 8))) return rc;
        HIPCHK(c, hipMemsetAsync(c->eig, 0, ((size_t)K + K * K + 8) * sizeof(double), c->stream));   // no previous solution
      }
      side = EigSide{use_gw ? c->gpart_w : c->gpart, use_gw ? c->ngp_w : c->ngp_gram, K, c->eig, gram_aside ? c->W : nullptr, c->N};
    }
    // curve columns ride along as side tasks too, a wave each (their sampler workgroups can solve them as well:
    // eig_cols_ready = 0 - kept as the fallback path)
    EigSideCols sidec{nullptr, 0, CurveLists{nullptr, nullptr, nullptr}, nullptr, 0.0, nullptr};
    const bool cols_aside = choice == 3 && cv;
    if (cols_aside) sidec = EigSideCols{c->cv_dcols, c->cv_ndef, CurveLists{c->cv_cptr, c->cv_crow, c->cv_cdef}, c->W, 1.0 / c->R, c->eig_cols};
    // the spectral sampler as the tail of the accumulation launch (btf_fused.h): complete-data stream on a 16-wave
    // instance, tf_order 2 with the pivot records in LDS, whole columns per 128-output tile and their layouts within the
    // instance's LDS, no curve columns, no split accumulation
    const bool fuse_v = c->fused_step >= 1 && choice == 3 && !wt && !cv && mode == 0 && c->TF == 2 && K <= 8 && !v_rest_only && !split_applies(c) &&
                        vf_fits(T, K, c->TF, c->nD, 16, rpb >= unr3_min_rpb()) && vs_lds_bytes(T, K, c->TF, c->nD, false) <= 160 * 1024 &&
                        !(rpb >= unr3_min_rpb() && K != 8);
    const ChunkMap cm = v_rest_only ? ChunkMap{0, 0, vsg.lo, vsg.hi - vsg.lo, c->N, 0} : ChunkMap{0, 0, INT_MAX, 0, c->N, 0};
    SweepSide sw{};
    if (c->lam_pending && c->dev_scalars && c->lsum && c->have_chain) {     // a queued lam2 | rest draw no scalar launch took
      sw.lam = LamSide{c->lsum, c->M, (double)c->nD * c->M * c->K + 1.0, c->lam_exact, c->lam_seed, c->hyp};
      c->lam_pending = false;
      ++c->prior_version;
    }
    if (!fuse_v) {
      K_SWITCH(K, launch_accum<KT>(c, BTF_K_V_ACCUM, mode, c->A_v, c->C_v, c->C8_v, c->W, c->srcmap_v, c->N, c->ldv, rpb,
                                   v_rest_only ? vsg.nch_r : nch_all, side, sidec, TauSide{}, GramSide{}, cm, sw));
    }
    hipError_t e = hipSuccess;
    if (choice == 3) {
      // spectral sampler (complete data): K scalar banded systems per column in the eigen-basis of the Gram
      VSpecArgs sa{};
      sa.part = c->part; sa.nch = nch; sa.ld = c->ldv; sa.eig = c->eig;
      sa.s = c->binomial ? 1.0 : 1.0 / c->nu2; sa.sR = sa.s * c->R;
      sa.Tau2 = c->Tau2; sa.lam2 = c->lam2; sa.nD = c->nD;
      sa.st_ptr = c->st_ptr; sa.st_row = c->st_row; sa.st_coef = c->st_coef;
      sa.st_drow = c->st_drow; sa.st_dcoef = c->st_dcoef;
      sa.T = T; sa.TF = c->TF; sa.K = K; sa.col0 = c->col0; sa.ml = c->ml;
      sa.V = c->V; sa.z = dz; sa.seed = seed; sa.stream = 2 * c->sweep_v + 0x10001ULL;
      sa.eps0 = eps0; sa.attempts = attempts; sa.status = c->status; sa.tries = c->tries;
      sa.hyp = c->dev_scalars ? c->hyp : nullptr; sa.Rrep = c->R; sa.hyp_noise = c->binomial ? 0 : 1; sa.dbg = c->dbg;
      if (cv) {
        sa.cv = CurveLists{c->cv_cptr, c->cv_crow, c->cv_cdef}; sa.cv_W = c->W;
        sa.gpart = side.gpart; sa.ngp = side.ngp; sa.eig_cols = c->eig_cols;
        sa.eig_cols_ready = cols_aside ? 1 : 0;
      }
      c->sse_cols_valid = false;
      // (only inside full sweeps - a nu2 draw since the last V half-sweep: a loop of bare W + V steps does not pay for it)
      if (c->fused_sweep && whole && !cv && mode == 0 && !c->binomial && c->dev_scalars && c->nu2_drawn_since_v) {
        if (!c->sse_cols) { if ((rc = dev_alloc(c, &c->sse_cols, (size_t)c->M))) return rc; }
        sa.sse_out = c->sse_cols;
        c->sse_cols_valid = true;             // (as of the end of this launch: W as it stands, V as drawn here)
      }
      const bool emit = whole && c->fuse_gram &&
                        (size_t)c->ml * KK + 16 * KK <= ws_gram_stage(K, false);
      if (emit) {
        if (!c->gpart_v) { if ((rc = dev_alloc(c, &c->gpart_v, (size_t)c->M * KK))) return rc; }
        sa.gout = c->gpart_v;
      }
      c->ngp_v = emit ? c->ml : 0;
      c->ngp_w = 0;
      const bool rg = vs_lds_bytes(T, K, c->TF, c->nD, false) > 160 * 1024;     // long depth axis: pivot records in HBM scratch
      if (rg) {
        const size_t need = (size_t)c->ml * n * (c->TF + 3);
        if (need > c->vs_rec_elems) { if ((rc = dev_alloc(c, &c->vs_rec, need))) return rc; c->vs_rec_elems = need; }
        sa.rec_g = c->vs_rec;
        // long depth axes: the prior band of every column by prior_band_kernel (one thread per entry) instead of six
        // stencil walks per thread inside the sampler
        const int TD1 = T * (c->TF + 2);
        if (!c->pband) { if ((rc = dev_alloc(c, &c->pband, (size_t)c->ml * TD1))) return rc; }
        {
          Prof p(c, BTF_K_PRIOR);
          p.launch(prior_band_kernel, dim3((c->ml * TD1 + 255) / 256), dim3(256), 0, (const double*)c->Tau2, c->lam2, c->nD,
                   (const int*)c->st_ptr, (const int*)c->st_row, (const double*)c->st_coef, TD1, c->col0, c->ml, c->pband,
                   (const double*)(c->dev_scalars ? c->hyp : nullptr), 0, (double*)nullptr, 0, 0, 0);
        }
        c->pband_version = 0;
        sa.pband = c->pband;
      }
      if (fuse_v) {
        if ((rc = ensure_fused(c))) return rc;
        ++c->fz_epoch;
        FuseV fv{};
        fv.a = sa;
        if (!sw.lam.hyp) {
          // the prior band of every column, precomputed (rebuilt only when Tau2 / lam2 changed since): the tails load their
          // entries at kernel start.  Not when lam2 is drawn by a side workgroup of this very launch - those tails form it
          const int TD1 = T * (c->TF + 2);
          if (!c->pband) { if ((rc = dev_alloc(c, &c->pband, (size_t)c->ml * TD1))) return rc; c->pband_version = 0; }
          // ... and, where the dataflow tail applies, the same band as LDS images [P | Pm] (v_fused_df copies them as they lie)
          const bool img = c->TF == 2 && K <= 6 && vf_df_fits(T, K, c->TF, c->nD, 16, K < 4 ? 4 : K);
          const int PB = df_layout(T, K, c->TF + 1).PB;
          if (img && !c->pimg) {
            if ((rc = dev_alloc(c, &c->pimg, (size_t)c->ml * 2 * PB))) return rc;
            HIPCHK(c, hipMemsetAsync(c->pimg, 0, (size_t)c->ml * 2 * PB * sizeof(double), c->stream));      // (the zero rows: once)
            c->pimg_version = 0;
          }
          // (full sweeps: the w_solve launch in front has already rebuilt it beside the solves - BandSide, btf_kernels.h)
          c->v_wants_band = true; c->band_img = img; c->band_PB = PB;
          if (c->pband_version != c->prior_version || (img && c->pimg_version != c->prior_version)) {
            Prof p(c, BTF_K_PRIOR);
            if (TD1 <= 512 && c->nD <= 256)      // (T <= 128: one workgroup per column)
              p.launch(prior_band_cols_kernel, dim3(c->ml), dim3(256), 0, (const double*)c->Tau2, c->lam2, c->nD,
                       (const int*)c->st_ptr, (const int*)c->st_row, (const double*)c->st_coef, TD1, c->col0, c->pband,
                       (const double*)(c->dev_scalars ? c->hyp : nullptr), img ? c->pimg : (double*)nullptr, T, c->TF + 2, PB);
            else
              p.launch(prior_band_kernel, dim3((c->ml * TD1 + 255) / 256), dim3(256), 0, (const double*)c->Tau2, c->lam2, c->nD,
                       (const int*)c->st_ptr, (const int*)c->st_row, (const double*)c->st_coef, TD1, c->col0, c->ml, c->pband,
                       (const double*)(c->dev_scalars ? c->hyp : nullptr), 1, img ? c->pimg : (double*)nullptr, T, c->TF + 2, PB);
            c->pband_version = c->prior_version;
            c->pimg_version = img ? c->prior_version : 0;
          }
          fv.a.pband = c->pband;
          if (img && c->pimg_version == c->prior_version) fv.pimg = c->pimg;
        }
        fv.cnt = nch_all > 1 ? c->fz_words + FZ_TICKETS + c->fz_tiles_w : nullptr;
        { static const int be = [] { const char* e = std::getenv("BTF_BAND_EARLY"); return e ? std::atoi(e) : 0; }(); fv.band_early = be; }      // (A/B aid)
        fv.eig_pub = c->fz_pub + FZ_PUB_EIG; fv.eig_flag = c->fz_words + FZ_EIG; fv.epoch = c->fz_epoch;
        side.pub = c->fz_pub + FZ_PUB_EIG; side.flag = c->fz_words + FZ_EIG; side.epoch = c->fz_epoch;
        side.gran = reinterpret_cast<unsigned long long*>(c->fz_pub + FZ_PUB_GRAN);
        fv.eig_gran = side.gran;
        if (sw.lam.hyp) {
          sw.lam.pub = c->fz_pub + FZ_PUB_HYP; sw.lam.flag = c->fz_words + FZ_LAM; sw.lam.epoch = c->fz_epoch;
          fv.hp = HypPub{c->fz_pub + FZ_PUB_HYP, c->fz_words + FZ_LAM, c->fz_epoch, 4};
        }
        K_SWITCH(K, launch_accum<KT>(c, BTF_K_V_ACCUM, mode, c->A_v, c->C_v, c->C8_v, c->W, c->srcmap_v, c->N, c->ldv, rpb,
                                     nch_all, side, sidec, TauSide{}, GramSide{}, cm, sw, nullptr, &fv));
        c->sweep_v++;
        c->nb_L_valid = false;
        c->w_part_valid = false;
        HIPCHK(c, hipGetLastError());
        return after_v_draw(c, compat);
      }
      if (!rg) {
        // the sampler launch of its own (shapes / data the fused tail does not take): the same precomputed band, from the
        // second V half-sweep on with unchanged Tau2 / lam2 (as v_banded_dispatch does for the twisted sampler)
        static const bool lazy_band = [] { const char* e2 = std::getenv("BTF_TWIST_PBAND"); return !e2 || std::atoi(e2) != 0; }();
        const int TD1 = T * (c->TF + 2);
        if (!lazy_band) { }
        else if (c->pband && c->pband_version == c->prior_version) sa.pband = c->pband;
        else if (c->last_v_prior_version == c->prior_version) {
          if (!c->pband) { if ((rc = dev_alloc(c, &c->pband, (size_t)c->ml * TD1))) return rc; }
          Prof p(c, BTF_K_PRIOR);
          p.launch(prior_band_kernel, dim3((c->ml * TD1 + 255) / 256), dim3(256), 0, (const double*)c->Tau2, c->lam2, c->nD,
                   (const int*)c->st_ptr, (const int*)c->st_row, (const double*)c->st_coef, TD1, c->col0, c->ml, c->pband,
                   (const double*)(c->dev_scalars ? c->hyp : nullptr), 1, (double*)nullptr, 0, 0, 0);
          c->pband_version = c->prior_version;
          sa.pband = c->pband;
        }
        c->last_v_prior_version = c->prior_version;
      }
      const size_t sl = vs_lds_bytes(T, K, c->TF, c->nD, rg);
      switch ((c->TF + 1) * 2 + (rg ? 1 : 0)) {
        case 2: e = launch_vspectral<1, false>(c, sa, sl); break;
        case 3: e = launch_vspectral<1, true>(c, sa, sl); break;
        case 4: e = launch_vspectral<2, false>(c, sa, sl); break;
        case 5: e = launch_vspectral<2, true>(c, sa, sl); break;
        case 6:                                  // tf_order = 2, records in LDS: the instances with nembeds compiled in
          switch (K) {
#define VS_FIXED(KV) case KV: e = launch_vspectral<3, false, KV>(c, sa, sl); break
            VS_FIXED(1); VS_FIXED(2); VS_FIXED(3); VS_FIXED(4); VS_FIXED(5); VS_FIXED(6); VS_FIXED(7); VS_FIXED(8); VS_FIXED(9); VS_FIXED(10);
#undef VS_FIXED
            default: e = launch_vspectral<3, false>(c, sa, sl); break;
          }
          break;
        case 7:                                  // tf_order = 2, records in HBM scratch (long depth axes): nembeds compiled in as well
          switch (K) {
#define VS_FIXED(KV) case KV: e = launch_vspectral<3, true, KV>(c, sa, sl); break
            VS_FIXED(1); VS_FIXED(2); VS_FIXED(3); VS_FIXED(4); VS_FIXED(5); VS_FIXED(6); VS_FIXED(7); VS_FIXED(8); VS_FIXED(9); VS_FIXED(10);
#undef VS_FIXED
            default: e = launch_vspectral<3, true>(c, sa, sl); break;
          }
          break;
        case 8: e = launch_vspectral<4, false>(c, sa, sl); break;
        default: e = launch_vspectral<4, true>(c, sa, sl); break;
      }
      HIPCHK(c, e);
      c->sweep_v++;
      c->nb_L_valid = false;
      c->w_part_valid = false;
      HIPCHK(c, hipGetLastError());
      return after_v_draw(c, compat);
    }
    if ((rc = v_banded_dispatch(c, choice, dz, seed, nch, use_gw, eps0, attempts, c->V, false))) return rc;
  }
  c->v_local_done = false;
  c->sweep_v++;
  c->nb_L_valid = false;
  c->w_part_valid = false;
  HIPCHK(c, hipGetLastError());
  return after_v_draw(c, compat);
}

// ------------------------------------------------------------------ elliptical slice sampling
namespace {
struct EssDims { int nchains, per, nbx, nsum; long long n; };
int ess_alloc(btf_ctx* c) {
  if (c->essX0) return BTF_OK;
  const size_t nx = std::max((size_t)c->N * c->K, (size_t)c->M * c->T * c->K);
  const size_t nc = (size_t)std::max(std::max(c->N, c->M), 1);
  int rc;
  if ((rc = dev_alloc(c, &c->essX0, nx))) return rc;
  if ((rc = dev_alloc(c, &c->essNu, nx))) return rc;
  if ((rc = dev_alloc(c, &c->ess_st, nc * 5))) return rc;
  if ((rc = dev_alloc(c, &c->ess_theta, nc))) return rc;
  if ((rc = dev_alloc(c, &c->ess_done, nc))) return rc;
  HIPCHK(c, hipMemsetAsync(c->ess_done, 0, nc * sizeof(int), c->stream));
  return BTF_OK;
}
// allow_host: only btf_ess_begin / btf_ess_eval take BTF_ESS_HOST_LIKELIHOOD (no device likelihood runs for it); every
// entry point that launches a likelihood kernel rejects a negative family instead of handing it to the kernels
int ess_check(btf_ctx* c, int what, int link, bool allow_host = false) {
  if (what < 0 || what > 1 || link < BTF_ESS_HOST_LIKELIHOOD || link >= ESS_FAM_COUNT) return fail(c, BTF_EINVAL, "bad elliptical-slice arguments");
  if (link == BTF_ESS_HOST_LIKELIHOOD && !allow_host)
    return fail(c, BTF_EINVAL, "BTF_ESS_HOST_LIKELIHOOD is taken by btf_ess_begin / btf_ess_eval only (a device likelihood family expected)");
  if (link == BTF_ESS_HOST_LIKELIHOOD) {                   // the caller's own likelihood: nothing of the data is needed here
    if (!c->have_W || !c->have_V || !c->have_hyper) return fail(c, BTF_ESTATE, "elliptical slice sampling needs W, V and hyper-parameters");
    if (c->nl != c->N || c->ml != c->M) return fail(c, BTF_ESTATE, "elliptical slice sampling needs an unsharded context");
    return BTF_OK;
  }
  if (!c->have_data || c->binomial || !c->have_W || !c->have_V || !c->have_hyper)
    return fail(c, BTF_ESTATE, "elliptical slice sampling needs count data (btf_set_data_gaussian statistics), W, V and hyper-parameters");
  if (c->nl != c->N || c->ml != c->M) return fail(c, BTF_ESTATE, "elliptical slice sampling needs an unsharded context");
  return BTF_OK;
}
EssDims ess_dims(const btf_ctx* c, int what, int mode) {
  EssDims d;
  d.n = what == 0 ? (long long)c->N * c->K : (long long)c->M * c->T * c->K;
  if (mode == 0) {                                   // joint: one chain over everything, rows-layout partials
    d.nchains = 1; d.per = 0;
    d.nbx = std::max(1, std::min((c->M * c->T + ESS_THREADS - 1) / ESS_THREADS, std::max(1, 2048 / std::max(c->N, 1))));
    d.nsum = c->N * d.nbx;
  } else if (what == 0) {                            // a chain per row
    d.nchains = c->N; d.per = c->K;
    d.nbx = std::max(1, std::min((c->M * c->T + ESS_THREADS - 1) / ESS_THREADS, std::max(1, 2048 / std::max(c->N, 1))));
    d.nsum = d.nbx;
  } else {                                           // a chain per column
    d.nchains = c->M; d.per = c->T * c->K;
    d.nbx = std::max(1, std::min((c->N + ESS_THREADS - 1) / ESS_THREADS, std::max(1, 2048 / std::max(c->M, 1))));
    d.nsum = d.nbx;
  }
  return d;
}
int ess_ensure_part(btf_ctx* c, const EssDims& d) {
  const size_t need = (size_t)std::max(c->N, c->M) * d.nbx;
  if (need > c->ess_part_elems) {
    int rc = dev_alloc(c, &c->ess_part, need);
    if (rc) return rc;
    c->ess_part_elems = need;
  }
  return BTF_OK;
}
// X0 <- current state, Nu <- prior draw (W: sigma z on the free entries; V: the banded sampler with the likelihood off)
int ess_begin(btf_ctx* c, int what, const double* z, uint64_t seed, double eps0, int attempts) {
  int rc;
  if ((rc = ess_alloc(c))) return rc;
  const int K = c->K, n = c->T * K;
  if (what == 0) {
    const size_t nw = (size_t)c->N * K;
    HIPCHK(c, hipMemcpyAsync(c->essX0, c->W, nw * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    const double* dz = nullptr;
    if (z) {
      const size_t nz = (size_t)w_z_offset(c->N, K);
      if ((rc = ensure_z(c, nz))) return rc;
      HIPCHK(c, hipMemcpyAsync(c->zbuf, z, nz * sizeof(double), hipMemcpyHostToDevice, c->stream));
      dz = c->zbuf;
    }
    Prof p(c, BTF_K_ESS);
    p.launch(ess_w_prior_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, c->essNu, c->N, K, c->sigma2,
             (const double*)(c->dev_scalars ? c->hyp : nullptr), dz, (unsigned long long)seed, 2 * c->sweep_w + 0x20000ULL);
    c->sweep_w++;
  } else {
    const size_t nv = (size_t)c->M * n;
    HIPCHK(c, hipMemcpyAsync(c->essX0, c->V, nv * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    const double* dz = nullptr;
    if (z) {
      if ((rc = ensure_z(c, nv))) return rc;
      HIPCHK(c, hipMemcpyAsync(c->zbuf, z, nv * sizeof(double), hipMemcpyHostToDevice, c->stream));
      dz = c->zbuf;
    }
    if ((rc = ensure_part(c, 1))) return rc;
    if ((rc = v_banded_dispatch(c, banded_choice(c, false), dz, seed, 0, false, eps0, attempts, c->essNu, true))) return rc;
    c->sweep_v++;
  }
  HIPCHK(c, hipGetLastError());
  return BTF_OK;
}
}  // namespace

int btf_ess_begin(btf_ctx* c, int what, const double* z, uint64_t seed, double eps0, int attempts) {
  if (!c) return BTF_EINVAL;
  int rc;
  if ((rc = ess_check(c, what, BTF_ESS_HOST_LIKELIHOOD, true))) return rc;      // (the prior draw needs no data; btf_ess_eval checks what its family needs)
  HIPCHK(c, hipSetDevice(c->dev));
  if ((rc = ess_begin(c, what, z, seed, eps0, attempts < 0 ? 0 : attempts))) return rc;
  // one joint chain: not done
  HIPCHK(c, hipMemsetAsync(c->ess_done, 0, sizeof(int), c->stream));
  return BTF_OK;
}

int btf_ess_eval(btf_ctx* c, int what, double theta, int current, int link, double* ll) {
  if (!c || !ll) return BTF_EINVAL;
  int rc;
  if ((rc = ess_check(c, what, link, true))) return rc;
  if (!c->essX0) return fail(c, BTF_ESTATE, "btf_ess_eval follows btf_ess_begin");
  HIPCHK(c, hipSetDevice(c->dev));
  const EssDims d = ess_dims(c, what, 0);
  if ((rc = ess_ensure_part(c, d))) return rc;
  if (!current) {
    HIPCHK(c, hipMemcpyAsync(c->ess_theta, &theta, sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));           // theta is a host temporary
    Prof p(c, BTF_K_ESS);
    p.launch(ess_combine_kernel, dim3((unsigned)((d.n + 255) / 256)), dim3(256), 0, (const double*)c->essX0, (const double*)c->essNu,
             what == 0 ? c->W : c->V, d.n, 0, (const double*)c->ess_theta, (const int*)c->ess_done, 0);
    c->w_local_done = c->v_local_done = false; c->sse_cols_valid = false;
    if (what == 0) { c->ngp_w = 0; c->w_part_valid = false; } else { c->ngp_v = 0; c->w_part_valid = false; }
    c->nb_L_valid = false;
  }
  if (link == BTF_ESS_HOST_LIKELIHOOD) {                   // the proposal stands in W / V: the caller evaluates its own function
    *ll = 0.0;
    HIPCHK(c, hipGetLastError());
    return check_status(c);
  }
  K_SWITCH(c->K, launch_ess_ll<KT>(c, what, 0, link, d.nbx));
  HIPCHK(c, hipGetLastError());
  std::vector<double> h((size_t)d.nsum);
  HIPCHK(c, hipMemcpyAsync(h.data(), c->ess_part, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double s = 0.0;
  for (double v : h) s += v;            // fixed order
  *ll = s;
  return check_status(c);
}

int btf_ess_run(btf_ctx* c, int what, int link, int mode, const double* z, uint64_t seed, int max_rounds, double eps0, int attempts) {
  if (!c) return BTF_EINVAL;
  int rc;
  if ((rc = ess_check(c, what, link))) return rc;
  if (mode < 0 || mode > 1 || max_rounds < 1) return fail(c, BTF_EINVAL, "bad elliptical-slice arguments");
  HIPCHK(c, hipSetDevice(c->dev));
  if ((rc = ess_begin(c, what, z, seed, eps0, attempts < 0 ? 0 : attempts))) return rc;
  const EssDims d = ess_dims(c, what, mode);
  if ((rc = ess_ensure_part(c, d))) return rc;
  HIPCHK(c, hipMemsetAsync(c->ess_done, 0, (size_t)d.nchains * sizeof(int), c->stream));
  double* X = what == 0 ? c->W : c->V;
  const unsigned long long dseed = seed * 0x9E3779B97F4A7C15ULL + 0x5851F42D4C957F2DULL;
  auto decide = [&](int round) {
    Prof p(c, BTF_K_ESS);
    p.launch(ess_decide_kernel, dim3(d.nchains), dim3(ESS_THREADS), 0, (const double*)c->ess_part, d.nsum, d.nchains, c->ess_st,
             c->ess_theta, c->ess_done, round, dseed);
  };
  K_SWITCH(c->K, launch_ess_ll<KT>(c, what, mode, link, d.nbx));      // ll of the current state
  decide(-1);
  for (int r = 0; r < max_rounds; ++r) {
    {
      Prof p(c, BTF_K_ESS);
      p.launch(ess_combine_kernel, dim3((unsigned)((d.n + 255) / 256)), dim3(256), 0, (const double*)c->essX0, (const double*)c->essNu, X,
               d.n, d.per, (const double*)c->ess_theta, (const int*)c->ess_done, 0);
    }
    K_SWITCH(c->K, launch_ess_ll<KT>(c, what, mode, link, d.nbx));
    decide(r);
  }
  {   // chains that used up the rounds keep the current state (never observed: the bracket halves every round)
    Prof p(c, BTF_K_ESS);
    p.launch(ess_combine_kernel, dim3((unsigned)((d.n + 255) / 256)), dim3(256), 0, (const double*)c->essX0, (const double*)c->essNu, X,
             d.n, d.per, (const double*)c->ess_theta, (const int*)c->ess_done, 1);
  }
  HIPCHK(c, hipGetLastError());
  c->ess_last_chains = d.nchains;
  if (what == 0) c->ngp_w = 0; else c->ngp_v = 0;
  c->w_part_valid = false; c->w_local_done = c->v_local_done = false; c->sse_cols_valid = false;
  c->nb_L_valid = false;
  return BTF_OK;
}

// ------------------------------------------------------------------ generalized analytic slice sampling
namespace {
int gass_alloc(btf_ctx* c) {
  if (c->gs_mask) return BTF_OK;
  const size_t nc = (size_t)std::max(c->N, c->M);
  int rc;
  if ((rc = dev_alloc(c, &c->gs_mask, nc * GASS_GRID))) return rc;
  if ((rc = dev_alloc(c, &c->gs_info, nc * 2))) return rc;
  if ((rc = dev_alloc(c, &c->gs_thetas, nc * GASS_MAXC))) return rc;
  if ((rc = dev_alloc(c, &c->gs_ntheta, nc))) return rc;
  if ((rc = dev_alloc(c, &c->gs_ll, nc * GASS_MAXC))) return rc;
  if ((rc = dev_alloc(c, &c->gs_hh, nc))) return rc;
  if ((rc = dev_alloc(c, &c->gs_cur, nc))) return rc;
  if ((rc = dev_alloc(c, &c->gs_nacc, nc))) return rc;
  if ((rc = dev_alloc(c, &c->gs_u, nc))) return rc;
  return BTF_OK;
}
int gass_check(btf_ctx* c, int what, int link) {
  int rc;
  if ((rc = ess_check(c, what, link))) return rc;
  if (!c->gs_cons) return fail(c, BTF_ESTATE, "btf_gass_set_constraints first");
  return BTF_OK;
}
int gass_eval_launch(btf_ctx* c, int what, int link) {
  GassEvalArgs a{};
  a.X0 = c->essX0; a.Nu = c->essNu; a.N = c->N; a.M = c->M; a.T = c->T; a.K = c->K; a.Rc = (double)c->R;
  a.thetas = c->gs_thetas; a.ntheta = c->gs_ntheta;
  // few chains: a chain's cell tiles are dealt to several workgroups (>= ~1024 in all: four waves per SIMD hide the LDS
  // and dependency stalls one 4-wave workgroup per CU leaves open), their partial sums added in order afterwards
  const int nch = what == 0 ? c->N : c->M;
  const int ntiles = ((what == 0 ? c->M : c->N) * c->T + GASS_CT - 1) / GASS_CT;
  const int nsplit = std::max(1, std::min(std::min(8, ntiles), (1024 + nch - 1) / nch));
  if (nsplit > 1) {
    const size_t need = (size_t)nch * nsplit * GASS_MAXC;
    if (need > c->gs_llp_elems) { int rc; if ((rc = dev_alloc(c, &c->gs_llp, need))) return rc; c->gs_llp_elems = need; }
  }
  a.ll = nsplit > 1 ? c->gs_llp : c->gs_ll;
  a.nsplit = nsplit;
  Prof p(c, BTF_K_ESS);
  const dim3 grid(nch, nsplit);
  a.lf = LikFam{link, c->lik_par[link]};
  const int tl = ess_link_of(link);
  if (what == 0) {
    a.F = c->V; a.A = c->A_v; a.C8 = c->C8_v; a.Cd = c->C_v; a.ld = c->ldv;
    if (tl == ESS_LINK_LOG) p.launch(gass_eval_kernel<ESS_LINK_LOG, true>, grid, dim3(GASS_THREADS), 0, a);
    else if (tl == ESS_LINK_IDENTITY) p.launch(gass_eval_kernel<ESS_LINK_IDENTITY, true>, grid, dim3(GASS_THREADS), 0, a);
    else p.launch(gass_eval_kernel<ESS_LINK_GENERIC, true>, grid, dim3(GASS_THREADS), 0, a);
  } else {
    a.F = c->W; a.A = c->A_wT; a.C8 = c->C8_wT; a.Cd = c->C_wT; a.ld = c->ldw;
    if (tl == ESS_LINK_LOG) p.launch(gass_eval_kernel<ESS_LINK_LOG, false>, grid, dim3(GASS_THREADS), 0, a);
    else if (tl == ESS_LINK_IDENTITY) p.launch(gass_eval_kernel<ESS_LINK_IDENTITY, false>, grid, dim3(GASS_THREADS), 0, a);
    else p.launch(gass_eval_kernel<ESS_LINK_GENERIC, false>, grid, dim3(GASS_THREADS), 0, a);
  }
  if (nsplit > 1)
    hipLaunchKernelGGL(gass_ll_sum_kernel, dim3((nch * GASS_MAXC + 255) / 256), dim3(256), 0, c->stream, (const double*)c->gs_llp, nsplit,
                       (const int*)c->gs_ntheta, nch, c->gs_ll);
  HIPCHK(c, hipGetLastError());
  return BTF_OK;
}
}  // namespace

int btf_gass_set_constraints(btf_ctx* c, const double* cons, int J, const double* row_cons, int nrc) {
  if (!c || !cons || J < 1 || nrc < 0 || (nrc > 0 && !row_cons)) return fail(c, BTF_EINVAL, "bad constraint arguments");
  HIPCHK(c, hipSetDevice(c->dev));
  const int T = c->T, K = c->K;
  std::vector<double> A((size_t)J * T), cc((size_t)J);
  for (int q = 0; q < J; ++q) {
    for (int t = 0; t < T; ++t) A[(size_t)q * T + t] = cons[(size_t)q * (T + 1) + t];
    cc[q] = cons[(size_t)q * (T + 1) + T];
  }
  int rc;
  if ((rc = dev_alloc(c, &c->gs_cons, A.size()))) return rc;
  if ((rc = dev_alloc(c, &c->gs_cc, cc.size()))) return rc;
  HIPCHK(c, hipMemcpy(c->gs_cons, A.data(), A.size() * sizeof(double), hipMemcpyHostToDevice));
  {   // the non-zeros of the constraint matrix, row by row in ascending depth (the column analysis walks them when they are few)
    std::vector<int> ptr((size_t)J + 1, 0), idx;
    std::vector<double> val;
    for (int q = 0; q < J; ++q) {
      for (int t = 0; t < T; ++t) if (A[(size_t)q * T + t] != 0.0) { idx.push_back(t); val.push_back(A[(size_t)q * T + t]); }
      ptr[(size_t)q + 1] = (int)idx.size();
    }
    c->gs_cnnz = 0;
    // used when it fits the dense matrix's LDS area and saves at least half of the work
    if (!idx.empty() && (size_t)3 * idx.size() + (size_t)J + 4 <= (size_t)2 * J * T && 2 * idx.size() <= (size_t)J * T) {
      if ((rc = dev_alloc(c, &c->gs_cptr, ptr.size()))) return rc;
      if ((rc = dev_alloc(c, &c->gs_cidx, idx.size()))) return rc;
      if ((rc = dev_alloc(c, &c->gs_cval, val.size()))) return rc;
      HIPCHK(c, hipMemcpy(c->gs_cptr, ptr.data(), ptr.size() * sizeof(int), hipMemcpyHostToDevice));
      HIPCHK(c, hipMemcpy(c->gs_cidx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
      HIPCHK(c, hipMemcpy(c->gs_cval, val.data(), val.size() * sizeof(double), hipMemcpyHostToDevice));
      c->gs_cnnz = (int)idx.size();
    }
  }
  HIPCHK(c, hipMemcpy(c->gs_cc, cc.data(), cc.size() * sizeof(double), hipMemcpyHostToDevice));
  if ((rc = dev_alloc(c, &c->gs_av, (size_t)c->M * J * K))) return rc;
  if (c->gs_rc) { (void)hipFree(c->gs_rc); c->gs_rc = nullptr; }
  if (nrc > 0) {
    if ((rc = dev_alloc(c, &c->gs_rc, (size_t)nrc * (K + 1)))) return rc;
    HIPCHK(c, hipMemcpy(c->gs_rc, row_cons, (size_t)nrc * (K + 1) * sizeof(double), hipMemcpyHostToDevice));
  }
  c->gs_J = J; c->gs_nrc = nrc;
  return BTF_OK;
}

int btf_gass_begin(btf_ctx* c, int what, int link, const double* z, const double* u, uint64_t seed, double eps0, int attempts,
                   int pick_ngrid) {
  if (!c) return BTF_EINVAL;
  int rc;
  if ((rc = gass_check(c, what, link))) return rc;
  if (pick_ngrid < 0 || pick_ngrid > GASS_MAXC) return fail(c, BTF_EINVAL, "at most 128 candidates per chain");
  HIPCHK(c, hipSetDevice(c->dev));
  if ((rc = gass_alloc(c))) return rc;
  if ((rc = ess_begin(c, what, z, seed, eps0, attempts < 0 ? 0 : attempts))) return rc;     // X0 <- state, Nu <- prior draw
  const int nch = what == 0 ? c->N : c->M;
  // slice heights from the likelihood of the current state, chain by chain
  const EssDims d = ess_dims(c, what, 1);
  if ((rc = ess_ensure_part(c, d))) return rc;
  HIPCHK(c, hipMemsetAsync(c->ess_done, 0, (size_t)d.nchains * sizeof(int), c->stream));
  K_SWITCH(c->K, launch_ess_ll<KT>(c, what, 1, link, d.nbx));
  const double* du = nullptr;
  if (u) {
    HIPCHK(c, hipMemcpyAsync(c->gs_u, u, (size_t)nch * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    du = c->gs_u;
  }
  const unsigned long long dseed = seed * 0x9E3779B97F4A7C15ULL + 0x6A09E667F3BCC909ULL;
  {
    Prof p(c, BTF_K_ESS);
    p.launch(gass_slice_kernel, dim3((nch + 255) / 256), dim3(256), 0, (const double*)c->ess_part, d.nsum, nch, du, dseed, c->gs_hh, c->gs_cur);
  }
  GassArgs a{};
  a.X0 = c->essX0; a.Nu = c->essNu; a.Cons = c->gs_cons; a.Cc = c->gs_cc; a.J = c->gs_J;
  if (c->gs_cnnz > 0) { a.cs_ptr = c->gs_cptr; a.cs_idx = c->gs_cidx; a.cs_val = c->gs_cval; a.cs_nnz = c->gs_cnnz; }
  a.AV = c->gs_av; a.Rc = c->gs_rc; a.nrc = what == 0 ? c->gs_nrc : 0; a.W = c->W;
  a.N = c->N; a.M = c->M; a.T = c->T; a.K = c->K;
  a.vmask = c->gs_mask; a.info = c->gs_info; a.pick = pick_ngrid > 0 ? 1 : 0; a.ngrid = pick_ngrid;
  a.thetas = c->gs_thetas; a.ntheta = c->gs_ntheta; a.seed = dseed;
  {
    Prof p(c, BTF_K_ESS);
    if (what == 0) {
      p.launch(gass_av_kernel, dim3(c->M), dim3(GASS_THREADS), 0, (const double*)c->V, (const double*)c->gs_cons, c->T, c->K, c->gs_J, c->gs_av);
      p.launch(gass_analyse_rows_kernel, dim3(c->N), dim3(GASS_THREADS), 0, a);
    } else {
      const size_t dyn = ((size_t)2 * GASS_RT * c->T + (size_t)c->gs_J * c->T) * sizeof(double);
      constexpr size_t GASS_DYN_MAX = 112 * 1024;      // (the kernel's static scratch takes the rest of the 160 KB)
      if (dyn > GASS_DYN_MAX || dyn + sizeof(GassScratch) > 158 * 1024)
        return fail(c, BTF_EINVAL, "constraint matrix too large for the column analysis");
      static std::atomic<unsigned long long> attr_set{0};
      if (!dev_flag_is_set(attr_set, c->dev)) {
        HIPCHK(c, hipFuncSetAttribute((const void*)gass_analyse_cols_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GASS_DYN_MAX));
        dev_flag_set(attr_set, c->dev);
      }
      p.launch(gass_analyse_cols_kernel, dim3(c->M), dim3(GASS_THREADS), dyn, a);
    }
  }
  HIPCHK(c, hipGetLastError());
  c->gs_chains = nch; c->gs_what = what; c->gs_link = link;
  return BTF_OK;
}

int btf_gass_grid(btf_ctx* c, int what, int32_t* info, uint8_t* mask, double* slice, double* cur_ll) {
  if (!c || !info) return BTF_EINVAL;
  if (c->gs_what != what || c->gs_chains < 1) return fail(c, BTF_ESTATE, "btf_gass_grid follows btf_gass_begin of the same factor");
  HIPCHK(c, hipSetDevice(c->dev));
  const size_t nch = (size_t)c->gs_chains;
  HIPCHK(c, hipMemcpyAsync(info, c->gs_info, nch * 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  if (mask) HIPCHK(c, hipMemcpyAsync(mask, c->gs_mask, nch * GASS_GRID, hipMemcpyDeviceToHost, c->stream));
  if (slice) HIPCHK(c, hipMemcpyAsync(slice, c->gs_hh, nch * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (cur_ll) HIPCHK(c, hipMemcpyAsync(cur_ll, c->gs_cur, nch * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  return check_status(c);
}

int btf_gass_eval(btf_ctx* c, int what, const double* thetas, const int32_t* ntheta, double* ll_out) {
  if (!c) return BTF_EINVAL;
  if (c->gs_what != what || c->gs_chains < 1) return fail(c, BTF_ESTATE, "btf_gass_eval follows btf_gass_begin of the same factor");
  HIPCHK(c, hipSetDevice(c->dev));
  const size_t nch = (size_t)c->gs_chains;
  if (thetas) {
    if (!ntheta) return fail(c, BTF_EINVAL, "candidate counts missing");
    for (size_t q = 0; q < nch; ++q) if (ntheta[q] < 0 || ntheta[q] > GASS_MAXC) return fail(c, BTF_EINVAL, "at most 128 candidates per chain");
    HIPCHK(c, hipMemcpyAsync(c->gs_thetas, thetas, nch * GASS_MAXC * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->gs_ntheta, ntheta, nch * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  int rc;
  if ((rc = gass_eval_launch(c, what, c->gs_link))) return rc;
  if (ll_out) {
    HIPCHK(c, hipMemcpyAsync(ll_out, c->gs_ll, nch * GASS_MAXC * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return check_status(c);
  }
  return BTF_OK;
}

int btf_gass_commit(btf_ctx* c, int what, const double* theta, const int32_t* keep) {
  if (!c || !theta || !keep) return BTF_EINVAL;
  if (c->gs_what != what || c->gs_chains < 1) return fail(c, BTF_ESTATE, "btf_gass_commit follows btf_gass_begin of the same factor");
  HIPCHK(c, hipSetDevice(c->dev));
  const size_t nch = (size_t)c->gs_chains;
  HIPCHK(c, hipMemcpyAsync(c->ess_theta, theta, nch * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->ess_done, keep, nch * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int per = what == 0 ? c->K : c->T * c->K;
  const long long n = (long long)nch * per;
  {
    Prof p(c, BTF_K_ESS);
    p.launch(ess_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (const double*)c->essX0, (const double*)c->essNu,
             what == 0 ? c->W : c->V, n, per, (const double*)c->ess_theta, (const int*)c->ess_done, 0);
  }
  HIPCHK(c, hipGetLastError());
  if (what == 0) c->ngp_w = 0; else c->ngp_v = 0;
  c->w_part_valid = false; c->w_local_done = c->v_local_done = false; c->sse_cols_valid = false; c->nb_L_valid = false;
  return BTF_OK;
}

int btf_gass_select(btf_ctx* c, int what, uint64_t seed, int32_t* naccept_out) {
  if (!c) return BTF_EINVAL;
  if (c->gs_what != what || c->gs_chains < 1) return fail(c, BTF_ESTATE, "btf_gass_select follows btf_gass_begin of the same factor");
  HIPCHK(c, hipSetDevice(c->dev));
  const int nch = c->gs_chains, per = what == 0 ? c->K : c->T * c->K;
  {
    Prof p(c, BTF_K_ESS);
    p.launch(gass_select_kernel, dim3(nch), dim3(GASS_THREADS), 0, (const double*)c->gs_ll, (const int*)c->gs_ntheta, (const double*)c->gs_thetas,
             (const double*)c->gs_hh, (const double*)c->essX0, (const double*)c->essNu, what == 0 ? c->W : c->V, per,
             (unsigned long long)(seed * 0x9E3779B97F4A7C15ULL + 0xBB67AE8584CAA73BULL), c->gs_nacc, (double*)nullptr);
  }
  HIPCHK(c, hipGetLastError());
  if (what == 0) c->ngp_w = 0; else c->ngp_v = 0;
  c->w_part_valid = false; c->w_local_done = c->v_local_done = false; c->sse_cols_valid = false; c->nb_L_valid = false;
  if (naccept_out) {
    HIPCHK(c, hipMemcpyAsync(naccept_out, c->gs_nacc, (size_t)nch * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    return check_status(c);
  }
  return BTF_OK;
}

int btf_gass_run(btf_ctx* c, int what, int link, uint64_t seed, int ngrid, double eps0, int attempts) {
  if (ngrid < 1) return fail(c, BTF_EINVAL, "ngrid must be positive");
  int rc;
  if ((rc = btf_gass_begin(c, what, link, nullptr, nullptr, seed, eps0, attempts, ngrid))) return rc;
  if ((rc = btf_gass_eval(c, what, nullptr, nullptr, nullptr))) return rc;
  return btf_gass_select(c, what, seed, nullptr);
}

// n whole Gibbs sweeps of the scalar-noise Gaussian model with every draw on the device, queued from here: the order and
// the launches of GaussianBTF.resample under rng="device" (nu2, sigma2 | Tau2 chain in the W accumulation launch | lam2 |
// W | V; factor.py:306-311, :112-128), the five seeds of sweep s being seed_base + draws0 + 5 s + 1..5 - the sequence
// functionalmf_amd.factor draws them in, so a chain driven from here equals one driven sweep by sweep from Python.
int btf_gibbs_sweeps(btf_ctx* c, int n, uint64_t seed_base, uint64_t draws0, int compat, double nu2_a, double nu2_b,
                     double sigma2_a, double sigma2_b, double stability, double eps0, int attempts) {
  if (!c || n < 0) return BTF_EINVAL;
  if (!c->dev_scalars || c->binomial || c->counts) return fail(c, BTF_ESTATE, "btf_gibbs_sweeps: Gaussian data with device-resident scalars");
  if (c->nl != c->N || c->ml != c->M) return fail(c, BTF_ESTATE, "btf_gibbs_sweeps: unsharded contexts (a sharded sweep has exchanges between its steps)");
  if (!c->have_data || !c->have_W || !c->have_V || !c->have_hyper || !c->have_chain)
    return fail(c, BTF_ESTATE, "set data, W, V, the hyper-parameters and the horseshoe+ chain first");
  int rc;
  for (int s = 0; s < n; ++s) {
    const uint64_t d = seed_base + draws0 + 5ULL * (uint64_t)s;
    if ((rc = btf_queue_Tau2(c, d + 1, stability))) return rc;
    if ((rc = btf_queue_lam2(c, d + 2, compat))) return rc;
    // four launches when the previous sweep's V sampler left the residual parts behind (BTF_OPT_FUSED_SWEEP): nu2 and
    // sigma2 ride in the W accumulation launch, lam2 in the V accumulation launch; six otherwise (first sweep, weighted data)
    int32_t queued = 0;
    if ((rc = btf_queue_scalars(c, d + 3, 3, nu2_a, nu2_b, sigma2_a, sigma2_b, &queued))) return rc;
    if ((rc = btf_w_accum(c, compat))) return rc;
    if (!queued) { if ((rc = btf_draw_scalars(c, d + 3, 7, nu2_a, nu2_b, sigma2_a, sigma2_b))) return rc; }
    if ((rc = btf_resample_W(c, nullptr, d + 4, compat))) return rc;
    if ((rc = btf_resample_V(c, nullptr, d + 5, compat, eps0, attempts))) return rc;
    if (c->col_every > 0 && --c->col_count == 0) {        // btf_collect_schedule: keep this state
      if ((rc = btf_collect(c, c->col_slot++))) return rc;
      c->col_count = c->col_every;
      if (c->col_slot >= c->smp_n) c->col_every = 0;
    }
  }
  return BTF_OK;
}

int btf_wv_steps(btf_ctx* c, int n, uint64_t seed_base, uint64_t draws0, int compat, double eps0, int attempts) {
  if (!c || n < 0) return BTF_EINVAL;
  if (c->nl != c->N || c->ml != c->M) return fail(c, BTF_ESTATE, "btf_wv_steps: unsharded contexts (a sharded step has an exchange behind each half-sweep)");
  if (!c->have_data || !c->have_W || !c->have_V || !c->have_hyper) return fail(c, BTF_ESTATE, "set data, W, V and hyper-parameters first");
  int rc;
  for (int s = 0; s < n; ++s) {
    const uint64_t d = seed_base + draws0 + 2ULL * (uint64_t)s;
    if ((rc = btf_resample_W(c, nullptr, d + 1, compat))) return rc;
    if ((rc = btf_resample_V(c, nullptr, d + 2, compat, eps0, attempts))) return rc;
  }
  return BTF_OK;
}

int btf_set_likelihood_param(btf_ctx* c, int family, double par) {
  if (!c || family < 0 || family >= ESS_FAM_COUNT || !(par == par)) return fail(c, BTF_EINVAL, "bad likelihood family / parameter");
  if ((family == ESS_FAM_GAUSSIAN || family == ESS_FAM_NEGBIN_LOGIT) && !(par > 0.0)) return fail(c, BTF_EINVAL, "the parameter must be positive");
  c->lik_par[family] = par;
  return BTF_OK;
}

int btf_ess_info(btf_ctx* c, int32_t* unfinished, double* ll_first) {
  if (!c || !unfinished) return BTF_EINVAL;
  if (!c->essX0 || c->ess_last_chains < 1) return fail(c, BTF_ESTATE, "no elliptical-slice run yet");
  HIPCHK(c, hipSetDevice(c->dev));
  std::vector<int> dn((size_t)c->ess_last_chains);
  double st[5];
  HIPCHK(c, hipMemcpyAsync(dn.data(), c->ess_done, dn.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(st, c->ess_st, sizeof(st), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int u = 0;
  for (int v : dn) u += v ? 0 : 1;
  *unfinished = u;
  if (ll_first) *ll_first = st[4];
  return check_status(c);
}

int btf_get_V_order(btf_ctx* c, int32_t* order) {
  if (!c || !order) return BTF_EINVAL;
  const int n = c->T * c->K;
  if (banded_choice(c) == 2) {
    const TwLayout W = tw_layout(c->T, c->K, c->TF, 0);
    for (int i = 0; i < n; ++i) order[i] = twist_order(i, n, W.nl, W.nr);
  } else {
    for (int i = 0; i < n; ++i) order[i] = i;
  }
  return BTF_OK;
}

int btf_get_V_attempts(btf_ctx* c, int32_t* tries) {
  if (!c || !tries) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipMemcpyAsync(tries, c->tries, (size_t)c->ml * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return BTF_OK;
}

// -------------------------------------------------------------------------- SSE
// launch the SSE reduction and queue the device-to-host copies of its block partials and of W
// into pinned memory; nothing is waited for (keeps a full sweep to two host syncs)
namespace {
// launch the SSE block reduction into c->bsum (nb block partials); no copies, no waits
int sse_launch(btf_ctx* c, size_t* nb_out) {
  const int ncols = c->ml * c->T;
  const int gx = (ncols + SSE_THREADS - 1) / SSE_THREADS;
  int nrb = std::max(1, std::min(c->N / 16, 2048 / std::max(1, gx)));
  const int rpb = (c->N + nrb - 1) / nrb;
  nrb = (c->N + rpb - 1) / rpb;
  const size_t nb = ncols > 0 ? (size_t)gx * nrb : 0;
  int rc;
  if (nb > c->bsum_elems) {
    if ((rc = dev_alloc(c, &c->bsum, nb))) return rc;
    c->bsum_elems = nb;
  }
  if (ncols > 0) { K_SWITCH(c->K, launch_sse<KT>(c, c->A_v, c->C_v, (double)c->R, ncols, c->ldv, rpb, nrb)); }
  HIPCHK(c, hipGetLastError());
  *nb_out = nb;
  return BTF_OK;
}
int ensure_hyp(btf_ctx* c) {
  if (c->hyp) return BTF_OK;
  int rc;
  if ((rc = dev_alloc(c, &c->hyp, (size_t)HYP_COUNT))) return rc;
  HIPCHK(c, hipHostMalloc((void**)&c->pin_hyp, HYP_COUNT * sizeof(double), hipHostMallocDefault));
  for (int i = 0; i < HYP_COUNT; ++i) c->pin_hyp[i] = 1.0;
  HIPCHK(c, hipMemcpyAsync(c->hyp, c->pin_hyp, HYP_COUNT * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return BTF_OK;
}
}  // namespace

int btf_sse_begin(btf_ctx* c) {
  if (!c) return BTF_EINVAL;
  if (!c->have_data || c->binomial || !c->have_W || !c->have_V) return fail(c, BTF_ESTATE, "btf_sse needs Gaussian data, W and V");
  HIPCHK(c, hipSetDevice(c->dev));
  size_t nb = 0;
  int rc;
  if ((rc = sse_launch(c, &nb))) return rc;
  const size_t need = nb + (size_t)c->N * c->K;
  if (need > c->pin_elems) {
    if (c->pin) (void)hipHostFree(c->pin);
    c->pin = nullptr;
    HIPCHK(c, hipHostMalloc((void**)&c->pin, need * sizeof(double), hipHostMallocDefault));
    c->pin_elems = need;
  }
  if (nb) HIPCHK(c, hipMemcpyAsync(c->pin, c->bsum, nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->pin + nb, c->W, (size_t)c->N * c->K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  c->sse_nb = nb;
  c->sse_pending = true;
  return BTF_OK;
}

// ------------------------------------------------------- Negative-Binomial counts
namespace {
size_t nb_rate_elems(const btf_ctx* c, const int32_t* sh) {
  return (size_t)(sh[0] ? 1 : c->N) * (sh[1] ? 1 : c->M) * (sh[2] ? 1 : c->T);
}
void nb_strides(const btf_ctx* c, const int32_t* sh, long long* sr) {   // R is C-contiguous over its unshared dims
  const long long e1 = sh[1] ? 1 : c->M, e2 = sh[2] ? 1 : c->T;
  sr[0] = sh[0] ? 0 : e1 * e2;
  sr[1] = sh[1] ? 0 : e2;
  sr[2] = sh[2] ? 0 : 1;
}
int nb_upload_rate(btf_ctx* c, const double* R, const double* cand, const int32_t* sh) {
  const size_t n = nb_rate_elems(c, sh);
  int rc;
  if (n > c->nb_relems) {
    if ((rc = dev_alloc(c, &c->nb_R, n))) return rc;
    if ((rc = dev_alloc(c, &c->nb_C, n))) return rc;
    c->nb_relems = n;
  }
  HIPCHK(c, hipMemcpyAsync(c->nb_R, R, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (cand) HIPCHK(c, hipMemcpyAsync(c->nb_C, cand, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return BTF_OK;
}
}  // namespace

int btf_set_data_counts(btf_ctx* c, const double* counts, int nreps) {
  if (!c || !counts || nreps < 1) return fail(c, BTF_EINVAL, "bad data arguments");
  // Sharded contexts take the WHOLE count tensor too (8.6 GB at C5 against 288 GB of HBM): the rate update needs all of it
  // and every rank computes it, like the other hyper-parameters; the augmented Binomial model - pseudo-data, trial counts,
  // Polya-Gamma weights - is kept for the rank's two slabs only (nb_trials_kernel).
  HIPCHK(c, hipSetDevice(c->dev));
  const int MT = c->M * c->T;
  const size_t cells = (size_t)c->N * MT;
  c->R = 1; c->binomial = true; c->counts = true; c->nb_Rr = nreps; c->nb_bwt_written = false;
  c->pg_has_small = c->pg_has_big = c->pg_has_frac = true;      // pseudo-trial counts change with the rate: every pass
  if (c->C8_wT) { (void)hipFree(c->C8_wT); c->C8_wT = nullptr; }
  if (c->C8_v) { (void)hipFree(c->C8_v); c->C8_v = nullptr; }
  c->ldw = round_up(std::max(slab_rows(c), 1), ACC_TILE);
  c->ldv = round_up(std::max(slab_cols(c) * c->T, 1), ACC_TILE);
  int rc;
  if ((rc = dev_alloc(c, &c->nb_data, cells * nreps))) return rc;
  if ((rc = dev_alloc(c, &c->nb_S, cells))) return rc;
  if ((rc = dev_alloc(c, &c->nb_cnt, cells))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->nb_data, counts, cells * nreps * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(nb_stats_kernel, dim3((unsigned)std::min<size_t>(4096, (cells + 255) / 256)), dim3(256), 0, c->stream,
                     (const double*)c->nb_data, nreps, cells, c->nb_S, c->nb_cnt);
  HIPCHK(c, hipGetLastError());
  const size_t ew = (size_t)MT * c->ldw, ev = (size_t)c->N * c->ldv;
  for (double** p : {&c->A_wT, &c->C_wT, &c->B_wT}) {
    if ((rc = dev_alloc(c, p, ew))) return rc;
    HIPCHK(c, hipMemsetAsync(*p, 0, ew * sizeof(double), c->stream));
  }
  for (double** p : {&c->A_v, &c->C_v, &c->B_v}) {
    if ((rc = dev_alloc(c, p, ev))) return rc;
    HIPCHK(c, hipMemsetAsync(*p, 0, ev * sizeof(double), c->stream));
  }
  // per-row histograms of the counts (for rates shared along (j,t): see nb_hist_loglik_kernel)
  {
    const size_t hn = (size_t)c->N * NB_TAB;
    if ((rc = dev_alloc(c, &c->nb_H, hn))) return rc;
    if ((rc = dev_alloc(c, &c->nb_Hd, hn))) return rc;
    if ((rc = dev_alloc(c, &c->nb_Hs, (size_t)NB_TAB))) return rc;
    if ((rc = dev_alloc(c, &c->nb_L, (size_t)c->N + 1))) return rc;
    HIPCHK(c, hipMemsetAsync(c->nb_H, 0, hn * sizeof(unsigned int), c->stream));
    if ((rc = dev_alloc(c, &c->nb_optr, (size_t)c->N + 1))) return rc;
    HIPCHK(c, hipMemsetAsync(c->nb_optr, 0, ((size_t)c->N + 1) * sizeof(int), c->stream));
    const int hbx = std::max(1, std::min((int)(((size_t)MT * nreps + 255) / 256), std::max(1, 4096 / c->N)));
    hipLaunchKernelGGL(nb_hist_kernel, dim3(hbx, c->N), dim3(256), 0, c->stream, (const double*)c->nb_data, nreps, MT, c->nb_H,
                       c->nb_optr);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(u32_to_f64_kernel, dim3((unsigned)((hn + 255) / 256)), dim3(256), 0, c->stream,
                       (const unsigned int*)c->nb_H, c->nb_Hd, hn);
    hipLaunchKernelGGL(nb_reduce_kernel, dim3(NB_TAB), dim3(256), 0, c->stream, (const double*)c->nb_Hd, c->N, NB_TAB, 1, 1, 0, 1, c->nb_Hs);
    HIPCHK(c, hipGetLastError());
    // outlier counts per row -> CSR row pointers (exclusive scan on the host: N is small)
    std::vector<int> cntv((size_t)c->N + 1, 0);
    HIPCHK(c, hipMemcpyAsync(cntv.data(), c->nb_optr, (size_t)c->N * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    long long tot = 0;
    for (int i = 0; i <= c->N; ++i) { const int k = cntv[i]; cntv[i] = (int)std::min<long long>(tot, 0x7fffffff); tot += k; }
    // few outliers: per-row lists summed term by term; many: the histograms buy nothing, keep the full kernel
    bool bad = tot > (long long)(cells * nreps / 8) || tot > 0x3fffffff;
    c->nb_nout = bad ? 0 : (int)tot;
    if (!bad && tot > 0) {
      if ((rc = dev_alloc(c, &c->nb_oval, (size_t)tot))) return rc;
      HIPCHK(c, hipMemcpyAsync(c->nb_optr, cntv.data(), ((size_t)c->N + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(nb_outlier_fill_kernel, dim3(c->N), dim3(256), 0, c->stream, (const double*)c->nb_data, nreps, MT,
                         (const int*)c->nb_optr, c->nb_oval);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipStreamSynchronize(c->stream));      // cntv is a host temporary
    }
    c->nb_tabulable = !bad;
    c->nb_L_valid = false;
    {
      std::vector<double> hs((size_t)NB_TAB, 0.0);
      HIPCHK(c, hipMemcpyAsync(hs.data(), c->nb_Hs, (size_t)NB_TAB * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      c->nb_ymax = 0;
      for (int y = 0; y < NB_TAB; ++y) if (hs[y] != 0.0) c->nb_ymax = y;
      std::vector<double> gs((size_t)NB_TAB, 0.0);
      double run = 0.0;
      for (int k = NB_TAB - 2; k >= 0; --k) { run += hs[k + 1]; gs[k] = run; }      // exact: integer counts < 2^53
      if ((rc = dev_alloc(c, &c->nb_G, (size_t)NB_TAB))) return rc;
      HIPCHK(c, hipMemcpy(c->nb_G, gs.data(), (size_t)NB_TAB * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->weighted = true;
  c->have_data = true;
  c->ssw = 0.0; c->nobs = 0.0;
  return BTF_OK;
}

// The trial counts in the transposed layout: sharded contexts draw their Polya-Gamma weights from them every sweep;
// an unsharded one draws from the other layout and only ever looks at their zero pattern (btf_set_omega's mask), which
// the first rebuild has written - later rebuilds skip the 8 B per cell.
static double* nb_bwt_target(btf_ctx* c) {
  const bool whole = c->nl == c->N && c->ml == c->M;
  if (whole && c->nb_bwt_written) return nullptr;
  c->nb_bwt_written = true;
  return c->B_wT;
}

int btf_nb_loglik(btf_ctx* c, const double* R, const double* cand, const int32_t* shared, double* ll) {
  if (!c || !R || !cand || !shared || !ll) return BTF_EINVAL;
  if (!c->counts || !c->have_W || !c->have_V) return fail(c, BTF_ESTATE, "btf_nb_loglik needs count data, W and V");
  HIPCHK(c, hipSetDevice(c->dev));
  int rc;
  if ((rc = nb_upload_rate(c, R, cand, shared))) return rc;
  const int MT = c->M * c->T;
  const size_t nR = nb_rate_elems(c, shared);
  if (shared[1] && shared[2] && c->nb_tabulable && c->nb_hist) {   // rate constant along (j,t), integer counts: histogram form
    if (!c->nb_L_valid) {
      const int lbx = std::max(1, std::min((MT + 255) / 256, std::max(1, 4096 / c->N)));
      if ((size_t)c->N * lbx > c->nb_tmp_elems) { if ((rc = dev_alloc(c, &c->nb_tmp, (size_t)c->N * lbx))) return rc; c->nb_tmp_elems = (size_t)c->N * lbx; }
      Prof p(c, BTF_K_NB);
      K_SWITCH(c->K, p.launch(nb_l1p_kernel<KT>, dim3(lbx, c->N), dim3(256), 0, (const double*)c->nb_cnt, (const double*)c->W,
                              (const double*)c->V, MT, c->nb_tmp));
      hipLaunchKernelGGL(nb_reduce_kernel, dim3(c->N), dim3(256), 0, c->stream, (const double*)c->nb_tmp, c->N, lbx, 1, 0, 1, 1, c->nb_L);
      hipLaunchKernelGGL(nb_reduce_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->nb_L, c->N, 1, 1, 1, 1, 1, c->nb_L + c->N);
      HIPCHK(c, hipGetLastError());
      c->nb_L_valid = true;
    }
    if (nR > c->nb_out_elems) { if ((rc = dev_alloc(c, &c->nb_out, nR))) return rc; c->nb_out_elems = nR; }
    {
      Prof p(c, BTF_K_NB);
      const int* optr = c->nb_nout > 0 ? c->nb_optr : nullptr;
      if ((size_t)c->N > c->nb_tmp_elems) { if ((rc = dev_alloc(c, &c->nb_tmp, (size_t)c->N))) return rc; c->nb_tmp_elems = (size_t)c->N; }
      p.launch(nb_hist_loglik_kernel, dim3(c->N), dim3(256), 0, (const double*)c->nb_Hd, (const double*)c->nb_L,
               (const double*)c->nb_R, (const double*)c->nb_C, optr, (const double*)c->nb_oval, shared[0] ? 0 : 1,
               shared[0] ? c->nb_tmp : c->nb_out);
      if (shared[0])
        hipLaunchKernelGGL(nb_reduce_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->nb_tmp, c->N, 1, 1, 1, 1, 1, c->nb_out);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(ll, c->nb_out, nR * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BTF_OK;
  }
  long long sr[3];
  nb_strides(c, shared, sr);
  // a few workgroups per row: enough to fill the chip at N >= 64, each amortising its table over many cells
  const int nbx = std::max(1, std::min((MT + 255) / 256, std::max(1, 4096 / std::max(c->N, 1))));
  const bool shared_jt = shared[1] && shared[2];
  const size_t tmp_need = shared_jt ? (size_t)c->N * nbx : (size_t)c->N * MT;
  const bool direct = !shared[0] && !shared[1] && !shared[2];      // nothing to reduce: per-cell terms are the answer
  if (tmp_need > c->nb_tmp_elems) { if ((rc = dev_alloc(c, &c->nb_tmp, tmp_need))) return rc; c->nb_tmp_elems = tmp_need; }
  if (nR > c->nb_out_elems) { if ((rc = dev_alloc(c, &c->nb_out, nR))) return rc; c->nb_out_elems = nR; }
  {
    Prof p(c, BTF_K_NB);
#define NB_LAUNCH(RRT)                                                                                              \
    K_SWITCH(c->K, p.launch(nb_loglik_kernel<KT, RRT>, dim3(nbx, c->N), dim3(256), 0, (const double*)c->nb_data,        \
                            c->nb_Rr, (const double*)c->W, (const double*)c->V, MT, c->T, (const double*)c->nb_R,      \
                            (const double*)c->nb_C, sr[0], sr[1], sr[2], shared_jt ? 1 : 0, direct ? c->nb_out : c->nb_tmp))
    switch (c->nb_Rr) {
      case 1: NB_LAUNCH(1); break;
      case 2: NB_LAUNCH(2); break;
      case 3: NB_LAUNCH(3); break;
      case 4: NB_LAUNCH(4); break;
      default: NB_LAUNCH(0); break;
    }
#undef NB_LAUNCH
  }
  HIPCHK(c, hipGetLastError());
  if (!direct) {
    Prof p(c, BTF_K_PROD);
    if (shared_jt) p.launch(nb_reduce_kernel, dim3((unsigned)nR), dim3(256), 0, (const double*)c->nb_tmp, c->N, nbx, 1, (int)shared[0], 1, 1, c->nb_out);
    else p.launch(nb_reduce_kernel, dim3((unsigned)nR), dim3(256), 0, (const double*)c->nb_tmp, c->N, c->M, c->T, (int)shared[0], (int)shared[1], (int)shared[2], c->nb_out);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipMemcpyAsync(ll, c->nb_out, nR * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return BTF_OK;
}

int btf_nb_set_rate(btf_ctx* c, const double* R, const int32_t* shared) {
  if (!c || !R || !shared) return BTF_EINVAL;
  c->w_part_valid = false; c->w_local_done = c->v_local_done = false; c->sse_cols_valid = false;
  if (!c->counts) return fail(c, BTF_ESTATE, "btf_nb_set_rate follows btf_set_data_counts");
  HIPCHK(c, hipSetDevice(c->dev));
  int rc;
  if ((rc = nb_upload_rate(c, R, nullptr, shared))) return rc;
  long long sr[3];
  nb_strides(c, shared, sr);
  const int MT = c->M * c->T;
  {
    Prof p(c, BTF_K_STATS);
    p.launch(nb_trials_kernel, dim3((MT + 63) / 64, (c->N + 63) / 64), dim3(256), 0, (const double*)c->nb_S,
             (const double*)c->nb_cnt, (const double*)c->nb_R, sr[0], sr[1], sr[2], c->N, MT, c->T, c->ldv, c->ldw, c->A_v,
             c->B_v, c->A_wT, nb_bwt_target(c), c->row0, c->nl, c->col0 * c->T, c->ml * c->T, c->hrow, c->hcol >= 0 ? c->hcol * c->T : -1);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));      // R is a borrowed host buffer
  return BTF_OK;
}

// whole random-walk MH loop on the device (rng="device"): needs the histogram form (rate shared along
// (cols, depth), few outliers); BTF_ESTATE otherwise - the caller then drives btf_nb_loglik step by step
int btf_nb_mh(btf_ctx* c, uint64_t seed, int nsteps, double rpropstdev, double rstdev, const int32_t* shared,
              const double* R_in) {
  if (!c || !shared || nsteps < 0 || !(rpropstdev > 0.0) || !(rstdev > 0.0)) return fail(c, BTF_EINVAL, "bad MH arguments");
  if (!c->counts || !c->have_W || !c->have_V) return fail(c, BTF_ESTATE, "btf_nb_mh needs count data, W and V");
  if (!(shared[1] && shared[2] && c->nb_tabulable && c->nb_hist))
    return fail(c, BTF_ESTATE, "device MH loop needs a rate shared along (cols, depth) and tabulable counts");
  HIPCHK(c, hipSetDevice(c->dev));
  int rc;
  const size_t nR = nb_rate_elems(c, shared);
  if (R_in) { if ((rc = nb_upload_rate(c, R_in, nullptr, shared))) return rc; }
  else if (nR > c->nb_relems) return fail(c, BTF_ESTATE, "no rate on the device yet");
  const int MT = c->M * c->T;
  if (!c->nb_L_valid) {
    const int lbx = std::max(1, std::min((MT + 255) / 256, std::max(1, 4096 / c->N)));
    if ((size_t)c->N * lbx > c->nb_tmp_elems) { if ((rc = dev_alloc(c, &c->nb_tmp, (size_t)c->N * lbx))) return rc; c->nb_tmp_elems = (size_t)c->N * lbx; }
    Prof p(c, BTF_K_NB);
    K_SWITCH(c->K, p.launch(nb_l1p_kernel<KT>, dim3(lbx, c->N), dim3(256), 0, (const double*)c->nb_cnt, (const double*)c->W,
                            (const double*)c->V, MT, c->nb_tmp));
    hipLaunchKernelGGL(nb_reduce_kernel, dim3(c->N), dim3(256), 0, c->stream, (const double*)c->nb_tmp, c->N, lbx, 1, 0, 1, 1, c->nb_L);
    hipLaunchKernelGGL(nb_reduce_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->nb_L, c->N, 1, 1, 1, 1, 1, c->nb_L + c->N);
    HIPCHK(c, hipGetLastError());
    c->nb_L_valid = true;
  }
  const size_t need_out = std::max((size_t)c->N, (size_t)1024);    // per-row values, or up to 1024 partial sums of a single rate
  if (need_out > c->nb_out_elems) { if ((rc = dev_alloc(c, &c->nb_out, need_out))) return rc; c->nb_out_elems = need_out; }
  const int scalar = shared[0] ? 1 : 0;
  const int* optr = c->nb_nout > 0 ? c->nb_optr : nullptr;
  // one rate for everything (the reference's default rdims): the whole loop is one launch of one workgroup.
  // BTF_NB_MH_STEPWISE=1 (test hook) keeps the two-launches-per-step form, which every other sharing pattern uses.
  const char* stepwise_env = getenv("BTF_NB_MH_STEPWISE");
  const bool single = scalar && nR == 1 && !(stepwise_env && stepwise_env[0] == '1');
  // (the one workgroup takes the outlier terms too: beyond one per thread the per-step launches spread them better)
  const bool fused = single && c->nb_nout <= 256;
  if (fused) {
    Prof p(c, BTF_K_NB);
    p.launch(nb_mh_scalar_kernel, dim3(1), dim3(256), 0, (const double*)c->nb_G, c->nb_ymax, (const double*)c->nb_L, c->N, optr,
             (const double*)c->nb_oval, c->nb_R, c->nb_C, rpropstdev, rstdev, nsteps, (unsigned long long)seed);
  } else if (single) {
    // ... with many outliers: per step one launch of a few workgroups over the suffix-sum form and the outlier list
    // (instead of one workgroup per row, each rebuilding the same table), then the decision kernel
    const int gparts = std::max((c->nb_ymax + 255) / 256, std::min(1024, (c->nb_nout + 255) / 256));      // ~ one outlier per thread
    const int gp = std::max(1, gparts);
    const bool merged = !(stepwise_env && stepwise_env[0] == '2');      // BTF_NB_MH_STEPWISE=2 (test hook): two launches per step
    if (merged) {
      // one launch per step: the decision of the previous step rides at the top of the next partial-sum launch
      if (need_out < (size_t)2 * gp + 4) { if ((rc = dev_alloc(c, &c->nb_out, (size_t)2 * gp + 4))) return rc; c->nb_out_elems = (size_t)2 * gp + 4; }
      double* pa = c->nb_out, *pb = c->nb_out + gp, *rcb = c->nb_out + 2 * gp;
      for (int l = 0; l <= nsteps; ++l) {
        Prof p(c, BTF_K_NB);
        p.launch(nb_scalar_step_kernel, dim3(gp), dim3(256), 0, (const double*)c->nb_G, c->nb_ymax, (const double*)(c->nb_L + c->N),
                 (const double*)c->nb_oval, c->nb_nout, c->nb_R, c->nb_C, rcb, (const double*)((l & 1) ? pa : pb), (l & 1) ? pb : pa,
                 rpropstdev, rstdev, l, nsteps, (unsigned long long)seed);
      }
    } else {
    hipLaunchKernelGGL(nb_mh_step_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->nb_out, gp, 1, 1, c->nb_R,
                       c->nb_C, rpropstdev, rstdev, -1, (unsigned long long)seed);
    for (int sidx = 0; sidx < nsteps; ++sidx) {
      {
        Prof p(c, BTF_K_NB);
        p.launch(nb_scalar_part_kernel, dim3(gp), dim3(256), 0, (const double*)c->nb_G, c->nb_ymax, (const double*)(c->nb_L + c->N),
                 (const double*)c->nb_oval, c->nb_nout, (const double*)c->nb_R, (const double*)c->nb_C, c->nb_out);
      }
      hipLaunchKernelGGL(nb_mh_step_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->nb_out, gp, 1, 1, c->nb_R,
                         c->nb_C, rpropstdev, rstdev, sidx, (unsigned long long)seed);
    }
    }
  }
  // one rate per row: the rows' chains are independent - the whole loop in one launch, a workgroup per row
  const bool rows_fused = !single && !scalar && nR == (size_t)c->N && !(stepwise_env && stepwise_env[0] == '1');
  if (rows_fused) {
    Prof p(c, BTF_K_NB);
    p.launch(nb_mh_rows_kernel, dim3(c->N), dim3(256), 0, (const double*)c->nb_Hd, (const double*)c->nb_L, optr, (const double*)c->nb_oval,
             c->nb_R, c->nb_C, rpropstdev, rstdev, nsteps, (unsigned long long)seed, std::min(c->nb_ymax + 1, (int)NB_TAB));
  } else if (!single)
  hipLaunchKernelGGL(nb_mh_step_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->nb_out, c->N, (int)nR, scalar, c->nb_R,
                     c->nb_C, rpropstdev, rstdev, -1, (unsigned long long)seed);
  for (int sidx = 0; !single && !rows_fused && sidx < nsteps; ++sidx) {
    {
      Prof p(c, BTF_K_NB);
      p.launch(nb_hist_loglik_kernel, dim3(c->N), dim3(256), 0, (const double*)c->nb_Hd, (const double*)c->nb_L,
               (const double*)c->nb_R, (const double*)c->nb_C, optr, (const double*)c->nb_oval, scalar ? 0 : 1, c->nb_out);
    }
    hipLaunchKernelGGL(nb_mh_step_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->nb_out, c->N, (int)nR, scalar, c->nb_R,
                       c->nb_C, rpropstdev, rstdev, sidx, (unsigned long long)seed);
  }
  HIPCHK(c, hipGetLastError());
  // Binomial pseudo-data for the new rate
  long long sr[3];
  nb_strides(c, shared, sr);
  {
    Prof p(c, BTF_K_STATS);
    p.launch(nb_trials_kernel, dim3((MT + 63) / 64, (c->N + 63) / 64), dim3(256), 0, (const double*)c->nb_S,
             (const double*)c->nb_cnt, (const double*)c->nb_R, sr[0], sr[1], sr[2], c->N, MT, c->T, c->ldv, c->ldw, c->A_v,
             c->B_v, c->A_wT, nb_bwt_target(c), c->row0, c->nl, c->col0 * c->T, c->ml * c->T, c->hrow, c->hcol >= 0 ? c->hcol * c->T : -1);
  }
  HIPCHK(c, hipGetLastError());
  if (R_in) HIPCHK(c, hipStreamSynchronize(c->stream));      // borrowed host buffer
  return BTF_OK;
}

int btf_nb_get_rate(btf_ctx* c, double* R, const int32_t* shared) {
  if (!c || !R || !shared) return BTF_EINVAL;
  const size_t nR = nb_rate_elems(c, shared);
  if (!c->nb_R || nR > c->nb_relems) return fail(c, BTF_ESTATE, "no rate on the device yet");
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipMemcpyAsync(R, c->nb_R, nR * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return BTF_OK;
}

// ------------------------------------------------------------ on-device sample collection
int btf_collect_begin(btf_ctx* c, int nsamples) {
  if (!c || nsamples < 1) return fail(c, BTF_EINVAL, "bad sample count");
  if (c->nl != c->N || c->ml != c->M) return fail(c, BTF_ESTATE, "on-device sample collection needs an unsharded context");
  HIPCHK(c, hipSetDevice(c->dev));
  int rc;
  if ((rc = dev_alloc(c, &c->smp_W, (size_t)nsamples * c->N * c->K))) return rc;
  if ((rc = dev_alloc(c, &c->smp_V, (size_t)nsamples * c->M * c->T * c->K))) return rc;
  if ((rc = dev_alloc(c, &c->smp_T, (size_t)nsamples * c->M * c->nD))) return rc;
  if ((rc = dev_alloc(c, &c->smp_s, (size_t)nsamples * HYP_COUNT))) return rc;
  c->smp_n = nsamples;
  c->col_every = 0; c->col_slot = 0; c->col_count = 0;     // (a schedule left armed by an interrupted run ends here)
  return BTF_OK;
}

int btf_collect(btf_ctx* c, int slot) {
  if (!c || slot < 0 || slot >= c->smp_n) return fail(c, BTF_EINVAL, "sample slot out of range");
  if (!c->have_W || !c->have_V || !c->have_hyper) return fail(c, BTF_ESTATE, "nothing to collect yet");
  HIPCHK(c, hipSetDevice(c->dev));
  const size_t nW = (size_t)c->N * c->K, nV = (size_t)c->M * c->T * c->K, nT = (size_t)c->M * c->nD;
  const size_t tot = nW + nV + nT + (c->hyp ? HYP_COUNT : 0);
  const unsigned blocks = (unsigned)std::min<size_t>(2048, (tot + 255) / 256);
  hipLaunchKernelGGL(collect_kernel, dim3(blocks), dim3(256), 0, c->stream, (const double*)c->W, nW, (const double*)c->V, nV,
                     (const double*)c->Tau2, nT, (const double*)c->hyp, c->hyp ? (int)HYP_COUNT : 0, c->smp_W + slot * nW,
                     c->smp_V + slot * nV, c->smp_T + slot * nT, c->smp_s + (size_t)slot * HYP_COUNT);
  HIPCHK(c, hipGetLastError());
  return BTF_OK;
}

int btf_collect_schedule(btf_ctx* c, int every, int first_slot, int countdown) {
  if (!c || every < 0 || first_slot < 0 || countdown < 0 || (every > 0 && (countdown < 1 || first_slot >= c->smp_n)))
    return fail(c, BTF_EINVAL, "bad collection schedule");
  c->col_every = every; c->col_slot = first_slot; c->col_count = countdown;
  return BTF_OK;
}

int btf_collect_end(btf_ctx* c, int nsamples, double* W, double* V, double* Tau2, double* scalars) {
  if (!c || nsamples < 1 || nsamples > c->smp_n) return fail(c, BTF_EINVAL, "bad sample count");
  HIPCHK(c, hipSetDevice(c->dev));
  const size_t nW = (size_t)c->N * c->K, nV = (size_t)c->M * c->T * c->K, nT = (size_t)c->M * c->nD;
  if (W) HIPCHK(c, hipMemcpyAsync(W, c->smp_W, nsamples * nW * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (V) HIPCHK(c, hipMemcpyAsync(V, c->smp_V, nsamples * nV * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (Tau2) HIPCHK(c, hipMemcpyAsync(Tau2, c->smp_T, nsamples * nT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (scalars) HIPCHK(c, hipMemcpyAsync(scalars, c->smp_s, (size_t)nsamples * HYP_COUNT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  return check_status(c);
}

// posterior summaries straight from the collected samples (no upload); see btf_posterior_summary
int btf_collect_summary(btf_ctx* c, int nsamples, int transform, const double* q, int nq, double* mean_out, double* q_out) {
  if (!c || nsamples < 1 || nsamples > c->smp_n || nsamples > 16384 || !mean_out || nq < 0 || (nq > 0 && (!q || !q_out)) ||
      transform < 0 || transform > 2)
    return fail(c, BTF_EINVAL, "bad collect_summary arguments");
  HIPCHK(c, hipSetDevice(c->dev));
  const int MT = c->M * c->T;
  const size_t cellsN = (size_t)c->N * MT;
  double *dq = nullptr, *dm = nullptr, *dqo = nullptr;
  int rc;
  if ((rc = dev_alloc(c, &dm, cellsN))) return rc;
  if ((rc = dev_alloc(c, &dq, (size_t)std::max(nq, 1)))) { (void)hipFree(dm); return rc; }
  if ((rc = dev_alloc(c, &dqo, std::max<size_t>(1, (size_t)nq * cellsN)))) { (void)hipFree(dm); (void)hipFree(dq); return rc; }
  auto cleanup = [&]() { (void)hipFree(dm); (void)hipFree(dq); (void)hipFree(dqo); };
#define CS(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return fail(c, BTF_EHIP, std::string(#call) + ": " + hipGetErrorString(e__)); } } while (0)
  if (nq) CS(hipMemcpyAsync(dq, q, (size_t)nq * sizeof(double), hipMemcpyHostToDevice, c->stream));
  int P = 2;
  while (P < nsamples) P <<= 1;
  const int cells = std::max(1, std::min(16, (int)((128 * 1024) / ((size_t)P * sizeof(double)))));
  const size_t lds = (size_t)cells * P * sizeof(double);
  dim3 grid((MT + cells - 1) / cells, c->N);
#define CS_LAUNCH(KT_)                                                                                           \
  case KT_: {                                                                                                    \
    CS(hipFuncSetAttribute((const void*)posterior_summary_kernel<KT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(posterior_summary_kernel<KT_>, grid, dim3(256), lds, c->stream, (const double*)c->smp_W,  \
                       (const double*)c->smp_V, nsamples, c->N, MT, P, cells, transform, (const double*)dq, nq, dm, dqo); \
  } break;
  switch (c->K) {
    CS_LAUNCH(1) CS_LAUNCH(2) CS_LAUNCH(3) CS_LAUNCH(4) CS_LAUNCH(5) CS_LAUNCH(6) CS_LAUNCH(7) CS_LAUNCH(8) CS_LAUNCH(9) CS_LAUNCH(10)
    default: break;
  }
#undef CS_LAUNCH
  CS(hipGetLastError());
  CS(hipMemcpyAsync(mean_out, dm, cellsN * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (nq) CS(hipMemcpyAsync(q_out, dqo, (size_t)nq * cellsN * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  CS(hipStreamSynchronize(c->stream));
#undef CS
  cleanup();
  return BTF_OK;
}

// ---------------------------------------------------------- device-resident scalars
int btf_device_scalars(btf_ctx* c, int enable) {
  if (!c) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (enable) {
    int rc;
    if ((rc = ensure_hyp(c))) return rc;
  }
  c->dev_scalars = enable != 0;
  ++c->prior_version;
  return BTF_OK;
}

int btf_set_scalars(btf_ctx* c, double nu2, double sigma2, double lam2, double lam2_a) {
  if (!c || !(sigma2 > 0.0) || !(lam2 > 0.0) || !(nu2 > 0.0) || !(lam2_a > 0.0)) return fail(c, BTF_EINVAL, "scalars must be positive");
  HIPCHK(c, hipSetDevice(c->dev));
  int rc;
  if ((rc = ensure_hyp(c))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));      // the staging word may still be in flight
  c->nu2 = nu2; c->sigma2 = sigma2; c->lam2 = lam2;
  ++c->prior_version;
  c->pin_hyp[HYP_NU2] = nu2; c->pin_hyp[HYP_SIGMA2] = sigma2; c->pin_hyp[HYP_LAM2] = lam2; c->pin_hyp[HYP_LAM2A] = lam2_a;
  HIPCHK(c, hipMemcpyAsync(c->hyp, c->pin_hyp, 4 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return BTF_OK;
}

int btf_get_scalars(btf_ctx* c, double* out) {
  if (!c || !out) return BTF_EINVAL;
  if (!c->hyp) return fail(c, BTF_ESTATE, "no device-resident scalars yet");
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipMemcpyAsync(c->pin_hyp, c->hyp, HYP_COUNT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < 6; ++i) out[i] = c->pin_hyp[i];
  // a read does not change the prior: with device-resident scalars the band is built from hyp[] itself (prior_band_kernel
  // reads lam2 there), so only a host copy that actually moved invalidates what was precomputed from it
  if (out[HYP_LAM2] != c->lam2 && !c->dev_scalars) ++c->prior_version;
  c->nu2 = out[HYP_NU2]; c->sigma2 = out[HYP_SIGMA2]; c->lam2 = out[HYP_LAM2];
  return BTF_OK;
}

void* btf_dev_hyp(btf_ctx* c) { return c ? (void*)c->hyp : nullptr; }

int btf_set_scalar_slot(btf_ctx* c, int slot, double value) {
  if (!c || slot < 0 || slot >= HYP_COUNT) return BTF_EINVAL;
  if (!c->hyp) return fail(c, BTF_ESTATE, "no device-resident scalars yet");
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->pin_hyp[slot] = value;
  HIPCHK(c, hipMemcpyAsync(c->hyp + slot, c->pin_hyp + slot, sizeof(double), hipMemcpyHostToDevice, c->stream));
  return BTF_OK;
}

int btf_set_global_nobs(btf_ctx* c, double nobs) {
  if (!c || !(nobs >= 0.0)) return BTF_EINVAL;
  c->nobs_global = nobs;
  return BTF_OK;
}

int btf_draw_scalars(btf_ctx* c, uint64_t seed, int which, double nu2_a, double nu2_b, double sigma2_a, double sigma2_b) {
  if (!c || !c->dev_scalars) return fail(c, BTF_ESTATE, "enable device-resident scalars first");
  // which & 8 / & 16: the two halves of a sharded nu2 draw (see scalars_kernel): 8 = reduce this rank's residual sum
  // of squares into the device scalar HYP_SSE and stop; 16 = draw from HYP_SSE (all-reduced by the caller in between)
  const int phase = (which & 8) ? 1 : ((which & 16) ? 2 : 0);
  if ((which & 1) && (!c->have_data || c->binomial)) return fail(c, BTF_ESTATE, "the scalar nu2 draw needs Gaussian data");
  if (which & 1) c->nu2_drawn_since_v = true;
  if (!c->have_W || ((which & 1) && !c->have_V)) return fail(c, BTF_ESTATE, "set W and V first");
  HIPCHK(c, hipSetDevice(c->dev));
  size_t nb = 0;
  int rc;
  double ssw = c->ssw;
  if (which & 1) {
    // which & 4: take the residual sum of squares from the W half-sweep's accumulation partials (btf_w_accum
    // must have run for the current V; the stale-weight mode of compat="reference" has no such identity)
    const bool from_part = phase != 2 && (which & 4) && c->w_part_valid && c->w_part_mode != 2;
    if (from_part) {
      const int blocks = (c->nl + WS_ROWS - 1) / WS_ROWS;
      if ((size_t)blocks > c->bsum_elems) { if ((rc = dev_alloc(c, &c->bsum, (size_t)blocks))) return rc; c->bsum_elems = (size_t)blocks; }
      Prof p(c, BTF_K_SSE);
      const double* gp = c->w_part_gv ? c->gpart_v : c->gpart;
      const int ngp = c->w_part_gv ? c->ngp_v : c->ngp_gram;
      const CurveLists cvl = c->w_part_curve ? CurveLists{c->cv_rptr, c->cv_rcol, c->cv_rdef} : CurveLists{nullptr, nullptr, nullptr};
      if (c->w_part_mode != 0) {
        K_SWITCH(c->K, p.launch(sse_part_kernel<KT, true>, dim3(blocks), dim3(WS_ROWS * ws_split(KT)), 0, (const double*)c->part,
                                c->w_part_nch, c->ldw, gp, ngp, (double)c->R, (const double*)c->W, c->row0, c->nl, c->bsum, cvl,
                                (const double*)c->gpart_v));
      } else {
        K_SWITCH(c->K, p.launch(sse_part_kernel<KT, false>, dim3(blocks), dim3(WS_ROWS * ws_split(KT)), 0, (const double*)c->part,
                                c->w_part_nch, c->ldw, gp, ngp, (double)c->R, (const double*)c->W, c->row0, c->nl, c->bsum, cvl,
                                (const double*)c->gpart_v));
      }
      HIPCHK(c, hipGetLastError());
      nb = (size_t)blocks;
      ssw += c->sa2;
    } else if (phase != 2 && (rc = sse_launch(c, &nb))) {
      return rc;
    }
  }
  const int h = std::min(c->K, c->N);
  const double nfree = (double)c->N * c->K - (double)h * (h - 1) / 2.0 - (double)(c->K - h) * c->N;   // factor.py:155-174
  {
    const bool with_lam = c->lam_pending && phase != 1 && c->lsum && c->have_chain;     // (not with the reduce-only half)
    Prof p(c, BTF_K_HYPER);
    p.launch(scalars_kernel, dim3(with_lam ? 2 : 1), dim3(256), 0, (const double*)c->bsum, (int)nb, ssw,
             c->nobs_global >= 0.0 ? c->nobs_global : c->nobs, (const double*)c->W,
             c->N, c->K, nfree, nu2_a, nu2_b, sigma2_a, sigma2_b, which & 3, (unsigned long long)seed, c->hyp, phase,
             (const double*)c->lsum, c->M, (double)c->nD * c->M * c->K + 1.0, c->lam_exact, c->lam_seed);
    if (with_lam) { c->lam_pending = false; ++c->prior_version; }
  }
  HIPCHK(c, hipGetLastError());
  return BTF_OK;
}

namespace {
// can the next W accumulation launch carry nu2 | rest / sigma2 | rest as a side workgroup?
bool scalars_can_ride(const btf_ctx* c, int which) {
  if (!c->fused_sweep || !c->dev_scalars || c->binomial || c->counts || !c->have_data || !c->have_W || !c->have_V) return false;
  if (c->nl != c->N || c->ml != c->M || c->nl < 1) return false;
  if (c->w_part_valid) return false;                      // the accumulation has already been launched
  if (lik_weighted(c) || c->weighted) return false;       // complete data: the residual identity of the spectral sampler
  if ((which & 1) && !(c->sse_cols && c->sse_cols_valid)) return false;
  return (which & 3) != 0;
}
}  // namespace

int btf_queue_scalars(btf_ctx* c, uint64_t seed, int which, double nu2_a, double nu2_b, double sigma2_a, double sigma2_b, int32_t* queued) {
  if (!c || !queued) return BTF_EINVAL;
  *queued = 0;
  if (!c->dev_scalars) return fail(c, BTF_ESTATE, "enable device-resident scalars first");
  if (!scalars_can_ride(c, which)) return BTF_OK;         // the caller draws them with btf_draw_scalars instead
  c->sc_pending = true; c->sc_seed = seed; c->sc_which = which & 3;
  if (which & 1) c->nu2_drawn_since_v = true;
  c->sc_prior[0] = nu2_a; c->sc_prior[1] = nu2_b; c->sc_prior[2] = sigma2_a; c->sc_prior[3] = sigma2_b;
  *queued = 1;
  return BTF_OK;
}

int btf_queue_lam2(btf_ctx* c, uint64_t seed, int compat) {
  if (!c || !c->dev_scalars) return fail(c, BTF_ESTATE, "enable device-resident scalars first");
  if (!c->have_chain) return fail(c, BTF_ESTATE, "set the horseshoe+ chain first");
  c->lam_pending = true; c->lam_seed = seed; c->lam_exact = compat == BTF_COMPAT_EXACT ? 1 : 0;
  return BTF_OK;
}

int btf_draw_lam2(btf_ctx* c, uint64_t seed, int compat) {
  if (!c || !c->dev_scalars) return fail(c, BTF_ESTATE, "enable device-resident scalars first");
  if (!c->lsum || !c->have_chain) return fail(c, BTF_ESTATE, "btf_draw_lam2 follows btf_resample_Tau2");
  HIPCHK(c, hipSetDevice(c->dev));
  const double shape = (double)c->nD * c->M * c->K + 1.0;
  c->lam_pending = false;
  ++c->prior_version;
  {
    Prof p(c, BTF_K_HYPER);
    p.launch(lam2_kernel, dim3(1), dim3(256), 0, (const double*)c->lsum, c->M, shape, compat == BTF_COMPAT_EXACT ? 1 : 0,
             (unsigned long long)seed, c->hyp);
  }
  HIPCHK(c, hipGetLastError());
  return BTF_OK;
}

int btf_sse_end(btf_ctx* c, double* sse, double* nobs, double* W_out) {
  if (!c || !sse || !nobs) return BTF_EINVAL;
  if (!c->sse_pending) return fail(c, BTF_ESTATE, "btf_sse_end without btf_sse_begin");
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->sse_pending = false;
  double s = 0.0;
  for (size_t b = 0; b < c->sse_nb; ++b) s += c->pin[b];     // fixed order
  // NOTE: in a sharded run this is the column-slab share of the between-cell part; the
  // within-cell part (ssw) and nobs were reduced over the ROW slab.  The host adds the
  // two kinds over ranks (see functionalmf_amd/factor.py).
  *sse = s + c->ssw;
  *nobs = c->nobs;
  if (W_out) std::memcpy(W_out, c->pin + c->sse_nb, (size_t)c->N * c->K * sizeof(double));
  return check_status(c);
}

int btf_sse(btf_ctx* c, double* sse, double* nobs) {
  int rc = btf_sse_begin(c);
  if (rc) return rc;
  rc = btf_sse_end(c, sse, nobs, nullptr);
  return rc == BTF_ENOTPD ? rc : rc;
}

int btf_pg_draw(btf_ctx* c, uint64_t seed) {
  if (!c) return BTF_EINVAL;
  c->w_part_valid = false; c->w_local_done = c->v_local_done = false; c->sse_cols_valid = false;          // the weights change
  if (!c->have_data || !c->binomial || !c->have_W || !c->have_V) return fail(c, BTF_ESTATE, "btf_pg_draw needs binomial data, W and V");
  HIPCHK(c, hipSetDevice(c->dev));
  const unsigned long long MT = (unsigned long long)c->M * c->T;
  if (c->nl == c->N && c->ml == c->M) {   // unsharded: every cell once, both layouts (LDS tile transpose)
    dim3 grid((unsigned)((MT + 63) / 64), (unsigned)((c->N + 63) / 64));
    const PgPasses ps = pg_passes(c);
    int fill = 1;
    if (ps.flat) {
      Prof p(c, BTF_K_PG);
      constexpr int TI = PGX_NW * PGX_CPL;
      dim3 gridx((unsigned)((MT + 63) / 64), (unsigned)((c->N + TI - 1) / TI));
      K_SWITCH(c->K, p.launch(pgx_tile_kernel<KT, PGX_NW, PGX_CPL>, gridx, dim3(PGX_NW * 64), pgx_tile_lds(PGX_NW, PGX_CPL),
                              (const double*)c->B_v, c->C_v, c->C_wT, (const double*)c->W, (const double*)c->V, c->N, (int)MT,
                              c->ldv, c->ldw, (unsigned long long)seed, ps.mode, fill));
      fill = 0;
    }
    if (ps.series) {
      Prof p(c, BTF_K_PG);
      K_SWITCH(c->K, p.launch(pg_tile_kernel<KT, PG_PATH_SERIES>, grid, dim3(256), 0, (const double*)c->B_v, c->C_v, c->C_wT,
                              (const double*)c->W, (const double*)c->V, c->N, (int)MT, c->ldv, c->ldw,
                              (unsigned long long)seed, ps.mode, fill));
      fill = 0;
    }
    if (ps.frac) {
      Prof p(c, BTF_K_PG);
      K_SWITCH(c->K, p.launch(pg_tile_kernel<KT, PG_PATH_EXACT>, grid, dim3(256), 0, (const double*)c->B_v, c->C_v, c->C_wT,
                              (const double*)c->W, (const double*)c->V, c->N, (int)MT, c->ldv, c->ldw,
                              (unsigned long long)seed, ps.mode, fill));
    }
    HIPCHK(c, hipGetLastError());
    return BTF_OK;
  }
  // sharded: every cell is drawn once per layout from the same (seed, cell) stream: identical values
  if (c->nl > 0) {  // W layout [jt][i_local]: lanes = rows of W, reduction axis = (j,t)
    K_SWITCH(c->K, launch_pg<KT>(c, c->B_wT, c->C_wT, c->W + (size_t)c->row0 * c->K, c->V, c->nl, c->ldw, (int)MT,
                                 (unsigned long long)c->row0 * MT, 1ULL, MT, seed));
  }
  if (c->ml > 0) {  // V layout [i][jt_local]
    K_SWITCH(c->K, launch_pg<KT>(c, c->B_v, c->C_v, c->V + (size_t)c->col0 * c->T * c->K, c->W, c->ml * c->T, c->ldv,
                                 c->N, (unsigned long long)c->col0 * c->T, MT, 1ULL, seed));
  }
  // the halo sources (btf_set_shard_halo): the weights of ONE more row / column, from the streams of its global cells -
  // the values its owner draws
  if (c->hrow >= 0) {
    K_SWITCH(c->K, launch_pg<KT>(c, c->B_wT + c->nl, c->C_wT + c->nl, c->W + (size_t)c->hrow * c->K, c->V, 1, c->ldw, (int)MT,
                                 (unsigned long long)c->hrow * MT, 1ULL, MT, seed));
  }
  if (c->hcol >= 0) {
    K_SWITCH(c->K, launch_pg<KT>(c, c->B_v + (size_t)c->ml * c->T, c->C_v + (size_t)c->ml * c->T, c->V + (size_t)c->hcol * c->T * c->K,
                                 c->W, c->T, c->ldv, c->N, (unsigned long long)c->hcol * c->T, MT, 1ULL, seed));
  }
  HIPCHK(c, hipGetLastError());
  return BTF_OK;
}

// ---------------------------------------------------------------- posterior summaries
int btf_posterior_summary(int device, int nsamples, int nrows, int ncols, int ndepth, int nembeds, const double* Ws,
                          const double* Vs, int transform, const double* q, int nq, double* mean_out, double* q_out) {
  if (nsamples < 1 || nsamples > 16384 || nrows < 1 || ncols < 1 || ndepth < 1 || nembeds < 1 || nembeds > MAX_K || !Ws || !Vs ||
      !mean_out || nq < 0 || (nq > 0 && (!q || !q_out)) || transform < 0 || transform > 2)
    return fail(nullptr, BTF_EINVAL, "bad posterior_summary arguments");
  for (int k = 0; k < nq; ++k)
    if (!(q[k] >= 0.0 && q[k] <= 100.0)) return fail(nullptr, BTF_EINVAL, "percentiles must lie in [0, 100]");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, BTF_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  const int MT = ncols * ndepth;
  const size_t cellsN = (size_t)nrows * MT;
  double *dW = nullptr, *dV = nullptr, *dq = nullptr, *dm = nullptr, *dqo = nullptr;
  auto cleanup = [&]() { for (void* p : {(void*)dW, (void*)dV, (void*)dq, (void*)dm, (void*)dqo}) if (p) (void)hipFree(p); };
#define PS(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return fail(nullptr, BTF_EHIP, std::string(#call) + ": " + hipGetErrorString(e__)); } } while (0)
  const size_t nW = (size_t)nsamples * nrows * nembeds, nV = (size_t)nsamples * MT * nembeds;
  PS(hipMalloc((void**)&dW, nW * sizeof(double)));
  PS(hipMalloc((void**)&dV, nV * sizeof(double)));
  PS(hipMalloc((void**)&dm, cellsN * sizeof(double)));
  PS(hipMalloc((void**)&dq, (size_t)std::max(nq, 1) * sizeof(double)));
  PS(hipMalloc((void**)&dqo, std::max<size_t>(1, (size_t)nq * cellsN) * sizeof(double)));
  PS(hipMemcpy(dW, Ws, nW * sizeof(double), hipMemcpyHostToDevice));
  PS(hipMemcpy(dV, Vs, nV * sizeof(double), hipMemcpyHostToDevice));
  if (nq) PS(hipMemcpy(dq, q, (size_t)nq * sizeof(double), hipMemcpyHostToDevice));
  int P = 2;
  while (P < nsamples) P <<= 1;
  const int cells = std::max(1, std::min(16, (int)((128 * 1024) / ((size_t)P * sizeof(double)))));
  const size_t lds = (size_t)cells * P * sizeof(double);
  dim3 grid((MT + cells - 1) / cells, nrows);
#define PS_LAUNCH(KT_)                                                                                           \
  case KT_: {                                                                                                    \
    PS(hipFuncSetAttribute((const void*)posterior_summary_kernel<KT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(posterior_summary_kernel<KT_>, grid, dim3(256), lds, 0, (const double*)dW, (const double*)dV, nsamples, \
                       nrows, MT, P, cells, transform, (const double*)dq, nq, dm, dqo);                         \
  } break;
  switch (nembeds) {
    PS_LAUNCH(1) PS_LAUNCH(2) PS_LAUNCH(3) PS_LAUNCH(4) PS_LAUNCH(5) PS_LAUNCH(6) PS_LAUNCH(7) PS_LAUNCH(8) PS_LAUNCH(9) PS_LAUNCH(10)
    default: break;
  }
#undef PS_LAUNCH
  PS(hipGetLastError());
  PS(hipDeviceSynchronize());
  PS(hipMemcpy(mean_out, dm, cellsN * sizeof(double), hipMemcpyDeviceToHost));
  if (nq) PS(hipMemcpy(q_out, dqo, (size_t)nq * cellsN * sizeof(double), hipMemcpyDeviceToHost));
#undef PS
  cleanup();
  return BTF_OK;
}

int btf_pg_batch(int device, int64_t n, const double* b, const double* psi, uint64_t seed, double* out) {
  return btf_pg_batch_mode(device, n, b, psi, seed, 0, out);
}

int btf_pg_batch_mode(int device, int64_t n, const double* b, const double* psi, uint64_t seed, int mode, double* out) {
  if (n < 1 || !b || !psi || !out || mode < 0 || mode > PG_MODE_REF_F64 + 1) return fail(nullptr, BTF_EINVAL, "bad pg_batch arguments");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, BTF_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  double *db = nullptr, *dp = nullptr, *dout = nullptr;
  auto cleanup = [&]() { for (void* p : {(void*)db, (void*)dp, (void*)dout}) if (p) (void)hipFree(p); };
#define PB(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return fail(nullptr, BTF_EHIP, std::string(#call) + ": " + hipGetErrorString(e__)); } } while (0)
  PB(hipMalloc((void**)&db, n * sizeof(double)));
  PB(hipMalloc((void**)&dp, n * sizeof(double)));
  PB(hipMalloc((void**)&dout, n * sizeof(double)));
  PB(hipMemcpy(db, b, n * sizeof(double), hipMemcpyHostToDevice));
  PB(hipMemcpy(dp, psi, n * sizeof(double), hipMemcpyHostToDevice));
  const bool allf64 = mode == PG_MODE_REF_F64 + 1;      // 4: PG_MODE_EXACT_ALL with every trip of the flat sampler repeated in f64
  if (allf64) mode = PG_MODE_EXACT_ALL;
  const bool flat = mode == PG_MODE_DEFAULT || mode == PG_MODE_EXACT_ALL;
  if (flat) {      // the integer counts the mode gives to the flat exact sampler
    constexpr int CPL = 4;
    const dim3 grid((unsigned)((n + 256 * CPL - 1) / (256 * CPL)));
    if (allf64) hipLaunchKernelGGL((pgx_batch_kernel<CPL, true>), grid, dim3(256), pgx_rows_lds(CPL), 0, db, dp, dout, (long long)n, (unsigned long long)seed, mode);
    else hipLaunchKernelGGL((pgx_batch_kernel<CPL, false>), grid, dim3(256), pgx_rows_lds(CPL), 0, db, dp, dout, (long long)n, (unsigned long long)seed, mode);
    PB(hipGetLastError());
  }
  hipLaunchKernelGGL(pg_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, db, dp, dout, (long long)n, seed, mode, flat ? 1 : 0);
  PB(hipGetLastError());
  PB(hipDeviceSynchronize());
  PB(hipMemcpy(out, dout, n * sizeof(double), hipMemcpyDeviceToHost));
  cleanup();
#undef PB
  return BTF_OK;
}

int btf_sym_eig(int device, int K, int nparts, const double* parts, double* out, const double* warm_from) {
  if (K < 1 || K > EIG_MAXK || nparts < 1 || !parts || !out) return fail(nullptr, BTF_EINVAL, "bad sym_eig arguments");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, BTF_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  const size_t np = (size_t)nparts * tri(K), no = (size_t)K + K * K + 2;
  double *dp = nullptr, *dout = nullptr;
  auto cleanup = [&]() { if (dp) (void)hipFree(dp); if (dout) (void)hipFree(dout); };
#define SE(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return fail(nullptr, BTF_EHIP, std::string(#call) + ": " + hipGetErrorString(e__)); } } while (0)
  SE(hipMalloc((void**)&dp, np * sizeof(double)));
  SE(hipMalloc((void**)&dout, no * sizeof(double)));
  SE(hipMemset(dout, 0, no * sizeof(double)));
  if (warm_from) {      // a previous solution to refine: eigenvalues, vectors, (sweeps), then the "valid" word
    std::vector<double> w(no, 0.0);
    std::memcpy(w.data(), warm_from, ((size_t)K + K * K) * sizeof(double));
    w[(size_t)K + K * K + 1] = 1.0;
    SE(hipMemcpy(dout, w.data(), no * sizeof(double), hipMemcpyHostToDevice));
  }
  SE(hipMemcpy(dp, parts, np * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(gram_eig_kernel, dim3(1), dim3(WAVE), 0, 0, (const double*)dp, nparts, K, dout, warm_from ? 1 : 0);
  SE(hipGetLastError());
  SE(hipDeviceSynchronize());
  SE(hipMemcpy(out, dout, ((size_t)K + K * K + 1) * sizeof(double), hipMemcpyDeviceToHost));
#undef SE
  cleanup();
  return BTF_OK;
}

// Measurement aid (bench.py roofline.read_ceiling_GBs): the rate of a plain streaming read - one 16-byte load
// per lane and trip, 512 workgroups of 1024 threads - over `bytes` of device memory, averaged over `reps` launches.
namespace {
__global__ __launch_bounds__(1024) void read_probe_kernel(const double2* __restrict__ x, size_t n2, double* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) s += x[i].x + x[i].y;
  if (s == 1.2345e300) out[0] = s;       // never true (the buffer holds zeros): keeps the loads alive
}
}  // namespace
int btf_read_probe(int device, size_t bytes, int reps, double* gb_per_s) {
  if (bytes < 1024 || reps < 1 || !gb_per_s) return fail(nullptr, BTF_EINVAL, "bad read_probe arguments");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, BTF_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  double2* x = nullptr; double* out = nullptr;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  auto cleanup = [&]() { if (x) (void)hipFree(x); if (out) (void)hipFree(out); if (t0) (void)hipEventDestroy(t0); if (t1) (void)hipEventDestroy(t1); };
#define RP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return fail(nullptr, BTF_EHIP, std::string(#call) + ": " + hipGetErrorString(e__)); } } while (0)
  const size_t n2 = bytes / sizeof(double2);
  RP(hipMalloc((void**)&x, n2 * sizeof(double2)));
  RP(hipMalloc((void**)&out, sizeof(double)));
  RP(hipMemset(x, 0, n2 * sizeof(double2)));
  RP(hipEventCreate(&t0));
  RP(hipEventCreate(&t1));
  hipLaunchKernelGGL(read_probe_kernel, dim3(512), dim3(1024), 0, 0, (const double2*)x, n2, out);      // warm-up
  RP(hipDeviceSynchronize());
  RP(hipEventRecord(t0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(read_probe_kernel, dim3(512), dim3(1024), 0, 0, (const double2*)x, n2, out);
  RP(hipEventRecord(t1, 0));
  RP(hipEventSynchronize(t1));
  float ms = 0.f;
  RP(hipEventElapsedTime(&ms, t0, t1));
#undef RP
  *gb_per_s = (double)n2 * sizeof(double2) * reps / (ms * 1e-3) / 1e9;
  cleanup();
  return BTF_OK;
}

int btf_sync(btf_ctx* c) {
  if (!c) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  return check_status(c);
}

// ---------------------------------------------------------------- stand-alone MVN
int btf_mvn_banded(int device, int batch, int n, int bw, const double* band, const double* mu_part, const double* z,
                   uint64_t seed, double eps0, int attempts, double* x_out, int32_t* tries_out) {
  if (batch < 1 || n < 1 || bw < 0 || bw > 63 || !band || !x_out) return fail(nullptr, BTF_EINVAL, "bad mvn arguments");
  btf_ctx* c = nullptr;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, BTF_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  const int R1 = bw + 1;
  const size_t nb = (size_t)batch * n * R1, nv = (size_t)batch * n;
  double *dband = nullptr, *dmu = nullptr, *dz = nullptr, *dx = nullptr, *dwork = nullptr;
  int *dtries = nullptr, *dstatus = nullptr;
  int rc = BTF_OK;
  auto cleanup = [&]() {
    for (void* p : {(void*)dband, (void*)dmu, (void*)dz, (void*)dx, (void*)dwork, (void*)dtries, (void*)dstatus})
      if (p) (void)hipFree(p);
  };
#define MV(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return fail(nullptr, BTF_EHIP, std::string(#call) + ": " + hipGetErrorString(e__)); } } while (0)
  MV(hipMalloc((void**)&dband, nb * sizeof(double)));
  MV(hipMemcpy(dband, band, nb * sizeof(double), hipMemcpyHostToDevice));
  if (mu_part) { MV(hipMalloc((void**)&dmu, nv * sizeof(double))); MV(hipMemcpy(dmu, mu_part, nv * sizeof(double), hipMemcpyHostToDevice)); }
  if (z) { MV(hipMalloc((void**)&dz, nv * sizeof(double))); MV(hipMemcpy(dz, z, nv * sizeof(double), hipMemcpyHostToDevice)); }
  MV(hipMalloc((void**)&dx, nv * sizeof(double)));
  MV(hipMalloc((void**)&dwork, (nb + 2 * nv) * sizeof(double)));
  MV(hipMalloc((void**)&dtries, batch * sizeof(int)));
  MV(hipMalloc((void**)&dstatus, 2 * sizeof(int)));
  MV(hipMemset(dstatus, 0, 2 * sizeof(int)));
  MvnArgs a{dband, dmu, dz, dx, dwork, n, bw, seed, eps0, attempts < 0 ? 0 : attempts, dtries, dstatus};
  const size_t lds = ((size_t)(bw * (bw + 1) / 2 + 3) / 4 + 1) * sizeof(double);
  hipLaunchKernelGGL(mvn_banded_kernel, dim3(batch), dim3(WAVE), lds, 0, a);
  MV(hipGetLastError());
  MV(hipDeviceSynchronize());
  int st[2];
  MV(hipMemcpy(st, dstatus, sizeof(st), hipMemcpyDeviceToHost));
  MV(hipMemcpy(x_out, dx, nv * sizeof(double), hipMemcpyDeviceToHost));
  if (tries_out) MV(hipMemcpy(tries_out, dtries, batch * sizeof(int), hipMemcpyDeviceToHost));
  cleanup();
#undef MV
  (void)c; (void)rc;
  if (st[0]) return fail(nullptr, BTF_ENOTPD, "precision not positive definite in batch item " + std::to_string(st[1]));
  return BTF_OK;
}

int btf_mvn_dense(int device, int batch, int n, const double* A, int form, const double* mu, const double* mu_part,
                  const double* z, uint64_t seed, double eps0, int attempts, double* x_out, int32_t* tries_out) {
  if (batch < 1 || n < 1 || n > 1024 || !A || !x_out || (form & ~3) || (mu && mu_part))
    return fail(nullptr, BTF_EINVAL, "bad dense mvn arguments (1 <= n <= 1024; mu and mu_part are mutually exclusive)");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, BTF_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  const size_t nm = (size_t)batch * n * n, nv = (size_t)batch * n;
  double *dA = nullptr, *dmu = nullptr, *dmp = nullptr, *dz = nullptr, *dx = nullptr, *dwork = nullptr;
  int *dtries = nullptr, *dstatus = nullptr;
  auto cleanup = [&]() {
    for (void* p : {(void*)dA, (void*)dmu, (void*)dmp, (void*)dz, (void*)dx, (void*)dwork, (void*)dtries, (void*)dstatus})
      if (p) (void)hipFree(p);
  };
#define MD(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return fail(nullptr, BTF_EHIP, std::string(#call) + ": " + hipGetErrorString(e__)); } } while (0)
  MD(hipMalloc((void**)&dA, nm * sizeof(double)));
  MD(hipMemcpy(dA, A, nm * sizeof(double), hipMemcpyHostToDevice));
  if (mu) { MD(hipMalloc((void**)&dmu, nv * sizeof(double))); MD(hipMemcpy(dmu, mu, nv * sizeof(double), hipMemcpyHostToDevice)); }
  if (mu_part) { MD(hipMalloc((void**)&dmp, nv * sizeof(double))); MD(hipMemcpy(dmp, mu_part, nv * sizeof(double), hipMemcpyHostToDevice)); }
  if (z) { MD(hipMalloc((void**)&dz, nv * sizeof(double))); MD(hipMemcpy(dz, z, nv * sizeof(double), hipMemcpyHostToDevice)); }
  MD(hipMalloc((void**)&dx, nv * sizeof(double)));
  MD(hipMalloc((void**)&dwork, nm * sizeof(double)));
  MD(hipMalloc((void**)&dtries, batch * sizeof(int)));
  MD(hipMalloc((void**)&dstatus, 2 * sizeof(int)));
  MD(hipMemset(dstatus, 0, 2 * sizeof(int)));
  MvnDenseArgs a{dA, dmu, dmp, dz, dx, dwork, n, form, (unsigned long long)seed, eps0, attempts < 0 ? 0 : attempts, dtries, dstatus};
  hipLaunchKernelGGL(mvn_dense_kernel, dim3(batch), dim3(MVD_THREADS), 2 * (size_t)n * sizeof(double), 0, a);
  MD(hipGetLastError());
  MD(hipDeviceSynchronize());
  int st[2];
  MD(hipMemcpy(st, dstatus, sizeof(st), hipMemcpyDeviceToHost));
  MD(hipMemcpy(x_out, dx, nv * sizeof(double), hipMemcpyDeviceToHost));
  if (tries_out) MD(hipMemcpy(tries_out, dtries, batch * sizeof(int), hipMemcpyDeviceToHost));
  cleanup();
#undef MD
  if (st[0]) return fail(nullptr, BTF_ENOTPD, "matrix not positive definite in batch item " + std::to_string(st[1]));
  return BTF_OK;
}

// ------------------------------------------------------------------ measurement
int btf_set_profiling(btf_ctx* c, int on) {
  if (!c) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (on && c->ev_pool.empty()) {
    c->ev_pool.resize(MAX_EVENTS);
    for (auto& e : c->ev_pool) {
      HIPCHK(c, hipEventCreate(&e.a));
      HIPCHK(c, hipEventCreate(&e.b));
    }
  }
  c->profiling = on != 0;
  return BTF_OK;
}

int btf_kernel_times(btf_ctx* c, double* ms_total, int64_t* launches) {
  if (!c || !ms_total || !launches) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int64_t timed[BTF_K_COUNT] = {0};
  for (size_t i = 0; i < c->ev_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b) == hipSuccess) {
      c->ms_total[c->ev_pool[i].kid] += ms;
      timed[c->ev_pool[i].kid]++;
    }
  }
  for (int k = 0; k < BTF_K_COUNT; ++k) {
    ms_total[k] = c->ms_total[k];
    launches[k] = c->profiling ? timed[k] : c->launches[k];
    c->ms_total[k] = 0.0;
    c->launches[k] = 0;
  }
  c->ev_used = 0;
  return BTF_OK;
}

// diagnostic (not in btf.h): the shared eigen-system as the side task left it, K + K*K + 8 doubles
// ([K + K*K]: Jacobi sweeps of the last solve, 0 = the warm refinement converged)
extern "C" int btf_debug_eig(btf_ctx* c, double* out) {
  if (!c || !out) return BTF_EINVAL;
  if (!c->eig) return fail(c, BTF_ESTATE, "no eigen-system yet");
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out, c->eig, ((size_t)c->K + c->K * c->K + 8) * sizeof(double), hipMemcpyDeviceToHost));
  return BTF_OK;
}
#ifdef BTF_ACC_STAMPS
// diagnostic builds only: the accumulation workgroups' wall-clock stamps of the LAST launch, [8192][4]
extern "C" int btf_debug_acc_stamps(btf_ctx* c, long long* out) {
  if (!c) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (!c->acc_stamps) {
    int rc = dev_alloc(c, &c->acc_stamps, (size_t)8192 * 8);
    if (rc) return rc;
  }
  HIPCHK(c, hipMemcpy(out, c->acc_stamps, sizeof(long long) * 8192 * 8, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemset(c->acc_stamps, 0, sizeof(long long) * 8192 * 8));
  return BTF_OK;
}
#endif
// diagnostic (not in btf.h): phase stamps of the fast banded kernel, [ncols_local][6] shader clocks
extern "C" int btf_debug_stamps(btf_ctx* c, long long* out) {
  if (!c) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (!c->dbg) {
    int rc = dev_alloc(c, &c->dbg, (size_t)c->M * 6);
    if (rc) return rc;
    HIPCHK(c, hipMemset(c->dbg, 0, (size_t)c->M * 6 * sizeof(long long)));
    return BTF_OK;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out, c->dbg, (size_t)c->ml * 6 * sizeof(long long), hipMemcpyDeviceToHost));
  return BTF_OK;
}

int btf_set_tuning(btf_ctx* c, int rows_per_block_w, int rows_per_block_v) {
  if (!c || rows_per_block_w < 0 || rows_per_block_v < 0) return fail(c, BTF_EINVAL, "rows per workgroup must be >= 0 (0 = default)");
  c->rpb_w = rows_per_block_w; c->rpb_v = rows_per_block_v;
  return BTF_OK;
}

int btf_set_option(btf_ctx* c, int option, int value) {
  if (!c) return BTF_EINVAL;
  switch (option) {
    case BTF_OPT_SAMPLER:
      if (value < BTF_SAMPLER_BANDED || value > BTF_SAMPLER_BANDED_NOPANEL) return fail(c, BTF_EINVAL, "unknown sampler");
      c->sampler = value;
      c->ngp_v = 0;
      return BTF_OK;
    case BTF_OPT_NB_HISTOGRAMS:
      c->nb_hist = value != 0;
      return BTF_OK;
    case BTF_OPT_PG_EXACT:
      if (value < PG_MODE_DEFAULT || value > PG_MODE_SERIES_ALL) return fail(c, BTF_EINVAL, "unknown Polya-Gamma mode");
      c->pg_mode = value;
      return BTF_OK;
    case BTF_OPT_FUSE_GRAM:
      c->fuse_gram = value != 0;
      c->ngp_v = c->ngp_w = 0;
      return BTF_OK;
    case BTF_OPT_FUSED_STEP:
      if (value < 0 || value > 2) return fail(c, BTF_EINVAL, "BTF_OPT_FUSED_STEP: 0, 1 or 2");
      c->fused_step = value;
      return BTF_OK;
    case BTF_OPT_FUSED_DATAFLOW:
      c->fused_dataflow = value != 0 ? 1 : 0;
      return BTF_OK;
    case BTF_OPT_FUSED_SWEEP:
      c->fused_sweep = value != 0;
      c->sse_cols_valid = false;
      return BTF_OK;
    case BTF_OPT_SPLIT_ACCUM:
      c->split_accum = value != 0;
      c->w_local_done = c->v_local_done = false; c->sse_cols_valid = false;
      return BTF_OK;
    case BTF_OPT_CURVE_COUNTS:
      c->curve_opt = value != 0;
      c->ngp_v = c->ngp_w = 0;
      c->w_part_valid = false; c->w_local_done = c->v_local_done = false; c->sse_cols_valid = false;
      return BTF_OK;
    default:
      return fail(c, BTF_EINVAL, "unknown option");
  }
}

int btf_get_likelihood_form(btf_ctx* c, int32_t* form) {
  if (!c || !form) return BTF_EINVAL;
  if (!c->have_data) return fail(c, BTF_ESTATE, "set data first");
  *form = !c->weighted ? BTF_LIK_COMPLETE : (curve_on(c) ? BTF_LIK_CURVE_COUNTS : BTF_LIK_WEIGHTED);
  return BTF_OK;
}

int btf_get_draw_counters(btf_ctx* c, uint64_t* w, uint64_t* v) {
  if (!c || !w || !v) return BTF_EINVAL;
  *w = c->sweep_w; *v = c->sweep_v;
  return BTF_OK;
}

int btf_set_draw_counters(btf_ctx* c, uint64_t w, uint64_t v) {
  if (!c) return BTF_EINVAL;
  c->sweep_w = w; c->sweep_v = v;
  // The spectral sampler refines the previous sweep's eigenvectors; a chain continued from a checkpoint has none.
  // Forget them here, so that the chain that goes on (checkpoint() calls this too) and the restored one both start
  // from a cold eigen-solve: the same basis bit for bit, also inside a cluster of near-equal eigenvalues.
  // Likewise the residual parts the V sampler left for the next nu2 draw (BTF_OPT_FUSED_SWEEP): both take the six-launch sweep once.
  c->sse_cols_valid = false;
  HIPCHK(c, hipSetDevice(c->dev));
  const size_t rec = (size_t)c->K + (size_t)c->K * c->K + 8;
  if (c->eig) HIPCHK(c, hipMemsetAsync(c->eig, 0, rec * sizeof(double), c->stream));
  if (c->eig_cols) HIPCHK(c, hipMemsetAsync(c->eig_cols, 0, (size_t)c->M * rec * sizeof(double), c->stream));
  return BTF_OK;
}

int btf_comm_fork(btf_ctx* c, void* comm_stream) {
  if (!c || !comm_stream) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  int rc;
  // sharded runs with BTF_OPT_SPLIT_ACCUM leave the event behind the draw kernel (work queued after it must not hold
  // the exchange back); otherwise: everything queued so far
  if (!(c->ev_draw && split_applies(c))) { if ((rc = mark_draw(c))) return rc; }
  HIPCHK(c, hipStreamWaitEvent((hipStream_t)comm_stream, c->ev_draw, 0));
  return BTF_OK;
}
int btf_comm_join(btf_ctx* c, void* comm_stream) {
  if (!c || !comm_stream) return BTF_EINVAL;
  HIPCHK(c, hipSetDevice(c->dev));
  if (!c->ev_join) HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->ev_join, (hipStream_t)comm_stream));
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
  return BTF_OK;
}

// ---- the ctx-owned communicator ---------------------------------------------------------------------------------
namespace {
#define RCCLCHK(ctx, api, call)                                                                     \
  do {                                                                                              \
    ncclResult_t r__ = (call);                                                                      \
    if (r__ != ncclSuccess)                                                                         \
      return fail(ctx, BTF_EHIP, std::string(#call) + ": " + (api)->GetErrorString(r__));           \
  } while (0)

int comm_ready(btf_ctx* c, RcclApi** api) {
  if (!c) return BTF_EINVAL;
  *api = nullptr;
  if (c->peer_on) { HIPCHK(c, hipSetDevice(c->dev)); return BTF_OK; }
  if (!c->comm) return fail(c, BTF_ESTATE, "no communicator: btf_comm_init (or btf_peer_init) first");
  std::string why;
  if (!(*api = rccl_api(&why))) return fail(c, BTF_EHIP, why);
  HIPCHK(c, hipSetDevice(c->dev));
  return BTF_OK;
}
// the rank's blocks must be the equal-chunk decomposition the in-place gather reassembles
int comm_blocks_match(btf_ctx* c) {
  const int r = c->gather_rank, w = c->gather_world;
  if (c->row0 != comm_block_lo(c->N, r, w) || c->nl != comm_block_len(c->N, r, w) ||
      c->col0 != comm_block_lo(c->M, r, w) || c->ml != comm_block_len(c->M, r, w))
    return fail(c, BTF_ESTATE, "btf_set_shard blocks are not rank " + std::to_string(r) + " of " + std::to_string(w) +
                                   "'s equal chunks ceil(n / world) (btf_comm_block)");
  return BTF_OK;
}
// one collective of the peer-window transport (btf_comm.h): this rank's block [off, off + len) of W (which 0) or V (1)
// into every peer's buffer, and / or the sum over the ranks of red_n doubles at red_src into red_dst (may alias)
int peer_collective(btf_ctx* c, int which, size_t off, size_t len, const double* red_src, double* red_dst, int red_n, hipStream_t s) {
  if (c->gather_world <= 1) {
    if (red_n && red_dst != red_src) HIPCHK(c, hipMemcpyAsync(red_dst, red_src, (size_t)red_n * sizeof(double), hipMemcpyDeviceToDevice, s));
    return BTF_OK;
  }
  PeerArgs a{};
  a.tab = c->peer_tab; a.rank = c->gather_rank; a.world = c->gather_world; a.which = which;
  static const int wpp_cap = [] { const char* e = std::getenv("BTF_PEER_WPP"); const int v = e && *e ? std::atoi(e) : 64; return std::min(64, std::max(1, v)); }();
  static const int wg_bytes = [] { const char* e = std::getenv("BTF_PEER_WG_BYTES"); const int v = e && *e ? std::atoi(e) : 16384; return std::max(4096, v); }();
  a.wpp = len ? (int)std::min<size_t>((size_t)wpp_cap, std::max<size_t>(1, (len * sizeof(double) + wg_bytes - 1) / wg_bytes)) : 1;
  a.epoch = ++c->peer_epoch; a.off = off; a.len = len;
  a.red_src = red_src; a.red_dst = red_dst; a.red_n = red_n;
  a.counters = c->peer_counters; a.status = c->status; a.timeout_ticks = c->peer_timeout_ticks;
  hipLaunchKernelGGL(peer_exchange_kernel, dim3((unsigned)((a.world - 1) * a.wpp)), dim3(PEER_THREADS), 0, s, a);
  HIPCHK(c, hipGetLastError());
  return BTF_OK;
}
// one gather of `chunk` doubles per rank into `buf` (in place), or - rehearsal - of the whole message through scratch
int comm_gather(btf_ctx* c, RcclApi* api, double* buf, size_t chunk, hipStream_t s) {
  if (c->peer_on) {
    const bool isW = buf == c->W;
    const size_t unit = isW ? (size_t)c->K : (size_t)c->T * c->K;
    return peer_collective(c, isW ? 0 : 1, (size_t)(isW ? c->row0 : c->col0) * unit, (size_t)(isW ? c->nl : c->ml) * unit, nullptr, nullptr, 0, s);
  }
  if (c->comm_rehearse) {
    const size_t n = chunk * (size_t)c->gather_world;
    if (2 * n > c->comm_scr_elems) {
      int rc;
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (c->comm_stream) HIPCHK(c, hipStreamSynchronize(c->comm_stream));
      if ((rc = dev_alloc(c, &c->comm_scr, 2 * n))) return rc;
      c->comm_scr_elems = 2 * n;
      HIPCHK(c, hipMemsetAsync(c->comm_scr, 0, 2 * n * sizeof(double), s));
    }
    RCCLCHK(c, api, api->AllGather(c->comm_scr, c->comm_scr + n, n, ncclDouble, c->comm, s));
    return BTF_OK;
  }
  RCCLCHK(c, api, api->AllGather(buf + (size_t)c->comm_rank * chunk, buf, chunk, ncclDouble, c->comm, s));
  return BTF_OK;
}
// in line on the ctx's stream, or (BTF_OPT_SPLIT_ACCUM) on the communication stream between btf_comm_fork / btf_comm_join
int comm_gather_ordered(btf_ctx* c, double* buf, size_t chunk) {
  RcclApi* api;
  int rc;
  if ((rc = comm_ready(c, &api))) return rc;
  if ((rc = comm_blocks_match(c))) return rc;
  if (!c->split_accum) return comm_gather(c, api, buf, chunk, c->stream);
  if (!c->comm_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  if ((rc = btf_comm_fork(c, c->comm_stream))) return rc;
  if ((rc = comm_gather(c, api, buf, chunk, c->comm_stream))) return rc;
  return btf_comm_join(c, c->comm_stream);
}
int comm_adopt(btf_ctx* c, RcclApi* api, int rank, int world, const ncclUniqueId& id) {
  if (c->comm) { int rc = btf_comm_destroy(c); if (rc) return rc; }
  HIPCHK(c, hipSetDevice(c->dev));
  RCCLCHK(c, api, api->CommInitRank(&c->comm, world, id, rank));
  c->comm_rank = rank; c->comm_world = world;
  if (!c->comm_words) { int rc = dev_alloc(c, &c->comm_words, (size_t)16); if (rc) return rc; }
  return BTF_OK;
}
}  // namespace

int btf_comm_unique_id(unsigned char* id, int nbytes) {
  if (!id || nbytes < BTF_COMM_ID_BYTES) return fail(nullptr, BTF_EINVAL, "id buffer of BTF_COMM_ID_BYTES bytes expected");
  static_assert(sizeof(ncclUniqueId) == BTF_COMM_ID_BYTES, "BTF_COMM_ID_BYTES is NCCL_UNIQUE_ID_BYTES");
  std::string why;
  RcclApi* api = rccl_api(&why);
  if (!api) return fail(nullptr, BTF_EHIP, why);
  ncclUniqueId u;
  RCCLCHK(nullptr, api, api->GetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return BTF_OK;
}

int btf_comm_block(int n, int rank, int world, int32_t* lo, int32_t* len) {
  if (n < 0 || world < 1 || rank < 0 || rank >= world || !lo || !len) return BTF_EINVAL;
  *lo = comm_block_lo(n, rank, world); *len = comm_block_len(n, rank, world);
  return BTF_OK;
}

int btf_comm_init(btf_ctx* c, int rank, int world, const unsigned char* id, int nbytes) {
  if (!c) return BTF_EINVAL;
  if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(c, BTF_EINVAL, "need 0 <= rank < world <= 64 (W / V are padded for 64 ranks)");
  if (!id || nbytes != BTF_COMM_ID_BYTES) return fail(c, BTF_EINVAL, "id: the BTF_COMM_ID_BYTES bytes rank 0 got from btf_comm_unique_id");
  std::string why;
  RcclApi* api = rccl_api(&why);
  if (!api) return fail(c, BTF_EHIP, why);
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  int rc;
  if ((rc = comm_adopt(c, api, rank, world, u))) return rc;
  c->gather_rank = rank; c->gather_world = world; c->comm_rehearse = false;
  return BTF_OK;
}

int btf_comm_rehearse(btf_ctx* c, int rank, int world) {
  if (!c) return BTF_EINVAL;
  if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(c, BTF_EINVAL, "need 0 <= rank < world <= 64");
  std::string why;
  RcclApi* api = rccl_api(&why);
  if (!api) return fail(c, BTF_EHIP, why);
  ncclUniqueId u;
  RCCLCHK(c, api, api->GetUniqueId(&u));
  int rc;
  if ((rc = comm_adopt(c, api, 0, 1, u))) return rc;
  c->gather_rank = rank; c->gather_world = world; c->comm_rehearse = true;
  return BTF_OK;
}

#define BTF_PEER_MAGIC 0x42544650
int btf_peer_export(btf_ctx* c, unsigned char* out, int nbytes) {
  if (!c) return BTF_EINVAL;
  if (!out || nbytes != BTF_PEER_DESC_BYTES) return fail(c, BTF_EINVAL, "desc: a buffer of BTF_PEER_DESC_BYTES bytes");
  static_assert(BTF_PEER_DESC_BYTES == PEER_DESC_BYTES, "include/btf.h and btf_comm.h disagree on the descriptor size");
  int rc;
  if ((rc = btf_comm_destroy(c))) return rc;
  HIPCHK(c, hipSetDevice(c->dev));
  HIPCHK(c, hipExtMallocWithFlags((void**)&c->peer_box, sizeof(PeerMailbox), hipDeviceMallocFinegrained));
  HIPCHK(c, hipMemset(c->peer_box, 0, sizeof(PeerMailbox)));
  HIPCHK(c, hipDeviceSynchronize());
  PeerDesc d;
  std::memset(&d, 0, sizeof(d));
  HIPCHK(c, hipIpcGetMemHandle(&d.W, c->W));
  HIPCHK(c, hipIpcGetMemHandle(&d.V, c->V));
  HIPCHK(c, hipIpcGetMemHandle(&d.box, c->peer_box));
  d.pW = (unsigned long long)(uintptr_t)c->W; d.pV = (unsigned long long)(uintptr_t)c->V; d.pbox = (unsigned long long)(uintptr_t)c->peer_box;
  d.pid = (long long)getpid(); d.dev = c->dev; d.magic = BTF_PEER_MAGIC;
  d.wbytes = (long long)((size_t)(c->N + 64) * c->K * sizeof(double)); d.vbytes = (long long)((size_t)(c->M + 64) * c->T * c->K * sizeof(double));
  std::memset(out, 0, (size_t)nbytes);
  std::memcpy(out, &d, sizeof(d));
  return BTF_OK;
}

int btf_peer_init(btf_ctx* c, int rank, int world, const unsigned char* descs, int nbytes) {
  if (!c) return BTF_EINVAL;
  if (world < 1 || world > PEER_MAX || rank < 0 || rank >= world) return fail(c, BTF_EINVAL, "need 0 <= rank < world <= 64 (W / V are padded for 64 ranks)");
  if (!descs || nbytes != world * BTF_PEER_DESC_BYTES) return fail(c, BTF_EINVAL, "descs: world x BTF_PEER_DESC_BYTES bytes, rank-major (what every rank's btf_peer_export gave)");
  if (!c->peer_box) return fail(c, BTF_ESTATE, "btf_peer_export first (it creates this rank's mailbox)");
  HIPCHK(c, hipSetDevice(c->dev));
  PeerTable tab;
  std::memset(&tab, 0, sizeof(tab));
  const long long me = (long long)getpid();
  for (int r = 0; r < world; ++r) {
    PeerDesc d;
    std::memcpy(&d, descs + (size_t)r * BTF_PEER_DESC_BYTES, sizeof(d));
    if (d.magic != BTF_PEER_MAGIC) return fail(c, BTF_EINVAL, "descriptor " + std::to_string(r) + " is not a btf_peer_export result");
    if (r == rank) {
      if (d.pid != me || d.pbox != (unsigned long long)(uintptr_t)c->peer_box) return fail(c, BTF_EINVAL, "descriptor " + std::to_string(r) + " is not this context's own export");
      tab.W[r] = c->W; tab.V[r] = c->V; tab.box[r] = c->peer_box;
      continue;
    }
    if (d.wbytes != (long long)((size_t)(c->N + 64) * c->K * sizeof(double)) || d.vbytes != (long long)((size_t)(c->M + 64) * c->T * c->K * sizeof(double)))
      return fail(c, BTF_EINVAL, "rank " + std::to_string(r) + "'s W / V buffers have another shape than this rank's");
    if (d.pid == me) {        // another context of this process: its pointers are ours too
      if (d.dev != c->dev) {
        int can = 0;
        HIPCHK(c, hipDeviceCanAccessPeer(&can, c->dev, d.dev));
        if (!can) return fail(c, BTF_EHIP, "device " + std::to_string(c->dev) + " cannot map device " + std::to_string(d.dev));
        const hipError_t e = hipDeviceEnablePeerAccess(d.dev, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(c, BTF_EHIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        (void)hipGetLastError();
      }
      tab.W[r] = (double*)(uintptr_t)d.pW; tab.V[r] = (double*)(uintptr_t)d.pV; tab.box[r] = (PeerMailbox*)(uintptr_t)d.pbox;
      continue;
    }
    void* p[3] = {nullptr, nullptr, nullptr};
    const hipIpcMemHandle_t* h[3] = {&d.W, &d.V, &d.box};
    for (int i = 0; i < 3; ++i) {
      const hipError_t e = hipIpcOpenMemHandle(&p[i], *h[i], hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess)
        return fail(c, BTF_EHIP, "hipIpcOpenMemHandle of rank " + std::to_string(r) + "'s buffers: " + hipGetErrorString(e) +
                                     " (the ranks must be processes of one node whose GPUs can map each other)");
      c->peer_opened.push_back(p[i]);
    }
    tab.W[r] = (double*)p[0]; tab.V[r] = (double*)p[1]; tab.box[r] = (PeerMailbox*)p[2];
  }
  int rc;
  if ((rc = dev_alloc(c, &c->peer_tab, (size_t)1))) return rc;
  if ((rc = dev_alloc(c, &c->peer_counters, (size_t)PEER_MAX + 1))) return rc;
  if (!c->comm_words) { if ((rc = dev_alloc(c, &c->comm_words, (size_t)16))) return rc; }
  HIPCHK(c, hipMemcpy(c->peer_tab, &tab, sizeof(tab), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemset(c->peer_counters, 0, (PEER_MAX + 1) * sizeof(unsigned)));
  const char* ms = std::getenv("BTF_PEER_TIMEOUT_MS");
  const long long msv = ms && *ms ? std::atoll(ms) : 20000;
  c->peer_timeout_ticks = std::max<long long>(1, msv) * 100000;          // wall_clock64 ticks at 100 MHz
  c->peer_epoch = 0;
  c->comm_rank = c->gather_rank = rank; c->comm_world = c->gather_world = world; c->comm_rehearse = false;
  c->peer_on = true;
  return BTF_OK;
}

int btf_comm_destroy(btf_ctx* c) {
  if (!c) return BTF_EINVAL;
  if (c->peer_box || c->peer_on) {
    (void)hipSetDevice(c->dev);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    for (void* p : c->peer_opened) (void)hipIpcCloseMemHandle(p);
    c->peer_opened.clear();
    if (c->peer_box) { (void)hipFree(c->peer_box); c->peer_box = nullptr; }
    if (c->peer_tab) { (void)hipFree(c->peer_tab); c->peer_tab = nullptr; }
    if (c->peer_counters) { (void)hipFree(c->peer_counters); c->peer_counters = nullptr; }
    c->peer_on = false; c->peer_epoch = 0;
  }
  if (c->comm) {
    std::string why;
    RcclApi* api = rccl_api(&why);
    (void)hipSetDevice(c->dev);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (api) (void)api->CommDestroy(c->comm);
    c->comm = nullptr;
  }
  if (c->comm_stream) { (void)hipStreamDestroy(c->comm_stream); c->comm_stream = nullptr; }
  if (c->comm_scr) { (void)hipFree(c->comm_scr); c->comm_scr = nullptr; c->comm_scr_elems = 0; }
  if (c->comm_words) { (void)hipFree(c->comm_words); c->comm_words = nullptr; }
  c->comm_rank = c->gather_rank = 0; c->comm_world = c->gather_world = 1; c->comm_rehearse = false;
  return BTF_OK;
}

int btf_comm_info(btf_ctx* c, int32_t* out) {
  if (!c || !out) return BTF_EINVAL;
  out[0] = c->peer_on ? 2 : c->comm ? 1 : 0; out[1] = c->comm_rank; out[2] = c->comm_world; out[3] = c->gather_rank; out[4] = c->gather_world;
  out[5] = c->comm_rehearse ? 1 : 0;
  int v = 0;
  std::string why;
  RcclApi* api = c->comm ? rccl_api(&why) : nullptr;
  if (api) (void)api->GetVersion(&v);
  out[6] = v;
  int n = 0;
  if (api && api->CommCount(c->comm, &n) == ncclSuccess) out[7] = n; else out[7] = 0;
  return BTF_OK;
}

int btf_allgather_W(btf_ctx* c) {
  if (!c) return BTF_EINVAL;
  if (!c->have_W) return fail(c, BTF_ESTATE, "set W first");
  return comm_gather_ordered(c, c->W, (size_t)comm_chunk(c->N, c->gather_world) * c->K);
}

int btf_allgather_V(btf_ctx* c) {
  if (!c) return BTF_EINVAL;
  if (!c->have_V) return fail(c, BTF_ESTATE, "set V first");
  return comm_gather_ordered(c, c->V, (size_t)comm_chunk(c->M, c->gather_world) * c->T * c->K);
}

int btf_allreduce_sse(btf_ctx* c) {
  RcclApi* api;
  int rc;
  if ((rc = comm_ready(c, &api))) return rc;
  if (!c->hyp) return fail(c, BTF_ESTATE, "no device-resident scalars yet (btf_device_scalars)");
  if (c->peer_on) return peer_collective(c, 2, 0, 0, c->hyp + HYP_SSE, c->hyp + HYP_SSE, 1, c->stream);
  RCCLCHK(c, api, api->AllReduce(c->hyp + HYP_SSE, c->hyp + HYP_SSE, 1, ncclDouble, ncclSum, c->comm, c->stream));
  return BTF_OK;
}

int btf_allreduce_sum(btf_ctx* c, double* vals, int n) {
  RcclApi* api;
  int rc;
  if ((rc = comm_ready(c, &api))) return rc;
  if (!vals || n < 1 || n > 16) return fail(c, BTF_EINVAL, "1 to 16 doubles");
  HIPCHK(c, hipMemcpyAsync(c->comm_words, vals, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (c->peer_on) { if ((rc = peer_collective(c, 2, 0, 0, c->comm_words, c->comm_words, n, c->stream))) return rc; }
  else RCCLCHK(c, api, api->AllReduce(c->comm_words, c->comm_words, (size_t)n, ncclDouble, ncclSum, c->comm, c->stream));
  HIPCHK(c, hipMemcpyAsync(vals, c->comm_words, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return BTF_OK;
}

// Host-only self-test: walks every piece of index arithmetic the host side feeds to the kernels - penalty matrix and
// stencil tables, the LDS layouts of the three V samplers, the band assembly program, elimination orders, row / chunk
// geometry of the (split) accumulation, Polya-Gamma sampler classes - over a grid of shapes, checking bounds and
// coverage.  No HIP call: it runs on a CPU-only box, and under the host sanitizers (scripts/asan_host.sh builds the
// library's host side with -fsanitize=address,undefined and runs this).  Returns 0, or the source line of the first
// failed check.
int btf_host_selftest(void) {
#define ST_CHECK(cond) do { if (!(cond)) return __LINE__; } while (0)
  const int Ts[] = {4, 6, 7, 12, 16, 33, 64, 65, 370};
  for (int T : Ts)
    for (int tf = 0; tf <= 3; ++tf) {
      std::vector<double> Delta;
      int nD = 0;
      build_delta_dense(T, tf, Delta, nD);
      ST_CHECK(nD > 0 && (size_t)nD * T == Delta.size());
      std::vector<int> ptr, row;
      std::vector<double> coef;
      stencil_csr(T, tf, Delta, nD, ptr, row, coef);
      ST_CHECK(ptr.size() == (size_t)T * (tf + 2) + 1 && ptr.front() == 0 && (size_t)ptr.back() == row.size() && row.size() == coef.size());
      for (size_t e = 0; e + 1 < ptr.size(); ++e) ST_CHECK(ptr[e] <= ptr[e + 1]);
      for (int r : row) ST_CHECK(r >= 0 && r < nD);
      for (int K = 1; K <= 10; ++K) {
        const int n = T * K, S = tf + 1;
        {   // spectral sampler: pivot order is a permutation, layout offsets increase
          std::vector<int> seen(T, 0);
          for (int i = 0; i < T; ++i) { const int d = spectral_depth_of_pivot(i, T, S); ST_CHECK(d >= 0 && d < T); seen[d]++; }
          for (int v : seen) ST_CHECK(v == 1);
          const VsLayout L = vs_layout(T, K, tf, nD);
          const int offs[] = {L.U, L.g, L.itau, L.P, L.Pm, L.mraw, L.mt, L.mtm, L.zz, L.rec, L.win, L.gs, L.flag, L.eG, L.eo, L.esc, L.total};
          for (size_t i = 0; i + 1 < sizeof(offs) / sizeof(int); ++i) ST_CHECK(offs[i] >= 0 && offs[i] < offs[i + 1]);
        }
        if (twist_ok(T, K, tf)) {
          for (int wt = 0; wt <= 2; ++wt) {      // 2: weighted, likelihood blocks fetched from the partials (sources < -1)
            const TwLayout W = tw_layout(T, K, tf, wt);
            ST_CHECK(W.nl + W.nr + W.ns == n && W.total > 0);
            std::vector<int> seen(n, 0);
            for (int i = 0; i < n; ++i) { const int g = twist_order(i, n, W.nl, W.nr); ST_CHECK(g >= 0 && g < n); seen[g]++; }
            for (int v : seen) ST_CHECK(v == 1);
            if (tw_lds_bytes(T, K, tf, wt) <= 160 * 1024) {
              std::vector<int> tab;
              fill_table_host(T, K, tf, wt, tab);
              ST_CHECK(!tab.empty() && tab.size() % (4 * VT_THREADS) == 0);
              for (size_t e = 0; e < tab.size(); e += 4) {
                ST_CHECK(tab[e] >= 0 && tab[e] < W.total && tab[e + 1] < W.total);
                if (wt < 2) ST_CHECK(tab[e + 1] >= 0);
                else ST_CHECK(tab[e + 1] >= 0 || (-2 - tab[e + 1] >= 0 && -2 - tab[e + 1] < T * tri(K)));
                if (wt == 2 && tab[e + 1] >= 0) ST_CHECK(tab[e + 1] < W.Ql || tab[e + 1] >= W.flag);     // no source inside the (empty) block area
                ST_CHECK(tab[e + 2] >= -1 && tab[e + 2] < W.total);
              }
            }
          }
        }
        const VbLayout L = vb_layout(T, K, tf, 1);
        ST_CHECK(L.total > 0 && L.band >= 0 && L.rhs > L.band && L.dummy + 64 * 9 + 8 <= L.total);
        // chunked chain: the chunk the budget allows fits it, leaves room for two band widths, and its layout is ordered
        for (int wt = 0; wt <= 1; ++wt) {
          const int bwc = (tf + 1) * K, ch = vc_pick_chunk(T, K, tf, wt, VC_LDS_BUDGET);
          if (ch > 0) {
            ST_CHECK(ch >= 2 * bwc + 2 && ch <= n && vc_lds_bytes(T, K, tf, wt, ch) <= VC_LDS_BUDGET);
            const VcLayout C = vc_layout(T, K, tf, wt, ch);
            ST_CHECK(C.V.band == 0 && C.V.rhs > C.V.band + C.V.npad * C.V.R1 && C.V.invd > C.V.rhs && C.m0 > C.V.dummy && C.zs == C.m0 + n);
            ST_CHECK(C.Ql > C.P && C.flag > C.Ql && C.xk > C.flag && C.total == C.xk + 64 && C.VC == ch + bwc && C.QT * K >= C.VC + K);
          }
        }
        for (int i = 0; i + 1 < 8; ++i) ST_CHECK(w_z_offset(i + 1, K) - w_z_offset(i, K) == std::min(i + 1, K));
      }
    }
  // rows per workgroup and the split of an accumulation around a rank's own block: every row exactly once
  const int Rs[] = {64, 512, 576, 1536, 4096, 16384, 65536}, tilesv[] = {1, 4, 64};
  for (int Rdim : Rs)
    for (int tiles : tilesv)
      for (int wt = 0; wt <= 1; ++wt)
        for (int world = 2; world <= 8; world *= 2)
          for (int rank = 0; rank < world; ++rank) {
            const int rpb = pick_rpb(Rdim, tiles, 0, wt != 0);
            ST_CHECK(rpb >= 64 && rpb % 64 == 0);
            const int chunk = (Rdim + world - 1) / world, lo = std::min(rank * chunk, Rdim), hi = std::min(lo + chunk, Rdim);
            const SplitGeom g = split_geom(lo, hi, Rdim, rpb, tiles);
            if (!g.ok) continue;
            std::vector<int> cover(Rdim, 0);
            auto walk = [&](ChunkMap cm, int rp, int nch) {
              for (int l = 0; l < nch; ++l) {
                int r0 = cm.row_base + l * rp;
                if (r0 >= cm.skip_at) r0 += cm.skip_rows;
                const int r1 = std::min(r0 + rp, cm.row_end);
                for (int r = r0; r < r1; ++r) { if (r < 0 || r >= Rdim) return false; cover[r]++; }
              }
              return true;
            };
            ST_CHECK(walk(ChunkMap{g.nch_r, g.lo, INT_MAX, 0, g.hi, 0}, g.rpb_l, g.nch_l));
            ST_CHECK(walk(ChunkMap{0, 0, g.lo, g.hi - g.lo, Rdim, 0}, rpb, g.nch_r));
            for (int v : cover) ST_CHECK(v == 1);
          }
  // one-round fitting of the accumulation launch: a rank's slabs of C5 / 8 (tiles x chunks + side workgroups <= CUs)
  {
    const int rw = pick_rpb(65536, 4, 0, false, 256, 32), rv = pick_rpb(4096, 64, 0, false, 256, 1);
    ST_CHECK(rw % 64 == 0 && rv % 64 == 0);
    ST_CHECK(4 * ((65536 + rw - 1) / rw) + 32 <= 256 && 4 * ((65536 + rw - 1) / rw) + 32 > 192);
    ST_CHECK(64 * ((4096 + rv - 1) / rv) + 1 <= 256);
    ST_CHECK(pick_rpb(16384, 4, 0, false, 256, 0) == 512 && pick_rpb(512, 128, 0, false, 256, 1) == 512);   // C3: untouched
    ST_CHECK(pick_rpb(65536, 4, 0, false, 0, 0) == 1024);                                                       // unknown chip: the old rule
  }
  // which Polya-Gamma sampler takes a count
  ST_CHECK(pg_class_of(0.0, PG_MODE_DEFAULT) == PG_CLASS_NONE && pg_class_of(4.0, PG_MODE_DEFAULT) == PG_CLASS_FLAT);
  ST_CHECK(pg_class_of(33.0, PG_MODE_DEFAULT) == PG_CLASS_SERIES && pg_class_of(33.0, PG_MODE_EXACT_ALL) == PG_CLASS_FLAT);
  ST_CHECK(pg_class_of(2.5, PG_MODE_EXACT_ALL) == PG_CLASS_FRAC && pg_class_of(2.0, PG_MODE_SERIES_ALL) == PG_CLASS_SERIES);
  ST_CHECK(pg_class_of(200.0, PG_MODE_EXACT_ALL) == PG_CLASS_NORMAL);
#undef ST_CHECK
  return 0;
}

int btf_get_accum_bytes_per_cell(btf_ctx* c, double* bytes) {
  if (!c || !bytes) return BTF_EINVAL;
  if (!c->have_data) return fail(c, BTF_ESTATE, "set data first");
  if (!lik_weighted(c)) *bytes = 8.0;                                 // the linear statistic alone
  else if (c->C8_wT) *bytes = 9.0;                                    // + replicate counts as bytes
  else if (c->A8_wT) *bytes = 9.0;                                    // pseudo-data as bytes + f64 weights
  else *bytes = 16.0;
  return BTF_OK;
}

int btf_get_V_sampler(btf_ctx* c, int32_t* which) {
  if (!c || !which) return BTF_EINVAL;
  const int ch = banded_choice(c);
  *which = ch == 3 ? BTF_SAMPLER_SPECTRAL : (ch == 2 ? (c->sampler == BTF_SAMPLER_BANDED_NOPANEL ? BTF_SAMPLER_BANDED_NOPANEL : BTF_SAMPLER_BANDED)
                                                     : (ch == 1 || ch == 4 ? BTF_SAMPLER_CHAIN : BTF_SAMPLER_GENERIC));
  return BTF_OK;
}

}  // extern "C"
