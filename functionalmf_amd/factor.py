"""Bayesian Tensor Filtering models with the Gibbs hot path on MI355X.

Counterpart of functionalmf/factor.py:23-460 (``BayesianTensorFiltering``,
``GaussianBayesianTensorFiltering``, ``BinomialBayesianTensorFiltering``): same
constructor keywords, public attributes (``W V Tau2 lam2 sigma2 nu2 Delta
sample_*``), method names and ``run_gibbs`` result layout, so a script written
for ``functionalmf.factor`` only changes its import.  The bodies of
``_resample_W`` / ``_resample_V`` / ``_resample_nu2`` (and ``_init_V``) are
calls into the C ABI of include/btf.h; the scalar / per-column hyper-parameter
draws stay on the host because in ``rng="host"`` mode they must consume the
legacy numpy stream in the reference's order.

Extra keywords (all optional):
  compat  "reference" (default) reproduces the reference's result-changing
          quirks Q1-Q3 (stale cached likelihood weights, lam2 rate overwrite);
          "exact" uses the textbook conditionals.
  rng     "host" (default): every normal / gamma comes from the global legacy
          numpy generator in the reference's order, so a seeded run reproduces
          the reference chain; "device": the normals of the W and V draws come
          from Philox4x32-10 on the GPU (nothing but scalars crosses PCIe).
  device  HIP device ordinal;  stream: hipStream_t handle (int) or None.
  shard   (rank, world) to update only this rank's block of rows / columns
          (see functionalmf_amd/parallel.py); W and V stay replicated.
  sampler square root used for the noise term of the V draw (fast_mvn.py:44 uses
          CHOLMOD's P' L^-T; scikit-sparse is not part of the reference tree, so the
          ordering is declared here - DESIGN.md section 2):
          "banded"   block-banded LDL' in the declared elimination order
                     (`v_order()`); default with rng="host";
          "spectral" complete Gaussian data only: the K x K likelihood block is the
                     same for every depth and column, its eigenvectors split each
                     column into K scalar banded systems (include/btf.h,
                     btf_get_V_sampler); default with rng="device", falls back to
                     "banded" for data with missing cells / Binomial weights;
          "auto"     (default) as described above.
"""
import ctypes

import numpy as np

from . import _native
from .genlasso import _BayesianModel, ConjugateInverseGammaPrior
from .utils import bayes_grid_penalty, sample_horseshoe_plus, sample_horseshoe
from .parallel import ShardPlan, Exchange


def _scalar(x):
    """Python float from a float, 0-d or 1-element array (the reference's prior draws
    are shape-(1,) arrays until the first resample replaces them)."""
    return float(np.asarray(x, dtype=float).reshape(-1)[0])


def _fingerprint(a):
    """Cheap content check of an observation array (see set_data): the bytes of 64 strided entries (NaN patterns
    included).  ~1 us - it runs in front of every half-sweep's launches (np.nansum + isnan took 8 us of the ~15 us a
    half-sweep spends on the host)."""
    flat = a.reshape(-1) if isinstance(a, np.ndarray) and a.flags.c_contiguous else None
    if flat is None or flat.size == 0:
        return None
    return flat[::max(1, flat.size // 61)][:64].tobytes()


def _digest(a):
    """Hash of every byte of an observation array (xxh3: ~10 GB/s, 30 ms for the 268 MB of config C3)."""
    if not (isinstance(a, np.ndarray) and a.flags.c_contiguous):
        return None
    try:
        import xxhash
        return xxhash.xxh3_64_intdigest(a.data)
    except ImportError:
        import warnings
        import zlib
        global _CRC_WARNED
        if not _CRC_WARNED:
            _CRC_WARNED = True
            warnings.warn("xxhash is not installed: the periodic check for in-place edits of the observation array hashes with "
                          "zlib.crc32 (~1 GB/s instead of ~10 GB/s); its period stretches accordingly (see set_data)", RuntimeWarning)
        return zlib.crc32(a.data)


_CRC_WARNED = False


def stale_row_sources(nrows, nembeds, any_nan):
    """Quirk Q1 (factor.py:320,349): without any NaN in the data the per-row design
    cache is refreshed only for rows < K, so rows >= K reuse row K-1's weights."""
    src = np.arange(nrows, dtype=np.int32)
    if not any_nan and nrows > nembeds:
        src[nembeds:] = nembeds - 1
    return src


def stale_col_sources(missing):
    """Quirk Q2 (factor.py:394-400): column j reuses the cached design (hence the
    weights) of the last column at which the NaN pattern of ybar[:, j, :] changed.
    `missing`: bool (N, M, T), True where the cell has no observation."""
    M = missing.shape[1]
    src = np.zeros(M, dtype=np.int32)
    for j in range(1, M):
        same = np.array_equal(missing[:, j, :], missing[:, src[j - 1], :])
        src[j] = src[j - 1] if same else j
    return src


class BayesianTensorFiltering(_BayesianModel):
    def __init__(self, nrows, ncols, ndepth,
                 nembeds=5, tf_order=2,
                 sigma2_init=None, sigma2_true=None,
                 sigma2_a=0.1, sigma2_b=0.1,
                 lam2_init=None, lam2_true=None,
                 Tau2_init=None, Tau2_true=None,
                 W_init=None, V_init=None,
                 W_true=None, V_true=None,
                 stability=1e-6,
                 force_psd=True,
                 force_psd_eps=1e-6,
                 force_psd_attempts=4,
                 compat="reference", rng="host", device=0, stream=None, shard=None, device_seed=0,
                 sampler="auto", overlap_exchange=False, rehearse_rank=False,
                 **kwargs):
        super().__init__(**kwargs)
        if compat not in _native.COMPAT:
            raise ValueError("compat must be 'reference' or 'exact'")
        if rng not in ("host", "device"):
            raise ValueError("rng must be 'host' or 'device'")
        if sampler != "auto" and sampler not in _native.SAMPLERS:
            raise ValueError("sampler must be 'auto' or one of %s" % sorted(_native.SAMPLERS))
        self.nrows, self.ncols, self.ndepth, self.nembeds = nrows, ncols, ndepth, nembeds
        self.tf_order = tf_order
        self.stability = stability
        self.linalg_opts = dict(force_psd=force_psd, force_psd_eps=force_psd_eps,
                                force_psd_attempts=force_psd_attempts)
        self.compat, self.rng, self.device = compat, rng, device
        self._device_seed = int(device_seed)
        self._draws = 0

        # Sharded runs exchange W / V through the context's own RCCL communicator (btf_allgather_W / _V; parallel.Exchange
        # bootstraps it over torch.distributed).  Default: the collectives are issued on the ctx's own stream, in line.  overlap_exchange=True: the all-gather of the freshly drawn block runs on a
        # communication stream ordered behind the draw kernel (btf_comm_fork), while the ctx's stream already accumulates
        # the chunks of the next half-sweep that reduce over this rank's own block (BTF_OPT_SPLIT_ACCUM); the rest of that
        # accumulation waits for the gather (btf_comm_join).  Measured on one MI355X playing rank 0 of 8 at C5
        # (bench.py --as-rank 0/8): the kernels get shorter (2 x 26 us against 57 us per accumulation) but the two
        # cross-stream edges per half-sweep cost more than the own-block eighth hides - 190 us per step against 157 in
        # line - so it stays opt-in until a run on 8 GPUs (longer, skewed gathers) says otherwise.
        import os
        # device context first: without the HIP library / a GPU nothing below can run
        self._ctx = _native.Context(nrows, ncols, ndepth, nembeds, tf_order, device=device, stream=stream)
        self._plan = ShardPlan(nrows, ncols, *(shard if shard is not None else (0, 1)))
        # rehearse_rank: this process plays rank shard[0] of shard[1] alone (timing rehearsal, see parallel.Exchange)
        self._exchange = Exchange(self._plan, self._ctx, overlap=bool(overlap_exchange), rehearse=bool(rehearse_rank))
        if self._plan.world > 1:
            self._ctx.call("btf_set_shard", *self._plan.mine())
        if self._exchange.active and self._exchange.overlap:
            self._ctx.call("btf_set_option", _native.OPT_SPLIT_ACCUM, 1)
        self.sampler = sampler
        self._ctx.call("btf_set_option", _native.OPT_SAMPLER,
                       _native.SAMPLERS[("spectral" if rng == "device" else "banded") if sampler == "auto" else sampler])
        self._data_key = None
        self._data_digest, self._data_checked, self._data_uploaded = None, 0.0, 0.0
        self.data_check_seconds = 5.0     # full re-hash of the bound observation array at most this often (None: never)
        self._W_host_new = self._V_host_new = True      # host copy must be pushed before the next kernel
        self._W_dev_new = self._V_dev_new = False       # device copy is newer than the host mirror
        self._tau_dirty = True
        self._v_version = 0
        # scalar hyper-parameters: host values, or (rng="device", unsharded) device-resident
        self._sc = {"nu2": 1.0, "sigma2": 1.0, "lam2": 1.0, "lam2_a": 1.0}
        self._sc_host_new, self._sc_dev_new = False, False
        # rng="device": nu2, sigma2, lam2, lam2_a live on the GPU.  Sharded runs keep them there too: W and V are
        # replicated, so only the residual sum of squares of nu2 | rest needs an exchange (one all-reduced double)
        self._dev_scalars = rng == "device"
        if self._dev_scalars:
            self._ctx.call("btf_device_scalars", 1)

        # trend-filtering prior (factor.py:50)
        self.Delta = bayes_grid_penalty(ndepth, tf_order)

        # construction draws in the reference's order (factor.py:53-110)
        self.sigma2_a, self.sigma2_b = sigma2_a, sigma2_b
        self.sigma2_model = ConjugateInverseGammaPrior(1, sigma2_a, sigma2_b)
        self.sample_sigma2 = sigma2_true is None
        if sigma2_true is not None:
            self.sigma2 = sigma2_true
        elif sigma2_init is not None:
            self.sigma2 = sigma2_init
        else:
            self._init_sigma2()

        self.sample_lam2 = lam2_true is None
        if lam2_true is not None:
            self.lam2 = lam2_true
        else:
            self._init_lam2()
            if lam2_init is not None:
                self.lam2 = lam2_init

        self.sample_Tau2 = Tau2_true is None
        if Tau2_true is not None:
            self.Tau2 = np.array(Tau2_true, dtype=float)
        elif Tau2_init is not None:
            self.Tau2 = np.array(Tau2_init, dtype=float)
        else:
            self._init_Tau2()
        assert self.Tau2.shape == (ncols, self.Delta.shape[0])

        self.sample_W = W_true is None
        if W_true is not None:
            self._W = np.array(W_true, dtype=float)
        elif W_init is not None:
            self._W = np.array(W_init, dtype=float)
        else:
            self._init_W()
        assert self._W.shape == (nrows, nembeds)

        self.sample_V = V_true is None
        if V_true is not None:
            self._V = np.array(V_true, dtype=float)
        elif V_init is not None:
            self._V = np.array(V_init, dtype=float)
        else:
            self._init_V()
        assert self._V.shape == (ncols, ndepth, nembeds)

    # ---- host mirrors of the device-resident factors -----------------------------
    # Reading .W / .V hands out the numpy mirror (refreshed from the GPU if a kernel
    # wrote since); because callers may write into it in place (the reference's
    # `model.W[:] = ...` idiom) the mirror is pushed back before the next kernel.
    @property
    def W(self):
        self._pull_W()
        self._W_host_new = True
        return self._W

    @W.setter
    def W(self, value):
        self._W = np.array(value, dtype=float)
        self._W_host_new, self._W_dev_new = True, False

    @property
    def V(self):
        self._pull_V()
        self._V_host_new = True
        self._v_version += 1           # may be written through the returned array
        return self._V

    @V.setter
    def V(self, value):
        self._V = np.array(value, dtype=float)
        self._V_host_new, self._V_dev_new = True, False
        self._v_version += 1

    # Tau2 (and, in rng="device" mode, its three auxiliary horseshoe+ levels) are handed out the
    # same way: refreshed from the GPU if a kernel wrote them, pushed back before the next kernel
    # because any access may be followed by an in-place write.
    def _pull_tau(self):
        if getattr(self, "_tau_dev_new", False):
            self._ctx.call("btf_get_tau", _native.dptr(self._Tau2), _native.dptr(self._Tau2_a),
                           _native.dptr(self._Tau2_b), _native.dptr(self._Tau2_c))
            self._tau_dev_new = False

    @property
    def Tau2(self):
        self._pull_tau()
        self._tau_dirty = True
        self._lsum_valid = False
        self._lsum_on_device = False
        return self._Tau2

    @Tau2.setter
    def Tau2(self, value):
        self._Tau2 = value
        self._tau_dirty = True
        self._tau_dev_new = False

    def _chain_get(self, name):
        self._pull_tau()
        self._chain_dirty = True
        return getattr(self, name)

    def _chain_set(self, name, value):
        setattr(self, name, value)
        self._chain_dirty = True

    Tau2_a = property(lambda self: self._chain_get("_Tau2_a"), lambda self, v: self._chain_set("_Tau2_a", v))
    Tau2_b = property(lambda self: self._chain_get("_Tau2_b"), lambda self, v: self._chain_set("_Tau2_b", v))
    Tau2_c = property(lambda self: self._chain_get("_Tau2_c"), lambda self, v: self._chain_set("_Tau2_c", v))

    # sigma2, lam2, lam2_a (and the Gaussian model's scalar nu2): in rng="device" mode they are
    # drawn on the GPU (btf_draw_scalars / btf_draw_lam2) and fetched when somebody looks.
    def _pull_scalars(self):
        if self._sc_dev_new:
            out = np.zeros(6)
            self._ctx.call("btf_get_scalars", _native.dptr(out))
            self._sc["nu2"], self._sc["sigma2"], self._sc["lam2"], self._sc["lam2_a"] = (float(x) for x in out[:4])
            self._sc_dev_new = False

    def _push_scalars(self):
        if self._dev_scalars and self._sc_host_new:
            self._ctx.call("btf_set_scalars", *(_scalar(self._sc[k]) for k in ("nu2", "sigma2", "lam2", "lam2_a")))
            self._sc_host_new = False

    def _sc_get(self, name):
        self._pull_scalars()
        return self._sc[name]

    def _sc_set(self, name, value):
        self._pull_scalars()
        self._sc[name] = value
        self._sc_host_new = True

    sigma2 = property(lambda self: self._sc_get("sigma2"), lambda self, v: self._sc_set("sigma2", v))
    lam2 = property(lambda self: self._sc_get("lam2"), lambda self, v: self._sc_set("lam2", v))
    lam2_a = property(lambda self: self._sc_get("lam2_a"), lambda self, v: self._sc_set("lam2_a", v))

    def _pull_W(self):
        if self._W_dev_new:
            self._ctx.call("btf_get_W", _native.dptr(self._W))
            self._W_dev_new = False

    def _pull_V(self):
        if self._V_dev_new:
            self._ctx.call("btf_get_V", _native.dptr(self._V))
            self._V_dev_new = False
            self._v_version += 1

    def _push_state(self):
        if self._W_host_new:
            self._W = _native.as_f64(self._W)
            self._ctx.call("btf_set_W", _native.dptr(self._W))
            self._W_host_new = False
        if self._V_host_new:
            self._V = _native.as_f64(self._V)
            self._ctx.call("btf_set_V", _native.dptr(self._V))
            self._V_host_new = False
        self._push_scalars()
        # (with device-resident scalars the kernels ignore the two host values below)
        lam2, sigma2 = _scalar(self._sc["lam2"]), _scalar(self._sc["sigma2"])
        if self._tau_dirty and not getattr(self, "_tau_dev_new", False):
            self._Tau2 = _native.as_f64(self._Tau2)
            self._ctx.call("btf_set_hyper", _native.dptr(self._Tau2), lam2, sigma2)
            self._tau_dirty = False
        elif not self._dev_scalars:
            self._ctx.call("btf_set_hyper", None, lam2, sigma2)

    def _next_seed(self):
        self._draws += 1
        return (self._device_seed * 0x9E3779B97F4A7C15 + self._draws) & 0xFFFFFFFFFFFFFFFF

    # ---- data -------------------------------------------------------------------
    def _bind_data(self, data):
        """Upload the observations once and hoist their sufficient statistics (what
        factor.py:329-330 / :374-375 recompute on every half-sweep)."""
        arrays = data if isinstance(data, (tuple, list)) else (data,)
        key = tuple((id(a), a.shape, _fingerprint(a)) for a in arrays)
        if key == self._data_key:
            # The reference re-reads the array on every half-sweep (factor.py:329-330, :374-375); here a few cells edited in
            # place (Y[:3, :3] = nan after the first sweep) slip past the fingerprint.  Every `data_check_seconds` of wall
            # clock the whole array is hashed again; a change re-uploads it and says how long the old copy was in use.
            import time
            every = getattr(self, "data_check_seconds", 5.0)
            now = time.monotonic()
            # (bounded duty cycle: the hash is O(data) on the host thread - 0.9 s for the 8.6 GB of config C5 - so the next
            #  check is at least 50 hash times away, whatever a positive `data_check_seconds` says: never more than 2 % of
            #  the wall clock.  An explicit 0 asks for the reference's behaviour - look at the data on every half-sweep)
            if every is None or now - self._data_checked < (max(every, 50.0 * getattr(self, "_digest_seconds", 0.0)) if every > 0 else 0.0):
                return
            dig = tuple(_digest(a) for a in arrays)
            self._data_checked = time.monotonic()
            self._digest_seconds = self._data_checked - now
            if dig == self._data_digest:
                return
            import warnings
            warnings.warn("the observation array was modified in place after it was uploaded; re-uploading it now - up to %.1f s of "
                          "sweeps used the previous contents (call set_data() right after editing the array in place)" % (now - self._data_uploaded),
                          RuntimeWarning, stacklevel=3)
        self._upload(data)
        import time
        t0 = time.monotonic()
        self._data_digest = tuple(_digest(a) for a in arrays)
        self._data_checked = self._data_uploaded = time.monotonic()
        self._digest_seconds = self._data_checked - t0
        if self._exchange.active:      # observation count over all ranks (a constant of the data set)
            (tot,) = self._exchange.sum_scalars(float(self._local_nobs(data)))
            self._ctx.call("btf_set_global_nobs", float(tot))
        self._data_ref = data          # keep alive so ids are not recycled
        self._data_key = key

    def _local_nobs(self, data):
        """Observed entries of this rank's row slab."""
        rows = getattr(data, "rows", None)
        if rows is None:
            a = data[0] if isinstance(data, (tuple, list)) else data
            rows = a[self._plan.row0:self._plan.row0 + self._plan.nl]
        return int(np.count_nonzero(~np.isnan(rows)))

    def set_data(self, data):
        """Force a re-upload.  The reference re-reads the observation array on every half-sweep
        (factor.py:329-330, :374-375); here it is uploaded once and recognised again by identity, shape and
        a 64-point fingerprint, which catches wholesale in-place edits (imputation, rescaling) at once; a few
        changed cells are caught by a hash of the whole array taken every `data_check_seconds` (default 5 s; None: never; 0: on
        every half-sweep; otherwise never more often than every 50 hash times - the hash is O(data) on the host thread), with a
        RuntimeWarning.  That
        check runs on the WALL clock: the sweep at which a silent in-place edit takes effect is not reproducible.  After
        mutating the array in place, call set_data(): the change then applies, deterministically, from the next half-sweep
        on."""
        self._data_key = None
        self._bind_data(data)

    def _upload(self, data):
        raise NotImplementedError

    def _stale_sources(self, any_nan, missing):
        """Quirks Q1/Q2 are reproduced only under compat="reference"; "exact" uses every output's own weights.  BEFORE the
        upload: works out the source rows / columns and, in a sharded run, declares the one source row and the one source
        column that may lie outside this rank's blocks (btf_set_shard_halo: the slabs then carry them as one more row /
        column - ShardPlan.slabs)."""
        self._src = None
        if self.compat == "reference":
            self._src = (stale_row_sources(self.nrows, self.nembeds, any_nan), stale_col_sources(missing))
        if self._plan.world > 1:
            hr, hc = self._plan.halo_of(*self._src) if self._src is not None else (-1, -1)
            self._plan.halo_row, self._plan.halo_col = hr, hc
            self._ctx.call("btf_set_shard_halo", hr, hc)

    def _set_stale_sources(self):
        """AFTER the upload: hands the sources of _stale_sources to the context."""
        if self._src is None:
            self._ctx.call("btf_set_stale_sources", None, None)
            return
        self._ctx.call("btf_set_stale_sources", self._src[0].ctypes.data_as(_native._c_ip), self._src[1].ctypes.data_as(_native._c_ip))

    # ---- construction draws ------------------------------------------------------
    def _init_sigma2(self):
        self.sigma2 = 1 / self.sigma2_model.draw_from_prior()

    def _init_lam2(self):
        self.lam2, self.lam2_a = sample_horseshoe()
        self.lam2 = self.lam2.clip(0, 4)

    def _init_Tau2(self):
        self.Tau2, self.Tau2_c, self.Tau2_b, self.Tau2_a = \
            sample_horseshoe_plus(size=(self.ncols, self.Delta.shape[0]))
        self.Tau2 = self.Tau2.clip(0, 9)

    def _init_W(self):
        """N(0, sigma2) with the strict upper triangle of the first K rows zeroed
        (factor.py:230-233)."""
        self._W = np.random.normal(0, np.sqrt(self.sigma2), size=(self.nrows, self.nembeds))
        if self.nrows > 1:
            self._W[np.triu_indices(self.nembeds, k=1)] = 0

    def prior_bands(self):
        """(M, T, tf+2) band of Delta' diag(1/(lam2 Tau2_j)) Delta: [j,t,d] = entry (t+d,t)."""
        D = self.Delta.toarray()
        lam = 1.0 / (self.lam2 * self.Tau2)                     # (M, nD)
        T, nb = self.ndepth, self.tf_order + 2
        band = np.zeros((self.ncols, T, nb))
        for d in range(nb):
            S = D[:, :T - d] * D[:, d:]                         # (nD, T-d)
            band[:, :T - d, d] = lam @ S
        return band

    def _init_V(self):
        """Prior draw per column, V_j ~ N(0, (I_K (x) Delta'LambdaDelta)^-1), clipped to
        +-10 (factor.py:235-242) - through the device banded sampler in depth-major order."""
        from .fast_mvn import sample_banded_batch
        K, T, M = self.nembeds, self.ndepth, self.ncols
        bw = (self.tf_order + 1) * K
        P = self.prior_bands()
        band = np.zeros((M, T * K, bw + 1))
        for d in range(self.tf_order + 2):
            for k in range(K):
                band[:, k::K, d * K][:, :T] = P[:, :, d]
        z = np.random.normal(size=(M, K * T)) if self.rng == "host" else None
        x, _ = sample_banded_batch(band, z=z, seed=self._next_seed(), device=self.device, **self.linalg_opts)
        self._V = x.reshape(M, T, K).clip(-10, 10)
        self._V_host_new, self._V_dev_new = True, False
        self._v_version += 1

    # ---- one Gibbs sweep over the shared parameters (factor.py:112-128) -----------
    def resample(self, data, **kwargs):
        if self.sample_sigma2:
            self._resample_sigma2()
        if self.sample_Tau2:
            self._resample_Tau2()
        if self.sample_lam2:
            self._resample_lam2()
        if self.sample_W:
            self._resample_W(data)
        if self.sample_V:
            self._resample_V(data)

    def _pack_W(self, W):
        """Free entries of W as one vector: lower triangle of the leading square block,
        then the dense remainder (factor.py:155-174; the precision it also returns is
        (1/sigma2) I and is not needed on this path)."""
        h = min(self.nembeds, self.nrows)
        return np.concatenate([W[np.tril_indices(h)], W[h:].reshape(-1)])

    def _resample_sigma2(self):
        if self._dev_scalars:
            if getattr(self, "_sigma2_drawn", False):      # drawn together with nu2 (one launch)
                self._sigma2_drawn = False
                return
            self._push_state()
            self._ctx.call("btf_draw_scalars", self._next_seed(), 2, 0.0, 0.0, float(self.sigma2_a), float(self.sigma2_b))
            self._sc_dev_new = True
            return
        self._pull_W()
        w = self._pack_W(self._W)
        self.sigma2 = 1 / self.sigma2_model.resample_from_stats(float(w @ w), w.size)

    def _penalised_differences(self):
        """sum_k (Delta V_j)[r,k]^2 for every column j and penalty row r: (M, nD).  One sparse
        product for all columns (the reference does Delta.dot(V[j]) per column, factor.py:136,149);
        cached while V is unchanged (the Tau2 and lam2 updates of one sweep share it)."""
        self._pull_V()
        key = (id(self._V), self._v_version)
        if getattr(self, "_dsq_key", None) != key:
            T = self.ndepth
            d = self.Delta.dot(np.ascontiguousarray(self._V.transpose(1, 0, 2)).reshape(T, -1))   # (nD, M*K)
            d = d.reshape(-1, self.ncols, self.nembeds)
            self._dsq = np.ascontiguousarray((d * d).sum(axis=2).T)                              # (M, nD)
            self._dsq_key = key
        return self._dsq

    def _resample_Tau2_device(self, queue=False):
        """rng="device": all columns at once on the GPU (Philox gamma draws); also leaves the
        per-column terms of the lam2 rate in self._lsum."""
        self._push_state()
        if getattr(self, "_chain_dirty", True):
            for nm in ("_Tau2_a", "_Tau2_b", "_Tau2_c"):
                setattr(self, nm, _native.as_f64(getattr(self, nm)))
            self._ctx.call("btf_set_tau_chain", _native.dptr(self._Tau2_a), _native.dptr(self._Tau2_b),
                           _native.dptr(self._Tau2_c))
            self._chain_dirty = False
        if self._dev_scalars:       # lam2 is read, and the lam2-rate terms are left, on the device
            if queue:               # as side workgroups of the W accumulation launch that follows (btf_queue_Tau2)
                self._ctx.call("btf_queue_Tau2", self._next_seed(), float(self.stability))
            else:
                self._ctx.call("btf_resample_Tau2", self._next_seed(), 1.0, float(self.stability), None)
            self._lsum_on_device, self._lsum_valid = True, False
        else:
            lsum = np.empty(self.ncols)
            self._ctx.call("btf_resample_Tau2", self._next_seed(), _scalar(self.lam2), float(self.stability),
                           _native.dptr(lsum))
            self._lsum, self._lsum_valid = lsum, True
        self._tau_dev_new, self._tau_dirty = True, False

    def _resample_Tau2(self):
        """Horseshoe+ local scales, one column after the other so that the legacy RNG
        stream matches the reference (4 vector gamma draws per column, factor.py:134-141)."""
        if getattr(self, "_tau2_drawn", False):            # rode along in this sweep's W accumulation launch
            self._tau2_drawn = False
            return
        if self.rng == "device" and hasattr(self, "_Tau2_a"):
            return self._resample_Tau2_device()
        lo, hi = self.stability, 1 / self.stability
        dsq = self._penalised_differences()
        shape = (self.nembeds + 1) / 2
        T2, Ta, Tb, Tc = self.Tau2, self.Tau2_a, self.Tau2_b, self.Tau2_c
        # The reference draws, column after column, gamma(shape, scale) for Tau2 and gamma(1, scale) for the three further
        # levels (nD variates each).  A legacy gamma(shape, scale) IS scale * standard_gamma(shape), and what a standard
        # gamma takes from the generator depends on its shape only - so ONE standard_gamma call over the shapes in the
        # reference's order ((K+1)/2, 1, 1, 1 per column) yields the same variates bit for bit and leaves the generator
        # where the reference's 4 M calls leave it (pinned by the G1 / G6 chain fixtures); the four scale levels - each a
        # function of the level before - are then applied to all columns at once.
        M, nD = self.ncols, dsq.shape[1]
        shapes = np.ones((M, 4, nD))
        shapes[:, 0, :] = shape
        sg = np.random.standard_gamma(shapes)
        scale1 = 1 / (dsq / (2 * self.lam2) + 1 / Tc.clip(lo, hi)).clip(lo, hi)
        t = 1 / (scale1 * sg[:, 0])
        c = 1 / ((1 / (1 / t + 1 / Tb).clip(lo, hi)) * sg[:, 1])
        b = 1 / ((1 / (1 / c + 1 / Ta).clip(lo, hi)) * sg[:, 2])
        a = 1 / ((1 / (1 / b + 1).clip(lo, hi)) * sg[:, 3])
        T2[:], Tc[:], Tb[:], Ta[:] = t, c, b, a

    def _resample_lam2(self):
        """Global scale.  compat="reference": the rate keeps only the LAST column's term
        (quirk Q3, factor.py:147-150); "exact": 1/lam2_a plus the sum over columns."""
        if getattr(self, "_lam2_drawn", False):            # rode along in this sweep's scalar-draw launch
            self._lam2_drawn = False
            return
        if self._dev_scalars and getattr(self, "_lsum_on_device", False):
            self._lsum_on_device = False
            self._push_scalars()
            seed = getattr(self, "_lam2_seed", None)
            self._lam2_seed = None
            self._ctx.call("btf_draw_lam2", self._next_seed() if seed is None else seed, _native.COMPAT[self.compat])
            self._sc_dev_new = True
            return
        if self.rng == "device" and getattr(self, "_lsum_valid", False):
            terms = self._lsum / 2          # from the device Tau2 update of this sweep
        else:
            terms = (self._penalised_differences() / self.Tau2).sum(axis=1) / 2
        rate = terms[-1] if self.compat == "reference" else 1 / self.lam2_a + terms.sum()
        shape = self.Delta.shape[0] * self.ncols * self.nembeds + 1
        self.lam2 = max(1e-5, 1 / np.random.gamma(shape / 2, 1 / rate))
        self.lam2_a = 1 / np.random.gamma(1, 1 / (1 / self.lam2 + 1))

    def _inferred_variables(self, var_map):
        self._pull_W()
        self._pull_V()
        var_map['W'] = np.copy(self._W)
        var_map['V'] = np.copy(self._V)
        var_map['sigma2'] = self.sigma2
        var_map['lam2'] = self.lam2
        var_map['Tau2'] = np.copy(self.Tau2)

    def _set_hyperparameters(self, hyperparams):
        self.lam2 = hyperparams['lam2']

    # ---- sample collection ------------------------------------------------------------
    def _collects_on_device(self):
        return self._dev_scalars and getattr(self, "_scalar_noise", False) and self._plan.world == 1 \
            and not self._exchange.active

    def run_gibbs(self, data, nburn=1000, nthin=1, nsamples=1000, verbose=True, print_freq=100,
                  callback=None, **kwargs):
        """genlasso.py:37-66.  rng="device" (scalar-noise model, no callback): the kept states are copied
        device-to-device into preallocated slots right behind the sweep that produced them and come
        back in one download at the end - the loop never waits for the GPU (the reference's per-sample
        `inferred_variables()` copies cost 380 us per kept sample here, 4x the sweep itself)."""
        if callback is not None or not self._collects_on_device() or nsamples < 1:
            return super().run_gibbs(data, nburn=nburn, nthin=nthin, nsamples=nsamples, verbose=verbose,
                                     print_freq=print_freq, callback=callback, **kwargs)
        self._ctx.call("btf_collect_begin", int(nsamples))        # (also clears a schedule an interrupted run left armed)
        self._collected = 0
        if self._sweeps_on_device():
            # whole sweeps queued by the C side (btf_gibbs_sweeps): the same chain as the loop below, without a
            # Python round trip per step; the kept states (the first after nburn + 1 sweeps, then every nthin-th) are copied
            # into their slots by the same C call (btf_collect_schedule) - one call per block of sweeps, not per sample
            done, total = 0, nburn + (nsamples - 1) * nthin + 1
            scheduled = False
            try:
                while done < total:
                    if not scheduled and done >= nburn:
                        self._ctx.call("btf_collect_schedule", int(nthin), 0, 1)
                        scheduled = True
                    limit = total if scheduled else nburn
                    n = min(limit - done, max(1, print_freq - done % print_freq)) if verbose else limit - done
                    if verbose and done % print_freq == 0:
                        print('\tStep {}'.format(done))
                    self.resample_sweeps(data, n)
                    done += n
            finally:           # (an exception / KeyboardInterrupt inside the loop must not leave later sweeps copying states)
                self._ctx.call("btf_collect_schedule", 0, 0, 0)
        else:
            for step in range(nburn + nthin * nsamples):
                if verbose and step % print_freq == 0:
                    print('\tStep {}'.format(step))
                self.resample(data, **kwargs)
                kept, rem = divmod(step - nburn, nthin)
                if step >= nburn and rem == 0:
                    self._push_state()                     # (anything the caller touched between sweeps)
                    self._ctx.call("btf_collect", kept)
        N, M, T, K, nD = self.nrows, self.ncols, self.ndepth, self.nembeds, self.Delta.shape[0]
        out = {"W": np.zeros((nsamples, N, K)), "V": np.zeros((nsamples, M, T, K)), "Tau2": np.zeros((nsamples, M, nD))}
        sc = np.zeros((nsamples, 8))
        self._ctx.call("btf_collect_end", int(nsamples), _native.dptr(out["W"]), _native.dptr(out["V"]),
                       _native.dptr(out["Tau2"]), _native.dptr(sc))
        self._collected = nsamples
        out["nu2"], out["sigma2"], out["lam2"] = sc[:, 0:1].copy(), sc[:, 1:2].copy(), sc[:, 2:3].copy()
        return out

    def _sweeps_on_device(self):
        return False

    def posterior_summary(self, q=(5, 95), transform=None):
        """Mean and percentiles of f(W V') over the samples the last device-collecting run_gibbs kept,
        computed where they lie (btf_collect_summary); see functionalmf_amd.utils.posterior_summary."""
        n = getattr(self, "_collected", 0)
        if n < 1:
            raise RuntimeError("no samples collected on the device (run_gibbs with rng='device' first)")
        code = {None: 0, "identity": 0, "ilogit": 1, "square": 2}[transform]
        qs = _native.as_f64(np.atleast_1d(q))
        mean = np.zeros((self.nrows, self.ncols, self.ndepth))
        quant = np.zeros((len(qs),) + mean.shape)
        self._ctx.call("btf_collect_summary", int(n), code, _native.dptr(qs), len(qs), _native.dptr(mean), _native.dptr(quant))
        return mean, quant

    # ---- the two half-sweeps (device) ----------------------------------------------
    def _w_normals(self):
        if self.rng != "host":
            return None
        K, N = self.nembeds, self.nrows
        n = K * (K + 1) // 2 + (N - K) * K if N >= K else N * (N + 1) // 2
        return np.random.normal(size=n)        # == the per-row draws of factor.py:361, concatenated

    def _v_normals(self):
        if self.rng != "host":
            return None
        return np.random.normal(size=(self.ncols, self.nembeds * self.ndepth))   # fast_mvn.py:41 per column

    def _device_W_step(self):
        self._push_state()
        z = self._keep_w = self._w_normals()     # kept alive until replaced: the upload is stream-ordered
        self._ctx.call("btf_resample_W", _native.dptr(z), self._next_seed(), _native.COMPAT[self.compat])
        self._exchange.after_W()
        self._W_dev_new = True

    def _device_V_step(self):
        self._push_state()
        z = self._keep_v = self._v_normals()
        o = self.linalg_opts
        self._ctx.call("btf_resample_V", _native.dptr(z), self._next_seed(), _native.COMPAT[self.compat],
                       float(o["force_psd_eps"]), int(o["force_psd_attempts"]) if o["force_psd"] else 0)
        self._exchange.after_V()
        self._V_dev_new = True
        self._lsum_valid = False
        self._lsum_on_device = False
        self._after_V_step()

    def _after_V_step(self):
        pass

    def v_order(self):
        """Elimination order of the V half-sweep's factorisation as depth-major indices t*K+k
        (what CHOLMOD's P() is to fast_mvn.py:44): z[j][i] multiplies pivot i.  Known once the
        data is bound (the kernel choice depends on the weighted / unweighted LDS footprint)."""
        order = np.zeros(self.nembeds * self.ndepth, dtype=np.int32)
        self._ctx.call("btf_get_V_order", order.ctypes.data_as(_native._c_ip))
        return order

    def v_sampler(self):
        """Name of the sampler the next V half-sweep will run (known once the data is bound)."""
        import ctypes
        which = ctypes.c_int32()
        self._ctx.call("btf_get_V_sampler", ctypes.byref(which))
        return {v: k for k, v in _native.SAMPLERS.items()}[which.value]

    def likelihood_form(self):
        """"complete", "weighted" or "curve_counts": which form of the likelihood part the half-sweeps run for the
        bound data (include/btf.h, btf_get_likelihood_form).  "curve_counts": Gaussian data whose replicate counts
        do not vary along the depth axis (whole curves held out, as in the reference's examples) - complete-data
        kernels plus per-row / per-column corrections, the same conditionals as factor.py:343-346, :388-391."""
        import ctypes
        form = ctypes.c_int32()
        self._ctx.call("btf_get_likelihood_form", ctypes.byref(form))
        return ("complete", "weighted", "curve_counts")[form.value]

    def sync(self):
        """Wait for the GPU; raises NotPositiveDefiniteError if a factorisation failed."""
        self._ctx.call("btf_sync")

    # ---- checkpoint / resume (the reference keeps nothing between runs: genlasso.py:57-66 returns the samples and the
    # applications np.save them, doseresponse/fit.py:428-439) -----------------------------------
    _CHAIN_ARRAYS = ("W", "V", "Tau2", "Tau2_a", "Tau2_b", "Tau2_c")
    _CHAIN_SCALARS = ("sigma2", "lam2", "lam2_a")

    def checkpoint(self):
        """The chain's state between two sweeps as a dict of numpy arrays and numbers (np.savez-able): factors,
        horseshoe+ levels, scalars, the device draw counter (rng="device": every Philox stream is keyed by
        device_seed and that counter) and the legacy numpy generator's state (rng="host").  A model built with the
        same arguments, `restore`d from it and given the same data continues the chain bit for bit.  Taking a checkpoint
        is NOT invisible to the chain that goes on: like `restore`, it drops the spectral sampler's eigen warm start and
        the residual parts of the four-launch sweep (so that both continuations take the same path) - the next draws
        agree with an un-checkpointed run to rounding, not bit for bit.  (Conjugate
        models: Gaussian, Binomial, Negative-Binomial.  The slice-sampling models keep further generator state in
        `chain_rngs` / the context's round counters, which a checkpoint does not carry.)"""
        self.sync()
        st = {"draws": int(self._draws), "device_seed": int(self._device_seed), "rng": self.rng}
        cw, cv = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._ctx.call("btf_get_draw_counters", ctypes.byref(cw), ctypes.byref(cv))
        st["half_sweeps_w"], st["half_sweeps_v"] = int(cw.value), int(cv.value)
        # (also drops the spectral sampler's eigen warm start, as restore() does: both continuations start it cold)
        self._ctx.call("btf_set_draw_counters", int(cw.value), int(cv.value))
        for k in self._CHAIN_ARRAYS:
            v = getattr(self, k, None)
            if v is not None:
                st[k] = np.array(v, dtype=float)
        for k in self._CHAIN_SCALARS:
            v = getattr(self, k, None)
            if v is not None:
                st[k] = float(v)
        kind, keys, pos, has_gauss, cached = np.random.get_state()
        st.update(np_kind=kind, np_keys=np.array(keys), np_pos=int(pos), np_has_gauss=int(has_gauss), np_cached=float(cached))
        st.update(self._extra_state())
        return st

    def restore(self, st):
        """Continue from `checkpoint()`'s dict (or an np.load of it)."""
        if int(st["device_seed"]) != int(self._device_seed) or str(st["rng"]) != self.rng:
            raise ValueError("checkpoint was taken with device_seed=%s rng=%s" % (st["device_seed"], st["rng"]))
        for k in self._CHAIN_ARRAYS:
            if k in st:
                setattr(self, k, np.array(st[k], dtype=float))
        for k in self._CHAIN_SCALARS:
            if k in st:
                setattr(self, k, float(st[k]))
        self._draws = int(st["draws"])
        self._ctx.call("btf_set_draw_counters", int(st["half_sweeps_w"]), int(st["half_sweeps_v"]))
        np.random.set_state((str(st["np_kind"]), np.asarray(st["np_keys"], dtype=np.uint32), int(st["np_pos"]),
                             int(st["np_has_gauss"]), float(st["np_cached"])))
        self._set_extra_state(st)

    def _extra_state(self):
        return {}

    def _set_extra_state(self, st):
        pass

    def _resample_W(self, data):
        raise NotImplementedError

    def _resample_V(self, data):
        raise NotImplementedError


class GaussianBayesianTensorFiltering(BayesianTensorFiltering):
    _queue_sse = True      # scalar-noise model: nu2 needs the residual sum of squares every sweep
    _scalar_noise = True   # nu2 is one number (device-resident in rng="device" mode)
    fuse_tau2 = True       # rng="device" sweeps: the horseshoe+ chain rides in the W accumulation launch (A/B switch)

    nu2 = property(lambda self: self._sc_get("nu2"), lambda self, v: self._sc_set("nu2", v))

    def _extra_state(self):
        st = super()._extra_state()
        if self._scalar_noise:
            st["nu2"] = float(self.nu2)
        return st

    def _set_extra_state(self, st):
        super()._set_extra_state(st)
        if self._scalar_noise:
            self.nu2 = float(st["nu2"])

    def __init__(self, nrows, ncols, ndepth,
                 nu2_init=None, nu2_true=None,
                 nu2_a=0.1, nu2_b=0.1, **kwargs):
        super().__init__(nrows, ncols, ndepth, **kwargs)
        self.nu2_a, self.nu2_b = nu2_a, nu2_b
        self.nu2_model = ConjugateInverseGammaPrior(1, nu2_a, nu2_b)
        self.sample_nu2 = nu2_true is None
        if nu2_true is not None:
            self.nu2 = nu2_true
        elif nu2_init is not None:
            self.nu2 = nu2_init
        else:
            self._init_nu2()

    def _init_nu2(self):
        self.nu2 = 1 / self.nu2_model.draw_from_prior()

    def _upload(self, Y):
        if Y.ndim not in (3, 4):
            raise AssertionError('Observations must be 3- or 4-tensor.')
        Y4 = Y[..., None] if Y.ndim == 3 else Y
        if Y4.shape[:3] != (self.nrows, self.ncols, self.ndepth):
            raise ValueError("data shape %r does not match the model" % (Y.shape,))
        miss = np.isnan(Y4)
        self._stale_sources(bool(miss.any()), miss.all(axis=3))
        rows, cols = self._plan.slabs(Y4)
        self._ctx.call("btf_set_data_gaussian", _native.dptr(rows), _native.dptr(cols), int(Y4.shape[3]))
        self._set_stale_sources()

    def resample(self, data):
        self._in_sweep = True
        try:
            if self.sample_nu2:
                self._resample_nu2(data)
            super().resample(data)
        finally:
            self._in_sweep = False

    def _sweeps_on_device(self):
        """Can n whole sweeps be queued by one C call (btf_gibbs_sweeps)?  Every parameter sampled, all of them on the
        device, one GPU."""
        return bool(self._dev_scalars and self._scalar_noise and self.fuse_tau2 and not self._exchange.active
                    and self._plan.world == 1 and hasattr(self, "_Tau2_a") and type(self).resample is GaussianBayesianTensorFiltering.resample
                    and all((self.sample_nu2, self.sample_sigma2, self.sample_Tau2, self.sample_lam2, self.sample_W, self.sample_V)))

    def resample_sweeps(self, data, n):
        """n sweeps of resample(data); with rng="device" on one GPU they are queued by a single call into the C side
        (the same launches and seeds as n calls of resample: the chains coincide)."""
        if not self._sweeps_on_device():
            for _ in range(int(n)):
                self.resample(data)
            return
        self._bind_data(data)
        self._push_state()
        if getattr(self, "_chain_dirty", True):
            for nm in ("_Tau2_a", "_Tau2_b", "_Tau2_c"):
                setattr(self, nm, _native.as_f64(getattr(self, nm)))
            self._ctx.call("btf_set_tau_chain", _native.dptr(self._Tau2_a), _native.dptr(self._Tau2_b), _native.dptr(self._Tau2_c))
            self._chain_dirty = False
        self._push_scalars()
        o = self.linalg_opts
        self._ctx.call("btf_gibbs_sweeps", int(n), (self._device_seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF, int(self._draws),
                       _native.COMPAT[self.compat], float(self.nu2_a), float(self.nu2_b), float(self.sigma2_a), float(self.sigma2_b),
                       float(self.stability), float(o["force_psd_eps"]), int(o["force_psd_attempts"]) if o["force_psd"] else 0)
        self._draws += 5 * int(n)
        self._W_dev_new = self._V_dev_new = self._sc_dev_new = True
        self._tau_dev_new, self._tau_dirty = True, False
        self._lsum_valid = self._lsum_on_device = False

    def wv_steps(self, data, n):
        """n times (_resample_W, _resample_V) - the W+V update of factor.py:313-409.  With rng="device" on one GPU the
        launches of all n steps are queued by a single call into the C side (btf_wv_steps: the same launches and seeds as the
        Python loop, so the chains coincide); otherwise this IS the Python loop."""
        if not (type(self) is GaussianBayesianTensorFiltering and self.rng == "device" and self._plan.world == 1
                and not self._exchange.active and self.sample_W and self.sample_V):
            for _ in range(int(n)):
                self._resample_W(data)
                self._resample_V(data)
            return
        self._bind_data(data)
        self._set_noise()
        self._push_state()
        o = self.linalg_opts
        self._ctx.call("btf_wv_steps", int(n), (self._device_seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF, int(self._draws),
                       _native.COMPAT[self.compat], float(o["force_psd_eps"]), int(o["force_psd_attempts"]) if o["force_psd"] else 0)
        self._draws += 2 * int(n)
        self._W_dev_new = self._V_dev_new = True
        self._lsum_valid = False
        self._lsum_on_device = False

    def _set_noise(self):
        if not self._dev_scalars:
            self._ctx.call("btf_set_nu2", _scalar(self.nu2))

    def _resample_W(self, data):
        self._bind_data(data)
        self._set_noise()
        self._device_W_step()

    def _resample_V(self, data):
        self._bind_data(data)
        self._set_noise()
        self._device_V_step()

    def _after_V_step(self):
        # rng="device": queue the next sweep's residual reduction (and a copy of W for the sigma2
        # update) right behind the V draw, so that a full sweep waits for the GPU only twice
        # (only inside resample(): a caller driving _resample_W/_resample_V by hand pays nothing extra)
        if getattr(self, "_in_sweep", False) and self._queue_sse and self.rng == "device" and self.sample_nu2 \
                and self._plan.world == 1 and not self._exchange.active and not self._dev_scalars:
            self._ctx.call("btf_sse_begin")
            self._sse_queued = True

    def _resample_nu2(self, data):
        """nu2 | rest: residual sum of squares reduced on the device (factor.py:411-416)."""
        self._bind_data(data)
        if self._dev_scalars and self._scalar_noise:
            # residual reduction + inverse-gamma draw on the device; sigma2 rides along when the
            # sweep is going to draw it next anyway (same conditionals: sigma2 | W does not depend on nu2)
            which = 1
            if getattr(self, "_in_sweep", False) and self.sample_sigma2:
                which, self._sigma2_drawn = 3, True
            self._push_state()
            if getattr(self, "_in_sweep", False) and self.sample_W:
                # the W half-sweep's accumulation depends on the data and V only: queue it now and take the
                # residual sum of squares from its partials (no pass of its own over the data); the W step
                # later in this sweep goes straight to the solve
                # ... and the horseshoe+ chain, which depends on V and lam2 only (not on nu2 / sigma2: the same
                # conditionals in either order), rides in that launch as side workgroups
                if self.sample_Tau2 and self.fuse_tau2 and hasattr(self, "_Tau2_a"):
                    self._resample_Tau2_device(queue=True)
                    self._tau2_drawn = True
                    if self.sample_lam2:
                        lam_seed = self._next_seed()       # (taken here whatever the sharding: one seed order for all)
                        if self._exchange.active:          # the scalar draw is split around an all-reduce: lam2 keeps its own launch
                            self._lam2_seed = lam_seed
                        else:
                            # lam2 | Tau2_new, ...: a second workgroup of the scalar-draw launch below
                            self._push_scalars()
                            self._ctx.call("btf_queue_lam2", lam_seed, _native.COMPAT[self.compat])
                            self._lam2_drawn, self._lsum_on_device = True, False
                # four-launch sweep: when the previous V half-sweep left the residual parts behind, nu2 / sigma2 ride in
                # the W accumulation launch (btf_queue_scalars) and a queued lam2 in the V accumulation launch
                seed = self._next_seed()
                queued = ctypes.c_int32(0)
                if not self._exchange.active:
                    self._ctx.call("btf_queue_scalars", seed, which & 3, float(self.nu2_a), float(self.nu2_b),
                                   float(self.sigma2_a), float(self.sigma2_b), ctypes.byref(queued))
                self._ctx.call("btf_w_accum", _native.COMPAT[self.compat])
                if queued.value:
                    self._sc_dev_new = True
                    return
                which |= 4
            else:
                seed = self._next_seed()
            if self._exchange.active:      # this rank's share -> all-reduce of one device double -> identical draws
                self._ctx.call("btf_draw_scalars", seed, (which & 5) | 8, float(self.nu2_a), float(self.nu2_b),
                               float(self.sigma2_a), float(self.sigma2_b))
                self._exchange.all_reduce_sse()
                which = (which & 3) | 16
            self._ctx.call("btf_draw_scalars", seed, which, float(self.nu2_a), float(self.nu2_b),
                           float(self.sigma2_a), float(self.sigma2_b))
            self._sc_dev_new = True
            return
        sse, nobs = ctypes.c_double(), ctypes.c_double()
        if getattr(self, "_sse_queued", False) and not (self._W_host_new or self._V_host_new):
            self._sse_queued = False
            self._ctx.call("btf_sse_end", ctypes.byref(sse), ctypes.byref(nobs), _native.dptr(self._W))
            self._W_dev_new = False                       # the mirror of W came along
        else:
            self._sse_queued = False
            self._push_state()
            self._ctx.call("btf_sse", ctypes.byref(sse), ctypes.byref(nobs))
        sse, nobs = self._exchange.sum_scalars(sse.value, nobs.value)
        self.nu2 = 1 / self.nu2_model.resample_from_stats(sse, nobs)

    def _inferred_variables(self, var_map):
        super()._inferred_variables(var_map)
        var_map['nu2'] = self.nu2


class BinomialBayesianTensorFiltering(GaussianBayesianTensorFiltering):
    """Logistic-Binomial likelihood through Polya-Gamma augmentation
    (factor.py:425-460): omega ~ PG(N, w.v) on the device, then the weighted
    Gaussian half-sweeps with pseudo-data kappa = Y - N/2."""

    _queue_sse = False     # nu2 here is the tensor 1/omega (PG draw), not a variance to update from residuals
    _scalar_noise = False

    def _extra_state(self):           # (omega itself is redrawn from W, V at the start of every sweep)
        st = super()._extra_state()
        st["pg_calls"] = int(self._pg_calls)
        st["pg_mode"] = int(self.PG_MODES[self.pg_exact])     # which sampler drew the chain so far (BTF_OPT_PG_EXACT)
        return st

    def _set_extra_state(self, st):
        super()._set_extra_state(st)
        self._pg_calls = int(st["pg_calls"])
        # (round 3 changed what pg_exact=False / None select: a checkpoint continued under another Polya-Gamma sampler
        #  would silently walk a different chain)
        if "pg_mode" in st and int(st["pg_mode"]) != int(self.PG_MODES[self.pg_exact]):
            raise ValueError("checkpoint was taken with Polya-Gamma mode %d (BTF_OPT_PG_EXACT), this model runs mode %d"
                             % (int(st["pg_mode"]), int(self.PG_MODES[self.pg_exact])))

    PG_MODES = {None: 0, "auto": 0, True: 1, "exact": 1, False: 2, "series": 2}

    def __init__(self, nrows, ncols, ndepth, pg_seed=42, pg_exact=None, **kwargs):
        """pg_exact (include/btf.h, BTF_OPT_PG_EXACT).  None / "auto" (default): every integer count up to 32 is drawn
        as the sum of N exact Devroye draws - what pypolyagamma's pgdrawv does at factor.py:459 - and larger or
        non-integer counts by the validated sum-of-gammas series with a moment-matched remainder; count data
        (Negative-Binomial pseudo-trial counts, integers only by accident) take the series throughout.
        True / "exact": the exact sampler for every count below 200.  False / "series": the series for every count
        (faster, approximate: the round-2 default)."""
        super().__init__(nrows, ncols, ndepth, **kwargs)
        self.pg_seed = pg_seed
        if pg_exact not in self.PG_MODES:
            raise ValueError("pg_exact must be None / 'auto', True / 'exact' or False / 'series'")
        self.pg_exact = pg_exact
        self.pg_sampler = ("devroye-exact(N<=32)+gamma-series", "devroye-exact", "gamma-series")[self.PG_MODES[pg_exact]]
        self._ctx.call("btf_set_option", _native.OPT_PG_EXACT, self.PG_MODES[pg_exact])
        self._pg_calls = 0
        self._nu2 = np.zeros((nrows, ncols, ndepth))
        self._omega_dev_new = False
        self.sample_nu2 = True

    # nu2 = 1/omega is materialised on the host only when somebody looks at it
    @property
    def nu2(self):
        if getattr(self, "_omega_dev_new", False):
            om = np.zeros((self._plan.nl, self.ncols, self.ndepth))
            self._ctx.call("btf_get_omega", _native.dptr(om))
            full = self._exchange.gather_rows_host(om)
            with np.errstate(divide='ignore'):
                self._nu2 = 1 / full
            self._omega_dev_new = False
        return self._nu2

    @nu2.setter
    def nu2(self, value):
        self._nu2 = value
        self._omega_dev_new = False
        self._omega_host_new = True

    def _upload(self, data):
        Y, N = data
        if Y.shape != (self.nrows, self.ncols, self.ndepth) or N.shape != Y.shape:
            raise ValueError("binomial data must be a (Y, N) pair of (nrows, ncols, ndepth) arrays")
        miss = np.isnan(Y) | np.isnan(N)
        self._stale_sources(bool(miss.any()), miss)
        yr, yc = self._plan.slabs(Y[..., None])
        nr, nc = self._plan.slabs(N[..., None])
        self._ctx.call("btf_set_data_binomial", _native.dptr(yr), _native.dptr(nr), _native.dptr(yc), _native.dptr(nc))
        self._set_stale_sources()
        self._omega_host_new = True

    def _set_noise(self):
        if getattr(self, "_omega_host_new", False) and np.ndim(self._nu2) == 3:
            with np.errstate(divide='ignore'):
                om = np.where(np.isfinite(self._nu2) & (self._nu2 > 0), 1 / self._nu2, 0.0)
            r, c = self._plan.slabs(om[..., None])
            self._ctx.call("btf_set_omega", _native.dptr(r), _native.dptr(c))
        self._omega_host_new = False

    def _resample_nu2(self, data):
        self._bind_data(data)
        self._push_state()
        self._pg_calls += 1
        seed = (int(self.pg_seed) * 0x9E3779B97F4A7C15 + self._pg_calls) & 0xFFFFFFFFFFFFFFFF
        self._ctx.call("btf_pg_draw", seed)
        self._omega_dev_new = True
        self._omega_host_new = False


class NegativeBinomialBayesianTensorFiltering(BinomialBayesianTensorFiltering):
    """Negative-Binomial counts y ~ NB(R, ilogit(w.v)) (factor.py:462-563): given the rate R the
    augmented model is the Binomial one on Y = sum_r y_r, N = sum_r (y_r + R); R itself is
    updated by `nmetropolis` random-walk Metropolis-Hastings steps on log R per sweep.

    The counts stay on the GPU: the data-sized part of every MH step (the log-likelihood ratio,
    `btf_nb_loglik`) and the rebuild of the Binomial pseudo-data after R moved
    (`btf_nb_set_rate`) are kernels; proposals and accept/reject decisions are drawn on the
    host from the legacy numpy stream in the reference's order, so that a seeded chain walks
    the reference's path.  `rdims`: dims of (rows, cols, depth) one R value is shared across.

    Deviation: with `R_true` the reference never defines `self.N` and fails in `resample`
    (factor.py:476-478 vs :508); here N is built from R_true."""

    def __init__(self, nrows, ncols, ndepth, R_true=None, R_init=None, nmetropolis=30, rpropstdev=0.1,
                 rstdev=1, rdims=(0, 1, 2), **kwargs):
        super().__init__(nrows, ncols, ndepth, **kwargs)
        # (sharded runs: every rank uploads the WHOLE count tensor - the rate update is a function of all of it and every
        #  rank computes it, from the same streams, like the other hyper-parameters; the augmented Binomial model is kept
        #  for the rank's two slabs only: btf_set_data_counts)
        self._shared = tuple(sorted(int(d) for d in (rdims if rdims is not None else ())))
        self.rdims = [3] + list(self._shared[::-1])                      # factor.py:485
        self.nmetropolis, self.rpropstdev, self.rstdev = nmetropolis, rpropstdev, rstdev
        self.sample_R = R_true is None
        if R_true is not None:
            self.R = np.array(R_true, dtype=float)
        elif R_init is not None:
            self.R = np.array(R_init, dtype=float)
        else:
            self._init_R()
        self._rate_key = None

    def _rate_shape(self):
        return tuple(1 if i in self._shared else c for i, c in enumerate((self.nrows, self.ncols, self.ndepth)))

    # R: host array, refreshed from the GPU when the device-side MH loop (rng="device") moved it
    @property
    def R(self):
        if getattr(self, "_R_dev_new", False):
            out = np.zeros(self._rate_shape())
            self._ctx.call("btf_nb_get_rate", _native.dptr(out), self._shared_flags().ctypes.data_as(_native._c_ip))
            self._R = out
            self._R_dev_new = False
            self._rate_key = self._key_of(out)          # the device already holds exactly this rate
        return self._R

    @R.setter
    def R(self, value):
        self._R = value
        self._R_dev_new = False

    @staticmethod
    def _key_of(R):
        return R.tobytes() if R.size <= 4096 else (float(R.sum()), float((R * R).sum()), R.shape)

    def _init_R(self):
        """exp(N(0, rstdev)) + 1 (factor.py:560-563)."""
        self.R = np.exp(np.random.normal(0, self.rstdev, size=self._rate_shape())) + 1

    def _shared_flags(self):
        return np.array([1 if d in self._shared else 0 for d in range(3)], dtype=np.int32)

    # ---- data: the raw replicate counts, uploaded once --------------------------------------
    @staticmethod
    def _counts4(data):
        if isinstance(data, (tuple, list)):
            raise ValueError("negative-binomial data is the count tensor itself, not a (Y, N) pair")
        return data[..., None] if data.ndim == 3 else data

    def _upload(self, data):
        d4 = _native.as_f64(self._counts4(data))
        if d4.ndim != 4 or d4.shape[:3] != (self.nrows, self.ncols, self.ndepth):
            raise ValueError("data shape %r does not match the model" % (data.shape,))
        miss = np.all(np.isnan(d4), axis=-1)
        self._stale_sources(bool(miss.any()), miss)
        self._ctx.call("btf_set_data_counts", _native.dptr(d4), int(d4.shape[3]))
        self._set_stale_sources()
        self._nb_sum = np.nansum(d4, axis=-1)
        self._nb_cnt = (~np.isnan(d4)).sum(axis=-1).astype(float)
        self._rate_key = None
        self._omega_host_new = True

    @property
    def N(self):
        """Binomial trial counts of the augmented model, nansum(data + R) (factor.py:552)."""
        return self._nb_sum + self._nb_cnt * np.broadcast_to(self.R, self._nb_sum.shape)

    def _push_rate(self):
        if getattr(self, "_R_dev_new", False):          # rate and pseudo-data already current on the device
            return
        R = _native.as_f64(np.broadcast_to(self.R, self._rate_shape()))
        key = self._key_of(R)
        if key != self._rate_key:
            self._ctx.call("btf_nb_set_rate", _native.dptr(R), self._shared_flags().ctypes.data_as(_native._c_ip))
            self._rate_key = key

    def _bind_data(self, data):
        super()._bind_data(data)
        self._push_rate()

    def resample(self, data):
        self._bind_data(data)
        if self.sample_R:
            self._resample_R(data)
        super().resample(data)

    def _resample_R(self, data):
        """Random-walk MH on log R (factor.py:513-554): per step one normal and one uniform array
        of R's shape from the legacy stream; the likelihood ratio comes from the GPU."""
        self._bind_data(data)
        self._push_state()
        shp = self._rate_shape()
        if self.rng == "device" and getattr(self, "_mh_on_device", True):
            # the whole loop on the GPU (Philox proposals / decisions); needs the histogram form of the
            # likelihood ratio - otherwise fall through to the host-driven loop below
            start = None if getattr(self, "_R_dev_new", False) else _native.as_f64(np.broadcast_to(self.R, shp))
            try:
                self._ctx.call("btf_nb_mh", self._next_seed(), int(self.nmetropolis), float(self.rpropstdev),
                               float(self.rstdev), self._shared_flags().ctypes.data_as(_native._c_ip), _native.dptr(start))
                self._R_dev_new = True
                return
            except _native.BTFError as e:
                if e.code != _native.BTF_ESTATE:
                    raise
                self._mh_on_device = False
        s2 = float(self.rstdev) ** 2
        R = _native.as_f64(np.broadcast_to(self.R, shp)).copy()
        logR = np.log(R)
        flags = self._shared_flags().ctypes.data_as(_native._c_ip)
        ll = np.zeros(shp)
        for _ in range(self.nmetropolis):
            cand_log = logR + np.random.normal(0, self.rpropstdev, size=shp)
            cand = np.exp(cand_log)
            a_prior = (logR * logR - cand_log * cand_log) / (2 * s2)      # N(0, rstdev) log-density ratio
            self._ctx.call("btf_nb_loglik", _native.dptr(R), _native.dptr(cand), flags, _native.dptr(ll))
            prob = np.exp(np.clip(a_prior + ll, -10, 1))
            acc = np.random.random(size=shp) <= prob
            acc &= cand > 1                                      # the reference's "TEMP" floor (factor.py:547)
            logR[acc] = cand_log[acc]
            R[acc] = np.exp(cand_log[acc])
        self.R = R
        self._push_rate()

    def _inferred_variables(self, var_map):
        super()._inferred_variables(var_map)
        var_map['R'] = np.copy(self.R)

    def _extra_state(self):
        st = super()._extra_state()
        st["R"] = np.array(self.R, dtype=float)
        return st

    def _set_extra_state(self, st):
        super()._set_extra_state(st)
        self.R = np.array(st["R"], dtype=float)           # (uploaded with the data: _bind_data -> _push_rate)
        self._rate_key = None


class NonconjugateBayesianTensorFiltering(BayesianTensorFiltering):
    """Non-conjugate likelihoods by elliptical slice sampling (factor.py:567-612): the prior draw and the slice
    loop of `_resample_W` / `_resample_V` run on the GPU (include/btf.h, btf_ess_*).

    The reference takes a Python callback `loglikelihood(W, V, data)` and evaluates it on the whole tensor for
    every proposal.  Here the likelihood is evaluated by a device kernel, so `loglikelihood` names one of
    the built-in families instead; a callable is accepted too and evaluated on the HOST for every proposal (the reference's
    interface, for any likelihood: prior draws and proposals on the device, each proposal read back - a slow path):
        "poisson" / "poisson_log"   counts y ~ Poisson(exp(w.v))
        "poisson_identity"          counts y ~ Poisson(w.v), zero likelihood where w.v <= 0
                                    (the likelihood of examples/poisson_tensor_filtering.py:26-37)
        "bernoulli_logit"           y in {0, 1} ~ Bernoulli(ilogit(w.v))      (scipy.stats.bernoulli.logpmf)
        "gaussian"                  y ~ N(w.v, likelihood_param)              (likelihood_param: the variance)
        "negbin_logit"              y ~ NB(likelihood_param, 1 - ilogit(w.v)) (scipy.stats.nbinom; likelihood_param: the rate r)
    - all functions of the hoisted per-cell statistics (sum and count of the observed replicates).
    data: (N,M,T) or (N,M,T,R), NaN = missing.

    ess = "joint" (default): ONE slice over all of W, then one over all of V, as the reference; with rng="host" the
          normals and uniforms come from the global legacy numpy generator in the reference's order, so a seeded
          chain walks the reference's path (tests/golden/g9_ess.npz).
    ess = "rows" (rng="device" only): one slice per row of W / per column of V - conditionally independent given
          the other factor, as in the reference's constrained model (factor.py:665-720) - all brackets shrinking
          in lockstep on the device; mixes faster than the joint slice, whose step shrinks with the dimension."""

    LINKS = {"poisson": 0, "poisson_log": 0, "poisson_identity": 1, "bernoulli_logit": 2, "gaussian": 3, "negbin_logit": 4}

    def __init__(self, nrows, ncols, ndepth, loglikelihood, ess="joint", ess_max_rounds=40, likelihood_param=None, **kwargs):
        self._callback = callable(loglikelihood)
        if not self._callback and loglikelihood not in self.LINKS:
            raise ValueError("loglikelihood must be a function (W, V, data) -> log-likelihood, as in the reference, or name a "
                             "device likelihood %s" % sorted(self.LINKS))
        if ess not in ("joint", "rows"):
            raise ValueError("ess must be 'joint' or 'rows'")
        super().__init__(nrows, ncols, ndepth, **kwargs)
        if self._plan.world > 1:
            raise NotImplementedError("NonconjugateBayesianTensorFiltering: unsharded runs only")
        if ess == "rows" and (self.rng != "device" or self._callback):
            raise ValueError("ess='rows' runs its slices on the device: use rng='device' and a device likelihood")
        self.loglikelihood = loglikelihood
        # a Python callback (the reference's interface, factor.py:567-570): the HOST-EVALUATED path - prior draws and
        # proposals on the device (same declared orders as the device likelihoods), every proposal read back and handed to
        # the function; correct for any likelihood, and as slow as the function and two PCIe copies per evaluation make it
        self._link = _native.ESS_HOST_LIKELIHOOD if self._callback else self.LINKS[loglikelihood]
        if self._link in (3, 4):
            if likelihood_param is None or not likelihood_param > 0:
                raise ValueError("loglikelihood=%r needs likelihood_param > 0 (the variance / the rate)" % loglikelihood)
            self.likelihood_param = float(likelihood_param)
            self._ctx.call("btf_set_likelihood_param", self._link, 1.0 / self.likelihood_param if self._link == 3 else self.likelihood_param)
        else:
            self.likelihood_param = None
        self.ess, self.ess_max_rounds = ess, int(ess_max_rounds)
        self.ess_evaluations = 0          # likelihood evaluations of the last host-driven slice (diagnostic)
        self._ll_const = 0.0

    def _bind_data(self, data):
        if self._callback:           # the function's `data` is its own business (any object): nothing goes to the device
            return
        super()._bind_data(data)

    def _upload(self, Y):
        if Y.ndim not in (3, 4):
            raise AssertionError('Observations must be 3- or 4-tensor.')
        Y4 = Y[..., None] if Y.ndim == 3 else Y
        if Y4.shape[:3] != (self.nrows, self.ncols, self.ndepth):
            raise ValueError("data shape %r does not match the model" % (Y.shape,))
        from scipy.special import gammaln
        rows, cols = self._plan.slabs(Y4)
        self._ctx.call("btf_set_data_gaussian", _native.dptr(rows), _native.dptr(cols), int(Y4.shape[3]))
        obs = ~np.isnan(Y4)
        y = np.where(obs, Y4, 0.0)
        # the state-independent part of the log-likelihood (the device kernels leave it out)
        if self._link <= 1:
            self._ll_const = -float(gammaln(y + 1.0)[obs].sum())                      # - sum lgamma(y+1)
        elif self._link == 2:
            self._ll_const = 0.0
        elif self._link == 3:
            s2 = self.likelihood_param
            self._ll_const = -float(0.5 * (y[obs] ** 2).sum() / s2 + 0.5 * obs.sum() * np.log(2 * np.pi * s2))
        else:
            r = self.likelihood_param
            self._ll_const = float((gammaln(y + r) - gammaln(r) - gammaln(y + 1.0))[obs].sum())

    def log_likelihood(self, data):
        """Log-likelihood of the current state (what the reference's callback returns), normalising terms included."""
        import ctypes
        if self._callback:
            return float(self.loglikelihood(self.W, self.V, data))
        self._bind_data(data)
        self._push_state()
        if not getattr(self, "_ess_ready", False):
            self._ctx.call("btf_ess_begin", 0, None, 0, 1e-6, 0)
            self._ess_ready = True
        ll = ctypes.c_double()
        self._ctx.call("btf_ess_eval", 0, 0.0, 1, self._link, ctypes.byref(ll))
        return ll.value + self._ll_const

    def _ess_step(self, what, data):
        import ctypes
        self._bind_data(data)
        self._push_state()
        o = self.linalg_opts
        eps, att = float(o["force_psd_eps"]), int(o["force_psd_attempts"]) if o["force_psd"] else 0
        if self._callback:
            # elliptical_slice.py:59-124 around the caller's function: prior draw and proposals x0 cos(phi) + nu sin(phi) on
            # the device (rng="host": from the reference's normals, fast_mvn.py:41; rng="device": Philox), the uniforms from
            # the global legacy generator in the reference's order, every proposal read back for the function
            z = self._w_normals() if what == 0 else self._v_normals()
            self._keep_z = z
            self._ctx.call("btf_ess_begin", what, _native.dptr(_native.as_f64(z)) if z is not None else None, self._next_seed(), eps, att)
            ll = ctypes.c_double()

            def value():
                if what == 0:
                    self._W_dev_new = True
                else:
                    self._V_dev_new = True
                return float(self.loglikelihood(self.W, self.V, data))
            cur = float(self.loglikelihood(self.W, self.V, data))
            hh = np.log(np.random.rand()) + cur
            phi = np.random.rand() * 2 * np.pi
            phi_min, phi_max = phi - 2 * np.pi, phi
            nev = 0
            while True:
                self._ctx.call("btf_ess_eval", what, float(phi), 0, self._link, ctypes.byref(ll))
                nev += 1
                if value() >= hh:
                    break
                if phi > 0:
                    phi_max = phi
                elif phi < 0:
                    phi_min = phi
                else:
                    break
                phi = np.random.rand() * (phi_max - phi_min) + phi_min
            self.ess_evaluations = nev
        elif self.rng == "device":
            self._ctx.call("btf_ess_run", what, self._link, 0 if self.ess == "joint" else 1, None, self._next_seed(),
                           self.ess_max_rounds, eps, att)
        else:
            # elliptical_slice.py:59-124 with the reference's draws: normals of the prior sample
            # (fast_mvn.py:41), log-uniform slice height, uniform angle, one uniform per shrink
            z = self._w_normals() if what == 0 else self._v_normals()
            self._keep_z = z
            self._ctx.call("btf_ess_begin", what, _native.dptr(_native.as_f64(z)), self._next_seed(), eps, att)
            ll = ctypes.c_double()
            self._ctx.call("btf_ess_eval", what, 0.0, 1, self._link, ctypes.byref(ll))
            hh = np.log(np.random.rand()) + ll.value
            phi = np.random.rand() * 2 * np.pi
            phi_min, phi_max = phi - 2 * np.pi, phi
            nev = 0
            while True:
                self._ctx.call("btf_ess_eval", what, float(phi), 0, self._link, ctypes.byref(ll))
                nev += 1
                if ll.value >= hh:
                    break
                if phi > 0:
                    phi_max = phi
                elif phi < 0:
                    phi_min = phi
                else:
                    break
                phi = np.random.rand() * (phi_max - phi_min) + phi_min
            self.ess_evaluations = nev
        self._ess_ready = True
        if what == 0:
            self._W_dev_new = True
        else:
            self._V_dev_new = True
            self._lsum_valid = False
            self._lsum_on_device = False

    def _resample_W(self, data):
        self._ess_step(0, data)

    def _resample_V(self, data):
        self._ess_step(1, data)

    def ess_unfinished(self):
        """Chains of the last device-driven slice that used up ess_max_rounds (they keep their current state)."""
        import ctypes
        u = ctypes.c_int32()
        self._ctx.call("btf_ess_info", ctypes.byref(u), None)
        return u.value


class ConstrainedNonconjugateBayesianTensorFiltering(NonconjugateBayesianTensorFiltering):
    """Non-conjugate likelihood under linear constraints on every curve tau_ij = (w_i . v_jt)_t, by generalized analytic
    slice sampling (factor.py:893-1010, worker functions :665-855, gass.py:13-130).  All rows of W (then all columns of
    V) are updated together on the GPU (include/btf.h, btf_gass_*): the reference maps them over a worker pool with the
    model in shared memory; `nthreads`, `multiprocessing`, `sharedprefix`, `worker_init` are accepted and ignored.

    Constraints: (J, T+1) array, row q = (c_q, bound_q): c_q . tau_ij >= bound_q for every curve (e.g. positivity
    `[I_T | 0]`, monotonicity rows `e_t - e_{t+1} >= -1e-2`, examples/poisson_tensor_filtering.py:42-48).
    Row_constraints: optional (n, K+1) fixed constraints on every row of W.
    loglikelihood: a device likelihood name as for NonconjugateBayesianTensorFiltering (the constrained Poisson model of
    the examples is "poisson_identity").  ep_approx (the optional Gaussian centering of the proposals) is not supported.

    rng="host": per row / column the slice height, the proposal normals, the grid subsample and the selection are drawn
    from `chain_rngs(what)[c]` (default: RandomState objects seeded from the global legacy generator) in the reference's
    order - with the same streams the update reproduces the reference's worker functions (tests/golden/g10_gass.npz).
    rng="device": everything on the GPU (btf_gass_run), nothing read back."""

    def __init__(self, nrows, ncols, ndepth, loglikelihood, Constraints, ep_approx=None, nthreads=3, gass_ngrid=100,
                 Row_constraints=None, multiprocessing=True, sharedprefix=None, worker_init=None, **kwargs):
        if ep_approx is not None:
            raise NotImplementedError("ep_approx (EP-centred proposals, factor.py:677-688) is not supported")
        kwargs.setdefault("ess", "joint")
        if callable(loglikelihood):
            raise NotImplementedError("ConstrainedNonconjugateBayesianTensorFiltering evaluates up to 100 candidate angles per curve "
                                      "and sweep on the device: `loglikelihood` must name a device likelihood (a Python callback is "
                                      "taken by NonconjugateBayesianTensorFiltering)")
        super().__init__(nrows, ncols, ndepth, loglikelihood, **kwargs)
        Constraints = _native.as_f64(np.atleast_2d(Constraints))
        if Constraints.shape[1] != ndepth + 1:
            raise ValueError("Constraints must be (J, ndepth + 1): the bound in the last column")
        if not 1 <= int(gass_ngrid) <= 128:
            raise ValueError("gass_ngrid must be in 1..128")
        self.Constraints_A, self.Constraints_C = Constraints[:, :-1], Constraints[:, -1:]
        self.nconstraints = Constraints.shape[0]
        self.gass_ngrid = int(gass_ngrid)
        self.Row_constraints = None if Row_constraints is None else _native.as_f64(np.atleast_2d(Row_constraints))
        if self.Row_constraints is not None and self.Row_constraints.shape[1] != self.nembeds + 1:
            raise ValueError("Row_constraints must be (n, nembeds + 1)")
        self._cons = Constraints
        self._cons_set = False
        self.chain_rngs = None            # optional: callable what -> list of RandomState, one per row / column
        self.gass_info = {}

    def shutdown(self):
        """(the reference releases its worker pool and shared arrays here)"""

    GRID = 10000

    def _gass_step(self, what, data):
        import ctypes
        self._bind_data(data)
        self._push_state()
        if not self._cons_set:
            rc = self.Row_constraints
            self._ctx.call("btf_gass_set_constraints", _native.dptr(self._cons), int(self._cons.shape[0]),
                           _native.dptr(rc), 0 if rc is None else int(rc.shape[0]))
            self._cons_set = True
        o = self.linalg_opts
        eps, att = float(o["force_psd_eps"]), int(o["force_psd_attempts"]) if o["force_psd"] else 0
        if self.rng == "device":
            self._ctx.call("btf_gass_run", what, self._link, self._next_seed(), self.gass_ngrid, eps, att)
        else:
            N, M, T, K = self.nrows, self.ncols, self.ndepth, self.nembeds
            nch = N if what == 0 else M
            rngs = self.chain_rngs(what) if self.chain_rngs is not None else \
                [np.random.RandomState(np.random.randint(0, 2 ** 31 - 1)) for _ in range(nch)]
            # gass.py:21-24 per chain: the slice uniform, then the proposal normals
            u = np.empty(nch)
            if what == 0:
                z = np.zeros(K * (K + 1) // 2 + (N - K) * K if N >= K else N * (N + 1) // 2)
                off = 0
                for i in range(N):
                    d = min(K, i + 1)
                    u[i] = rngs[i].random_sample()
                    z[off:off + d] = rngs[i].normal(size=d)
                    off += d
            else:
                z = np.empty((M, K * T))
                for j in range(M):
                    u[j] = rngs[j].random_sample()
                    z[j] = rngs[j].normal(size=K * T)
            self._keep_z = (z, u)
            self._ctx.call("btf_gass_begin", what, self._link, _native.dptr(z), _native.dptr(u), self._next_seed(), eps, att, 0)
            info = np.zeros((nch, 2), dtype=np.int32)
            mask = np.zeros((nch, self.GRID), dtype=np.uint8)
            hh = np.empty(nch)
            self._ctx.call("btf_gass_grid", what, info.ctypes.data_as(_native._c_ip), mask.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                           _native.dptr(hh), None)
            full = np.linspace(-np.pi, np.pi, self.GRID)
            thetas = np.zeros((nch, 128))
            nth = np.zeros(nch, dtype=np.int32)
            for c in range(nch):
                grid = np.linspace(-np.pi, np.pi, self.gass_ngrid) if info[c, 1] else full[mask[c] != 0]
                if len(grid) > self.gass_ngrid:
                    grid = rngs[c].choice(grid, size=self.gass_ngrid, replace=False)      # gass.py:110-111
                nth[c] = len(grid)
                thetas[c, :len(grid)] = grid
            ll = np.empty((nch, 128))
            self._ctx.call("btf_gass_eval", what, _native.dptr(thetas), nth.ctypes.data_as(_native._c_ip), _native.dptr(ll))
            theta = np.zeros(nch)
            keep = np.ones(nch, dtype=np.int32)
            acc = np.zeros(nch, dtype=np.int64)
            for c in range(nch):
                ok = np.nonzero(ll[c, :nth[c]] >= hh[c])[0]
                acc[c] = len(ok)
                if len(ok) > 0:
                    theta[c] = thetas[c, ok[rngs[c].choice(len(ok))]]                      # gass.py:121-124
                    keep[c] = 0
            self._ctx.call("btf_gass_commit", what, _native.dptr(theta), keep.ctypes.data_as(_native._c_ip))
            self.gass_info = {"valid": info[:, 0].copy(), "unrestricted": info[:, 1].copy(), "candidates": nth, "accepted": acc}
        self._ess_ready = True
        if what == 0:
            self._W_dev_new = True
        else:
            self._V_dev_new = True
            self._lsum_valid = False
            self._lsum_on_device = False

    def _resample_W(self, data):
        self._gass_step(0, data)

    def _resample_V(self, data):
        self._gass_step(1, data)
