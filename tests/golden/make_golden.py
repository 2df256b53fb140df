#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the REAL
reference code (/root/reference/functionalmf) in the build container.

Run:  python tests/golden/make_golden.py          (build container only; the
reference does not exist on the GPU box, the .npz files travel instead).

Only *data* is written: inputs, injected states, the normals consumed, and the
arrays the reference produced.  No reference source is copied.

Three third-party modules the reference imports at module top are absent from
this image (ordinary ModuleNotFoundError, no permission denial):

  * ``SharedArray``            - only used by the out-of-scope constrained model;
                                 an empty module object is registered.
  * ``sksparse.cholmod``       - CHOLMOD wrapper used by fast_mvn.py:38-47.  A
                                 shim with the call surface the reference uses
                                 (cholesky(Q) -> .P() .solve_Lt() .solve_A()) is
                                 registered.  It factorises P Q P' with dense
                                 LAPACK for a *declared* permutation P (set per
                                 fixture: depth-major or identity).  Everything
                                 the reference computes itself - the precision
                                 matrix Q, mu_part, control flow, RNG order,
                                 un-permutation, jitter retries - is exercised
                                 for real; Q and mu_part are captured on entry.
                                 What is NOT pinned: CHOLMOD's own ordering
                                 ("parity unpinned" items).
  * ``pypolyagamma``           - shim whose pgdrawv fills the output from a
                                 caller-supplied omega array, so the Binomial
                                 W/V steps are pinned *given* omega.
"""
import os
import sys
import types
import numpy as np
import scipy.linalg as sla

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

# ----------------------------------------------------------------- shims ----
_cfg = {"perm": None, "log": None, "omega": None}


class _NotPD(Exception):
    pass


class _Factor:
    def __init__(self, Q):
        Qd = Q.toarray() if hasattr(Q, "toarray") else np.asarray(Q, float)
        n = Qd.shape[0]
        p = np.arange(n) if _cfg["perm"] is None else _cfg["perm"](n)
        self.p = np.asarray(p)
        try:
            self.Lo = np.linalg.cholesky(Qd[np.ix_(self.p, self.p)])
        except np.linalg.LinAlgError:
            if _cfg["log"] is not None:
                _cfg["log"].append(("fail", Qd.copy()))
            raise _NotPD()
        if _cfg["log"] is not None:
            _cfg["log"].append(("ok", Qd.copy()))

    def P(self):
        return self.p

    def solve_Lt(self, b, use_LDLt_decomposition=True):
        return sla.solve_triangular(self.Lo.T, b, lower=False)

    def solve_A(self, b):
        if _cfg["log"] is not None:
            _cfg["log"].append(("mu", np.array(b, float)))
        out = np.empty_like(np.asarray(b, float))
        out[self.p] = sla.cho_solve((self.Lo, True), np.asarray(b, float)[self.p])
        return out


def _install_shims():
    sa = types.ModuleType("SharedArray")
    sk = types.ModuleType("sksparse")
    ch = types.ModuleType("sksparse.cholmod")
    ch.cholesky = lambda Q, **kw: _Factor(Q)
    ch.CholmodNotPositiveDefiniteError = _NotPD
    sk.cholmod = ch
    pg = types.ModuleType("pypolyagamma")

    class PyPolyaGamma:
        def __init__(self, seed=0):
            self.seed = seed

        def pgdrawv(self, n, z, out):
            out[:] = _cfg["omega"].reshape(-1)

    pg.PyPolyaGamma = PyPolyaGamma
    sys.modules.update({"SharedArray": sa, "sksparse": sk, "sksparse.cholmod": ch,
                        "pypolyagamma": pg})


def _twist_perm_factory(K, T, tf=2):
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.btf_oracle import twisted_perm

    def f(n):
        assert n == K * T
        return twisted_perm(K, T, tf)
    return f


def _depth_perm_factory(K, T):
    def f(n):
        assert n == K * T
        t, k = np.meshgrid(np.arange(T), np.arange(K), indexing="ij")
        return (k * T + t).reshape(-1)
    return f


# ---------------------------------------------------------------- helpers ---
def synth(N, M, T, R, K, seed, noise=0.5):
    """SURVEY 8(d) synthetic generator."""
    np.random.seed(seed)
    Wt = np.random.normal(0, 1, size=(N, K))
    Wt[np.triu_indices(K, k=1)] = 0
    Vt = 0.1 * np.cumsum(np.random.normal(0, 1, size=(M, T, K)), axis=1)
    Mu = np.einsum("nk,mtk->nmt", Wt, Vt)
    Y = Mu[..., None] + np.random.normal(0, noise, size=(N, M, T, R))
    return Y, Wt, Vt


def snapshot(model):
    keys = ["W", "V", "Tau2", "Tau2_a", "Tau2_b", "Tau2_c", "lam2", "lam2_a", "sigma2", "nu2"]
    return {k: np.array(getattr(model, k), dtype=float).copy() for k in keys}


def pack(prefix, d):
    return {prefix + k: v for k, v in d.items()}


def collect_systems(log, M):
    """Split the shim log of one _resample_V call into per-column final Q, mu and
    number of failed attempts."""
    Qs, mus, tries = [], [], []
    fails = 0
    for kind, arr in log:
        if kind == "fail":
            fails += 1
        elif kind == "ok":
            Qs.append(arr)
            tries.append(fails)
            fails = 0
        elif kind == "mu":
            mus.append(arr)
    assert len(Qs) == M and len(mus) == M
    return np.stack(Qs), np.stack(mus), np.array(tries)


def half_sweeps(factor, model, data, K, T, seed, want_systems=True):
    """From the model's current (injected) state: W step, then V step under both
    declared permutations, each under a fresh legacy seed.  Returns dict."""
    out = {}
    st0 = snapshot(model)
    N = model.nrows
    M = model.ncols
    nz_w = sum(min(i + 1, K) for i in range(N))
    # ---- W step
    np.random.seed(seed)
    out["z_W"] = np.random.normal(size=nz_w)
    np.random.seed(seed)
    model._resample_W(data)
    out["W_after"] = model.W.copy()
    # ---- V step (state = after W step), depth-major then identity from same state
    st1 = snapshot(model)
    for name, pf in (("depth", _depth_perm_factory(K, T)), ("ident", None), ("twist", _twist_perm_factory(K, T))):
        model.V[:] = st1["V"]
        _cfg["perm"] = pf
        _cfg["log"] = []
        np.random.seed(seed + 1)
        model._resample_V(data)
        Qs, mus, tries = collect_systems(_cfg["log"], M)
        _cfg["log"] = None
        out["V_after_" + name] = model.V.copy()
        out["V_tries_" + name] = tries
        if want_systems and name == "depth":
            out["V_Q"] = Qs
            out["V_mu"] = mus
    np.random.seed(seed + 1)
    out["z_V"] = np.random.normal(size=(M, K * T))   # valid when no retry consumed... (retries draw nothing)
    out.update(pack("s0_", st0))
    return out


def main():
    _install_shims()
    sys.path.insert(0, REF)
    from functionalmf import utils as rutils
    from functionalmf import factor as rfactor
    from functionalmf import genlasso as rgen

    # ------------------------------------------------------------ G0: Delta
    g0 = {}
    for T in (6, 12, 16, 64):
        for k in (0, 1, 2):
            g0["delta_T%d_k%d" % (T, k)] = rutils.bayes_grid_penalty(T, k).toarray()
    np.savez_compressed(os.path.join(HERE, "g0_delta.npz"), **g0)

    def build(N, M, T, K, seed, cls=None, **kw):
        _cfg["perm"] = _depth_perm_factory(K, T)
        np.random.seed(seed)
        cls = cls or rfactor.GaussianBayesianTensorFiltering
        return cls(N, M, T, nembeds=K, tf_order=2, sigma2_init=0.5, lam2_init=0.1,
                   nthreads=1, **kw)

    # ---------------------- G1: C1 dims, [:3,:3] held out, all steps pinned
    N, M, T, R, K = 10, 11, 12, 3, 3
    Y, Wt, Vt = synth(N, M, T, R, K, seed=1)
    Y[:3, :3] = np.nan
    model = build(N, M, T, K, seed=11, nu2_init=1.0)
    g1 = {"Y": Y, "dims": np.array([N, M, T, R, K, 2])}
    g1.update(pack("init_", snapshot(model)))
    # a couple of real sweeps to leave the prior draw, then pin every step
    np.random.seed(12)
    for _ in range(2):
        model.resample(Y)
    g1.update(half_sweeps(rfactor, model, Y, K, T, seed=100))
    # hyper steps from the state after the half sweeps (V = depth variant restored)
    model.V[:] = g1["V_after_depth"]
    g1.update(pack("h0_", snapshot(model)))
    np.random.seed(200)
    model._resample_nu2(Y)
    g1["h_nu2"] = float(model.nu2)
    Mu = np.einsum("nk,mtk->nmt", model.W, model.V)
    g1["h_sse"] = float(np.nansum((Mu[..., None] - Y) ** 2))
    g1["h_nobs"] = int(np.sum(~np.isnan(Y)))
    np.random.seed(201)
    model._resample_sigma2()
    g1["h_sigma2"] = float(model.sigma2)
    np.random.seed(202)
    model._resample_Tau2()
    g1.update(pack("h_tau_", {k: getattr(model, k).copy() for k in ("Tau2", "Tau2_a", "Tau2_b", "Tau2_c")}))
    np.random.seed(203)
    model._resample_lam2()
    g1["h_lam2"] = float(model.lam2)
    g1["h_lam2_a"] = float(model.lam2_a)
    np.savez_compressed(os.path.join(HERE, "g1_c1_heldout.npz"), **g1)

    # ------------------- G6: run_gibbs chain at C1 dims (3 burn + 2*4 kept)
    model = build(N, M, T, K, seed=21, nu2_init=1.0)
    g6 = {"Y": Y}
    g6.update(pack("init_", snapshot(model)))
    np.random.seed(22)
    res = model.run_gibbs(Y, nburn=3, nthin=2, nsamples=4, verbose=False)
    g6.update(pack("res_", {k: np.asarray(v) for k, v in res.items()}))
    model = build(N, M, T, K, seed=21, nu2_init=1.0)            # same construction (depth-major prior draw)
    _cfg["perm"] = _twist_perm_factory(K, T)
    np.random.seed(22)
    res = model.run_gibbs(Y, nburn=3, nthin=2, nsamples=4, verbose=False)
    g6.update(pack("rest_", {k: np.asarray(v) for k, v in res.items()}))
    np.savez_compressed(os.path.join(HERE, "g6_c1_chain.npz"), **g6)

    # --------------------- G2: (64,32,16,2) K=3 complete data (Q1 cache path)
    N, M, T, R, K = 64, 32, 16, 2, 3
    Y, Wt, Vt = synth(N, M, T, R, K, seed=2)
    model = build(N, M, T, K, seed=31, nu2_init=1.0)
    np.random.seed(32)
    for _ in range(2):
        model.resample(Y)
    g2 = {"Y": Y, "dims": np.array([N, M, T, R, K, 2])}
    g2.update(half_sweeps(rfactor, model, Y, K, T, seed=300, want_systems=False))
    np.savez_compressed(os.path.join(HERE, "g2_c2_complete.npz"), **g2)

    # ------------- G3: partially missing replicates + some fully missing (Q2)
    N, M, T, R, K = 12, 9, 10, 3, 3
    Y, Wt, Vt = synth(N, M, T, R, K, seed=3)
    rs = np.random.RandomState(33)
    Y[rs.rand(N, M, T, R) < 0.15] = np.nan          # single replicates
    Y[2:5, 4:6] = np.nan                            # whole curves: pattern change at cols 4,6
    model = build(N, M, T, K, seed=34, nu2_init=0.7)
    np.random.seed(35)
    model.resample(Y)
    g3 = {"Y": Y, "dims": np.array([N, M, T, R, K, 2])}
    g3.update(half_sweeps(rfactor, model, Y, K, T, seed=400))
    np.savez_compressed(os.path.join(HERE, "g3_partial_reps.npz"), **g3)

    # ---- G4: Binomial W/V given omega (tensor nu2), with and without NaNs
    for tag, hold in (("nan", True), ("full", False)):
        N, M, T, K = 10, 8, 9, 3
        rs = np.random.RandomState(41)
        _, Wt, Vt = synth(N, M, T, 1, K, seed=4)
        Mu = np.einsum("nk,mtk->nmt", Wt, Vt)
        Ntr = np.full((N, M, T), 6.0)
        Ys = rs.binomial(6, 1 / (1 + np.exp(-Mu))).astype(float)
        if hold:
            Ys[:3, :3] = np.nan
            Ys[5, 6, 2:5] = np.nan
            Ntr[np.isnan(Ys)] = np.nan
        model = build(N, M, T, K, seed=42, cls=rfactor.BinomialBayesianTensorFiltering)
        omega = rs.gamma(2.0, 0.6, size=(N, M, T))
        _cfg["omega"] = omega
        model._resample_nu2((Ys, Ntr))            # nu2 = 1/omega through the reference code
        g4 = {"Ysucc": Ys, "Ntrials": Ntr, "omega": omega, "dims": np.array([N, M, T, 1, K, 2])}
        g4.update(half_sweeps(rfactor, model, (Ys, Ntr), K, T, seed=500))
        np.savez_compressed(os.path.join(HERE, "g4_binomial_%s.npz" % tag), **g4)

    # -- G5: ill-conditioned state (lam2 floor, wide Tau2) + forced jitter retry
    N, M, T, R, K = 10, 6, 12, 2, 3
    Y, Wt, Vt = synth(N, M, T, R, K, seed=5)
    model = build(N, M, T, K, seed=51, nu2_init=1.0)
    rs = np.random.RandomState(52)
    model.lam2 = 1e-5
    model.Tau2[:] = 10.0 ** rs.uniform(-7, 6, size=model.Tau2.shape)
    g5 = {"Y": Y, "dims": np.array([N, M, T, R, K, 2])}
    import signal
    signal.alarm(120)
    g5.update(half_sweeps(rfactor, model, Y, K, T, seed=600))
    signal.alarm(0)
    # forced retry: column 2's system is made slightly indefinite (min eigenvalue
    # about -5e-6) through one negative Tau2 entry found by bisection, so the
    # reference's jitter loop (fast_mvn.py:62-68) needs exactly two shifts
    # (1e-6, then +1e-5).  A grossly indefinite system would spin forever in the
    # reference (fast_mvn.py:69-72), hence the alarm.
    model.V[:] = g5["s0_V"]
    model.W[:] = g5["W_after"]
    model.lam2 = 0.1
    model.nu2 = 1e4
    model.Tau2[:] = 1.0
    Dl = model.Delta.toarray()
    cnt = (~np.isnan(Y)).sum(-1)

    def min_eig(tau):
        lamT = 1.0 / (model.lam2 * np.where(np.arange(Dl.shape[0]) == 5, tau, 1.0))
        P1 = Dl.T @ (Dl * lamT[:, None])
        G = np.einsum("it,ik,il->tkl", cnt[:, 2, :] / model.nu2, model.W, model.W)
        Q = np.kron(np.eye(K), P1)
        for k in range(K):
            for l in range(K):
                Q[k * T + np.arange(T), l * T + np.arange(T)] += G[:, k, l]
        return np.linalg.eigvalsh(Q)[0]
    lo_t, hi_t = -1e3, -1e-3          # tau -> -inf : tiny negative term ; tau -> -0 : huge
    assert min_eig(lo_t) > -5e-6 > min_eig(hi_t)
    for _ in range(200):
        mid = -np.sqrt(lo_t * hi_t)
        if min_eig(mid) > -5e-6:
            lo_t = mid
        else:
            hi_t = mid
    model.Tau2[2, 5] = lo_t
    print("G5 retry: Tau2[2,5] = %r  min eig = %.3e" % (lo_t, min_eig(lo_t)))
    st = snapshot(model)
    _cfg["perm"] = _depth_perm_factory(K, T)
    _cfg["log"] = []
    np.random.seed(601)
    import warnings
    import signal
    signal.alarm(60)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model._resample_V(Y)
    signal.alarm(0)
    Qs, mus, tries = collect_systems(_cfg["log"], M)
    _cfg["log"] = None
    assert tries[2] == 2 and tries.sum() == 2, tries
    g5["retry_V_after"] = model.V.copy()
    g5["retry_tries"] = tries
    # the same forced retry under the twisted ordering
    for k_, v_ in st.items():
        if k_ in ("W", "V", "Tau2"):
            getattr(model, k_)[:] = v_
    _cfg["perm"] = _twist_perm_factory(K, T)
    _cfg["log"] = []
    np.random.seed(601)
    signal.alarm(60)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model._resample_V(Y)
    signal.alarm(0)
    _, _, tries_t = collect_systems(_cfg["log"], M)
    _cfg["log"] = None
    g5["retry_V_after_twist"] = model.V.copy()
    g5["retry_tries_twist"] = tries_t
    g5["retry_Q_final"] = Qs
    np.random.seed(601)
    g5["retry_z_V"] = np.random.normal(size=(M, K * T))
    g5.update(pack("retry_s0_", st))
    np.savez_compressed(os.path.join(HERE, "g5_illcond.npz"), **g5)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-28s %8.1f KB" % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))


if __name__ == "__main__":
    main()
