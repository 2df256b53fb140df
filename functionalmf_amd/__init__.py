"""functionalmf_amd - MI355X-native Gibbs core for Bayesian Tensor Filtering.

Drop-in for the hot path of tansey/functionalmf (``functionalmf.factor``'s
``GaussianBayesianTensorFiltering`` / ``BinomialBayesianTensorFiltering``:
``run_gibbs`` -> ``resample`` -> ``_resample_W`` / ``_resample_V`` /
``_resample_nu2``), with the conditional-posterior draws executed by
hand-written HIP kernels for gfx950 behind the C ABI in ``include/btf.h``.

Module names mirror the reference package (``factor``, ``genlasso``,
``fast_mvn``, ``utils``) so user scripts only change the import root.
"""
__all__ = ["factor", "genlasso", "fast_mvn", "utils"]
__version__ = "0.1.0"
