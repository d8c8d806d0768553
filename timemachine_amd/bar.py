"""Free-energy estimators over reduced-potential matrices: the consumers of the HREX / lambda-window energy matrices.

Mirrors the function surface of the reference's `timemachine/fe/bar.py` (EXP :19-41, BARzero :44-80, dG_dw :83-102,
ukln_to_ukn :105-128, df_and_err_from_u_kln :131-149, df_from_u_kln :152-166, bootstrap_bar :169-218,
bar_with_pessimistic_uncertainty :221-259, bar :262-285, works_from_ukln :288-294, df_from_ukln_by_lambda :297-319,
pair_overlap_from_ukln :322-353, compute_fwd_and_reverse_df_over_time :356-401).

The reference delegates the numerics to pymbar == 4.0.3 (third party, absent from the reference tree and from this
image).  What is implemented here is the published algorithm, not pymbar's code:

* MBAR (Shirts & Chodera, J. Chem. Phys. 129, 124105 (2008)): the self-consistent equations (eq. 11), solved by an
  adaptive mix of self-consistent and Newton-Raphson steps on the negative log-likelihood (eqs. C1-C9: whichever step
  leaves the smaller gradient is taken); asymptotic covariance from the SVD form of Theta (eq. D8); overlap matrix
  O = W^T W N (Klimovich, Shirts & Mobley 2015).
* BAR (Bennett, J. Comput. Phys. 22, 245 (1976)): root of the implicit equation (eq. 12a with the T_F/T_R shift),
  bracketed by the two exponential-averaging estimates and solved by bisection + Newton polishing; variance eq. 10a.

PARITY UNPINNED against pymbar itself (it cannot be run here); pinned instead by the checks the reference's own
`tests/test_bar.py` uses: analytic Gaussian / uniform examples with known log Z ratios, two-state MBAR == BAR (they are
the same estimator), and self-consistency of the estimating equations.  Host-side numpy, off the device path.
"""

import logging
from typing import Optional

import numpy as np
from scipy.special import logsumexp
from scipy.stats import normaltest

DG_KEY = "Delta_f"
DG_ERR_KEY = "dDelta_f"

DEFAULT_RELATIVE_TOLERANCE = 1e-6
DEFAULT_MAXIMUM_ITERATIONS = 1_000
DEFAULT_SOLVER_PROTOCOL = "robust"

logger = logging.getLogger(__name__)


class ParameterError(ValueError):
    """covariance could not be computed (the role pymbar.utils.ParameterError plays at fe/bar.py:145)"""


# ---------------------------------------------------------------------------------------------------------------------
# exponential averaging and BAR
# ---------------------------------------------------------------------------------------------------------------------
def EXP(w_raw):
    """-log < exp(-w) >, skipping None entries (fe/bar.py:19-41)"""
    w = np.array([ww for ww in w_raw if ww is not None], dtype=np.float64)
    return float(-(logsumexp(-w) - np.log(w.size)))


def _log_fermi(x):
    """log(1 / (1 + exp(x))), overflow-free"""
    x = np.asarray(x, dtype=np.float64)
    return -np.logaddexp(0.0, x)


def BARzero(w, deltaF):
    """log sum_F f(M + w_F - dF) - log sum_R f(-M + w_R + dF), f = Fermi function; zero at the BAR estimate
    (fe/bar.py:44-80).  `w` = (w_F, w_R), possibly of different lengths."""
    w_F = np.asarray(w[0], dtype=np.float64)
    w_R = np.asarray(w[1], dtype=np.float64)
    M = np.log(len(w_F) / len(w_R))
    log_numer = logsumexp(_log_fermi(M + w_F - deltaF))
    log_denom = logsumexp(_log_fermi(-(M - w_R - deltaF)))
    return float(log_numer - log_denom)


def _bar_solve(w_F, w_R, relative_tolerance=1e-12, maximum_iterations=500):
    # BARzero is strictly increasing in deltaF (each Fermi term of the numerator grows, each of the denominator shrinks):
    # bracket, bisect to a tight interval, then the interval midpoint.  The exponential-averaging estimates bracket the
    # root when they are finite; otherwise expand.
    lo, hi = sorted([EXP(w_F), -EXP(w_R)])
    if not np.isfinite(lo) or not np.isfinite(hi):
        finite = [v for v in (lo, hi) if np.isfinite(v)]
        lo = hi = finite[0] if finite else 0.0
    pad = 1e-3
    while BARzero((w_F, w_R), lo) > 0:
        lo -= pad
        pad *= 2
        if pad > 1e12:
            raise ParameterError("BAR: cannot bracket the root (no overlap)")
    pad = 1e-3
    while BARzero((w_F, w_R), hi) < 0:
        hi += pad
        pad *= 2
        if pad > 1e12:
            raise ParameterError("BAR: cannot bracket the root (no overlap)")
    for _ in range(maximum_iterations):
        mid = 0.5 * (lo + hi)
        if BARzero((w_F, w_R), mid) < 0:
            lo = mid
        else:
            hi = mid
        if hi - lo <= relative_tolerance * max(1.0, abs(mid)):
            break
    return 0.5 * (lo + hi)


def _bar_variance(w_F, w_R, df):
    """Bennett 1976 eq. 10a: var = [<f^2>_F / <f>_F^2 - 1] / T_F + [<f^2>_R / <f>_R^2 - 1] / T_R"""
    T_F, T_R = len(w_F), len(w_R)
    M = np.log(T_F / T_R)
    lf_F = _log_fermi(M + w_F - df)
    lf_R = _log_fermi(-(M - w_R - df))

    def ratio(lf, T):
        # <f^2> / <f>^2 = T * sum f^2 / (sum f)^2
        return np.exp(np.log(T) + logsumexp(2 * lf) - 2 * logsumexp(lf))

    return (ratio(lf_F, T_F) - 1.0) / T_F + (ratio(lf_R, T_R) - 1.0) / T_R


def bar(w_F, w_R, compute_uncertainty=True, **kwargs):
    """(df, ddf | None) from forward / reverse works (fe/bar.py:262-285)"""
    w_F = np.asarray(w_F, dtype=np.float64)
    w_R = np.asarray(w_R, dtype=np.float64)
    df = _bar_solve(w_F, w_R, **{k: v for k, v in kwargs.items() if k in ("relative_tolerance", "maximum_iterations")})
    if not compute_uncertainty:
        return df, None
    return df, float(np.sqrt(max(_bar_variance(w_F, w_R, df), 0.0)))


def dG_dw(w):
    """d(BAR estimate) / d(works), by the implicit-function theorem on BARzero (fe/bar.py:83-102 does the same with
    jax.grad): dF/dw = -(dZ/dw) / (dZ/dF)."""
    w_F = np.asarray(w[0], dtype=np.float64)
    w_R = np.asarray(w[1], dtype=np.float64)
    df, _ = bar(w_F, w_R, compute_uncertainty=False)
    M = np.log(len(w_F) / len(w_R))
    a_F = M + w_F - df
    a_R = -(M - w_R - df)
    # d/da log f(a) = -sigmoid(a); softmax weights of the two log-sum-exps
    p_F = np.exp(_log_fermi(a_F) - logsumexp(_log_fermi(a_F)))
    p_R = np.exp(_log_fermi(a_R) - logsumexp(_log_fermi(a_R)))
    s_F = np.exp(-np.logaddexp(0.0, -a_F))
    s_R = np.exp(-np.logaddexp(0.0, -a_R))
    dZ_dwF = -p_F * s_F
    dZ_dwR = p_R * s_R
    dZ_dF = np.sum(p_F * s_F) + np.sum(p_R * s_R)
    return np.array([-dZ_dwF / dZ_dF, -dZ_dwR / dZ_dF], dtype=object if len(w_F) != len(w_R) else np.float64)


# ---------------------------------------------------------------------------------------------------------------------
# MBAR
# ---------------------------------------------------------------------------------------------------------------------
def kln_to_kn(u_kln):
    """[k, l, n] -> [l, k*n]: energies of all samples (grouped by sampled state) in every state"""
    k, l, n = u_kln.shape
    return np.transpose(u_kln, (1, 0, 2)).reshape(l, k * n)


def ukln_to_ukn(u_kln):
    """2-state u_kln -> (u_kn, N_k), the MBAR inputs (fe/bar.py:105-128)"""
    u_kn = kln_to_kn(u_kln)
    k, l, n = u_kln.shape
    assert k == l == 2
    assert u_kn.shape == (k, l * n)
    N_k = n * np.ones(l)
    return u_kn, N_k


class MBAR:
    """Multistate Bennett acceptance ratio over u_kn[K, N_total] (reduced potentials of every sample in every state)
    with N_k samples drawn from state k (samples ordered by state)."""

    def __init__(
        self,
        u_kn,
        N_k,
        initial_f_k=None,
        maximum_iterations=DEFAULT_MAXIMUM_ITERATIONS,
        relative_tolerance=DEFAULT_RELATIVE_TOLERANCE,
        solver_protocol=DEFAULT_SOLVER_PROTOCOL,
    ):
        self.u_kn = np.asarray(u_kn, dtype=np.float64)
        self.N_k = np.asarray(N_k, dtype=np.float64)
        K, N = self.u_kn.shape
        if self.N_k.shape != (K,) or int(self.N_k.sum()) != N:
            raise ParameterError("N_k must have one entry per state and sum to the number of samples")
        if np.isnan(self.u_kn).any():
            raise ParameterError("u_kn contains NaN")
        self.sampled = self.N_k > 0
        self.log_N_k = np.where(self.sampled, np.log(np.maximum(self.N_k, 1)), -np.inf)
        f0 = np.zeros(K) if initial_f_k is None else np.asarray(initial_f_k, dtype=np.float64).copy()
        self.f_k, self.iterations, self.converged = self._solve(f0, maximum_iterations, relative_tolerance)

    # log of the mixture denominator per sample: log sum_k N_k exp(f_k - u_k(x_n))
    def _log_denom(self, f_k):
        with np.errstate(invalid="ignore"):
            a = (self.log_N_k + f_k)[:, None] - self.u_kn
        return logsumexp(a, axis=0)

    def _self_consistent(self, f_k):
        ld = self._log_denom(f_k)
        new = -logsumexp(-self.u_kn - ld[None, :], axis=1)
        return new - new[0]

    def _log_W(self, f_k):
        return f_k[:, None] - self.u_kn - self._log_denom(f_k)[None, :]

    def _gradient_hessian(self, f_k):
        # negative log-likelihood (up to constants): sum_n log_denom_n - sum_k N_k f_k; eqs. C6, C9
        W = np.exp(self._log_W(f_k))  # [K, N]
        NW = self.N_k[:, None] * W
        g = NW.sum(axis=1) - self.N_k
        H = np.diag(NW.sum(axis=1)) - NW @ NW.T
        return g, H

    def _newton(self, f_k):
        g, H = self._gradient_hessian(f_k)
        s = self.sampled.copy()
        s[np.argmax(s)] = False  # gauge: first sampled state fixed
        step = np.zeros_like(f_k)
        if s.any():
            Hs = H[np.ix_(s, s)]
            step[s] = np.linalg.lstsq(Hs, g[s], rcond=None)[0]
        new = f_k - step
        # unsampled states follow from the sampled ones
        new_sc = self._self_consistent(new)
        new = np.where(self.sampled, new - new[0], new_sc)
        return new

    def _gnorm(self, f_k):
        W = np.exp(self._log_W(f_k))
        g = self.N_k * W.sum(axis=1) - self.N_k
        return float(np.sqrt(np.sum(g[self.sampled] ** 2)))

    def _solve(self, f_k, maximum_iterations, relative_tolerance):
        f_k = f_k - f_k[0]
        converged = False
        it = 0
        for it in range(1, maximum_iterations + 1):
            cand_sc = self._self_consistent(f_k)
            try:
                cand_nr = self._newton(f_k)
                use_nr = np.all(np.isfinite(cand_nr)) and self._gnorm(cand_nr) < self._gnorm(cand_sc)
            except np.linalg.LinAlgError:
                use_nr = False
            new = cand_nr if use_nr else cand_sc
            finite = np.isfinite(new) & np.isfinite(f_k)
            denom = np.max(np.abs(new[finite])) if finite.any() else 0.0
            delta = np.max(np.abs(new[finite] - f_k[finite])) if finite.any() else 0.0
            f_k = new
            if denom == 0.0 or delta / denom < relative_tolerance:
                converged = True
                break
        return f_k, it, converged

    def W_nk(self):
        return np.exp(self._log_W(self.f_k)).T  # [N, K]

    def _theta(self):
        # eq. D8: Theta = V S (I - S V^T N V S)^+ S V^T with W = U S V^T
        W = self.W_nk()
        K = W.shape[1]
        _, S, Vt = np.linalg.svd(W, full_matrices=False)
        Sig = np.diag(S)
        V = Vt.T
        Ndiag = np.diag(self.N_k)
        inner = np.eye(K) - Sig @ V.T @ Ndiag @ V @ Sig
        return V @ Sig @ np.linalg.pinv(inner, rcond=1e-10) @ Sig @ V.T

    def compute_free_energy_differences(self, compute_uncertainty=True):
        f = self.f_k
        with np.errstate(invalid="ignore"):
            out = {DG_KEY: f[None, :] - f[:, None]}
        if compute_uncertainty:
            theta = self._theta()
            d = np.diag(theta)
            var = d[:, None] + d[None, :] - 2 * theta
            if not np.all(np.isfinite(var)) or np.any(var < -1e-8 * max(1.0, np.max(np.abs(var)))):
                raise ParameterError("MBAR covariance is not positive: incomplete convergence or no overlap")
            out[DG_ERR_KEY] = np.sqrt(np.maximum(var, 0.0))
        return out

    def compute_overlap(self):
        W = self.W_nk()
        O = (W.T @ W) * self.N_k[None, :]
        eig = np.sort(np.linalg.eigvals(O).real)[::-1]
        return {"scalar": 1.0 - eig[1] if len(eig) > 1 else 0.0, "eigenvalues": eig, "matrix": O}


def df_and_err_from_u_kln(u_kln, maximum_iterations=DEFAULT_MAXIMUM_ITERATIONS):
    """2-state df and its asymptotic uncertainty; NaN uncertainty when the covariance cannot be formed
    (fe/bar.py:131-149)"""
    u_kn, N_k = ukln_to_ukn(np.asarray(u_kln))
    mbar = MBAR(u_kn, N_k, maximum_iterations=maximum_iterations, relative_tolerance=DEFAULT_RELATIVE_TOLERANCE)
    try:
        res = mbar.compute_free_energy_differences()
        return float(res[DG_KEY][0, 1]), float(res[DG_ERR_KEY][0, 1])
    except ParameterError:
        df = mbar.compute_free_energy_differences(compute_uncertainty=False)[DG_KEY]
        return float(df[0, 1]), float("nan")


def df_from_u_kln(u_kln, initial_f_k: Optional[np.ndarray] = None, maximum_iterations=DEFAULT_MAXIMUM_ITERATIONS):
    """2-state df (fe/bar.py:152-166)"""
    u_kn, N_k = ukln_to_ukn(np.asarray(u_kln))
    mbar = MBAR(
        u_kn, N_k, initial_f_k=initial_f_k, maximum_iterations=maximum_iterations, relative_tolerance=DEFAULT_RELATIVE_TOLERANCE
    )
    return float(mbar.compute_free_energy_differences(compute_uncertainty=False)[DG_KEY][0, 1])


def bootstrap_bar(u_kln, n_bootstrap=100, maximum_iterations=DEFAULT_MAXIMUM_ITERATIONS):
    """(df, ddf, bootstrap df samples): frames resampled with replacement, seed 2022, warm-started from the full
    estimate (fe/bar.py:169-218)"""
    u_kln = np.asarray(u_kln)
    full_df, full_err = df_and_err_from_u_kln(u_kln, maximum_iterations=maximum_iterations)
    n = u_kln.shape[2]
    rng = np.random.default_rng(2022)
    samples = []
    for _ in range(n_bootstrap):
        sample = rng.choice(u_kln, size=(n,), replace=True, axis=2)
        samples.append(df_from_u_kln(sample, initial_f_k=np.array([0.0, full_df]), maximum_iterations=maximum_iterations))
    return full_df, full_err, np.array(samples)


def bar_with_pessimistic_uncertainty(u_kln, n_bootstrap=100, maximum_iterations=DEFAULT_MAXIMUM_ITERATIONS):
    """df and max(asymptotic, bootstrap) uncertainty (fe/bar.py:221-259)"""
    df, ddf, boot = bootstrap_bar(u_kln, n_bootstrap=n_bootstrap, maximum_iterations=maximum_iterations)
    if len(boot) >= 8 and np.ptp(boot) > 0:
        res = normaltest(boot)
        if res.pvalue < 1e-3:
            logger.warning(f"bootstrapped errors non-normal: {res}")
    if not np.isfinite(ddf):
        logger.warning(f"BAR error estimate is not finite, setting to zero: {ddf}")
        ddf = 0.0
    return df, float(np.maximum(ddf, np.std(boot)))


def works_from_ukln(u_kln):
    """forward / reverse works of a 2-state u_kln (fe/bar.py:288-294)"""
    k, l, _ = u_kln.shape
    assert k == l == 2
    return u_kln[0, 1, :] - u_kln[0, 0, :], u_kln[1, 0, :] - u_kln[1, 1, :]


def df_from_ukln_by_lambda(ukln_by_lambda):
    """sum of pair dfs over adjacent windows, errors in quadrature (fe/bar.py:297-319)"""
    dfs, errs = [], []
    for window in ukln_by_lambda:
        df, err = df_and_err_from_u_kln(window)
        dfs.append(df)
        errs.append(err)
    return float(np.sum(dfs)), float(np.linalg.norm(errs))


def pair_overlap_from_ukln(u_kln, maximum_iterations=DEFAULT_MAXIMUM_ITERATIONS, relative_tolerance=DEFAULT_RELATIVE_TOLERANCE):
    """2 x off-diagonal of the 2x2 MBAR overlap matrix, clipped to [0, 1] (fe/bar.py:322-353)"""
    u_kn, N_k = ukln_to_ukn(np.asarray(u_kln))
    O = MBAR(u_kn, N_k, maximum_iterations=maximum_iterations, relative_tolerance=relative_tolerance).compute_overlap()["matrix"]
    return float(np.clip(2 * O[0, 1], 0.0, 1.0))


def compute_fwd_and_reverse_df_over_time(ukln_by_lambda, frames_per_step=100):
    """df (and error) from growing prefixes of the frames, forward and time-reversed (fe/bar.py:356-401)"""
    ukln_by_lambda = np.asarray(ukln_by_lambda)
    assert ukln_by_lambda.ndim == 4
    assert ukln_by_lambda.shape[1] == 2
    total = ukln_by_lambda.shape[-1]
    assert total >= frames_per_step, "fewer samples than frames_per_step"
    rev = np.flip(ukln_by_lambda, 3)
    fwd_out, rev_out = [], []
    for n in range(frames_per_step, total + 1, frames_per_step):
        fwd_out.append(df_from_ukln_by_lambda(ukln_by_lambda[..., :n]))
        rev_out.append(df_from_ukln_by_lambda(rev[..., :n]))
    fwd_out, rev_out = np.array(fwd_out), np.array(rev_out)
    return fwd_out[:, 0], fwd_out[:, 1], rev_out[:, 0], rev_out[:, 1]


def mbar_from_u_kln(u_kln, **kwargs):
    """K-state MBAR straight from the [K, K, n] matrix an HREX run accumulates (equal n per state)"""
    u_kln = np.asarray(u_kln)
    k, l, n = u_kln.shape
    assert k == l
    return MBAR(kln_to_kn(u_kln), n * np.ones(k), **kwargs)
