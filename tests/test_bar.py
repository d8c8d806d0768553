"""Host-side free-energy estimators (timemachine_amd/bar.py): analytic known answers, BAR == two-state MBAR, and the
estimating equations themselves.  The cases follow what the reference checks in tests/test_bar.py (Gaussian pairs with
known log Z ratio :28-53, partially overlapping uniforms :56-78, bootstrap :81-110, over-time :205-232); pymbar is a
third-party dependency that is absent here, so nothing is compared with pymbar output."""

import numpy as np
import pytest

from timemachine_amd import bar as B


def gaussian_ukln(params_a, params_b, seed=0, n=2000):
    (mu_a, s_a), (mu_b, s_b) = params_a, params_b
    rng = np.random.default_rng(seed)
    x_a = rng.normal(mu_a, s_a, n)
    x_b = rng.normal(mu_b, s_b, n)
    u = lambda mu, s, x: (x - mu) ** 2 / (2 * s**2)
    u_kln = np.array([[u(mu_a, s_a, x_a), u(mu_b, s_b, x_a)], [u(mu_a, s_a, x_b), u(mu_b, s_b, x_b)]])
    return u_kln, np.log(s_a) - np.log(s_b)


def uniform_ukln(dlogZ, n=100):
    rng = np.random.default_rng(2023)
    u_a = lambda x: np.where((x > 0) & (x < 1), 0.0, np.inf)
    u_b = lambda x: u_a(x - 0.5) + dlogZ
    x_a = rng.uniform(0, 1, n)
    x_b = rng.uniform(0.5, 1.5, n)
    return np.array([[u_a(x_a), u_b(x_a)], [u_a(x_b), u_b(x_b)]])


@pytest.mark.parametrize("sigma", [0.1, 1.0, 10.0])
def test_two_state_estimate_brackets_the_exact_answer(sigma):
    u_kln, dlogZ = gaussian_ukln((0.0, 1.0), (1.0, sigma))
    df, err = B.df_and_err_from_u_kln(u_kln)
    assert np.isfinite(err) and err > 0
    assert df == pytest.approx(dlogZ, abs=3.0 * err)
    assert B.df_from_u_kln(u_kln) == df


@pytest.mark.parametrize("sigma", [0.3, 1.0, 10.0])
def test_bar_and_two_state_mbar_are_the_same_estimator(sigma):
    u_kln, _ = gaussian_ukln((0.0, 1.0), (1.0, sigma))
    w_F, w_R = B.works_from_ukln(u_kln)
    df_bar, err_bar = B.bar(w_F, w_R)
    df, err = B.df_and_err_from_u_kln(u_kln)
    assert df == pytest.approx(df_bar, abs=1e-5)
    assert err == pytest.approx(err_bar, rel=0.02)
    assert abs(B.BARzero((w_F, w_R), df_bar)) < 1e-9
    df2, none = B.bar(w_F, w_R, compute_uncertainty=False)
    assert none is None and df2 == df_bar


def test_mbar_solution_satisfies_the_self_consistent_equations():
    rng = np.random.default_rng(5)
    K, n = 5, 300
    # harmonic oscillators with different centres and widths: f_k = -log(sigma_k) + const
    mus = np.linspace(0, 2, K)
    sig = np.array([1.0, 0.8, 1.2, 0.7, 1.5])
    xs = [rng.normal(mus[k], sig[k], n) for k in range(K)]
    u_kln = np.array([[(xs[k] - mus[l]) ** 2 / (2 * sig[l] ** 2) for l in range(K)] for k in range(K)])
    m = B.mbar_from_u_kln(u_kln, relative_tolerance=1e-12)
    assert m.converged
    f = m.f_k
    ld = m._log_denom(f)
    from scipy.special import logsumexp

    resid = f + logsumexp(-m.u_kn - ld[None, :], axis=1)
    assert np.allclose(resid - resid[0], 0, atol=1e-9)
    res = m.compute_free_energy_differences()
    exact = -np.log(sig) + np.log(sig[0])
    err = res[B.DG_ERR_KEY][0]
    assert np.all(np.abs(res[B.DG_KEY][0] - exact)[1:] < 4 * err[1:])
    # weights are normalised per state and the overlap matrix is row-stochastic
    W = m.W_nk()
    assert np.allclose(W.sum(axis=0), 1.0)
    O = m.compute_overlap()["matrix"]
    assert np.allclose(O.sum(axis=1), 1.0)


def test_partial_overlap_uniforms():
    dlogZ = 5.0
    u_kln = uniform_ukln(dlogZ)
    df, err = B.df_and_err_from_u_kln(u_kln)
    assert np.isfinite(df)
    assert df == pytest.approx(dlogZ, abs=3.0 * err if np.isfinite(err) else 1.0)
    ov = B.pair_overlap_from_ukln(u_kln)
    assert 0.0 < ov < 1.0


def test_pair_overlap_limits():
    same, _ = gaussian_ukln((0.0, 1.0), (0.0, 1.0))
    assert B.pair_overlap_from_ukln(same) == pytest.approx(1.0, abs=1e-9)
    far, _ = gaussian_ukln((0.0, 1.0), (40.0, 1.0), n=200)
    assert B.pair_overlap_from_ukln(far) < 1e-6
    mid, _ = gaussian_ukln((0.0, 1.0), (1.0, 1.0))
    assert 0.3 < B.pair_overlap_from_ukln(mid) < 0.95


@pytest.mark.parametrize("sigma", [0.1, 1.0])
def test_bootstrap_and_pessimistic_uncertainty(sigma):
    u_kln, dlogZ = gaussian_ukln((0.0, 1.0), (1.0, sigma), n=500)
    df_ref, err_ref = B.df_and_err_from_u_kln(u_kln)
    df0, err0, samples = B.bootstrap_bar(u_kln, n_bootstrap=30)
    assert (df0, err0) == (df_ref, err_ref)
    assert samples.shape == (30,)
    df1, pess = B.bar_with_pessimistic_uncertainty(u_kln, n_bootstrap=30)
    assert df1 == df_ref
    assert pess >= err_ref
    np.testing.assert_approx_equal(pess, err_ref, significant=1)
    assert df1 == pytest.approx(dlogZ, abs=3.0 * pess)


def test_exp_and_gradient_of_bar():
    rng = np.random.default_rng(1)
    w = rng.normal(2.0, 1.0, 5000)
    # <exp(-w)> for a Gaussian: dF = mu - sigma^2 / 2
    assert B.EXP(list(w) + [None]) == pytest.approx(1.5, abs=0.1)
    w_F = rng.normal(1.0, 0.7, 40)
    w_R = rng.normal(-0.6, 0.7, 30)
    g = B.dG_dw((w_F, w_R))
    h = 1e-5
    for side, idx in [(0, 3), (1, 7)]:
        wp = [w_F.copy(), w_R.copy()]
        wm = [w_F.copy(), w_R.copy()]
        wp[side][idx] += h
        wm[side][idx] -= h
        fd = (B.bar(*wp, compute_uncertainty=False)[0] - B.bar(*wm, compute_uncertainty=False)[0]) / (2 * h)
        assert g[side][idx] == pytest.approx(fd, rel=1e-4, abs=1e-8)


def test_df_over_lambda_and_over_time():
    windows = []
    exact = 0.0
    sig = [1.0, 0.8, 0.6, 0.5]
    for i in range(3):
        u, d = gaussian_ukln((0.0, sig[i]), (0.2, sig[i + 1]), seed=i, n=400)
        windows.append(u)
        exact += d
    ukln_by_lambda = np.array(windows)
    df, err = B.df_from_ukln_by_lambda(ukln_by_lambda)
    assert df == pytest.approx(exact, abs=4 * err)
    fwd, fwd_err, rev, rev_err = B.compute_fwd_and_reverse_df_over_time(ukln_by_lambda, frames_per_step=100)
    assert fwd.shape == fwd_err.shape == rev.shape == rev_err.shape == (4,)
    assert fwd[-1] == pytest.approx(df) and rev[-1] == pytest.approx(df, abs=1e-6)
    assert fwd_err[-1] < fwd_err[0]
    with pytest.raises(AssertionError, match="fewer samples than frames_per_step"):
        B.compute_fwd_and_reverse_df_over_time(ukln_by_lambda, frames_per_step=1000)


def test_ukln_to_ukn_layout():
    u_kln = np.arange(2 * 2 * 3, dtype=float).reshape(2, 2, 3)
    u_kn, N_k = B.ukln_to_ukn(u_kln)
    assert u_kn.shape == (2, 6) and list(N_k) == [3, 3]
    # row l = energies in state l of samples from state 0 then state 1
    assert list(u_kn[1]) == list(u_kln[0, 1]) + list(u_kln[1, 1])
