"""The conditional sampler P(theta_h | theta_i) (SURVEY.md §8f item 2 in its survey form; definition: oracle/merl_oracle.h)
— the oracle's restatement pinned on the CPU: every row is a distribution, the mixture pdf integrates to the accepted
fraction, sample() reports pdf(wi, wo') exactly and weight == eval / pdf, a chi-square test of the sampled half vectors
against the pdf, and the reason it exists: a lower variance of the weight than the row marginal's.  Parity unpinned (the
reference's sample() strategy is unknown: SURVEY.md Appendix B 5)."""
import numpy as np
import pytest

from mitsuba_customization_amd import synth
from oracle import binding as ob


@pytest.fixture(scope="module")
def table():
    tab = synth.make_table("ggx_tab", 0)
    T = ob.OracleTable(tab)
    return T, T.sampling2d(32)


def test_rows_are_distributions(table):
    T, sp = table
    A = T.sampling2d_arrays(sp)
    n = T.c.n_th
    s = T.sampling_arrays()[0]
    assert A.shape == (32, 2 * n + 1)
    cdf, c = A[:, :n + 1], A[:, n + 1:]
    assert (cdf[:, 0] == 0).all() and (cdf[:, -1] == 1).all() and (np.diff(cdf, axis=1) > 0).all()
    # c_j pi ds_j is the bin's probability
    assert np.allclose((c * np.pi * np.diff(s)).sum(axis=1), 1.0, rtol=1e-12)
    assert np.allclose(c * np.pi * np.diff(s), np.diff(cdf, axis=1), rtol=1e-9, atol=1e-15)
    # the rows differ with the incident angle (that is the point), and the floor keeps every bin alive
    assert np.abs(cdf[2] - cdf[30]).max() > 1e-3 and (c > 0).all()


def test_pdf_integrates_to_the_accepted_fraction(table):
    """One incident direction per bin family: quadrature of pdf(wi, .) over the upper hemisphere equals the fraction of
    samples that land above the horizon (the half-vector lobe sends some below it)."""
    T, sp = table
    rng = np.random.default_rng(5)
    for mu in (0.98, 0.7, 0.3):
        wi1 = np.array([np.sqrt(1 - mu * mu), 0.0, mu], np.float32)
        # a stratified quadrature in (cos theta, phi)
        nz, nphi = 800, 720
        z = (np.arange(nz) + 0.5) / nz
        ph = (np.arange(nphi) + 0.5) / nphi * 2 * np.pi
        Z, P = np.meshgrid(z, ph, indexing="ij")
        r = np.sqrt(1 - Z * Z)
        wo = np.stack([r * np.cos(P), r * np.sin(P), Z], -1).reshape(-1, 3).astype(np.float32)
        wi = np.tile(wi1, (wo.shape[0], 1))
        integral = T.pdf_table2d(sp, wi, wo).astype(np.float64).mean() * 2 * np.pi
        u = rng.random((400000, 2)).astype(np.float32)
        _, pdf, _ = T.sample_table2d(sp, np.tile(wi1, (u.shape[0], 1)), u)
        accepted = (pdf > 0).mean()
        assert abs(integral - accepted) < 4e-3, (mu, integral, accepted)


def test_sample_reports_its_own_pdf_and_weight_is_eval_over_pdf(table):
    T, sp = table
    wi, _, u = ob.generate_pairs(77, 0, 100000)
    wo, pdf, w = T.sample_table2d(sp, wi, u)
    live = pdf > 0
    assert 0.7 < live.mean() <= 1.0
    assert np.array_equal(T.pdf_table2d(sp, wi[live], wo[live]), pdf[live])
    f = T.eval(wi[live], wo[live])
    assert np.array_equal(w[live], (f / pdf[live, None]).astype(np.float32))
    assert not w[~live].any() and not wo[~live].any()
    # the cosine branch: u0 < 1/8 -> the cosine-hemisphere direction of (8 u0, u1), bit for bit
    lo = u[:, 0] < 0.125
    c_wo, _, _ = T.sample(wi[lo], np.stack([u[lo, 0] * 8.0, u[lo, 1]], 1).astype(np.float32))
    assert np.array_equal(wo[lo][pdf[lo] > 0], c_wo[pdf[lo] > 0])


def test_chi_square_of_sampled_directions_against_the_pdf(table):
    T, sp = table
    mu = 0.6
    wi1 = np.array([np.sqrt(1 - mu * mu), 0.0, mu], np.float32)
    n = 1_000_000
    rng = np.random.default_rng(11)
    u = rng.random((n, 2)).astype(np.float32)
    wo, pdf, _ = T.sample_table2d(sp, np.tile(wi1, (n, 1)), u)
    live = pdf > 0
    # bins in (cos theta_o, phi_o): expected counts by quadrature of the pdf (4 x 4 sub-samples per bin)
    nz, nphi, sub = 16, 32, 6
    zi = np.minimum((wo[live, 2] * nz).astype(int), nz - 1)
    pi_ = np.minimum(((np.arctan2(wo[live, 1], wo[live, 0]) + np.pi) / (2 * np.pi) * nphi).astype(int), nphi - 1)
    obs = np.bincount(zi * nphi + pi_, minlength=nz * nphi).astype(np.float64)
    zz = (np.arange(nz * sub) + 0.5) / (nz * sub)
    pp = (np.arange(nphi * sub) + 0.5) / (nphi * sub) * 2 * np.pi - np.pi
    Z, P = np.meshgrid(zz, pp, indexing="ij")
    r = np.sqrt(1 - Z * Z)
    q = np.stack([r * np.cos(P), r * np.sin(P), Z], -1).reshape(-1, 3).astype(np.float32)
    dens = T.pdf_table2d(sp, np.tile(wi1, (q.shape[0], 1)), q).astype(np.float64).reshape(nz, sub, nphi, sub).mean(axis=(1, 3))
    exp = dens.reshape(-1) * (2 * np.pi / (nz * nphi)) * n
    keep = exp > 50
    chi2 = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum()
    dof = keep.sum() - 1
    # the lobe is piecewise constant over narrow theta_h rings: the bin quadrature carries a few percent of error in the
    # bins a ring edge crosses, so the statistic is compared with a generous bound rather than a p-value
    assert chi2 < 3.0 * dof, (chi2, dof)
    assert abs(obs.sum() / n - exp.sum() / n) < 5e-3


def test_variance_is_below_the_row_marginal(table):
    T, sp = table
    wi, _, u = ob.generate_pairs(3, 0, 400000)
    lum = lambda w: 0.2126 * w[:, 0] + 0.7152 * w[:, 1] + 0.0722 * w[:, 2]
    w2 = lum(T.sample_table2d(sp, wi, u)[2].astype(np.float64))
    w1 = lum(T.sample_table(wi, u)[2].astype(np.float64))
    wc = lum(T.sample(wi, u)[2].astype(np.float64))
    assert abs(w2.mean() - w1.mean()) < 0.01 * w1.mean() and abs(w2.mean() - wc.mean()) < 0.02 * wc.mean()    # same integral
    assert w2.var() < 0.75 * w1.var() < 0.75 * wc.var(), (w2.var(), w1.var(), wc.var())


def test_standard_parameterisation_gets_flat_rows():
    tab = synth.make_table("ggx_std", 3, (12, 10, 16))
    T = ob.OracleTable(tab, (1.0, 1.0, 1.0), param=1)
    sp = T.sampling2d(8)
    A = T.sampling2d_arrays(sp)
    n = 12
    s = T.sampling_arrays()[0]
    assert np.allclose(A[:, :n + 1], np.tile((s - s[0]) / (s[-1] - s[0]), (8, 1)), atol=1e-15)   # uniform in s: p_h = cos(theta_h) / pi
    assert np.allclose(A[:, n + 1:], 1.0 / np.pi, rtol=1e-12)
