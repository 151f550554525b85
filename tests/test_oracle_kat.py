"""Pins the CPU oracle against the reference's own fixtures.

The fixtures under tests/golden/sample_syn/ are the reference's sample data files
(data, not source): sample_syn/true/true.velmod, sample_syn/data/sample_{1,2}.trc,
sample_syn/params.in, sample_syn/model/sample.velmod.
"""
import os

import numpy as np
import pytest


def _true_model(golden_dir):
    vm = np.loadtxt(os.path.join(golden_dir, "sample_syn", "true", "true.velmod"))
    return vm.T  # alpha, beta, rho, h


@pytest.mark.parametrize("fname,rayp,rms_max", [("sample_1.trc", 0.06, 2.5e-9), ("sample_2.trc", 0.08, 6e-9)])
def test_kat_sample_syn_traces(oracle, golden_dir, fname, rayp, rms_max):
    """calc_rf(true.velmod, land, P, deconv 0, nfft 256, a 4) == shipped SAC trace
    to float32 quantisation: every sample rounds to the stored float32."""
    alpha, beta, rho, h = _true_model(golden_dir)
    obs, delta, nsmp = oracle.read_sac(os.path.join(golden_dir, "sample_syn", "data", fname), 0.0, 5.0)
    assert nsmp == 101
    assert delta == float(np.float32(0.05))
    cfg = dict(nfft=256, deconv_mode=0, delta=delta, t_start=0.0, sdep=0.0,
               rayps=[rayp], a_gus=[4.0], ipha=[1])
    rft = oracle.calc_rf(cfg, alpha, beta, rho, h)
    d = rft[0, :nsmp] - obs
    assert np.sqrt(np.mean(d * d)) < rms_max
    assert np.array_equal(rft[0, :nsmp].astype(np.float32), obs.astype(np.float32))


def test_kat_vp_to_rho(oracle, golden_dir):
    alpha, _, rho, _ = _true_model(golden_dir)
    assert oracle.vp_to_rho(alpha[0]) == rho[0] == 2.5347508187769563
    # exact-decimal coefficients would give a different double
    a = 5.0
    exact = 1.6612 * a - 0.4721 * a**2 + 0.0671 * a**3 - 0.0043 * a**4 + 0.000106 * a**5
    assert exact != rho[0]


def test_multi_trace_noncommon_rays_matches_single(oracle, golden_dir):
    """2-trace call with different rays == two 1-trace calls (forward.f90:141)."""
    alpha, beta, rho, h = _true_model(golden_dir)
    delta = float(np.float32(0.05))
    base = dict(nfft=256, deconv_mode=0, delta=delta, t_start=0.0, sdep=0.0)
    both = oracle.calc_rf(dict(base, rayps=[0.06, 0.08], a_gus=[4.0, 4.0], ipha=[1, 1]), alpha, beta, rho, h)
    for i, p in enumerate([0.06, 0.08]):
        one = oracle.calc_rf(dict(base, rayps=[p], a_gus=[4.0], ipha=[1]), alpha, beta, rho, h)
        assert np.array_equal(both[i], one[0])


def _random_stack(rng, nlay, ocean):
    alpha = rng.uniform(4.0, 7.5, nlay)
    beta = alpha / rng.uniform(1.6, 1.9, nlay)
    rho = np.array([0.77 + 0.32 * a for a in alpha])
    h = rng.uniform(0.3, 6.0, nlay)
    h[-1] = 999.0
    if ocean:
        alpha[0], beta[0], rho[0], h[0] = 1.5, -999.0, 1.0, 2.0
    return alpha, beta, rho, h


@pytest.mark.parametrize("ocean", [False, True])
@pytest.mark.parametrize("ipha", [1, -1])
@pytest.mark.parametrize("nlay", [2, 3, 9])
def test_c_vs_numpy_restatement(oracle, ocean, ipha, nlay):
    """Two independent restatements of calc_seis (C scalar loops; numpy dense 4x4
    complex matmul) agree to rounding on land/ocean x P/S."""
    if ocean and nlay < 3:
        pytest.skip("ocean needs >= 1 solid layer above the half-space to be interesting")
    rng = np.random.default_rng(100 * nlay + 10 * ocean + (ipha > 0))
    alpha, beta, rho, h = _random_stack(rng, nlay, ocean)
    delta = float(np.float32(0.05))
    ur, uz = oracle.calc_seis(256, delta, 0.07, ipha, alpha, beta, rho, h)
    ur2, uz2 = oracle.calc_seis_numpy(256, delta, 0.07, ipha, alpha, beta, rho, h)
    scale = max(np.abs(ur).max(), np.abs(uz).max())
    assert np.abs(ur - ur2).max() <= 1e-11 * scale
    assert np.abs(uz - uz2).max() <= 1e-11 * scale


@pytest.mark.parametrize("n", [8, 256, 4096])
def test_c2r_definition(oracle, n):
    """The oracle FFT equals the published c2r definition (long-double O(n^2) sum)."""
    rng = np.random.default_rng(n)
    nh = n // 2 + 1
    cx = rng.standard_normal(nh) + 1j * rng.standard_normal(nh)
    fast = oracle.c2r(cx, n)
    nn = n if n <= 256 else None
    if nn is None:
        # O(n^2) in long double is slow at 4096: check against numpy's irfft instead
        ref = np.fft.irfft(cx, n) * n
    else:
        ref = oracle.c2r(cx, n, naive=True)
    assert np.abs(fast - ref).max() <= 1e-12 * np.abs(ref).max()
    # imaginary parts of DC and Nyquist bins are ignored
    cx2 = cx.copy(); cx2[0] = cx[0].real; cx2[-1] = cx[-1].real
    assert np.array_equal(oracle.c2r(cx2, n), fast)


def test_non_pow2_falls_back_to_definition(oracle):
    rng = np.random.default_rng(7)
    n = 12
    cx = rng.standard_normal(n // 2 + 1) + 1j * rng.standard_normal(n // 2 + 1)
    assert np.allclose(oracle.c2r(cx, n), np.fft.irfft(cx, n) * n, rtol=0, atol=1e-13)


def test_s_wave_trace_is_time_reversed_and_negated(oracle, golden_dir):
    """forward.f90:185-194: for ipha = -1 the shift map is rft(i) = -rx(mod(n+npre-i+1, n))."""
    alpha, beta, rho, h = _true_model(golden_dir)
    delta = float(np.float32(0.05))
    cfg = dict(nfft=256, deconv_mode=1, delta=delta, t_start=-1.0, sdep=0.0,
               rayps=[0.1], a_gus=[4.0], ipha=[-1])
    rft, npre, rff, fv = oracle.calc_rf(cfg, alpha, beta, rho, h, want_stages=True)
    assert npre[0] == 20  # nint((1.0 + 0) / delta), tp = 0 in deconv mode
    flt = oracle.init_filter(256, delta, [4.0])
    rx = oracle.c2r(rff[0] * flt[0], 256)
    n = 256
    j = (n + npre[0] - np.arange(1, n + 1) + 1) % n
    j[j == 0] = n
    assert np.array_equal(rft[0], -rx[j - 1])


def test_evanescent_gives_nan(oracle, golden_dir):
    """Out-of-domain physics (1/v^2 < p^2) propagates NaN, never traps."""
    alpha, beta, rho, h = _true_model(golden_dir)
    cfg = dict(nfft=256, deconv_mode=0, delta=0.05, t_start=0.0, sdep=0.0,
               rayps=[0.25], a_gus=[4.0], ipha=[1])  # p > 1/alpha = 0.2
    rft = oracle.calc_rf(cfg, alpha, beta, rho, h)
    assert np.isnan(rft).all()


def test_r_inv_rank_and_pinv_property(oracle):
    """init_r_inv (likelihood.f90:168-222): rank 40 of 101 for the sample_syn
    geometry, and R+ is a truncated pseudo-inverse."""
    delta = float(np.float32(0.05))
    r_inv, ranks = oracle.build_r_inv(101, [4.0], delta, return_rank=True)
    assert ranks == [40]
    idx = np.arange(101)
    R = np.exp(-16.0 * delta * delta) ** ((idx[:, None] - idx[None, :]) ** 2.0)
    P = r_inv[0].T  # P[i, j] = r_inv(i, j)
    assert np.abs(P @ R @ P - P).max() < 1e-6 * np.abs(P).max()
    assert np.abs(P - P.T).max() < 1e-7 * np.abs(P).max()


def test_log_likelihood_formula(oracle):
    rng = np.random.default_rng(3)
    nsmp, nfft, ntrc = 17, 32, 2
    rft = rng.standard_normal((ntrc, nfft))
    obs = rng.standard_normal((ntrc, 40))  # leading dimension > nsmp, like obs(npts_max, ntrc)
    A = rng.standard_normal((ntrc, nsmp, nsmp))
    sig = np.array([0.3, 0.7])
    ll = oracle.log_likelihood(rft, obs, A, sig, nsmp)
    ref = 0.0
    for t in range(ntrc):
        m = rft[t, :nsmp] - obs[t, :nsmp]
        ref += -0.5 * (m @ A[t].T @ m) / sig[t] ** 2 - nsmp * np.log(sig[t])
    assert abs(ll - ref) < 1e-10 * abs(ref)


def test_format_model_sample_reference(oracle, golden_dir):
    """format_model on sample.velmod (uniform 5.0 / 2.89): bit-exact bookkeeping."""
    ref = np.loadtxt(os.path.join(golden_dir, "sample_syn", "model", "sample.velmod"))
    mcfg = dict(k_max=10, vp_mode=0, sdep=2.0, z_max=20.0, h_min=0.05,
                z_ref_min=ref[0, 0], dz_ref=ref[-1, 0] - ref[-2, 0],
                vp_min=0.1, vp_max=8.6, vs_min=0.001, vs_max=5.0, vpvs_min=0.0, vpvs_max=5.0,
                vp_ref=ref[:, 1], vs_ref=ref[:, 2])
    z = np.zeros(9); dvp = np.zeros(10); dvs = np.zeros(10)
    z[:3] = [12.0, 4.0, 7.5]
    dvs[:3] = [0.3, -0.2, 0.1]
    dvs[9] = 0.5
    nlay, alpha, beta, rho, h, ok = oracle.format_model(mcfg, 3, z, dvp, dvs)
    assert nlay == 5 and ok
    assert np.array_equal(alpha, [1.5, 5.0, 5.0, 5.0, 5.0])
    assert np.array_equal(beta, [-999.0, 2.89 - 0.2, 2.89 + 0.1, 2.89 + 0.3, 2.89 + 0.5])
    assert np.array_equal(h, [2.0, 2.0, 3.5, 4.5, 999.0])
    assert rho[0] == 1.0 and rho[1] == 2.5347508187769563
    # top-layer rule h < 0.125 * alpha (model.f90:229), not h_min
    z[:3] = [12.0, 2.5, 7.5]
    assert oracle.format_model(mcfg, 3, z, dvp, dvs)[5] is False
    # thin middle layer (h < h_min)
    z[:3] = [12.0, 4.0, 4.04]
    assert oracle.format_model(mcfg, 3, z, dvp, dvs)[5] is False


@pytest.mark.parametrize("ipha", [1, -1])
def test_homogeneous_halfspace_apparent_angle(oracle, ipha):
    """Physics known answer, independent of the reference: when the layer equals the half-space the
    surface ratio u_r/u_z is the classic free-surface apparent-incidence relation, real and
    frequency independent --  P: -2 b^2 p eta / (1 - 2 b^2 p^2),  SV: (1 - 2 b^2 p^2) / (2 b^2 p xi)
    (eta, xi = vertical S, P slowness).  Pins the S-incidence boundary-condition lines
    (forward.f90:273-274) and the E^-1 entries they use, which no reference fixture covers."""
    a, b, rho, p = 6.1, 3.4, 2.7, 0.07
    alpha, beta, dens, h = [a, a], [b, b], [rho, rho], [7.3, 999.0]
    ur, uz = oracle.calc_seis(256, 0.05, p, ipha, alpha, beta, dens, h)
    eta = np.sqrt(1 / b**2 - p**2)
    xi = np.sqrt(1 / a**2 - p**2)
    bp = 1 - 2 * b**2 * p**2
    want = -2 * b**2 * p * eta / bp if ipha == 1 else bp / (2 * b**2 * p * xi)
    ratio = ur / uz
    assert np.abs(ratio.imag).max() < 1e-10
    assert np.abs(ratio.real - want).max() < 1e-10 * abs(want)
    # and with no thickness dependence at all
    ur2, uz2 = oracle.calc_seis(256, 0.05, p, ipha, alpha, beta, dens, [0.4, 999.0])
    assert np.abs(ur2 / uz2 - ratio).max() < 1e-10


@pytest.mark.parametrize("ipha", [1, -1])
def test_water_level_decon_against_numpy_restatement(oracle, golden_dir, ipha):
    """water_level_decon (forward.f90:447-470) and its call sites (:148-153): the C restatement
    against a separate numpy one built from calc_seis' raw output."""
    alpha, beta, rho, h = _true_model(golden_dir)
    delta = float(np.float32(0.05))
    p = 0.07
    cfg = dict(nfft=256, deconv_mode=1, delta=delta, t_start=-1.0, sdep=0.0, rayps=[p], a_gus=[4.0], ipha=[ipha])
    _, npre, rff, fv = oracle.calc_rf(cfg, alpha, beta, rho, h, want_stages=True)
    ur, uz = oracle.calc_seis(256, delta, p, ipha, alpha, beta, rho, h)
    freq_r, freq_v = np.conj(ur), -np.conj(uz)            # :145-146
    y, x = (freq_r, freq_v) if ipha == 1 else (freq_v, freq_r)
    amp = (x * np.conj(x)).real
    want = y * np.conj(x) / np.maximum(amp, 0.001 * amp.max())
    assert np.abs(rff[0] - want).max() <= 1e-13 * np.abs(want).max()
    assert np.array_equal(fv[0], freq_v) and npre[0] == 20  # tp = 0 in deconvolution mode


def test_speed_build_of_the_oracle_returns_identical_values(oracle):
    """bench.py times the oracle compiled -O3 -march=native (oracle.lib_fast: still no FMA contraction, no
    fast-math); the checker build is -O2.  Same source, same values, bit for bit -- incl. S traces, the ocean
    branch and deconvolution."""
    from helpers import DELTA, make_cfg, pack_layers, random_stack

    rng = np.random.default_rng(8)
    for sdep, dec in ((0.0, 0), (2.0, 1)):
        cfg = make_cfg(nfft=1024, deconv_mode=dec, sdep=sdep, t_start=-1.0, rayps=[0.06, 0.10], ipha=[1, -1])
        stacks = [random_stack(rng, int(n), sdep > 0, sdep) for n in (3, 8, 21)]
        nlay, layers = pack_layers(stacks, 23)
        obs = rng.normal(size=(2, 101)) * 0.1
        r_inv = oracle.build_r_inv(101, cfg["a_gus"], DELTA)
        sig = np.full((3, 2), 0.02)
        a, ra = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, 101, want_rft=True)
        b, rb = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, 101, want_rft=True, nthreads=2, fast=True)
        assert np.array_equal(a, b) and np.array_equal(ra, rb)


@pytest.mark.parametrize("nfft", [8, 250, 256, 1000, 4096])
def test_oracle_c2r_against_the_fftw3_interface_the_reference_build_uses(oracle, nfft):
    """rfo_c2r (this oracle's transform) and rfo_c2r_naive (the O(n^2) long-double definition) against the third-party
    transform of the reference's CPU build: MKL's FFTW3 interface, called exactly as src/fftw.f90:44 / src/forward.f90:172
    call it.  Im of the DC and Nyquist bins must be ignored by all three."""
    from helpers import mkl_fftw3_c2r

    rng = np.random.default_rng(nfft)
    nh = nfft // 2 + 1
    spec = rng.normal(0, 1, nh) + 1j * rng.normal(0, 1, nh)
    want = mkl_fftw3_c2r(spec, nfft)
    if want is None:
        pytest.skip("no /opt/conda/lib/libmkl_rt.so here")
    full = np.concatenate([spec, np.zeros(nfft - nh)])
    for naive in (False, True):
        got = oracle.c2r(full, nfft, naive=naive)
        assert np.abs(got - want).max() <= 4e-15 * np.abs(want).max() * max(1.0, np.log2(nfft) / 4), (nfft, naive)
