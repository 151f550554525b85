"""Host-side mirrors (rf_inv_amd.params / model / likelihood.init_r_inv) against the oracle
and the reference's sample files.  CPU only."""
import os
import sys

import numpy as np
import pytest

from rf_inv_amd import format_model, get_params, read_obs, read_ref_model, vp_to_rho
from rf_inv_amd.likelihood import init_r_inv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sample(golden_dir):
    p = get_params(os.path.join(golden_dir, "sample_syn", "params.in"))
    read_obs(p)
    ref = read_ref_model(os.path.join(p.base_dir, p.vel_file))
    return p, ref


def test_params_in_positional_parse(sample):
    p, _ = sample
    assert (p.nburn, p.niter, p.ncorr, p.nchains, p.ncool) == (3000, 8000, 10, 5, 1)
    assert p.t_high == 15.0 and p.iseed == 12345678 and p.ntrc == 2
    assert list(p.rayps) == [0.06, 0.08] and list(p.a_gus) == [4.0, 4.0] and list(p.ipha) == [1, 1]
    assert p.nfft == 256 and p.obs_files == ["data/sample_1.trc", "data/sample_2.trc"]
    assert (p.t_start, p.t_end, p.deconv_mode, p.sdep) == (0.0, 5.0, 0, 2.0)
    assert p.vel_file == "model/sample.velmod" and p.vp_mode == 0
    assert (p.k_min, p.k_max, p.z_min, p.z_max, p.h_min) == (1, 10, 0.0, 20.0, 0.05)
    assert (p.prior_mode, p.dvs_prior, p.dvp_prior) == (2, 2.0, 0.2)
    assert list(p.sig_mode) == [0, 0] and list(p.sig_min) == [0.01, 0.01]
    assert (p.dev_z, p.dev_dvs, p.dev_dvp, p.dev_sig) == (0.02, 0.02, 0.02, 0.002)
    assert (p.nbin_z, p.nbin_vs, p.nbin_vp, p.nbin_vpvs, p.nbin_sig, p.nbin_amp) == (100, 50, 50, 100, 50, 100)
    assert (p.amp_min, p.amp_max, p.vp_min, p.vp_max) == (-0.8, 0.8, 0.1, 8.6)
    assert (p.vs_min, p.vs_max, p.vpvs_min, p.vpvs_max) == (0.001, 5.0, 0.0, 5.0)


def test_read_obs_sac(sample, oracle, golden_dir):
    p, _ = sample
    assert p.nsmp == 101 and p.delta == float(np.float32(0.05))
    assert p.obs.shape == (2, 2000)  # leading dimension npts_max like obs(npts_max, ntrc)
    for i, f in enumerate(("sample_1.trc", "sample_2.trc")):
        o, d, n = oracle.read_sac(os.path.join(golden_dir, "sample_syn", "data", f), 0.0, 5.0)
        assert np.array_equal(p.obs[i, :n], o) and np.all(p.obs[i, n:] == 0)


def test_read_ref_model(sample):
    _, ref = sample
    assert ref.vp_ref.size == 61 and ref.dz_ref == 0.5 and ref.z_ref_min == 0.0 and ref.z_ref_max == 30.0
    assert np.all(ref.vp_ref == 5.0) and np.all(ref.vs_ref == 2.89)


def test_vp_to_rho_matches_oracle_bitwise(oracle):
    for a in np.linspace(1.5, 8.5, 57):
        assert vp_to_rho(float(a)) == oracle.vp_to_rho(float(a))
    assert vp_to_rho(5.0) == 2.5347508187769563


@pytest.mark.parametrize("sdep,vp_mode", [(2.0, 0), (0.0, 0), (0.0, 1)])
def test_format_model_matches_oracle_bitwise(sample, oracle, sdep, vp_mode):
    """Layer bookkeeping (sort permutation, iz lookups, ocean prepend, validity rules) is bit-exact."""
    p, ref = sample
    import copy

    p = copy.copy(p)
    p.sdep, p.vp_mode = sdep, vp_mode
    # a non-uniform reference so that iz lookups matter
    rm = copy.copy(ref)
    rm.vp_ref = 5.0 + 0.03 * np.arange(ref.vp_ref.size)
    rm.vs_ref = 2.8 + 0.02 * np.arange(ref.vs_ref.size)
    mcfg = dict(k_max=p.k_max, vp_mode=vp_mode, sdep=sdep, z_max=p.z_max, h_min=p.h_min, z_ref_min=rm.z_ref_min,
                dz_ref=rm.dz_ref, vp_min=p.vp_min, vp_max=p.vp_max, vs_min=p.vs_min, vs_max=p.vs_max,
                vpvs_min=p.vpvs_min, vpvs_max=p.vpvs_max, vp_ref=rm.vp_ref, vs_ref=rm.vs_ref)
    rng = np.random.default_rng(42)
    n_valid = 0
    for _ in range(300):
        k = int(rng.integers(p.k_min, p.k_max))
        z = np.zeros(p.k_max - 1); dvp = np.zeros(p.k_max); dvs = np.zeros(p.k_max)
        z[:k] = rng.uniform(p.z_min + sdep, p.z_max, k)
        dvs[:k] = rng.normal(0, 0.6, k); dvs[-1] = rng.normal(0, 0.6)
        dvp[:k] = rng.normal(0, 0.3, k); dvp[-1] = rng.normal(0, 0.3)
        a = format_model(p, rm, k, z, dvp, dvs)
        b = oracle.format_model(mcfg, k, z, dvp, dvs)
        assert a[0] == b[0] and a[5] == b[5]
        for x, y in zip(a[1:5], b[1:5]):
            assert np.array_equal(x, y)
        n_valid += a[5]
    assert 20 < n_valid < 300  # both outcomes exercised


def test_init_r_inv_host_equals_oracle(oracle):
    d = float(np.float32(0.05))
    a = init_r_inv(101, [4.0, 2.5], d)
    b = oracle.build_r_inv(101, [4.0, 2.5], d)
    assert np.array_equal(a, b)


def test_sac_writer_roundtrip_and_layout(golden_dir, tmp_path):
    """write_sac emits the records make_syn.f90:121-137 writes; read_sac (params.f90 read_obs)
    reads them back; re-writing the shipped sample trace reproduces its payload bytes."""
    from rf_inv_amd import read_sac, write_sac

    src = os.path.join(golden_dir, "sample_syn", "data", "sample_1.trc")
    data, delta, nsmp = read_sac(src, 0.0, 5.0)
    out = tmp_path / "copy.trc"
    write_sac(str(out), data, delta, 0.0, 5.0)
    a = np.fromfile(src, dtype="<f4")
    b = np.fromfile(out, dtype="<f4")
    assert a.size == b.size == 158 + 101
    assert np.array_equal(a[158:], b[158:]) and a[0] == b[0] and a[5] == b[5]
    assert np.array_equal(a.view("<i4")[[79]], b.view("<i4")[[79]])          # npts @ record 80
    d2, delta2, n2 = read_sac(str(out), 0.0, 5.0)
    assert n2 == nsmp and delta2 == delta and np.array_equal(d2, data)
    # a sub-window is addressed like the reference does (it1/it2: double arithmetic on float32 header fields)
    d3, _, n3 = read_sac(str(out), 1.0, 2.0)
    assert n3 == 21 and np.array_equal(d3, data[20:41])


def test_sac_window_edge_half_a_sample_off_grid(golden_dir, oracle):
    """read_obs evaluates nint((t_start - t_beg4) / delta4) in DOUBLE (t_start is real(8), src/params.f90:66,
    :449-450; the REAL header fields are promoted): with delta4 = 0.05f = 0.05000000074505806 the edge
    t_start = 0.025 gives 0.4999999925 -> 0, where float32 arithmetic would give exactly 0.5 -> 1 and shift
    the window (and nsmp) by one sample against the reference."""
    from rf_inv_amd import read_sac

    src = os.path.join(golden_dir, "sample_syn", "data", "sample_1.trc")
    full, delta, _ = read_sac(src, 0.0, 5.0)
    assert (0.025 - 0.0) / delta < 0.5 and np.float32(0.025) / np.float32(delta) == np.float32(0.5)
    for reader in (read_sac, oracle.read_sac):
        d, _, n = reader(src, 0.025, 4.975)
        # it1 = nint(0.49999999) + 1 = 1, it2 = nint(99.4999985) + 1 = 100
        assert n == 100 and np.array_equal(d, full[0:100]), reader


def _noise_params(ntrc, rays_common, nfft=128):
    from rf_inv_amd.params import Params

    p = Params()
    p.nfft, p.ntrc, p.nsmp = nfft, ntrc, 41
    p.sig_min = np.array([0.01, 0.02, 0.005][:ntrc])
    p.sig_max = np.array([0.05, 0.02, 0.02][:ntrc])
    p.delta = float(np.float32(0.05))
    p.a_gus = np.array([4.0, 2.5, 1.5][:ntrc])
    return p


# FFTW's r2c / c2r definitions through numpy, for the CPU tests of reference_noise's draw order (the product executes
# the two plans on the GPU: rf_fft_r2c / rf_fft_c2r, tests/test_gpu_parity.py::test_fftw_plans_on_the_gpu)
HOST_PLANS = (np.fft.rfft, lambda spec, n: np.fft.irfft(spec, n) * n)


def test_make_syn_noise_consumes_the_stream_like_the_reference(oracle):
    """src/make_syn.f90:100-115 (rays not common): per trace ONE grnd() -> sigma = grnd() * (sig_max - sig_min) +
    sig_min, then nfft gauss() values (two grnd() each) times sigma, in trace order; r2c -> flt -> c2r unnormalised.
    Checked against a hand replay of the MT19937 stream; the stream position afterwards is
    ntrc * (1 + 2 nfft) draws on."""
    from rf_inv_amd.make_syn import reference_noise
    from rf_inv_amd.mcmc import gauss
    from rf_inv_amd.mt19937 import MT19937

    p = _noise_params(3, False)
    flt = oracle.init_filter(p.nfft, p.delta, p.a_gus).T                     # flt(nh, ntrc)
    noise, sigma, white = reference_noise(MT19937(4321), p, flt, False, plans=HOST_PLANS)
    g = MT19937(4321)
    for t in range(3):
        s = g.grnd() * (p.sig_max[t] - p.sig_min[t]) + p.sig_min[t]
        assert s == sigma[t] and p.sig_min[t] <= s <= p.sig_max[t]
        w = np.array([gauss(g) * s for _ in range(p.nfft)])
        assert np.array_equal(w, white[:, t])
        # the filtered series by the DEFINITIONS of FFTW's r2c / c2r (unnormalised), summed directly
        j = np.arange(p.nfft)
        spec = np.array([np.sum(w * np.exp(-2j * np.pi * j * k / p.nfft)) for k in range(p.nfft // 2 + 1)]) * flt[:, t]
        full = np.concatenate([spec, np.conj(spec[-2:0:-1])])
        full[0], full[p.nfft // 2] = full[0].real, full[p.nfft // 2].real
        direct = np.array([np.sum(full * np.exp(2j * np.pi * j * m / p.nfft)) for m in range(p.nfft)]).real
        assert np.abs(noise[:, t] - direct).max() <= 1e-12 * np.abs(direct).max()
    g2 = MT19937(4321)
    for _ in range(3 * (1 + 2 * p.nfft)):
        g2.grnd()
    assert g.grnd() == g2.grnd()
    assert sigma[1] == 0.02                                                  # sig_min == sig_max: the draw is still consumed


def test_make_syn_noise_common_rays_share_one_white_series(oracle):
    """src/make_syn.f90:84-98: with a common ray geometry ONE sigma and ONE white series are drawn (1 + 2 nfft draws
    in all).  The loop :90-96 takes its input from noise(:, 1) and stores trace itrc's result in noise(:, itrc): its
    first pass overwrites the white series with trace 1's filtered, unnormalised output, which is what the later
    passes then filter -- trace t >= 2 = flt_t(flt_1(white)) with FFTW's factor nfft twice.  Pinned here literally
    (a restatement of those seven lines on plain arrays), not as one would have meant it."""
    from rf_inv_amd.make_syn import reference_noise
    from rf_inv_amd.mt19937 import MT19937

    p = _noise_params(3, True)
    flt = oracle.init_filter(p.nfft, p.delta, p.a_gus).T
    rng = MT19937(99)
    noise, sigma, white = reference_noise(rng, p, flt, True, plans=HOST_PLANS)
    assert white.shape == (p.nfft, 1) and np.all(sigma == sigma[0])
    g2 = MT19937(99)
    s = g2.grnd() * (p.sig_max[0] - p.sig_min[0]) + p.sig_min[0]
    assert s == sigma[0]
    for _ in range(2 * p.nfft):
        g2.grnd()
    assert rng.grnd() == g2.grnd()                                           # nothing else was drawn
    # :90-96 on plain arrays: rx <- noise(:, 1); r2c; cx *= flt(:, itrc); c2r; noise(:, itrc) <- rx
    col = [white[:, 0].copy(), None, None]
    for t in range(3):
        rx = col[0].copy()
        cx = np.fft.rfft(rx) * flt[:, t]
        col[t] = np.fft.irfft(cx, p.nfft) * p.nfft
    for t in range(3):
        assert np.allclose(noise[:, t], col[t], rtol=1e-12, atol=1e-14 * np.abs(col[t]).max())
    # trace 1: the white series through its own filter; trace 2: trace 1's output through flt(:, 2), nfft twice
    w_spec = np.fft.rfft(white[:, 0])
    k = slice(1, 12)   # where all three filters are far from underflow
    assert np.allclose(np.fft.rfft(noise[:, 0])[k] / p.nfft, (w_spec * flt[:, 0])[k], rtol=1e-9)
    assert np.allclose(np.fft.rfft(noise[:, 1])[k] / p.nfft ** 2, (w_spec * flt[:, 0] * flt[:, 1])[k], rtol=1e-9)


def test_make_syn_file_names_are_the_ones_the_reference_creates(tmp_path):
    """'(A10,I2.2,A2)' applied to the 11-character literal "test_trace." keeps ten characters
    (src/make_syn.f90:120,140): the reference's files are test_traceNNwn / test_traceNN."""
    from rf_inv_amd.make_syn import _names

    assert _names(3, False) == ("test_trace03", "test_trace03wn")
    assert _names(3, True) == ("test_trace.03", "test_trace.03wn")


@pytest.mark.parametrize("shape", ["sample_syn", "c4", "c4vp"])
def test_fast_validity_verdict_equals_the_reference_format_model(tmp_path, shape):
    """rf_inv_amd/fortran/model_check.f90 -- what pt_control_batched asks instead of calling the reference's
    format_model for its verdict alone (the random stream depends on it) -- against that very routine
    (src/model.f90:175-290, compiled unmodified).  proposal_is_valid: 600 000 proposals of eight kinds (the sampler's
    own, exact ties of two depths, thicknesses at and one ulp around h_min and the 0.125 alpha rule, velocities at
    their limits, depths at the ends of the range, large perturbations); velocity_move_is_valid (one layer examined):
    valid models + a change of one dVs / dVp (small, large, exactly at and one ulp beyond the Vs and Vp/Vs limits, the
    half-space slot, the top layer); interface_move_is_valid / interface_removal_is_valid (two layers / one): valid
    models + a depth move (small, passing other interfaces, at h_min +- one ulp from a neighbour, at the 0.125 alpha
    rule, exact ties), a birth (anywhere, at h_min +- one ulp from an interface, on top of one) or a death.  Ocean / land, k_max 10 / 30, Vp fixed / solved for (c4vp): not one verdict differs."""
    import shutil
    import subprocess

    exe = os.path.join(ROOT, "oracle", "_ref", "check_model_verdict")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/check_model_verdict not built (no Fortran compiler / reference tree at build time)")
    work = tmp_path / shape
    if shape == "sample_syn":
        shutil.copytree(os.path.join(ROOT, "tests", "golden", "sample_syn"), work)
        os.makedirs(work / "rslt", exist_ok=True)
    else:
        subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "shape_run.py"), "c4", "16", str(work)], check=True,
                       capture_output=True, timeout=300)
        if shape == "c4vp":                    # vp_mode 1: the line after the velocity file's name
            lines = open(work / "params.in").read().splitlines()
            at = [i for i, x in enumerate(lines) if "sample.velmod" in x][0]
            assert lines[at + 1].strip() == "0"
            lines[at + 1] = "1"
            open(work / "params.in", "w").write("\n".join(lines) + "\n")
    r = subprocess.run([exe, "params.in", "600000"], cwd=work, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    full, moves, ifaces = [x.split() for x in r.stdout.splitlines() if "check_model_verdict:" in x][-3:]
    n, valid, bad = int(full[1]), int(full[3]), int(full[5])
    assert n == 600000 and bad == 0, r.stdout
    assert 0.1 * n < valid < 0.9 * n        # both verdicts well represented
    assert moves[2] == "velocity" and int(moves[1]) > 50000 and int(moves[6]) == 0, r.stdout
    assert 0.1 * int(moves[1]) < int(moves[4]) < 0.95 * int(moves[1])
    assert ifaces[2] == "interface" and int(ifaces[1]) > 50000 and int(ifaces[6]) == 0, r.stdout
    assert 0.1 * int(ifaces[1]) < int(ifaces[4]) < 0.95 * int(ifaces[1])


def test_write_params_is_the_inverse_of_get_params(golden_dir, tmp_path):
    """rf_inv_amd.params.write_params: the shipped params.in read, written and read again -- every field of the file
    equal -- and a modified copy (other traces, nfft, window, ocean depth, deconvolution) survives the round trip."""
    from rf_inv_amd import get_params, write_params

    def same(a, b):
        for k, v in a.__dict__.items():
            if k in ("obs", "base_dir", "nsmp", "delta"):
                continue
            w = b.__dict__[k]
            assert (np.array_equal(v, w) if isinstance(v, np.ndarray) else v == w), k

    p = get_params(os.path.join(golden_dir, "sample_syn", "params.in"))
    q = get_params(write_params(str(tmp_path / "a.in"), p, header="round trip\nof the shipped file"))
    same(p, q)
    p.ntrc, p.nfft, p.deconv_mode, p.sdep, p.t_start, p.t_end = 3, 4096, 1, 0.0, -3.0, 20.0
    p.rayps, p.a_gus, p.ipha = np.array([0.06, 0.08, 0.1]), np.array([4.0, 2.5, 1.5]), np.array([1, 1, -1], dtype=np.int32)
    p.obs_files = ["data/a.trc", "data/b.trc", "data/c.trc"]
    p.sig_min, p.sig_max, p.sig_mode = np.array([0.01, 0.005, 0.01]), np.array([0.01, 0.08, 0.01]), np.array([0, 1, 0], dtype=np.int32)
    p.k_max = 30
    same(p, get_params(write_params(str(tmp_path / "b.in"), p)))
