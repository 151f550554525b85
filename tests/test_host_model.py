"""Host-side mirrors (rf_inv_amd.params / model / likelihood.init_r_inv) against the oracle
and the reference's sample files.  CPU only."""
import os

import numpy as np
import pytest

from rf_inv_amd import format_model, get_params, read_obs, read_ref_model, vp_to_rho
from rf_inv_amd.likelihood import init_r_inv


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
