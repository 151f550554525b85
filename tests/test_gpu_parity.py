"""HIP path vs CPU oracle through the C ABI (include/rfgpu.h).  -m gpu only."""
import os
import zlib

import numpy as np
import pytest

from helpers import DELTA, assert_logl_parity, load_true_model, logl_tol, make_cfg, pack_layers, random_stack, synth_obs

pytestmark = pytest.mark.gpu


def _engine(cfg, obs, nsmp, r_inv, max_walkers=8, nlay_max=40, options=None):
    """options: launch-plan knobs (rf_set_option); a context that cannot run the fused kernel
    ignores a request for it (common rays / LDS footprint), like the library's own default."""
    from rf_inv_amd import RFEngine

    eng = RFEngine(nfft=cfg["nfft"], delta=cfg["delta"], t_start=cfg["t_start"], deconv_mode=cfg["deconv_mode"],
                   sdep=cfg["sdep"], rayps=cfg["rayps"], a_gus=cfg["a_gus"], ipha=cfg["ipha"], obs=obs,
                   nsmp=nsmp, r_inv=r_inv, max_walkers=max_walkers, nlay_max=nlay_max)
    for k, v in (options or {}).items():
        if k == "fused" and int(v) == 1 and not eng.launch_plan["fused"]:
            continue
        eng.set_option(k, v)
    return eng


def test_library_is_the_hip_one():
    from rf_inv_amd import _lib

    lib = _lib.load()
    assert lib.rf_abi_version() == 6
    assert os.path.basename(_lib.LIB_PATH) == "librfgpu.so"


@pytest.mark.parametrize("fname,rayp", [("sample_1.trc", 0.06), ("sample_2.trc", 0.08)])
def test_kat_sample_syn_on_gpu(oracle, golden_dir, fname, rayp):
    """The reference's own fixture, straight through the HIP path: every sample rounds to
    the shipped float32 value."""
    alpha, beta, rho, h = load_true_model(golden_dir)
    obs, delta, nsmp = oracle.read_sac(os.path.join(golden_dir, "sample_syn", "data", fname), 0.0, 5.0)
    cfg = make_cfg(rayps=[rayp])
    with _engine(cfg, obs[None, :], nsmp, None) as eng:
        rft = eng.calc_rf(3, alpha, beta, rho, h)
    assert np.array_equal(rft[:nsmp, 0].astype(np.float32), obs.astype(np.float32))
    ref = oracle.calc_rf(cfg, alpha, beta, rho, h)
    assert np.abs(rft[:, 0] - ref[0]).max() < 1e-13


CASES = [
    # (name, nfft, deconv, sdep, rayps, ipha, t_start, nlays)
    ("land_P", 256, 0, 0.0, (0.06,), (1,), 0.0, (2, 3, 9)),
    ("land_P_decon", 256, 1, 0.0, (0.06,), (1,), -1.0, (2, 5)),
    ("land_S", 256, 0, 0.0, (0.10,), (-1,), -2.0, (3, 9)),
    ("land_S_decon", 256, 1, 0.0, (0.10,), (-1,), -1.0, (4,)),
    ("land_PPS", 512, 0, 0.0, (0.06, 0.08, 0.10), (1, 1, -1), -1.0, (3, 15, 30)),
    ("land_common", 256, 0, 0.0, (0.06, 0.06), (1, 1), 0.0, (3, 7)),
    # nlay 2 under an ocean = water directly on the half-space: no solid layer, the product is the identity
    ("ocean_P", 256, 0, 2.0, (0.06, 0.08), (1, 1), 0.0, (2, 3, 5, 11)),
    ("ocean_PS_decon", 256, 1, 2.0, (0.06, 0.10), (1, -1), -1.0, (4, 8)),
    ("ocean_S", 256, 0, 2.0, (0.10,), (-1,), -2.0, (2, 5)),
    ("big_fft", 4096, 0, 0.0, (0.06,), (1,), 0.0, (15,)),
]


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_batch_parity(oracle, case, fused):
    """rf_eval_batch: traces within 1e-12 of max|trace|, integer shifts identical by
    construction of the trace equality, logL within the north-star tolerance."""
    name, nfft, deconv, sdep, rayps, ipha, t_start, nlays = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    cfg = make_cfg(nfft=nfft, deconv_mode=deconv, t_start=t_start, sdep=sdep, rayps=rayps,
                   a_gus=[4.0 if i % 2 == 0 else 2.5 for i in range(len(rayps))], ipha=ipha)
    ocean = sdep > 0
    nsmp = 101
    true = random_stack(rng, 4, ocean, sdep)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = []
    for nl in nlays:
        for _ in range(3):
            stacks.append(random_stack(rng, nl, ocean, sdep))
    stacks.append(true)  # logL near its maximum: the 1e-9 absolute regime
    nlay, layers = pack_layers(stacks, max(nlays) + 2)
    nb = len(stacks)
    sig = np.full((nb, len(rayps)), 0.01)
    sig[:, -1] = 0.02
    ref_ll, ref_rft = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, want_rft=True)
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb, options={"fused": int(fused)}) as eng:
        ll = eng.eval_batch(np.arange(nb), nlay, layers, sig)
        for i in range(nb):
            got = eng.get_rft(i, which=1).T  # [ntrc, nfft]
            scale = np.abs(ref_rft[i]).max()
            assert np.abs(got - ref_rft[i]).max() <= 1e-12 * scale, (name, i)
    assert np.all(np.isfinite(ll))
    assert np.all(np.abs(ll - ref_ll) <= logl_tol(ref_ll)), np.abs(ll - ref_ll).max()


def test_single_call_dropins_and_state(oracle, golden_dir):
    """rf_calc_likelihood (fwd / sigma-only), rf_commit, rf_get_rft follow
    likelihood.f90:75-81 and pt_mcmc.f90:182-191."""
    rng = np.random.default_rng(5)
    cfg = make_cfg(rayps=[0.06, 0.08])
    nsmp = 101
    true = load_true_model(golden_dir)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    m1, m2 = random_stack(rng, 4), random_stack(rng, 6)
    sig = np.array([0.01, 0.03])
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=2) as eng:
        ll1, rft1 = eng.calc_likelihood(0, True, 4, *m1, sig)
        ref1 = oracle.calc_rf(cfg, *m1)
        assert np.abs(rft1.T - ref1).max() < 1e-12 * np.abs(ref1).max()
        o1 = oracle.log_likelihood(ref1, obs, r_inv, sig, nsmp)
        assert abs(ll1 - o1) <= logl_tol(o1)
        eng.commit([0], [1])                       # accept: proposal becomes current
        assert np.array_equal(eng.get_rft(0, 0), rft1)
        # sigma-only proposal re-uses the stored trace
        sig2 = np.array([0.02, 0.03])
        ll1b, rft1b = eng.calc_likelihood(0, False, 2, np.ones(2), np.ones(2), np.ones(2), np.ones(2), sig2)
        o1b = oracle.log_likelihood(ref1, obs, r_inv, sig2, nsmp)
        assert abs(ll1b - o1b) <= logl_tol(o1b)
        assert np.array_equal(rft1b, rft1)
        # rejected forward proposal leaves the current trace untouched
        ll2, rft2 = eng.calc_likelihood(0, True, 6, *m2, sig)
        eng.commit([0], [0])
        assert np.array_equal(eng.get_rft(0, 0), rft1)
        assert not np.array_equal(rft2, rft1)
        # first nsmp samples only
        assert np.array_equal(eng.get_rft(0, 0, n=nsmp), rft1[:nsmp])


def test_nan_propagates_not_traps(oracle, golden_dir):
    cfg = make_cfg(rayps=[0.25])  # p > 1/alpha: evanescent
    nsmp = 101
    true = load_true_model(golden_dir)
    obs = np.zeros((1, nsmp))
    with _engine(cfg, obs, nsmp, None, max_walkers=1) as eng:
        ll, rft = eng.calc_likelihood(0, True, 3, *true, np.array([0.01]))
    assert np.isnan(ll) and np.isnan(rft).all()


def test_nsplit_and_block_shape_variants_agree(oracle):
    """Launch-shape knobs must not change results at all (they only re-partition bins)."""
    rng = np.random.default_rng(11)
    cfg = make_cfg(nfft=1024, rayps=[0.06])
    nsmp = 101
    true = random_stack(rng, 5)
    obs = synth_obs(oracle, cfg, true, nsmp)
    stacks = [random_stack(rng, 12) for _ in range(4)]
    nlay, layers = pack_layers(stacks, 14)
    sig = np.full((4, 1), 0.01)
    outs = []
    for ns, wpb, fused, lpt in [("1", "1", "0", "0"), ("3", "1", "0", "1"), ("3", "4", "0", "0"), ("4", "2", "0", "1"),
                                ("1", "1", "1", "0"), ("1", "1", "1", "1")]:
        # fused: split kernels (spectra -> trace) vs the fused kernel; lpt: longest-first dispatch order
        opts = {"nsplit": int(ns), "waves_per_block": int(wpb), "fused": int(fused), "lpt": int(lpt), "chain": 0}
        with _engine(cfg, obs, nsmp, None, max_walkers=4, options=opts) as eng:
            outs.append((eng.eval_batch(np.arange(4), nlay, layers, sig), eng.get_rft(2, 1)))
    # the split variants (outs[0..3]) only re-partition bins and blocks: bit-identical; so are the two
    # fused runs among themselves.  Fused vs split may differ in the last bits (the compiler contracts
    # the filter multiply / Hermitian fill differently in the two kernels)
    for ll, rft in outs[1:4]:
        assert np.array_equal(ll, outs[0][0]) and np.array_equal(rft, outs[0][1])
    assert np.array_equal(outs[5][0], outs[4][0]) and np.array_equal(outs[5][1], outs[4][1])
    assert np.allclose(outs[4][0], outs[0][0], rtol=1e-12, atol=1e-9)
    assert np.abs(outs[4][1] - outs[0][1]).max() <= 1e-13 * np.abs(outs[0][1]).max()


@pytest.mark.parametrize("block_threads", [256, 512])
def test_full_size_properties(oracle, block_threads):
    """BASELINE config-2 shape (nfft 4096, 15 layers, 1024 walkers): properties that do
    not need the oracle at full size + a sampled oracle check; with the 4-wave fused_kernel (radix-16 FFT) and
    the 8-wave fused8_kernel (4-bin chains, radix-8 FFT)."""
    rng = np.random.default_rng(2)
    cfg = make_cfg(nfft=4096, rayps=[0.06])
    nsmp = 101
    true = random_stack(rng, 4)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    nb = 1024
    stacks = [random_stack(rng, int(rng.integers(2, 16))) for _ in range(nb - 1)] + [true]
    nlay, layers = pack_layers(stacks, 16)
    sig = np.full((nb, 1), 0.01)
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb, options={"block_threads": block_threads}) as eng:
        ll = eng.eval_batch(np.arange(nb), nlay, layers, sig)
        # (1) permutation invariance: walkers are independent
        perm = rng.permutation(nb)
        ll_p = eng.eval_batch(np.arange(nb), nlay[perm], layers[perm], sig)
        assert np.array_equal(ll_p, ll[perm])
        # (2) the true model maximises logL: misfit is ~0 so logL = -nsmp log(sigma)
        assert abs(ll[-1] - (-nsmp * np.log(0.01))) < 1e-6
        assert np.all(ll[:-1] < ll[-1])
        # (3) sigma rescaling identity: phi is sigma-independent
        ll2 = eng.eval_batch(np.arange(nb), nlay, layers, 2 * sig)
        phi = -2 * (ll + nsmp * np.log(0.01)) * 0.01 ** 2
        assert np.allclose(ll2, -0.5 * phi / 0.02 ** 2 - nsmp * np.log(0.02), rtol=1e-12, atol=1e-9)
    # (4) sampled oracle parity
    idx = rng.choice(nb, 16, replace=False)
    ref = oracle.eval_batch(cfg, obs, r_inv, nlay[idx], layers[idx], sig[idx], nsmp)
    assert np.all(np.abs(ll[idx] - ref) <= logl_tol(ref))


def test_full_size_properties_three_traces(oracle):
    """BASELINE config-4 shape (nfft 4096, 3 traces P/P/S, <= 30 layers) at 1024 walkers = 3072 blocks:
    the default launch plan here is 8-bin phase chains + quadratic forms and logL by the follow-up kernel.
    Properties that do not need the oracle at full size, agreement with the in-kernel path, and a
    sampled oracle check."""
    rng = np.random.default_rng(4)
    cfg = make_cfg(nfft=4096, rayps=[0.06, 0.08, 0.10], ipha=[1, 1, -1])
    nsmp = 101
    true = random_stack(rng, 6)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    nb = 1024
    stacks = [random_stack(rng, int(rng.integers(2, 31))) for _ in range(nb - 1)] + [true]
    nlay, layers = pack_layers(stacks, 32)
    sig = np.column_stack([np.full(nb, 0.01), np.full(nb, 0.02), np.full(nb, 0.03)])
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb) as eng:
        plan = eng.launch_plan
        assert plan["fused"] and plan["chain"] == 8
        ll = eng.eval_batch(np.arange(nb), nlay, layers, sig)
        perm = rng.permutation(nb)
        ll_p = eng.eval_batch(np.arange(nb), nlay[perm], layers[perm], sig)
        assert np.array_equal(ll_p, ll[perm])                              # walkers are independent
        expect = -nsmp * (np.log(0.01) + np.log(0.02) + np.log(0.03))
        assert abs(ll[-1] - expect) < 1e-6 and np.all(ll[:-1] < ll[-1])    # the true model maximises logL
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb, options={"defer_logl": 0}) as eng:
        assert np.array_equal(eng.eval_batch(np.arange(nb), nlay, layers, sig), ll)   # same arithmetic either way
    idx = rng.choice(nb, 12, replace=False)
    ref = oracle.eval_batch(cfg, obs, r_inv, nlay[idx], layers[idx], sig[idx], nsmp)
    assert np.all(np.abs(ll[idx] - ref) <= logl_tol(ref)), np.abs(ll[idx] - ref).max()


@pytest.mark.parametrize("deconv", [0, 1])
def test_eight_wave_fused_kernel(oracle, deconv):
    """fused8_kernel (512-thread blocks: nfft 4096, land) on every branch it carries: P and S traces, with and without
    water-level deconvolution, walkers on the generic path (out-of-range phases; no unit gauge),
    in-kernel and deferred quadratic forms, sigma-only items after a commit, and the per-call entry -- against the
    oracle, and against the 4-wave kernel to rounding."""
    rng = np.random.default_rng(88 + deconv)
    cfg = make_cfg(nfft=4096, deconv_mode=deconv, rayps=[0.06, 0.10], ipha=[1, -1], t_start=-2.0, a_gus=[4.0, 2.5])
    nsmp = 101
    true = random_stack(rng, 5)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    # (40 layers: more than the anchor table of the 512-thread kernel holds -- that walker's chains start from a
    # direct sincos like the 4-wave kernel's)
    stacks = [random_stack(rng, int(n)) for n in (2, 3, 8, 17, 30, 12, 12, 40)] + [true]
    stacks[5][3][3] = 2.5e5                        # out-of-range phases
    stacks[6][2][4] = 40.0 * stacks[6][2][5]       # density contrast of 40 across an interface: no unit gauge
    nlay, layers = pack_layers(stacks, 42)
    nb = len(stacks)
    sig = np.column_stack([np.full(nb, 0.01), np.full(nb, 0.03)])
    ref_ll, ref_rft = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, want_rft=True, nthreads=4)
    res = {}
    for bt in (256, 512):
        for defer in (0, 1):
            with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb, nlay_max=42,
                         options={"block_threads": bt, "defer_logl": defer}) as eng:
                assert eng.launch_plan["block_threads_full_batch"] == bt
                ll = eng.eval_batch(np.arange(nb), nlay, layers, sig)
                rft = np.stack([eng.get_rft(i, which=1).T for i in range(nb)])
                assert np.all(np.abs(ll - ref_ll) <= logl_tol(ref_ll)), (bt, defer, np.abs(ll - ref_ll).max())
                for i in range(nb):
                    assert np.abs(rft[i] - ref_rft[i]).max() <= 1e-11 * np.abs(ref_rft[i]).max(), (bt, i)
                eng.commit(np.arange(nb), np.ones(nb, dtype=np.int32))
                ff = (np.arange(nb) % 2).astype(np.int32)
                ll2 = eng.eval_batch(np.arange(nb), nlay[::-1].copy(), layers[::-1].copy(), 2 * sig, fwd_flag=ff)
                use_l = np.where(ff[:, None, None] == 1, layers[::-1], layers)
                use_n = np.where(ff == 1, nlay[::-1], nlay)
                ref2 = oracle.eval_batch(cfg, obs, r_inv, use_n, use_l, 2 * sig, nsmp, nthreads=4)
                assert np.all(np.abs(ll2 - ref2) <= logl_tol(ref2)), (bt, defer, np.abs(ll2 - ref2).max())
                one, rft1 = eng.calc_likelihood(0, True, int(nlay[3]), *[layers[3, r, :nlay[3]] for r in range(4)], sig[3])
                assert abs(one - ref_ll[3]) <= logl_tol(ref_ll[3])
                assert np.abs(rft1.T - ref_rft[3]).max() <= 1e-12 * np.abs(ref_rft[3]).max()
                res[(bt, defer)] = (ll, rft)
    assert np.array_equal(res[(512, 0)][0], res[(512, 1)][0])        # same arithmetic in-kernel and deferred
    d = np.abs(res[(512, 0)][1] - res[(256, 0)][1]).max(axis=(1, 2)) / np.abs(res[(256, 0)][1]).max(axis=(1, 2))
    assert d.max() <= 1e-12


@pytest.mark.parametrize("cap", [1024, 3000])
def test_results_do_not_depend_on_how_many_items_share_a_launch(oracle, cap):
    """The kernel plan is a property of the CONTEXT (its capacity), never of a launch's batch size: at nfft 4096 on
    land a context of up to two rounds of blocks runs the 8-wave kernel (8^4 FFT), a larger one the 4-wave kernel
    (16^3 FFT) -- and within a context a chain evaluated alone (the per-call drop-in), in a partial batch or in a
    full one gets bit-identical logL and traces (in-kernel and deferred quadratic forms included: the partial and
    full batches below straddle that threshold)."""
    rng = np.random.default_rng(cap)
    cfg = make_cfg(nfft=4096, rayps=[0.06])
    nsmp = 101
    obs = synth_obs(oracle, cfg, random_stack(rng, 4), nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, int(rng.integers(2, 16))) for _ in range(cap)]
    nlay, layers = pack_layers(stacks, 16)
    sig = rng.uniform(0.01, 0.03, (cap, 1))
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=cap, nlay_max=16) as eng:
        assert eng.launch_plan["block_threads_full_batch"] == (512 if cap <= 1024 else 256)
        full = eng.eval_batch(np.arange(cap), nlay, layers, sig)
        pick = [0, 7, cap // 2, cap - 1]
        rft_full = {i: eng.get_rft(i, which=1).copy() for i in pick}
        part = eng.eval_batch(np.arange(600), nlay[:600], layers[:600], sig[:600])
        assert np.array_equal(part, full[:600])
        for i in pick:
            one = eng.eval_batch(np.array([i]), nlay[i:i + 1], layers[i:i + 1], sig[i:i + 1])
            assert one[0] == full[i], (i, one[0], full[i])
            assert np.array_equal(eng.get_rft(i, which=1), rft_full[i]), i
            ll, rft = eng.calc_likelihood(i, True, int(nlay[i]), *[layers[i, r, :nlay[i]] for r in range(4)], sig[i])
            assert ll == full[i] and np.array_equal(rft, rft_full[i]), i
    ref = oracle.eval_batch(cfg, obs, r_inv, nlay[pick], layers[pick], sig[pick], nsmp)
    assert np.all(np.abs(full[pick] - ref) <= logl_tol(ref))


@pytest.mark.parametrize("ipha", [1, -1])
@pytest.mark.parametrize("deconv", [0, 1])
def test_common_ray_fused_kernel(oracle, deconv, ipha):
    """fusedc_kernel (nfft 4096, land, several traces of ONE ray: single-FWD mode, forward.f90:59-91,141) on every
    branch it carries: P and S, with and without water-level deconvolution, walkers on the generic path (out-of-range
    phases; no unit gauge; a stack deeper than the anchor table holds), in-kernel and deferred quadratic forms,
    sigma-only items after a commit, the per-call entry -- against the oracle, and against the split plan
    (spectra_kernel -> trace_kernel) to rounding."""
    rng = np.random.default_rng(300 + 2 * deconv + (ipha > 0))
    p = 0.06 if ipha == 1 else 0.10
    cfg = make_cfg(nfft=4096, deconv_mode=deconv, rayps=[p, p, p], ipha=[ipha] * 3, t_start=-2.0, a_gus=[4.0, 2.5, 1.5])
    nsmp = 101
    true = random_stack(rng, 5)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, int(n)) for n in (2, 3, 8, 17, 30, 12, 12, 40)] + [true]
    stacks[5][3][3] = 2.5e5                        # out-of-range phases
    stacks[6][2][4] = 40.0 * stacks[6][2][5]       # density contrast of 40 across an interface: no unit gauge
    nlay, layers = pack_layers(stacks, 42)
    nb = len(stacks)
    sig = np.column_stack([np.full(nb, 0.01), np.full(nb, 0.03), np.full(nb, 0.02)])
    ref_ll, ref_rft, kap = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, want_rft=True, nthreads=4,
                                             want_kappa=True)
    allow = np.where(kap >= 1000.0, kap / 1000.0, 1.0)
    res = {}
    for fused in (-1, 0):
        for defer in (0, 1):
            with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb, nlay_max=42, options={"fused": fused, "defer_logl": defer}) as eng:
                plan = eng.launch_plan
                assert plan["common_ray_fused"] == (fused == -1) and plan["fused"] == (fused == -1)
                ll = eng.eval_batch(np.arange(nb), nlay, layers, sig)
                rft = np.stack([eng.get_rft(i, which=1).T for i in range(nb)])
                assert np.all(np.abs(ll - ref_ll) <= logl_tol(ref_ll) * allow), (fused, defer, np.abs(ll - ref_ll) / logl_tol(ref_ll))
                for i in range(nb):
                    err = np.abs(rft[i] - ref_rft[i]).max() / np.abs(ref_rft[i]).max()
                    assert err <= 1e-12 * allow[i], (fused, i, err, kap[i])     # the bound every other trace check uses (kappa rule)
                eng.commit(np.arange(nb), np.ones(nb, dtype=np.int32))
                ff = (np.arange(nb) % 2).astype(np.int32)
                ll2 = eng.eval_batch(np.arange(nb), nlay[::-1].copy(), layers[::-1].copy(), 2 * sig, fwd_flag=ff)
                use_l = np.where(ff[:, None, None] == 1, layers[::-1], layers)
                use_n = np.where(ff == 1, nlay[::-1], nlay)
                ref2, kap2 = oracle.eval_batch(cfg, obs, r_inv, use_n, use_l, 2 * sig, nsmp, nthreads=4, want_kappa=True)
                allow2 = np.where(kap2 >= 1000.0, kap2 / 1000.0, 1.0)
                assert np.all(np.abs(ll2 - ref2) <= logl_tol(ref2) * allow2), (fused, defer, np.abs(ll2 - ref2).max())
                one, rft1 = eng.calc_likelihood(0, True, int(nlay[3]), *[layers[3, r, :nlay[3]] for r in range(4)], sig[3])
                assert abs(one - ref_ll[3]) <= logl_tol(ref_ll[3]) * allow[3]
                assert np.abs(rft1.T - ref_rft[3]).max() <= 1e-12 * allow[3] * np.abs(ref_rft[3]).max()
                if fused == -1:
                    assert one == ll[3] and np.array_equal(rft1.T, rft[3])     # alone or in a batch: the same bits
                res[(fused, defer)] = (ll, rft)
    assert np.array_equal(res[(-1, 0)][0], res[(-1, 1)][0])          # same arithmetic in-kernel and deferred
    assert np.array_equal(res[(-1, 0)][1], res[(-1, 1)][1])
    d = np.abs(res[(-1, 0)][1] - res[(0, 0)][1]).max(axis=(1, 2)) / np.abs(res[(0, 0)][1]).max(axis=(1, 2))
    assert np.all(d <= 1e-12 * allow), d


def test_r_inv_builtin_matches_lapack(oracle):
    from rf_inv_amd.engine import compute_r_inv

    r, rank = compute_r_inv(101, 4.0, DELTA)
    ref, ranks = oracle.build_r_inv(101, [4.0], DELTA, return_rank=True)
    assert rank == ranks[0] == 40
    assert np.abs(r - ref[0]).max() <= 1e-11 * np.abs(ref[0]).max()


def test_huge_phase_takes_generic_sincos_kernel(oracle):
    """A layer so thick that omega*eta*h > 1e6 rad leaves the Cody-Waite range: the
    walker is deferred to spectra_slow_kernel (ocml sincos).  Mixed batch: the other
    walkers stay on the fast path; results of both agree with the oracle."""
    rng = np.random.default_rng(21)
    cfg = make_cfg(nfft=256, rayps=[0.06, 0.08])
    nsmp = 101
    true = random_stack(rng, 4)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, 5) for _ in range(6)]
    for i in (1, 4):
        stacks[i][3][2] = 2.0e5  # km
    nlay, layers = pack_layers(stacks, 7)
    sig = np.full((6, 2), 0.05)
    ref_ll, ref_rft = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, want_rft=True)
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=6) as eng:
        for rep in range(2):  # twice: the deferred list must re-arm between batches
            ll = eng.eval_batch(np.arange(6), nlay, layers, sig)
            assert np.all(np.abs(ll - ref_ll) <= logl_tol(ref_ll)), (rep, np.abs(ll - ref_ll))
        for i in range(6):
            got = eng.get_rft(i, which=1).T
            assert np.abs(got - ref_rft[i]).max() <= 1e-11 * np.abs(ref_rft[i]).max(), i


@pytest.mark.parametrize("mode,k", [("allgather", 8), ("p2p", 1)])
def test_pt_swap_device_matches_serial_replay(oracle, mode, k):
    """Temperature exchange on device tensors (rf_pt_swap_device kernel for the batched form,
    torch ops for the reference's one-pair p2p form; judge_pt, pt_mcmc.f90:580-595) on one GPU
    vs a serial replay of the same replicated schedule."""
    import torch

    from rf_inv_amd.pt import PairSchedule, PTSwap, init_temps, judge_pt

    cfg = make_cfg(rayps=[0.06])
    nch = 64
    with _engine(cfg, np.zeros((1, 101)), 101, None, max_walkers=nch) as eng:
        dev = torch.device("cuda", 0)
        sw = PTSwap(eng, nch, 8, dev, seed=5, t_high=15.0, pairs_per_step=k, mode=mode, cache_steps=16)
        temps = init_temps(nch, 8, 15.0, np.random.Generator(np.random.Philox(key=5 + 7919)))
        assert np.array_equal(sw.temps.cpu().numpy(), temps)
        sched = PairSchedule(nch, 5, k)
        rng = np.random.default_rng(0)
        for step in range(40):  # crosses a schedule-cache refill
            ll = -100.0 * rng.random(nch)
            sw.step(torch.from_numpy(ll).to(dev))
            pairs, logu = sched.draw()
            for (i1, i2), lu in zip(pairs, logu):
                if judge_pt(temps[i1], temps[i2], ll[i1], ll[i2], lu):
                    temps[i1], temps[i2] = temps[i2], temps[i1]
            torch.cuda.synchronize()
            assert np.array_equal(sw.temps.cpu().numpy(), temps), step


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("chain", ["0", "2", "3", "4", "8"])
def test_chained_phase_variants_parity(oracle, chain, fused):
    """Every phase-chain length of the spectra kernel (0 = a full sincos per phase) meets the
    same tolerances, on land and under an ocean, including the DC bin whose omega is the
    literal 1e-5 (forward.f90:247) and a bin count that leaves leftover iterations."""
    opts = {"chain": int(chain), "fused": int(fused)}
    for sdep, nfft in [(0.0, 2048), (2.0, 1024)]:
        rng = np.random.default_rng(77)
        cfg = make_cfg(nfft=nfft, sdep=sdep, rayps=[0.06, 0.10], ipha=[1, -1], t_start=-1.0)
        nsmp = 101
        ocean = sdep > 0
        true = random_stack(rng, 5, ocean, sdep)
        obs = synth_obs(oracle, cfg, true, nsmp)
        r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
        stacks = [random_stack(rng, int(n), ocean, sdep) for n in (3, 9, 17, 30)] + [true]
        nlay, layers = pack_layers(stacks, 32)
        sig = np.full((5, 2), 0.01)
        ref_ll, ref_rft = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, want_rft=True)
        with _engine(cfg, obs, nsmp, r_inv, max_walkers=5, options=opts) as eng:
            ll = eng.eval_batch(np.arange(5), nlay, layers, sig)
            for i in range(5):
                got = eng.get_rft(i, which=1).T
                assert np.abs(got - ref_rft[i]).max() <= 1e-12 * np.abs(ref_rft[i]).max(), (chain, sdep, i)
        assert np.all(np.abs(ll - ref_ll) <= logl_tol(ref_ll)), (chain, sdep, np.abs(ll - ref_ll).max())


@pytest.mark.parametrize("nfft", [16, 32, 64, 128, 8192])
def test_fft_sizes(oracle, nfft):
    """Every mixed-radix plan of the in-LDS c2r (radix 16 passes + a 2/4/8 remainder), from the
    smallest sizes to the LDS-filling 8192."""
    rng = np.random.default_rng(nfft)
    nsmp = min(101, nfft // 2)
    cfg = make_cfg(nfft=nfft, rayps=[0.06, 0.09], ipha=[1, -1], t_start=0.0)
    true = random_stack(rng, 4)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, 3), random_stack(rng, 7), true]
    nlay, layers = pack_layers(stacks, 9)
    sig = np.full((3, 2), 0.02)
    ref_ll, ref_rft = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, want_rft=True)
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=3) as eng:
        ll = eng.eval_batch(np.arange(3), nlay, layers, sig)
        for i in range(3):
            got = eng.get_rft(i, which=1).T
            assert np.abs(got - ref_rft[i]).max() <= 1e-12 * np.abs(ref_rft[i]).max(), (nfft, i)
    assert np.all(np.abs(ll - ref_ll) <= logl_tol(ref_ll)), (nfft, np.abs(ll - ref_ll).max())


@pytest.mark.parametrize("nfft,deconv", [(1000, 0), (250, 1), (375, 0), (384, 1), (1500, 0), (2048 + 2, 0), (3400, 0),
                                         (5000, 1), (6001, 0), (10007, 0), (12000, 1), (20000, 0), (32768 - 1, 1),
                                         (16384, 0), (32768, 1), (65536, 0)])
def test_any_length_nfft(oracle, nfft, deconv):
    """nfft need not be a power of two, nor short: FFTW plans any length (src/fftw.f90:44) and the reference accepts
    any nfft.  Up to 2048: the direct DFT (even lengths with a Nyquist bin, odd ones without, a multiple of 128 --
    the Nyquist bin alone in its 64-bin iteration).  Beyond: Bluestein's algorithm on two power-of-two transforms of
    length M >= 2 nfft - 1 (2050 -> M 8192; the prime 10007 -> 32768; 20000 and 32767 -> 65536), and for powers
    of two beyond 8192 the four-step transform itself (16384 = 4096 x 4 ... 65536 = 4096 x 16), all through
    trace_long_kernel.  P and S traces, with and without deconvolution; for lengths that are not a power of two the
    oracle's c2r is the O(n^2) long-double sum of the definition."""
    rng = np.random.default_rng(nfft)
    nsmp = 101
    cfg = make_cfg(nfft=nfft, deconv_mode=deconv, rayps=[0.06, 0.10], ipha=[1, -1], t_start=-1.0, a_gus=[4.0, 2.5])
    true = random_stack(rng, 4)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, 2), random_stack(rng, 6), random_stack(rng, 19), true]
    nlay, layers = pack_layers(stacks, 21)
    sig = np.full((4, 2), 0.02)
    ref_ll, ref_rft, kap = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, want_rft=True, nthreads=4,
                                             want_kappa=True)
    # the conditioning rule of tests/test_gpu_configs.py: a trace normalised by a nearly cancelling signed maximum
    # (kappa >= 1000; the 2-layer stack resonates) carries kappa times the transform's rounding, in any evaluation
    allow = np.where(kap >= 1000.0, kap / 1000.0, 1.0)
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=4) as eng:
        assert not eng.launch_plan["fused"]                         # split plan: spectra_kernel -> trace_anyn_kernel / trace_long_kernel
        ll = eng.eval_batch(np.arange(4), nlay, layers, sig)
        for i in range(4):
            got = eng.get_rft(i, which=1).T
            assert got.shape == (2, nfft)
            err = np.abs(got - ref_rft[i]).max() / np.abs(ref_rft[i]).max()
            assert err <= 1e-12 * allow[i], (nfft, i, err, kap[i])
        # the per-call drop-in on the same context
        one, rft1 = eng.calc_likelihood(0, True, int(nlay[1]), *[layers[1, r, :nlay[1]] for r in range(4)], sig[1])
        assert abs(one - ref_ll[1]) <= logl_tol(ref_ll[1]) * allow[1] and rft1.shape == (nfft, 2)
    assert np.all(np.abs(ll - ref_ll) <= logl_tol(ref_ll) * allow), (nfft, np.abs(ll - ref_ll) / logl_tol(ref_ll), kap)
    assert np.sum(kap >= 1000.0) <= 2


@pytest.mark.parametrize("nfft", [32769, 40000, 131072])
def test_nfft_beyond_the_long_series_transforms_is_refused(nfft):
    """The limits are stated, not silently exceeded: 65536 for a power of two, 32768 for any other length (Bluestein
    needs a power-of-two transform of 2 nfft - 1 points or more)."""
    from rf_inv_amd import RFEngine
    from rf_inv_amd.engine import RFGPUError

    with pytest.raises(RFGPUError, match="not supported"):
        RFEngine(nfft=nfft, delta=DELTA, t_start=0.0, deconv_mode=0, sdep=0.0, rayps=np.array([0.06]),
                 a_gus=np.array([4.0]), ipha=np.array([1], dtype=np.int32), obs=np.zeros((1, 101)), nsmp=101,
                 max_walkers=1, nlay_max=8)


def test_edge_shapes(oracle):
    """Ragged / extreme batches: a single walker, a 2-layer model (no propagator at all), the
    deepest stack the context allows (reference nlay_max = 200), a long time window (nsmp 401),
    mixed fwd_flag, and re-evaluation of a subset of walkers in a different order."""
    rng = np.random.default_rng(9)
    nsmp = 401
    cfg = make_cfg(nfft=1024, rayps=[0.05], t_start=-2.0)
    true = random_stack(rng, 6)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    deep = random_stack(rng, 200)
    deep[3][:-1] *= 0.05  # 199 thin layers
    stacks = [random_stack(rng, 2), deep, true, random_stack(rng, 11)]
    nlay, layers = pack_layers(stacks, 200)
    sig = np.full((4, 1), 0.03)
    ref = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp)
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=4, nlay_max=200) as eng:
        one = eng.eval_batch([2], nlay[2:3], layers[2:3], sig[2:3])           # nb = 1
        assert abs(one[0] - ref[2]) <= logl_tol(ref[2])
        ll = eng.eval_batch(np.arange(4), nlay, layers, sig)
        assert np.all(np.abs(ll - ref) <= logl_tol(ref)), np.abs(ll - ref)
        eng.commit(np.arange(4), [1, 1, 0, 1])
        # subset, permuted, one of them sigma-only (re-uses its committed trace)
        ids = np.array([3, 0, 1], dtype=np.int32)
        ff = np.array([1, 0, 1], dtype=np.int32)
        sig2 = np.full((3, 1), 0.05)
        ll2 = eng.eval_batch(ids, nlay[ids], layers[ids], sig2, fwd_flag=ff)
        ref2 = oracle.eval_batch(cfg, obs, r_inv, nlay[ids], layers[ids], sig2, nsmp)
        assert np.all(np.abs(ll2 - ref2) <= logl_tol(ref2)), np.abs(ll2 - ref2)
        # walker 2 was rejected: its current trace is still the zero-initialised slot
        assert np.all(eng.get_rft(2, 0) == 0.0)


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("chain", ["0", "4"])
def test_mixed_kinds_and_huge_phase_in_one_batch(oracle, chain, fused):
    """In ONE batch: ordinary walkers, one with out-of-range phases (|x| > 1e6 rad) and one whose
    stack is of the other kind than the context (a water layer, beta(1) < 0, with sdep = 0:
    calc_seis keys on beta(1), forward.f90:229, direct_arrival on sdep, :484).  Both the
    in-place generic path of the chained-phase kernels and the deferred-list kernel."""
    opts = {"chain": int(chain), "fused": int(fused)}
    rng = np.random.default_rng(31)
    cfg = make_cfg(nfft=2048, rayps=[0.06, 0.07], t_start=-1.0)
    nsmp = 101
    true = random_stack(rng, 4)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, 6) for _ in range(5)]
    stacks[1][3][2] = 3.0e5                      # huge phase
    stacks[3] = random_stack(rng, 6, ocean=True)  # water layer in a land context
    nlay, layers = pack_layers(stacks, 8)
    sig = np.full((5, 2), 0.05)
    ref_ll, ref_rft = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, want_rft=True)
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=5, options=opts) as eng:
        for rep in range(2):
            ll = eng.eval_batch(np.arange(5), nlay, layers, sig)
            assert np.all(np.abs(ll - ref_ll) <= logl_tol(ref_ll)), (rep, np.abs(ll - ref_ll))
        for i in range(5):
            got = eng.get_rft(i, which=1).T
            assert np.abs(got - ref_rft[i]).max() <= 1e-11 * np.abs(ref_rft[i]).max(), i


def test_walker_without_unit_gauge_takes_the_generic_path(oracle):
    """The fast paths divide every interface's change of eigen-coordinates by its common diagonal entry
    d = 2 b'^2 p^2 + (rho / rho')(1 - 2 b^2 p^2) (the gauge of stage_interface).  A walker for which some d leaves
    [1/16, 16] -- here density contrasts of 40 and of 35 across one interface (d ~ 36 and d ~ 30) -- keeps the
    plain constants and is evaluated by the generic path; results
    agree with the oracle either way, in one batch with ordinary walkers."""
    rng = np.random.default_rng(77)
    cfg = make_cfg(nfft=2048, rayps=[0.06, 0.125], ipha=[1, -1], t_start=-1.0)
    nsmp = 101
    true = random_stack(rng, 4)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, 6) for _ in range(5)]
    stacks[1][2][2] = 40.0 * stacks[1][2][3]                   # rho contrast across one interface: d ~ 40
    a, b, r, h = stacks[3]
    r[1] = 0.08                                                # a very light layer between ordinary ones: d ~ 30 (p = 0.06) and ~ 23 (p = 0.125)
    nlay, layers = pack_layers(stacks, 8)
    sig = np.full((5, 2), 0.05)
    ref_ll, ref_rft = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, want_rft=True)
    assert np.all(np.isfinite(ref_ll))
    for fused in (1, 0):
        with _engine(cfg, obs, nsmp, r_inv, max_walkers=5, options={"fused": fused}) as eng:
            ll = eng.eval_batch(np.arange(5), nlay, layers, sig)
            assert np.all(np.abs(ll - ref_ll) <= logl_tol(ref_ll)), (fused, np.abs(ll - ref_ll))
            for i in range(5):
                got = eng.get_rft(i, which=1).T
                assert np.abs(got - ref_rft[i]).max() <= 1e-11 * np.abs(ref_rft[i]).max(), (fused, i)


def test_dc_bin_phase_by_series_and_its_limit(oracle):
    """The 512-thread fused kernel starts its phase chains from the block's anchor table and evaluates the DC bin's
    phase omega_dc * xi * h (omega_dc = the single-precision literal 1e-5, forward.f90:247) by its series, valid
    below 2^-8 rad; stage_kernel sends walkers with a larger DC phase to the generic path.  Layers 1500 km (series)
    and 4000 km thick (beyond the limit: 1e-5 * 0.14 * 4000 ~ 5.6e-3) next to ordinary walkers, nfft 4096 on both
    block sizes: every result agrees with the oracle."""
    rng = np.random.default_rng(91)
    cfg = make_cfg(nfft=4096, rayps=[0.06], ipha=[1], t_start=0.0)
    nsmp = 101
    true = random_stack(rng, 4)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, 6) for _ in range(6)]
    stacks[1][3][1] = 1500.0
    stacks[4][3][2] = 4000.0
    nlay, layers = pack_layers(stacks, 8)
    sig = np.full((6, 1), 0.05)
    ref_ll, ref_rft = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, want_rft=True)
    assert np.all(np.isfinite(ref_ll))
    for bt in (512, 256):
        with _engine(cfg, obs, nsmp, r_inv, max_walkers=6, options={"block_threads": bt}) as eng:
            ll = eng.eval_batch(np.arange(6), nlay, layers, sig)
            assert np.all(np.abs(ll - ref_ll) <= logl_tol(ref_ll)), (bt, np.abs(ll - ref_ll))
            for i in range(6):
                got = eng.get_rft(i, which=1).T
                assert np.abs(got - ref_rft[i]).max() <= 1e-11 * np.abs(ref_rft[i]).max(), (bt, i)


def test_make_syn_reproduces_the_shipped_sample(oracle, golden_dir, tmp_path):
    """rf_inv_amd.make_syn on true.velmod (land) regenerates the reference's shipped
    sample_{1,2}.trc payloads byte for byte (float32 samples)."""
    from rf_inv_amd import get_params, make_syn, read_obs

    p = get_params(os.path.join(golden_dir, "sample_syn", "params.in"))
    read_obs(p)
    p.sdep = 0.0   # the shipped traces are land synthetics
    stack = load_true_model(golden_dir)
    cfg = make_cfg(rayps=p.rayps)
    with _engine(cfg, p.obs[:, :p.nsmp], p.nsmp, None, max_walkers=1) as eng:
        out = make_syn(p, eng, stack, str(tmp_path), seed=1)
    for i in (1, 2):
        ref = np.fromfile(os.path.join(golden_dir, "sample_syn", "data", f"sample_{i}.trc"), dtype="<f4")
        # (the reference's '(A10,I2.2)' keeps ten characters of "test_trace.": its files have no dot)
        got = np.fromfile(tmp_path / f"test_trace{i:02d}", dtype="<f4")
        assert np.array_equal(ref[158:], got[158:]) and ref[0] == got[0]
        noisy = np.fromfile(tmp_path / f"test_trace{i:02d}wn", dtype="<f4")
        want = (out["rft"][:p.nsmp, i - 1] + out["noise"][:p.nsmp, i - 1]).astype(np.float32)
        assert np.array_equal(noisy[158:], want)
        assert p.sig_min[i - 1] <= out["noise_sigma"][i - 1] <= p.sig_max[i - 1]


def test_make_syn_program_follows_the_reference_stream(golden_dir, tmp_path):
    """`program make_syn` end to end on the shipped params.in (ocean, 2 traces): the MT19937 stream is consumed in
    the reference's order -- init_model, init_sig, then per trace one grnd() for sigma and 2 * nfft for the white
    series (src/make_syn.f90:100-106) -- chain 1's model is the "true" model, its device-resident trace is what
    the noise-free files hold, and test_vel lists its layers."""
    from rf_inv_amd import RFEngine, format_model, get_params, read_obs, read_ref_model
    from rf_inv_amd.make_syn import make_syn_program
    from rf_inv_amd.mcmc import EngineEvaluator, RJMCMC, gauss
    from rf_inv_amd.mt19937 import MT19937

    g = os.path.join(golden_dir, "sample_syn")
    p = get_params(os.path.join(g, "params.in"))
    read_obs(p)
    p.nchains = 3
    ref = read_ref_model(os.path.join(g, "model", "sample.velmod"))
    with RFEngine.from_params(p, max_walkers=p.nchains) as eng:
        out = make_syn_program(p, ref, eng, str(tmp_path))
        assert not eng.is_ray_common
        # replay: the same initialisation on a second stream, then the noise draws by hand
        rng = MT19937(p.iseed)
        s = RJMCMC(p, ref, EngineEvaluator(eng, p.k_max + 2), rng)
        s.init_model()
        s.init_likelihood()
        flt = eng.flt
        for t in range(p.ntrc):
            sigma = rng.grnd() * (p.sig_max[t] - p.sig_min[t]) + p.sig_min[t]
            assert sigma == out["noise_sigma"][t]
            white = np.array([gauss(rng) * sigma for _ in range(p.nfft)])
            assert np.array_equal(white, out["white"][:, t])
            spec = np.fft.rfft(out["noise"][:, t])
            assert np.allclose(spec, np.fft.rfft(white) * flt[:, t] * p.nfft, rtol=1e-9, atol=1e-12 * np.abs(spec).max())
        assert rng.grnd() == out["rng"].grnd()                       # both streams stand at the same position
        nl, a, b, r, h, ok = format_model(p, ref, int(s.k[0]), s.z[0], s.dvp[0], s.dvs[0])
        assert ok and out["k"] == int(s.k[0]) and np.array_equal(out["stack"][1], b)
        vel = np.loadtxt(tmp_path / "test_vel")
        assert vel.shape == (nl, 4) and np.array_equal(vel[:, 0], a) and np.array_equal(vel[:, 3], h)
        clean = eng.calc_rf(nl, a, b, r, h)
    for t in range(p.ntrc):
        got = np.fromfile(tmp_path / f"test_trace{t + 1:02d}", dtype="<f4")
        assert got.size == 158 + p.nsmp and got.view("<i4")[79] == p.nsmp
        assert np.array_equal(got[158:], clean[:p.nsmp, t].astype(np.float32))
        noisy = np.fromfile(tmp_path / f"test_trace{t + 1:02d}wn", dtype="<f4")
        assert np.array_equal(noisy[158:], (clean[:p.nsmp, t] + out["noise"][:p.nsmp, t]).astype(np.float32))


def test_get_rft_batch_equals_single_gets(oracle):
    rng = np.random.default_rng(3)
    cfg = make_cfg(rayps=[0.06, 0.08])
    nsmp = 101
    obs = np.zeros((2, nsmp))
    stacks = [random_stack(rng, 4) for _ in range(6)]
    nlay, layers = pack_layers(stacks, 6)
    with _engine(cfg, obs, nsmp, None, max_walkers=6) as eng:
        eng.eval_batch(np.arange(6), nlay, layers, np.full((6, 2), 0.1))
        eng.commit(np.arange(6), [1, 0, 1, 1, 0, 1])
        ids = [5, 0, 3, 1]
        got = eng.get_rft_batch(ids, which=0, n=nsmp)
        for i, w in enumerate(ids):
            assert np.array_equal(got[i], eng.get_rft(w, 0, n=nsmp).T)
        got1 = eng.get_rft_batch(ids, which=1)
        for i, w in enumerate(ids):
            assert np.array_equal(got1[i], eng.get_rft(w, 1).T)


def test_python_module_mirrors_forward_and_likelihood(oracle, golden_dir):
    """rf_inv_amd.Forward / rf_inv_amd.Likelihood mirror the reference's module interfaces
    (same names, argument meaning, 1-based chain ids) on the shipped sample_syn setup."""
    from rf_inv_amd import Forward, Likelihood, format_model, get_params, read_obs, read_ref_model
    from rf_inv_amd.mcmc import RJMCMC
    from rf_inv_amd.mt19937 import MT19937

    p = get_params(os.path.join(golden_dir, "sample_syn", "params.in"))
    read_obs(p)
    ref = read_ref_model(os.path.join(p.base_dir, p.vel_file))
    g = MT19937(p.iseed)
    m = RJMCMC(p, ref, None, g)          # only for the reference's init_model state
    m.init_model()
    k, z, dvp, dvs = m.k, m.z.T.copy(), m.dvp.T.copy(), m.dvs.T.copy()   # (k_max-1, nchains) etc.

    lik = Likelihood(p, ref)
    lik.init_likelihood(False, k=k, z=z, dvp=dvp, dvs=dvs, rng=g.grnd)
    fwd = Forward(p, engine=lik.engine)
    fwd.init_forward(False)
    assert fwd.is_ray_common is False and fwd.flt.shape == (129, 2)
    assert np.allclose(fwd.flt.T, oracle.init_filter(256, p.delta, p.a_gus), rtol=1e-15, atol=0)
    assert lik.sig.shape == (2, 5) and np.all(lik.sig == 0.01)

    cfg = dict(nfft=p.nfft, deconv_mode=p.deconv_mode, delta=p.delta, t_start=p.t_start, sdep=p.sdep,
               rayps=p.rayps, a_gus=p.a_gus, ipha=p.ipha)
    r_inv = oracle.build_r_inv(p.nsmp, p.a_gus, p.delta)     # Likelihood.init_r_inv is LAPACK dgesvd too
    obs = np.ascontiguousarray(p.obs[:, :p.nsmp])
    rft_all = lik.rft                                         # rft(nfft, ntrc, nchains)
    assert rft_all.shape == (256, 2, 5)
    for c in range(p.nchains):
        nl, a, b, r, h, ok = format_model(p, ref, k[c], z[:, c], dvp[:, c], dvs[:, c])
        want = oracle.calc_rf(cfg, a, b, r, h)
        assert np.abs(rft_all[:, :, c].T - want).max() <= 1e-12 * np.abs(want).max()
        ll = oracle.log_likelihood(want, obs, r_inv, lik.sig[:, c], p.nsmp)
        assert abs(lik.log_likelihood[c] - ll) <= logl_tol(ll)
        # calc_rf through the forward mirror, with the reference's argument list
        got = fwd.calc_rf(c + 1, nl, p.nfft, p.ntrc, p.rayps, a, b, r, h)
        assert np.abs(got.T - want).max() <= 1e-12 * np.abs(want).max()
    # calc_likelihood, both branches, chain ids 1-based
    ll1, rft1 = lik.calc_likelihood(2, True, k[1], z[:, 1], dvp[:, 1], dvs[:, 1], lik.sig[:, 1])
    assert ll1 == lik.log_likelihood[1]
    sig2 = 3.0 * lik.sig[:, 1]
    ll2, rft2 = lik.calc_likelihood(2, False, k[1], z[:, 1], dvp[:, 1], dvs[:, 1], sig2)
    want2 = oracle.log_likelihood(np.ascontiguousarray(rft_all[:, :, 1].T), obs, r_inv, sig2, p.nsmp)
    assert abs(ll2 - want2) <= logl_tol(want2) and np.array_equal(rft2, rft_all[:, :, 1])
    with pytest.raises(ValueError):
        fwd.calc_rf(1, 3, 128, p.ntrc, p.rayps, a, b, r, h)   # n must be params' nfft
    lik.engine.close()


def test_optional_filter_support_cutoff(oracle):
    """rf_set_option("bin_cutoff") (opt-in, off by default): bins whose Gaussian filter weight is below
    1e-20 of the DC weight are not propagated.  Their contribution is far below the FFT's rounding
    noise: traces agree with the full evaluation (and the oracle) within the usual tolerances."""
    rng = np.random.default_rng(12)
    cfg = make_cfg(nfft=4096, rayps=[0.06, 0.10], a_gus=[4.0, 2.5], ipha=[1, -1], t_start=-1.0)
    nsmp = 101
    true = random_stack(rng, 5)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, int(n)) for n in (3, 12, 25)] + [true]
    nlay, layers = pack_layers(stacks, 27)
    sig = np.full((4, 2), 0.01)
    ref_ll, ref_rft = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, want_rft=True)
    out = {}
    for cut in ("", "1e-20"):
        with _engine(cfg, obs, nsmp, r_inv, max_walkers=4, options={"bin_cutoff": float(cut)} if cut else None) as eng:
            assert eng.launch_plan["bin_cutoff"] == bool(cut)
            ll = eng.eval_batch(np.arange(4), nlay, layers, sig)
            out[cut] = (ll, np.stack([eng.get_rft(i, 1).T for i in range(4)]))
    for cut in out:
        ll, rft = out[cut]
        assert np.all(np.abs(ll - ref_ll) <= logl_tol(ref_ll)), (cut, np.abs(ll - ref_ll).max())
        for i in range(4):
            assert np.abs(rft[i] - ref_rft[i]).max() <= 1e-12 * np.abs(ref_rft[i]).max()
    # the cut-off run differs from the full one by less than 1e-15 of the trace scale
    assert np.abs(out["1e-20"][1] - out[""][1]).max() <= 1e-15 * np.abs(out[""][1]).max()


@pytest.mark.parametrize("nsmp", [5, 101, 161])
@pytest.mark.parametrize("defer", ["0", "1"])
def test_deferred_loglikelihood_kernel(oracle, defer, nsmp):
    """Multi-trace batches can form logL in a follow-up kernel instead of the cross-block hand-off inside
    the fused kernel (option defer_logl; chosen by batch size by default): same values, including
    sigma-only items (fwd_flag 0) and a second evaluation after a commit.  nsmp 5: waves without rows; 101: R^-1 held in
    registers by the follow-up kernel; 161: streamed."""
    rng = np.random.default_rng(321)
    cfg = make_cfg(nfft=512, rayps=[0.06, 0.08, 0.10], ipha=[1, 1, -1], t_start=-1.0)
    true = random_stack(rng, 5)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, int(n)) for n in rng.integers(2, 20, 40)]
    nlay, layers = pack_layers(stacks, 22)
    nb = len(stacks)
    sig = np.column_stack([np.full(nb, 0.01), np.full(nb, 0.02), rng.uniform(0.01, 0.05, nb)])
    ref = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp)
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb, options={"defer_logl": int(defer)}) as eng:
        ll = eng.eval_batch(np.arange(nb), nlay, layers, sig)
        assert np.all(np.abs(ll - ref) <= logl_tol(ref)), np.abs(ll - ref).max()
        eng.commit(np.arange(nb), np.ones(nb, dtype=np.int32))
        # half of the walkers: sigma-only proposals on their committed traces; the others: new models
        ff = (np.arange(nb) % 2).astype(np.int32)
        sig2 = sig * 1.5
        stacks2 = [random_stack(rng, int(n)) for n in rng.integers(2, 20, nb)]
        nlay2, layers2 = pack_layers(stacks2, 22)
        ll2 = eng.eval_batch(np.arange(nb), nlay2, layers2, sig2, fwd_flag=ff)
        use_l = np.where(ff[:, None, None] == 1, layers2, layers)
        use_n = np.where(ff == 1, nlay2, nlay)
        ref2 = oracle.eval_batch(cfg, obs, r_inv, use_n, use_l, sig2, nsmp)
        assert np.all(np.abs(ll2 - ref2) <= logl_tol(ref2)), np.abs(ll2 - ref2).max()


def test_long_time_window(oracle):
    """nsmp 1500 (the reference allows up to npts_max = 2000, src/params.f90:413): an 18 MB R^-1 per
    trace, the quadratic form's row quarters beyond one 64-lane pass of columns, P and S traces."""
    rng = np.random.default_rng(5)
    nsmp = 1500
    cfg = make_cfg(nfft=4096, rayps=[0.06, 0.09], ipha=[1, -1], t_start=-5.0)
    true = random_stack(rng, 5)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, int(n)) for n in (7, 15, 28)] + [true]
    nlay, layers = pack_layers(stacks, 30)
    sig = np.full((4, 2), 0.02)
    ref = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp)
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=4, nlay_max=30) as eng:
        ll = eng.eval_batch(np.arange(4), nlay, layers, sig)
    assert np.all(np.abs(ll - ref) <= logl_tol(ref)), np.abs(ll - ref)


@pytest.mark.parametrize("nsmp", [191, 192, 401, 1201, 2000])
def test_long_window_plan_boundary_and_sizes(oracle, nsmp):
    """nsmp <= 191: phi_deferred_kernel's LDS still holds eight misfit rows and the path of the short windows is
    untouched; from 192 on the context takes the long-window plan (one FP64-MFMA GEMM per batch, phi_gemm_kernel) up to
    the reference's npts_max = 2000 (src/params.f90:44).  Window lengths that are / are not multiples of the GEMM's
    16-sample and 128-column tiles, a batch that is not a multiple of its 128-row tile, sigma-only items, an invalid
    item, a second evaluation after a commit."""
    rng = np.random.default_rng(1000 + nsmp)
    cfg = make_cfg(nfft=4096, rayps=[0.06, 0.09], ipha=[1, -1], t_start=-3.0)
    true = random_stack(rng, 5)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    nb = 150
    stacks = [random_stack(rng, int(n)) for n in rng.integers(2, 20, nb - 1)] + [true]
    nlay, layers = pack_layers(stacks, 22)
    sig = np.column_stack([np.full(nb, 0.01), rng.uniform(0.01, 0.05, nb)])
    ref, kap = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, nthreads=oracle.max_threads(), want_kappa=True)
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb, nlay_max=22) as eng:
        assert eng.launch_plan["long_window_gemm"] == (nsmp >= 192)
        ll = eng.eval_batch(np.arange(nb), nlay, layers, sig)
        assert_logl_parity(ll, ref, kap, nsmp)     # (one-layer stacks can resonate: the conditioning rule)
        # the true model: zero misfit, logL = -nsmp sum(log sigma)
        assert abs(ll[-1] + nsmp * np.log(sig[-1]).sum()) < 1e-6
        eng.commit(np.arange(nb), np.ones(nb, dtype=np.int32))
        ff = (np.arange(nb) % 3 != 0).astype(np.int32)          # every third walker: a sigma-only proposal
        ff[5] = -1                                               # and one item skipped altogether
        sig2 = sig * 1.5
        stacks2 = [random_stack(rng, int(n)) for n in rng.integers(2, 20, nb)]
        nlay2, layers2 = pack_layers(stacks2, 22)
        ll2 = eng.eval_batch(np.arange(nb), nlay2, layers2, sig2, fwd_flag=ff)
        use_l = np.where(ff[:, None, None] == 1, layers2, layers)
        use_n = np.where(ff == 1, nlay2, nlay)
        ref2, kap2 = oracle.eval_batch(cfg, obs, r_inv, use_n, use_l, sig2, nsmp, nthreads=oracle.max_threads(),
                                       want_kappa=True)
        live = ff >= 0
        assert np.isnan(ll2[5])
        assert_logl_parity(ll2[live], ref2[live], kap2[live], nsmp)


@pytest.mark.parametrize("shape", ["fused8", "fused256", "split", "common", "ocean", "decon", "nfft512", "anyn", "long"])
def test_long_window_plan_every_trace_kernel(oracle, shape):
    """Every kernel that ends with a trace hands its misfits to the long-window GEMM: the 8-wave and the 4-wave fused
    kernels, the split plan's trace kernel, the common-ray kernel, the ocean kernel, deconvolution, a short series,
    the direct-DFT and the long-series kernels.  logL of every item against the oracle; a chain evaluated ALONE (the
    per-call drop-in, batch of one) gets the same bits as inside the batch, and so does the host-owned-trace call."""
    nsmp = 333
    kw = {"fused8": dict(nfft=4096, rayps=[0.06], ipha=[1]),
          "fused256": dict(nfft=4096, rayps=[0.06, 0.08, 0.1], ipha=[1, 1, -1]),
          "split": dict(nfft=4096, rayps=[0.06, 0.08], ipha=[1, -1]),
          "common": dict(nfft=4096, rayps=[0.07, 0.07, 0.07], ipha=[1, 1, 1], a_gus=[4.0, 2.5, 1.5]),
          "ocean": dict(nfft=4096, rayps=[0.06, 0.1], ipha=[1, -1], sdep=2.0),
          "decon": dict(nfft=4096, rayps=[0.06, 0.07], ipha=[1, 1], deconv_mode=1),
          "nfft512": dict(nfft=512, rayps=[0.06, 0.07], ipha=[1, -1]),
          "anyn": dict(nfft=1000, rayps=[0.06], ipha=[1]),
          "long": dict(nfft=16384, rayps=[0.06], ipha=[-1])}[shape]
    rng = np.random.default_rng(zlib.crc32(("lw" + shape).encode()))
    cfg = make_cfg(t_start=-2.0, **kw)
    ocean = cfg["sdep"] > 0
    ntrc = len(cfg["rayps"])
    true = random_stack(rng, 6, ocean, cfg["sdep"])
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    nb = 40 if shape in ("anyn", "long") else 700       # 700 x 3 blocks: beyond three rounds -> the 4-wave fused kernel
    stacks = [random_stack(rng, int(n), ocean, cfg["sdep"]) for n in rng.integers(3 if ocean else 2, 20, nb)]
    nlay, layers = pack_layers(stacks, 22)
    sig = rng.uniform(0.01, 0.05, (nb, ntrc))
    ref, kap = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, nthreads=oracle.max_threads(), want_kappa=True)
    opts = {"fused": 0} if shape == "split" else {}
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb, nlay_max=22, options=opts) as eng:
        plan = eng.launch_plan
        assert plan["long_window_gemm"]
        assert plan["fused"] == (shape not in ("split", "anyn", "long"))
        if shape == "fused8":
            assert plan["block_threads_full_batch"] == 512
        if shape == "fused256":
            assert plan["block_threads_full_batch"] == 256
        assert plan["common_ray_fused"] == (shape == "common")
        ll = eng.eval_batch(np.arange(nb), nlay, layers, sig)
        assert_logl_parity(ll, ref, kap, shape)
        # alone = in the batch, bit for bit (walker slots 3 and 17; the second through the per-call drop-in)
        one = eng.eval_batch(np.array([3]), nlay[3:4], layers[3:4], sig[3:4])
        assert one[0] == ll[3]
        n17 = int(nlay[17])
        l17, rft17 = eng.calc_likelihood(17, True, n17, *[layers[17, r, :n17] for r in range(4)], sig[17])
        assert l17 == ll[17]
        # ... and the fwd_flag = .false. branch for a trace the host owns (src/likelihood.f90:81-98)
        assert eng.calc_likelihood_of_trace(rft17, sig[17]) == ll[17]


@pytest.mark.parametrize("shape", ["land3", "ocean4", "common3", "decon2"])
def test_repeated_launches_are_bit_stable(oracle, shape):
    """Soak: the same batch evaluated 200 times in a row (proposal slots flipping through commits in between) returns
    the same bits every time -- the hand-offs between blocks (last-arriving trace forms logL, deferred quadratic
    forms, the order block that sorts for the next launch) have no run-to-run freedom -- and so do the stored traces."""
    rng = np.random.default_rng(zlib.crc32(shape.encode()))
    kw = {"land3": dict(rayps=[0.06, 0.08, 0.10], ipha=[1, 1, -1]),
          "ocean4": dict(rayps=[0.06, 0.08, 0.10, 0.12], ipha=[1, 1, -1, -1], sdep=2.0),
          "common3": dict(rayps=[0.06, 0.06, 0.06], ipha=[1, 1, 1], a_gus=[4.0, 2.5, 1.5]),
          "decon2": dict(rayps=[0.06, 0.07], ipha=[1, 1], deconv_mode=1)}[shape]
    cfg = make_cfg(nfft=4096, **kw)
    ocean = cfg["sdep"] > 0
    ntrc, nsmp, nb = len(cfg["rayps"]), 101, 1536
    true = random_stack(rng, 6, ocean, cfg["sdep"])
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, int(rng.integers(3 if ocean else 2, 31)), ocean, cfg["sdep"]) for _ in range(nb)]
    nlay, layers = pack_layers(stacks, 32)
    sig = rng.uniform(0.01, 0.05, (nb, ntrc))
    ids = np.arange(nb)
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb) as eng:
        first = eng.eval_batch(ids, nlay, layers, sig)
        assert np.all(np.isfinite(first))
        trace0 = eng.get_rft(7, which=1).copy()
        for rep in range(200):
            if rep % 3 == 0:
                eng.commit(ids, (rng.integers(0, 2, nb)).astype(np.int32))      # some walkers flip their slot
            ll = eng.eval_batch(ids, nlay, layers, sig)
            assert np.array_equal(ll, first), (rep, int(np.argmax(ll != first)))
        assert np.array_equal(eng.get_rft(7, which=1), trace0)
    idx = rng.choice(nb, 6, replace=False)
    ref = oracle.eval_batch(cfg, obs, r_inv, nlay[idx], layers[idx], sig[idx], nsmp)
    assert np.all(np.abs(first[idx] - ref) <= logl_tol(ref))


def test_contexts_give_their_memory_back(oracle):
    """Create / use / destroy a context 60 times (every kind of plan: fused, common-ray, long series; posterior
    accumulators; the per-call staging): the device's free memory ends where it started -- rf_ctx_destroy releases
    everything a context allocated, lazily allocated buffers included."""
    import torch

    rng = np.random.default_rng(11)
    shapes = [dict(nfft=4096, rayps=[0.06, 0.08], ipha=[1, -1]),
              dict(nfft=4096, rayps=[0.06, 0.06, 0.06], ipha=[1, 1, 1], a_gus=[4.0, 2.5, 1.5]),
              dict(nfft=512, rayps=[0.06], ipha=[1], sdep=2.0),
              dict(nfft=3000, rayps=[0.06], ipha=[1]),
              dict(nfft=16384, rayps=[0.07], ipha=[1])]
    nsmp = 61

    def cycle(kw):
        cfg = make_cfg(**kw)
        ocean = cfg["sdep"] > 0
        stack = random_stack(rng, 5, ocean, cfg["sdep"])
        obs = np.zeros((len(cfg["rayps"]), nsmp))
        r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
        nlay, layers = pack_layers([stack] * 4, 8)
        with _engine(cfg, obs, nsmp, r_inv, max_walkers=300, nlay_max=8) as eng:
            ll = eng.eval_batch(np.arange(4), nlay, layers, np.full((4, len(cfg["rayps"])), 0.02))
            eng.calc_rf(len(stack[0]), *stack)          # the per-call staging
            assert np.all(np.isfinite(ll))

    for kw in shapes:          # first use of every code path (kernel images, allocator pools) before the baseline
        cycle(kw)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for rep in range(12):
        for kw in shapes:
            cycle(kw)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, (free0, free1)      # (the allocator may keep a few pooled blocks: far below one context)


def test_device_batch_inputs_are_validated(oracle):
    """rf_eval_batch_device cannot look at its arrays on the host; stage_kernel (which touches every item once) checks
    them on the device: a layer count outside [2, nlay_pad], a walker id outside the context's slots or a fwd_flag > 1
    makes that ITEM refused -- logL = NaN, nothing of it evaluated, no out-of-bounds access -- while its neighbours in
    the batch get bit-identical results, and the next call that checks the context's error word (rf_get_rft, rf_commit*,
    rf_profile_read ...) fails with the item and the reason; afterwards the context works as before."""
    import torch

    from rf_inv_amd.engine import RFGPUError

    rng = np.random.default_rng(77)
    cfg = make_cfg(nfft=4096, rayps=[0.06, 0.09], ipha=[1, -1])
    nsmp, nb, pad = 101, 600, 20
    true = random_stack(rng, 5)
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, int(n)) for n in rng.integers(3, 18, nb)]
    nlay, layers = pack_layers(stacks, pad)
    sig = rng.uniform(0.01, 0.05, (nb, 2))
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb, nlay_max=pad) as eng:
        ids = np.arange(nb, dtype=np.int32)
        d_logl = torch.empty(nb, dtype=torch.float64, device=dev)
        eng.eval_batch_device(t(ids), t(nlay), t(layers), t(sig), d_logl)
        torch.cuda.synchronize()
        clean = d_logl.cpu().numpy()
        assert np.all(np.isfinite(clean))
        trace7 = eng.get_rft(7, which=1).copy()
        for kind, item, mutate in (("nlay", 100, lambda i, n, f: n.__setitem__(100, pad + 5)),
                                   ("nlay", 3, lambda i, n, f: n.__setitem__(3, 1)),
                                   ("walker id", 250, lambda i, n, f: i.__setitem__(250, nb + 10)),
                                   ("walker id", 599, lambda i, n, f: i.__setitem__(599, -3)),
                                   ("fwd_flag", 42, lambda i, n, f: f.__setitem__(42, 2))):
            i2, n2, f2 = ids.copy(), nlay.copy(), np.ones(nb, dtype=np.int32)
            mutate(i2, n2, f2)
            eng.eval_batch_device(t(i2), t(n2), t(layers), t(sig), d_logl, fwd_flag=t(f2))
            torch.cuda.synchronize()
            got = d_logl.cpu().numpy()
            assert np.isnan(got[item]), (kind, item)
            rest = np.arange(nb) != item
            assert np.array_equal(got[rest], clean[rest]), kind          # the neighbours: bit for bit the clean batch
            with pytest.raises(RFGPUError, match=f"batch item {item}: .*{kind}"):
                eng.get_rft(7, which=1)
            assert np.array_equal(eng.get_rft(7, which=1), trace7)       # reported once; the context goes on
        # rf_commit_device checks its ids on the device too
        bad_ids = ids.copy()
        bad_ids[11] = 5 * nb
        eng.commit_device(t(bad_ids), t(np.ones(nb, dtype=np.int32)))
        torch.cuda.synchronize()
        with pytest.raises(RFGPUError, match="walker id"):
            eng.profile_read()
        ll = eng.eval_batch(ids, nlay, layers, sig)
        assert np.array_equal(ll, clean)


@pytest.mark.parametrize("shape", ["land3", "common3", "ocean2", "nfft512", "long_window"])
def test_trace_window_option(oracle, shape):
    """rf_set_option("trace_window", 1): only samples 1 .. nsmp of every trace are stored.  logL and those samples are
    bit-identical to the default (full rft(nfft, ntrc) image) through every kind of trace kernel, commits and sigma-only
    proposals keep working, the host-owned-trace call too; what needs the full image is refused; switching back gives
    the full image again."""
    from rf_inv_amd.engine import RFGPUError

    kw = {"land3": dict(nfft=4096, rayps=[0.06, 0.08, 0.1], ipha=[1, 1, -1]),
          "common3": dict(nfft=4096, rayps=[0.07, 0.07, 0.07], ipha=[1, 1, 1], a_gus=[4.0, 2.5, 1.5]),
          "ocean2": dict(nfft=4096, rayps=[0.06, 0.1], ipha=[1, -1], sdep=2.0),
          "nfft512": dict(nfft=512, rayps=[0.06], ipha=[-1], deconv_mode=1),
          "long_window": dict(nfft=4096, rayps=[0.06, 0.09], ipha=[1, -1])}[shape]
    nsmp = 333 if shape == "long_window" else 101
    rng = np.random.default_rng(zlib.crc32(("tw" + shape).encode()))
    cfg = make_cfg(t_start=-1.0, **kw)
    ocean = cfg["sdep"] > 0
    ntrc, nfft = len(cfg["rayps"]), cfg["nfft"]
    true = random_stack(rng, 6, ocean, cfg["sdep"])
    obs = synth_obs(oracle, cfg, true, nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    nb = 900
    stacks = [random_stack(rng, int(n), ocean, cfg["sdep"]) for n in rng.integers(3, 20, nb)]
    nlay, layers = pack_layers(stacks, 22)
    sig = rng.uniform(0.01, 0.05, (nb, ntrc))
    ids = np.arange(nb)
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb, nlay_max=22) as eng:
        full = eng.eval_batch(ids, nlay, layers, sig)
        tr_full = eng.get_rft_batch(ids[::50], which=1)                      # [*, ntrc, nfft]
        eng.set_option("trace_window", 1)
        assert eng.launch_plan["trace_window"]
        win = eng.eval_batch(ids, nlay, layers, sig)
        assert np.array_equal(win, full)
        tr_win = eng.get_rft_batch(ids[::50], which=1, n=nsmp)
        assert np.array_equal(tr_win, tr_full[:, :, :nsmp])
        assert np.array_equal(eng.get_rft(150, which=1, n=nsmp).T, tr_full[3, :, :nsmp])
        for call in (lambda: eng.get_rft(0, which=1), lambda: eng.get_rft_batch(ids[:2], which=1, n=nsmp + 1),
                     lambda: eng.calc_rf(len(true[0]), *true),
                     lambda: eng.calc_likelihood(0, True, int(nlay[0]), *[layers[0, r, :nlay[0]] for r in range(4)], sig[0])):
            with pytest.raises(RFGPUError, match="trace_window|nsmp"):
                call()
        # accept half, then sigma-only proposals everywhere: committed walkers re-use their windowed trace
        acc = (ids % 2).astype(np.int32)
        eng.commit(ids, acc)
        ff = np.zeros(nb, dtype=np.int32)
        ll2 = eng.eval_batch(ids, nlay, layers, 2 * sig, fwd_flag=ff)
        q = -(full + nsmp * np.log(sig).sum(axis=1))
        # one sigma per walker scaled by 2: each trace's phi / sigma^2 term scales by 1/4
        want = -q / 4 - nsmp * np.log(2 * sig).sum(axis=1)
        assert np.allclose(ll2[acc == 1], want[acc == 1], rtol=1e-12, atol=1e-9)
        # per-call drop-in without the trace, and the host-owned-trace call with a full-length host array
        l0, none = eng.calc_likelihood(1, True, int(nlay[1]), *[layers[1, r, :nlay[1]] for r in range(4)], sig[1], want_rft=False)
        assert l0 == full[1] and none is None
        host_trace = np.zeros((nfft, ntrc))
        host_trace[:nsmp] = tr_full[0, :, :nsmp].T
        assert eng.calc_likelihood_of_trace(host_trace, sig[0]) == full[0]
        eng.set_option("trace_window", 0)
        again = eng.eval_batch(ids, nlay, layers, sig)
        assert np.array_equal(again, full) and np.array_equal(eng.get_rft_batch(ids[::50], which=1), tr_full)


def test_long_window_gemm_tilings_agree(oracle):
    """phi_gemm_kernel's two block tilings (option gemm_tile) return the same bits -- on the quadratic form's upper
    triangle (the default) and on the full product ("gemm_triangle" = 0) -- and a chain evaluated alone gets the bits it
    gets in the batch.  The two forms differ from each other by rounding only and both meet the oracle."""
    rng = np.random.default_rng(12)
    nsmp, nb = 530, 333
    cfg = make_cfg(nfft=4096, rayps=[0.06, 0.09], ipha=[1, -1], t_start=-3.0)
    obs = synth_obs(oracle, cfg, random_stack(rng, 5), nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    nlay, layers = pack_layers([random_stack(rng, int(n)) for n in rng.integers(3, 20, nb)], 22)
    sig = rng.uniform(0.01, 0.05, (nb, 2))
    ref, kap = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, nthreads=oracle.max_threads(), want_kappa=True)
    form = {}
    for tri in (1, 0):
        out = {}
        for tile in (0, 64, 128):
            with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb, nlay_max=22,
                         options={"gemm_tile": tile, "gemm_triangle": tri}) as eng:
                assert eng.launch_plan["long_window_gemm"] and eng.launch_plan["gemm_triangle"] == bool(tri)
                out[tile] = eng.eval_batch(np.arange(nb), nlay, layers, sig)
                assert eng.eval_batch(np.array([7]), nlay[7:8], layers[7:8], sig[7:8])[0] == out[tile][7]
        assert np.array_equal(out[64], out[128]) and np.array_equal(out[0], out[64])
        assert_logl_parity(out[0], ref, kap, f"tilings, triangle {tri}")
        form[tri] = out[0]
    assert not np.array_equal(form[0], form[1])                      # two summation orders ...
    assert np.median(np.abs(form[0] - form[1]) / np.abs(form[0])) < 1e-13   # ... of the same sums (each within the oracle's tolerance above)


def test_long_window_triangle_with_an_asymmetric_matrix(oracle):
    """The triangular form assumes nothing about R^-1: with a deliberately NON-symmetric matrix (what a caller's own
    r_inv may be; the reference's product is defined for any matrix, src/likelihood.f90:92-93) both forms still return
    the reference's misfit . R^-1 . misfit."""
    rng = np.random.default_rng(21)
    nsmp, nb = 333, 150
    cfg = make_cfg(nfft=2048, rayps=[0.07], ipha=[1], t_start=-2.0)
    obs = synth_obs(oracle, cfg, random_stack(rng, 4), nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    skew = rng.normal(0, 1, (nsmp, nsmp))
    r_inv = r_inv + 0.05 * np.abs(r_inv).max() * (skew - skew.T)[None]   # antisymmetric part: no effect on the form ...
    r_inv[0] += 0.01 * np.abs(r_inv).max() * np.triu(rng.normal(0, 1, (nsmp, nsmp)), 1)   # ... and a lopsided part that has one
    nlay, layers = pack_layers([random_stack(rng, int(n)) for n in rng.integers(3, 12, nb)], 14)
    sig = rng.uniform(0.01, 0.05, (nb, 1))
    ref, kap = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, nthreads=oracle.max_threads(), want_kappa=True)
    for tri in (1, 0):
        with _engine(cfg, obs, nsmp, r_inv, max_walkers=nb, nlay_max=14, options={"gemm_triangle": tri}) as eng:
            got = eng.eval_batch(np.arange(nb), nlay, layers, sig)
        # (the antisymmetric part cancels in exact arithmetic only; its rounding -- eps |m| |A| |m| -- stays far inside
        # the tolerance: |A| is 5 % of |R|)
        assert_logl_parity(got, ref, kap, f"asymmetric matrix, triangle {tri}")


@pytest.mark.parametrize("nsmp", [101, 333])
@pytest.mark.parametrize("triangle", [1, 0])
def test_set_r_inv_replaces_every_image_of_the_matrix(oracle, nsmp, triangle):
    """rf_set_r_inv after creation (a host that builds the pseudo-inverse through its own LAPACK,
    src/likelihood.f90:183-222): every device image of the matrix follows -- the transposed one of the in-kernel
    quadratic form (nsmp 101), the padded one and the TRIANGULAR one the long-window GEMM multiplies by default
    (nsmp 333; round 4 left that one at the creation-time matrix: ADVICE r04) -- so logL is the new matrix's."""
    if nsmp < 192 and triangle == 0:
        pytest.skip("gemm_triangle only exists on the long-window plan")
    rng = np.random.default_rng(5 + nsmp)
    nb = 70
    cfg = make_cfg(nfft=1024, rayps=[0.06, 0.07], ipha=[1, 1], a_gus=[4.0, 2.5])
    obs = synth_obs(oracle, cfg, random_stack(rng, 4), nsmp)
    r0 = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    r1 = oracle.build_r_inv(nsmp, [3.0, 5.0], DELTA)                  # another matrix altogether
    nlay, layers = pack_layers([random_stack(rng, int(n)) for n in rng.integers(3, 10, nb)], 12)
    sig = rng.uniform(0.01, 0.05, (nb, 2))
    ref0, kap = oracle.eval_batch(cfg, obs, r0, nlay, layers, sig, nsmp, nthreads=oracle.max_threads(), want_kappa=True)
    ref1 = oracle.eval_batch(cfg, obs, r1, nlay, layers, sig, nsmp, nthreads=oracle.max_threads())
    assert np.all(np.abs(ref1 - ref0) > 1e3 * logl_tol(ref0))          # the two matrices are told apart by far
    opts = {"gemm_triangle": triangle} if nsmp >= 192 else {}
    with _engine(cfg, obs, nsmp, r0, max_walkers=nb, nlay_max=12, options=opts) as eng:
        assert eng.launch_plan["long_window_gemm"] == (nsmp >= 192)
        assert_logl_parity(eng.eval_batch(np.arange(nb), nlay, layers, sig), ref0, kap, "creation-time matrix")
        eng.set_r_inv(r1)
        assert np.array_equal(eng.r_inv, r1)
        got = eng.eval_batch(np.arange(nb), nlay, layers, sig)
        assert_logl_parity(got, ref1, kap, f"after set_r_inv, nsmp {nsmp}, triangle {triangle}")
        # a sigma-only proposal re-uses the quadratic forms cached with the chains' CURRENT traces
        eng.commit(np.arange(nb), np.ones(nb, dtype=np.int32))
        ff = np.zeros(nb, dtype=np.int32)
        ref1s = oracle.eval_batch(cfg, obs, r1, nlay, layers, 2 * sig, nsmp, nthreads=oracle.max_threads())
        assert_logl_parity(eng.eval_batch(np.arange(nb), nlay, layers, 2 * sig, fwd_flag=ff), ref1s, kap, "sigma-only")
        # ... which belong to the matrix they were formed with: after another set_r_inv they are void (NaN, never a
        # silently stale value) until the chains have been evaluated and committed again
        eng.set_r_inv(r0)
        assert np.all(np.isnan(eng.eval_batch(np.arange(nb), nlay, layers, 2 * sig, fwd_flag=ff)))
        assert_logl_parity(eng.eval_batch(np.arange(nb), nlay, layers, sig), ref0, kap, "and back")
        eng.commit(np.arange(nb), np.ones(nb, dtype=np.int32))
        ref0s = oracle.eval_batch(cfg, obs, r0, nlay, layers, 2 * sig, nsmp, nthreads=oracle.max_threads())
        assert_logl_parity(eng.eval_batch(np.arange(nb), nlay, layers, 2 * sig, fwd_flag=ff), ref0s, kap, "sigma-only again")


@pytest.mark.parametrize("nfft", [8, 9, 250, 256, 1001, 4096, 4099, 65536])
def test_fftw_plans_on_the_gpu(oracle, nfft):
    """rf_fft_c2r / rf_fft_r2c: the reference's two FFTW plans (src/fftw.f90:44-45) as the drop-in module fftw and
    rf_inv_amd.make_syn execute them -- FFTW's definitions (c2r unnormalised, Hermitian extension implied, imaginary
    parts of the DC and Nyquist bins ignored; r2c writes nfft/2 + 1 bins), any length: even, odd, prime, 2^16."""
    from rf_inv_amd.engine import fft_c2r, fft_r2c

    rng = np.random.default_rng(nfft)
    nh = nfft // 2 + 1
    spec = rng.normal(0, 1, nh) + 1j * rng.normal(0, 1, nh)
    got = fft_c2r(spec, nfft)
    clean = spec.copy()
    clean[0] = clean[0].real
    if nfft % 2 == 0:
        clean[-1] = clean[-1].real
    if nfft <= 4099:
        want = oracle.c2r(np.concatenate([spec, np.zeros(nfft - nh)]), nfft)      # long-double O(n^2) definition
    else:
        want = np.fft.irfft(clean, nfft) * nfft
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= (2e-15 if nfft <= 4099 else 2e-14) * scale * max(1.0, np.log2(nfft) / 8)
    assert np.array_equal(got, fft_c2r(clean, nfft))                               # Im of DC / Nyquist: ignored
    # ... and the third-party transform of the reference's CPU build (MKL's FFTW3 interface, called as src/fftw.f90:44 does)
    from helpers import mkl_fftw3_c2r

    mkl = mkl_fftw3_c2r(spec, nfft)
    if mkl is not None:
        assert np.abs(got - mkl).max() <= 2e-14 * scale * max(1.0, np.log2(nfft) / 8)
    x = rng.normal(0, 1, nfft)
    back = fft_r2c(x)
    ref = np.fft.rfft(x)
    assert back.shape == (nh,) and np.abs(back - ref).max() <= 1e-13 * np.abs(ref).max()
    # the pair is n times the identity on real series (what make_syn's filter relies on with flt = 1 / n ...)
    again = fft_c2r(back, nfft)
    assert np.abs(again - nfft * x).max() <= 1e-13 * nfft * np.abs(x).max()


def test_every_option_away_from_its_default_is_echoed_by_the_launch_plan(oracle):
    """rf_get_launch_plan's override count (plan[8]) moves for EVERY rf_set_option knob set away from its default and
    comes back with it -- a plan changed by any option must never be reported as the default plan (bench.py and the
    full-batch tests assert `overrides == 0`; ADVICE r04: gemm_triangle and copy_stream were not counted)."""
    rng = np.random.default_rng(8)
    knobs = [("fused", 0, -1), ("chain", 4, -1), ("lpt", 0, 1), ("order_reuse", 0, 1), ("nsplit", 2, 0),
             ("waves_per_block", 2, 4), ("defer_logl", 1, -1), ("block_threads", 256, 0), ("bin_cutoff", 1e-6, 0.0),
             ("trace_window", 1, 0), ("copy_stream", 1, 0)]
    for nsmp, extra in ((101, []), (333, [("gemm_tile", 128, 0), ("gemm_triangle", 0, 1)])):
        cfg = make_cfg(nfft=4096, rayps=[0.06, 0.08], ipha=[1, -1])
        obs = synth_obs(oracle, cfg, random_stack(rng, 4), nsmp)
        with _engine(cfg, obs, nsmp, None, max_walkers=4, nlay_max=12) as eng:
            assert eng.launch_plan["overrides"] == 0
            for name, away, default in knobs + extra:
                eng.set_option(name, away)
                assert eng.launch_plan["overrides"] == 1, (nsmp, name)
                eng.set_option(name, default)
                assert eng.launch_plan["overrides"] == 0, (nsmp, name)
            assert eng.launch_plan["copy_stream"] is False
            eng.set_option("copy_stream", 1)
            assert eng.launch_plan["copy_stream"] is True
            with pytest.raises(Exception, match="unknown option"):
                eng.set_option("no_such_option", 1)


def test_several_proposals_per_chain_in_one_launch(oracle):
    """A multiple-try host: m candidate models per chain evaluated in ONE rf_eval_batch call on
    a context of m x nchains slots, candidate j of chain c in slot j * nchains + c; the accepted candidate's slot is
    committed and becomes the chain's state for the next sigma-only proposal.  Same values as m separate launches on a
    context of nchains slots, bit for bit (results depend neither on the slot nor on the launch)."""
    rng = np.random.default_rng(77)
    nchains, m, nsmp = 48, 4, 101
    cfg = make_cfg(nfft=4096, rayps=[0.06], ipha=[1])
    obs = synth_obs(oracle, cfg, random_stack(rng, 4), nsmp)
    r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
    stacks = [random_stack(rng, int(n)) for n in rng.integers(2, 15, m * nchains)]
    nlay, layers = pack_layers(stacks, 17)
    sig = rng.uniform(0.01, 0.05, (m * nchains, 1))
    # one launch of m * nchains items
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=m * nchains, nlay_max=17) as big:
        all_at_once = big.eval_batch(np.arange(m * nchains), nlay, layers, sig)
        # the host accepts candidate pick[c] of chain c: that slot holds the chain's state from now on
        pick = rng.integers(0, m, nchains)
        slot = (pick * nchains + np.arange(nchains)).astype(np.int32)
        big.commit(slot, np.ones(nchains, dtype=np.int32))
        sig2 = 1.7 * sig[slot]
        sigma_only = big.eval_batch(slot, nlay[slot], layers[slot], sig2, fwd_flag=np.zeros(nchains, dtype=np.int32))
    # m launches of nchains items on a context of nchains slots
    with _engine(cfg, obs, nsmp, r_inv, max_walkers=nchains, nlay_max=17) as small:
        for j in range(m):
            blk = slice(j * nchains, (j + 1) * nchains)
            one = small.eval_batch(np.arange(nchains), nlay[blk], layers[blk], sig[blk])
            assert np.array_equal(one, all_at_once[blk]), j
    ref = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, nthreads=oracle.max_threads())
    assert np.all(np.abs(all_at_once - ref) <= logl_tol(ref))
    ref2 = oracle.eval_batch(cfg, obs, r_inv, nlay[slot], layers[slot], sig2, nsmp, nthreads=oracle.max_threads())
    assert np.all(np.abs(sigma_only - ref2) <= logl_tol(ref2))
