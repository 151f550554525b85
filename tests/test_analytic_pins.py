"""S-phase and water-level pins that do not come from our own restatement of the reference.

tests/analytic_layered.py computes the receiver function of a stack of solid layers by the reflectivity method
(scattering matrices from solving the boundary conditions, Kennett's addition rules, the reverberation operator)
-- a different formulation from the propagator-matrix chain of src/forward.f90:212-287 that both the oracle
(oracle/rf_oracle.c) and the HIP kernels restate -- plus the textbook receiver-function processing.  Both the
oracle (CPU, `-m "not gpu"`) and the HIP path (`-m gpu`) must reproduce it:

  * S incidence: the S rows of the boundary condition (forward.f90:272-274), `freq_v` as the S receiver function
    (:159-163), the reversed / negated shift map (:185-194), the direct S time over beta (:161), the
    normalisation of an S trace by the VERTICAL maximum (:197-203);
  * water-level deconvolution: R/V for P, V/R for S (:148-153), level 0.001 of the maximum over the nh bins
    (:447-470), no direct-arrival shift (tp = 0) -- on models where the level really clips bins;
  * P incidence and several layers as a control;
  * an ocean over the stack with the receiver on the sea floor (the reference's OBS case, `sdep > 0`): the
    liquid-layer matrix and the sea-floor rows of the boundary condition (forward.f90:276-287, 424-442), the
    direct-arrival time without the water layer (:484), P and S, with and without deconvolution -- from the sea
    floor's boundary conditions solved together with the acoustic waves of the water column, not from a
    liquid-layer propagator.

Agreement is at rounding level (1e-12 of the trace scale), far inside the 1e-9 logL tolerance.
"""
import numpy as np
import pytest

import analytic_layered as al
from helpers import DELTA, make_cfg, random_stack

NFFT = 2048
A_GUS = 4.0
T_START = -3.0

# (name, alpha, beta, rho, h): random crustal stacks + two deliberately resonant ones whose spectra dip
# below the water level (a soft / slow surface layer on a fast half-space)
_rng = np.random.default_rng(20260)
MODELS = [("crust%d" % n, *random_stack(_rng, n)) for n in (2, 3, 4, 6, 9)]
MODELS += [
    ("soft_sediment", np.array([1.6, 6.0, 8.0]), np.array([0.2, 3.5, 4.5]), np.array([1.5, 2.7, 3.3]),
     np.array([0.5, 30.0, 999.0])),
    ("resonant_layer", np.array([0.5, 8.0]), np.array([0.25, 4.6]), np.array([1.0, 3.5]), np.array([0.4, 999.0])),
]
CASES = [(m, ipha, p, dec) for m in MODELS for ipha, p in ((1, 0.06), (-1, 0.10)) for dec in (0, 1)]
IDS = [f"{m[0]}-{'P' if ipha == 1 else 'S'}-{'decon' if dec else 'norm'}" for m, ipha, p, dec in CASES]

# ocean-bottom models: the reference's water layer (alpha 1.5, rho 1.0, beta < 0 as the marker, thickness = sdep)
# on top of random crustal stacks; nlay 2 = water directly on the half-space
SDEP = 2.0
OCEAN_MODELS = [("obs%d" % n, *random_stack(_rng, n, ocean=True, sdep=SDEP)) for n in (2, 3, 5, 8)]
OCEAN_MODELS.append(("deep_water", np.array([1.5, 5.5, 7.9]), np.array([-999.0, 3.1, 4.4]), np.array([1.0, 2.6, 3.3]),
                     np.array([5.0, 12.0, 999.0])))
OCEAN_CASES = [(m, ipha, p, dec) for m in OCEAN_MODELS for ipha, p in ((1, 0.06), (-1, 0.10)) for dec in (0, 1)]
OCEAN_IDS = [f"{m[0]}-{'P' if ipha == 1 else 'S'}-{'decon' if dec else 'norm'}" for m, ipha, p, dec in OCEAN_CASES]


def _expected(model, ipha, p, dec, nfft=NFFT, **kw):
    return al.receiver_function(nfft, DELTA, T_START, A_GUS, p, ipha, dec, *model[1:], **kw)


def _check(got, want, what):
    scale = np.abs(want).max()
    assert np.isfinite(got).all(), what
    assert np.abs(got - want).max() <= 1e-11 * scale, (what, np.abs(got - want).max() / scale)


def test_water_level_is_active_in_the_resonant_models():
    """The deconvolution cases below are only a pin of the water LEVEL if it clips bins."""
    nh = NFFT // 2 + 1
    w = np.arange(nh) * (2.0 * al.PI / (NFFT * DELTA))
    w[0] = al.REFERENCE_QUIRKS["dc_omega"]
    clipped = {}
    for m in MODELS[-2:]:
        for ipha, p in ((1, 0.06), (-1, 0.10)):
            ux, uz = al.surface_response(w, p, ipha, *m[1:])
            amp = np.abs(uz if ipha == 1 else ux) ** 2
            clipped[(m[0], ipha)] = int((amp < 0.001 * amp.max()).sum())
    assert clipped[("resonant_layer", 1)] > 100 and clipped[("resonant_layer", -1)] > 100
    assert clipped[("soft_sediment", -1)] > 10


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_matches_reflectivity_solution(oracle, case):
    model, ipha, p, dec = case
    cfg = make_cfg(nfft=NFFT, deconv_mode=dec, t_start=T_START, rayps=[p], a_gus=[A_GUS], ipha=[ipha])
    got = oracle.calc_rf(cfg, *model[1:])[0]
    _check(got, _expected(model, ipha, p, dec), case[0][0])
    if ipha == -1 and dec == 0:
        # normalised by the signed maximum of the vertical trace, reversed and negated: the minimum is exactly -1
        assert got.min() == -1.0
        # the opposite sign convention for the incident SV wave is a different trace (the pin is not vacuous)
        other = _expected(model, ipha, p, dec, s_polarity=-1.0)
        assert np.abs(other - got).max() > 1e-3 * np.abs(got).max()


@pytest.mark.parametrize("case", OCEAN_CASES, ids=OCEAN_IDS)
def test_oracle_matches_sea_floor_reflectivity_solution(oracle, case):
    model, ipha, p, dec = case
    sdep = float(model[4][0])
    cfg = make_cfg(nfft=NFFT, deconv_mode=dec, t_start=T_START, rayps=[p], a_gus=[A_GUS], ipha=[ipha], sdep=sdep)
    got = oracle.calc_rf(cfg, *model[1:])[0]
    _check(got, _expected(model, ipha, p, dec), case[0][0])


def test_sea_floor_solution_differs_from_the_land_one():
    """The ocean pin is not vacuous: the same solid stack without its water column is a different trace, and so is
    a water column of another depth."""
    m = OCEAN_MODELS[2]
    wet = _expected(m, 1, 0.06, 0)
    dry = al.receiver_function(NFFT, DELTA, T_START, A_GUS, 0.06, 1, 0, *[x[1:] for x in m[1:]])
    assert np.abs(wet - dry).max() > 1e-2 * np.abs(wet).max()
    deeper = (m[0], m[1], m[2], m[3], np.concatenate([[3.0], m[4][1:]]))
    assert np.abs(_expected(deeper, 1, 0.06, 0) - wet).max() > 1e-3 * np.abs(wet).max()


COMMON_A = (4.0, 2.5, 1.5)
COMMON_CASES = [(ipha, p, dec, ocean) for ipha, p in ((1, 0.06), (-1, 0.10)) for dec in (0, 1) for ocean in (0, 1)]


def _common_model(ocean):
    return OCEAN_MODELS[2] if ocean else MODELS[3]


@pytest.mark.parametrize("ipha,p,dec,ocean", COMMON_CASES)
def test_oracle_single_fwd_mode_matches_reflectivity_solution(oracle, ipha, p, dec, ocean):
    """Common ray geometry (forward.f90:59-91, 141): ONE forward computation feeds every trace, which differ by their
    Gaussian filter only -- each must equal the single-trace known answer of its own filter width."""
    model = _common_model(ocean)
    cfg = make_cfg(nfft=NFFT, deconv_mode=dec, t_start=T_START, rayps=[p] * 3, a_gus=list(COMMON_A), ipha=[ipha] * 3,
                   sdep=float(model[4][0]) if ocean else 0.0)
    got = oracle.calc_rf(cfg, *model[1:])
    for t, a in enumerate(COMMON_A):
        want = al.receiver_function(NFFT, DELTA, T_START, a, p, ipha, dec, *model[1:])
        _check(got[t], want, (ipha, dec, ocean, t))


@pytest.mark.gpu
@pytest.mark.parametrize("nfft", [NFFT, 4096])
@pytest.mark.parametrize("ipha,p,dec,ocean", COMMON_CASES)
def test_hip_single_fwd_mode_matches_reflectivity_solution(ipha, p, dec, ocean, nfft):
    """The same through the C ABI, on both launch plans of common rays: the split spectra -> trace kernels (nfft 2048,
    and every ocean context) and, at nfft 4096 on land, fusedc_kernel -- one block per walker, one propagator pass,
    the three traces' tails from the spectra it keeps in registers."""
    from rf_inv_amd import RFEngine

    model = _common_model(ocean)
    nlay = len(model[1])
    with RFEngine(nfft=nfft, delta=DELTA, t_start=T_START, deconv_mode=dec, sdep=float(model[4][0]) if ocean else 0.0,
                  rayps=np.full(3, p), a_gus=np.array(COMMON_A), ipha=np.full(3, ipha, dtype=np.int32),
                  obs=np.zeros((3, 101)), nsmp=101, max_walkers=1, nlay_max=nlay + 2) as eng:
        assert eng.launch_plan["common_ray_fused"] == (nfft == 4096 and not ocean)
        assert eng.launch_plan["fused"] == eng.launch_plan["common_ray_fused"]
        got = eng.calc_rf(nlay, *model[1:])
    for t, a in enumerate(COMMON_A):
        want = al.receiver_function(nfft, DELTA, T_START, a, p, ipha, dec, *model[1:])
        _check(got[:, t], want, (ipha, dec, ocean, t, nfft))


def test_s_trace_is_offset_by_one_sample_like_the_reference(oracle):
    """forward.f90:188: j = mod(nfft + npre - i + 1, nfft) puts lag t_start + i * delta into the 1-based sample i
    -- the lag that nominally belongs to sample i + 1 (the P map of :179 has no such offset).  The physical
    formulation reproduces the oracle only with that quirk; without it the traces differ by a whole sample."""
    model = MODELS[2]
    cfg = make_cfg(nfft=NFFT, deconv_mode=1, t_start=T_START, rayps=[0.10], a_gus=[A_GUS], ipha=[-1])
    got = oracle.calc_rf(cfg, *model[1:])[0]
    saved = al.REFERENCE_QUIRKS["s_lag_samples"]
    try:
        al.REFERENCE_QUIRKS["s_lag_samples"] = 0
        nominal = _expected(model, -1, 0.10, 1)
    finally:
        al.REFERENCE_QUIRKS["s_lag_samples"] = saved
    assert np.abs(nominal - got).max() > 1e-2 * np.abs(got).max()
    assert np.abs(np.roll(nominal, -1) - got).max() <= 1e-11 * np.abs(got).max()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_hip_matches_reflectivity_solution(case, fused):
    """The same known answers straight through the C ABI (rf_calc_rf), fused and split launch plans."""
    from rf_inv_amd import RFEngine

    model, ipha, p, dec = case
    nlay = len(model[1])
    with RFEngine(nfft=NFFT, delta=DELTA, t_start=T_START, deconv_mode=dec, sdep=0.0, rayps=np.array([p]),
                  a_gus=np.array([A_GUS]), ipha=np.array([ipha], dtype=np.int32), obs=np.zeros((1, 101)), nsmp=101,
                  max_walkers=1, nlay_max=nlay + 2, options={"fused": fused}) as eng:
        got = eng.calc_rf(nlay, *model[1:])[:, 0]
    _check(got, _expected(model, ipha, p, dec), (case[0][0], fused))
    if ipha == -1 and dec == 0:
        # the kernel forms the S trace and the vertical trace it is normalised by as the two channels of ONE complex
        # transform (different roundings of the same sequence): -1 to rounding, not bit for bit like the oracle
        assert abs(got.min() + 1.0) <= 1e-14


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("case", OCEAN_CASES, ids=OCEAN_IDS)
def test_hip_matches_sea_floor_reflectivity_solution(case, fused):
    """The ocean-bottom known answers straight through the C ABI (3-column kernels), fused and split launch plans."""
    from rf_inv_amd import RFEngine

    model, ipha, p, dec = case
    nlay = len(model[1])
    with RFEngine(nfft=NFFT, delta=DELTA, t_start=T_START, deconv_mode=dec, sdep=float(model[4][0]), rayps=np.array([p]),
                  a_gus=np.array([A_GUS]), ipha=np.array([ipha], dtype=np.int32), obs=np.zeros((1, 101)), nsmp=101,
                  max_walkers=1, nlay_max=nlay + 2, options={"fused": fused}) as eng:
        got = eng.calc_rf(nlay, *model[1:])[:, 0]
    _check(got, _expected(model, ipha, p, dec), (case[0][0], fused))


@pytest.mark.gpu
def test_hip_c5_traces_match_sea_floor_reflectivity_solution():
    """The C5 trace set (P .06, P .08, S .10, S .12; nfft 4096; 2 km of water) on 2 .. 31-layer stacks through the
    batched entry and the library's default plan (4-bin ocean chains)."""
    from rf_inv_amd import RFEngine
    from helpers import pack_layers

    rng = np.random.default_rng(505)
    stacks = [random_stack(rng, n, ocean=True, sdep=SDEP) for n in (2, 3, 9, 31)]
    rayps, ipha = np.array([0.06, 0.08, 0.10, 0.12]), np.array([1, 1, -1, -1], dtype=np.int32)
    nlay, layers = pack_layers(stacks, 32)
    with RFEngine(nfft=4096, delta=DELTA, t_start=T_START, deconv_mode=0, sdep=SDEP, rayps=rayps,
                  a_gus=np.full(4, A_GUS), ipha=ipha, obs=np.zeros((4, 101)), nsmp=101, max_walkers=4,
                  nlay_max=32) as eng:
        assert eng.launch_plan["fused"]
        eng.eval_batch(np.arange(4), nlay, layers, np.full((4, 4), 0.01))
        for i, st in enumerate(stacks):
            got = eng.get_rft(i, which=1)
            for t in range(4):
                want = al.receiver_function(4096, DELTA, T_START, A_GUS, rayps[t], int(ipha[t]), 0, *st)
                _check(got[:, t], want, (i, t))


@pytest.mark.gpu
@pytest.mark.parametrize("block_threads", [256, 512])
def test_hip_c4_traces_match_reflectivity_solution(block_threads):
    """The C4 trace set (P .06, P .08, S .10; nfft 4096) on 2 .. 29-layer stacks, batched entry: the 4-wave fused kernel
    (8-bin phase chains, radix-16 FFT) and the 8-wave one (4-bin chains, radix-8 FFT)."""
    from rf_inv_amd import RFEngine

    rng = np.random.default_rng(404)
    stacks = [random_stack(rng, n) for n in (2, 5, 12, 29)]
    rayps, ipha = np.array([0.06, 0.08, 0.10]), np.array([1, 1, -1], dtype=np.int32)
    from helpers import pack_layers

    nlay, layers = pack_layers(stacks, 32)
    with RFEngine(nfft=4096, delta=DELTA, t_start=T_START, deconv_mode=0, sdep=0.0, rayps=rayps,
                  a_gus=np.full(3, A_GUS), ipha=ipha, obs=np.zeros((3, 101)), nsmp=101, max_walkers=4,
                  nlay_max=32, options={"block_threads": block_threads}) as eng:
        assert eng.launch_plan["fused"] and eng.launch_plan["chain"] == 8
        assert eng.launch_plan["block_threads_full_batch"] == block_threads
        eng.eval_batch(np.arange(4), nlay, layers, np.full((4, 3), 0.01))
        for i, st in enumerate(stacks):
            got = eng.get_rft(i, which=1)
            for t in range(3):
                want = al.receiver_function(4096, DELTA, T_START, A_GUS, rayps[t], int(ipha[t]), 0, *st)
                _check(got[:, t], want, (i, t))


# ---------------------------------------------------------------------------------------------------------------
# The checker at the shapes the all-walker tests lean on it (tests/test_gpu_configs.py: C4 = 3 traces P .06 / P .08 /
# S .10 on <= 30 layers, C5 = 4 traces P / P / S / S under 2 km of water on <= 31 layers, nfft 4096): the ORACLE
# against the independent reflectivity solution on deep stacks, with and without deconvolution -- on the CPU, so that
# the oracle is pinned there before any GPU comparison uses it (Next #8).
# ---------------------------------------------------------------------------------------------------------------
DEEP_NFFT = 4096
DEEP_LAND = [random_stack(np.random.default_rng(404), n) for n in (2, 5, 12, 29, 30)]
DEEP_OCEAN = [random_stack(np.random.default_rng(505), n, ocean=True, sdep=SDEP) for n in (2, 3, 9, 30, 31)]


@pytest.mark.parametrize("dec", [0, 1])
@pytest.mark.parametrize("ocean", [0, 1])
def test_oracle_deep_stacks_at_the_benchmark_trace_sets(oracle, ocean, dec):
    """C4's (land) and C5's (ocean) trace sets at nfft 4096 on stacks of 2 .. 31 layers: every trace of every stack."""
    if ocean:
        rayps, ipha, stacks, sdep = [0.06, 0.08, 0.10, 0.12], [1, 1, -1, -1], DEEP_OCEAN, SDEP
    else:
        rayps, ipha, stacks, sdep = [0.06, 0.08, 0.10], [1, 1, -1], DEEP_LAND, 0.0
    cfg = make_cfg(nfft=DEEP_NFFT, deconv_mode=dec, t_start=T_START, rayps=rayps, a_gus=[A_GUS] * len(rayps), ipha=ipha,
                   sdep=sdep)
    for i, st in enumerate(stacks):
        got = oracle.calc_rf(cfg, *st)
        assert got.shape == (len(rayps), DEEP_NFFT)
        for t in range(len(rayps)):
            want = al.receiver_function(DEEP_NFFT, DELTA, T_START, A_GUS, rayps[t], ipha[t], dec, *st)
            _check(got[t], want, ("deep", ocean, dec, len(st[0]), t))


def test_oracle_deep_stack_benchmark_geometry(oracle):
    """bench.py's own geometry (t_start 0, the 5 s window of 101 samples the likelihood reads): the first nsmp samples
    of the C4 trace set on a 30-layer stack, oracle against the reflectivity solution."""
    st = DEEP_LAND[-1]
    cfg = make_cfg(nfft=DEEP_NFFT, deconv_mode=0, t_start=0.0, rayps=[0.06, 0.08, 0.10], a_gus=[A_GUS] * 3, ipha=[1, 1, -1])
    got = oracle.calc_rf(cfg, *st)
    for t, (p, ph) in enumerate(((0.06, 1), (0.08, 1), (0.10, -1))):
        want = al.receiver_function(DEEP_NFFT, DELTA, 0.0, A_GUS, p, ph, 0, *st)
        assert np.abs(got[t, :101] - want[:101]).max() <= 1e-11 * np.abs(want).max()
