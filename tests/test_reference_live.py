"""Randomised contexts against the reference ITSELF, run live on the host CPU.

oracle/_ref/cpu_o0/ref_forward_dump = the whole reference built on the CPU (oracle/Makefile.ref: all of its sources
unmodified, MKL's FFTW3 interface; no product code linked, no GPU).  The committed fixtures (tests/golden/ref/) freeze its
outputs on a fixed matrix; here fresh seeded random contexts -- nfft 256 .. 4096 and odd lengths, 1 .. 4 traces, P / S,
ocean, water-level deconvolution, common rays, windows, stacks of 2 .. 31 layers -- go through it at test time:
  -m "not gpu":  the CPU oracle against it;   -m gpu:  the HIP path (rf_eval_batch / rf_calc_rf) against it.
The binaries exist where /root/reference was present at build time and travel with the snapshot (oracle/_ref/ is
git-ignored): without them these tests skip -- the frozen fixtures of tests/test_reference_fixtures.py never do."""
import os

import numpy as np
import pytest

from helpers import DELTA, pack_layers, random_stack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEEDS = [11, 12, 13]
CASES_PER_SEED = 8


def random_context(rng):
    nfft = int(rng.choice([256, 512, 1024, 2048, 4096, 4096, 1000, 1500]))
    ntrc = int(rng.integers(1, 5))
    ocean = bool(rng.integers(0, 2))
    sdep = 2.0 if ocean else 0.0
    deconv = int(rng.integers(0, 2))
    ipha = [int(rng.choice([1, -1])) for _ in range(ntrc)]
    rayps = [float(rng.uniform(0.04, 0.075)) if ph == 1 else float(rng.uniform(0.09, 0.12)) for ph in ipha]
    if rng.integers(0, 4) == 0 and ntrc > 1:              # common rays now and then
        rayps, ipha = [rayps[0]] * ntrc, [ipha[0]] * ntrc
    a_gus = [float(rng.choice([2.5, 4.0, 6.0])) for _ in range(ntrc)]
    t_start = float(rng.choice([0.0, -1.0, -3.0]))
    lo = 3 if ocean else 2
    stacks = [random_stack(rng, int(rng.integers(lo, 32)), ocean, sdep) for _ in range(6)]
    return dict(nfft=nfft, rayps=rayps, ipha=ipha, a_gus=a_gus, deconv=deconv, sdep=sdep, t_start=t_start, stacks=stacks)


def reference_traces(ctx, tmp_path):
    from oracle import gen_golden, refrun

    if not refrun.available("cpu_o0"):
        pytest.skip("oracle/_ref/cpu_o0 not built (no Fortran compiler / reference tree at build time)")
    p = gen_golden.forward_params(ctx["nfft"], ctx["rayps"], ctx["ipha"], ctx["a_gus"], ctx["deconv"], ctx["sdep"], ctx["t_start"])
    work = str(tmp_path)
    refrun.write_run_dir(work, p)
    refrun.write_stacks(os.path.join(work, "stacks.txt"), ctx["stacks"])
    r = refrun.run_forward("cpu_o0", work, len(ctx["stacks"]), ctx["nfft"], p.ntrc)
    cfg = dict(nfft=ctx["nfft"], deconv_mode=ctx["deconv"], delta=DELTA, t_start=ctx["t_start"], sdep=ctx["sdep"],
               rayps=np.asarray(ctx["rayps"], float), a_gus=np.asarray(ctx["a_gus"], float),
               ipha=np.asarray(ctx["ipha"], dtype=np.int32))
    return p, cfg, r


def allowance(oracle, cfg, stack):
    """1e-12 of the trace scale; kappa / 1000 of slack for an ill-conditioned normalising maximum (tests/helpers.py)."""
    if cfg["deconv_mode"] == 1:
        return 1e-12
    nfft = int(cfg["nfft"])
    nh = nfft // 2 + 1
    _, _, _, fv = oracle.calc_rf(cfg, *stack, want_stages=True)
    flt = oracle.init_filter(nfft, cfg["delta"], cfg["a_gus"])
    kap = 1.0
    for t in range(len(cfg["rayps"])):
        cx = np.zeros(nfft, dtype=np.complex128)
        cx[:nh] = fv[t] * flt[t]
        rx = oracle.c2r(cx, nfft)
        kap = max(kap, np.abs(rx).max() / abs(rx.max()))
    return 1e-12 * max(1.0, kap / 1000.0)


@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_against_the_live_reference(oracle, tmp_path, seed):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for case in range(CASES_PER_SEED):
        ctx = random_context(rng)
        d = tmp_path / f"c{case}"
        d.mkdir()
        p, cfg, r = reference_traces(ctx, d)
        assert np.array_equal(oracle.init_filter(ctx["nfft"], DELTA, cfg["a_gus"]), r["flt"])
        for i, st in enumerate(ctx["stacks"]):
            got, npre, _, _ = oracle.calc_rf(cfg, *st, want_stages=True)
            assert np.array_equal(npre, r["npre"][i]), (seed, case, i)
            scale = np.abs(r["rft"][i]).max(axis=1, keepdims=True)
            err = (np.abs(got - r["rft"][i]) / scale).max()
            assert err <= allowance(oracle, cfg, st), (seed, case, i, err, ctx["nfft"], ctx["ipha"], ctx["deconv"], ctx["sdep"])
            worst = max(worst, err)
    print(f"seed {seed}: {CASES_PER_SEED} random contexts x 6 stacks, oracle against the live reference: max |d trace| / scale {worst:.2e}")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_hip_against_the_live_reference(oracle, tmp_path, seed):
    from rf_inv_amd import RFEngine

    rng = np.random.default_rng(seed)
    worst = 0.0
    for case in range(CASES_PER_SEED):
        ctx = random_context(rng)
        d = tmp_path / f"c{case}"
        d.mkdir()
        p, cfg, r = reference_traces(ctx, d)
        n, ntrc = len(ctx["stacks"]), p.ntrc
        nlay, layers = pack_layers(ctx["stacks"], 33)
        with RFEngine(nfft=ctx["nfft"], delta=DELTA, t_start=ctx["t_start"], deconv_mode=ctx["deconv"], sdep=ctx["sdep"],
                      rayps=cfg["rayps"], a_gus=cfg["a_gus"], ipha=cfg["ipha"], obs=np.zeros((ntrc, p.nsmp)), nsmp=p.nsmp,
                      max_walkers=n, nlay_max=33) as eng:
            assert np.array_equal(eng.flt.T, r["flt"]) and eng.is_ray_common == r["common"]
            eng.eval_batch(np.arange(n), nlay, layers, np.full((n, ntrc), 0.02))
            got = eng.get_rft_batch(np.arange(n), which=1)
            one = eng.calc_rf(int(nlay[2]), *ctx["stacks"][2]).T
        for i, st in enumerate(ctx["stacks"]):
            scale = np.abs(r["rft"][i]).max(axis=1, keepdims=True)
            err = (np.abs(got[i] - r["rft"][i]) / scale).max()
            tol = allowance(oracle, cfg, st)
            assert err <= tol, (seed, case, i, err, ctx["nfft"], ctx["ipha"], ctx["deconv"], ctx["sdep"])
            if i == 2:
                assert (np.abs(one - r["rft"][i]) / scale).max() <= tol
            worst = max(worst, err)
    print(f"seed {seed}: {CASES_PER_SEED} random contexts x 6 stacks, HIP against the live reference: max |d trace| / scale {worst:.2e}")


# ---------------------------------------------------------------------------------------------------------------
# calc_likelihood (format_model + calc_rf + misfit + matmul(misfits, r_inv) + logL) against the live reference
# ---------------------------------------------------------------------------------------------------------------
def random_workload(rng):
    ntrc = int(rng.integers(1, 4))
    ocean = bool(rng.integers(0, 2))
    ipha = [int(rng.choice([1, -1])) for _ in range(ntrc)]
    rayps = [float(rng.uniform(0.04, 0.075)) if ph == 1 else float(rng.uniform(0.09, 0.12)) for ph in ipha]
    if rng.integers(0, 4) == 0 and ntrc > 1:
        rayps, ipha = [rayps[0]] * ntrc, [ipha[0]] * ntrc
    return dict(walkers=16, nfft=int(rng.choice([256, 1024, 4096])), rayps=rayps, ipha=ipha,
                a_gus=[float(rng.choice([2.5, 4.0])) for _ in range(ntrc)], k_max=int(rng.choice([6, 15, 30])),
                sdep=2.0 if ocean else 0.0, deconv=int(rng.integers(0, 2)), temps=1, t_end=float(rng.choice([3.0, 5.0, 8.0])),
                desc="random")


def reference_likelihoods(w, seed, tmp_path):
    import sys

    sys.path.insert(0, ROOT)
    import bench
    from oracle import refrun
    from rf_inv_amd import read_ref_model

    if not refrun.available("cpu_o0"):
        pytest.skip("oracle/_ref/cpu_o0 not built (no Fortran compiler / reference tree at build time)")
    p = bench.make_params(w)
    refm = read_ref_model(os.path.join(ROOT, "tests", "golden", "sample_syn", "model", "sample.velmod"))
    nlay, layers, (m_k, m_z, m_dvp, m_dvs) = bench.draw_walkers(p, refm, 1000 * seed, w["walkers"], return_models=True, procs=1)
    rng = np.random.default_rng(seed)
    sig = rng.uniform(0.01, 0.05, (w["walkers"], p.ntrc))
    # observed traces: the reference's own synthetic of the first model (so that one logL sits at its maximum), float32
    work0, work = tmp_path / "truth", tmp_path / "run"
    work0.mkdir(); work.mkdir()
    refrun.write_run_dir(str(work0), p)
    refrun.write_models(str(work0 / "models.txt"), p.k_max, m_k[:1], m_z[:1], m_dvp[:1], m_dvs[:1], sig[:1])
    truth = refrun.run_path("cpu_o0", str(work0), 1, p)
    refrun.write_run_dir(str(work), p, obs=truth["rft"][0])
    refrun.write_models(str(work / "models.txt"), p.k_max, m_k, m_z, m_dvp, m_dvs, sig)
    r = refrun.run_path("cpu_o0", str(work), w["walkers"], p)
    obs = np.stack([truth["rft"][0][t, :p.nsmp].astype(np.float32).astype(np.float64) for t in range(p.ntrc)])
    assert np.array_equal(r["nlay"], nlay) and r["valid"].all()
    return p, refm, obs, sig, (m_k, m_z, m_dvp, m_dvs), nlay, layers, r


@pytest.mark.parametrize("seed", [21, 22])
def test_oracle_log_likelihood_against_the_live_reference(oracle, tmp_path, seed):
    from helpers import logl_tol

    rng = np.random.default_rng(seed)
    for case in range(4):
        w = random_workload(rng)
        d = tmp_path / f"c{case}"
        d.mkdir()
        p, refm, obs, sig, models, nlay, layers, r = reference_likelihoods(w, seed + case, d)
        cfg = dict(nfft=p.nfft, deconv_mode=p.deconv_mode, delta=p.delta, t_start=p.t_start, sdep=p.sdep, rayps=p.rayps,
                   a_gus=p.a_gus, ipha=p.ipha)
        for i in range(len(nlay)):                              # format_model through the Python host mirror == the reference's
            assert np.array_equal(layers[i, :, :nlay[i]], r["layers"][i, :, :nlay[i]]), (seed, case, i)
        ll, kap = oracle.eval_batch(cfg, obs, r["r_inv"], nlay, layers, sig, p.nsmp, want_kappa=True)
        slack = np.where(kap >= 1000.0, kap / 1000.0, 1.0)
        assert (np.abs(ll - r["logl"]) <= logl_tol(r["logl"]) * slack).all(), (seed, case, w, np.abs(ll - r["logl"]) / logl_tol(r["logl"]))
        assert abs(ll[0] + p.nsmp * np.log(sig[0]).sum()) < 1e-2          # the model that made the data: phi ~ 0 (float32 data)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [21, 22])
def test_hip_log_likelihood_against_the_live_reference(oracle, tmp_path, seed):
    from helpers import logl_tol
    from rf_inv_amd import RFEngine

    rng = np.random.default_rng(seed)
    for case in range(4):
        w = random_workload(rng)
        d = tmp_path / f"c{case}"
        d.mkdir()
        p, refm, obs, sig, (m_k, m_z, m_dvp, m_dvs), nlay, layers, r = reference_likelihoods(w, seed + case, d)
        cfg = dict(nfft=p.nfft, deconv_mode=p.deconv_mode, delta=p.delta, t_start=p.t_start, sdep=p.sdep, rayps=p.rayps,
                   a_gus=p.a_gus, ipha=p.ipha)
        n = w["walkers"]
        ids = np.arange(n, dtype=np.int32)
        with RFEngine(nfft=p.nfft, delta=p.delta, t_start=p.t_start, deconv_mode=p.deconv_mode, sdep=p.sdep, rayps=p.rayps,
                      a_gus=p.a_gus, ipha=p.ipha, obs=obs, nsmp=p.nsmp, r_inv=r["r_inv"], max_walkers=n, nlay_max=p.k_max + 2) as eng:
            eng.set_model(p, refm)
            ll = eng.eval_models(ids, m_k, m_z[:, :max(p.k_max - 1, 1)], m_dvp, m_dvs, sig)
        _, kap = oracle.eval_batch(cfg, obs, r["r_inv"], nlay, layers, sig, p.nsmp, want_kappa=True)
        slack = np.where(kap >= 1000.0, kap / 1000.0, 1.0)
        assert (np.abs(ll - r["logl"]) <= logl_tol(r["logl"]) * slack).all(), (seed, case, w, np.abs(ll - r["logl"]) / logl_tol(r["logl"]))
