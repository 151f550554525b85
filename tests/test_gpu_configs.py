"""BASELINE.json configs C3, C4, C5 through the library's DEFAULT launch plan at their full per-GPU batch
(C3 8192, C4 8192, C5 32768 walkers per GPU), with the parallel-tempering swap in the loop, and a seeded randomised
sweep over contexts.  -m gpu only.

Full-size checks: the size-independent properties of the domain (walkers are independent; the true model
maximises logL; phi does not depend on sigma; temperatures follow a serial replay of the replicated swap
schedule) AND the logL of every walker of the batch against the CPU oracle on the same inputs, under the
conditioning (kappa) rule; traces of a sample of walkers incl. the highest walker ids."""
import os
import sys
import zlib

import numpy as np
import pytest

from helpers import DELTA, logl_tol, make_cfg, pack_layers, random_stack, synth_obs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

KAPPA_MIN, KAPPA_SCALE = 1000.0, 1000.0      # the conditioning rule (see above test_randomised_contexts_against_oracle)


def _workload(name, walkers=None):
    """The bench.py workload of that name: params, walker models of rank 0, observed traces (noise-free
    synthetic of a fixed 3-interface model through the ORACLE here), R^-1."""
    import bench
    from oracle import rf_oracle as oracle
    from rf_inv_amd import format_model, read_ref_model

    oracle.build()
    w = dict(bench.WORKLOADS[name])
    if walkers:
        w["walkers"] = walkers
    p = bench.make_params(w)
    ref = read_ref_model(os.path.join(ROOT, "tests", "golden", "sample_syn", "model", "sample.velmod"))
    nb = w["walkers"]
    nlay, layers = bench.draw_walkers(p, ref, 0, nb)
    cfg = dict(nfft=p.nfft, deconv_mode=p.deconv_mode, delta=p.delta, t_start=p.t_start, sdep=p.sdep,
               rayps=p.rayps, a_gus=p.a_gus, ipha=p.ipha)
    zt = np.zeros(max(p.k_max - 1, 1)); dvt = np.zeros(p.k_max); dst = np.zeros(p.k_max)
    zt[:3] = [3.1 + p.sdep, 7.7 + p.sdep, 14.2 + p.sdep]; dst[:3] = [-0.6, 0.2, 0.5]; dst[p.k_max - 1] = 0.9
    nl_t, a_t, b_t, r_t, h_t, ok = format_model(p, ref, 3, zt, dvt, dst)
    assert ok
    true = (a_t, b_t, r_t, h_t)
    obs = np.ascontiguousarray(oracle.calc_rf(cfg, *true)[:, :p.nsmp])
    r_inv = oracle.build_r_inv(p.nsmp, p.a_gus, p.delta)
    # the last walker carries the true model: logL at its maximum, the 1e-9 absolute regime
    nlay[-1] = nl_t
    layers[-1] = 1.0
    for r in range(4):
        layers[-1, r, :nl_t] = true[r]
    return w, p, cfg, obs, r_inv, nlay, layers


def _run_config(name, expect_defer, nsample, extra_check=None, walkers=None, oracle_sample=None):
    import torch

    from oracle import rf_oracle as oracle
    from rf_inv_amd import RFEngine
    from rf_inv_amd.pt import PairSchedule, PTSwap, init_temps, judge_pt

    w, p, cfg, obs, r_inv, nlay, layers = _workload(name, walkers)
    nb, ntrc, nsmp = w["walkers"], p.ntrc, p.nsmp
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    sigv = np.linspace(0.01, 0.02, ntrc)
    sig = np.tile(sigv, (nb, 1))
    dev = torch.device("cuda", 0)
    with RFEngine(nfft=p.nfft, delta=p.delta, t_start=p.t_start, deconv_mode=p.deconv_mode, sdep=p.sdep,
                  rayps=p.rayps, a_gus=p.a_gus, ipha=p.ipha, obs=obs, nsmp=nsmp, r_inv=r_inv, max_walkers=nb,
                  nlay_max=p.k_max + 2) as eng:
        plan = eng.launch_plan
        # the DEFAULT plan is the subject: no option set, production build
        assert plan["overrides"] == 0 and plan["build"] == "production" and plan["fused"] and plan["defer_logl"] == -1
        assert plan["common_ray_fused"] == bool(eng.is_ray_common and ntrc > 1)
        assert plan["chain"] == 8 or plan["common_ray_fused"]   # the "chain" option's default, land and ocean
        # contexts of <= 3 rounds of blocks at nfft 4096 on land (C2) and the common-ray kernel: 512-thread blocks
        assert (plan["block_threads_full_batch"] == 512) == (plan["common_ray_fused"] or nb * ntrc <= 3 * 512)
        # by batch size the library defers the quadratic form + logL to the follow-up kernel(s) here
        # (rfgpu_api.cpp run_batch: >= 2 rounds of blocks with several traces, >= 4 rounds with one)
        blocks, rnd = nb * ntrc, 2 * 256
        if expect_defer == "gemm":
            # long time windows: the misfits always go to HBM and the batch's quadratic forms are one FP64-MFMA GEMM
            assert plan["long_window_gemm"] and nsmp >= 192
        else:
            assert not plan["long_window_gemm"]
            if not plan["common_ray_fused"]:
                assert (blocks >= (2 if ntrc > 1 else 4) * rnd) == expect_defer
        stream = torch.cuda.Stream(device=dev)
        d_ids = torch.arange(nb, dtype=torch.int32, device=dev)
        d_nlay, d_layers = torch.from_numpy(nlay).to(dev), torch.from_numpy(layers).to(dev)
        d_sig = torch.from_numpy(sig).to(dev)
        d_logl = torch.empty(nb, dtype=torch.float64, device=dev)
        tempered = w["temps"] > 1                   # (C2 / c2d: 1024 chains at T = 1, no swap step)
        if tempered:
            swap = PTSwap(eng, nb, w["temps"], dev, seed=99, t_high=15.0, mode="allgather")
            temps = init_temps(nb, max(1, nb // w["temps"]), 15.0, np.random.Generator(np.random.Philox(key=99 + 7919)))
            sched = PairSchedule(nb, 99, swap.k)
        lls = []
        for step in range(3):                       # evaluation + swap, like bench.py's step
            with torch.cuda.stream(stream):
                eng.eval_batch_device(d_ids, d_nlay, d_layers, d_sig, d_logl, stream=stream)
                if tempered:
                    swap.step(d_logl, stream)
            stream.synchronize()
            ll = d_logl.cpu().numpy()
            lls.append(ll)
            if not tempered:
                continue
            pairs, logu = sched.draw()
            for (i1, i2), lu in zip(pairs, logu):
                if judge_pt(temps[i1], temps[i2], ll[i1], ll[i2], lu):
                    temps[i1], temps[i2] = temps[i2], temps[i1]
            assert np.array_equal(swap.temps.cpu().numpy(), temps), step       # serial replay of the schedule
        ll = lls[0]
        assert np.all(np.isfinite(ll))
        assert np.array_equal(lls[1], ll) and np.array_equal(lls[2], ll)       # deterministic; the swap moves temperatures only
        if tempered:
            assert np.sum(temps != init_temps(nb, max(1, nb // w["temps"]), 15.0,
                                              np.random.Generator(np.random.Philox(key=99 + 7919)))) > 0   # swaps did happen
        # (1) walkers are independent: a permuted batch gives the permuted result, bit for bit
        perm = rng.permutation(nb)
        ll_p = eng.eval_batch(np.arange(nb), nlay[perm], layers[perm], sig)
        assert np.array_equal(ll_p, ll[perm])
        # (2) the true model maximises logL: zero misfit, logL = -nsmp * sum(log sigma)
        expect = -nsmp * np.log(sigv).sum()
        assert abs(ll[-1] - expect) < 1e-6 and np.all(ll[:-1] < ll[-1])
        # (3) phi does not depend on sigma: logL(2 sigma) follows from logL(sigma) trace by trace; with one
        # common factor: sum_t phi_t / sigma_t^2 scales by 1/4
        ll2 = eng.eval_batch(np.arange(nb), nlay, layers, 2 * sig)
        q = -(ll + nsmp * np.log(sigv).sum())                                   # = 0.5 sum phi_t / sigma_t^2
        assert np.allclose(ll2, -q / 4 - nsmp * np.log(2 * sigv).sum(), rtol=1e-12, atol=1e-9)
        # (4) EVERY walker of the batch against the oracle, conditioning accounted for (the rule above
        # test_randomised_contexts_against_oracle): an item may exceed the plain tolerance only if kappa >= KAPPA_MIN
        # and must then stay within tolerance * kappa / KAPPA_SCALE; items below KAPPA_MIN get no allowance
        # (oracle_sample: the whole-job-on-one-GPU sizes compare that many walkers -- the first, the last and a random
        # draw -- and mark the others as conditioned well enough not to be looked at)
        if oracle_sample and oracle_sample < nb:
            cmp_idx = np.unique(np.concatenate([np.arange(oracle_sample // 4), np.arange(nb - oracle_sample // 4, nb),
                                                rng.choice(nb, oracle_sample // 2, replace=False)]))
        else:
            cmp_idx = np.arange(nb)
        ref_c, kap_c = oracle.eval_batch(cfg, obs, r_inv, nlay[cmp_idx], layers[cmp_idx], sig[cmp_idx], nsmp,
                                         nthreads=oracle.max_threads(), want_kappa=True)
        ref, kap = np.full(nb, np.nan), np.ones(nb)
        ref[cmp_idx], kap[cmp_idx] = ref_c, kap_c
        d = np.abs(ll[cmp_idx] - ref_c)
        tol = logl_tol(ref_c)
        over = np.nonzero(~(d <= tol))[0]
        for i in over:
            assert kap_c[i] >= KAPPA_MIN, (name, int(cmp_idx[i]), "well-conditioned walker off tolerance", ll[cmp_idx[i]], ref_c[i], kap_c[i])
            assert d[i] <= tol[i] * kap_c[i] / KAPPA_SCALE, (name, int(cmp_idx[i]), ll[cmp_idx[i]], ref_c[i], kap_c[i])
        assert len(over) <= max(2, len(cmp_idx) // 100), (name, len(over))
        worst = int(np.argmax(d / tol))
        report = {"config": name, "walkers": int(nb), "compared": int(len(cmp_idx)), "n_kappa_ge_1000": int(np.sum(kap_c >= KAPPA_MIN)),
                  "n_used_kappa_allowance": int(len(over)), "max_rel_dlogl": float((d / np.abs(ref_c)).max()),
                  "max_abs_dlogl": float(d.max()), "highest_walker_compared": int(cmp_idx.max()),
                  "worst": {"walker": int(cmp_idx[worst]), "nlay": int(nlay[cmp_idx[worst]]), "logl": float(ref_c[worst]),
                            "abs": float(d[worst]), "rel": float(d[worst] / abs(ref_c[worst])),
                            "tolerance_used": float(d[worst] / tol[worst]), "kappa": float(kap_c[worst])}}
        print("full-batch parity:", report)
        out_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out_dir):
            import json

            with open(os.path.join(out_dir, f"parity_full_batch_{name}.json"), "w") as fh:
                json.dump(report, fh, indent=1)
        # traces of sampled walkers (the deepest, the true model, the highest walker ids -- at C5 beyond the 4 GiB
        # mark of the trace array -- and the worst-conditioned ones)
        deep = np.argsort(nlay)[-4:]
        idx = np.unique(np.concatenate([rng.choice(cmp_idx, nsample, replace=False), deep, [nb - 1, nb - 2, nb - 3],
                                        np.argsort(kap)[-2:], np.arange(nb - 1, nb // 2, -(nb // 16))]))
        if len(cmp_idx) < nb:               # (the conditioning of walkers outside the compared sample)
            kap[idx] = oracle.eval_batch(cfg, obs, r_inv, nlay[idx], layers[idx], sig[idx], nsmp, nthreads=oracle.max_threads(),
                                         want_kappa=True)[1]
        _, ref_rft = oracle.eval_batch(cfg, obs, r_inv, nlay[idx], layers[idx], sig[idx], nsmp, want_rft=True,
                                       nthreads=oracle.max_threads())
        eng.eval_batch(np.arange(nb), nlay, layers, sig)                        # proposals = the compared batch again
        got_all = eng.get_rft_batch(idx, which=1)                                # [len(idx), ntrc, nfft]
        for j in range(len(idx)):
            scale = np.abs(ref_rft[j]).max()
            allow = 1e-12 * max(1.0, kap[idx[j]] / KAPPA_SCALE if kap[idx[j]] >= KAPPA_MIN else 1.0)
            assert np.abs(got_all[j] - ref_rft[j]).max() <= allow * scale, (name, int(idx[j]), kap[idx[j]])
        j = int(rng.integers(len(idx)))
        assert np.array_equal(eng.get_rft(int(idx[j]), which=1).T, got_all[j])
        if extra_check:
            extra_check(eng, p, cfg, obs, r_inv, oracle)


def test_c2_default_plan_full_batch():
    """C2 = BASELINE configs[1]: 1024 chains at T = 1, one P trace, nfft 4096, <= 15 layers: two rounds of blocks --
    fused8_kernel (512 threads, 4-bin chains), quadratic form + logL inside the block; no swap step.  Every walker
    against the oracle + the size-independent properties."""
    _run_config("c2", expect_defer=False, nsample=48)


def test_c2d_default_plan_full_batch():
    """C2 with water-level deconvolution (deconv_mode 1, forward.f90:148-153,447-470): every one of the 1024 walkers."""
    _run_config("c2d", expect_defer=False, nsample=48)


def test_c4common_default_plan_full_batch():
    """C4 in single-FWD mode (three traces of ONE ray, Gaussian a 4.0 / 2.5 / 1.5; forward.f90:59-91,141): fusedc_kernel,
    one block per walker, one propagator pass feeding three trace tails.  Every one of the 8192 walkers."""
    _run_config("c4common", expect_defer=True, nsample=36)


def test_c3_default_plan_full_batch():
    """C3: 8192 walkers x 1 trace: the single-trace deferred-logL plan (phi_deferred_kernel forms logL itself)."""
    _run_config("c3", expect_defer=True, nsample=40)


def test_c4_default_plan_full_batch():
    """C4 per-GPU shard: 8192 walkers x (P, P, S) x <= 30 layers: 8-bin chains, misfits to HBM,
    phi_deferred_kernel."""
    _run_config("c4", expect_defer=True, nsample=36)


def test_c5_default_plan_full_batch():
    """C5 at BASELINE's per-GPU batch (16384 chains x 16 temperatures / 8 GPUs = 32768 walkers: 8.6 GB of traces,
    offsets beyond 4 GiB): sdep 2.0, traces P, P, S, S, nfft 4096, <= 31 layers (ocean kernel, 8-bin chains, 3
    propagated columns), plus -- in the same context -- walkers whose own beta(1) >= 0 (a land stack under sdep > 0:
    calc_seis keys on beta(1), forward.f90:229; direct_arrival on sdep, :484)."""

    def land_in_ocean_context(eng, p, cfg, obs, r_inv, oracle):
        rng = np.random.default_rng(55)
        stacks = [random_stack(rng, n) for n in (2, 3, 9, 31)] + [random_stack(rng, 7, True, p.sdep)]
        nlay, layers = pack_layers(stacks, p.k_max + 2)
        sig = np.full((5, p.ntrc), 0.02)
        ref, ref_rft = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, p.nsmp, want_rft=True)
        ll = eng.eval_batch(np.arange(5), nlay, layers, sig)
        assert np.all(np.abs(ll - ref) <= logl_tol(ref)), np.abs(ll - ref)
        for i in range(5):
            got = eng.get_rft(i, which=1).T
            assert np.abs(got - ref_rft[i]).max() <= 1e-12 * np.abs(ref_rft[i]).max(), i

    _run_config("c5", expect_defer=True, nsample=64, extra_check=land_in_ocean_context)


def test_c4d_default_plan_full_batch():
    """C4 with water-level deconvolution (deconv_mode 1: P traces R / V, the S trace V / R, src/forward.f90:148-153,
    447-470; the 256-thread kernel's in-place water level): every one of the 8192 walkers x 3 traces."""
    _run_config("c4d", expect_defer=True, nsample=36)


def test_c5d_default_plan_full_batch():
    """C5 with water-level deconvolution: 32768 walkers x (P, P, S, S) under the ocean layer, every walker."""
    _run_config("c5d", expect_defer=True, nsample=48)


def test_c4full_all_of_configs3_on_one_gpu():
    """ALL of BASELINE configs[3] on one GPU: 65536 walkers x 3 traces (12.9 GB of double-buffered traces).  The
    size-independent properties and the swap replay at full size; every walker against the oracle."""
    _run_config("c4full", expect_defer=True, nsample=48)


def test_c5full_all_of_configs4_on_one_gpu():
    """ALL of BASELINE configs[4] on one GPU: 262144 walkers x 4 traces under the ocean layer, 68.7 GB of
    double-buffered traces (byte offsets to 2^36).  The size-independent properties (permutation invariance bit for bit,
    sigma scaling, the true model at the last walker id, the swap replay) on all 262144; logL of 32768 walkers -- the
    first, the LAST 8192 ids and a random draw -- and traces incl. the highest ids against the oracle."""
    _run_config("c5full", expect_defer=True, nsample=48, oracle_sample=32768)


def test_c4_20s_window_full_batch():
    """C4 with a 20 s time window (nsmp 401, bench.py's c4w20) at the full 8192 walkers: fused_kernel<8,2> leaves the
    misfits in HBM, phi_gemm_kernel forms the 24576 quadratic forms as one GEMM on the FP64 matrix cores.  Every
    walker against the oracle + the size-independent properties + the swap replay, like the BASELINE configs."""
    _run_config("c4w20", expect_defer="gemm", nsample=24)


def test_c4_60s_window_2048_walkers():
    """C4 with a 60 s time window (nsmp 1201: 11.5 MB of R^-1 per trace, bench.py's c4w60), 2048 walkers."""
    _run_config("c4w60", expect_defer="gemm", nsample=12, walkers=2048)


# ---------------------------------------------------------------------------------------------------------
# Seeded randomised sweep (promoted from tests/tools/fuzz_parity.py).
#
# Conditioning rule.  Without deconvolution a trace is divided by maxval(rx) of the filtered VERTICAL trace
# (forward.f90:201-202) -- the signed maximum.  kappa = max|rx| / |maxval(rx)| measures how much of the
# vertical trace's scale cancels in that divisor: its absolute rounding error is ~1e-15..1e-14 of max|rx|
# in ANY double evaluation (the reference's included), i.e. a relative error kappa times that in the
# divisor, in every sample of the trace, and twice that in logL.  Resonating one-layer models reach kappa
# 1e4 .. 1e15 (|logL| 1e11 .. 1e31).  So: an item may exceed the plain tolerance only if its kappa
# (computed from the ORACLE's own vertical spectrum, in the test) is >= KAPPA_MIN, and then it must stay
# within the plain tolerance times kappa / KAPPA_SCALE.  Items with kappa below KAPPA_MIN get no allowance.
# ---------------------------------------------------------------------------------------------------------
def _kappa(oracle, cfg, stack):
    """max over traces of max|rx_v| / |maxval(rx_v)| for the filtered vertical trace of forward.f90:197-201."""
    nfft = int(cfg["nfft"])
    _, _, _, freq_v = oracle.calc_rf(cfg, *stack, want_stages=True)
    flt = oracle.init_filter(nfft, cfg["delta"], cfg["a_gus"])
    k = 1.0
    for t in range(freq_v.shape[0]):
        rx = oracle.c2r(np.concatenate([freq_v[t] * flt[t], np.zeros(nfft - freq_v.shape[1])]), nfft)
        m = rx.max()
        k = max(k, np.inf if m == 0 else np.abs(rx).max() / abs(m))
    return k


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_randomised_contexts_against_oracle(oracle, seed):
    """(seeds 4 and 5: time windows of 12 .. 32 s -- the long-window plan, quadratic forms as one FP64-MFMA GEMM)"""
    from rf_inv_amd import RFEngine

    rng = np.random.default_rng(seed)
    n_items = n_allow = 0
    for case in range(14):
        nfft = int(rng.choice([256, 512, 1024, 2048, 4096]))
        ntrc = int(rng.integers(1, 5))
        ocean = bool(rng.integers(0, 2))
        sdep = 2.0 if ocean else 0.0
        deconv = int(rng.integers(0, 2))
        ipha = [int(rng.choice([1, -1])) for _ in range(ntrc)]
        rayps = [float(rng.uniform(0.04, 0.075)) if ph == 1 else float(rng.uniform(0.09, 0.12)) for ph in ipha]
        if rng.integers(0, 4) == 0 and ntrc > 1:      # common rays now and then
            rayps, ipha = [rayps[0]] * ntrc, [ipha[0]] * ntrc
        a_gus = [float(rng.choice([2.5, 4.0, 6.0])) for _ in range(ntrc)]
        t_start = float(rng.choice([0.0, -1.0, -3.0]))
        nsmp = min(nfft, int(rng.choice([61, 101, 161] if seed <= 3 else [250, 401, 640])))
        kmax = int(rng.choice([6, 15, 30]))
        nb = int(rng.choice([1, 3, 17, 130, 300]))
        cfg = make_cfg(nfft=nfft, deconv_mode=deconv, t_start=t_start, sdep=sdep, rayps=rayps, a_gus=a_gus, ipha=ipha)
        true = random_stack(rng, int(rng.integers(3, 7)), ocean, sdep)
        obs = synth_obs(oracle, cfg, true, nsmp)
        r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
        lo = 3 if ocean else 2
        stacks = [random_stack(rng, int(rng.integers(lo, kmax + 2)), ocean, sdep) for _ in range(nb)]
        # the shallowest stacks are where the normalisation can be ill-conditioned: always a few of them
        stacks += [random_stack(rng, lo, ocean, sdep) for _ in range(4)]
        nb = len(stacks)
        nlay, layers = pack_layers(stacks, kmax + 2)
        sig = rng.uniform(0.01, 0.05, (nb, ntrc))
        ref = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, nthreads=oracle.max_threads())
        with RFEngine(nfft=nfft, delta=cfg["delta"], t_start=t_start, deconv_mode=deconv, sdep=sdep, rayps=cfg["rayps"],
                      a_gus=cfg["a_gus"], ipha=cfg["ipha"], obs=obs, nsmp=nsmp, r_inv=r_inv, max_walkers=nb,
                      nlay_max=kmax + 2) as eng:
            ll = eng.eval_batch(np.arange(nb), nlay, layers, sig)
            eng.commit(np.arange(nb), np.ones(nb, dtype=np.int32))
            # second evaluation: mixed forward / sigma-only items on the committed traces
            ff = rng.integers(0, 2, nb).astype(np.int32)
            sig2 = sig * rng.uniform(0.8, 1.6)
            stacks2 = [random_stack(rng, int(rng.integers(lo, kmax + 2)), ocean, sdep) for _ in range(nb)]
            nlay2, layers2 = pack_layers(stacks2, kmax + 2)
            ll2 = eng.eval_batch(np.arange(nb), nlay2, layers2, sig2, fwd_flag=ff)
        use_l = np.where(ff[:, None, None] == 1, layers2, layers)
        use_n = np.where(ff == 1, nlay2, nlay)
        ref2 = oracle.eval_batch(cfg, obs, r_inv, use_n, use_l, sig2, nsmp, nthreads=oracle.max_threads())
        for got, want, nl_, lay_ in ((ll, ref, nlay, layers), (ll2, ref2, use_n, use_l)):
            assert np.array_equal(np.isnan(got), np.isnan(want)), (seed, case)
            fin = np.isfinite(want)
            d = np.abs(got - want)
            n_items += int(fin.sum())
            for i in np.nonzero(fin & ~(d <= logl_tol(want)))[0]:
                n = int(nl_[i])
                kap = _kappa(oracle, cfg, tuple(lay_[i, r, :n] for r in range(4))) if deconv == 0 else 1.0
                assert kap >= KAPPA_MIN, (seed, case, int(i), "well-conditioned item off tolerance", got[i], want[i], kap)
                assert d[i] <= logl_tol(want[i]) * kap / KAPPA_SCALE, (seed, case, int(i), got[i], want[i], kap)
                n_allow += 1
    # the allowance is the exception, not the rule (every batch deliberately carries four of the shallowest stacks)
    assert n_allow <= max(2, n_items // 50), (n_allow, n_items)


def _replay_bench_state(dump, nb, ntemps, nranks):
    """Final temperatures of a bench.py run == serial replay of the replicated swap schedule on the logL the run
    produced (rf_inv_amd.pt.replay_swap_schedule: the check bench.py applies to itself when N > 1); swaps across the
    rank boundary present."""
    from rf_inv_amd.pt import replay_swap_schedule

    st = np.load(dump)
    temps, logl = st["temps"], st["logl"]
    assert temps.shape == (nranks, nb)
    r = replay_swap_schedule(temps, logl, nb, ntemps, int(st["swap_steps"]), int(st["pairs_per_step"]), 1234, 15.0)
    assert r["ok"]
    assert r["moved"] > 0                       # swaps did happen
    assert r["cross_rank_swaps"] > 0            # ... also across the ranks' blocks
    return st


@pytest.mark.parametrize("launcher", ["torchrun", "direct"])
def test_bench_two_ranks_on_one_gpu_swap_matches_serial_replay(tmp_path, launcher):
    """bench.py's N > 1 path end to end as two ranks that share the one GPU, with gloo as the process-group backend
    (RCCL cannot put two ranks on one device) -- sharding, barrier, max-over-ranks timing, the gathered temperature
    exchange (rf_pt_swap_gathered_device) and the JSON line.  launcher = torchrun: the driver's own command line;
    direct: `python bench.py --gpus 2` with no launcher around it -- bench.py starts the two ranks itself."""
    import json
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dump = tmp_path / "state.npz"
    nb, ntemps, steps, warm = 256, 8, 5, 1
    env = dict(os.environ, RFGPU_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c3", "--walkers", str(nb), "--steps", str(steps),
            "--warmup", str(warm), "--prewarm-seconds", "0", "--no-cpu-baseline", "--also", "", "--dump-state", str(dump)]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == steps
    # (the line carries 6 significant digits of each number)
    assert abs(d["value"] - 2 * nb * steps / (d["ms_per_step"] * 1e-3 * steps)) < 1e-4 * d["value"]
    assert d["config"]["parallelism"] == "walkers sharded x2"
    assert d["config"]["rccl"]["ranks"] == 0 and d["config"]["rccl"]["transport"] == "process_group"
    assert len(line) < 4096 and line == r.stdout.splitlines()[-1]          # one compact line, the last of stdout
    assert d["swap_replay_ok"] is True and d["cross_rank_swaps"] > 0       # the run's own check of its exchange
    st = _replay_bench_state(dump, nb, ntemps, 2)
    assert int(st["swap_steps"]) == 16 + warm + steps


def test_bench_rccl_route_with_two_ranks_on_one_gpu(tmp_path):
    """bench.py's N > 1 path over librfgpu's own communicator -- the route a multi-GPU SCALE run takes:
    open_exchange's bootstrap (rank 0's id through the process group, rf_comm_init on every rank, unanimous
    agreement), then rf_pt_swap_allgather_device every step -- as two ranks on the one GPU, with librfgpu pointed at
    the RCCL test double (tests/c/rccl_double.cpp; real RCCL refuses two ranks on one device).  The JSON line says
    config.rccl.ranks == 2 and the final temperatures equal the serial replay, swaps across the rank boundary
    included."""
    import json
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = str(tmp_path / "librccl_double.so")
    subprocess.run([hipcc, "-shared", "-fPIC", "-O2", "-o", lib, os.path.join(ROOT, "tests", "c", "rccl_double.cpp")],
                   check=True, capture_output=True, timeout=300)
    dump = tmp_path / "state.npz"
    nb, ntemps, steps, warm = 256, 8, 5, 1
    env = dict(os.environ, RFGPU_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c3", "--walkers", str(nb), "--steps",
           str(steps), "--warmup", str(warm), "--prewarm-seconds", "0", "--no-cpu-baseline", "--also", "", "--dump-state",
           str(dump), "--rccl-library", lib, "--detail-file", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert len(r.stdout.splitlines()[-1]) < 4096
    d = json.loads(r.stdout.splitlines()[-1])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "walkers sharded x2"
    assert d["config"]["rccl"]["ranks"] == 2 and d["config"]["rccl"]["library"] == lib
    assert d["config"]["rccl"]["transport"] == "rccl_allgather" and d["config"]["rccl"]["control_plane"] == "gloo"
    assert d["swap_replay_ok"] is True and d["cross_rank_swaps"] > 0
    full = json.load(open(tmp_path / "detail.json"))                       # the full record of the same run
    assert full["value"] == pytest.approx(d["value"], rel=1e-5) and full["swap_replay"]["swap_steps"] == 16 + warm + steps
    assert full["config"]["launch_plan"]["build"] == "production"
    st = _replay_bench_state(dump, nb, ntemps, 2)
    assert int(st["swap_steps"]) == 16 + warm + steps


def test_bench_eight_ranks_over_the_rccl_double(tmp_path):
    """The command line of the driver's 8-GPU SCALE run -- `bench.py --gpus 8`, workload c4 -- as eight ranks on the one
    GPU over the RCCL test double: gloo control plane, rf_comm_init with nranks = 8, one grouped pair of all-gathers per
    step over eight rank blocks, the run's own replay of the swap schedule over all 8 x 1024 walkers, ONE compact line."""
    import json
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = str(tmp_path / "librccl_double.so")
    subprocess.run([hipcc, "-shared", "-fPIC", "-O2", "-o", lib, os.path.join(ROOT, "tests", "c", "rccl_double.cpp")],
                   check=True, capture_output=True, timeout=300)
    dump = tmp_path / "state.npz"
    nb, ntemps, steps, warm, nr = 1024, 8, 4, 1, 8
    env = dict(os.environ, RFGPU_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nr), "--workload", "c4", "--walkers", str(nb),
           "--steps", str(steps), "--warmup", str(warm), "--prewarm-seconds", "0", "--dump-state", str(dump),
           "--rccl-library", lib, "--detail-file", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = r.stdout.splitlines()
    assert len(out) == 1 and len(out[0]) < 4096                 # the ranks' relay prints rank 0's line and nothing else
    d = json.loads(out[0])
    assert d["n_gpus"] == nr and d["config"]["parallelism"] == f"walkers sharded x{nr}" and d["scaling"] == "weak"
    assert d["config"]["rccl"]["ranks"] == nr and d["config"]["rccl"]["transport"] == "rccl_allgather"
    assert d["config"]["walkers_per_gpu"] == nb and d["config"]["ntrc"] == 3
    assert d["value"] == pytest.approx(nr * nb / (d["ms_per_step"] * 1e-3), rel=1e-4)
    assert d["swap_replay_ok"] is True and d["cross_rank_swaps"] > 0
    assert "cpu_baseline" not in d                              # rank 0 at N = 1 only
    st = _replay_bench_state(dump, nb, ntemps, nr)
    assert int(st["swap_steps"]) == 16 + warm + steps


def test_bench_gpus_flag_is_honoured_or_refused(tmp_path):
    """`python bench.py --gpus 2` the way the driver's SCALE run may invoke it (no launcher, RCCL backend): on a node
    with two GPUs it must run two ranks over librfgpu's RCCL communicator (n_gpus == 2, config.rccl.ranks == 2, final
    temperatures == the serial replay); on a one-GPU box it must REFUSE with a non-zero exit code and a message --
    never print a line that carries a rate.  And a launcher whose WORLD_SIZE disagrees with --gpus is refused too."""
    import json
    import subprocess

    import torch

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                              "RFGPU_BENCH_BACKEND")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dump = tmp_path / "state.npz"
    nb, ntemps, steps, warm = 256, 8, 5, 1
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c3", "--walkers", str(nb), "--steps",
           str(steps), "--warmup", str(warm), "--prewarm-seconds", "0", "--no-cpu-baseline", "--also", "", "--dump-state",
           str(dump)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
        assert d["n_gpus"] == 2 and d["config"]["rccl"]["ranks"] == 2 and d["config"]["rccl"]["version"]
        _replay_bench_state(dump, nb, ntemps, 2)
    else:
        assert r.returncode != 0
        # (round 6: the refusal is also ONE JSON line on stdout -- value null, the error, never a rate)
        lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
        assert "needs 2 visible GPUs" in r.stderr and len(lines) == 1
        d = json.loads(lines[0])
        assert d["value"] is None and d["n_gpus"] == 2 and "visible GPUs" in d["error"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="3", RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "must agree" in r.stderr
