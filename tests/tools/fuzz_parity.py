"""Randomised parity sweep (GPU box): random contexts (nfft, traces, phases, ocean, deconvolution, window,
batch size, fwd flags) against the CPU oracle with the tolerances of tests/helpers.py and the conditioning rule of
tests/test_gpu_configs.py (an item may exceed the plain tolerance only when kappa = max|rx| / |maxval(rx)| of the
vertical trace it is normalised by is >= 1000, and then by at most kappa / 1000; kappa from the oracle's own traces).  Not part of the
pytest suites (minutes of oracle time); run by hand:  python tests/tools/fuzz_parity.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DELTA, logl_tol, make_cfg, pack_layers, random_stack, synth_obs  # noqa: E402
from oracle import rf_oracle as oracle  # noqa: E402
from rf_inv_amd import RFEngine  # noqa: E402

KAPPA_MIN, KAPPA_SCALE = 1000.0, 1000.0      # as tests/test_gpu_configs.py


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    oracle.build()
    rng = np.random.default_rng(seed)
    worst, n_allow, n_items = 0.0, 0, 0
    dump = []           # every item that used the conditioning allowance: its context, stack, sigma, both logL, kappa
    for case in range(ncases):
        nfft = int(rng.choice([256, 512, 1024, 2048, 4096]))
        long_series = rng.integers(0, 7) == 0       # now and then: the split plans (direct DFT, four-step, Bluestein)
        if long_series:
            nfft = int(rng.choice([1000, 1500, 3000, 6000, 8192, 16384]))
        ntrc = int(rng.integers(1, 5))
        ocean = bool(rng.integers(0, 2))
        sdep = 2.0 if ocean else 0.0
        deconv = int(rng.integers(0, 2))
        ipha = [int(rng.choice([1, -1])) for _ in range(ntrc)]
        rayps = [float(rng.uniform(0.04, 0.075)) if ph == 1 else float(rng.uniform(0.09, 0.12)) for ph in ipha]
        if rng.integers(0, 4) == 0 and ntrc > 1:      # common rays now and then
            rayps = [rayps[0]] * ntrc
            ipha = [ipha[0]] * ntrc
        a_gus = [float(rng.choice([2.5, 4.0, 6.0])) for _ in range(ntrc)]
        t_start = float(rng.choice([0.0, -1.0, -3.0]))
        nsmp = int(rng.choice([61, 101, 161]))
        kmax = int(rng.choice([6, 15, 30]))
        nb = int(rng.choice([1, 3, 17, 130, 300, 700]))
        if long_series:
            nb = int(rng.choice([1, 3, 17]))           # (the oracle's any-length transform is O(n^2))
        cfg = make_cfg(nfft=nfft, deconv_mode=deconv, t_start=t_start, sdep=sdep, rayps=rayps, a_gus=a_gus, ipha=ipha)
        true = random_stack(rng, int(rng.integers(3, 7)), ocean, sdep)
        obs = synth_obs(oracle, cfg, true, nsmp)
        r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
        lo = 3 if ocean else 2
        stacks = [random_stack(rng, int(rng.integers(lo, kmax + 2)), ocean, sdep) for _ in range(nb)]
        nlay, layers = pack_layers(stacks, kmax + 2)
        sig = rng.uniform(0.01, 0.05, (nb, ntrc))
        ref, kap1 = oracle.eval_batch(cfg, obs, r_inv, nlay, layers, sig, nsmp, nthreads=oracle.max_threads(), want_kappa=True)
        with RFEngine(nfft=nfft, delta=cfg["delta"], t_start=t_start, deconv_mode=deconv, sdep=sdep, rayps=cfg["rayps"],
                      a_gus=cfg["a_gus"], ipha=cfg["ipha"], obs=obs, nsmp=nsmp, r_inv=r_inv, max_walkers=nb,
                      nlay_max=kmax + 2) as eng:
            ll = eng.eval_batch(np.arange(nb), nlay, layers, sig)
            eng.commit(np.arange(nb), np.ones(nb, dtype=np.int32))
            # second evaluation: mixed forward / sigma-only items
            ff = rng.integers(0, 2, nb).astype(np.int32)
            sig2 = sig * rng.uniform(0.8, 1.6)
            stacks2 = [random_stack(rng, int(rng.integers(lo, kmax + 2)), ocean, sdep) for _ in range(nb)]
            nlay2, layers2 = pack_layers(stacks2, kmax + 2)
            ll2 = eng.eval_batch(np.arange(nb), nlay2, layers2, sig2, fwd_flag=ff)
        use_l = np.where(ff[:, None, None] == 1, layers2, layers)
        use_n = np.where(ff == 1, nlay2, nlay)
        ref2, kap2 = oracle.eval_batch(cfg, obs, r_inv, use_n, use_l, sig2, nsmp, nthreads=oracle.max_threads(),
                                       want_kappa=True)
        ok = True
        for got, want, kap in ((ll, ref, kap1), (ll2, ref2, kap2)):
            fin = np.isfinite(want)
            allow = np.where(kap >= KAPPA_MIN, np.maximum(kap / KAPPA_SCALE, 1.0), 1.0)
            over = fin & ~(np.abs(got - want) <= logl_tol(want))
            n_items += int(fin.sum())
            used = over & (np.abs(got - want) <= logl_tol(want) * allow)
            n_allow += int(np.sum(used))
            for i in np.nonzero(used)[0]:
                lay = (layers if got is ll else use_l)[i]
                pad = np.ones((4, 33)); pad[:, :lay.shape[1]] = lay
                obs_pad = np.zeros((4, 161)); obs_pad[:ntrc, :nsmp] = obs
                tr = lambda a, fill: np.concatenate([np.asarray(a, float), np.full(4 - ntrc, fill)])
                dump.append(dict(case=case, item=int(i), nfft=nfft, ntrc=ntrc, deconv=deconv, sdep=sdep, t_start=t_start, nsmp=nsmp,
                                 rayps=tr(rayps, 0.0), a_gus=tr(a_gus, 0.0), ipha=tr(ipha, 0.0), nlay=int((nlay if got is ll else use_n)[i]),
                                 layers=pad, sig=tr((sig if got is ll else sig2)[i], 0.0), obs=obs_pad, logl_hip=float(got[i]),
                                 logl_oracle=float(want[i]), kappa=float(kap[i])))
            bad = (np.isnan(got) != np.isnan(want)) | (fin & ~(np.abs(got - want) <= logl_tol(want) * allow))
            if bad.any():
                ok = False
                i = int(np.argmax(bad))
                nl_i = int((nlay if got is ll else use_n)[i])
                print("  MISMATCH item", i, got[i], want[i], "nlay", nl_i, "kappa %.3g" % kap[i])
                if os.environ.get("FUZZ_TRUTH"):
                    # who is closer to the exactly evaluated formulas?  (80-bit long double, tests/tools/truth_check.py)
                    import types

                    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
                    import truth_check as tc

                    pp = types.SimpleNamespace(nfft=nfft, nsmp=nsmp, delta=cfg["delta"], t_start=t_start, sdep=sdep,
                                               ntrc=ntrc, ipha=cfg["ipha"], rayps=cfg["rayps"])
                    lay_i = (layers if got is ll else use_l)[i]
                    sg_i = (sig if got is ll else sig2)[i]
                    if deconv == 0:
                        t = tc.logl_truth(pp, oracle.init_filter(nfft, cfg["delta"], cfg["a_gus"]), obs, r_inv, nl_i,
                                          lay_i, sg_i, oracle)
                        print("    truth %.17g  |gpu-truth|/|truth| %.2e  |oracle-truth|/|truth| %.2e"
                              % (float(t), float(abs(got[i] - t) / abs(t)), float(abs(want[i] - t) / abs(t))))
            rel = (np.abs(got - want) / np.maximum(np.abs(want), 1.0) / allow)[fin]
            worst = max(worst, float(rel.max()) if rel.size else 0.0)
        print(f"case {case:3d} nfft {nfft} ntrc {ntrc} ipha {ipha} ocean {int(ocean)} decon {deconv} nsmp {nsmp} "
              f"kmax {kmax} nb {nb}: {'ok' if ok else 'FAIL'}", flush=True)
    print("worst relative difference / conditioning allowance over all cases: %.2e; %d of %d items used the allowance"
          % (worst, n_allow, n_items))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if dump and os.path.isdir(out_dir):
        # input of tests/tools/kappa_reference_spread.py (the reference's own spread on exactly these items)
        np.savez(os.path.join(out_dir, f"fuzz_allowance_items_seed{seed}.npz"),
                 **{k: np.array([d[k] for d in dump]) for k in dump[0]})


if __name__ == "__main__":
    main()
