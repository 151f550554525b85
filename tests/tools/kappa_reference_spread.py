"""Is the conditioning (kappa) allowance of tests/helpers.py backed by the reference's OWN behaviour?   (CPU only)

Input: gpurun_out/fuzz_allowance_items_seed<seed>.npz, written by tests/tools/fuzz_parity.py on a GPU box -- every item
of three randomised sweeps whose |logL(HIP) - logL(oracle)| exceeded the plain tolerance max(1e-9, 1e-12 |logL|) and was
accepted under the rule of rounds 3-5, "kappa >= 100 and within tolerance * kappa / 10" (kappa = max|rx| / |maxval(rx)| of the filtered
vertical trace the reference divides by, src/forward.f90:197-202).

For exactly those items this evaluates the trace with
  (a) the reference's own code, built -O0 and built -O2 (oracle/_ref/cpu_o0, cpu_o2: all of the reference's sources
      unmodified, MKL's FFTW3 interface; oracle/Makefile.ref) -- two builds of the SAME program,
  (b) the CPU oracle with its two inverse transforms (the O(n^2) definition in long double and its FFT),
forms logL from each trace set with one and the same quadratic form (the oracle's log_likelihood on the item's observed
traces, R^-1 and sigma: the spread is in the traces, the quadratic form is well conditioned) and reports, per item, in
units of the plain tolerance:  what the HIP path used, the reference's -O0 / -O2 spread, the oracle's two-transform
spread, and the allowance kappa / 10 those rounds granted.  Outcome (round 6): the rule became kappa / 1000
(tests/helpers.py).

    python tests/tools/kappa_reference_spread.py [seeds...]  ->  profiles/r06_kappa_reference_spread.json"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DELTA, logl_tol  # noqa: E402
from oracle import gen_golden, refrun  # noqa: E402
from oracle import rf_oracle as oracle  # noqa: E402


def oracle_rft_two_transforms(cfg, stack):
    """calc_rf's tail (src/forward.f90:166-203) applied to the oracle's spectra with BOTH of the oracle's c2r's."""
    nfft = int(cfg["nfft"])
    nh = nfft // 2 + 1
    _, npre, rff, fv = oracle.calc_rf(cfg, *stack, want_stages=True)
    flt = oracle.init_filter(nfft, cfg["delta"], cfg["a_gus"])
    out = {}
    for naive in (False, True):
        rft = np.empty((len(npre), nfft))
        for t in range(len(npre)):
            cx = np.zeros(nfft, dtype=np.complex128)
            cx[:nh] = rff[t] * flt[t]
            rx = oracle.c2r(cx, nfft, naive=naive)
            i = np.arange(1, nfft + 1)
            if cfg["ipha"][t] == 1:
                j = (nfft - npre[t] + i) % nfft
                j[j == 0] = nfft
                tr = rx[j - 1]
            else:
                j = (nfft + npre[t] - i + 1) % nfft
                j[j == 0] = nfft
                tr = -rx[j - 1]
            cx[:nh] = fv[t] * flt[t]
            rft[t] = tr / oracle.c2r(cx, nfft, naive=naive).max()
        out[naive] = rft
    return out[False], out[True]


def main():
    seeds = [int(a) for a in sys.argv[1:]] or [101, 102, 103]
    oracle.build()
    assert refrun.available("cpu_o0") and refrun.available("cpu_o2"), "make -C oracle -f Makefile.ref first"
    items = []
    for seed in seeds:
        z = np.load(os.path.join(ROOT, "gpurun_out", f"fuzz_allowance_items_seed{seed}.npz"))
        for case in np.unique(z["case"]):
            sel = np.nonzero(z["case"] == case)[0]
            i0 = sel[0]
            ntrc, nfft, nsmp = int(z["ntrc"][i0]), int(z["nfft"][i0]), int(z["nsmp"][i0])
            rayps, a_gus, ipha = z["rayps"][i0][:ntrc], z["a_gus"][i0][:ntrc], z["ipha"][i0][:ntrc].astype(np.int32)
            deconv, sdep, t_start = int(z["deconv"][i0]), float(z["sdep"][i0]), float(z["t_start"][i0])
            t_end = t_start + (nsmp - 1) * 0.05
            p = gen_golden.forward_params(nfft, list(rayps), list(ipha), list(a_gus), deconv, sdep, t_start, t_end=t_end)
            assert p.nsmp == nsmp, (p.nsmp, nsmp)
            stacks = [tuple(z["layers"][i][r, :int(z["nlay"][i])] for r in range(4)) for i in sel]
            ref = {}
            for build in refrun.BUILDS:
                with tempfile.TemporaryDirectory() as work:
                    refrun.write_run_dir(work, p)
                    refrun.write_stacks(os.path.join(work, "stacks.txt"), stacks)
                    ref[build] = refrun.run_forward(build, work, len(stacks), nfft, ntrc)["rft"]
            cfg = dict(nfft=nfft, deconv_mode=deconv, delta=DELTA, t_start=t_start, sdep=sdep, rayps=rayps.astype(float),
                       a_gus=a_gus.astype(float), ipha=ipha)
            r_inv = oracle.build_r_inv(nsmp, cfg["a_gus"], DELTA)
            for n, i in enumerate(sel):
                obs, sig = z["obs"][i][:ntrc, :nsmp], z["sig"][i][:ntrc]
                ll = lambda rft: float(oracle.log_likelihood(np.ascontiguousarray(rft), obs, r_inv, sig, nsmp))
                l0, l2 = ll(ref["cpu_o0"][n]), ll(ref["cpu_o2"][n])
                fft_t, naive_t = oracle_rft_two_transforms(cfg, stacks[n])
                lf, ln = ll(fft_t), ll(naive_t)
                want, hip, kap = float(z["logl_oracle"][i]), float(z["logl_hip"][i]), float(z["kappa"][i])
                tol = float(logl_tol(want))
                assert abs(lf - want) <= 1e-6 * abs(want) + 1e-9, (lf, want)        # the tool's tail == the oracle's own
                items.append({"seed": seed, "case": int(case), "item": int(z["item"][i]), "nfft": nfft, "ntrc": ntrc,
                              "nlay": int(z["nlay"][i]), "kappa": kap, "logl": want, "tolerance": tol,
                              "hip_minus_oracle": abs(hip - want) / tol,
                              "hip_minus_reference_o0": abs(hip - l0) / tol, "oracle_minus_reference_o0": abs(want - l0) / tol,
                              "reference_o2_minus_o0": abs(l2 - l0) / tol, "oracle_fft_minus_definition": abs(lf - ln) / tol,
                              "allowance": kap / 10.0})
    k = lambda name: np.array([it[name] for it in items])
    used = np.maximum(k("hip_minus_oracle"), k("hip_minus_reference_o0"))
    spread = np.maximum(k("reference_o2_minus_o0"), k("oracle_fft_minus_definition"))
    summary = {
        "what": "every item of tests/tools/fuzz_parity.py 120 {101,102,103} that used the kappa allowance, re-evaluated with the "
                "reference's own code (-O0 and -O2 builds, CPU) and the oracle's two inverse transforms; all in units of the plain "
                "tolerance max(1e-9, 1e-12 |logL|)",
        "items": len(items), "kappa_min": float(k("kappa").min()), "kappa_max": float(k("kappa").max()),
        "nlay": sorted(set(int(x) for x in k("nlay"))),
        "max_used_over_allowance": float((used / k("allowance")).max()),
        "max_used_over_kappa": float((used / k("kappa")).max()),
        "reference_o2_vs_o0": {"n_beyond_plain_tolerance": int((k("reference_o2_minus_o0") > 1.0).sum()),
                               "median_over_kappa": float(np.median(k("reference_o2_minus_o0") / k("kappa"))),
                               "max_over_kappa": float((k("reference_o2_minus_o0") / k("kappa")).max())},
        "oracle_fft_vs_definition": {"n_beyond_plain_tolerance": int((k("oracle_fft_minus_definition") > 1.0).sum()),
                                     "median_over_kappa": float(np.median(k("oracle_fft_minus_definition") / k("kappa"))),
                                     "max_over_kappa": float((k("oracle_fft_minus_definition") / k("kappa")).max())},
        "hip_vs_reference_o0": {"max_over_kappa": float((k("hip_minus_reference_o0") / k("kappa")).max()),
                                "n_beyond_plain_tolerance": int((k("hip_minus_reference_o0") > 1.0).sum())},
        "items_where_a_reference_spread_exceeds_what_hip_used": int((spread >= k("hip_minus_oracle")).sum()),
        "n_reference_spread_beyond_plain_tolerance": int((spread > 1.0).sum()),
    }
    out = os.path.join(ROOT, "profiles", "r06_kappa_reference_spread.json")
    with open(out, "w") as fh:
        json.dump({"summary": summary, "items": items}, fh, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
