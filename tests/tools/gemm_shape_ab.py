#!/usr/bin/env python3
"""Long-window likelihood: the two MFMA shapes of phi_gemm_kernel (rf_set_option "gemm_shape" 16 | 4) give the same bits;
their kernel times at bench.py's c4w20 / c4w60, interleaved.   usage: tests/tools/gemm_shape_ab.py [reps]"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def same_bits():
    from helpers import DELTA, make_cfg, pack_layers, random_stack, synth_obs
    from oracle import rf_oracle as orc
    from rf_inv_amd import RFEngine

    orc.build()
    rng = np.random.default_rng(5)
    for nsmp, nb in ((401, 777), (1201, 300)):
        cfg = make_cfg(nfft=4096, rayps=[0.06, 0.09], ipha=[1, -1], t_start=-3.0)
        obs = synth_obs(orc, cfg, random_stack(rng, 5), nsmp)
        r_inv = orc.build_r_inv(nsmp, cfg["a_gus"], DELTA)
        nlay, layers = pack_layers([random_stack(rng, int(n)) for n in rng.integers(3, 20, nb)], 22)
        sig = rng.uniform(0.01, 0.05, (nb, 2))
        out = {}
        for shape in (16, 4):
            with RFEngine(nfft=4096, delta=cfg["delta"], t_start=cfg["t_start"], deconv_mode=0, sdep=0.0, rayps=cfg["rayps"],
                          a_gus=cfg["a_gus"], ipha=cfg["ipha"], obs=obs, nsmp=nsmp, r_inv=r_inv, max_walkers=nb, nlay_max=22,
                          options={"gemm_shape": shape}) as eng:
                out[shape] = eng.eval_batch(np.arange(nb), nlay, layers, sig)
        assert np.array_equal(out[16], out[4]), (nsmp, np.abs(out[16] - out[4]).max())
        print(f"nsmp {nsmp}: the two shapes agree bit for bit on {nb} walkers")


def times(reps):
    for rep in range(reps):
        for wl in ("c4w60", "c4w20"):
            for shape in (16, 4):
                r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--also", "", "--steps", "60",
                                    "--warmup", "10", "--no-cpu-baseline", "--opt", f"gemm_shape={shape}"], capture_output=True, text=True)
                d = json.loads(r.stdout.strip().splitlines()[-1])
                q = d["quadratic_form_gemm"]
                print(f"rep{rep} {wl} shape {shape:2d}: step {d['ms_per_step']:.3f} ms, fused {d['kernel_ms']['fused']:.3f}, GEMM + logL "
                      f"{q['ms']:.3f} ms = {q['achieved']:.1f} TF algorithmic; parity worst rel {d['parity_in_bench']['max_rel_dlogl']:.1e}",
                      flush=True)


if __name__ == "__main__":
    same_bits()
    times(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
