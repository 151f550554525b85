#!/usr/bin/env python3
"""Long-window likelihood: the two block tilings of phi_gemm_kernel (rf_set_option "gemm_tile" 128 | 64; 0 = by launch
size) give the same bits; their kernel times at bench.py's c4w20 / c4w60, interleaved.
usage: tests/tools/gemm_tile_ab.py [reps]"""
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
        for shape in (128, 64):
            with RFEngine(nfft=4096, delta=cfg["delta"], t_start=cfg["t_start"], deconv_mode=0, sdep=0.0, rayps=cfg["rayps"],
                          a_gus=cfg["a_gus"], ipha=cfg["ipha"], obs=obs, nsmp=nsmp, r_inv=r_inv, max_walkers=nb, nlay_max=22,
                          options={"gemm_tile": shape}) as eng:
                out[shape] = eng.eval_batch(np.arange(nb), nlay, layers, sig)
        assert np.array_equal(out[128], out[64]), (nsmp, np.abs(out[128] - out[64]).max())
        print(f"nsmp {nsmp}: the two tilings agree bit for bit on {nb} walkers")


def times(reps):
    for rep in range(reps):
        for wl in ("c4w60", "c4w20"):
            for shape in (128, 64):
                r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--also", "", "--steps", "60",
                                    "--warmup", "10", "--no-cpu-baseline", "--opt", f"gemm_tile={shape}"], capture_output=True, text=True)
                d = json.loads(r.stdout.strip().splitlines()[-1])
                q = d["quadratic_form_gemm"]
                print(f"rep{rep} {wl} tile {shape:3d}: step {d['ms_per_step']:.3f} ms, fused {d['kernel_ms']['fused']:.3f}, GEMM + logL "
                      f"{q['ms']:.3f} ms = {q['achieved']:.1f} TF run; parity worst rel {d['parity_in_bench']['max_rel_dlogl']:.1e}",
                      flush=True)


if __name__ == "__main__":
    same_bits()
    times(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
