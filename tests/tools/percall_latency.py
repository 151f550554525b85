"""Latency of the per-call drop-in (rf_calc_likelihood, one chain per call: src/pt_mcmc.f90:178-180) on the shipped
sample_syn shape and on a C2-shaped context: microseconds per call, with and without the trace copied back.
usage: python tests/tools/percall_latency.py [ncalls] [path of another librfgpu.so build]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import DELTA, random_stack  # noqa: E402
from rf_inv_amd import RFEngine, _lib  # noqa: E402

if len(sys.argv) > 2:
    _lib.load(sys.argv[2])
    print("library:", sys.argv[2])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(1)
for name, nfft, rayps, ipha, sdep, nl in (("sample_syn shape (nfft 256, 2 P traces, ocean, 5 layers)", 256, [0.06, 0.08], [1, 1], 2.0, 5),
                                          ("C2 shape (nfft 4096, 1 P trace, 12 layers)", 4096, [0.06], [1], 0.0, 12)):
    ntrc = len(rayps)
    with RFEngine(nfft=nfft, delta=DELTA, t_start=0.0, deconv_mode=0, sdep=sdep, rayps=np.array(rayps), a_gus=np.full(ntrc, 4.0),
                  ipha=np.array(ipha, dtype=np.int32), obs=np.zeros((ntrc, 101)), nsmp=101, max_walkers=8, nlay_max=32) as eng:
        st = random_stack(rng, nl, sdep > 0, sdep)
        sig = np.full(ntrc, 0.02)
        for want in (True, False):
            for _ in range(200):
                eng.calc_likelihood(0, True, nl, *st, sig, want_rft=want)
            t0 = time.perf_counter()
            for _ in range(n):
                eng.calc_likelihood(0, True, nl, *st, sig, want_rft=want)
            dt = time.perf_counter() - t0
            print(f"{name}: {1e6 * dt / n:7.2f} us per call ({'with' if want else 'without'} the trace read back)")
