#!/usr/bin/env python3
"""How fast do proposals travel from the three kinds of host memory rf_eval_models accepts -- pageable (staged through
the context's arena), pinned (rf_host_alloc) and shared + registered (rf_host_alloc_shared) -- at the C4 shape?
Times eval_models_begin (the host call) and begin + wait (the evaluation) for 4096 items.
usage: tests/tools/shared_dma_bench.py [items]"""
import os
import sys
import time
import uuid

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bench
    from rf_inv_amd import RFEngine, read_ref_model
    import ctypes as C

    from rf_inv_amd.engine import _dptr, _iptr, host_alloc, host_alloc_shared
    from rf_inv_amd.likelihood import init_r_inv

    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    w = dict(bench.WORKLOADS["c4"])
    p = bench.make_params(w)
    ref = read_ref_model(os.path.join(ROOT, "tests", "golden", "sample_syn", "model", "sample.velmod"))
    rng = np.random.default_rng(1)
    k = rng.integers(3, p.k_max - 1, nb).astype(np.int32)
    z = np.zeros((nb, p.k_max)); dvp = np.zeros((nb, p.k_max)); dvs = np.zeros((nb, p.k_max))
    for i in range(nb):
        z[i, :k[i]] = np.sort(rng.uniform(1.0, 55.0, k[i]))
        dvs[i, :k[i]] = rng.normal(0, 0.1, k[i])
    sig = np.full((nb, p.ntrc), 0.01)
    ids = np.arange(nb, dtype=np.int32)
    src = dict(ids=ids, k=k, z=z, dvp=dvp, dvs=dvs, sig=sig)
    tag = "/rfgpu_dma_" + uuid.uuid4().hex[:10]
    with RFEngine(nfft=p.nfft, delta=p.delta, t_start=p.t_start, deconv_mode=0, sdep=0.0, rayps=p.rayps, a_gus=p.a_gus,
                  ipha=p.ipha, obs=np.zeros((p.ntrc, p.nsmp)), nsmp=p.nsmp, r_inv=init_r_inv(p.nsmp, p.a_gus, p.delta),
                  max_walkers=2 * nb, nlay_max=p.k_max + 2) as eng:
        eng.set_model(p, ref)
        kinds = {"pageable": src,
                 "pinned": {n: host_alloc(a.shape, a.dtype) for n, a in src.items()},
                 "shared+registered": {n: host_alloc_shared(f"{tag}_{n}", a.shape, a.dtype, create=True, gpu=True) for n, a in src.items()}}
        for name in ("pinned", "shared+registered"):
            for n, a in src.items():
                kinds[name][n][...] = a
        ref_ll = None
        L, out = eng._lib, np.empty(nb)
        for name, a in kinds.items():
            for rep in range(3):
                ll = eng.eval_models(a["ids"], a["k"], a["z"], a["dvp"], a["dvs"], a["sig"])
            ref_ll = ll if ref_ll is None else ref_ll
            assert np.array_equal(ll, ref_ll, equal_nan=True)
            t_b, t_all = [], []
            for rep in range(20):
                tk = C.c_int32(0)
                t0 = time.perf_counter()
                eng._chk(L.rf_eval_models_begin(eng._ctx, nb, _iptr(a["ids"]), None, _iptr(a["k"]), _dptr(a["z"]), p.k_max,
                                                _dptr(a["dvp"]), _dptr(a["dvs"]), _dptr(a["sig"]), 0, C.byref(tk)))
                t1 = time.perf_counter()
                eng._chk(L.rf_eval_wait(eng._ctx, tk, _dptr(out), None))
                t2 = time.perf_counter()
                t_b.append(t1 - t0); t_all.append(t2 - t0)
            print(f"{name:18s}: begin {1e3 * np.median(t_b):.3f} ms, begin + wait {1e3 * np.median(t_all):.3f} ms "
                  f"({nb} items, staged arrays {eng.launch_plan['staged_host_arrays']})", flush=True)


if __name__ == "__main__":
    main()
