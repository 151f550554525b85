"""One rank of the posterior-merge test (tests/test_gpu_posterior.py::test_posterior_merge_over_the_communicator): a
fresh process that evaluates and records its own chains (rf_post_record), saves its accumulators, then takes part in
the end-of-run merge over the engine's communicator -- rf_comm_post_gather + rf_comm_post_reduce through
Posterior.merge_over_comm (the top of output_results, src/mcmc_out.f90:52-93) -- and saves what it holds afterwards.

usage: post_merge_worker.py RANK WORLD OUT_DIR DEVICE [RCCL_LIBRARY|- [sequential]]   (sequential: the reduce's twelve
ncclReduce calls one by one instead of as one group, rf_comm_set_option "sequential_reduce")
Writes OUT_DIR/own_RANK.npz (before the merge), merged_RANK.npz (merge_over_comm's return), after_RANK.npz (a plain read
after the merge)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NCHAINS, MAX_MODELS = 40, 60


def save(path, r):
    from dataclasses import fields

    np.savez(path, **{f.name: np.asarray(getattr(r, f.name)) for f in fields(r) if getattr(r, f.name) is not None})


def main():
    rank, world, out, device = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    import test_gpu_posterior as tp
    from helpers import DELTA
    from oracle import rf_oracle as orc
    from rf_inv_amd import RFEngine
    from rf_inv_amd.posterior import Posterior

    orc.build()
    if len(sys.argv) > 5 and sys.argv[5] != "-":
        RFEngine.comm_set_library(sys.argv[5])
    p, ref, mcfg = tp._setup(os.path.join(ROOT, "tests", "golden"), 2.0, 0, 10)
    ntrc, n = 2, NCHAINS
    p.ntrc, p.nsmp = ntrc, 101
    p.sig_mode, p.sig_min, p.sig_max = [1, 0], [0.005, 0.01], [0.08, 0.01]
    p.amp_min, p.amp_max, p.nbin_amp = -0.05, 0.25, 40
    p.nbin_z, p.nbin_vs, p.nbin_vp, p.nbin_vpvs, p.nbin_sig = 37, 25, 20, 15, 11
    p.nchains, p.niter, p.ncorr = n, 2, 1
    rng = np.random.default_rng(500 + rank)
    k, z, dvp, dvs = tp._valid_states(orc, rng, p, mcfg, n)
    sig = np.stack([rng.uniform(0.005, 0.0799, n), np.full(n, 0.01)], axis=1).copy()
    pad = p.k_max + 2
    nlay, lay = tp._layers(orc, mcfg, k, z, dvp, dvs, pad)
    obs = np.random.default_rng(7).normal(0, 0.05, (ntrc, 101))      # the same data on every rank
    ids = np.arange(n, dtype=np.int32)
    eng = RFEngine(nfft=256, delta=DELTA, t_start=0.0, deconv_mode=0, sdep=2.0, rayps=[0.06, 0.075], a_gus=[4.0, 3.0],
                   ipha=[1, 1], obs=obs, nsmp=101, max_walkers=n, nlay_max=pad, device=device)
    logl = eng.eval_batch(ids, nlay, lay, sig)
    eng.commit(ids, np.ones(n, dtype=np.int32))
    eng.set_model(p, ref)
    post = Posterior(eng, p, max_models=MAX_MODELS)
    # ranks record different numbers of models (temperatures move between ranks): a filter that depends on the rank
    temps = rng.permutation(np.where(np.arange(n) < 12 + 7 * rank, 1.0, 3.0))
    if not (world == 3 and rank == 1):                               # (three ranks: the middle one records nothing at all)
        post.record(ids, k, z, dvp, dvs, sig, logl, temps=temps)
        post.record(ids, k, z, dvp, dvs, sig, logl, temps=temps[::-1].copy())
    if rank == world - 1:
        post.record(ids, k, z, dvp, dvs, sig, logl)                  # the last rank overflows its max_models rows
    save(os.path.join(out, f"own_{rank}.npz"), post.read())

    idf = os.path.join(out, "rccl_id")
    if rank == 0:
        with open(idf + ".tmp", "wb") as fh:
            fh.write(RFEngine.comm_unique_id())
        os.replace(idf + ".tmp", idf)
    t0 = time.time()
    while not os.path.exists(idf):
        assert time.time() - t0 < 120, "no RCCL id from rank 0"
        time.sleep(0.05)
    eng.comm_init(open(idf, "rb").read(), rank, world)
    if len(sys.argv) > 6 and sys.argv[6] == "sequential":
        eng.comm_set_option("sequential_reduce", 1)
    save(os.path.join(out, f"merged_{rank}.npz"), post.merge_over_comm(root=0))
    save(os.path.join(out, f"after_{rank}.npz"), post.read())
    eng.comm_destroy()
    eng.close()


if __name__ == "__main__":
    main()
