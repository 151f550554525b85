"""`python -m rf_inv_amd.run [params.in]` -- the flow of the reference main program
(src/rf_inv.f90:45-110) on the HIP engine: get_params -> read_obs -> seed -> init_forward ->
read_ref_model -> init_model -> init_likelihood -> init_pt_mcmc -> pt_control -> output_results.

One process per GPU.  Multi-GPU: launch with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        -m rf_inv_amd.run params.in
(backend "nccl" = RCCL; RF_INV_BACKEND=gloo keeps the control messages on the host).  Every rank runs
N_CHAINS chains with its own random stream (src/rf_inv.f90:73-74); the temperature exchange spans all
ranks' chains (src/pt_mcmc.f90:498-571); rank 0 merges and writes the result files.
"""
from __future__ import annotations

import os
import sys

from . import RFEngine, get_params, read_obs, read_ref_model
from .mcmc import RJMCMC, EngineEvaluator, TorchComm, rank_seed
from .mcmc_out import output_results, reduce_counters
from .mt19937 import MT19937
from .posterior import Posterior, reduce_results


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    param_file = argv[0] if argv else "params.in"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    comm = device = None
    if world > 1:
        import torch
        import torch.distributed as dist

        backend = os.environ.get("RF_INV_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            device = torch.device("cuda", local_rank)
        dist.init_process_group(backend)
        comm = TorchComm(device=device)
    rank = comm.rank if comm else 0
    verb = rank == 0

    p = get_params(param_file)
    read_obs(p)
    rng = MT19937(rank_seed(p.iseed, rank))
    ref = read_ref_model(os.path.join(p.base_dir, p.vel_file))
    if verb:
        print(f" rf_inv_amd: {world} rank(s) x {p.nchains} chains, {p.nburn}+{p.niter} iterations, "
              f"nfft {p.nfft}, {p.ntrc} trace(s)")
    with RFEngine.from_params(p, device=local_rank) as eng:
        eng.set_model(p, ref)
        m = RJMCMC(p, ref, EngineEvaluator(eng, p.k_max + 2), rng, comm=comm)
        m.init_model()
        m.init_likelihood()
        m.init_pt_mcmc()
        m.posterior = Posterior(eng, p)
        n_tot = p.nburn + p.niter
        for it in range(1, n_tot + 1):
            if verb and it % p.ncorr == 0:
                print(f" Iteration #: {it} / {n_tot}")
            m.iterate(it)
        res = m.posterior.read()
    if comm is not None:
        res = reduce_results(res, device=device)
        cnt = reduce_counters(m.counters, device=device)
    else:
        cnt = m.counters
    if rank == 0:
        out_dir = p.out_dir if os.path.isabs(p.out_dir) else os.path.join(p.base_dir, p.out_dir)
        output_results(p, res, cnt, nproc=world, out_dir=out_dir, verb=True)
    if comm is not None:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
