"""One rank of the multi-GPU temperature-exchange test (tests/test_pt_swap.py): a FRESH process per GPU that drives
librfgpu's RCCL entry points exactly like the Fortran host does (rf_inv_amd/fortran/pt_mcmc_batched.f90):

  bootstrap   rank 0 draws the 128-byte id (rf_comm_get_unique_id) and publishes it through a file -- the host's own
              channel, MPI in the Fortran program -- every rank joins with rf_comm_init
  p2p form    one pair per iteration as src/pt_mcmc.f90:498-571: rank 0 draws the pair and broadcasts it
              (rf_comm_bcast_i32); a cross-rank pair is ONE grouped send/receive (rf_pt_swap_exchange)
  allgather   K disjoint pairs per iteration: rf_pt_swap_allgather_device

usage: rccl_two_rank_worker.py RANK WORLD OUT_DIR [DEVICE [RCCL_LIBRARY]]
DEVICE: the GPU of this rank (default: RANK).  RCCL_LIBRARY: rf_comm_set_library -- the one-GPU variant of the test
runs both ranks on device 0 over tests/c/rccl_double.cpp, a host-staged test double (real RCCL refuses two ranks on
one device); everything of librfgpu above the twelve nccl* calls is the code a multi-GPU run executes.
Writes OUT_DIR/p2p_RANK.npy and OUT_DIR/allgather_RANK.npy: the temperature history [steps + 1, nchains]."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

NCHAINS, NTEMPS, STEPS, SEED, K = 48, 4, 60, 99, 12


def logl_of(rank, step):
    g = np.random.Generator(np.random.Philox(key=1000 + 17 * rank + step))
    return -50.0 * g.random(NCHAINS)


def start_temps(rank):
    from rf_inv_amd.pt import init_temps

    return init_temps(NCHAINS, max(1, NCHAINS // NTEMPS), 15.0, np.random.Generator(np.random.Philox(key=SEED + 7919 * (rank + 1))))


def main():
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    device = int(sys.argv[4]) if len(sys.argv) > 4 else rank
    import torch

    from rf_inv_amd import RFEngine
    from rf_inv_amd.pt import PairSchedule

    if len(sys.argv) > 5:
        RFEngine.comm_set_library(sys.argv[5])
    delta = float(np.float32(0.05))
    eng = RFEngine(nfft=256, delta=delta, t_start=0.0, deconv_mode=0, sdep=0.0, rayps=np.array([0.06]),
                   a_gus=np.array([4.0]), ipha=np.array([1], dtype=np.int32), obs=np.zeros((1, 101)), nsmp=101,
                   max_walkers=NCHAINS, device=device)
    ok, key = eng.comm_probe()
    assert ok
    open(os.path.join(out, f"key_{rank}"), "w").write(str(key))
    idf = os.path.join(out, "rccl_id")
    if rank == 0:
        tok = RFEngine.comm_unique_id()
        with open(idf + ".tmp", "wb") as fh:
            fh.write(tok)
        os.replace(idf + ".tmp", idf)
    t0 = time.time()
    while not os.path.exists(idf):
        assert time.time() - t0 < 120, "no RCCL id from rank 0"
        time.sleep(0.05)
    eng.comm_init(open(idf, "rb").read(), rank, world)
    info = eng.comm_info()
    assert info["rank"] == rank and info["nranks"] == world and info["rccl_version"]

    # ---- p2p form: the reference's one pair per iteration --------------------------------------------------
    temps = start_temps(rank)
    rng = np.random.Generator(np.random.Philox(key=SEED))      # rank 0's stream: the pair, then (first walker's rank) the uniform
    hist = [temps.copy()]
    n_all = world * NCHAINS
    for s in range(STEPS):
        ll = logl_of(rank, s)
        # every rank advances a replica of the stream so that the uniform is the one a serial replay draws; only
        # rank 0's pair is USED -- it travels by rf_comm_bcast_i32 like mpi_bcast in :518
        i1 = int(rng.random() * n_all)
        while True:
            i2 = int(rng.random() * n_all)
            if i2 != i1:
                break
        logu = float(np.log(max(rng.random(), np.finfo(float).tiny)))
        pick = eng.comm_bcast_i32([i1, i2] if rank == 0 else [-1, -1], root=0)
        assert list(pick) == [i1, i2]
        own = [int(pick[0]) // NCHAINS, int(pick[1]) // NCHAINS]
        slot = [int(pick[0]) % NCHAINS, int(pick[1]) % NCHAINS]
        if rank in own:
            if own[0] == own[1]:
                if logu <= (ll[slot[1]] - ll[slot[0]]) * (1.0 / temps[slot[0]] - 1.0 / temps[slot[1]]):
                    temps[slot[0]], temps[slot[1]] = temps[slot[1]], temps[slot[0]]
            else:
                mine = 0 if own[0] == rank else 1
                t_new, _ = eng.pt_swap_exchange(own[1 - mine], mine == 0, temps[slot[mine]], ll[slot[mine]],
                                                logu if mine == 0 else 0.0)
                temps[slot[mine]] = t_new
        hist.append(temps.copy())
    np.save(os.path.join(out, f"p2p_{rank}.npy"), np.stack(hist))

    # ---- all-gather form: K disjoint pairs per iteration, everything on the device --------------------------
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    d_t = torch.from_numpy(start_temps(rank)).to(dev)
    sched = PairSchedule(n_all, SEED, K)
    hist = [d_t.cpu().numpy().copy()]
    stream = torch.cuda.Stream(device=dev)
    for s in range(STEPS):
        pairs, logu = sched.draw()
        d_l = torch.from_numpy(logl_of(rank, s)).to(dev)
        d_p, d_u = torch.from_numpy(pairs).to(dev), torch.from_numpy(logu).to(dev)
        torch.cuda.synchronize(dev)
        eng.pt_swap_allgather_device(d_p, d_u, d_t, d_l, stream=stream)   # a caller stream, like bench.py
        stream.synchronize()
        hist.append(d_t.cpu().numpy().copy())
    np.save(os.path.join(out, f"allgather_{rank}.npy"), np.stack(hist))
    eng.comm_destroy()
    eng.close()


if __name__ == "__main__":
    main()
