"""Parallel-tempering temperature exchange across ranks (rf_inv_amd/pt.py), world size 2 on
gloo/CPU: the reference's p2p protocol (pt_mcmc.f90:498-571) and the batched all_gather form."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from rf_inv_amd.pt import PairSchedule, PTSwap, init_temps, judge_pt  # noqa: E402

NCHAINS, NTEMPS, STEPS, SEED = 6, 3, 40, 99


def _logl(rank, step):
    g = np.random.Generator(np.random.Philox(key=1000 + 17 * rank + step))
    return -50.0 * g.random(NCHAINS)


def _worker(rank, world, port, mode, k, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sw = PTSwap(None, NCHAINS, NTEMPS, "cpu", seed=SEED, t_high=15.0, pairs_per_step=k, mode=mode)
    hist = [sw.temps.numpy().copy()]
    for s in range(STEPS):
        sw.step(torch.from_numpy(_logl(rank, s)))
        hist.append(sw.temps.numpy().copy())
    np.save(os.path.join(out, f"temps_{mode}_{rank}.npy"), np.stack(hist))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode,k", [("p2p", 1), ("allgather", 3)])
def test_two_rank_exchange_matches_serial_replay(tmp_path, mode, k):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, k, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"temps_{mode}_{r}.npy") for r in range(world)]
    # serial replay of the same replicated schedule on the concatenated ensemble
    temps = np.concatenate([init_temps(NCHAINS, max(1, NCHAINS // NTEMPS), 15.0,
                                       np.random.Generator(np.random.Philox(key=SEED + 7919 * (r + 1))))
                            for r in range(world)])
    assert np.array_equal(np.concatenate([g[0] for g in got]), temps)
    sched = PairSchedule(world * NCHAINS, SEED, k)
    n_swaps = 0
    for s in range(STEPS):
        ll = np.concatenate([_logl(r, s) for r in range(world)])
        pairs, logu = sched.draw()
        for (i1, i2), lu in zip(pairs, logu):
            if judge_pt(temps[i1], temps[i2], ll[i1], ll[i2], lu):
                temps[i1], temps[i2] = temps[i2], temps[i1]
                n_swaps += 1
        now = np.concatenate([g[s + 1] for g in got])
        assert np.array_equal(now, temps), (mode, s)
        # temperatures move, states stay: the multiset of temperatures is conserved
        assert np.array_equal(np.sort(now), np.sort(np.concatenate([g[0] for g in got])))
    assert n_swaps > 0


def test_pair_schedule_reference_rule():
    """One pair per iteration = two DISTINCT global ids; id -> (rank, chain) as pt_mcmc.f90:508-511."""
    s = PairSchedule(10, 3, 1)
    for _ in range(200):
        p, lu = s.draw()
        assert p.shape == (1, 2) and p[0, 0] != p[0, 1] and 0 <= p.min() and p.max() < 10 and lu[0] <= 0
    s = PairSchedule(64, 5, 16)
    p, _ = s.draw()
    assert len(set(p.ravel().tolist())) == 32  # disjoint


def test_judge_pt_rule():
    # hotter chain with higher likelihood always swaps (del_s >= 0 >= log u)
    assert judge_pt(1.0, 4.0, -10.0, -5.0, np.log(0.999))
    assert not judge_pt(1.0, 4.0, -5.0, -50.0, np.log(0.5))
