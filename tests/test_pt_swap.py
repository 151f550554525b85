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


@pytest.mark.gpu
def test_rccl_exchange_entry_points_on_one_rank():
    """The C-ABI temperature exchange over RCCL (rfgpu_comm.cpp) on a communicator of ONE rank -- all a one-GPU box
    can form (RCCL refuses two ranks on one device): id, init, broadcast, the grouped send/receive with itself as
    the peer, and the all-gather form against rf_pt_swap_device.  Multi-rank behaviour is by construction only until
    a multi-GPU node runs it."""
    from rf_inv_amd import RFEngine
    from rf_inv_amd.pt import PairSchedule, judge_pt

    nch = 64
    delta = float(np.float32(0.05))
    with RFEngine(nfft=256, delta=delta, t_start=0.0, deconv_mode=0, sdep=0.0, rayps=np.array([0.06]),
                  a_gus=np.array([4.0]), ipha=np.array([1], dtype=np.int32), obs=np.zeros((1, 101)), nsmp=101,
                  max_walkers=nch) as eng:
        ok, key = eng.comm_probe()
        assert ok and key >= 0
        assert eng.comm_info()["nranks"] == 1 and eng.comm_info()["rank"] == 0 and eng.comm_info()["rccl_version"]
        eng.comm_init(RFEngine.comm_unique_id(), 0, 1)
        assert eng.comm_info() == {"rank": 0, "nranks": 1, "rccl_version": eng.comm_info()["rccl_version"]}
        assert list(eng.comm_bcast_i32([7, 11, 13, 17])) == [7, 11, 13, 17]
        with pytest.raises(Exception, match="root out of range"):
            eng.comm_bcast_i32([1, 2], root=1)
        # self-exchange: both "sides" are this rank's chain; the rule is judge_pt with chain 1 = chain 2
        t_new, acc = eng.pt_swap_exchange(0, True, 2.5, -10.0, np.log(0.3))
        assert t_new == 2.5 and acc == judge_pt(2.5, 2.5, -10.0, -10.0, np.log(0.3))
        dev = torch.device("cuda", 0)
        rng = np.random.default_rng(3)
        temps = np.exp(rng.random(nch) * np.log(15.0))
        ll = -100.0 * rng.random(nch)
        sched = PairSchedule(nch, 5, 8)
        d_t = torch.from_numpy(temps.copy()).to(dev)
        d_t2 = d_t.clone()
        d_l = torch.from_numpy(ll).to(dev)
        for _ in range(5):
            pairs, logu = sched.draw()
            d_p, d_u = torch.from_numpy(pairs).to(dev), torch.from_numpy(logu).to(dev)
            eng.pt_swap_allgather_device(d_p, d_u, d_t, d_l)
            eng.pt_swap_device(d_p, d_u, d_t2, d_l)
            for (i1, i2), lu in zip(pairs, logu):
                if judge_pt(temps[i1], temps[i2], ll[i1], ll[i2], lu):
                    temps[i1], temps[i2] = temps[i2], temps[i1]
            torch.cuda.synchronize()
            assert np.array_equal(d_t.cpu().numpy(), temps) and np.array_equal(d_t2.cpu().numpy(), temps)
        eng.comm_destroy()


@pytest.mark.gpu
def test_rccl_two_stream_use_of_one_communicator_on_one_rank():
    """The communicator is used from TWO streams: the host-value exchanges (rf_comm_bcast_i32, rf_pt_swap_exchange) on the
    communicator's own stream, the all-gather swap on the evaluation stream -- a sampler makes the former while a segment's
    evaluation kernels are in flight.  Real RCCL, one rank (all a one-GPU box can form): full-occupancy evaluation launches
    queued on the evaluation stream, the exchanges issued from the host meanwhile, an all-gather swap between two
    evaluations -- every value as if each had run alone.  (Across GPUs this interleaving has never run: DESIGN.md 7.)"""
    import sys

    sys.path.insert(0, ROOT)
    from helpers import make_cfg, pack_layers, random_stack

    from rf_inv_amd import RFEngine
    from rf_inv_amd.pt import PairSchedule, judge_pt

    nb, nfft, nsmp = 2048, 4096, 101
    delta = float(np.float32(0.05))
    rng = np.random.default_rng(8)
    stacks = [random_stack(rng, int(rng.integers(8, 16))) for _ in range(nb)]
    nlay, layers = pack_layers(stacks, 17)
    sig = np.full((nb, 1), 0.02)
    obs = rng.normal(0, 0.05, (1, nsmp))
    dev = torch.device("cuda", 0)
    with RFEngine(nfft=nfft, delta=delta, t_start=0.0, deconv_mode=0, sdep=0.0, rayps=np.array([0.06]), a_gus=np.array([4.0]),
                  ipha=np.array([1], dtype=np.int32), obs=obs, nsmp=nsmp, max_walkers=nb, nlay_max=17) as eng:
        want = eng.eval_batch(np.arange(nb), nlay, layers, sig)                       # alone
        eng.comm_init(RFEngine.comm_unique_id(), 0, 1)
        stream = torch.cuda.Stream(device=dev)
        d_ids = torch.arange(nb, dtype=torch.int32, device=dev)
        d_nlay, d_layers, d_sig = (torch.from_numpy(x).to(dev) for x in (nlay, layers, sig))
        d_l1, d_l2 = (torch.empty(nb, dtype=torch.float64, device=dev) for _ in range(2))
        temps = np.exp(rng.random(nb) * np.log(15.0))
        d_t = torch.from_numpy(temps.copy()).to(dev)
        sched = PairSchedule(nb, 11, 64)
        for rep in range(6):
            pairs, logu = sched.draw()
            d_p, d_u = torch.from_numpy(pairs).to(dev), torch.from_numpy(logu).to(dev)
            with torch.cuda.stream(stream):
                for _ in range(4):                                                     # ~1 ms of kernels in flight
                    eng.eval_batch_device(d_ids, d_nlay, d_layers, d_sig, d_l1, stream=stream)
                eng.pt_swap_allgather_device(d_p, d_u, d_t, d_l1, stream=stream)       # RCCL on the evaluation stream
                eng.eval_batch_device(d_ids, d_nlay, d_layers, d_sig, d_l2, stream=stream)
            # ... and meanwhile, from the host, RCCL on the communicator's own stream
            for j in range(5):
                assert list(eng.comm_bcast_i32([rep, j, 13, 17])) == [rep, j, 13, 17]
                t_new, acc = eng.pt_swap_exchange(0, True, 2.5 + j, -10.0, np.log(0.3))
                assert t_new == 2.5 + j and acc == judge_pt(2.5 + j, 2.5 + j, -10.0, -10.0, np.log(0.3))
            stream.synchronize()
            for (i1, i2), lu in zip(pairs, logu):
                if judge_pt(temps[i1], temps[i2], want[i1], want[i2], lu):
                    temps[i1], temps[i2] = temps[i2], temps[i1]
            assert np.array_equal(d_l1.cpu().numpy(), want) and np.array_equal(d_l2.cpu().numpy(), want), rep
            assert np.array_equal(d_t.cpu().numpy(), temps), rep
        eng.comm_destroy()


@pytest.mark.gpu
def test_gathered_swap_kernel_as_three_ranks_on_one_gpu():
    """rf_pt_swap_gathered_device -- the ONE kernel of the multi-rank swap step -- driven as three ranks in turn on
    the one GPU: every "rank" reads the same gathered snapshot [3 * nchains] (global id = rank * nchains + chain,
    src/pt_mcmc.f90:508-511) and writes only its own temperatures; together they equal the serial replay, pairs that
    straddle ranks included, and the snapshot is left untouched."""
    from rf_inv_amd import RFEngine

    nch, world, K = 40, 3, 25
    delta = float(np.float32(0.05))
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(12)
    with RFEngine(nfft=256, delta=delta, t_start=0.0, deconv_mode=0, sdep=0.0, rayps=np.array([0.06]),
                  a_gus=np.array([4.0]), ipha=np.array([1], dtype=np.int32), obs=np.zeros((1, 101)), nsmp=101,
                  max_walkers=nch) as eng:
        temps = np.exp(rng.random(world * nch) * np.log(15.0))
        sched = PairSchedule(world * nch, 7, K)
        n_cross = 0
        for step in range(6):
            ll = -100.0 * rng.random(world * nch)
            pairs, logu = sched.draw()
            d_p, d_u = torch.from_numpy(pairs).to(dev), torch.from_numpy(logu).to(dev)
            g_t, g_l = torch.from_numpy(temps.copy()).to(dev), torch.from_numpy(ll).to(dev)
            new = []
            acc = torch.zeros(K, dtype=torch.int32, device=dev)
            for r in range(world):
                mine = torch.from_numpy(temps[r * nch:(r + 1) * nch].copy()).to(dev)
                eng.pt_swap_gathered_device(d_p, d_u, g_t, g_l, mine, r, world, accepted=acc)
                torch.cuda.synchronize()
                new.append(mine.cpu().numpy())
            assert np.array_equal(g_t.cpu().numpy(), temps)             # the snapshot is read only
            want = []
            for (i1, i2), lu in zip(pairs, logu):
                yes = judge_pt(temps[i1], temps[i2], ll[i1], ll[i2], lu)
                want.append(int(yes))
                if yes:
                    temps[i1], temps[i2] = temps[i2], temps[i1]
                    n_cross += (i1 // nch) != (i2 // nch)
            assert np.array_equal(np.concatenate(new), temps), step
            assert acc.cpu().numpy().tolist() == want
        assert n_cross > 0


def _run_two_rank_exchange(tmp_path, world, devices, rccl_library=None):
    """Two FRESH processes drive librfgpu's RCCL entry points like the Fortran host does
    (tests/tools/rccl_two_rank_worker.py); the temperature history of the ensemble must equal the single-process
    replay step by step, in the reference's one-pair protocol and in the all-gather form."""
    import subprocess

    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import rccl_two_rank_worker as wk

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "rccl_two_rank_worker.py"), str(r),
                               str(world), str(tmp_path), str(devices[r])] + ([rccl_library] if rccl_library else []),
                              env=env, cwd=ROOT) for r in range(world)]
    for q in procs:
        assert q.wait(timeout=600) == 0
    keys = [open(tmp_path / f"key_{r}").read() for r in range(world)]
    assert len(set(keys)) == len(set(devices))                       # one key per physical GPU
    n_all = world * wk.NCHAINS
    # p2p: rank 0's stream names the pair, then the uniform
    temps = np.concatenate([wk.start_temps(r) for r in range(world)])
    got = [np.load(tmp_path / f"p2p_{r}.npy") for r in range(world)]
    rng = np.random.Generator(np.random.Philox(key=wk.SEED))
    n_cross = 0
    for s in range(wk.STEPS):
        ll = np.concatenate([wk.logl_of(r, s) for r in range(world)])
        i1 = int(rng.random() * n_all)
        while True:
            i2 = int(rng.random() * n_all)
            if i2 != i1:
                break
        lu = float(np.log(max(rng.random(), np.finfo(float).tiny)))
        if judge_pt(temps[i1], temps[i2], ll[i1], ll[i2], lu):
            temps[i1], temps[i2] = temps[i2], temps[i1]
            n_cross += (i1 // wk.NCHAINS) != (i2 // wk.NCHAINS)
        assert np.array_equal(np.concatenate([g[s + 1] for g in got]), temps), ("p2p", s)
    assert n_cross > 0
    # all-gather form
    temps = np.concatenate([wk.start_temps(r) for r in range(world)])
    got = [np.load(tmp_path / f"allgather_{r}.npy") for r in range(world)]
    sched = PairSchedule(n_all, wk.SEED, wk.K)
    n_cross = 0
    for s in range(wk.STEPS):
        ll = np.concatenate([wk.logl_of(r, s) for r in range(world)])
        pairs, logu = sched.draw()
        for (i1, i2), lu in zip(pairs, logu):
            if judge_pt(temps[i1], temps[i2], ll[i1], ll[i2], lu):
                temps[i1], temps[i2] = temps[i2], temps[i1]
                n_cross += (i1 // wk.NCHAINS) != (i2 // wk.NCHAINS)
        assert np.array_equal(np.concatenate([g[s + 1] for g in got]), temps), ("allgather", s)
    assert n_cross > 0


@pytest.mark.gpu
def test_rccl_exchange_between_two_gpus(tmp_path):
    """The cross-rank RCCL traffic for real: one process per GPU bootstraps the communicator from rank 0's id and
    runs both exchange forms.  Needs two GPUs: skipped on a one-GPU box (where the next test runs the same ranks
    over a test double of RCCL)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    _run_two_rank_exchange(tmp_path, 2, [0, 1])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_cross_rank_entry_points_with_several_ranks_on_one_gpu(tmp_path, world):
    """rf_comm_init, rf_comm_bcast_i32, rf_pt_swap_exchange and rf_pt_swap_allgather_device with nranks > 1 on a
    ONE-GPU box: the ranks share device 0 and librfgpu is pointed (rf_comm_set_library) at tests/c/rccl_double.cpp,
    a host-staged stand-in for the twelve nccl* calls -- real RCCL refuses two ranks on one device.  Everything
    above those calls (rank -> walker block mapping, the grouped send/receive and the decision both ranks form from
    it, the gathered layout the swap kernel indexes, the in-place temperature update) is the code a multi-GPU run
    executes."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = str(tmp_path / "librccl_double.so")
    subprocess.run([hipcc, "-shared", "-fPIC", "-O2", "-o", lib, os.path.join(ROOT, "tests", "c", "rccl_double.cpp")],
                   check=True, capture_output=True, timeout=300)
    _run_two_rank_exchange(tmp_path, world, [0] * world, rccl_library=lib)


def test_swap_replay_detects_a_wrong_exchange():
    """The self-check is a check: a run whose temperatures differ from the replay by one transposition fails it."""
    from rf_inv_amd.pt import PairSchedule, init_temps, judge_pt, replay_swap_schedule

    nb, nranks, ntemps, steps, k = 64, 2, 8, 7, 8
    rng = np.random.default_rng(3)
    logl = -rng.uniform(10, 1000, (nranks, nb))
    ref = np.concatenate([init_temps(nb, nb // ntemps, 15.0, np.random.Generator(np.random.Philox(key=1234 + 7919 * (rk + 1))))
                          for rk in range(nranks)])
    sched = PairSchedule(nranks * nb, 1234, k)
    for _ in range(steps):
        pairs, logu = sched.draw()
        for (i1, i2), lu in zip(pairs, logu):
            if judge_pt(ref[i1], ref[i2], logl.reshape(-1)[i1], logl.reshape(-1)[i2], lu):
                ref[i1], ref[i2] = ref[i2], ref[i1]
    good = replay_swap_schedule(ref.reshape(nranks, nb), logl, nb, ntemps, steps, k, 1234, 15.0)
    assert good["ok"] and good["moved"] > 0
    bad = ref.copy()
    i, j = np.argmax(bad), np.argmin(bad + (bad == bad.max()))
    bad[[i, j]] = bad[[j, i]]
    assert not replay_swap_schedule(bad.reshape(nranks, nb), logl, nb, ntemps, steps, k, 1234, 15.0)["ok"]
