"""Parallel-tempering temperature exchange (reference src/pt_mcmc.f90:498-571).

Walkers shard across ranks in contiguous blocks exactly like the reference's
`global id -> (rank = id / nchains, chain = mod(id, nchains) + 1)` (:508-511);
per-walker state never migrates, only temperatures move (:532-535, :551-554, :570).

Two exchange protocols over torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests):

  * mode="p2p"   -- the reference's protocol, one pair per iteration: the pair is drawn
    by a replicated RNG (replacing rank 0's draw + mpi_bcast, :501-519); if it spans two
    ranks, rank2 sends (T2, L2) to rank1 (:564-567), rank1 judges and returns the
    temperature rank2 must hold (:544-556, :568-570).
  * mode="allgather" -- throughput generalisation: K disjoint pairs per iteration; one
    all_gather of every rank's (T, logL) (16 B / walker), after which every rank applies
    the same decisions locally (device kernel rf_pt_swap_device) and keeps its own slice.
    One collective per iteration whatever K is; xGMI is latency- not bandwidth-bound here.
"""
from __future__ import annotations

import numpy as np


def judge_pt(temp1, temp2, l1, l2, log_u):
    """subroutine judge_pt (src/pt_mcmc.f90:580-595): accept iff log(u) <= del_s."""
    return log_u <= (l2 - l1) * (1.0 / temp1 - 1.0 / temp2)


def init_temps(nchains, ncool, t_high, rng):
    """src/pt_mcmc.f90:444-452: first ncool chains at T = 1, the rest exp(u ln T_high)."""
    t = np.ones(nchains)
    for c in range(ncool, nchains):
        t[c] = np.exp(rng.random() * np.log(t_high))
    return t


class PairSchedule:
    """Replicated pair draws: every rank constructs it with the same seed and obtains the
    same sequence, which replaces the reference's rank-0 draw + mpi_bcast (:501-519).
    pairs_per_step = 1 reproduces the reference (two distinct global ids); more pairs
    are drawn as a random partial matching (disjoint)."""

    def __init__(self, n_all, seed, pairs_per_step=1):
        if n_all < 2:
            raise ValueError("need at least two walkers to swap")
        self.n_all = n_all
        self.k = min(int(pairs_per_step), n_all // 2)
        self.rng = np.random.Generator(np.random.Philox(key=seed))

    def draw(self):
        if self.k == 1:
            i1 = int(self.rng.random() * self.n_all)                 # :502
            while True:
                i2 = int(self.rng.random() * self.n_all)             # :503-506
                if i2 != i1:
                    break
            pairs = np.array([[i1, i2]], dtype=np.int32)
        else:
            perm = self.rng.permutation(self.n_all)[: 2 * self.k]
            pairs = perm.reshape(self.k, 2).astype(np.int32)
        u = self.rng.random(self.k)
        u = np.where(u <= 0.0, np.finfo(np.float64).tiny, u)
        return pairs, np.log(u)


def replay_swap_schedule(temps_final, logl, nchains, ntemps, swap_steps, pairs_per_step, seed, t_high=15.0):
    """Serial replay of the replicated swap schedule of a run whose walkers kept the same logL at every step
    (bench.py: the same models are evaluated each step): start from every rank's initial temperatures (PTSwap's
    draw), apply `swap_steps` steps of `pairs_per_step` disjoint pairs with judge_pt (src/pt_mcmc.f90:580-595) on
    global walker ids (rank * nchains + chain, :508-511), compare with the temperatures the run ended with.
    temps_final / logl: [nranks, nchains].  Returns {"ok", "moved", "cross_rank_swaps"}: the replay equals the run;
    walkers that do not hold their initial temperature; accepted swaps whose two walkers live on different ranks."""
    temps_final, logl = np.asarray(temps_final, dtype=np.float64), np.asarray(logl, dtype=np.float64)
    nranks = temps_final.shape[0]
    ref = np.concatenate([init_temps(nchains, max(1, nchains // max(1, int(ntemps))), t_high,
                                     np.random.Generator(np.random.Philox(key=seed + 7919 * (rk + 1))))
                          for rk in range(nranks)])
    start = ref.copy()
    sched = PairSchedule(nranks * nchains, seed, pairs_per_step)
    ll = logl.reshape(-1)
    cross = 0
    for _ in range(int(swap_steps)):
        pairs, logu = sched.draw()
        i1, i2 = pairs[:, 0], pairs[:, 1]
        t1, t2 = ref[i1], ref[i2]
        yes = judge_pt(t1, t2, ll[i1], ll[i2], logu)              # pairs are disjoint: one vector step
        ref[i1] = np.where(yes, t2, t1)
        ref[i2] = np.where(yes, t1, t2)
        cross += int(np.sum(yes & (i1 // nchains != i2 // nchains)))
    return {"ok": bool(np.array_equal(temps_final.reshape(-1), ref)), "moved": int(np.sum(ref != start)),
            "cross_rank_swaps": cross}


def open_exchange(engine, dist, shared_gpu_ok=False):
    """Decide once, unanimously, how this run's ranks exchange temperatures -- the Python host's
    `open_temperature_exchange` (rf_inv_amd/fortran/pt_mcmc_batched.f90): every rank probes RCCL and names its
    physical GPU (rf_comm_probe); if all can and all GPUs differ, rank 0 draws the RCCL id (rf_comm_get_unique_id),
    the process group that launched the run carries its 128 bytes to everybody, and every rank joins
    (rf_comm_init) -- from then on the swap step is librfgpu's own RCCL group over xGMI
    (rf_pt_swap_allgather_device).  Ranks that share a GPU (functional tests on a one-GPU box) return False and keep
    the process group as the transport.  Collective: every rank must call it.
    shared_gpu_ok: functional tests only -- ranks on one GPU join too (over a test double of RCCL selected with
    RFEngine.comm_set_library; real RCCL refuses two ranks on one device)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    # which GPU every rank drives (loads nothing); RCCL itself -- about a second to load -- only if each has its own
    keys = [None] * world
    dist.all_gather_object(keys, int(engine.comm_device_key()))
    if not shared_gpu_ok and len(set(keys)) != world:
        return False
    usable, key = engine.comm_probe()
    seen = [None] * world
    dist.all_gather_object(seen, (bool(usable), int(key)))
    ok = all(u for u, _ in seen) and (shared_gpu_ok or len({k for _, k in seen}) == world)
    token = [None]
    if ok and rank == 0:
        from .engine import RFEngine, RFGPUError

        try:
            token[0] = RFEngine.comm_unique_id()
        except RFGPUError:
            pass
    dist.broadcast_object_list(token, src=0)
    if token[0] is None:
        return False
    # ncclCommInitRank is collective; if it fails anywhere (it fails everywhere or nowhere in practice) every rank
    # must know, drop what it has and keep the process group as the transport -- a benchmark that dies in its
    # bootstrap measures nothing
    from .engine import RFGPUError

    try:
        engine.comm_init(token[0], rank, world)
        mine = True
    except RFGPUError as e:
        print(f"rank {rank}: librfgpu's RCCL communicator could not be formed ({e}); the launcher's process group "
              "carries the temperature exchange instead", flush=True)
        mine = False
    oks = [None] * world
    dist.all_gather_object(oks, mine)
    if not all(oks):
        if mine:
            engine.comm_destroy()
        return False
    return True


class PTSwap:
    """Temperature state of this rank's walkers + the exchange step."""

    def __init__(self, engine, nchains, ntemps, device, seed=0, t_high=15.0, pairs_per_step=None,
                 mode="allgather", cache_steps=256, rccl=None):
        """rccl: True -- the engine holds an RCCL communicator (open_exchange): the device step is
        rf_pt_swap_allgather_device; False -- gather through torch.distributed, judge with the same kernel
        (rf_pt_swap_gathered_device); None -- whatever the engine reports (rf_comm_info)."""
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.engine = engine
        self.nchains = int(nchains)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.device = torch.device(device)
        self.mode = mode
        if rccl is None:
            rccl = engine is not None and self.world > 1 and engine.comm_info(version=False)["nranks"] == self.world
        self.rccl = bool(rccl) and self.world > 1
        n_all = self.world * self.nchains
        ncool = max(1, self.nchains // max(1, int(ntemps)))
        trng = np.random.Generator(np.random.Philox(key=seed + 7919 * (self.rank + 1)))
        self.temps = torch.from_numpy(init_temps(self.nchains, ncool, t_high, trng)).to(self.device)
        if pairs_per_step is None:
            pairs_per_step = 1 if mode == "p2p" else max(1, n_all // 16)
        self.sched = PairSchedule(n_all, seed, pairs_per_step)
        self.k = self.sched.k
        # pre-drawn schedule (replicated) so the step needs no host->device copy
        self._cache = []
        self._cache_steps = cache_steps
        self._cursor = 0
        if self.device.type == "cuda" and mode != "p2p":
            self._fill_cache()
        # gathered (T, logL) of every rank, rank blocks in rank order = indexed by global walker id
        # (only the transport-by-process-group modes use them; the RCCL step gathers inside librfgpu)
        self._g_t = torch.empty(n_all, dtype=torch.float64, device=self.device)
        self._g_l = torch.empty(n_all, dtype=torch.float64, device=self.device)

    def _fill_cache(self):
        torch = self.torch
        ps, us = zip(*(self.sched.draw() for _ in range(self._cache_steps)))
        self._pairs = torch.from_numpy(np.stack(ps)).to(self.device)       # [S, K, 2] global ids
        self._logu = torch.from_numpy(np.stack(us)).to(self.device)        # [S, K]
        self._cursor = 0

    # -- device, throughput mode ------------------------------------------------
    def step(self, logl, stream=None):
        """One exchange step on device tensors (logl[nchains] float64 on self.device)."""
        if self.device.type != "cuda":
            return self.step_host(logl)
        if self.mode == "p2p":
            return self.step_p2p(logl)
        if self._cursor >= self._cache_steps:
            self._fill_cache()
        pairs = self._pairs[self._cursor]
        logu = self._logu[self._cursor]
        self._cursor += 1
        if self.world == 1:
            self.engine.pt_swap_device(pairs, logu, self.temps, logl, None, stream)
        elif self.rccl:
            # one RCCL group (two all-gathers straight from temps / logl) + one kernel, all inside librfgpu
            self.engine.pt_swap_allgather_device(pairs, logu, self.temps, logl, stream)
        else:
            # ranks share a GPU: the process group gathers, the same kernel judges -- all of it on `stream`, the
            # stream the producer of logl ran on (the gather must not overtake it, the kernel not the gather)
            st = stream if stream is not None else self.torch.cuda.current_stream(self.device)
            with self.torch.cuda.stream(st):
                self._gather_global(self.temps, logl)
            self.engine.pt_swap_gathered_device(pairs, logu, self._g_t, self._g_l, self.temps, self.rank, self.world,
                                                stream=st)

    def _gather_global(self, temps, logl):
        """Every rank's T and logL -> self._g_t / self._g_l, indexed by global id = rank * nchains + chain
        (src/pt_mcmc.f90:508-511): all_gather_into_tensor concatenates the rank blocks in rank order, which IS
        that order -- no repacking."""
        dist = self.dist
        if self.device.type == "cuda" and dist.get_backend() == "gloo":
            # functional mode (several ranks on ONE GPU, where RCCL cannot form a communicator): gloo moves host
            # tensors, so stage through the host
            h_t = self.torch.empty(self._g_t.shape, dtype=self.torch.float64)
            h_l = self.torch.empty(self._g_l.shape, dtype=self.torch.float64)
            self.torch.cuda.current_stream(self.device).synchronize()
            dist.all_gather_into_tensor(h_t, temps.cpu())
            dist.all_gather_into_tensor(h_l, logl.cpu())
            self._g_t.copy_(h_t)
            self._g_l.copy_(h_l)
        else:
            dist.all_gather_into_tensor(self._g_t, temps)
            dist.all_gather_into_tensor(self._g_l, logl)
        return self._g_t, self._g_l

    # -- the reference's p2p protocol, device-agnostic (RCCL send/recv on the GPU box, gloo in
    #    the CPU tests): one pair per iteration ---------------------------------------------
    def step_p2p(self, logl):
        """src/pt_mcmc.f90:498-571 with torch.distributed p2p: the pair comes from the replicated
        schedule (every rank knows its role without a broadcast); if it spans two ranks, rank2
        sends (T2, L2) (tag 2018), rank1 judges and returns the temperature rank2 must now hold
        (tag 1988).  All tensor work stays on logl's device and is asynchronous there."""
        torch, dist = self.torch, self.dist
        pairs, logu = self.sched.draw()
        temps = self.temps
        for (i1, i2), lu in zip(pairs.tolist(), logu.tolist()):
            r1, r2 = i1 // self.nchains, i2 // self.nchains                  # :508-509
            c1, c2 = i1 % self.nchains, i2 % self.nchains                    # :510-511 (0-based)
            if r1 != self.rank and r2 != self.rank:
                continue
            if r1 == self.rank and r2 == self.rank:                          # :525-535
                t1, t2, e1, e2 = temps[c1].clone(), temps[c2].clone(), logl[c1], logl[c2]
                yn = lu <= (e2 - e1) * (1.0 / t1 - 1.0 / t2)                # judge_pt :586-592
                temps[c1] = torch.where(yn, t2, t1)
                temps[c2] = torch.where(yn, t1, t2)
            elif r1 == self.rank:                                            # :542-556
                rp = torch.empty(2, dtype=torch.float64, device=temps.device)
                dist.recv(rp, src=r2, tag=2018)
                t1, t2, e1, e2 = temps[c1].clone(), rp[0], logl[c1], rp[1]
                yn = lu <= (e2 - e1) * (1.0 / t1 - 1.0 / t2)
                temps[c1] = torch.where(yn, t2, t1)
                dist.send(torch.where(yn, t1, t2).reshape(1), dst=r2, tag=1988)
            else:                                                            # :564-570
                dist.send(torch.stack([temps[c2], logl[c2]]), dst=r1, tag=2018)
                back = torch.empty(1, dtype=torch.float64, device=temps.device)
                dist.recv(back, src=r1, tag=1988)
                temps[c2] = back[0]

    # -- host tensors (gloo tests, sequential host drivers) --------------------------------
    def step_host(self, logl):
        """Host-resident variant (temps/logl CPU tensors or numpy)."""
        torch = self.torch
        if self.mode == "p2p":
            return self.step_p2p(torch.as_tensor(logl))
        pairs, logu = self.sched.draw()
        temps = self.temps.numpy()
        ll = logl.numpy() if hasattr(logl, "numpy") else np.asarray(logl)
        if self.world > 1:
            gt, gl = self._gather_global(self.temps, torch.as_tensor(ll))
            g_t, g_l = gt.numpy().copy(), gl.numpy()
        else:
            g_t, g_l = temps.copy(), ll
        for (i1, i2), lu in zip(pairs, logu):
            if judge_pt(g_t[i1], g_t[i2], g_l[i1], g_l[i2], lu):
                g_t[i1], g_t[i2] = g_t[i2], g_t[i1]
        temps[:] = g_t[self.rank * self.nchains:(self.rank + 1) * self.nchains]
