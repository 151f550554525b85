"""Batched RJ-MCMC driver: the reference's sampler (src/pt_mcmc.f90 `mcmc`, `judge_mcmc`,
`pt_control`, `init_pt_mcmc`; src/model.f90 `init_model`; src/math.f90 `gauss`; src/prior.f90)
restated so that one iteration is  propose-all -> ONE batched forward+likelihood call ->
accept-all,  while consuming the random stream in exactly the reference's order.

Why the order is preserved: in the reference a chain's step draws
its proposal numbers, evaluates, then `judge_mcmc` draws the acceptance uniform
(src/pt_mcmc.f90:610-615).  That uniform does not depend on the likelihood, so it can be
drawn right after the proposal, before the (batched) evaluation; chain after chain the
stream is identical, and so is the trajectory.

The evaluator is the GPU engine (`EngineEvaluator`); anything with the same two methods
works (the CPU tests plug the oracle in, from tests/ only).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from .model import RefModel, format_model
from .mt19937 import MT19937
from .params import Params

_PI2 = 2.0 * 3.1415926535897931            # src/math.f90:36
_TINY_V1 = float(np.float32(1.0e-16))      # src/math.f90:43 (single-precision literal)
_T1_TOL = 1.0 + float(np.float32(1.0e-6))  # src/pt_mcmc.f90:195 `temp <= 1.d0 + 1.0e-6`
_LN2 = 0.69314718055994529                 # src/prior.f90:60
_EPS = float(np.finfo(np.float64).eps)     # epsilon(1.d0), src/pt_mcmc.f90:612


def gauss(rng: MT19937) -> float:
    """real(8) function gauss() (src/math.f90:33-50): Box-Muller, one value per two draws."""
    v1 = rng.grnd()
    v2 = rng.grnd()
    if v1 == 0.0:
        v1 = _TINY_V1
    return math.sqrt(-2.0 * math.log(v1)) * math.cos(_PI2 * v2)


def laplace(rng: MT19937) -> float:
    """real(8) function laplace() (src/prior.f90:54-121)."""
    u1 = rng.grnd()
    u1p = 2.0 * u1
    if u1p < 1.0:
        i_sign, u1pp = 1, 1.0 - u1p
    else:
        i_sign, u1pp = -1, 2.0 - u1p
    a = 0.0
    while True:
        u1ppp = 2.0 * u1pp
        if u1ppp >= 1.0:
            u1 = u1ppp - 1.0
            break
        a += _LN2
        u1pp = u1ppp
    while True:
        w = _LN2 * u1
        val = i_sign * (a + w)
        k = 1
        while True:
            u2 = rng.grnd()
            if u2 >= w:
                u1 = (u2 - w) / (1.0 - w)
                break
            w = u2
            k += 1
        if k % 2 == 1:
            return val


def log_prior_ratio(x_new, x_old, dev, prior_mode):
    """src/prior.f90:35-50."""
    if prior_mode == 1:
        return -(abs(x_new) - abs(x_old)) / dev
    return -((x_new * x_new) - (x_old * x_old)) / (2.0 * dev * dev)


class EngineEvaluator:
    """Forward+likelihood through the HIP engine (rf_eval_batch / rf_commit)."""

    def __init__(self, engine, nlay_pad):
        self.engine = engine
        self.nlay_pad = nlay_pad

    def eval_batch(self, chains, fwd_flags, stacks, sigs):
        nb = len(chains)
        layers = np.ones((nb, 4, self.nlay_pad))
        nlay = np.full(nb, 2, dtype=np.int32)
        for i, st in enumerate(stacks):
            if st is not None:
                n = len(st[0])
                nlay[i] = n
                for r in range(4):
                    layers[i, r, :n] = st[r]
        return self.engine.eval_batch(np.asarray(chains, dtype=np.int32), nlay, layers, np.asarray(sigs),
                                      np.asarray(fwd_flags, dtype=np.int32))

    def commit(self, chains, accepts):
        self.engine.commit(np.asarray(chains, dtype=np.int32), np.asarray(accepts, dtype=np.int32))


@dataclass
class _Proposal:
    itype: int
    null: bool
    k: int = 0
    z: np.ndarray | None = None
    dvp: np.ndarray | None = None
    dvs: np.ndarray | None = None
    sig: np.ndarray | None = None
    log_prior12: float = 0.0
    fwd_flag: bool = True
    stack: tuple | None = None
    log_r: float = 0.0


@dataclass
class Counters:
    nprop: np.ndarray
    naccept: np.ndarray
    likelihood_hist: np.ndarray
    labels: list = field(default_factory=list)


class TorchComm:
    """The three MPI calls of pt_control (src/pt_mcmc.f90:518,544-556,564-570) over
    torch.distributed: one rank per GPU, backend "nccl" (= RCCL, tensors on `device`) or "gloo"
    (device=None, host tensors)."""

    def __init__(self, device=None, group=None):
        import torch
        import torch.distributed as dist

        self._t, self._d, self.device, self.group = torch, dist, device, group
        self.rank, self.nproc = dist.get_rank(group), dist.get_world_size(group)

    def bcast_ints(self, vals, n):
        t = self._t.tensor(list(vals) if self.rank == 0 else [0] * n, dtype=self._t.int32, device=self.device)
        self._d.broadcast(t, src=0, group=self.group)
        return [int(v) for v in t.cpu()]

    def send(self, vals, dst):
        self._d.send(self._t.tensor(list(vals), dtype=self._t.float64, device=self.device), dst=dst,
                     group=self.group)

    def recv(self, n, src):
        t = self._t.zeros(n, dtype=self._t.float64, device=self.device)
        self._d.recv(t, src=src, group=self.group)
        return [float(v) for v in t.cpu()]


def rank_seed(iseed: int, rank: int) -> int:
    """src/rf_inv.f90:73-74: every rank seeds its own MT19937 stream."""
    return iseed + rank * rank * 10000 + 23 * rank


class RJMCMC:
    """Chain state of one rank + the iteration loop.  Single rank by default; with `comm`
    (TorchComm) the temperature exchange runs over all ranks' chains as in pt_control."""

    def __init__(self, p: Params, ref: RefModel, evaluator, rng: MT19937, comm=None):
        self.p, self.ref, self.ev, self.rng = p, ref, evaluator, rng
        self.comm = comm
        self.rank = comm.rank if comm is not None else 0
        self.nproc = comm.nproc if comm is not None else 1
        n, kmax = p.nchains, p.k_max
        self.k = np.zeros(n, dtype=np.int64)
        self.z = np.zeros((n, max(kmax - 1, 1)))
        self.dvp = np.zeros((n, kmax))
        self.dvs = np.zeros((n, kmax))
        self.sig = np.zeros((n, p.ntrc))
        self.log_likelihood = np.zeros(n)
        self.temps = np.ones(n)
        self.counters: Counters | None = None
        self.posterior = None   # rf_inv_amd.posterior.Posterior: device-side record step when set

    # ---- initialisation, in the order of src/rf_inv.f90:83-91 --------------------------
    def _draw_prior(self):
        p = self.p
        if p.prior_mode == 1:
            dvs = laplace(self.rng) * p.dvs_prior
            dvp = laplace(self.rng) * p.dvp_prior
        else:
            dvs = gauss(self.rng) * p.dvs_prior
            dvp = gauss(self.rng) * p.dvp_prior
        return dvs, dvp

    def init_model(self):
        """subroutine init_model (src/model.f90:41-101): whole-model rejection until valid."""
        p, g = self.p, self.rng
        for c in range(p.nchains):
            while True:
                k = p.k_min + int(g.grnd() * (p.k_max - p.k_min))
                self.k[c] = k
                for i in range(k):
                    self.z[c, i] = p.z_min + g.grnd() * (p.z_max - p.z_min)
                for i in range(k):
                    self.dvs[c, i], self.dvp[c, i] = self._draw_prior()
                self.dvs[c, p.k_max - 1], self.dvp[c, p.k_max - 1] = self._draw_prior()
                if format_model(p, self.ref, k, self.z[c], self.dvp[c], self.dvs[c])[5]:
                    break

    def init_likelihood(self):
        """init_sig + init_rft (src/likelihood.f90:107-163): sigma draws, then the first
        evaluation of every chain (one batch)."""
        p, g = self.p, self.rng
        for c in range(p.nchains):
            for t in range(p.ntrc):
                self.sig[c, t] = p.sig_min[t]
                if p.sig_mode[t] == 1:
                    self.sig[c, t] = p.sig_min[t] + g.grnd() * (p.sig_max[t] - p.sig_min[t])
        stacks = []
        for c in range(p.nchains):
            nl, a, b, r, h, _ = format_model(p, self.ref, self.k[c], self.z[c], self.dvp[c], self.dvs[c])
            stacks.append((a, b, r, h))
        chains = list(range(p.nchains))
        self.log_likelihood[:] = self.ev.eval_batch(chains, [1] * p.nchains, stacks, self.sig)
        self.ev.commit(chains, [1] * p.nchains)

    def init_pt_mcmc(self):
        """subroutine init_pt_mcmc (src/pt_mcmc.f90:296-464): proposal types and temperatures."""
        p = self.p
        self.itype_birth, self.itype_death, self.itype_z, self.itype_dvs = 1, 2, 3, 4
        self.ntype = 4
        self.itype_dvp = -1
        if p.vp_mode == 1:
            self.ntype += 1
            self.itype_dvp = self.ntype
        self.isig_trc = [t for t in range(p.ntrc) if p.sig_mode[t] == 1]
        self.itype_sig = -1
        if self.isig_trc:
            self.ntype += 1
            self.itype_sig = self.ntype
        self.temps[:p.ncool] = 1.0
        for c in range(p.ncool, p.nchains):
            self.temps[c] = math.exp(self.rng.grnd() * math.log(p.t_high))   # :450-452
        labels = {self.itype_birth: "Birth proposal", self.itype_death: "Death proposal",
                  self.itype_z: "Moving interface depth proposal", self.itype_dvs: "Perturbing dVs proposal",
                  self.itype_dvp: "Perturbing dVs proposal",          # sic, src/pt_mcmc.f90:380
                  self.itype_sig: "Perturbing sigma proposal"}
        self.counters = Counters(nprop=np.zeros(self.ntype + 1, dtype=np.int64),
                                 naccept=np.zeros(self.ntype + 1, dtype=np.int64),
                                 likelihood_hist=np.zeros(p.nburn + p.niter + 1),
                                 labels=[labels[i] for i in range(1, self.ntype + 1)])

    # ---- one chain's proposal (src/pt_mcmc.f90:74-169) -------------------------------------
    def _propose(self, c) -> _Proposal:
        p, g = self.p, self.rng
        kmax = p.k_max
        k = int(self.k[c])
        z, dvp, dvs, sig = self.z[c].copy(), self.dvp[c].copy(), self.dvs[c].copy(), self.sig[c].copy()
        lp = 0.0
        null = False
        itype = int(g.grnd() * self.ntype) + 1                                  # :88
        if itype == self.itype_birth:                                            # :91-105
            k += 1
            if k < kmax:
                if p.prior_mode == 1:
                    dvp[k - 1] = laplace(g) * p.dvp_prior
                    dvs[k - 1] = laplace(g) * p.dvs_prior
                else:
                    dvp[k - 1] = gauss(g) * p.dvp_prior
                    dvs[k - 1] = gauss(g) * p.dvs_prior
                z[k - 1] = p.z_min + g.grnd() * (p.z_max - p.z_min)
            else:
                null = True
        elif itype == self.itype_death:                                          # :107-122
            k -= 1
            if k >= p.k_min:
                itarget = int(g.grnd() * (k + 1)) + 1
                for il in range(itarget, k + 1):
                    dvp[il - 1] = self.dvp[c, il]
                    dvs[il - 1] = self.dvs[c, il]
                    z[il - 1] = self.z[c, il] if il < z.size else 0.0
                dvp[k] = 0.0
                dvs[k] = 0.0
                if k < z.size:
                    z[k] = 0.0
            else:
                null = True
        elif itype == self.itype_z:                                              # :124-131
            itarget = int(g.grnd() * k) + 1
            z[itarget - 1] = z[itarget - 1] + gauss(g) * p.dev_z
            if z[itarget - 1] < p.z_min or z[itarget - 1] > p.z_max:
                null = True
        elif itype == self.itype_dvs:                                            # :133-140
            itarget = int(g.grnd() * (k + 1)) + 1
            if itarget == k + 1:
                itarget = kmax
            dvs[itarget - 1] = dvs[itarget - 1] + gauss(g) * p.dev_dvs
            lp = log_prior_ratio(dvs[itarget - 1], self.dvs[c, itarget - 1], p.dvs_prior, p.prior_mode)
        elif itype == self.itype_dvp:                                            # :142-149
            itarget = int(g.grnd() * (k + 1)) + 1
            if itarget == k + 1:
                itarget = kmax
            dvp[itarget - 1] = dvp[itarget - 1] + gauss(g) * p.dev_dvp
            lp = log_prior_ratio(dvp[itarget - 1], self.dvp[c, itarget - 1], p.dvp_prior, p.prior_mode)
        elif itype == self.itype_sig:                                            # :151-158
            itarget = self.isig_trc[int(g.grnd() * len(self.isig_trc))]
            sig[itarget] = sig[itarget] + gauss(g) * p.dev_sig
            if sig[itarget] < p.sig_min[itarget] or sig[itarget] > p.sig_max[itarget]:
                null = True
        stack = None
        if not null:                                                             # :163-169
            nl, a, b, r, h, ok = format_model(p, self.ref, k, z, dvp, dvs)
            if not ok:
                null = True
            else:
                stack = (a, b, r, h)
        prop = _Proposal(itype=itype, null=null, k=k, z=z, dvp=dvp, dvs=dvs, sig=sig, log_prior12=lp,
                         fwd_flag=(itype != self.itype_sig), stack=stack)
        if not null:
            # judge_mcmc's uniform (src/pt_mcmc.f90:610-615), drawn now: it does not depend on logL
            while True:
                r = g.grnd()
                if r >= _EPS:
                    break
            prop.log_r = math.log(r)
        return prop

    # ---- one iteration of pt_control (src/pt_mcmc.f90:488-535, single rank) -----------------
    def iterate(self, it: int):
        p, cnt = self.p, self.counters
        props = [self._propose(c) for c in range(p.nchains)]
        live = [c for c in range(p.nchains) if not props[c].null]
        accepted = [False] * p.nchains
        if live:
            ll = self.ev.eval_batch(live, [int(props[c].fwd_flag) for c in live],
                                    [props[c].stack if props[c].fwd_flag else None for c in live],
                                    np.stack([props[c].sig for c in live]))
            acc = []
            for c, l2 in zip(live, ll):
                pr = props[c]
                del_s = (l2 - self.log_likelihood[c]) / self.temps[c] + pr.log_prior12   # :609
                yn = pr.log_r <= del_s                                                    # :616 (NaN -> reject)
                if yn:                                                                    # :182-191
                    self.log_likelihood[c] = l2
                    self.k[c] = pr.k
                    self.dvp[c], self.dvs[c], self.z[c], self.sig[c] = pr.dvp, pr.dvs, pr.z, pr.sig
                accepted[c] = bool(yn)
                acc.append(int(yn))
            self.ev.commit(live, acc)
        for c in range(p.nchains):                                                        # :195-201
            if self.temps[c] <= _T1_TOL:
                cnt.nprop[props[c].itype] += 1
                if accepted[c]:
                    cnt.naccept[props[c].itype] += 1
                cnt.likelihood_hist[it] += self.log_likelihood[c]
        if self.posterior is not None and it > p.nburn and it % p.ncorr == 0:              # :204-286
            # every chain goes down; the device keeps the non-tempered ones (temps filter) in chain order
            self.posterior.record(np.arange(p.nchains), self.k, self.z[:, :max(p.k_max - 1, 1)], self.dvp,
                                  self.dvs, self.sig, self.log_likelihood, temps=self.temps)
        if p.nchains >= 2:
            self._swap_temperatures()
        return accepted

    def _swap_temperatures(self):
        """One temperature-exchange proposal for the whole ensemble (src/pt_mcmc.f90:498-571):
        rank 0 picks two distinct global chains; temperatures move, states stay."""
        p, g, nc = self.p, self.rng, self.p.nchains
        n_all = nc * self.nproc
        pack = None
        if self.rank == 0:                                                                # :501-515
            i1 = int(g.grnd() * n_all)
            while True:
                i2 = int(g.grnd() * n_all)
                if i2 != i1:
                    break
            pack = [i1 // nc, i2 // nc, i1 % nc, i2 % nc]
        if self.nproc > 1:
            pack = self.comm.bcast_ints(pack, 4)                                          # :518
        rank1, rank2, c1, c2 = pack
        if rank1 == self.rank and rank2 == self.rank:                                     # :525-535
            t1, t2 = self.temps[c1], self.temps[c2]
            del_s = (self.log_likelihood[c2] - self.log_likelihood[c1]) * (1.0 / t1 - 1.0 / t2)   # :586
            if math.log(g.grnd()) <= del_s:
                self.temps[c2], self.temps[c1] = t1, t2
        elif rank1 == self.rank:                                                          # :538-557
            t2, e2 = self.comm.recv(2, rank2)
            t1, e1 = self.temps[c1], self.log_likelihood[c1]
            back = t2
            if math.log(g.grnd()) <= (e2 - e1) * (1.0 / t1 - 1.0 / t2):
                self.temps[c1] = t2
                back = t1
            self.comm.send([back], rank2)
        elif rank2 == self.rank:                                                          # :560-570
            self.comm.send([self.temps[c2], self.log_likelihood[c2]], rank1)
            self.temps[c2] = self.comm.recv(1, rank1)[0]

    def mean_t1_likelihood(self, it: int) -> float:
        """The value written to rslt/likelihood (src/mcmc_out.f90:142), single rank."""
        return self.counters.likelihood_hist[it] / float(self.p.ncool)
