"""Posterior accumulators of module pt_mcmc (src/pt_mcmc.f90:394-430) kept on the GPU and the
"record sampled model" step (src/pt_mcmc.f90:204-286) as one batched call (rf_post_record*):
the recorded chains' traces are read where the evaluation left them in HBM.

Array shapes are C-order with the reference's first index LAST, so `.ravel()` is the memory
of the Fortran array: nsig[ntrc, nbin_sig] == nsig(nbin_sig, ntrc), namp[ntrc, nsmp, nbin_amp],
nvpz[nbin_vp, nbin_z], vp_model[max_models, nbin_z] ...
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, fields

import numpy as np

from . import _lib
from .engine import RFEngine, _dptr, _iptr
from .params import Params

_INT_FIELDS = ("nk", "nz", "nsig", "namp", "nvpz", "nvsz", "nvpvsz")
_SUM_FIELDS = ("vp_mean", "vs_mean", "vpvs_mean")
_GATHER_FIELDS = ("vp_model", "vs_model", "all_likelihood")


@dataclass
class PosteriorResult:
    nmod: int
    nk: np.ndarray
    nz: np.ndarray
    nsig: np.ndarray
    namp: np.ndarray
    nvpz: np.ndarray
    nvsz: np.ndarray
    nvpvsz: np.ndarray
    vp_mean: np.ndarray
    vs_mean: np.ndarray
    vpvs_mean: np.ndarray
    vp_model: np.ndarray
    vs_model: np.ndarray
    all_likelihood: np.ndarray
    amp_out_of_range: int = 0
    nmod_rank: np.ndarray | None = None   # merge_over_comm: models recorded by each rank


class Posterior:
    """Device-resident accumulators of one engine context (one rank)."""

    def __init__(self, engine: RFEngine, p: Params, max_models: int | None = None):
        """Needs engine.set_model(p, ref) first (the V-z profile runs format_model)."""
        self.engine, self.p = engine, p
        if max_models is None:
            max_models = int(p.nchains * p.niter / p.ncorr)                      # :407-409
        self.max_models = int(max_models)
        self._smin = np.ascontiguousarray(p.sig_min[:p.ntrc], dtype=np.float64)
        self._smax = np.ascontiguousarray(p.sig_max[:p.ntrc], dtype=np.float64)
        self._smode = np.ascontiguousarray(p.sig_mode[:p.ntrc], dtype=np.int32)
        cfg = _lib.RFPostConfig(int(p.nbin_z), int(p.nbin_vs), int(p.nbin_vp), int(p.nbin_vpvs), int(p.nbin_sig),
                                int(p.nbin_amp), float(p.amp_min), float(p.amp_max), float(p.z_min),
                                _dptr(self._smin), _dptr(self._smax), _iptr(self._smode), self.max_models)
        engine._chk(engine._lib.rf_post_create(engine._ctx, C.byref(cfg)))

    def reset(self):
        self.engine._chk(self.engine._lib.rf_post_reset(self.engine._ctx))

    def record(self, chains, k, z, dvp, dvs, sig, logl, temps=None):
        """Record the chains `chains` (ids into the engine's walkers) in order; k[n], z[n, k_max-1],
        dvp/dvs[n, k_max], sig[n, ntrc], logl[n] are their CURRENT state.  temps[n] (optional) applies
        the reference's temp <= 1 + 1e-6 filter on the device.  Pageable arrays are copied before the call returns;
        arrays from host_alloc (pinned) are read by DMA after it returns: leave them alone until a later call on the
        engine has waited for the device (read(), eval_wait ...)."""
        ids = np.ascontiguousarray(chains, dtype=np.int32)
        n = ids.size
        if n == 0:
            return
        kmax, ntrc = self.p.k_max, self.p.ntrc
        k = np.ascontiguousarray(k, dtype=np.int32)
        z = np.ascontiguousarray(z, dtype=np.float64)
        dvp = np.ascontiguousarray(dvp, dtype=np.float64)
        dvs = np.ascontiguousarray(dvs, dtype=np.float64)
        sig = np.ascontiguousarray(sig, dtype=np.float64)
        logl = np.ascontiguousarray(logl, dtype=np.float64)
        if k.shape != (n,) or z.shape != (n, kmax - 1) or dvp.shape != (n, kmax) or dvs.shape != (n, kmax) \
                or sig.shape != (n, ntrc) or logl.shape != (n,):
            raise ValueError("Posterior.record: array shapes do not match (n, k_max, ntrc)")
        t = None
        if temps is not None:
            t = np.ascontiguousarray(temps, dtype=np.float64)
            if t.shape != (n,):
                raise ValueError("Posterior.record: temps must be [n]")
        e = self.engine
        e._chk(e._lib.rf_post_record(e._ctx, n, _iptr(ids), _iptr(k), _dptr(z), _dptr(dvp), _dptr(dvs), _dptr(sig),
                                     _dptr(logl), _dptr(t) if t is not None else None))

    def record_device(self, walker_ids, k, z, dvp, dvs, sig, logl, temps=None, stream=None):
        """The same with torch CUDA tensors (int32 / float64, contiguous); asynchronous on `stream`."""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(sig.device)
        e = self.engine
        e._chk(e._lib.rf_post_record_device(
            e._ctx, walker_ids.numel(), walker_ids.data_ptr(), k.data_ptr(), z.data_ptr(), dvp.data_ptr(),
            dvs.data_ptr(), sig.data_ptr(), logl.data_ptr(), temps.data_ptr() if temps is not None else None,
            st.cuda_stream))

    def read(self, with_models=True) -> PosteriorResult:
        p, nm = self.p, self.max_models
        nmod = C.c_int32(0)
        oor = C.c_int64(0)
        r = PosteriorResult(
            nmod=0,
            nk=np.zeros(p.k_max, dtype=np.int32), nz=np.zeros(p.nbin_z, dtype=np.int32),
            nsig=np.zeros((p.ntrc, p.nbin_sig), dtype=np.int32),
            namp=np.zeros((p.ntrc, p.nsmp, p.nbin_amp), dtype=np.int32),
            nvpz=np.zeros((p.nbin_vp, p.nbin_z), dtype=np.int32),
            nvsz=np.zeros((p.nbin_vs, p.nbin_z), dtype=np.int32),
            nvpvsz=np.zeros((p.nbin_vpvs, p.nbin_z), dtype=np.int32),
            vp_mean=np.zeros(p.nbin_z), vs_mean=np.zeros(p.nbin_z), vpvs_mean=np.zeros(p.nbin_z),
            vp_model=np.zeros((nm if with_models else 0, p.nbin_z)),
            vs_model=np.zeros((nm if with_models else 0, p.nbin_z)),
            all_likelihood=np.zeros(nm if with_models else 0))
        keep = with_models and nm > 0
        if keep:
            r.vs_model[:, 0] = -999.9        # unused slots as init_pt_mcmc leaves them (src/pt_mcmc.f90:419)
        out = _lib.RFPostResult(
            C.pointer(nmod), _iptr(r.nk), _iptr(r.nz), _iptr(r.nsig), _iptr(r.namp), _iptr(r.nvpz), _iptr(r.nvsz),
            _iptr(r.nvpvsz), _dptr(r.vp_mean), _dptr(r.vs_mean), _dptr(r.vpvs_mean),
            _dptr(r.vp_model) if keep else None, _dptr(r.vs_model) if keep else None,
            _dptr(r.all_likelihood) if keep else None, C.pointer(oor))
        e = self.engine
        e._chk(e._lib.rf_post_read(e._ctx, C.byref(out)))
        r.nmod, r.amp_out_of_range = int(nmod.value), int(oor.value)
        return r


    def merge_over_comm(self, root=0, with_models=True) -> PosteriorResult:
        """The merge at the top of output_results (src/mcmc_out.f90:52-93) over the engine's RCCL communicator
        (engine.comm_init): rf_comm_post_gather (the model rows, rank blocks in rank order) + rf_comm_post_reduce
        (histograms and mean sums, ncclReduce in place into the root's device accumulators).  Collective; once per
        run.  Returns the merged result on `root`, this rank's own (unmerged) result elsewhere."""
        e, p, nm = self.engine, self.p, self.max_models
        info = e.comm_info(version=False)
        rank, nranks = info["rank"], info["nranks"]
        counts = np.zeros(nranks, dtype=np.int32)
        keep = with_models and nm > 0
        if rank == root and keep:
            vp = np.zeros((nranks * nm, p.nbin_z))
            vs = np.zeros((nranks * nm, p.nbin_z))
            vs[:, 0] = -999.9                    # unused slots as init_pt_mcmc leaves them (src/pt_mcmc.f90:419)
            al = np.zeros(nranks * nm)
            e._chk(e._lib.rf_comm_post_gather(e._ctx, int(root), _iptr(counts), _dptr(vp), _dptr(vs), _dptr(al)))
        else:
            e._chk(e._lib.rf_comm_post_gather(e._ctx, int(root), _iptr(counts), None, None, None))
        total = C.c_int32(0)
        e._chk(e._lib.rf_comm_post_reduce(e._ctx, int(root), C.byref(total)))
        r = self.read(with_models=with_models and rank != root)
        if rank == root:
            r.nmod = int(total.value)
            if keep:
                r.vp_model, r.vs_model, r.all_likelihood = vp, vs, al
        r.nmod_rank = counts
        return r


def reduce_results(r: PosteriorResult, group=None, device=None) -> PosteriorResult:
    """The merge at the top of output_results (src/mcmc_out.f90:52-99) over torch.distributed:
    SUM-reduce of the counters / histograms / mean sums to rank 0 and a gather of the per-model
    profiles in rank order.  Returns the merged result on rank 0 (other ranks: their own, unmerged).
    `device`: where the collective buffers live (cuda device for the RCCL backend, None = CPU/gloo)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return r
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def to_t(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device) if device is not None \
            else torch.from_numpy(np.ascontiguousarray(a).copy())

    merged = {f.name: getattr(r, f.name) for f in fields(r)}
    scal = to_t(np.array([r.nmod, r.amp_out_of_range], dtype=np.int64))
    dist.reduce(scal, dst=0, op=dist.ReduceOp.SUM, group=group)
    for name in _INT_FIELDS + _SUM_FIELDS:
        t = to_t(getattr(r, name))
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM, group=group)
        if rank == 0:
            merged[name] = t.cpu().numpy()
    for name in _GATHER_FIELDS:
        t = to_t(getattr(r, name))
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        if rank == 0:
            merged[name] = torch.cat(parts, dim=0).cpu().numpy()
    if rank == 0:
        merged["nmod"], merged["amp_out_of_range"] = int(scal[0]), int(scal[1])
        return PosteriorResult(**merged)
    return r
