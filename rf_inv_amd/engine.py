"""RFEngine: one librfgpu context (one GPU), the object both interface mirrors
(`Forward`, `Likelihood`) and the batched drivers talk to.  Thin: every method is
one C-ABI call (include/rfgpu.h).  torch is used only for device buffers/streams
in the *_device entry points.
"""
from __future__ import annotations

import os

import ctypes as C

import numpy as np

from . import _lib
from .params import NLAY_MAX, Params


class RFGPUError(RuntimeError):
    pass


def _dptr(a):
    return a.ctypes.data_as(_lib.dp)


def _iptr(a):
    return a.ctypes.data_as(_lib.ip)


class RFEngine:
    def __init__(self, *, nfft, delta, t_start, deconv_mode, sdep, rayps, a_gus, ipha, obs, nsmp,
                 r_inv=None, max_walkers=1, nlay_max=NLAY_MAX, device=0, options=None):
        """obs[ntrc, >= nsmp] (row t = trace t, the reference's obs(:, t));
        r_inv[ntrc, nsmp, nsmp] with r_inv[t].ravel() == Fortran r_inv(:, :, t) or None;
        options: {name: value} launch-plan options (rf_set_option, include/rfgpu.h)."""
        self._lib = _lib.load()
        self.nfft, self.nsmp = int(nfft), int(nsmp)
        self.nh = self.nfft // 2 + 1
        self._rayps = np.ascontiguousarray(rayps, dtype=np.float64)
        self._a_gus = np.ascontiguousarray(a_gus, dtype=np.float64)
        self._ipha = np.ascontiguousarray(ipha, dtype=np.int32)
        self.ntrc = int(self._rayps.size)
        self._obs = np.ascontiguousarray(obs, dtype=np.float64)
        if self._obs.ndim != 2 or self._obs.shape[0] != self.ntrc or self._obs.shape[1] < self.nsmp:
            raise ValueError("obs must be [ntrc, >= nsmp]")
        self._r_inv = None if r_inv is None else np.ascontiguousarray(r_inv, dtype=np.float64)
        if self._r_inv is not None and self._r_inv.shape != (self.ntrc, self.nsmp, self.nsmp):
            raise ValueError("r_inv must be [ntrc, nsmp, nsmp]")
        self.max_walkers, self.nlay_max, self.device = int(max_walkers), int(nlay_max), int(device)
        cfg = _lib.RFConfig(self.nfft, self.ntrc, self.nsmp, int(deconv_mode), float(delta), float(t_start),
                            float(sdep), _dptr(self._rayps), _dptr(self._a_gus), _iptr(self._ipha),
                            _dptr(self._obs), int(self._obs.shape[1]),
                            _dptr(self._r_inv) if self._r_inv is not None else None,
                            self.max_walkers, self.nlay_max, self.device)
        self._ctx = C.c_void_p()
        self._chk(self._lib.rf_ctx_create(C.byref(cfg), C.byref(self._ctx)))
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def set_option(self, name, value):
        """rf_set_option: a launch-plan knob (results do not depend on it, except the opt-in bin_cutoff)."""
        self._chk(self._lib.rf_set_option(self._ctx, str(name).encode(), float(value)))

    @classmethod
    def from_params(cls, p: Params, r_inv=None, max_walkers=None, nlay_max=None, device=0):
        return cls(nfft=p.nfft, delta=p.delta, t_start=p.t_start, deconv_mode=p.deconv_mode, sdep=p.sdep,
                   rayps=p.rayps, a_gus=p.a_gus, ipha=p.ipha, obs=p.obs, nsmp=p.nsmp, r_inv=r_inv,
                   max_walkers=max_walkers or p.nchains,
                   nlay_max=nlay_max or (p.k_max + 2), device=device)

    # ------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise RFGPUError(self._lib.rf_last_error().decode())

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.rf_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- tables --------------------------------------------------------
    @property
    def flt(self):
        """flt(nh, ntrc) of module forward, Fortran-ordered (src/forward.f90:30)."""
        out = np.empty((self.ntrc, self.nh))
        self._chk(self._lib.rf_get_flt(self._ctx, _dptr(out)))
        return out.T

    @property
    def is_ray_common(self):
        f = C.c_int32()
        self._chk(self._lib.rf_get_is_ray_common(self._ctx, C.byref(f)))
        return bool(f.value)

    @property
    def r_inv(self):
        out = np.empty((self.ntrc, self.nsmp, self.nsmp))
        self._chk(self._lib.rf_get_r_inv(self._ctx, _dptr(out)))
        return out

    @property
    def r_inv_info(self):
        """(rank[ntrc], cut_gap[ntrc]) of the library-built pseudo-inverse (-1 / NaN for a supplied r_inv)."""
        rank = np.zeros(self.ntrc, dtype=np.int32)
        gap = np.zeros(self.ntrc)
        self._chk(self._lib.rf_get_r_inv_info(self._ctx, _iptr(rank), _dptr(gap)))
        return rank, gap

    # ---- single evaluations ---------------------------------------------
    def calc_rf(self, nlay, alpha, beta, rho, h):
        """rft(nfft, ntrc) -- src/forward.f90:123-208."""
        a, b, r, hh = (np.ascontiguousarray(x, dtype=np.float64) for x in (alpha, beta, rho, h))
        out = np.empty((self.ntrc, self.nfft))
        self._chk(self._lib.rf_calc_rf(self._ctx, int(nlay), _dptr(a), _dptr(b), _dptr(r), _dptr(hh), _dptr(out)))
        return out.T

    def calc_likelihood(self, walker, fwd_flag, nlay, alpha, beta, rho, h, sig, want_rft=True):
        """(prop_log_likelihood, prop_rft(nfft, ntrc) or None) -- src/likelihood.f90:56-101
        after format_model."""
        a, b, r, hh = (np.ascontiguousarray(x, dtype=np.float64) for x in (alpha, beta, rho, h))
        s = np.ascontiguousarray(sig, dtype=np.float64)
        ll = C.c_double()
        out = np.empty((self.ntrc, self.nfft)) if want_rft else None
        self._chk(self._lib.rf_calc_likelihood(self._ctx, int(walker), int(bool(fwd_flag)), int(nlay), _dptr(a),
                                               _dptr(b), _dptr(r), _dptr(hh), _dptr(s), C.byref(ll),
                                               _dptr(out) if want_rft else None))
        return ll.value, (out.T if want_rft else None)

    def set_r_inv(self, r_inv):
        r = np.ascontiguousarray(r_inv, dtype=np.float64)
        if r.shape != (self.ntrc, self.nsmp, self.nsmp):
            raise ValueError("r_inv must be [ntrc, nsmp, nsmp]")
        self._chk(self._lib.rf_set_r_inv(self._ctx, _dptr(r)))

    def calc_likelihood_of_trace(self, rft, sig):
        """logL of a host-owned trace rft(nfft, ntrc) -- src/likelihood.f90:81-98."""
        r = np.ascontiguousarray(np.asarray(rft, dtype=np.float64).T)
        s = np.ascontiguousarray(sig, dtype=np.float64)
        if r.shape != (self.ntrc, self.nfft):
            raise ValueError("rft must be (nfft, ntrc)")
        ll = C.c_double()
        self._chk(self._lib.rf_calc_likelihood_of_trace(self._ctx, _dptr(r), _dptr(s), C.byref(ll)))
        return ll.value

    # ---- batched -----------------------------------------------------------
    def eval_batch(self, walker_ids, nlay, layers, sig, fwd_flag=None):
        """layers[nb, 4, nlay_pad] (alpha, beta, rho, h), sig[nb, ntrc] -> logL[nb]."""
        ids = np.ascontiguousarray(walker_ids, dtype=np.int32)
        nl = np.ascontiguousarray(nlay, dtype=np.int32)
        L = np.ascontiguousarray(layers, dtype=np.float64)
        s = np.ascontiguousarray(sig, dtype=np.float64)
        ff = None if fwd_flag is None else np.ascontiguousarray(fwd_flag, dtype=np.int32)
        nb = ids.size
        if L.shape[:2] != (nb, 4) or s.shape != (nb, self.ntrc) or nl.size != nb:
            raise ValueError("bad batch shapes")
        out = np.empty(nb)
        self._chk(self._lib.rf_eval_batch(self._ctx, nb, _iptr(ids), _iptr(ff) if ff is not None else None,
                                          _iptr(nl), int(L.shape[2]), _dptr(L), _dptr(s), _dptr(out)))
        return out

    def eval_batch_device(self, walker_ids, nlay, layers, sig, logl, fwd_flag=None, stream=None):
        """Same with torch CUDA tensors (int32 / float64, contiguous); asynchronous on
        `stream` (default: torch's current stream)."""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(layers.device)
        nb = walker_ids.numel()
        assert layers.dtype == torch.float64 and layers.is_contiguous() and layers.shape[:2] == (nb, 4)
        assert sig.dtype == torch.float64 and sig.is_contiguous() and logl.dtype == torch.float64
        assert walker_ids.dtype == torch.int32 and nlay.dtype == torch.int32
        self._chk(self._lib.rf_eval_batch_device(
            self._ctx, nb, walker_ids.data_ptr(), fwd_flag.data_ptr() if fwd_flag is not None else None,
            nlay.data_ptr(), int(layers.shape[2]), layers.data_ptr(), sig.data_ptr(), logl.data_ptr(),
            st.cuda_stream))

    def commit(self, walker_ids, accept):
        ids = np.ascontiguousarray(walker_ids, dtype=np.int32)
        acc = np.ascontiguousarray(accept, dtype=np.int32)
        self._chk(self._lib.rf_commit(self._ctx, ids.size, _iptr(ids), _iptr(acc)))

    def commit_device(self, walker_ids, accept, stream=None):
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(walker_ids.device)
        self._chk(self._lib.rf_commit_device(self._ctx, walker_ids.numel(), walker_ids.data_ptr(),
                                             accept.data_ptr(), st.cuda_stream))

    def get_rft(self, walker, which=0, n=None):
        """rft(1:n, 1:ntrc, walker) (which=0) or the last proposal (which=1)."""
        n = self.nfft if n is None else int(n)
        out = np.empty((self.ntrc, n))
        self._chk(self._lib.rf_get_rft(self._ctx, int(walker), int(which), n, _dptr(out)))
        return out.T

    def get_rft_batch(self, walker_ids, which=0, n=None):
        """rft(1:n, 1:ntrc) of many walkers in one gather: returns [len(ids), ntrc, n]."""
        ids = np.ascontiguousarray(walker_ids, dtype=np.int32)
        n = self.nfft if n is None else int(n)
        out = np.empty((ids.size, self.ntrc, n))
        self._chk(self._lib.rf_get_rft_batch(self._ctx, ids.size, _iptr(ids), int(which), n, _dptr(out)))
        return out

    # ---- format_model on the device -----------------------------------------------
    def set_model(self, p: Params, ref):
        """Hands format_model's inputs (params limits + reference velocity table) to the engine."""
        vp = np.ascontiguousarray(ref.vp_ref, dtype=np.float64)
        vs = np.ascontiguousarray(ref.vs_ref, dtype=np.float64)
        m = _lib.RFModelConfig(int(p.k_max), int(p.vp_mode), int(vp.size), float(p.z_max), float(p.h_min),
                               float(ref.z_ref_min), float(ref.dz_ref), float(p.vp_min), float(p.vp_max),
                               float(p.vs_min), float(p.vs_max), float(p.vpvs_min), float(p.vpvs_max),
                               _dptr(vp), _dptr(vs))
        self._chk(self._lib.rf_set_model(self._ctx, C.byref(m)))
        self.k_max = int(p.k_max)

    def format_models_device(self, k, z, dvp, dvs, nlay, layers, valid=None, stream=None):
        """format_model for a batch, torch CUDA tensors: k[nb] int32, z[nb, k_max-1], dvp/dvs[nb, k_max]
        -> nlay[nb] int32, layers[nb, 4, nlay_pad], valid[nb] int32."""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(layers.device)
        self._chk(self._lib.rf_format_models_device(
            self._ctx, k.numel(), k.data_ptr(), z.data_ptr(), dvp.data_ptr(), dvs.data_ptr(), nlay.data_ptr(),
            layers.data_ptr(), int(layers.shape[2]), valid.data_ptr() if valid is not None else None,
            st.cuda_stream))

    def eval_models_device(self, walker_ids, k, z, dvp, dvs, sig, logl, valid=None, fwd_flag=None, stream=None):
        """format_model + forward + likelihood for proposals given as (k, z, dVp, dVs); invalid models
        are not evaluated (valid = 0, logL = NaN)."""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(sig.device)
        self._chk(self._lib.rf_eval_models_device(
            self._ctx, walker_ids.numel(), walker_ids.data_ptr(),
            fwd_flag.data_ptr() if fwd_flag is not None else None, k.data_ptr(), z.data_ptr(), dvp.data_ptr(),
            dvs.data_ptr(), sig.data_ptr(), logl.data_ptr(), valid.data_ptr() if valid is not None else None,
            st.cuda_stream))

    def eval_models(self, walker_ids, k, z, dvp, dvs, sig, fwd_flag=None, want_valid=False):
        """rf_eval_models: format_model + forward + likelihood from HOST arrays in the batched sampler's layout --
        k[nb] int32, z[nb, k_max - 1 or k_max], dvp / dvs[nb, k_max], sig[nb, ntrc] (row i = chain i, i.e. the
        memory of Fortran's z(ldz, nb) ...).  Arrays from host_alloc() go down by DMA as they are.
        Returns logL[nb] (and valid[nb])."""
        ids = np.ascontiguousarray(walker_ids, dtype=np.int32)
        nb = ids.size
        kk = np.ascontiguousarray(k, dtype=np.int32)
        zz, vp, vs = (np.ascontiguousarray(a, dtype=np.float64) for a in (z, dvp, dvs))
        s = np.ascontiguousarray(sig, dtype=np.float64)
        ff = None if fwd_flag is None else np.ascontiguousarray(fwd_flag, dtype=np.int32)
        if zz.shape[0] != nb or vp.shape != vs.shape or vp.shape[0] != nb or s.shape != (nb, self.ntrc):
            raise ValueError("bad batch shapes")
        out = np.empty(nb)
        valid = np.empty(nb, dtype=np.int32) if want_valid else None
        self._chk(self._lib.rf_eval_models(self._ctx, nb, _iptr(ids), _iptr(ff) if ff is not None else None, _iptr(kk),
                                           _dptr(zz), int(zz.shape[1]), _dptr(vp), _dptr(vs), _dptr(s), _dptr(out),
                                           _iptr(valid) if want_valid else None))
        return (out, valid) if want_valid else out

    def pt_swap_device(self, pairs, log_u, temps, logl, accepted=None, stream=None):
        """judge_pt over pairs[npairs, 2] (torch int32), applied in order."""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(temps.device)
        self._chk(self._lib.rf_pt_swap_device(self._ctx, pairs.shape[0], pairs.data_ptr(), log_u.data_ptr(),
                                              temps.data_ptr(), logl.data_ptr(),
                                              accepted.data_ptr() if accepted is not None else None,
                                              st.cuda_stream))

    # ---- temperature exchange over RCCL (rfgpu_comm.cpp; one process per GPU) -----------------
    @staticmethod
    def comm_unique_id():
        """rank 0: the 128-byte RCCL id every rank passes to comm_init (distribute it with whatever the
        host already has: torch.distributed, MPI, a file)."""
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        if lib.rf_comm_get_unique_id(buf):
            raise RFGPUError(lib.rf_last_error().decode())
        return buf.raw

    @staticmethod
    def comm_set_library(path: str):
        """Load RCCL from this file instead of the default search (process-wide; before any other comm_* call)."""
        lib = _lib.load()
        if lib.rf_comm_set_library(os.fsencode(path)):
            raise RFGPUError(lib.rf_last_error().decode())

    def comm_device_key(self):
        """The physical GPU this context drives (machine + PCI address); loads nothing."""
        key = C.c_int64()
        self._chk(self._lib.rf_comm_device_key(self._ctx, C.byref(key)))
        return key.value

    def comm_probe(self):
        """(usable, device_key): can this rank join an RCCL communicator, and on which physical GPU it sits."""
        key = C.c_int64()
        rc = self._lib.rf_comm_probe(self._ctx, C.byref(key))
        return rc == 0, key.value

    def comm_init(self, unique_id: bytes, rank: int, nranks: int):
        self._chk(self._lib.rf_comm_init(self._ctx, unique_id, int(rank), int(nranks)))

    def comm_info(self, version=True):
        """{rank, nranks} of the context's communicator (0 of 1 without one) and the RCCL version in use
        (version=False: not asked for -- asking loads librccl)."""
        r, n, v = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self._lib.rf_comm_info(self._ctx, C.byref(r), C.byref(n), C.byref(v) if version else None))
        ver = v.value
        return {"rank": r.value, "nranks": n.value,
                "rccl_version": f"{ver // 10000}.{ver // 100 % 100}.{ver % 100}" if ver else None}

    def comm_set_option(self, name: str, value: float):
        """Options of the context's communicator (rf_comm_set_option): "sequential_reduce" 0 | 1."""
        self._chk(self._lib.rf_comm_set_option(self._ctx, name.encode(), float(value)))

    def comm_destroy(self):
        self._chk(self._lib.rf_comm_destroy(self._ctx))

    def comm_bcast_i32(self, values, root=0):
        v = np.ascontiguousarray(values, dtype=np.int32)
        self._chk(self._lib.rf_comm_bcast_i32(self._ctx, _iptr(v), v.size, int(root)))
        return v

    def pt_swap_exchange(self, peer, judge, temp, logl, log_u):
        """The cross-rank swap of src/pt_mcmc.f90:542-571 as one grouped ncclSend + ncclRecv: returns
        (temperature this rank's chain holds afterwards, accepted)."""
        t = C.c_double()
        acc = C.c_int32()
        self._chk(self._lib.rf_pt_swap_exchange(self._ctx, int(peer), int(bool(judge)), float(temp), float(logl),
                                                float(log_u), C.byref(t), C.byref(acc)))
        return t.value, bool(acc.value)

    def pt_swap_allgather_device(self, pairs, log_u, temps, logl, stream=None):
        """K disjoint pairs of GLOBAL walker ids: one ncclAllGather of (T, logL) + the swap kernel; temps[nchains]
        (this rank's, torch float64 on the device) is updated in place."""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(temps.device)
        self._chk(self._lib.rf_pt_swap_allgather_device(self._ctx, temps.numel(), pairs.shape[0], pairs.data_ptr(),
                                                        log_u.data_ptr(), temps.data_ptr(), logl.data_ptr(),
                                                        st.cuda_stream))

    def pt_swap_gathered_device(self, pairs, log_u, g_temps, g_logl, temps, rank, nranks, accepted=None, stream=None):
        """The same judgement on arrays another transport gathered: g_temps / g_logl [nranks * nchains] by global
        walker id (read only), temps[nchains] = this rank's, updated in place."""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(temps.device)
        self._chk(self._lib.rf_pt_swap_gathered_device(
            self._ctx, temps.numel(), int(rank), int(nranks), pairs.shape[0], pairs.data_ptr(), log_u.data_ptr(),
            g_temps.data_ptr(), g_logl.data_ptr(), temps.data_ptr(),
            accepted.data_ptr() if accepted is not None else None, st.cuda_stream))

    # ---- instrumentation -----------------------------------------------------
    @property
    def launch_plan(self):
        plan = (C.c_int32 * 16)()
        self._chk(self._lib.rf_get_launch_plan(self._ctx, plan))
        return {"fused": bool(plan[0]), "common_ray_fused": plan[0] == 2, "chain": plan[1], "waves_per_block": plan[2], "nsplit": plan[3],
                "lpt": bool(plan[4]), "order_reuse": bool(plan[5]), "defer_logl": plan[6],
                "bin_cutoff": bool(plan[7]), "overrides": plan[8],
                "build": ("production", "diagnostics", "diagnostics+ablate")[plan[9]],
                "block_threads_option": plan[10], "block_threads_full_batch": plan[11],
                "long_window_gemm": bool(plan[12]), "gemm_triangle": plan[12] == 2,
                "trace_window": bool(plan[13]), "staged_host_arrays": plan[14], "copy_stream": bool(plan[15])}

    def profile_enable(self, on=True):
        """on: False / True (every batch) / k > 1 (every k-th batch is timed)."""
        self._chk(self._lib.rf_profile_enable(self._ctx, int(on)))

    def profile_read(self, reset=True):
        ms = np.zeros(3)
        n = (C.c_int64 * 4)()
        self._chk(self._lib.rf_profile_read(self._ctx, _dptr(ms), n, int(reset)))
        return {"spectra_ms": ms[0], "trace_ms": ms[1], "logl_ms": ms[2], "launches": n[0],
                "spectra_launches": n[1], "trace_launches": n[2], "logl_launches": n[3]}


def compute_r_inv(nsmp, a_gus, delta, with_gap=False):
    """librfgpu's own init_r_inv (one-sided Jacobi SVD): returns (r_inv[nsmp, nsmp]
    with .ravel() == Fortran column-major r_inv(:, :), rank[, relative gap at the 1e-3 cut-off])."""
    lib = _lib.load()
    out = np.empty((nsmp, nsmp))
    rank = C.c_int32()
    gap = C.c_double()
    if lib.rf_compute_r_inv(int(nsmp), float(a_gus), float(delta), _dptr(out), C.byref(rank), C.byref(gap)):
        raise RFGPUError(lib.rf_last_error().decode())
    return (out, rank.value, gap.value) if with_gap else (out, rank.value)


def host_alloc(shape, dtype=np.float64):
    """A numpy array in pinned (page-locked, device-mapped) host memory (rf_host_alloc): host arrays handed to
    eval_batch / eval_models from such memory travel to the GPU by DMA without a staging copy.  Freed with the array."""
    import weakref

    lib = _lib.load()
    dt = np.dtype(dtype)
    n = int(np.prod(shape))
    ptr = C.c_void_p()
    if lib.rf_host_alloc(max(1, n * dt.itemsize), C.byref(ptr)):
        raise RFGPUError(lib.rf_last_error().decode())
    buf = (C.c_char * max(1, n * dt.itemsize)).from_address(ptr.value)
    arr = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
    weakref.finalize(buf, lib.rf_host_free, ptr)
    return arr


def fft_c2r(cx, nfft):
    """The reference's FFTW plan `ifft` (dfftw_plan_dft_c2r_1d, src/fftw.f90:44) on the GPU: unnormalised inverse real
    transform of cx[0 : nfft // 2 + 1] (complex128; Im of the DC and Nyquist bins ignored) -> rx[nfft] (rf_fft_c2r)."""
    lib = _lib.load()
    nfft = int(nfft)
    c = np.ascontiguousarray(np.asarray(cx, dtype=np.complex128)[: nfft // 2 + 1])
    if c.size != nfft // 2 + 1:
        raise ValueError("cx must hold nfft // 2 + 1 bins")
    out = np.empty(nfft)
    if lib.rf_fft_c2r(nfft, _dptr(c.view(np.float64)), _dptr(out)):
        raise RFGPUError(lib.rf_last_error().decode())
    return out


def fft_r2c(rx):
    """The reference's FFTW plan `ifft2` (dfftw_plan_dft_r2c_1d, src/fftw.f90:45) on the GPU: rx[nfft] ->
    cx[nfft // 2 + 1] complex128 (rf_fft_r2c)."""
    lib = _lib.load()
    r = np.ascontiguousarray(rx, dtype=np.float64)
    out = np.empty(r.size // 2 + 1, dtype=np.complex128)
    if lib.rf_fft_r2c(int(r.size), _dptr(r), _dptr(out.view(np.float64))):
        raise RFGPUError(lib.rf_last_error().decode())
    return out
