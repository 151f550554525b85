"""The result files of subroutine output_results (src/mcmc_out.f90:35-319), written from the
device-side posterior accumulators (rf_inv_amd.posterior) and the proposal counters.

Same twelve files, same row order and the same columns, so util/plot.py of the reference reads
them unchanged (it parses by whitespace).  Files the reference writes with an explicit format
((3F10.5,I6), (3F10.5), (2F10.5): syn_trace.ppd, vs_z.ppd, vp_z.ppd, vpvs_z.ppd, *.mean) are
reproduced character for character; the list-directed ones (all_models, likelihood,
num_interface.ppd, interface_depth.ppd, sigma.ppd) have a compiler-defined layout in the
reference, here: one leading blank, values at full double precision, blank separated.
"""
from __future__ import annotations

import os

import numpy as np

from .params import Params
from .posterior import PosteriorResult


def _g(x: float) -> str:
    return f"{float(x):.17g}"


def _f10_5(x: float) -> str:
    """Fortran F10.5: ten columns, asterisks when the value does not fit."""
    x = float(x)
    if x != x:
        return "       NaN"
    if x in (float("inf"), float("-inf")):
        return "  Infinity" if x > 0 else " -Infinity"
    s = f"{x:10.5f}"
    return s if len(s) == 10 else "*" * 10


def reduce_counters(counters, group=None, device=None):
    """mpi_reduce of nprop, naccept and likelihood_hist to rank 0 (src/mcmc_out.f90:54-57,72-73)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return counters
    out = type(counters)(nprop=counters.nprop.copy(), naccept=counters.naccept.copy(),
                         likelihood_hist=counters.likelihood_hist.copy(), labels=list(counters.labels))
    for name in ("nprop", "naccept", "likelihood_hist"):
        t = torch.from_numpy(getattr(out, name))
        if device is not None:
            t = t.to(device)
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM, group=group)
        setattr(out, name, t.cpu().numpy())
    return out


def output_results(p: Params, post: PosteriorResult, counters, nproc: int = 1, out_dir: str | None = None,
                   verb: bool = False):
    """Rank 0's part of output_results: `post` / `counters` are the merged (reduced) values.
    counters.likelihood_hist is indexed by iteration (1-based, entry 0 unused)."""
    out_dir = p.out_dir if out_dir is None else out_dir
    os.makedirs(out_dir, exist_ok=True)
    nmod = float(post.nmod)
    dbin_z = (p.z_max - 0.0) / p.nbin_z                                          # src/pt_mcmc.f90:423-430
    dbin_vp = (p.vp_max - p.vp_min) / p.nbin_vp
    dbin_vs = (p.vs_max - p.vs_min) / p.nbin_vs
    dbin_vpvs = (p.vpvs_max - p.vpvs_min) / p.nbin_vpvs
    dbin_amp = (p.amp_max - p.amp_min) / p.nbin_amp
    zc = [(iz - 0.5) * dbin_z for iz in range(1, p.nbin_z + 1)]

    def path(name):
        return os.path.join(out_dir, name)

    if verb:                                                                     # :104-112
        print(" --- Summary ---")
        print(" # of sampled models:", post.nmod)
        for i, lab in enumerate(counters.labels, start=1):
            print(f" # of {lab}: {counters.naccept[i]} / {counters.nprop[i]}")

    with np.errstate(divide="ignore", invalid="ignore"):
        # all_models (:114-131): one block per recorded model, unused slots skipped
        with open(path("all_models"), "w") as f:
            n_slots = int(p.niter * p.nchains * nproc / p.ncorr)
            for imod in range(min(n_slots, post.vs_model.shape[0])):
                if post.vs_model[imod, 0] < -900.0:
                    continue
                f.write(" \n")
                for iz in range(p.nbin_z):
                    f.write(f" {_g(zc[iz])} {_g(post.vp_model[imod, iz])} {_g(post.vs_model[imod, iz])}\n")
                f.write(" \n")

        # likelihood (:133-145)
        with open(path("likelihood"), "w") as f:
            for it in range(1, p.nburn + p.niter + 1):
                f.write(f" {it} {_g(counters.likelihood_hist[it] / float(p.ncool * nproc))}\n")

        # num_interface.ppd (:147-159)
        with open(path("num_interface.ppd"), "w") as f:
            for ik in range(1, p.k_max):
                f.write(f" {ik} {_g(np.float64(post.nk[ik - 1]) / nmod)}\n")

        # syn_trace.ppd (:162-181)  '(3F10.5,I6)'
        with open(path("syn_trace.ppd"), "w") as f:
            for itrc in range(p.ntrc):
                prob = post.namp[itrc].astype(np.float64) / nmod
                for it in range(p.nsmp):
                    t = _f10_5(it * p.delta + p.t_start)
                    row = prob[it]
                    f.write("".join(f"{t}{_f10_5(p.amp_min + (i + 0.5) * dbin_amp)}{_f10_5(row[i])}{itrc + 1:6d}\n"
                                    for i in range(p.nbin_amp)))

        # interface_depth.ppd (:183-196)
        with open(path("interface_depth.ppd"), "w") as f:
            for i in range(p.nbin_z):
                f.write(f" {_g(zc[i])} {_g(np.float64(post.nz[i]) / nmod)}\n")

        # sigma.ppd (:198-218)
        with open(path("sigma.ppd"), "w") as f:
            for itrc in range(p.ntrc):
                if p.sig_mode[itrc] == 1:
                    dbin_sig = (p.sig_max[itrc] - p.sig_min[itrc]) / p.nbin_sig
                    for i in range(1, p.nbin_sig + 1):
                        f.write(f" {_g((i - 0.5) * dbin_sig + p.sig_min[itrc])} "
                                f"{_g(np.float64(post.nsig[itrc, i - 1]) / nmod)} {itrc + 1}\n")

        # vs_z.ppd / vp_z.ppd / vpvs_z.ppd (:220-273)  '(3F10.5)'
        for name, hist, nbin, dbin, vmin in (("vs_z.ppd", post.nvsz, p.nbin_vs, dbin_vs, p.vs_min),
                                             ("vp_z.ppd", post.nvpz, p.nbin_vp, dbin_vp, p.vp_min),
                                             ("vpvs_z.ppd", post.nvpvsz, p.nbin_vpvs, dbin_vpvs, p.vpvs_min)):
            with open(path(name), "w") as f:
                for iv in range(1, nbin + 1):
                    v = _f10_5((iv - 0.5) * dbin + vmin)
                    prob = hist[iv - 1].astype(np.float64) / nmod
                    f.write("".join(f"{v}{_f10_5(zc[iz])}{_f10_5(prob[iz])}\n" for iz in range(p.nbin_z)))

        # vs_z.mean / vp_z.mean / vpvs_z.mean (:275-316)  '(2F10.5)'
        for name, sums in (("vs_z.mean", post.vs_mean), ("vp_z.mean", post.vp_mean),
                           ("vpvs_z.mean", post.vpvs_mean)):
            with open(path(name), "w") as f:
                for iz in range(p.nbin_z):
                    f.write(f"{_f10_5(sums[iz] / nmod)}{_f10_5(zc[iz])}\n")
