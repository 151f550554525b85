#!/usr/bin/env python3
"""bench.py -- forward+likelihood evaluations/s of the HIP hot path on N MI355X.

One "step" = one batched pass of the hot path (rf_eval_batch_device: propagator spectra ->
trace -> logL) over every walker resident on the rank, inputs already in HBM, followed by the
parallel-tempering swap step and the logL read-back to pinned host memory.  Walkers shard
across ranks with no data-path collective (weak scaling: per-GPU work fixed); the swap step
is one all_gather of (T, logL) per step over RCCL when N > 1.

Workloads (BASELINE.json configs):
  c4  (default)  the north-star shape, one GPU's shard of configs[3]: 8192 walkers/GPU
                 (1024 chains x 8 temperatures), 3 traces (P .06, P .08, S .10), nfft 4096
                 (2049 bins), <= 30 layers, PT swap
  c2             1024 walkers/GPU, 1 P trace (p 0.06), nfft 4096, <= 15 layers
  c3             8192 walkers/GPU (1024 x 8 temperatures), 1 P trace, PT swap on the device
  c5             32768 walkers/GPU (BASELINE configs[4] / 8 GPUs), 4 traces (2 P + 2 S), ocean layer, <= 31 layers,
                 PT swap
  c4common       c4 with ONE ray for all three traces (common-ray / "single FWD" mode,
                 src/forward.f90:59-91,141): fusedc_kernel, one block per walker
  c1, c2d        sample_syn shape (nfft 256) / c2 with water-level deconvolution
  c4w20, c4w60   c4 with a 20 s / 60 s time window (nsmp 401 / 1201): the long-window likelihood plan, the batch's
                 quadratic forms as one FP64-MFMA GEMM

At N = 1 the default run also measures c2, c2d, c3, c5 (BASELINE's 32768 walkers/GPU), c4common, c4w20 and c4w60
briefly into "also", plus the c4 shape with walker depths that change every step (`--perturb-nlay`: the dispatch order the
previous launch prepared is then one proposal stale, as in a real chain).  `c4host` (any `<name>host`): the c4 walkers
handed over every step as (k, z, dVp, dVs, sigma) in pinned HOST arrays through rf_eval_models + rf_commit -- the boundary
the batched sampler uses, PCIe included; reported in "also", never as the headline.

`--gpus N` without a launcher (WORLD_SIZE unset) starts N rank processes itself -- before anything touches a GPU,
one per device, rendezvous on 127.0.0.1 -- relays rank 0's JSON line and exits non-zero if any rank fails or
fewer than N devices are visible.  Under torchrun (WORLD_SIZE set) it must equal --gpus.  With N > 1 the swap
step runs through librfgpu's own RCCL communicator (rf_comm_init / rf_pt_swap_allgather_device, include/rfgpu.h);
torch.distributed is only the launcher's process group (rendezvous, barrier, max-over-ranks of the time).
Rank 0 prints ONE compact JSON line (< 4 KB: the contract's keys, `roofline`, `cpu_baseline`, the `also` rates) as the
LAST line of stdout; the full record -- launch plan, every `also` workload with its own roofline -- goes to
`bench_detail.json` (`--detail-file`) and to stderr.  Refuses to run with RFGPU_* variables in the environment
(the library reads none; a stray one must not be mistaken for a setting) -- non-default
launch plans are explicit `--opt name=value` flags and are echoed in `config`.
"""
import argparse
import glob
import hashlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector == matrix peak (datasheet)
ALSO_NOMINAL_MS = {"c1": 0.03, "c2": 0.075, "c2d": 0.08, "c3": 0.45, "c4": 1.9, "c4common": 0.95, "c5": 13.0,
                   "c4w20": 2.2, "c4w60": 4.0, "c4win": 1.8, "c5win": 12.5, "c4d": 1.9, "c5d": 13.0, "c4full": 15.0,
                   "c5full": 100.0}   # ms per step
SPEC_CLOCK_GHZ = 2.4      # the engine clock 78.6 TF is quoted at (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4e9)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
ENV_ALLOWED = {"RFGPU_BENCH_BACKEND"}   # "gloo": functional test of the N > 1 path on one GPU

WORKLOADS = {
    "c2": dict(walkers=1024, nfft=4096, rayps=[0.06], ipha=[1], k_max=15, sdep=0.0, deconv=0, temps=1,
               desc="c2: 1024 walkers/GPU x 1 P trace (p=0.06) x nfft 4096 (2049 bins) x <=15 layers, T=1"),
    "c2d": dict(walkers=1024, nfft=4096, rayps=[0.06], ipha=[1], k_max=15, sdep=0.0, deconv=1, temps=1,
                desc="c2 with water-level deconvolution (deconv_mode 1): 1024 walkers/GPU x 1 P trace x nfft 4096 "
                     "x <=15 layers"),
    "c3": dict(walkers=8192, nfft=4096, rayps=[0.06], ipha=[1], k_max=15, sdep=0.0, deconv=0, temps=8,
               desc="c3: 8192 walkers/GPU (1024 chains x 8 temperatures) x 1 P trace x nfft 4096 x <=15 layers, "
                    "PT swap on the device"),
    "c4": dict(walkers=8192, nfft=4096, rayps=[0.06, 0.08, 0.10], ipha=[1, 1, -1], k_max=30, sdep=0.0, deconv=0,
               temps=8,
               desc="c4 (north-star shape, one GPU's shard of BASELINE configs[3]): 8192 walkers/GPU (1024 chains x 8 "
                    "temperatures) x 3 traces (P .06, P .08, S .10) x nfft 4096 (2049 bins) x <=30 layers, PT swap"),
    "c4common": dict(walkers=8192, nfft=4096, rayps=[0.06, 0.06, 0.06], ipha=[1, 1, 1], a_gus=[4.0, 2.5, 1.5],
                     k_max=30, sdep=0.0, deconv=0, temps=8,
                     desc="c4common (single-FWD / common-ray mode, forward.f90:59-91,141): 8192 walkers/GPU x 3 P traces "
                          "of ONE ray (p .06; Gaussian a 4.0, 2.5, 1.5) x nfft 4096 x <=30 layers, PT swap; one "
                          "propagator pass per walker feeds three trace tails inside one fusedc_kernel block"),
    "c5": dict(walkers=32768, nfft=4096, rayps=[0.06, 0.08, 0.10, 0.12], ipha=[1, 1, -1, -1], k_max=30, sdep=2.0,
               deconv=0, temps=16,
               desc="c5 (one GPU's shard of BASELINE configs[4]: 16384 chains x 16 temperatures / 8 GPUs): 32768 "
                    "walkers/GPU x 4 traces (P .06, P .08, S .10, S .12) x nfft 4096 x ocean layer (sdep 2 km) x <=31 "
                    "layers, PT swap (BASELINE's 'buried station' has no reference behaviour: the borehole branch of "
                    "forward.f90:289-338 is commented out)"),
    # water-level deconvolution (deconv_mode 1, src/forward.f90:148-153,447-470) at the multi-trace shapes
    "c4d": dict(walkers=8192, nfft=4096, rayps=[0.06, 0.08, 0.10], ipha=[1, 1, -1], k_max=30, sdep=0.0, deconv=1,
                temps=8,
                desc="c4d: c4 with water-level deconvolution (deconv_mode 1; P: R/V, S: V/R): 8192 walkers/GPU x 3 traces "
                     "x nfft 4096 x <=30 layers, PT swap"),
    "c5d": dict(walkers=32768, nfft=4096, rayps=[0.06, 0.08, 0.10, 0.12], ipha=[1, 1, -1, -1], k_max=30, sdep=2.0,
                deconv=1, temps=16,
                desc="c5d: c5 with water-level deconvolution (deconv_mode 1): 32768 walkers/GPU x 4 traces x nfft 4096 x "
                     "ocean layer x <=31 layers, PT swap"),
    # BASELINE configs[3] / configs[4] at their FULL size on ONE GPU (the 8-GPU job's walkers all resident here)
    "c4full": dict(walkers=65536, nfft=4096, rayps=[0.06, 0.08, 0.10], ipha=[1, 1, -1], k_max=30, sdep=0.0, deconv=0,
                   temps=8,
                   desc="c4full: ALL of BASELINE configs[3] on one GPU: 65536 walkers (8192 chains x 8 temperatures) x 3 "
                        "traces x nfft 4096 x <=30 layers, PT swap; 12.9 GB of double-buffered resident traces"),
    "c5full": dict(walkers=262144, nfft=4096, rayps=[0.06, 0.08, 0.10, 0.12], ipha=[1, 1, -1, -1], k_max=30, sdep=2.0,
                   deconv=0, temps=16,
                   desc="c5full: ALL of BASELINE configs[4] on one GPU: 262144 walkers (16384 chains x 16 temperatures) x 4 "
                        "traces x nfft 4096 x ocean layer x <=31 layers, PT swap; 68.7 GB of double-buffered resident "
                        "traces"),
    # real time windows: the reference takes any window up to npts_max = 2000 samples (src/params.f90:44) and its
    # quadratic form grows with nsmp^2 (src/likelihood.f90:92-93): the long-window plan (one FP64-MFMA GEMM per batch)
    "c4w20": dict(walkers=8192, nfft=4096, rayps=[0.06, 0.08, 0.10], ipha=[1, 1, -1], k_max=30, sdep=0.0, deconv=0,
                  temps=8, t_end=20.0,
                  desc="c4w20: the c4 shape with a 20 s time window (t_end 20 -> nsmp 401; R^-1 1.29 MB per trace): "
                       "8192 walkers/GPU x 3 traces x nfft 4096 x <=30 layers, PT swap; quadratic forms as one "
                       "FP64-MFMA GEMM per batch (phi_gemm_kernel)"),
    "c4w60": dict(walkers=8192, nfft=4096, rayps=[0.06, 0.08, 0.10], ipha=[1, 1, -1], k_max=30, sdep=0.0, deconv=0,
                  temps=8, t_end=60.0,
                  desc="c4w60: the c4 shape with a 60 s time window (t_end 60 -> nsmp 1201; R^-1 11.5 MB per trace): "
                       "8192 walkers/GPU x 3 traces x nfft 4096 x <=30 layers, PT swap; quadratic forms as one "
                       "FP64-MFMA GEMM per batch (phi_gemm_kernel)"),
    # the option a host that only ever reads samples 1 .. nsmp can set (pt_control_batched does): never the headline
    "c4win": dict(walkers=8192, nfft=4096, rayps=[0.06, 0.08, 0.10], ipha=[1, 1, -1], k_max=30, sdep=0.0, deconv=0,
                  temps=8, options={"trace_window": 1.0},
                  desc="c4win: c4 with rf_set_option('trace_window', 1): only samples 1 .. nsmp of every trace are stored "
                       "(what the likelihood and the histograms read, src/likelihood.f90:88, src/pt_mcmc.f90:273-274); "
                       "same logL bit for bit; NOT the reference's full rft(nfft, ntrc) image"),
    "c5win": dict(walkers=32768, nfft=4096, rayps=[0.06, 0.08, 0.10, 0.12], ipha=[1, 1, -1, -1], k_max=30, sdep=2.0,
                  deconv=0, temps=16, options={"trace_window": 1.0},
                  desc="c5win: c5 with rf_set_option('trace_window', 1): 0.2 GB of resident traces instead of 8.6 GB"),
    "c1": dict(walkers=1024, nfft=256, rayps=[0.06, 0.08], ipha=[1, 1], k_max=10, sdep=2.0, deconv=0, temps=1,
               desc="c1-shape: 1024 walkers/GPU x 2 P traces x nfft 256 x ocean x <=11 layers"),
}


def make_params(w):
    """sample_syn-shaped params (tests/golden/sample_syn/params.in) with the workload's
    geometry; obs filled later."""
    from rf_inv_amd import get_params

    p = get_params(os.path.join(ROOT, "tests", "golden", "sample_syn", "params.in"))
    n = len(w["rayps"])
    p.ntrc, p.nfft = n, w["nfft"]
    p.rayps = np.array(w["rayps"], dtype=np.float64)
    p.ipha = np.array(w["ipha"], dtype=np.int32)
    p.a_gus = np.array(w.get("a_gus", [4.0] * n), dtype=np.float64)
    p.sig_min = np.full(n, 0.01); p.sig_max = np.full(n, 0.01); p.sig_mode = np.zeros(n, dtype=np.int32)
    p.k_min, p.k_max, p.sdep, p.deconv_mode = 1, w["k_max"], w["sdep"], w["deconv"]
    p.delta = float(np.float32(0.05))
    # nsmp as read_obs forms it (src/params.f90:449-451) for a SAC file that starts at b = 0: 101 for the 5 s window
    from rf_inv_amd.params import _nint

    p.t_start, p.t_end = 0.0, float(w.get("t_end", 5.0))
    p.nsmp = _nint(p.t_end / p.delta) - _nint(p.t_start / p.delta) + 1
    return p


def draw_walkers(p, ref, first_id, count, seed=12345678, return_models=False, procs=0):
    """Walker models from the init_model prior (reference src/model.f90:66-95) conditioned on
    validity, with k uniform in [k_min, k_max), from a counter-based RNG
    keyed by seed + global walker id.  init_model rejects whole models, which makes deep
    stacks vanishingly rare (P(valid) ~ 0.68^nlay at dVs sigma 2.0); here k is drawn once and
    the rejection is applied per component (interface set until the thickness rules hold,
    each dVs until its layer passes the range rules) -- the same distribution as whole-model
    rejection given k, since the validity rules factorise that way for vp_mode = 0.
    Every walker's model depends on seed + its global id only; 65536 walkers or more are drawn by child processes
    (procs: 0 = one per physical core, 1 = in this process)."""
    from rf_inv_amd import format_model

    if count >= 65536 and procs != 1:
        return _draw_walkers_pool(p, ref, first_id, count, seed, return_models, procs)
    pad = p.k_max + 2
    layers = np.ones((count, 4, pad))
    nlay = np.zeros(count, dtype=np.int32)
    m_k = np.zeros(count, dtype=np.int32)
    m_z, m_dvp, m_dvs = np.zeros((count, p.k_max)), np.zeros((count, p.k_max)), np.zeros((count, p.k_max))
    vs0, vp0 = float(ref.vs_ref[0]), float(ref.vp_ref[0])
    assert np.all(ref.vs_ref == vs0) and np.all(ref.vp_ref == vp0) and p.vp_mode == 0

    def draw_dvs(g):
        while True:
            d = g.standard_normal() * p.dvs_prior
            b = vs0 + d
            if p.vs_min <= b <= p.vs_max and p.vpvs_min <= vp0 / b <= p.vpvs_max:
                return d

    for i in range(count):
        g = np.random.Generator(np.random.Philox(key=seed + first_id + i))
        k = p.k_min + int(g.random() * (p.k_max - p.k_min))
        # rejection on the interface set, a block of tries at a time: the stream is consumed exactly as by one
        # g.random(k) per try up to and including the accepted one (state restored, then that many doubles drawn)
        while True:
            st = g.bit_generator.state
            blk = np.sort(p.z_min + g.random((64, k)) * (p.z_max - p.z_min), axis=1)
            th = np.diff(np.concatenate([np.full((64, 1), p.sdep), blk], axis=1), axis=1)
            good = (th[:, 0] >= 0.125 * vp0) & np.all(th[:, 1:] >= p.h_min, axis=1)
            if good.any():
                j = int(np.argmax(good))
                zs = blk[j]
                g.bit_generator.state = st
                g.random((j + 1) * k)
                break
        z = np.zeros(max(p.k_max - 1, 1)); dvp = np.zeros(p.k_max); dvs = np.zeros(p.k_max)
        z[:k] = g.permutation(zs)
        dvs[:k] = [draw_dvs(g) for _ in range(k)]
        dvs[p.k_max - 1] = draw_dvs(g)
        nl, a, b, r, h, ok = format_model(p, ref, k, z, dvp, dvs)
        assert ok
        nlay[i] = nl
        layers[i, 0, :nl], layers[i, 1, :nl], layers[i, 2, :nl], layers[i, 3, :nl] = a, b, r, h
        m_k[i] = k
        m_z[i, :z.size], m_dvp[i], m_dvs[i] = z, dvp, dvs
    if return_models:
        return nlay, layers, (m_k, m_z, m_dvp, m_dvs)
    return nlay, layers


def _draw_walkers_pool(p, ref, first_id, count, seed, return_models, procs):
    """draw_walkers over child processes (`python bench.py --draw-worker job.pkl`, started like any other child program:
    fresh interpreters that never touch the GPU), one contiguous block of walker ids each."""
    import pickle
    import subprocess
    import tempfile

    procs = procs or max(1, min(physical_cores(), 32))
    chunk = -(-count // procs)
    with tempfile.TemporaryDirectory() as work:
        jobs = []
        for j, a in enumerate(range(0, count, chunk)):
            job = os.path.join(work, f"job{j}.pkl")
            with open(job, "wb") as fh:
                pickle.dump((p, ref, first_id + a, min(chunk, count - a), seed, os.path.join(work, f"out{j}.npz")), fh)
            # (no OpenMP binding of this process, and no profiler preload: the children are plain numpy programs)
            env = {k: v for k, v in os.environ.items()
                   if not (k.startswith("OMP_") or k.startswith("ROCP") or k.startswith("ROCPROF") or k == "LD_PRELOAD")}
            env.update(OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
            jobs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), "--draw-worker", job], env=env,
                                          stdout=subprocess.DEVNULL), os.path.join(work, f"out{j}.npz")))
        parts = []
        try:
            for q, out in jobs:
                if q.wait(timeout=1800) != 0:
                    raise RuntimeError("bench.py: a --draw-worker child failed")
                with np.load(out) as z:
                    parts.append([z[n] for n in ("nlay", "layers", "k", "z", "dvp", "dvs")])
        finally:
            for q, _ in jobs:
                if q.poll() is None:
                    q.kill()
                    q.wait()
    nlay, layers = np.concatenate([q[0] for q in parts]), np.concatenate([q[1] for q in parts])
    models = tuple(np.concatenate([q[j] for q in parts]) for j in range(2, 6))
    return (nlay, layers, models) if return_models else (nlay, layers)


def _draw_worker(job):
    import pickle

    with open(job, "rb") as fh:
        p, ref, first_id, count, seed, out = pickle.load(fh)
    nlay, layers, (k, z, dvp, dvs) = draw_walkers(p, ref, first_id, count, seed=seed, return_models=True, procs=1)
    np.savez(out, nlay=nlay, layers=layers, k=k, z=z, dvp=dvp, dvs=dvs)


def run_host_boundary(workload, steps, device, prewarm_s=1.0, warmup=5):
    """The boundary as the batched sampler uses it (rf_eval_models + rf_commit, include/rfgpu.h): the same walkers as
    `workload`, handed over as (k, z, dVp, dVs, sigma) in PINNED host arrays every step -- H2D by DMA, format_model on the
    device, the kernels of the resident path, logL back into host memory -- i.e. the PCIe-inclusive rate (DESIGN.md
    section 6).  No temperature swap: a host that hands models over owns the temperatures.  Never the headline."""
    import time

    import torch

    from rf_inv_amd import RFEngine, format_model, read_ref_model
    from rf_inv_amd.engine import host_alloc
    from rf_inv_amd.likelihood import init_r_inv

    w = dict(WORKLOADS[workload])
    p = make_params(w)
    ref = read_ref_model(os.path.join(ROOT, "tests", "golden", "sample_syn", "model", "sample.velmod"))
    nb = w["walkers"]
    nlay, layers, (m_k, m_z, m_dvp, m_dvs) = draw_walkers(p, ref, 0, nb, return_models=True)
    sig = np.full((nb, p.ntrc), 0.01)
    r_inv = init_r_inv(p.nsmp, p.a_gus, p.delta)
    zt = np.zeros(max(p.k_max - 1, 1)); dvt = np.zeros(p.k_max); dst = np.zeros(p.k_max)
    zt[:3] = [3.1 + p.sdep, 7.7 + p.sdep, 14.2 + p.sdep]; dst[:3] = [-0.6, 0.2, 0.5]; dst[p.k_max - 1] = 0.9
    nl_t, a_t, b_t, r_t, h_t, ok = format_model(p, ref, 3, zt, dvt, dst)
    kw = dict(nfft=p.nfft, delta=p.delta, t_start=p.t_start, deconv_mode=p.deconv_mode, sdep=p.sdep, rayps=p.rayps,
              a_gus=p.a_gus, ipha=p.ipha, nsmp=p.nsmp, nlay_max=p.k_max + 2, device=device)
    with RFEngine(obs=np.zeros((p.ntrc, p.nsmp)), r_inv=r_inv, max_walkers=1, **kw) as e0:
        obs = np.ascontiguousarray(e0.calc_rf(nl_t, a_t, b_t, r_t, h_t)[:p.nsmp].T)

    def pin(a):
        b = host_alloc(a.shape, a.dtype)
        b[...] = a
        return b

    with RFEngine(obs=obs, r_inv=r_inv, max_walkers=nb, **kw) as eng:
        eng.set_model(p, ref)
        ids, k, z, dvp, dvs, sg = (pin(a) for a in (np.arange(nb, dtype=np.int32), m_k, m_z, m_dvp, m_dvs, sig))
        acc = np.zeros(nb, dtype=np.int32)
        acc[::2] = 1
        want = eng.eval_batch(np.arange(nb), nlay, layers, sig)            # the resident path's values on these walkers
        # Pre-warm by TIME like the resident runs (drawing the walkers above kept the host busy for seconds with the GPU
        # idle: its clocks are down, and five 2 ms steps do not bring them back -- the driver's r04 run of this leg, 20
        # steps after 5 warm-ups, read 3.57 ms per step against 2.05 here after 200: profiles/r05_host_boundary_probe.txt)
        n_pre, t_pre = 0, time.perf_counter()
        while n_pre < warmup or time.perf_counter() - t_pre < prewarm_s:
            ll = eng.eval_models(ids, k, z, dvp, dvs, sg)
            eng.commit(ids, acc)
            n_pre += 1
        assert eng.launch_plan["staged_host_arrays"] == 0
        torch.cuda.synchronize(device)
        eng.profile_enable(max(1, steps // 16))
        t_eval = t_commit = 0.0
        per_step = np.zeros(steps)
        t0 = time.perf_counter()
        for i in range(steps):
            ta = time.perf_counter()
            ll = eng.eval_models(ids, k, z, dvp, dvs, sg)
            tb = time.perf_counter()
            eng.commit(ids, acc)
            tc = time.perf_counter()
            t_eval += tb - ta
            t_commit += tc - tb
            per_step[i] = tc - ta
        dt = time.perf_counter() - t0
        eng.profile_enable(False)
        prof = eng.profile_read()
        n_l = max(prof["launches"], 1)
        bytes_in = ids.nbytes + k.nbytes + z.nbytes + dvs.nbytes + sg.nbytes + (dvp.nbytes if p.vp_mode == 1 else 0)
        return {"value": nb * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
                "prewarm": {"seconds": prewarm_s, "steps": n_pre},
                # (a host hiccup inside a short timed region shows as mean >> median)
                "ms_per_step_median": 1e3 * float(np.median(per_step)), "ms_per_step_max": 1e3 * float(per_step.max()),
                # host-side split of a step: the synchronous rf_eval_models call (DMA in, format_model, stage, the
                # evaluation kernels, logL out), rf_commit (returns without waiting), and inside the former the
                # HIP-event time of the evaluation kernels
                "phase_ms": {"eval_models_call": 1e3 * t_eval / steps, "commit_call": 1e3 * t_commit / steps,
                             "kernels": (prof["spectra_ms"] + prof["trace_ms"] + prof["logl_ms"]) / n_l},
                "config": {"workload": w["desc"].replace(", PT swap on the device", "").replace(", PT swap", "") + " -- handed over every step as (k, z, dVp, dVs, sigma) in pinned HOST arrays "
                           "(rf_eval_models + rf_commit): DMA, format_model on the device, logL into host memory; no swap",
                           "walkers_per_gpu": nb, "host_bytes_in_per_step": int(bytes_in), "host_bytes_out_per_step": 8 * nb,
                           "pcie_inclusive": True},
                "same_logl_as_the_resident_path": bool(np.array_equal(ll, want))}


def alg_work(p, nlay, common):
    """Algorithmic flops / bytes per evaluation in the REFERENCE's arithmetic
    (SURVEY.md section 8d): returns (F_spectra[nb], F_total[nb], B_alg[nb])."""
    nh = p.nfft // 2 + 1
    sea = 1 if p.sdep > 0 else 0
    nfwd = 1 if common else p.ntrc
    n_ifft = 2 if p.deconv_mode == 0 else 1
    f_spec = nfwd * nh * ((nlay - 1 - sea) * 570.0 + 580.0)
    f_rest = p.ntrc * (n_ifft * 2.5 * p.nfft * math.log2(p.nfft) + 2.0 * p.nsmp ** 2 + 4.0 * p.nsmp + 3.0 * p.nfft)
    b = 8.0 * (4 * nlay + p.ntrc) + 8.0 * p.nfft * p.ntrc + 8.0
    return f_spec, f_spec + f_rest, b


def quad_flops_run(nsmp, plan):
    """Multiply-add flops the likelihood's quadratic form executes per (walker, trace): 2 nsmp^2 (quad_form /
    phi_deferred_kernel: the reference's matmul); on the long-window plan the GEMM's K loop per 64-column chunk c --
    rows 0 .. 64 (c + 1) - 1 of the triangular image ("gemm_triangle", default), all kp rows of the full one."""
    if not plan.get("long_window_gemm"):
        return 2.0 * nsmp ** 2
    kp = (nsmp + 15) // 16 * 16
    rows = sum(min(kp, 64 * (c + 1)) if plan.get("gemm_triangle") else kp for c in range((kp + 63) // 64))
    return 2.0 * 64.0 * rows


def committed_counters(kernel, grid_threads, long_window=False, decon=False):
    """Per-launch hardware counters of `kernel` at exactly this launch shape from the newest committed
    profile (profiles/rNN_counters.json, written by tools/collect_counters.sh from separate rocprofv3
    --pmc passes over THIS script; FETCH_SIZE doubled per the gfx950 correction of
    MI355X_MICROARCH.md).  bench.py itself cannot collect PMC counters.  None when no launch of that
    kernel and grid is in the file.  Workloads that launch the same kernel at the same grid but execute different
    instructions or write different bytes keep their counters in files of their own, which the other workloads' look-ups
    skip: the long-window workloads (c4w20, c4w60: rNN_longwindow_counters.json, collected over `--workload c4w60 --also
    c4w20`) and the water-level deconvolution workloads (c2d, c4d, c5d: rNN_decon_counters.json, collected over `--workload
    c4d --also c2d,c5d`)."""
    tag = "longwindow" if long_window else "decon" if decon else ""
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")), reverse=True)
             if (tag in os.path.basename(f) if tag else not any(t in os.path.basename(f) for t in ("longwindow", "decon")))]
    for f in files:
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for e in d.get("kernels", []):
            if e["kernel"].split("<")[0] == kernel and int(e["grid_threads"]) == int(grid_threads):
                out = dict(e["counters"])
                out["_file"] = os.path.relpath(f, ROOT)
                out["_lib_sha256"] = d.get("lib_sha256")
                out["_kernels_sha256"] = d.get("kernels_sha256")
                out["_clock_ghz"] = e.get("clock_ghz")
                out["_mean_s"] = e.get("mean_s")
                return out
    return None


def executed_fp64_flops(c):
    """fp64 flops one launch EXECUTED, from the SQ_INSTS_VALU_*_F64 wave-instruction counters
    (64 lanes; an FMA is 2 flops) -- the count the fp64 VALU roofline is priced on."""
    try:
        return 64.0 * (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]
                       + 2.0 * c["SQ_INSTS_VALU_FMA_F64"])
    except (KeyError, TypeError):
        return None


def physical_cores():
    """Physical cores of the host (unique (package, core) pairs), not SMT threads."""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        n = len(seen)
    except OSError:
        n = 0
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = min(n or avail, avail)
    # a container's CPU quota (cgroup v2 cpu.max / v1 cfs_quota) caps what the threads can actually use
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(p, obs, r_inv, nlay, layers, sig, budget_s=15.0):
    """The CPU side of the comparison, on a bounded sample of the same workload.
    kind "reference" (when oracle/_ref/cpu_o2/ref_path_dump is there and runs): ALL of the reference's own sources compiled
    unmodified and run on the host cores only -- its own module fftw on the FFTW3 interface of the image's Intel MKL, dgesvd
    from MKL, no GPU (reference_path_rate) --, one process per physical core at once: the reference's deployment, one MPI
    rank per core, no communication inside an evaluation.  `o0`: the same with the reference Makefile's default -O0
    class.  The port's figures sit beside it under `port`.
    kind "port" (otherwise): oracle/rf_oracle.c, a scalar fp64 restatement of the reference's arithmetic, in its speed
    build (-O3 -march=native, same values as the checker build), one OpenMP thread per physical core.
    Returns (record, the port's logL on the sample, how many of the rank's walkers the sample covers)."""
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")
    from oracle import rf_oracle as orc

    orc.build()
    cfg = dict(nfft=p.nfft, deconv_mode=p.deconv_mode, delta=p.delta, t_start=p.t_start, sdep=p.sdep,
               rayps=p.rayps, a_gus=p.a_gus, ipha=p.ipha)
    cores = max(1, min(physical_cores(), orc.max_threads()))
    nb = len(nlay)
    # parity sample + a first timing on a slice that one pass finishes quickly
    n0 = min(nb, 16 * cores)
    t0 = time.perf_counter()
    orc.lib_fast()
    ll = orc.eval_batch(cfg, obs, r_inv, nlay[:n0], layers[:n0], sig[:n0], p.nsmp, nthreads=cores, fast=True)
    dt0 = time.perf_counter() - t0
    n = int(min(max(n0, budget_s / max(dt0, 1e-3) * n0), 200 * nb))
    idx = np.arange(n) % nb
    t0 = time.perf_counter()
    ll_all = orc.eval_batch(cfg, obs, r_inv, nlay[idx], layers[idx], sig[idx], p.nsmp, nthreads=cores, fast=True)
    dt = time.perf_counter() - t0
    # single-core rate of the port on the same sample (for the per-core comparison with the reference probe)
    n1 = max(8, min(nb, int(2.0 / max(dt / n * cores, 1e-4))))
    t0 = time.perf_counter()
    orc.eval_batch(cfg, obs, r_inv, nlay[:n1], layers[:n1], sig[:n1], p.nsmp, nthreads=1, fast=True)
    dt1 = time.perf_counter() - t0
    base = {"value": n / dt, "unit": "evals/s", "cores": cores, "kind": "port",
            "per_core": n / dt / cores, "single_core": n1 / dt1,
            "sample_short": f"{n} evals: this workload's walkers cycled, oracle/rf_oracle.c -O3, {cores} OpenMP threads, "
                            f"{dt:.1f} s wall",
            "sample": f"{n} evals = the rank's walker set cycled ({nb} walkers, mean {float(nlay.mean()):.1f} layers), "
                      f"oracle/rf_oracle.c (gcc {' '.join(orc.FAST_FLAGS)}, OpenMP x{cores} threads = physical cores, "
                      f"{dt:.1f} s wall); single-core rate on {n1} evals"}
    # The reference's own forward + likelihood code on this box's cores (sigma 0.01 like `sig`): one process alone, then one
    # per physical core at once -- the reference's deployment (one MPI rank per core).  When it runs, IT is the CPU
    # baseline (kind "reference") and the port's figures move to `port`; otherwise the port stays the baseline.
    n24 = min(nb, 24)
    t0 = time.perf_counter()
    orc.eval_batch(cfg, obs, r_inv, nlay[:n24], layers[:n24], sig[:n24], p.nsmp, nthreads=1, fast=True)
    port24 = n24 / (time.perf_counter() - t0)
    # (the reference reads its observed traces from SAC files: float32 samples -- the port is given the same here)
    obs32 = obs.astype(np.float32).astype(np.float64)
    one, ll_ref = reference_path_rate(p, obs32, budget_s=2.5, count=n24, procs=1)
    full = reference_path_rate(p, obs32, budget_s=6.0, count=n24, procs=cores)[0] if one is not None and cores > 1 else one
    if full is not None:
        ll24 = orc.eval_batch(cfg, obs32, r_inv, nlay[:n24], layers[:n24], sig[:n24], p.nsmp, nthreads=1, fast=True)
        port = dict(base)
        base = dict(full)
        base["single_core"] = one["value"]
        # the reference Makefile's default class (-O0, Makefile:11-13) beside the optimised build
        one0 = reference_path_rate(p, obs32, budget_s=2.0, count=n24, procs=1, build="cpu_o0")[0]
        full0 = reference_path_rate(p, obs32, budget_s=4.0, count=n24, procs=cores, build="cpu_o0")[0] if one0 and cores > 1 else one0
        if full0 is not None:
            base["o0"] = {"value": full0["value"], "cores": full0["cores"], "per_core": full0["per_core"],
                          "single_core": one0["value"], "build": "cpu_o0", "sample": full0["sample"]}
        base["port"] = {k: port[k] for k in ("value", "unit", "cores", "kind", "per_core", "single_core", "sample")}
        base["port"]["single_core_same_walkers"] = port24
        base["reference_over_port"] = {"all_cores": full["value"] / port["value"], "single_core_same_walkers": one["value"] / port24}
        # (R^-1 of the port comes from scipy's dgesvd, the reference's from MKL's: two SVDs of an ill-conditioned matrix)
        base["max_rel_dlogl_port_vs_reference"] = float(np.max(np.abs(ll24 - ll_ref) / np.abs(ll_ref)))
    nuse = min(nb, n)
    return base, ll_all[:nuse], nuse


def reference_path_rate(p, obs, budget_s=8.0, count=24, procs=1, build="cpu_o2"):
    """The reference's OWN forward + likelihood code timed on this box's HOST CORES ONLY: oracle/_ref/<build>/ref_path_dump
    = all of the reference's sources compiled unmodified (oracle/Makefile.ref: amdflang, `cpu_o2` = -O2 -ffp-contract=off,
    `cpu_o0` = the reference Makefile's default -O0 class), its own src/fftw.f90 on the FFTW3 interface of the image's Intel
    MKL, dgesvd from the same MKL -- no product object linked, no GPU touched -- looping calc_likelihood(fwd_flag = .true.)
    (src/likelihood.f90:56-101: format_model, calc_rf, misfit, quadratic form, logL) over the first `count` of this
    workload's walkers.  The timed loop is inside the program (system_clock around the calls: start-up, init_r_inv's SVD
    and file IO are outside).
    procs > 1: that many processes at once, one per host core, the way the reference runs (one MPI rank per core, no
    communication inside an evaluation): the box's rate = all their evaluations / the slowest one's time.
    Returns (record or None, logL of the sample or None)."""
    import subprocess
    import tempfile

    from oracle import refrun

    if not refrun.available(build):
        return None, None
    from rf_inv_amd import read_ref_model

    ref = read_ref_model(os.path.join(ROOT, "tests", "golden", "sample_syn", "model", "sample.velmod"))
    nlay, _, (m_k, m_z, m_dvp, m_dvs) = draw_walkers(p, ref, 0, count, return_models=True, procs=1)   # the rank-0 walkers again
    n = count
    try:
        with tempfile.TemporaryDirectory() as work:
            refrun.write_run_dir(work, p, obs=obs, header="bench.py cpu_baseline")
            refrun.write_models(os.path.join(work, "models.txt"), p.k_max, m_k, m_z, m_dvp, m_dvs, np.full((n, p.ntrc), 0.01))
            # one pass first (page-in; its time sizes the timed run to the budget)
            first = refrun.run_path(build, work, n, p, reps=1)
            reps = int(max(1, min(200, budget_s / max(first["seconds"], 1e-3))))
            exe = refrun.exe(build, "ref_path_dump")
            runs = [subprocess.Popen([exe, "params.in", "models.txt", f"ref_{i}.bin", str(reps)], cwd=work, env=refrun.clean_env(),
                                     stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(procs)]
            try:
                outs = [q.communicate(timeout=300)[0] for q in runs]
            finally:
                for q in runs:              # (a child that outlives its time must not keep the cores busy under the GPU runs)
                    if q.poll() is None:
                        q.kill()
                        q.wait()
            if any(q.returncode != 0 for q in runs):
                return None, None
            each = []
            for o in outs:
                tok = [l for l in o.splitlines() if "ref_path_dump: seconds" in l][0].split()
                each.append((float(tok[2]), int(tok[4])))
            secs, evals = max(e[0] for e in each), sum(e[1] for e in each)
        flags = "-O2 -ffp-contract=off" if build == "cpu_o2" else "-O0 -ffp-contract=off (the reference Makefile's default class)"
        return ({"value": evals / secs, "unit": "evals/s", "cores": procs, "kind": "reference", "per_core": evals / secs / procs,
                 "build": build,
                 "sample_short": f"{evals} calc_likelihood calls of the reference's own code ({build}) on host cores only, "
                                 f"{procs} process(es) = cores, {secs:.1f} s",
                 "sample": f"{evals} calc_likelihood(fwd_flag = .true.) calls on {n} of this workload's walkers (mean "
                           f"{float(np.mean(nlay)):.1f} layers) by {procs} concurrent process(es), one per core, {secs:.1f} s (the "
                           f"slowest); all of the reference's sources unmodified, amdflang {flags}, its own module fftw on the "
                           "FFTW3 interface of Intel MKL, dgesvd from MKL; no GPU involved"},
                first["logl"])
    except (OSError, subprocess.SubprocessError, ValueError, IndexError, RuntimeError, AssertionError, TypeError, KeyError):
        return None, None


def visible_gpus():
    """GPUs this process may use, counted WITHOUT the HIP runtime (the launcher must never initialise a GPU): the
    KFD topology's nodes that have SIMDs, cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set."""
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(f):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
        except OSError:
            pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(n, argv):
    """`bench.py --gpus N` without a launcher: start N rank processes, one per GPU, and relay rank 0's JSON line.
    This parent never touches a GPU (it counts them in sysfs, visible_gpus) and never
    exec()s: the ranks are ordinary children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment,
    exactly what `python -m torch.distributed.run --nproc-per-node N` would give them.  Returns the exit code:
    non-zero if any rank failed."""
    import signal
    import socket
    import subprocess

    have = visible_gpus()
    shared = os.environ.get("RFGPU_BENCH_BACKEND", "nccl") != "nccl"   # functional test: ranks may share a GPU
    if have < n and not shared:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs, this node shows {have}: refusing to report a {n}-GPU "
              "number from fewer devices (RCCL needs one GPU per rank)", file=sys.stderr)
        # (the one JSON line of a failed run: value null, what happened, how far it got)
        print(json.dumps({"metric": METRIC, "value": None, "unit": "evals/s", "n_gpus": n, "ms_per_step": None,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "error": f"--gpus {n} needs {n} visible GPUs, this node shows {have}", "reached": "launcher",
                          "config": {"workload": None, "rccl": {"ranks": 0, "transport": None}}}, separators=(",", ":")), flush=True)
        return 2
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver supports only dmabuf IPC; with the legacy mode RCCL's (and torch's)
    # cross-process device-memory handles fail with `hipIpcGetMemHandle: invalid argument`.  The image exports it already
    # (here and on the GPU boxes); it is repeated so that ranks started from a scrubbed environment still get it.
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # every rank is watched: the first one that fails takes the others down at once (a rank that dies in its
    # rendezvous would otherwise leave rank 0 waiting for the process group's timeout)
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    while True:
        codes = [q.poll() for q in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            for q in procs:
                if q.poll() is None:
                    q.send_signal(signal.SIGTERM)
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    for q in procs:
        try:
            q.wait(timeout=10)
        except subprocess.TimeoutExpired:
            q.kill()
            q.wait()
    reader.join(timeout=10)
    out0 = out0[0] if out0 else ""
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    if rc:
        print(f"bench.py: a rank exited with code {rc}", file=sys.stderr)
    return rc or 0


METRIC = "forward+likelihood evals/sec (whole node)"
HEADLINE_MAX_BYTES = 4096


def _sig(x, digits=6):
    """Numbers of the headline rounded to `digits` significant digits (the side file keeps them in full)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if math.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def headline_line(full, detail_file):
    """The ONE line the driver parses: the contract's keys + `roofline` + `cpu_baseline`, nothing that repeats (launch
    plan, notes, per-workload records live in `detail_file`).  Always shorter than HEADLINE_MAX_BYTES -- round 4's line
    carried eleven `also` records (34.8 KB) and the driver could not parse it (tests/test_bench_line.py)."""
    pick = lambda d, keys: {k: d.get(k) for k in keys if d is not None and k in d}
    cfg = full.get("config", {})
    roof = full.get("roofline") or {}
    hbm = full.get("roofline_hbm") or {}
    cpu = full.get("cpu_baseline")
    par = full.get("parity_in_bench")
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                     "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    rccl = cfg.get("rccl") or {}
    line["config"] = dict(pick(cfg, ("workload", "walkers_per_gpu", "nfft", "ntrc", "nsmp", "k_max", "mean_nlay",
                                     "temperatures", "pt_swap", "parallelism")),
                          rccl={k: (v[:160] if isinstance(v, str) else v) for k, v in
                                pick(rccl, ("ranks", "version", "transport", "control_plane", "init_s", "library")).items()},
                          lib_sha256=(cfg.get("lib") or {}).get("sha256", "")[:16],
                          overrides=cfg.get("overrides") or {})
    line["roofline"] = dict(pick(roof, ("bound", "unit", "peak", "achieved", "frac", "traffic", "kernel", "kernel_ms",
                                        "clock_ghz", "frac_at_clock", "frac_step", "frac_with_stale_counters")),
                            algorithmic_bytes=full.get("alg_bytes_per_step"),
                            counters_file=(roof.get("counters") or {}).get("file"))
    line["roofline_hbm"] = pick(hbm, ("bound", "unit", "peak", "achieved", "frac"))
    if cpu:
        line["cpu_baseline"] = dict(pick(cpu, ("value", "unit", "cores", "kind", "per_core", "single_core", "build")),
                                    sample=cpu.get("sample_short") or (cpu.get("sample") or "")[:120])
        if cpu.get("o0"):               # the reference Makefile's default -O0 class beside the -O2 build
            line["cpu_baseline"]["o0"] = pick(cpu["o0"], ("value", "cores", "per_core", "single_core"))
        if cpu.get("port"):             # kind "reference": the C port's figures beside it
            line["cpu_baseline"]["port"] = pick(cpu["port"], ("value", "cores", "per_core"))
            line["cpu_baseline"]["reference_over_port"] = (cpu.get("reference_over_port") or {}).get("all_cores")
    if par:
        line["parity_in_bench"] = pick(par, ("n", "max_abs_dlogl", "max_rel_dlogl", "within_tolerance",
                                             "n_used_kappa_allowance", "within_kappa_rule"))
    for k in ("swap_replay_ok", "cross_rank_swaps", "cpu_leg_error"):
        if k in full:
            line[k] = full[k]
    if full.get("also"):
        line["also"] = {k: (v.get("value") if v.get("value") is not None else "failed")
                        for k, v in full["also"].items()}     # evals/s only; records in detail_file
    line["detail_file"] = detail_file
    def dumps(x):
        y = _sig(x)
        y["value"], y["ms_per_step"] = x.get("value"), x.get("ms_per_step")     # the headline pair in full precision
        return json.dumps(y, separators=(",", ":"))

    s = dumps(line)
    if len(s) >= HEADLINE_MAX_BYTES:                      # cannot happen with the keys above; never print a long line
        line.pop("also", None)
        line["config"].pop("overrides", None)
        s = dumps(line)
    assert len(s) < HEADLINE_MAX_BYTES, len(s)
    return s


KAPPA_MIN, KAPPA_SCALE = 1000.0, 1000.0      # the conditioning rule of tests/test_gpu_configs.py


def parity_report(orc, cfg, obs, r_inv, nlay, layers, sig, nsmp, ll_gpu, ll_cpu, nthreads=1):
    """GPU vs oracle logL over the compared walkers, with the conditioning accounted for.  Tolerance:
    |dlogL| <= max(1e-9, 1e-12 |logL|).  Without deconvolution a trace is divided by the SIGNED maximum of the
    filtered vertical trace (forward.f90:201-202); kappa = max|rx| / |maxval(rx)| says how much of that trace's
    scale cancels in the divisor -- its rounding, in any double evaluation including the reference's, is amplified
    kappa-fold in every sample and 2 kappa-fold in logL.  An item may exceed the plain tolerance only if its kappa
    (from the oracle's own vertical trace) is >= 1000, and must then stay within tolerance * kappa / 1000 (the scale the reference's own spread supports:
    profiles/r06_kappa_reference_spread.json, tests/helpers.py)."""
    d = np.abs(ll_gpu - ll_cpu)
    tol = np.maximum(1e-9, 1e-12 * np.abs(ll_cpu))
    rel = d / np.abs(ll_cpu)
    over = np.nonzero(~(d <= tol))[0]
    worst = int(np.argmax(d / tol))
    look = np.unique(np.concatenate([over[:256], [worst]])).astype(np.int64)
    _, kap = orc.eval_batch(cfg, obs, r_inv, nlay[look], layers[look], sig[look], nsmp, nthreads=nthreads,
                            want_kappa=True)
    kappa = dict(zip(look.tolist(), kap.tolist()))
    rule_ok = all(kappa[i] >= KAPPA_MIN and d[i] <= tol[i] * kappa[i] / KAPPA_SCALE for i in over[:256].tolist())
    return {"n": int(len(d)), "max_abs_dlogl": float(d.max()), "max_rel_dlogl": float(rel.max()),
            "n_over_1e-13": int(np.sum(rel > 1e-13)),
            "within_tolerance": bool(len(over) == 0),
            "n_used_kappa_allowance": int(len(over)),
            "within_kappa_rule": bool(rule_ok and len(over) <= 256),
            "worst": {"walker": worst, "nlay": int(nlay[worst]), "logl": float(ll_cpu[worst]),
                      "abs": float(d[worst]), "rel": float(rel[worst]), "tolerance_used": float(d[worst] / tol[worst]),
                      "kappa": float(kappa[worst])}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--debug-steps", action="store_true", help="stderr: the event-timed step durations of the timed region")
    ap.add_argument("--walkers", type=int, default=0, help="override walkers per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--swap", default="allgather", choices=["allgather", "p2p"],
                    help="temperature exchange of tempered workloads: K disjoint pairs via one all_gather, or the "
                         "reference's one pair per iteration via send/recv")
    ap.add_argument("--also", default=None,
                    help="comma list of extra workloads measured briefly into 'also' (default at N = 1: "
                         "c2,c2d,c3,c5,c4common,c4d,c5d,c4w20,c4w60,c4win,c5win,c4stale,c4host,c4full,c5full; '' for none; <name>host = that workload "
                         "handed over from pinned host arrays every step, the PCIe-inclusive boundary)")
    ap.add_argument("--prewarm-seconds", type=float, default=1.0,
                    help="untimed steps run for at least this long before --warmup (clock ramp; independent of --warmup)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="launch-plan option (rf_set_option); echoed in config.overrides")
    ap.add_argument("--lib", default=None, help="another build of librfgpu.so (A/B timing); echoed in config.lib")
    ap.add_argument("--rccl-library", default=None, metavar="PATH",
                    help="functional tests on a one-GPU box: librfgpu loads RCCL from this file (rf_comm_set_library; "
                         "tests/c/rccl_double.cpp) and ranks that share a GPU join its communicator; echoed in config.rccl")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not time the dominant kernel with HIP events")
    ap.add_argument("--copy-logl", action="store_true", help="always read logL back with an async copy")
    ap.add_argument("--dump-state", default=None, metavar="PATH.npz",
                    help="rank 0 writes every rank's final temperatures and logL (tests replay the swap schedule)")
    ap.add_argument("--control-plane", default="gloo", choices=["gloo", "nccl"],
                    help="N > 1: the launcher's process group (rendezvous, barrier, max of the time; NOT the temperature "
                         "exchange, which is librfgpu's RCCL communicator): gloo on the host (default), or torch's RCCL group")
    ap.add_argument("--comm-init-timeout", type=float, default=120.0, metavar="SECONDS",
                    help="N > 1: a rank whose RCCL bootstrap (rf_comm_init) takes longer exits with code 4")
    ap.add_argument("--detail-file", default=None, metavar="PATH.json",
                    help="where rank 0 writes the full record (default bench_detail.json next to this script); stdout "
                         "carries one compact line")
    ap.add_argument("--perturb-nlay", type=float, default=0.0, metavar="FRAC",
                    help="every step a fresh FRAC of the walkers evaluates its model one layer shallower (a birth / "
                         "death proposal changes nlay by one, pt_mcmc.f90:88-160): the dispatch order the previous "
                         "launch prepared is then one proposal stale, as in a real chain")
    args = ap.parse_args()

    # ---- N ranks asked for, no launcher around us: become the launcher (before torch / HIP are touched)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: the launcher's "
                         "rank count and --gpus must agree (a line saying n_gpus = N must come from N ranks)")

    # stdout carries ONE line, the last thing rank 0 says.  Libraries write there too, at the C level and at exit time
    # (gloo: "[Gloo] Rank 0 is connected to 7 peer ranks"; RCCL's version banner, flushed when the process ends): from
    # here on file descriptor 1 IS stderr, and the headline goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    # N > 1: a run that dies in the rendezvous, in librfgpu's RCCL bootstrap or in the first steps must still say so in ONE
    # JSON line on rank 0's stdout ("value": null, "error", how far it got) before the non-zero exit -- a plain exit,
    # never a re-exec.  `progress` is what the line reports.
    progress = {"stage": "start", "rccl": {"control_plane": None, "init_s": None, "transport": None, "version": None}}
    line_done = [False]

    def failure_line(error):
        if line_done[0] or int(os.environ.get("RANK", "0")) != 0:
            return
        line_done[0] = True
        rec = {"metric": METRIC, "value": None, "unit": "evals/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic", "error": str(error)[:400], "reached": progress["stage"],
               "config": {"workload": WORKLOADS[args.workload]["desc"][:200], "rccl": progress["rccl"]}}
        try:
            os.write(real_stdout, (json.dumps(rec, separators=(",", ":")) + "\n").encode())
        except OSError:
            pass

    if args.gpus > 1:
        import signal

        def _terminated(signum, frame):       # the launcher takes the ranks down when one of them failed
            failure_line(f"rank 0 received signal {signum} at stage '{progress['stage']}' (another rank failed, or the launcher timed out)")
            os._exit(6)

        signal.signal(signal.SIGTERM, _terminated)

    stray = sorted(k for k in os.environ if k.startswith("RFGPU_") and k not in ENV_ALLOWED)
    if stray:
        raise SystemExit(f"bench.py: refusing to run with {stray} in the environment: librfgpu reads no environment "
                         "variables; use --opt name=value / --lib (both are echoed in the JSON line)")
    overrides = {}
    for kv in args.opt:
        k, _, v = kv.partition("=")
        overrides[k.strip()] = float(v)

    import torch

    from rf_inv_amd import _lib

    _lib.load(args.lib)
    lib_sha = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
    kernels_sha = _lib.kernels_sha256(_lib.LIB_PATH)     # the gfx950 code objects alone (.hip_fatbin)
    if args.rccl_library:
        from rf_inv_amd import RFEngine as _E

        _E.comm_set_library(args.rccl_library)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Two planes.  DATA: the temperature exchange of every step runs over librfgpu's own RCCL communicator (xGMI), one
    # GPU per rank.  CONTROL: rendezvous, the 128-byte RCCL id, barrier and max-over-ranks of the time go over the
    # launcher's process group -- gloo by default (host side, TCP on the loopback interface): the device then carries ONE
    # communicator, librfgpu's (`--control-plane nccl` puts torch's RCCL group there as well).
    # RFGPU_BENCH_BACKEND=gloo (functional tests on a one-GPU box): the ranks may share a GPU.
    shared = os.environ.get("RFGPU_BENCH_BACKEND", "nccl") != "nccl"
    backend = "gloo" if shared else args.control_plane
    if world > 1:
        import datetime

        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: never resolve the container's host name
        progress["stage"], progress["rccl"]["control_plane"] = "process_group_init", backend
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                        timeout=datetime.timedelta(seconds=300))
            else:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
        except Exception as e:      # noqa: BLE001
            failure_line(f"process group ({backend}) rendezvous failed: {type(e).__name__}: {e}")
            raise
        progress["stage"] = "process_group_formed"
    if not torch.cuda.is_available():
        failure_line("no GPU visible: the hot path has no CPU fallback")
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if not shared and world > torch.cuda.device_count():
        failure_line(f"{world} ranks but {torch.cuda.device_count()} visible GPU(s): one GPU per rank")
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPU(s): one GPU per rank")
    if shared:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    comm_boot_s = [None]

    def run(workload, steps, warmup, with_cpu, parity_n=64):
        from rf_inv_amd import RFEngine, format_model, read_ref_model
        from rf_inv_amd.likelihood import init_r_inv
        from rf_inv_amd.pt import PTSwap, open_exchange

        w = dict(WORKLOADS[workload])
        if args.walkers:
            w["walkers"] = args.walkers
        p = make_params(w)
        ref = read_ref_model(os.path.join(ROOT, "tests", "golden", "sample_syn", "model", "sample.velmod"))
        nb = w["walkers"]
        nlay, layers = draw_walkers(p, ref, rank * nb, nb)
        sig = np.full((nb, p.ntrc), 0.01)
        r_inv = init_r_inv(p.nsmp, p.a_gus, p.delta)
        # observed traces: noise-free synthetic of a fixed 3-interface model, produced by the HIP path itself
        zt = np.zeros(max(p.k_max - 1, 1)); dvt = np.zeros(p.k_max); dst = np.zeros(p.k_max)
        zt[:3] = [3.1 + p.sdep, 7.7 + p.sdep, 14.2 + p.sdep]; dst[:3] = [-0.6, 0.2, 0.5]; dst[p.k_max - 1] = 0.9
        nl_t, a_t, b_t, r_t, h_t, ok = format_model(p, ref, 3, zt, dvt, dst)
        assert ok
        kw = dict(nfft=p.nfft, delta=p.delta, t_start=p.t_start, deconv_mode=p.deconv_mode, sdep=p.sdep,
                  rayps=p.rayps, a_gus=p.a_gus, ipha=p.ipha, nsmp=p.nsmp, nlay_max=p.k_max + 2,
                  device=local_rank)
        with RFEngine(obs=np.zeros((p.ntrc, p.nsmp)), r_inv=r_inv, max_walkers=1, **kw) as e0:
            obs = np.ascontiguousarray(e0.calc_rf(nl_t, a_t, b_t, r_t, h_t)[:p.nsmp].T)
        eng = RFEngine(obs=obs, r_inv=r_inv, max_walkers=nb, options={**w.get("options", {}), **overrides}, **kw)

        stream = torch.cuda.Stream(device=dev)
        d_ids = torch.arange(nb, dtype=torch.int32, device=dev)
        d_nlay = torch.from_numpy(nlay).to(dev)
        d_layers = torch.from_numpy(layers).to(dev)
        d_sig = torch.from_numpy(sig).to(dev)
        d_logl = torch.empty(nb, dtype=torch.float64, device=dev)
        h_logl = torch.empty(nb, dtype=torch.float64).pin_memory()
        # N > 1: the temperature exchange runs over librfgpu's own RCCL communicator (one GPU per rank); ranks
        # that share a GPU (functional test) keep the launcher's process group as the transport
        over_rccl = False
        if world > 1:
            # ncclCommInitRank is collective and cannot be interrupted: a rank stuck in it for --comm-init-timeout
            # seconds says so and leaves (exit code 4; the launcher takes the other ranks down) -- never a silent hang
            import threading

            def _stuck():
                print(f"bench.py: rank {rank}: librfgpu's RCCL bootstrap (rf_comm_init, {world} ranks) did not finish "
                      f"within {args.comm_init_timeout:.0f} s: giving up (exit code 4)", file=sys.stderr, flush=True)
                failure_line(f"rf_comm_init ({world} ranks) did not finish within {args.comm_init_timeout:.0f} s")
                os._exit(4)

            dog = threading.Timer(args.comm_init_timeout, _stuck)
            dog.daemon = True
            dog.start()
            t_boot = time.perf_counter()
            progress["stage"] = "rf_comm_init"
            over_rccl = open_exchange(eng, dist, shared_gpu_ok=bool(args.rccl_library))
            dog.cancel()
            comm_boot_s[0] = time.perf_counter() - t_boot
            progress["stage"] = "communicator_formed"
            progress["rccl"].update(init_s=comm_boot_s[0], transport="rccl_allgather" if over_rccl else "process_group",
                                    version=eng.comm_info()["rccl_version"])
        swap = (PTSwap(eng, nb, w["temps"], dev, seed=1234, t_high=15.0, mode=args.swap, rccl=over_rccl)
                if w["temps"] > 1 else None)
        # --perturb-nlay: NV pre-built depth vectors cycled through, so that the timed loop does nothing extra
        nlay_var = None
        if args.perturb_nlay > 0.0:
            g = np.random.Generator(np.random.Philox(key=4242 + rank))
            nv = []
            for _ in range(16):
                pick = (g.random(nb) < args.perturb_nlay) & (nlay >= 3 + (1 if p.sdep > 0 else 0))
                nv.append(torch.from_numpy((nlay - pick.astype(np.int32)).astype(np.int32)).to(dev))
            nlay_var = nv
        step_no = [0]

        # logL read-back: without a swap step the kernel writes logL straight into the pinned
        # (device-mapped) host buffer -- no copy kernel after the evaluation; with a swap step logL
        # is consumed on the device first and copied afterwards
        zero_copy = swap is None and not args.copy_logl

        def step():
            nl_now = d_nlay
            if nlay_var is not None:
                nl_now = nlay_var[step_no[0] % len(nlay_var)]
                step_no[0] += 1
            with torch.cuda.stream(stream):
                if zero_copy:
                    eng.eval_batch_device(d_ids, nl_now, d_layers, d_sig, h_logl, stream=stream)
                else:
                    eng.eval_batch_device(d_ids, nl_now, d_layers, d_sig, d_logl, stream=stream)
                    if swap is not None:
                        swap.step(d_logl, stream)
                    h_logl.copy_(d_logl, non_blocking=True)

        # ---- pre-warm by TIME (clock ramp, allocator, dispatch order of the running batch), then --warmup steps.
        # Nothing that leaves the GPU idle for milliseconds may sit between the last warm-up step and the timed
        # region -- the clocks drop within a few ms of idleness and take tens of ms of work to come back (a
        # gc.collect() there cost the first 20 timed steps 15 %): the collector is run and disabled, and the event
        # machinery switched on, BEFORE the pre-warm; only the barrier separates warm-up and timed steps.
        import gc

        gc.collect()
        gc.disable()            # no collector pause inside the (possibly short) timed region
        # HIP events (library side, on the launch stream) around the dominant kernel, and torch events on the same
        # stream around whole steps, both on every `every`-th step of the timed region: an event pair costs ~4 us of
        # stream time -- at C2 (0.09 ms steps) timing every step would slow the loop it measures by ~8 %
        every = 1 if steps <= 16 else min(8, max(1, steps // 8))
        marks = []
        def timed_step(i, marks):
            if i % every == 0:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                step()
                e1.record(stream)
                marks.append((e0, e1))
            else:
                step()

        # (the pre-warm and the warm-up run the very code of the timed loop, events included: on a fresh box the first
        # pass through any code path pages it in from the image -- a 10 ms stall if that happens inside the timed
        # region.  The library-side kernel events are exercised once, flushed -- a synchronisation -- and switched on
        # again, which costs nothing, right before the timed region.)
        eng.profile_enable(0 if args.no_kernel_events else every)
        scratch = []
        first = None
        if world > 1:
            # the FIRST steps across ranks (the first collectives of the exchange on the device): a rank whose stream has not
            # drained after --comm-init-timeout seconds says so and leaves (exit code 4) instead of hanging the job
            import threading

            def _stuck_step():
                print(f"bench.py: rank {rank}: the first steps with the temperature exchange did not finish within "
                      f"{args.comm_init_timeout:.0f} s (transport: {'librfgpu RCCL' if over_rccl else 'process group'}): giving up "
                      "(exit code 4)", file=sys.stderr, flush=True)
                failure_line(f"the first steps with the temperature exchange did not finish within {args.comm_init_timeout:.0f} s")
                os._exit(4)

            first = threading.Timer(args.comm_init_timeout, _stuck_step)
            first.daemon = True
            first.start()
        if world > 1:
            progress["stage"] = "first_steps"
        for i in range(8):
            timed_step(i, scratch)
        torch.cuda.synchronize(dev)
        if first is not None:
            first.cancel()
            progress["stage"] = "steps_running"
        eng.profile_enable(False)
        eng.profile_read()
        t_pre = time.perf_counter()
        n_pre = 8
        while True:
            scratch = []
            for i in range(8):
                timed_step(i, scratch)
            n_pre += 8
            torch.cuda.synchronize(dev)
            _ = [a.elapsed_time(b) for a, b in scratch]
            more = time.perf_counter() - t_pre < args.prewarm_seconds
            if world > 1:
                # every rank must run the SAME number of steps (each carries a collective of the temperature exchange):
                # the ranks go on while any of them wants to
                flag = torch.tensor([1 if more else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                more = int(flag.item()) != 0
            if not more:
                break
        scratch = []
        for i in range(warmup):
            timed_step(i, scratch)
        barrier()
        eng.profile_enable(0 if args.no_kernel_events else every)
        t0 = time.perf_counter()
        for i in range(steps):
            timed_step(i, marks)
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        eng.profile_enable(False)
        prof = eng.profile_read()
        step_ms = np.array([a.elapsed_time(b) for a, b in marks])
        if args.debug_steps:
            print("step_ms", workload, np.round(step_ms, 3).tolist(), "dt", dt, file=sys.stderr)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        ll_gpu = h_logl.numpy().copy()
        # A multi-rank run checks its own temperature exchange: every rank's final temperatures and logL are gathered
        # (after the timed region) and rank 0 replays the replicated swap schedule serially
        # (rf_inv_amd.pt.replay_swap_schedule) -- a wrong permutation of temperatures must not print a clean line
        swap_check = None
        if (world > 1 or args.dump_state) and workload == args.workload:
            t_loc = swap.temps.clone() if swap is not None else torch.ones(nb, dtype=torch.float64, device=dev)
            l_loc = torch.from_numpy(ll_gpu).to(dev)
            if world > 1:
                on_host = backend != "nccl"
                src_t, src_l = (t_loc.cpu(), l_loc.cpu()) if on_host else (t_loc, l_loc)
                all_t = [torch.empty_like(src_t) for _ in range(world)]
                all_l = [torch.empty_like(src_l) for _ in range(world)]
                dist.all_gather(all_t, src_t)
                dist.all_gather(all_l, src_l)
            else:
                all_t, all_l = [t_loc], [l_loc]
            if rank == 0:
                g_t = np.stack([t.cpu().numpy() for t in all_t])
                g_l = np.stack([t.cpu().numpy() for t in all_l])
                if args.dump_state:
                    np.savez(args.dump_state, temps=g_t, logl=g_l, swap_steps=n_pre + warmup + steps,
                             pairs_per_step=swap.k if swap is not None else 0)
                if swap is not None and args.swap == "allgather" and nlay_var is None:
                    from rf_inv_amd.pt import replay_swap_schedule

                    swap_check = replay_swap_schedule(g_t, g_l, nb, w["temps"], n_pre + warmup + steps, swap.k, 1234, 15.0)
                    swap_check["swap_steps"] = n_pre + warmup + steps
        if world > 1 and over_rccl:
            info = eng.comm_info()
            assert info["nranks"] == world, (info, world)
            if rank == 0:
                print(f"bench.py: temperature exchange over librfgpu's RCCL communicator, {info['nranks']} ranks, RCCL "
                      f"{info['rccl_version']}", file=sys.stderr)
        elif world > 1 and rank == 0:
            print("bench.py: temperature exchange over the launcher's process group (ranks share a GPU, or RCCL's "
                  "bootstrap failed)", file=sys.stderr)
        plan = eng.launch_plan
        assert plan["build"] == "production" or overrides or args.lib, plan
        assert plan["trace_window"] == bool(w.get("options", {}).get("trace_window", overrides.get("trace_window", 0)))
        assert np.all(np.isfinite(ll_gpu)), "non-finite logL in the benchmark batch"

        # the depths the LAST step evaluated (they are what h_logl holds and what the checker must be given)
        nlay_eval = nlay if nlay_var is None else nlay_var[(step_no[0] - 1) % len(nlay_var)].cpu().numpy()
        f_spec, f_tot, b_alg = alg_work(p, nlay_eval.astype(np.float64), eng.is_ray_common)
        n_l = max(prof["launches"], 1)          # batches timed
        # dominant kernel: fused_kernel (propagator + trace + logL in one launch) where every trace has
        # its own forward computation, else spectra_kernel (then trace_kernel follows it)
        kernel_ms = prof["spectra_ms"] / n_l if prof["launches"] else None
        f_dom = f_tot if plan["fused"] else f_spec
        bt = plan["block_threads_full_batch"]      # 512: fused8_kernel (contexts of up to two rounds of blocks)
        kname = ("rfgpu::fusedc_kernel" if plan["common_ray_fused"] else
                 "rfgpu::fused8_kernel" if bt == 512 else "rfgpu::fused_kernel") if plan["fused"] else "rfgpu::spectra_kernel"
        if plan["fused"]:
            nblk = nb if plan["common_ray_fused"] else nb * p.ntrc       # common rays: one block per walker
            grid_threads = bt * (nblk + (1 if (plan["lpt"] and plan["order_reuse"] and nb >= 512) else 0))
        else:
            grid_threads = None   # split path: matched by name only (nsplit decides the grid)
        ctr = (committed_counters(kname, grid_threads, long_window=plan["long_window_gemm"], decon=p.deconv_mode == 1)
               if grid_threads else None)
        exe = executed_fp64_flops(ctr) if ctr else None
        t_k = kernel_ms * 1e-3 if kernel_ms else None
        # counters describe kernels: they are this build's if its gfx950 code objects (.hip_fatbin) are the ones they
        # were collected on -- a host-only change of the library keeps them -- or, for files without that hash, if the
        # whole library is the same file
        counters_fresh = bool(ctr) and ((bool(ctr.get("_kernels_sha256")) and ctr["_kernels_sha256"] == kernels_sha)
                                        or ctr["_lib_sha256"] == lib_sha)
        roof = {
            "bound": "fp64_valu", "unit": "TFLOP/s", "peak": FP64_PEAK_TFLOPS,
            # EXECUTED fp64 flops of one launch (committed SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 counters of this
            # kernel at this launch shape) / live HIP-event kernel time: a hardware fraction, <= 1
            # (counters collected on ANOTHER build of the library say nothing about this one: no fraction then)
            "achieved": exe / t_k / 1e12 if exe and t_k and counters_fresh else None,
            "frac": exe / t_k / 1e12 / FP64_PEAK_TFLOPS if exe and t_k and counters_fresh else None,
            "frac_with_stale_counters": (exe / t_k / 1e12 / FP64_PEAK_TFLOPS
                                         if exe and t_k and not counters_fresh else None),
            # (the windowed-trace workloads launch the same kernel at the same grid as their full-trace twins but write
            # 40x fewer trace bytes: the committed counters are the twins', so no traffic figure for them)
            "traffic": (2048.0 * ctr["FETCH_SIZE"] + 1024.0 * ctr["WRITE_SIZE"]) if ctr and "FETCH_SIZE" in ctr
                       and "WRITE_SIZE" in ctr and not plan["trace_window"] else None,
            "kernel": kname, "kernel_ms": kernel_ms, "grid_threads": grid_threads,
            "executed_gflop_per_launch": exe / 1e9 if exe else None,
            # shader clock this kernel ran at in the counter pass (GRBM_GUI_ACTIVE / 8 XCDs / its duration,
            # tools/summarize_counters.py): the part lowers it below the 2.4 GHz `peak` is quoted at when the fp64
            # vector ALUs are the load (HBM-bound kernels of the same pass run at 2.4).  `frac_at_clock` = frac
            # priced at that clock -- `frac` stays the spec-clock figure.
            # the same at STEP level: the fp64 flops every kernel of a step executes / ms_per_step.  The dominant kernel's
            # from its counters; the follow-up kernel's quadratic forms as quad_flops_run + 2 nsmp per (walker, trace) -- what
            # phi_deferred_kernel / phi_gemm_kernel execute (the latter on the form's upper triangle) --; stage_kernel, the swap and the order kernels
            # execute < 0.2 % of a step's flops and are left out (a lower bound by that much)
            "frac_step": ((exe + nb * p.ntrc * (quad_flops_run(p.nsmp, plan) + 2.0 * p.nsmp) * (1.0 if prof["logl_launches"] else 0.0))
                          / (dt / steps) / 1e12 / FP64_PEAK_TFLOPS if exe and counters_fresh else None),
            "clock_ghz": ctr.get("_clock_ghz") if ctr else None,
            "frac_at_clock": (exe / t_k / 1e12 / (FP64_PEAK_TFLOPS * ctr["_clock_ghz"] / SPEC_CLOCK_GHZ)
                              if exe and t_k and ctr.get("_clock_ghz") and counters_fresh else None),
            "counters": ({"file": ctr["_file"], "lib_sha256_matches_this_build": ctr["_lib_sha256"] == lib_sha,
                          "kernels_sha256_matches_this_build": bool(ctr.get("_kernels_sha256")) and ctr["_kernels_sha256"] == kernels_sha}
                         if ctr else None),
            # SURVEY.md 8d's ALGORITHMIC figure (reference arithmetic: 570 flop/(bin*layer) + 580/bin, + FFT /
            # shift / quadratic form when fused) over the same kernel time.  The eigen-coordinate real-form
            # propagator executes ~1/5 of those flops, so this ratio exceeds 1; it is not a hardware fraction.
            "algorithmic": {"gflop_per_launch": float(f_dom.sum()) / 1e9,
                            "tflops": float(f_dom.sum()) / t_k / 1e12 if t_k else None,
                            "ratio_to_peak": float(f_dom.sum()) / t_k / 1e12 / FP64_PEAK_TFLOPS if t_k else None},
            "note": "fp64 vector ALU roofline: MI355X FP64 matrix (MFMA) peak == vector peak = 78.6 TF, the kernel "
                    "issues fp64 VALU FMA/MUL/ADD, MFMA not used.",
        }
        res = {
            "value": world * nb * steps / dt,
            "ms_per_step": 1e3 * dt / steps,
            "steps": steps,
            "ms_per_step_median": float(np.median(step_ms)) if len(step_ms) else None,
            "ms_per_step_p10_p90": [float(np.percentile(step_ms, 10)), float(np.percentile(step_ms, 90))]
                                   if len(step_ms) else None,
            "config": {"workload": w["desc"], "walkers_per_gpu": nb, "nfft": p.nfft, "ntrc": p.ntrc,
                       "nsmp": p.nsmp, "k_max": p.k_max, "mean_nlay": float(nlay.mean()),
                       "max_nlay": int(nlay.max()), "deconv_mode": p.deconv_mode, "sdep": p.sdep,
                       "ray_common": bool(eng.is_ray_common),
                       "logl_readback": "kernel writes pinned host memory" if zero_copy else "device buffer + async copy",
                       "temperatures": w["temps"], "parallelism": f"walkers sharded x{world}",
                       "pt_swap": (f"{args.swap}, {swap.k} pair(s)/step" if swap is not None else "none"),
                       # transport: "rccl_allgather" = librfgpu's RCCL group, 2 x ncclAllGather + 1 kernel per step
                       # (rf_pt_swap_allgather_device); "process_group" = the launcher's process group gathers,
                       # rf_pt_swap_gathered_device judges (librfgpu's communicator not formed: ranks share a GPU, or its
                       # bootstrap failed); "none" = one rank
                       "rccl": ({"ranks": eng.comm_info()["nranks"], "version": eng.comm_info()["rccl_version"],
                                 "transport": "rccl_allgather", "control_plane": backend, "init_s": comm_boot_s[0],
                                 **({"library": args.rccl_library} if args.rccl_library else {})} if over_rccl else
                                {"ranks": 0, "version": eng.comm_info()["rccl_version"],
                                 "transport": "none" if world == 1 else "process_group",
                                 **({"control_plane": backend} if world > 1 else {})}),
                       "perturb_nlay": args.perturb_nlay, "workload_options": w.get("options", {}),
                       "launch_plan": plan, "overrides": overrides,
                       "lib": {"path": os.path.relpath(_lib.LIB_PATH, ROOT), "sha256": lib_sha, "kernels_sha256": kernels_sha,
                               "default_build": args.lib is None},
                       "prewarm": {"seconds": args.prewarm_seconds, "steps": n_pre},
                       "events_every": every},
            "roofline": roof,
            "roofline_hbm": {
                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "achieved": float(b_alg.sum()) * steps / dt / 1e9,
                "frac": float(b_alg.sum()) * steps / dt / 1e9 / HBM_PEAK_GBS,
                "note": "algorithmic bytes/eval (layers+sigma in, prop_rft(nfft,ntrc)+logL out) x evals/s of this rank; "
                        "the path is FP64-ALU bound, not HBM bound",
            },
            # HIP-event times per step of the kernels of the batch: the fused kernel (or spectra + trace), and the
            # follow-up that forms the quadratic forms + logL when the trace kernel leaves its misfits in HBM
            # (phi_deferred_kernel, or on the long-window plan phi_gemm_kernel + phi_gemm_finish_kernel)
            "kernel_ms": dict(({"fused": kernel_ms} if plan["fused"] else
                               {"spectra": kernel_ms, "trace": prof["trace_ms"] / n_l}),
                              **({"quadratic_form_logl": prof["logl_ms"] / n_l} if prof["logl_launches"] else {})),
            "alg_gflop_per_step": float(f_tot.sum()) / 1e9,
            "alg_bytes_per_step": float(b_alg.sum()),
        }
        if swap_check is not None:
            res["swap_replay_ok"] = swap_check["ok"]
            res["cross_rank_swaps"] = swap_check["cross_rank_swaps"]
            res["swap_replay"] = swap_check
        if plan["long_window_gemm"] and prof["logl_launches"]:
            # the long-window plan's GEMM: Phi1 = M_t R^-1_t per trace on the FP64 matrix cores, 2 nb ntrc nsmp^2
            # algorithmic flops (src/likelihood.f90:92: matmul(misfits, r_inv)); the time includes the logL kernel
            gf = 2.0 * nb * p.ntrc * float(p.nsmp) ** 2
            t_q = prof["logl_ms"] / n_l * 1e-3
            # committed counters of the GEMM at this launch shape (128 x 64 blocks of 256 threads): MFMA wave-instructions
            # (2048 flop each), the matrix pipe's busy cycles (64 per instruction) / (kernel time x its clock x 1024 SIMDs)
            kp = (p.nsmp + 15) // 16 * 16
            g_thr = 256 * ((nb + 127) // 128) * ((kp + 63) // 64) * p.ntrc
            gc = committed_counters("rfgpu::phi_gemm_kernel", g_thr, long_window=True)
            g_fresh = bool(gc) and bool(gc.get("_kernels_sha256")) and gc["_kernels_sha256"] == kernels_sha
            mf = None
            if gc and g_fresh and gc.get("SQ_INSTS_MFMA"):
                mf = {"mfma_wave_instructions": gc["SQ_INSTS_MFMA"], "executed_gflop": 2048.0 * gc["SQ_INSTS_MFMA"] / 1e9,
                      "busy_cycles_per_instruction": gc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / gc["SQ_INSTS_MFMA"],
                      "clock_ghz": gc.get("_clock_ghz"),
                      "pipe_busy_frac_in_counter_pass": (gc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 /
                                                         (gc["_clock_ghz"] * 1e9 * gc["_mean_s"])
                                                         if gc.get("_clock_ghz") and gc.get("_mean_s") else None),
                      "hbm_mb": ((2048.0 * gc["FETCH_SIZE"] + 1024.0 * gc["WRITE_SIZE"]) / 1e6
                                 if "FETCH_SIZE" in gc and "WRITE_SIZE" in gc else None),
                      "l2_hit_rate": (gc["TCC_HIT_sum"] / (gc["TCC_HIT_sum"] + gc["TCC_MISS_sum"])
                                      if gc.get("TCC_HIT_sum") is not None and gc.get("TCC_MISS_sum") else None),
                      "file": gc["_file"]}
            # what the kernel multiplies: with the triangular image ("gemm_triangle", default) column chunk c of 64 runs
            # rows 0 .. 64 (c + 1) - 1 only; the full product runs all kp rows for every chunk
            tri = bool(plan.get("gemm_triangle"))
            gf_run = nb * p.ntrc * quad_flops_run(p.nsmp, plan)
            res["quadratic_form_gemm"] = {
                "kernel": "rfgpu::phi_gemm_kernel + phi_gemm_finish_kernel", "bound": "fp64_mfma", "unit": "TFLOP/s",
                "peak": FP64_PEAK_TFLOPS, "triangle": tri, "gflop_per_launch": gf_run / 1e9, "ms": 1e3 * t_q,
                "achieved": gf_run / t_q / 1e12, "frac": gf_run / t_q / 1e12 / FP64_PEAK_TFLOPS, "grid_threads": g_thr,
                "algorithmic": {"gflop_per_launch": gf / 1e9, "tflops": gf / t_q / 1e12,
                                "ratio_to_peak": gf / t_q / 1e12 / FP64_PEAK_TFLOPS},
                "mfma": mf,
                "note": "achieved / frac: the multiply-adds the kernel runs (triangle: 2 nb ntrc 64 sum_c min(kp, 64 (c + 1)), "
                        "about half of the reference's matmul(misfits, r_inv) = 2 nb ntrc nsmp^2, which is `algorithmic`) over "
                        "the HIP-event time of the GEMM + logL kernels; the FP64 matrix peak equals the vector peak (78.6 TF at "
                        "2.4 GHz: one 16x16x4 instruction per 64 cycles and SIMD); R^-1 is streamed once per 128 walkers"}
        if rank == 0 and (with_cpu or parity_n):
          try:
            from oracle import rf_oracle as orc

            cfg = dict(nfft=p.nfft, deconv_mode=p.deconv_mode, delta=p.delta, t_start=p.t_start, sdep=p.sdep,
                       rayps=p.rayps, a_gus=p.a_gus, ipha=p.ipha)
            nthr = max(1, min(physical_cores(), orc.max_threads()))
            if with_cpu:
                base, ll_cpu, n = cpu_baseline(p, obs, r_inv, nlay_eval, layers, sig)
                res["cpu_baseline"] = base
            else:
                orc.build()
                n = min(nb, parity_n)
                ll_cpu = None
            # the compared walkers: the first n, and (side workloads) the n // 4 highest walker ids of the rank as well
            pidx = np.arange(n) if with_cpu or n == nb else np.unique(np.concatenate([np.arange(n - n // 4), np.arange(nb - n // 4, nb)]))
            if ll_cpu is None:
                ll_cpu = orc.eval_batch(cfg, obs, r_inv, nlay_eval[pidx], layers[pidx], sig[pidx], p.nsmp, nthreads=nthr)   # the checker build
            res["parity_in_bench"] = parity_report(orc, cfg, obs, r_inv, nlay_eval[pidx], layers[pidx], sig[pidx], p.nsmp,
                                                   ll_gpu[pidx], ll_cpu, nthr)
            res["parity_in_bench"]["walker_id_range"] = [int(pidx.min()), int(pidx.max())]
          except Exception as e:      # noqa: BLE001  (the CPU side failing must not take the measured line with it)
            print(f"bench.py: the CPU leg (cpu_baseline / parity_in_bench) of {workload} failed: {type(e).__name__}: {e}",
                  file=sys.stderr)
            res["cpu_leg_error"] = f"{type(e).__name__}: {e}"[:300]
        eng.close()
        return res

    try:
        main_res = run(args.workload, args.steps, args.warmup, not args.no_cpu_baseline and world == 1)
    except BaseException as e:      # noqa: BLE001  (N > 1: the failure still gets its JSON line; then the error as it was)
        if world > 1 and not isinstance(e, KeyboardInterrupt):
            failure_line(f"{type(e).__name__}: {e}")
        raise
    progress["stage"] = "measured"
    also_list = args.also if args.also is not None else ("c2,c2d,c3,c5,c4common,c4d,c5d,c4w20,c4w60,c4win,c5win,c4stale,c4host,c4full,c5full" if world == 1 else "")
    also = {}
    keep = ("value", "ms_per_step", "ms_per_step_median", "steps", "config", "roofline", "kernel_ms", "parity_in_bench",
            "quadratic_form_gemm")
    def also_run(wl):
        if wl.endswith("host"):
            # the PCIe-inclusive boundary of the batched sampler (never the headline)
            return run_host_boundary(wl[:-4], 200 if args.steps >= 10 else max(20, args.steps), local_rank,
                                     prewarm_s=args.prewarm_seconds)
        if wl == "c4stale":
            # the c4 shape with 30 % of the walkers changing depth every step: the order the previous launch
            # prepared is one proposal stale (order_reuse, the default), against a fresh order_kernel per launch
            args.perturb_nlay, sv = 0.3, dict(overrides)
            n_c4 = 200 if args.steps >= 10 else max(30, min(200, args.steps))
            r = run("c4", n_c4, max(5, min(20, args.warmup)), False)
            overrides["order_reuse"] = 0.0
            r2 = run("c4", n_c4, max(5, min(20, args.warmup)), False, parity_n=0)
            overrides.clear(); overrides.update(sv)
            args.perturb_nlay = 0.0
            rec = {k: r[k] for k in keep if k in r}
            rec["fresh_order_every_launch"] = {"value": r2["value"], "ms_per_step": r2["ms_per_step"],
                                               "kernel_ms": r2["kernel_ms"]}
            return rec
        # long enough that one host hiccup does not show: about 0.4 s of steps for the small shapes
        n_also = max(30, min(200, args.steps))
        if wl.endswith("full"):      # the whole 8-GPU job on one GPU: 15 / 100 ms per step
            n_also = min(n_also, 60)
        if args.steps >= 10:     # (the counter passes of tools/collect_counters.sh ask for 3 steps and get them)
            n_also = max(n_also, min(6000, int(400.0 / ALSO_NOMINAL_MS.get(wl, 2.0))))
        r = run(wl, n_also, max(5, min(20, args.warmup)), False)
        return {k: r[k] for k in keep if k in r}

    # (a side workload that fails must not take the headline with it: its record says what happened, the line is printed)
    for wl in [x for x in also_list.split(",") if x and x != args.workload]:
        try:
            also[wl] = also_run(wl)
        except Exception as e:      # noqa: BLE001
            print(f"bench.py: side workload {wl} failed: {type(e).__name__}: {e}", file=sys.stderr)
            also[wl] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
            args.perturb_nlay = 0.0
    if rank == 0:
        out = {"metric": METRIC, "value": main_res["value"], "unit": "evals/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic"}
        out.update({k: v for k, v in main_res.items() if k not in ("value", "ms_per_step")})
        if also:
            out["also"] = also
        # the full record (launch plan, every `also` workload with its roofline, notes) goes to a side file and to
        # stderr; stdout gets ONE compact line (< 4 KB), the last thing this process prints there
        detail = args.detail_file or os.path.join(ROOT, "bench_detail.json")
        try:
            with open(detail, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:
            print(f"bench.py: could not write {detail}: {e}", file=sys.stderr)
            detail = None
        print("bench.py detail: " + json.dumps(out), file=sys.stderr)
        sys.stderr.flush()
        sys.stdout.flush()
        try:
            line = headline_line(out, os.path.relpath(detail, ROOT) if detail else None)
        except Exception as e:      # noqa: BLE001  (never lose a measured number to the formatter)
            print(f"bench.py: headline formatter failed ({type(e).__name__}: {e}): printing the contract keys only", file=sys.stderr)
            line = json.dumps({k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                       "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
                              | {"config": {"workload": (out.get("config") or {}).get("workload")}})
        os.write(real_stdout, (line + "\n").encode())
        line_done[0] = True
    # parity of the headline workload: the line is printed either way; a run whose checker leg died, or whose walkers are
    # outside both the tolerance and the conditioning rule, ends with exit code 5
    par = main_res.get("parity_in_bench")
    parity_failed = rank == 0 and world == 1 and (
        "cpu_leg_error" in main_res or (par is not None and not (par["within_tolerance"] or par["within_kappa_rule"])))
    if parity_failed:
        print("bench.py: parity of the headline workload is NOT established (cpu_leg_error, or walkers outside the tolerance "
              "and the conditioning rule): exit code 5 -- the JSON line above carries the details", file=sys.stderr)
    failed = rank == 0 and main_res.get("swap_replay_ok") is False
    if failed:
        print("bench.py: the final temperatures do NOT equal the serial replay of the swap schedule: the temperature "
              "exchange is wrong (the JSON line above carries swap_replay_ok = false)", file=sys.stderr)
    if world > 1:
        dist.destroy_process_group()
    if failed:
        raise SystemExit(3)
    if parity_failed:
        raise SystemExit(5)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--draw-worker":
        _draw_worker(sys.argv[2])
    else:
        main()
