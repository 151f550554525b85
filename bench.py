#!/usr/bin/env python3
"""bench.py -- forward+likelihood evaluations/s of the HIP hot path on N MI355X.

One "step" = one batched pass of the hot path (rf_eval_batch_device: spectra ->
trace -> logL) over every walker resident on the rank, inputs already in HBM, logL
read back to pinned host memory.  Walkers shard across ranks with no data-path
collective (weak scaling: per-GPU work fixed); workloads with tempered chains add the
parallel-tempering swap exchange (one all_gather of (T, logL) per step over RCCL).

Workloads (BASELINE.json configs):
  c2  (default) 1024 walkers/GPU, 1 P trace (p 0.06), nfft 4096, k_max 15 (<= 15 layers)
  c4            8192 walkers/GPU, 3 traces (P .06, P .08, S .10), k_max 30, PT swap
  c5            8192 walkers/GPU, 4 traces (2 P + 2 S), ocean layer, k_max 30, PT swap
  c1            sample_syn shape: nfft 256, 2 traces, ocean, k_max 10 (plumbing size)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector == matrix peak (datasheet)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec

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
               desc="c4: 8192 walkers/GPU (1024 chains x 8 temperatures) x 3 traces (P .06, P .08, S .10) x nfft 4096 "
                    "x <=30 layers, PT swap"),
    "c5": dict(walkers=8192, nfft=4096, rayps=[0.06, 0.08, 0.10, 0.12], ipha=[1, 1, -1, -1], k_max=30, sdep=2.0,
               deconv=0, temps=16,
               desc="c5-shape: 8192 walkers/GPU x 4 traces (P .06, P .08, S .10, S .12) x nfft 4096 x ocean layer "
                    "(sdep 2 km) x <=31 layers, PT swap (BASELINE's 'buried station' has no reference behaviour: "
                    "the borehole branch of forward.f90:289-338 is commented out)"),
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
    p.a_gus = np.full(n, 4.0)
    p.sig_min = np.full(n, 0.01); p.sig_max = np.full(n, 0.01); p.sig_mode = np.zeros(n, dtype=np.int32)
    p.k_min, p.k_max, p.sdep, p.deconv_mode = 1, w["k_max"], w["sdep"], w["deconv"]
    p.delta = float(np.float32(0.05))
    p.t_start, p.t_end, p.nsmp = 0.0, 5.0, 101
    return p


def draw_walkers(p, ref, first_id, count, seed=12345678):
    """Walker models from the init_model prior (reference src/model.f90:66-95) conditioned on
    validity, with k uniform in [k_min, k_max), from a counter-based RNG
    keyed by seed + global walker id.  init_model rejects whole models, which makes deep
    stacks vanishingly rare (P(valid) ~ 0.68^nlay at dVs sigma 2.0); here k is drawn once and
    the rejection is applied per component (interface set until the thickness rules hold,
    each dVs until its layer passes the range rules) -- the same distribution as whole-model
    rejection given k, since the validity rules factorise that way for vp_mode = 0."""
    from rf_inv_amd import format_model

    pad = p.k_max + 2
    layers = np.ones((count, 4, pad))
    nlay = np.zeros(count, dtype=np.int32)
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
        while True:
            zs = np.sort(p.z_min + g.random(k) * (p.z_max - p.z_min))
            th = np.diff(np.concatenate([[p.sdep], zs]))
            if th[0] >= 0.125 * vp0 and np.all(th[1:] >= p.h_min):
                break
        z = np.zeros(max(p.k_max - 1, 1)); dvp = np.zeros(p.k_max); dvs = np.zeros(p.k_max)
        z[:k] = g.permutation(zs)
        dvs[:k] = [draw_dvs(g) for _ in range(k)]
        dvs[p.k_max - 1] = draw_dvs(g)
        nl, a, b, r, h, ok = format_model(p, ref, k, z, dvp, dvs)
        assert ok
        nlay[i] = nl
        layers[i, 0, :nl], layers[i, 1, :nl], layers[i, 2, :nl], layers[i, 3, :nl] = a, b, r, h
    return nlay, layers


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


def measured_traffic(kernel, workload, nb):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/<round>_<workload>_hbm_traffic.json, produced by tools/profile_gpu.sh: FETCH_SIZE and
    WRITE_SIZE in separate runs, FETCH doubled per the gfx950 correction).  None when no profile of
    this workload / walker count is committed -- bench.py itself cannot collect PMC counters."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("_walkers_per_gpu") not in (None, nb):
            continue
        for k, v in d.items():
            if isinstance(v, dict) and k.split("<")[0] == kernel:
                return v.get("hbm_bytes_per_launch")
    return None


def cpu_baseline(p, obs, r_inv, nlay, layers, sig, budget_s=15.0):
    """Oracle (CPU restatement, kind 'port') timed on this box's host cores on a bounded
    sample of the same workload: the rank's walker set, repeated until about budget_s of
    wall time.  The reference itself cannot be built in this image (needs FFTW3 + LAPACK),
    see DESIGN.md."""
    from oracle import rf_oracle as orc

    orc.build()
    cfg = dict(nfft=p.nfft, deconv_mode=p.deconv_mode, delta=p.delta, t_start=p.t_start, sdep=p.sdep,
               rayps=p.rayps, a_gus=p.a_gus, ipha=p.ipha)
    cores = max(1, min(orc.max_threads(), os.cpu_count() or 1))
    nb = len(nlay)
    t0 = time.perf_counter()
    ll = orc.eval_batch(cfg, obs, r_inv, nlay, layers, sig, p.nsmp, nthreads=cores)   # also the parity sample
    dt1 = time.perf_counter() - t0
    reps = int(max(1, min(2000, budget_s / max(dt1, 1e-3))))
    big = (np.tile(nlay, reps), np.tile(layers, (reps, 1, 1)), np.tile(sig, (reps, 1)))
    t0 = time.perf_counter()
    orc.eval_batch(cfg, obs, r_inv, big[0], big[1], big[2], p.nsmp, nthreads=cores)
    dt = time.perf_counter() - t0
    n = nb * reps
    return {"value": n / dt, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"{reps} passes over the rank's {nb} walkers ({n} evals), oracle/rf_oracle.c "
                      f"(gcc -O2 -ffp-contract=off, OpenMP x{cores} threads), {dt:.1f} s wall"}, ll, nb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default=os.environ.get("RFGPU_BENCH_WORKLOAD", "c2"), choices=sorted(WORKLOADS))
    ap.add_argument("--walkers", type=int, default=0, help="override walkers per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--swap", default="allgather", choices=["allgather", "p2p"],
                    help="temperature exchange of tempered workloads: K disjoint pairs via one all_gather, or the "
                         "reference's one pair per iteration via send/recv")
    ap.add_argument("--also", default="", help="comma list of extra workloads measured briefly into 'also'")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RFGPU_BENCH_BACKEND", "nccl")   # "gloo": functional test of the N > 1 path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if os.environ.get("RFGPU_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def run(workload, steps, warmup, with_cpu):
        from rf_inv_amd import RFEngine, read_ref_model
        from rf_inv_amd.likelihood import init_r_inv
        from rf_inv_amd.pt import PTSwap

        w = dict(WORKLOADS[workload])
        if args.walkers:
            w["walkers"] = args.walkers
        p = make_params(w)
        ref = read_ref_model(os.path.join(ROOT, "tests", "golden", "sample_syn", "model", "sample.velmod"))
        nb = w["walkers"]
        nlay, layers = draw_walkers(p, ref, rank * nb, nb)
        sig = np.full((nb, p.ntrc), 0.01)
        r_inv = init_r_inv(p.nsmp, p.a_gus, p.delta)
        # observed traces: noise-free synthetic of a fixed 3-interface model, produced by the
        # HIP path itself
        from rf_inv_amd import format_model
        zt = np.zeros(max(p.k_max - 1, 1)); dvt = np.zeros(p.k_max); dst = np.zeros(p.k_max)
        zt[:3] = [3.1 + p.sdep, 7.7 + p.sdep, 14.2 + p.sdep]; dst[:3] = [-0.6, 0.2, 0.5]; dst[p.k_max - 1] = 0.9
        nl_t, a_t, b_t, r_t, h_t, ok = format_model(p, ref, 3, zt, dvt, dst)
        assert ok
        kw = dict(nfft=p.nfft, delta=p.delta, t_start=p.t_start, deconv_mode=p.deconv_mode, sdep=p.sdep,
                  rayps=p.rayps, a_gus=p.a_gus, ipha=p.ipha, nsmp=p.nsmp, nlay_max=p.k_max + 2,
                  device=local_rank)
        with RFEngine(obs=np.zeros((p.ntrc, p.nsmp)), r_inv=r_inv, max_walkers=1, **kw) as e0:
            obs = np.ascontiguousarray(e0.calc_rf(nl_t, a_t, b_t, r_t, h_t)[:p.nsmp].T)
        eng = RFEngine(obs=obs, r_inv=r_inv, max_walkers=nb, **kw)

        stream = torch.cuda.Stream(device=dev)
        d_ids = torch.arange(nb, dtype=torch.int32, device=dev)
        d_nlay = torch.from_numpy(nlay).to(dev)
        d_layers = torch.from_numpy(layers).to(dev)
        d_sig = torch.from_numpy(sig).to(dev)
        d_logl = torch.empty(nb, dtype=torch.float64, device=dev)
        h_logl = torch.empty(nb, dtype=torch.float64).pin_memory()
        swap = PTSwap(eng, nb, w["temps"], dev, seed=1234, t_high=15.0, mode=args.swap) if w["temps"] > 1 else None

        # logL read-back: without a swap step the kernel writes logL straight into the pinned
        # (device-mapped) host buffer -- no copy kernel after the evaluation; with a swap step logL
        # is consumed on the device first and copied afterwards
        zero_copy = swap is None and not os.environ.get("RFGPU_BENCH_COPY")

        def step():
            with torch.cuda.stream(stream):
                if zero_copy:
                    eng.eval_batch_device(d_ids, d_nlay, d_layers, d_sig, h_logl, stream=stream)
                else:
                    eng.eval_batch_device(d_ids, d_nlay, d_layers, d_sig, d_logl, stream=stream)
                    if swap is not None:
                        swap.step(d_logl, stream)
                    h_logl.copy_(d_logl, non_blocking=True)

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

        for _ in range(warmup):
            step()
        barrier()
        # HIP events around the dominant kernel of every 8th step of the timed region: an event record costs
        # ~4 us of stream time, so timing every step would slow the loop it measures by ~8 % at C2
        eng.profile_enable(0 if os.environ.get("RFGPU_BENCH_NOPROF") else min(8, max(1, steps // 8)))
        import gc

        gc.collect()
        gc.disable()            # no collector pause inside the (short) timed region
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        eng.profile_enable(False)
        prof = eng.profile_read()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        ll_gpu = h_logl.numpy().copy()
        assert np.all(np.isfinite(ll_gpu)) or os.environ.get("RFGPU_ABLATE"), "non-finite logL in the benchmark batch"

        f_spec, f_tot, b_alg = alg_work(p, nlay.astype(np.float64), eng.is_ray_common)
        plan = eng.launch_plan
        n_l = max(prof["launches"], 1)          # batches timed
        # dominant kernel: fused_kernel (propagator + trace + logL in one launch) where every trace has
        # its own forward computation, else spectra_kernel (then trace_kernel follows it)
        spectra_ms = prof["spectra_ms"] / n_l   # one launch per batch
        f_dom = f_tot if plan["fused"] else f_spec
        kname = "rfgpu::fused_kernel" if plan["fused"] else "rfgpu::spectra_kernel"
        res = {
            "value": world * nb * steps / dt,
            "ms_per_step": 1e3 * dt / steps,
            "config": {"workload": w["desc"], "walkers_per_gpu": nb, "nfft": p.nfft, "ntrc": p.ntrc,
                       "nsmp": p.nsmp, "k_max": p.k_max, "mean_nlay": float(nlay.mean()),
                       "max_nlay": int(nlay.max()), "deconv_mode": p.deconv_mode, "sdep": p.sdep,
                       "logl_readback": "kernel writes pinned host memory" if zero_copy else "device buffer + async copy",
                       "temperatures": w["temps"], "parallelism": f"walkers sharded x{world}",
                       "pt_swap": (f"{args.swap}, {swap.k} pair(s)/step" if swap is not None else "none")},
            "roofline": {
                "bound": "mfma", "unit": "TFLOP/s", "peak": FP64_PEAK_TFLOPS,
                "achieved": float(f_dom.sum()) / (spectra_ms * 1e-3) / 1e12 if spectra_ms > 0 else None,
                "frac": float(f_dom.sum()) / (spectra_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if spectra_ms > 0 else None,
                "traffic": measured_traffic(kname, workload, nb),
                "kernel": kname, "kernel_ms": spectra_ms, "launch_plan": plan,
                # executed (not algorithmic) fp64 rate, an ESTIMATE from the ISA of the chained-phase loop:
                # ~65 fp64 VALU ops per (bin, layer), ~55 % of them FMAs -> ~100 flop; + ~300 flop per bin
                # for the boundary condition
                "executed_tflops_est": (float((nlay - 1 - (1 if p.sdep > 0 else 0)).sum() * 100.0 + 300.0 * nb)
                                        * (1 if eng.is_ray_common else p.ntrc) * (p.nfft // 2 + 1)
                                        / (spectra_ms * 1e-3) / 1e12) if spectra_ms > 0 else None,
                "note": "fp64: MI355X matrix (MFMA) peak == vector peak = 78.6 TF; the kernel issues fp64 VALU FMA, "
                        "MFMA not used (no rate advantage). achieved = reference-arithmetic flops (SURVEY 8d: "
                        "570/(bin*layer)+580/bin, + FFT/shift/quadratic form when fused) per launch / live "
                        "HIP-event kernel time; the eigen-coordinate real-form propagator executes ~1/6 of those flops, so the "
                        "algorithmic fraction can exceed 1 (DESIGN.md section 3).",
            },
            "roofline_hbm": {
                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "achieved": float(b_alg.sum()) * steps / dt / 1e9,
                "frac": float(b_alg.sum()) * steps / dt / 1e9 / HBM_PEAK_GBS,
                "note": "algorithmic bytes/eval (layers+sigma in, prop_rft(nfft,ntrc)+logL out) x evals/s of this rank",
            },
            "kernel_ms": ({"fused": spectra_ms} if plan["fused"] else
                          {"spectra": spectra_ms, "trace": prof["trace_ms"] / n_l}),
            "alg_gflop_per_step": float(f_tot.sum()) / 1e9,
        }
        if with_cpu and rank == 0:
            base, ll_cpu, n = cpu_baseline(p, obs, r_inv, nlay, layers, sig)
            res["cpu_baseline"] = base
            d = np.abs(ll_gpu[:n] - ll_cpu)
            res["parity_in_bench"] = {"n": n, "max_abs_dlogl": float(d.max()),
                                      "max_rel_dlogl": float((d / np.abs(ll_cpu)).max())}
        eng.close()
        return res

    main_res = run(args.workload, args.steps, args.warmup, not args.no_cpu_baseline and world == 1)
    also = {}
    for wl in [x for x in args.also.split(",") if x]:
        r = run(wl, max(30, min(200, args.steps // 2)), max(5, min(20, args.warmup // 2)), False)
        also[wl] = {k: r[k] for k in ("value", "ms_per_step", "config", "roofline", "kernel_ms")}
    if rank == 0:
        out = {"metric": "forward+likelihood evals/sec (whole node)", "value": main_res["value"], "unit": "evals/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic"}
        out.update({k: v for k, v in main_res.items() if k not in ("value", "ms_per_step")})
        if also:
            out["also"] = also
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
