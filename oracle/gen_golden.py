"""Generate tests/golden/ref/*.npz: inputs and outputs of the reference's OWN code, run on the CPU (test infrastructure).

    make -C oracle -f Makefile.ref        # the whole reference, unmodified, + MKL's FFTW3 interface / LAPACK; no GPU
    python oracle/gen_golden.py              # needs /root/reference (the build) -- runs HERE only; the fixtures travel

Every number in a fixture's outputs comes out of oracle/_ref/cpu_o0/ref_forward_dump or ref_path_dump (the reference
Makefile's default -O0 class; `*_o2` fields: the same run through the -O2 build, i.e. the reference's own spread between
two builds of itself).  No product code and no oracle code computes anything here: rf_inv_amd's Python host mirror only
WRITES the input files the reference reads (params.in, SAC traces) and draws bench.py's walker models; the observed
traces of the likelihood fixtures are the reference's own synthetic of bench.py's fixed three-interface model.

  forward_<case>.npz   calc_rf (src/forward.f90:123-208) on hand-built layer stacks: the matrix of SURVEY.md section 8c
                       (land / ocean x P / S x deconvolution 0 / 1 x common / separate rays x 2 .. 31 layers x nfft 256 ..
                       4096, an odd length, the DC bin in every trace, an evanescent (NaN) trace, the scenario of the
                       reference's own src/forward_test.f90)
  path_<workload>.npz  calc_likelihood (src/likelihood.f90:56-101) with fwd_flag = .true. on bench.py's own walkers of
                       that workload (the first ones, the deepest, the shallowest), and with fwd_flag = .false. on
                       host-stored traces; format_model's layer stacks, the pseudo-inverse as init_r_inv forms it
  format_model.npz     format_model (src/model.f90:175-290) on 3 x 700 random proposals -- valid and invalid, with equal
                       interface depths (the unstable quicksort's permutation), ocean, vp_mode 0 / 1, a NON-uniform
                       reference velocity table: layer counts, verdicts and layer values
  run_sample_syn.npz   the reference's main program rf_inv on the shipped sample_syn directory (60 + 240 iterations x 5 chains;
                       1 and 2 MPI ranks; and 1 rank at the shipped full length 3000 + 8000): sha256 of each of the
                       eleven model / histogram / mean files it writes and the whole rslt/likelihood table
  MANIFEST.json        file list with sha256 (tests fail, not skip, on a missing or altered fixture)

Consumers: tests/test_reference_fixtures.py (-m "not gpu": the oracle against the fixtures; -m gpu: the HIP path)."""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "tests", "golden", "ref")

# name -> (nfft, rayps, ipha, a_gus, deconv_mode, sdep, t_start)        [the cases of tests/test_reference_forward.py + more]
FORWARD_CASES = {
    "c1_ocean_2P_nfft256": (256, [0.06, 0.08], [1, 1], [4.0, 4.0], 0, 2.0, 0.0),
    "land_S_nfft256_decon": (256, [0.09], [-1], [3.0], 1, 0.0, -1.0),
    "c2_land_P": (4096, [0.06], [1], [4.0], 0, 0.0, 0.0),
    "c2d_land_P_decon": (4096, [0.06], [1], [4.0], 1, 0.0, 0.0),
    "c4_land_PPS": (4096, [0.06, 0.08, 0.10], [1, 1, -1], [4.0, 4.0, 4.0], 0, 0.0, 0.0),
    "c4_land_PPS_decon_tstart": (4096, [0.06, 0.08, 0.10], [1, 1, -1], [4.0, 2.5, 4.0], 1, 0.0, -3.0),
    "c5_ocean_PPSS": (4096, [0.06, 0.08, 0.10, 0.12], [1, 1, -1, -1], [4.0] * 4, 0, 2.0, 0.0),
    "c5_ocean_PPSS_decon": (4096, [0.06, 0.08, 0.10, 0.12], [1, 1, -1, -1], [4.0] * 4, 1, 2.0, -1.0),
    "c4common_land_3P_one_ray": (4096, [0.06, 0.06, 0.06], [1, 1, 1], [4.0, 2.5, 1.5], 0, 0.0, 0.0),
    "common_ocean_3S_one_ray_nfft2048": (2048, [0.10, 0.10, 0.10], [-1, -1, -1], [4.0, 2.5, 1.5], 0, 2.0, -2.0),
    "common_land_2S_decon_nfft512": (512, [0.07, 0.07], [-1, -1], [4.0, 2.0], 1, 0.0, -2.0),
    "odd_length_nfft1000_S": (1000, [0.11], [-1], [3.0], 0, 0.0, 0.0),
    # the long-series plans of the library (row a12): a power of two beyond the in-LDS transform (four-step through HBM) and a
    # length that is neither a power of two nor short (Bluestein); the reference's FFTW interface takes any length
    "long_nfft8192_PS": (8192, [0.06, 0.10], [1, -1], [4.0, 2.5], 0, 0.0, -1.0),
    "bluestein_nfft3000_ocean_P_decon": (3000, [0.07], [1], [4.0], 1, 2.0, 0.0),
}
FORWARD_FEWER_STACKS = {"long_nfft8192_PS": 4, "bluestein_nfft3000_ocean_P_decon": 4}      # (keeps the files small)
PATH_WORKLOADS = {"c1": 40, "c2": 48, "c2d": 48, "c4": 56, "c4d": 48, "c4common": 40, "c5": 40, "c5d": 40, "c4w20": 20}
N_FULL = 3          # models per path fixture whose whole rft(nfft, ntrc) is kept (all keep samples 1 .. nsmp)


def forward_stacks(rng, ocean, sdep):
    from helpers import random_stack

    sizes = (3, 4, 7, 15, 20, 31) if ocean else (2, 3, 6, 15, 20, 30)
    stacks = [random_stack(rng, n, ocean, sdep) for n in sizes]
    # a soft surface layer on a fast half-space: spectra that dip below the water level (the level clips bins)
    soft = (np.array([1.6, 6.0, 8.0]), np.array([0.2, 3.5, 4.5]), np.array([1.5, 2.7, 3.3]), np.array([0.5, 30.0, 999.0]))
    if ocean:
        soft = tuple(np.concatenate([[w], x]) for w, x in zip((1.5, -999.0, 1.0, sdep), soft))
    return stacks + [soft]


def forward_params(nfft, rayps, ipha, a_gus, deconv, sdep, t_start, t_end=5.0):
    from helpers import DELTA
    from rf_inv_amd import get_params
    from rf_inv_amd.params import _nint

    p = get_params(os.path.join(ROOT, "tests", "golden", "sample_syn", "params.in"))
    n = len(rayps)
    p.ntrc, p.nfft, p.deconv_mode, p.sdep, p.t_start, p.t_end, p.k_max = n, nfft, deconv, float(sdep), t_start, t_end, 31
    p.rayps, p.a_gus, p.ipha = np.asarray(rayps, float), np.asarray(a_gus, float), np.asarray(ipha, dtype=np.int32)
    p.sig_min = p.sig_max = np.full(n, 0.01)
    p.delta = DELTA
    p.nsmp = int(round((t_end - t_start) / DELTA)) + 1
    assert p.nsmp == _nint(t_end / DELTA) - _nint(t_start / DELTA) + 1
    return p


def gen_forward(name, case, stacks=None):
    from helpers import pack_layers
    from oracle import refrun

    nfft, rayps, ipha, a_gus, deconv, sdep, t_start = case
    if stacks is None:
        stacks = forward_stacks(np.random.default_rng(sum(map(ord, name))), sdep > 0, sdep)
        if name in FORWARD_FEWER_STACKS:
            stacks = stacks[1:FORWARD_FEWER_STACKS[name]] + stacks[-1:]
    p = forward_params(nfft, rayps, ipha, a_gus, deconv, sdep, t_start)
    res = {}
    for build in refrun.BUILDS:
        with tempfile.TemporaryDirectory() as work:
            text = refrun.write_run_dir(work, p, header=f"oracle/gen_golden.py: forward case {name}")
            refrun.write_stacks(os.path.join(work, "stacks.txt"), stacks)
            res[build] = refrun.run_forward(build, work, len(stacks), nfft, p.ntrc)
    r0, r2 = res["cpu_o0"], res["cpu_o2"]
    assert np.array_equal(r0["npre"], r2["npre"]) and np.array_equal(r0["flt"], r2["flt"])
    nlay, layers = pack_layers(stacks, 33)
    scale = np.nanmax(np.abs(r0["rft"]), axis=2, keepdims=True)
    with np.errstate(invalid="ignore"):
        spread = np.nanmax(np.abs(r2["rft"] - r0["rft"]) / scale, axis=2)
    np.savez(os.path.join(OUT, f"forward_{name}.npz"), name=name, params_in=text, nfft=nfft, ntrc=p.ntrc, nsmp=p.nsmp,
             delta=p.delta, t_start=t_start, t_end=p.t_end, deconv_mode=deconv, sdep=float(sdep), rayps=p.rayps, ipha=p.ipha,
             a_gus=p.a_gus, nlay=nlay, layers=layers, flt=r0["flt"], rft=r0["rft"], tp=r0["tp"], npre=r0["npre"],
             is_ray_common=r0["common"], o2_rel_trace_spread=spread, build="cpu_o0")
    print(f"forward_{name}: {len(stacks)} stacks, -O2 against -O0 max rel trace difference {np.nanmax(spread):.2e}", flush=True)


def gen_path(workload, count):
    import bench
    from oracle import refrun
    from rf_inv_amd import read_ref_model

    w = dict(bench.WORKLOADS[workload])
    p = bench.make_params(w)
    refm = read_ref_model(os.path.join(ROOT, "tests", "golden", "sample_syn", "model", "sample.velmod"))
    nlay_b, _, (m_k, m_z, m_dvp, m_dvs) = bench.draw_walkers(p, refm, 0, 4 * count, return_models=True, procs=1)
    pick = np.unique(np.concatenate([np.arange(count), np.argsort(nlay_b, kind="stable")[-6:], np.argsort(nlay_b, kind="stable")[:3]]))
    m_k, m_z, m_dvp, m_dvs = m_k[pick], m_z[pick], m_dvp[pick], m_dvs[pick]
    n = len(pick)
    rng = np.random.default_rng(sum(map(ord, workload)) + 1)
    sig = rng.uniform(0.01, 0.03, (n, p.ntrc))
    sig[: n // 2] = 0.01                                   # half at bench.py's sigma, half spread
    kz = max(p.k_max - 1, 1)
    # pass 1: the reference's own synthetic of bench.py's fixed three-interface model = the observed traces
    zt = np.zeros((1, p.k_max)); dvt = np.zeros((1, p.k_max)); dst = np.zeros((1, p.k_max))
    zt[0, :3] = [3.1 + p.sdep, 7.7 + p.sdep, 14.2 + p.sdep]; dst[0, :3] = [-0.6, 0.2, 0.5]; dst[0, p.k_max - 1] = 0.9
    with tempfile.TemporaryDirectory() as work:
        refrun.write_run_dir(work, p)
        refrun.write_models(os.path.join(work, "models.txt"), p.k_max, np.array([3]), zt, dvt, dst, np.full((1, p.ntrc), 0.01))
        truth = refrun.run_path("cpu_o0", work, 1, p)
    obs_full = truth["rft"][0]
    # what the reference reads back from a SAC file: float32 samples
    obs = np.stack([obs_full[t, :p.nsmp].astype(np.float32).astype(np.float64) for t in range(p.ntrc)])
    res = {}
    for build in refrun.BUILDS:
        with tempfile.TemporaryDirectory() as work:
            text = refrun.write_run_dir(work, p, obs=obs_full, header=f"oracle/gen_golden.py: bench.py workload {workload}")
            refrun.write_models(os.path.join(work, "models.txt"), p.k_max, m_k, m_z, m_dvp, m_dvs, sig)
            res[build] = refrun.run_path(build, work, n, p)
    r0, r2 = res["cpu_o0"], res["cpu_o2"]
    assert np.array_equal(r0["nlay"], r2["nlay"]) and np.array_equal(r0["npre"], r2["npre"]) and r0["valid"].all()
    assert np.array_equal(r0["nlay"], nlay_b[pick]) and r0["nlay"].max() <= p.k_max + 2
    pad = p.k_max + 2
    # one stored matrix per distinct Gaussian parameter
    a_unique, r_index = np.unique(p.a_gus, return_inverse=True)
    r_inv = np.stack([r0["r_inv"][int(np.nonzero(r_index == j)[0][0])] for j in range(len(a_unique))])
    for t in range(p.ntrc):
        assert np.array_equal(r0["r_inv"][t], r_inv[r_index[t]])
    m = len(r0["probe_logl"])
    np.savez(os.path.join(OUT, f"path_{workload}.npz"), workload=workload, params_in=text, walker_ids=pick.astype(np.int32),
             nfft=p.nfft, ntrc=p.ntrc, nsmp=p.nsmp, delta=p.delta, t_start=p.t_start, t_end=p.t_end, deconv_mode=p.deconv_mode,
             sdep=p.sdep, rayps=p.rayps, ipha=p.ipha, a_gus=p.a_gus, k_max=p.k_max,
             obs=obs, truth_nlay=truth["nlay"][0], truth_layers=truth["layers"][0][:, :pad],
             k=m_k, z=m_z[:, :kz], dvp=m_dvp, dvs=m_dvs, sig=sig,
             logl=r0["logl"], logl_o2=r2["logl"], nlay=r0["nlay"], layers=r0["layers"][:, :, :pad], tp=r0["tp"], npre=r0["npre"],
             rft_window=r0["rft"][:, :, :p.nsmp], rft_full=r0["rft"][:N_FULL],
             o2_rel_trace_spread=np.max(np.abs(r2["rft"] - r0["rft"]) / np.max(np.abs(r0["rft"]), axis=2, keepdims=True), axis=2),
             r_inv=r_inv, r_index=r_index.astype(np.int32),
             probe_logl=r0["probe_logl"], probe_window=r0["probe_trace"][:, :, :p.nsmp], probe_sig=r0["probe_sig"].reshape(m, p.ntrc),
             build="cpu_o0")
    d = np.abs(r2["logl"] - r0["logl"])
    print(f"path_{workload}: {n} models (nlay {r0['nlay'].min()} .. {r0['nlay'].max()}), |logL| {np.abs(r0['logl']).min():.3g} .. "
          f"{np.abs(r0['logl']).max():.3g}; the reference -O2 against -O0: max |dlogL| {d.max():.2e}, max rel "
          f"{(d / np.abs(r0['logl'])).max():.2e}", flush=True)


FORMAT_CASES = [(2.0, 0, 10, False), (0.0, 1, 30, False), (0.0, 0, 12, True)]      # (sdep, vp_mode, k_max, ties)


def format_proposals(rng, p, nb, ties=False):
    """Proposals as a chain may make them (tests/test_gpu_format_model.py draws the same way): some valid, some not."""
    k = rng.integers(p.k_min, p.k_max, nb).astype(np.int32)
    z = np.zeros((nb, p.k_max - 1)); dvp = np.zeros((nb, p.k_max)); dvs = np.zeros((nb, p.k_max))
    for i in range(nb):
        z[i, :k[i]] = rng.uniform(p.z_min + p.sdep, p.z_max, k[i])
        if ties and k[i] >= 3:
            z[i, 1] = z[i, 0]            # equal interface depths: the unstable quicksort's permutation matters
        dvs[i, :k[i]] = rng.normal(0, 0.5, k[i]); dvs[i, -1] = rng.normal(0, 0.5)
        dvp[i, :k[i]] = rng.normal(0, 0.3, k[i]); dvp[i, -1] = rng.normal(0, 0.3)
        z[i, k[i]:] = rng.uniform(0, 20, p.k_max - 1 - k[i])   # stale entries beyond k must be ignored
    return k, z, dvp, dvs


def gen_format():
    """format_model through the reference (ref_path_dump's extras: its public format_model on every model handed in)."""
    from oracle import refrun

    nref = 61
    zr = 0.5 * np.arange(nref)
    vp_ref, vs_ref = 5.0 + 0.03 * np.arange(nref), 2.8 + 0.02 * np.arange(nref)     # non-uniform: the iz look-ups matter
    velmod = "".join(f"{float(zr[i])!r} {float(vp_ref[i])!r} {float(vs_ref[i])!r}\n" for i in range(nref))
    out = {}
    for ci, (sdep, vp_mode, k_max, ties) in enumerate(FORMAT_CASES):
        p = forward_params(256, [0.06], [1], [4.0], 0, sdep, 0.0)
        p.k_max, p.vp_mode, p.k_min = k_max, vp_mode, 1
        p.z_min, p.z_max = 0.0 + sdep, 20.0 + sdep
        k, z, dvp, dvs = format_proposals(np.random.default_rng(1000 + k_max), p, 700, ties)
        with tempfile.TemporaryDirectory() as work:
            text = refrun.write_run_dir(work, p, header=f"oracle/gen_golden.py: format_model case {ci}", velmod=velmod)
            refrun.write_models(os.path.join(work, "models.txt"), k_max, k, np.pad(z, ((0, 0), (0, 1))), dvp, dvs,
                                np.full((len(k), 1), 0.01))
            r = refrun.run_path("cpu_o0", work, len(k), p)
        pad = k_max + 2
        assert r["nlay"].max() <= pad and 0 < r["valid"].sum() < len(k)
        out.update({f"c{ci}_params_in": text, f"c{ci}_k": k, f"c{ci}_z": z, f"c{ci}_dvp": dvp, f"c{ci}_dvs": dvs,
                    f"c{ci}_nlay": r["nlay"], f"c{ci}_valid": r["valid"], f"c{ci}_layers": r["layers"][:, :, :pad]})
        print(f"format_model case {ci} (sdep {sdep}, vp_mode {vp_mode}, k_max {k_max}, ties {ties}): {len(k)} proposals, "
              f"{int(r['valid'].sum())} valid, nlay {r['nlay'].min()} .. {r['nlay'].max()}", flush=True)
    np.savez_compressed(os.path.join(OUT, "format_model.npz"), cases=np.array(FORMAT_CASES, dtype=float), velmod=velmod,
                        z_ref=zr, vp_ref=vp_ref, vs_ref=vs_ref, build="cpu_o0", **out)


RUN_FILES = ["all_models", "num_interface.ppd", "syn_trace.ppd", "interface_depth.ppd", "sigma.ppd", "vs_z.ppd", "vp_z.ppd",
             "vpvs_z.ppd", "vs_z.mean", "vp_z.mean", "vpvs_z.mean"]
RUNS = [(1, 60, 240), (2, 60, 240), (1, 3000, 8000)]         # (MPI ranks, N_BURN, N_ITER)


def gen_runs():
    """The reference's main program end to end on tests/golden/sample_syn (params.in with the iteration counts replaced)."""
    import shutil
    import subprocess

    from oracle import refrun

    out = {}
    for nranks, nburn, niter in RUNS:
        tag = f"np{nranks}_{nburn}_{niter}"
        res = {}
        for build in refrun.BUILDS:
            with tempfile.TemporaryDirectory() as tmp:
                work = os.path.join(tmp, "run")
                shutil.copytree(os.path.join(ROOT, "tests", "golden", "sample_syn"), work)
                os.makedirs(os.path.join(work, "rslt"))
                lines = open(os.path.join(work, "params.in")).read().split("\n")
                i = next(j for j, l in enumerate(lines) if l.startswith("# N_BURN"))
                assert lines[i + 1].strip() == "3000" and lines[i + 3].strip() == "8000"
                lines[i + 1], lines[i + 3] = str(nburn), str(niter)
                open(os.path.join(work, "params.in"), "w").write("\n".join(lines))
                r = subprocess.run(["/opt/conda/bin/mpiexec", "-np", str(nranks), refrun.exe(build, "rf_inv"), "params.in"], cwd=work,
                                   env=refrun.clean_env(), capture_output=True, text=True, timeout=3600)
                assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
                res[build] = ({f: hashlib.sha256(open(os.path.join(work, "rslt", f), "rb").read()).hexdigest() for f in RUN_FILES},
                              np.loadtxt(os.path.join(work, "rslt", "likelihood")))
        (h0, l0), (h2, l2) = res["cpu_o0"], res["cpu_o2"]
        same = h0 == h2
        out[f"{tag}_sha256"] = np.array([h0[f] for f in RUN_FILES])
        out[f"{tag}_likelihood"] = l0
        out[f"{tag}_o2_same_files"] = same
        out[f"{tag}_o2_max_rel_dlikelihood"] = float(np.max(np.abs(l2[:, 1] - l0[:, 1]) / np.abs(l0[:, 1])))
        print(f"run_sample_syn {tag}: rslt/likelihood[0] = {l0[0, 1]!r}; the -O2 build writes the same eleven files: {same}, "
              f"its likelihood column within {out[f'{tag}_o2_max_rel_dlikelihood']:.1e}", flush=True)
    # BASELINE.json configs[0]: the shipped params.in with ONE P trace, 4 chains per rank (1 at T = 1), -np 8 = "8 x 4"
    from rf_inv_amd import get_params, write_params

    tag = "configs0_np8"
    res = {}
    for build in refrun.BUILDS:
        with tempfile.TemporaryDirectory() as tmp:
            work = os.path.join(tmp, "run")
            shutil.copytree(os.path.join(ROOT, "tests", "golden", "sample_syn"), work)
            os.makedirs(os.path.join(work, "rslt"))
            p = get_params(os.path.join(work, "params.in"))
            p.ntrc, p.nchains, p.ncool, p.nburn, p.niter = 1, 4, 1, 60, 240
            write_params(os.path.join(work, "params.in"), p, header="BASELINE configs[0] (oracle/gen_golden.py)")
            text = open(os.path.join(work, "params.in")).read()
            r = subprocess.run(["/opt/conda/bin/mpiexec", "-np", "8", refrun.exe(build, "rf_inv"), "params.in"], cwd=work,
                               env=refrun.clean_env(), capture_output=True, text=True, timeout=3600)
            assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
            res[build] = ({f: hashlib.sha256(open(os.path.join(work, "rslt", f), "rb").read()).hexdigest() for f in RUN_FILES},
                          np.loadtxt(os.path.join(work, "rslt", "likelihood")))
    (h0, l0), (h2, l2) = res["cpu_o0"], res["cpu_o2"]
    out[f"{tag}_sha256"], out[f"{tag}_likelihood"], out[f"{tag}_params_in"] = np.array([h0[f] for f in RUN_FILES]), l0, text
    out[f"{tag}_o2_same_files"] = h0 == h2
    print(f"run_sample_syn {tag}: rslt/likelihood[0] = {l0[0, 1]!r}; the -O2 build writes the same eleven files: {h0 == h2}", flush=True)
    np.savez(os.path.join(OUT, "run_sample_syn.npz"), files=np.array(RUN_FILES), build="cpu_o0", **out)


def main():
    from oracle import refrun

    if not (refrun.available("cpu_o0") and refrun.available("cpu_o2")):
        raise SystemExit("oracle/gen_golden.py: build oracle/_ref/cpu_o0 and cpu_o2 first (make -C oracle -f Makefile.ref)")
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])
    for name, case in FORWARD_CASES.items():
        if not only or f"forward_{name}" in only:
            gen_forward(name, case)
    # the scenario of the reference's own src/forward_test.f90:39-56 (a 20 km layer over a half-space, one S trace, a = 8)
    if not only or "forward_reference_forward_test" in only:
        gen_forward("reference_forward_test", (1024, [0.06], [-1], [8.0], 0, 0.0, -3.0),
                    stacks=[(np.array([5.0, 8.0]), np.array([2.5, 4.0]), np.array([3.0, 3.3]), np.array([20.0, -10.0]))])
    # an evanescent layer for the second ray (1 / v^2 < p^2: sqrt of a negative number, src/forward.f90:396-397): NaN trace
    if not only or "forward_evanescent_nan" in only:
        gen_forward("evanescent_nan", (256, [0.06, 0.30], [1, 1], [4.0, 4.0], 0, 0.0, 0.0),
                    stacks=[(np.array([3.0, 6.0]), np.array([1.7, 3.4]), np.array([2.3, 2.8]), np.array([2.0, 999.0]))])
    for workload, count in PATH_WORKLOADS.items():
        if not only or f"path_{workload}" in only:
            gen_path(workload, count)
    if not only or "format_model" in only:
        gen_format()
    if not only or "run_sample_syn" in only:
        gen_runs()
    files = sorted(f for f in os.listdir(OUT) if f.endswith(".npz"))
    man = {f: {"sha256": hashlib.sha256(open(os.path.join(OUT, f), "rb").read()).hexdigest(),
               "bytes": os.path.getsize(os.path.join(OUT, f))} for f in files}
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as fh:
        json.dump({"generator": "oracle/gen_golden.py", "build": "oracle/Makefile.ref (cpu_o0; *_o2 fields: cpu_o2)",
                   "files": man}, fh, indent=1, sort_keys=True)
    print(f"{len(files)} fixtures, {sum(v['bytes'] for v in man.values()) / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
