"""Row f-3: posterior accumulation on the device (rf_post_*) and the result files.

(a) against oracle/posterior_oracle.py (restatement of src/pt_mcmc.f90:204-286) on synthetic
    chain states -- every histogram and the order-dependent fp64 sums bit-exact;
(b) against the reference's OWN pt_mcmc.f90 + mcmc_out.f90 (compiled unmodified into oracle/_ref):
    the files they write for a sample_syn run equal what rf_inv_amd.mcmc_out writes from the device
    accumulators on the same trajectory.
"""
import copy
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import DELTA

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RFINV = os.path.join(ROOT, "oracle", "_ref", "drive_rfinv")
INT_FIELDS = ("nk", "nz", "nsig", "namp", "nvpz", "nvsz", "nvpvsz")
F64_FIELDS = ("vp_mean", "vs_mean", "vpvs_mean", "vp_model", "vs_model", "all_likelihood")


def _setup(golden_dir, sdep, vp_mode, k_max):
    from rf_inv_amd import get_params, read_obs, read_ref_model

    p = get_params(os.path.join(golden_dir, "sample_syn", "params.in"))
    read_obs(p)
    p.sdep, p.vp_mode, p.k_max = sdep, vp_mode, k_max
    ref = copy.copy(read_ref_model(os.path.join(p.base_dir, p.vel_file)))
    ref.vp_ref = 5.0 + 0.03 * np.arange(ref.vp_ref.size)
    ref.vs_ref = 2.8 + 0.02 * np.arange(ref.vs_ref.size)
    mcfg = dict(k_max=p.k_max, vp_mode=vp_mode, sdep=sdep, z_max=p.z_max, h_min=p.h_min, z_ref_min=ref.z_ref_min,
                dz_ref=ref.dz_ref, vp_min=p.vp_min, vp_max=p.vp_max, vs_min=p.vs_min, vs_max=p.vs_max,
                vpvs_min=p.vpvs_min, vpvs_max=p.vpvs_max, vp_ref=ref.vp_ref, vs_ref=ref.vs_ref)
    return p, ref, mcfg


def _valid_states(oracle, rng, p, mcfg, n):
    """Chain states whose models pass format_model's validity rules (recorded models always do)."""
    k = np.zeros(n, dtype=np.int32)
    z = np.zeros((n, p.k_max - 1)); dvp = np.zeros((n, p.k_max)); dvs = np.zeros((n, p.k_max))
    for i in range(n):
        while True:
            ki = int(rng.integers(p.k_min, p.k_max))
            zi = np.zeros(p.k_max - 1); a = np.zeros(p.k_max); b = np.zeros(p.k_max)
            zi[:ki] = rng.uniform(p.z_min + p.sdep + 0.3, p.z_max, ki)
            b[:ki] = rng.normal(0, 0.25, ki); b[-1] = rng.normal(0, 0.25)
            a[:ki] = rng.normal(0, 0.2, ki); a[-1] = rng.normal(0, 0.2)
            zi[ki:] = rng.uniform(0, 20, p.k_max - 1 - ki)      # stale entries beyond k
            if oracle.format_model(mcfg, ki, zi, a, b)[5]:
                k[i], z[i], dvp[i], dvs[i] = ki, zi, a, b
                break
    return k, z, dvp, dvs


def _layers(oracle, mcfg, k, z, dvp, dvs, pad):
    n = k.size
    nlay = np.zeros(n, dtype=np.int32)
    lay = np.ones((n, 4, pad))
    for i in range(n):
        nl, a, b, r, h, ok = oracle.format_model(mcfg, int(k[i]), z[i], dvp[i], dvs[i])
        assert ok
        nlay[i] = nl
        lay[i, 0, :nl], lay[i, 1, :nl], lay[i, 2, :nl], lay[i, 3, :nl] = a, b, r, h
    return nlay, lay


def _compare(res, orc, nm):
    assert res.nmod == orc.nmod
    for f in INT_FIELDS:
        assert np.array_equal(getattr(res, f), getattr(orc, f)), f
    for f in F64_FIELDS:
        a, b = getattr(res, f), getattr(orc, f)
        assert a.shape == b.shape, f
        assert np.array_equal(a, b), (f, np.abs(a - b).max())      # bit-exact, order-dependent sums included
    assert res.amp_out_of_range == orc.amp_out_of_range


@pytest.mark.parametrize("sdep,vp_mode,k_max,ntrc,device_api", [(2.0, 0, 10, 2, False), (0.0, 1, 16, 1, True)])
def test_posterior_record_matches_oracle(oracle, golden_dir, sdep, vp_mode, k_max, ntrc, device_api):
    """Three record calls (a temperature filter on the second) over evaluated chains: ocean layer
    (the vs_mean / vpvs_mean ASSIGNMENT quirk), sigma histogram, a narrow amplitude range that sends
    samples to the edge bins, stale z entries beyond k."""
    import torch

    from oracle.posterior_oracle import PosteriorOracle
    from rf_inv_amd import RFEngine
    from rf_inv_amd.posterior import Posterior

    p, ref, mcfg = _setup(golden_dir, sdep, vp_mode, k_max)
    p.ntrc, p.nsmp = ntrc, 101
    p.sig_mode, p.sig_min, p.sig_max = [1, 0][:ntrc], [0.005, 0.01][:ntrc], [0.08, 0.01][:ntrc]
    p.amp_min, p.amp_max, p.nbin_amp = -0.05, 0.25, 40            # narrow: out-of-range samples occur
    p.nbin_z, p.nbin_vs, p.nbin_vp, p.nbin_vpvs, p.nbin_sig = 37, 25, 20, 15, 11
    n = 96
    p.nchains, p.niter, p.ncorr = n, 3, 1
    nm = 150                                                       # < the ~200 recorded: the last rows are dropped
    rng = np.random.default_rng(100 + k_max)
    k, z, dvp, dvs = _valid_states(oracle, rng, p, mcfg, n)
    sig = np.stack([rng.uniform(0.005, 0.0799, n), np.full(n, 0.01)], axis=1)[:, :ntrc].copy()
    pad = k_max + 2
    nlay, lay = _layers(oracle, mcfg, k, z, dvp, dvs, pad)
    obs = rng.normal(0, 0.05, (ntrc, 101))
    ids = np.arange(n, dtype=np.int32)
    with RFEngine(nfft=256, delta=DELTA, t_start=0.0, deconv_mode=0, sdep=sdep, rayps=[0.06, 0.075][:ntrc],
                  a_gus=[4.0, 3.0][:ntrc], ipha=[1, 1][:ntrc], obs=obs, nsmp=101, max_walkers=n,
                  nlay_max=pad) as eng:
        logl = eng.eval_batch(ids, nlay, lay, sig)
        eng.commit(ids, np.ones(n, dtype=np.int32))
        traces = eng.get_rft_batch(ids, 0, 101)
        eng.set_model(p, ref)
        post = Posterior(eng, p, max_models=nm)
        orc = PosteriorOracle(mcfg=mcfg, ntrc=ntrc, nsmp=101, nbin_z=p.nbin_z, nbin_vs=p.nbin_vs, nbin_vp=p.nbin_vp,
                              nbin_vpvs=p.nbin_vpvs, nbin_sig=p.nbin_sig, nbin_amp=p.nbin_amp, amp_min=p.amp_min,
                              amp_max=p.amp_max, z_min=p.z_min, sig_min=p.sig_min, sig_max=p.sig_max,
                              sig_mode=p.sig_mode, max_models=3 * n)
        temps = np.where(rng.uniform(size=n) < 0.5, 1.0, np.exp(rng.uniform(size=n) * np.log(15.0)))
        temps[3] = 1.0 + 5e-7                                       # inside the 1 + 1e-6 tolerance
        for call, tt in enumerate((None, temps, None)):
            sel = slice(0, n) if call != 2 else slice(10, 70)       # a shorter third batch
            if device_api:
                dev = torch.device("cuda", 0)
                t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                post.record_device(t(ids[sel]), t(k[sel]), t(z[sel]), t(dvp[sel]), t(dvs[sel]), t(sig[sel]),
                                   t(logl[sel]), t(tt[sel]) if tt is not None else None)
                torch.cuda.synchronize()
            else:
                post.record(ids[sel], k[sel], z[sel], dvp[sel], dvs[sel], sig[sel], logl[sel],
                            temps=tt[sel] if tt is not None else None)
            for i in range(n)[sel]:
                orc.record(int(k[i]), z[i], dvp[i], dvs[i], sig[i], logl[i], traces[i],
                           temp=1.0 if tt is None else tt[i])
        res = post.read()
        assert res.nmod == orc.nmod > nm
        assert res.amp_out_of_range > 0 and (sdep == 0.0 or res.vs_mean[0] == p.vs_min)
        # rows beyond max_models are dropped by the device (the reference would overrun its arrays)
        for f in ("vp_model", "vs_model", "all_likelihood"):
            setattr(orc, f, getattr(orc, f)[:nm])
        _compare(res, orc, nm)
        # reset: everything back to the state of init_pt_mcmc
        post.reset()
        z0 = post.read()
        assert z0.nmod == 0 and not z0.namp.any() and not z0.vp_mean.any() and np.all(z0.vs_model[:, 0] == -999.9)


def _read_cols(path):
    return [[float(t) for t in line.split()] for line in open(path) if line.strip()]


@pytest.mark.parametrize("vp_mode", [0, 1])
def test_python_driver_result_files_equal_reference_writer(golden_dir, tmp_path, vp_mode):
    """The reference's own sampler + output_results (pt_mcmc.f90, mcmc_out.f90 unmodified, on the GPU
    drop-in modules) against the batched Python driver recording on the device and
    rf_inv_amd.mcmc_out: the explicitly formatted files are identical text, the list-directed ones
    hold identical numbers."""
    if not os.path.exists(RFINV):
        pytest.skip("oracle/_ref/drive_rfinv not built (no Fortran compiler / reference tree at build time)")
    nburn, niter = 40, 200
    work = tmp_path / "ref"
    shutil.copytree(os.path.join(golden_dir, "sample_syn"), work)
    os.makedirs(work / "rslt")
    par = work / "params.in"
    if vp_mode == 1:
        # second variant: dVp solved and the noise level of trace 1 solved (sigma histogram, 6 proposal types)
        txt = open(par).read().splitlines()
        for key, val in (("# VP_MODE", "1"), ("# SIG_MIN(1:N_TRC)", "0.005 0.05")):
            hits = [i for i, line in enumerate(txt) if line.startswith(key)]
            assert hits, key + " key line not found in params.in"
            j = hits[0] + 1
            while txt[j].startswith("#") or not txt[j].strip():
                j += 1
            txt[j] = val
        open(par, "w").write("\n".join(txt) + "\n")
    r = subprocess.run([RFINV, "params.in", str(nburn), str(niter), "0", "out"], cwd=work, env=dict(os.environ),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "drive_rfinv: ok" in r.stdout, r.stdout + r.stderr

    from rf_inv_amd import RFEngine, get_params, read_obs, read_ref_model
    from rf_inv_amd.mcmc import RJMCMC, EngineEvaluator
    from rf_inv_amd.mcmc_out import output_results
    from rf_inv_amd.mt19937 import MT19937
    from rf_inv_amd.posterior import Posterior

    p = get_params(str(par))
    read_obs(p)
    assert p.vp_mode == vp_mode and list(p.sig_mode[:2]) == [vp_mode, 0]
    p.nburn, p.niter = nburn, niter
    ref = read_ref_model(os.path.join(p.base_dir, p.vel_file))
    out = tmp_path / "ours"
    with RFEngine.from_params(p) as eng:
        eng.set_model(p, ref)
        m = RJMCMC(p, ref, EngineEvaluator(eng, p.k_max + 2), MT19937(p.iseed))
        m.init_model(); m.init_likelihood(); m.init_pt_mcmc()
        m.posterior = Posterior(eng, p)
        for it in range(1, nburn + niter + 1):
            m.iterate(it)
        res = m.posterior.read()
    assert res.nmod == niter // p.ncorr * p.ncool > 0
    output_results(p, res, m.counters, nproc=1, out_dir=str(out))
    names = ["all_models", "likelihood", "num_interface.ppd", "syn_trace.ppd", "interface_depth.ppd", "sigma.ppd",
             "vs_z.ppd", "vp_z.ppd", "vpvs_z.ppd", "vs_z.mean", "vp_z.mean", "vpvs_z.mean"]
    for name in names:
        assert os.path.exists(work / "rslt" / name), name
    for name in ("syn_trace.ppd", "vs_z.ppd", "vp_z.ppd", "vpvs_z.ppd", "vs_z.mean", "vp_z.mean", "vpvs_z.mean"):
        assert open(out / name).read() == open(work / "rslt" / name).read(), name
    for name in ("all_models", "num_interface.ppd", "interface_depth.ppd", "sigma.ppd"):
        a, b = _read_cols(out / name), _read_cols(work / "rslt" / name)
        assert len(a) == len(b) and a == b, name
        assert len(a) > 0 or (name == "sigma.ppd" and vp_mode == 0), name
    a, b = np.array(_read_cols(out / "likelihood")), np.array(_read_cols(work / "rslt" / "likelihood"))
    assert a.shape == b.shape == (nburn + niter, 2)
    assert np.allclose(a, b, rtol=1e-12, atol=1e-9)      # logL: the two hosts add the terms identically


def _compare_result_dirs(ours, theirs, n_it, sigma_solved):
    for name in ("syn_trace.ppd", "vs_z.ppd", "vp_z.ppd", "vpvs_z.ppd", "vs_z.mean", "vp_z.mean", "vpvs_z.mean"):
        assert open(os.path.join(ours, name)).read() == open(os.path.join(theirs, name)).read(), name
    for name in ("all_models", "num_interface.ppd", "interface_depth.ppd", "sigma.ppd"):
        a, b = _read_cols(os.path.join(ours, name)), _read_cols(os.path.join(theirs, name))
        assert len(a) == len(b) and a == b, name
        assert len(a) > 0 or (name == "sigma.ppd" and not sigma_solved), name
    a, b = np.array(_read_cols(os.path.join(ours, "likelihood"))), np.array(_read_cols(os.path.join(theirs, "likelihood")))
    assert a.shape == b.shape == (n_it, 2)
    assert np.allclose(a, b, rtol=1e-12, atol=1e-9)


def test_two_rank_python_main_equals_two_rank_reference_run(golden_dir, tmp_path):
    """`python -m rf_inv_amd.run` as two ranks (gloo control messages, both on the one GPU of the test
    box) against the reference's own pt_control + output_results under `mpiexec -np 2`: per-rank random
    streams, the cross-rank temperature exchange and the merge of the two ranks' device accumulators give
    the same twelve files."""
    import sys

    mpiexec = "/opt/conda/bin/mpiexec"
    if not os.path.exists(RFINV) or not os.path.exists(mpiexec):
        pytest.skip("drive_rfinv or mpiexec not available")
    nburn, niter = 30, 120

    def prepare(name):
        work = tmp_path / name
        shutil.copytree(os.path.join(golden_dir, "sample_syn"), work)
        os.makedirs(work / "rslt")
        txt = open(work / "params.in").read().splitlines()
        vals = [i for i, line in enumerate(txt) if line.strip() and not line.startswith("#")]
        txt[vals[1]], txt[vals[2]] = str(nburn), str(niter)       # N_BURN, N_ITER follow the output directory
        open(work / "params.in", "w").write("\n".join(txt) + "\n")
        return work

    ref_dir = prepare("ref")
    r = subprocess.run([mpiexec, "-np", "2", RFINV, "params.in", str(nburn), str(niter), "0", "out"], cwd=ref_dir,
                       env=dict(os.environ), capture_output=True, text=True, timeout=900)
    if r.returncode != 0 and ("hydra" in r.stderr.lower() or "unable" in r.stderr.lower()):
        pytest.skip("mpiexec cannot start processes here: " + r.stderr[-200:])
    assert r.returncode == 0 and r.stdout.count("drive_rfinv: ok") == 2, r.stdout + r.stderr

    our_dir = prepare("ours")
    port = 29500 + os.getpid() % 1000
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RF_INV_BACKEND="gloo", PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, "-m", "rf_inv_amd.run", "params.in"], cwd=our_dir, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [pr.communicate(timeout=900)[0] for pr in procs]
    assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)
    _compare_result_dirs(our_dir / "rslt", ref_dir / "rslt", nburn + niter, sigma_solved=False)


def _run_post_merge(tmp_path, world, devices, rccl_library=None, sequential=False):
    sys_path_tools = os.path.join(ROOT, "tests", "tools")
    import sys

    sys.path.insert(0, sys_path_tools)
    import post_merge_worker as wk

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(sys_path_tools, "post_merge_worker.py"), str(r), str(world),
                               str(tmp_path), str(devices[r]), rccl_library or "-"] + (["sequential"] if sequential else []),
                              env=env, cwd=ROOT)
             for r in range(world)]
    for q in procs:
        assert q.wait(timeout=600) == 0
    own = [np.load(tmp_path / f"own_{r}.npz") for r in range(world)]
    merged = [np.load(tmp_path / f"merged_{r}.npz") for r in range(world)]
    after = [np.load(tmp_path / f"after_{r}.npz") for r in range(world)]
    nm = wk.MAX_MODELS
    counts = np.array([int(o["nmod"]) for o in own])
    assert len(set(counts)) == world and counts[-1] > nm > counts[0]         # ragged, one rank overflowing
    assert (world == 3) == (counts.min() == 0)                                # three ranks: one has nothing to send
    m0 = merged[0]
    assert int(m0["nmod"]) == counts.sum()                                    # mpi_reduce of nmod, src/mcmc_out.f90:52
    assert np.array_equal(m0["nmod_rank"], counts)
    assert int(m0["amp_out_of_range"]) == sum(int(o["amp_out_of_range"]) for o in own) > 0
    for f in INT_FIELDS:                                                      # :58-71
        want = sum(o[f].astype(np.int64) for o in own)
        assert np.array_equal(m0[f], want), f
        assert want.any()
    for f in ("vp_mean", "vs_mean", "vpvs_mean"):                             # :74-79 (rank order; exact for 2 ranks)
        want = np.zeros_like(own[0][f])
        for o in own:
            want = want + o[f]
        assert np.allclose(m0[f], want, rtol=4e-16, atol=0), f
        if world == 2:
            assert np.array_equal(m0[f], want), f
    # :88-93 -- rank blocks of max_models rows in rank order; unused rows as init_pt_mcmc leaves them
    for f, width in (("vp_model", 37), ("vs_model", 37), ("all_likelihood", None)):
        got = m0[f]
        assert got.shape[0] == world * nm
        for r in range(world):
            rows = min(counts[r], nm)
            blk = got[r * nm:(r + 1) * nm]
            assert np.array_equal(blk[:rows], own[r][f][:rows]), (f, r)
            rest = blk[rows:]
            if f == "vs_model":
                assert np.all(rest[:, 0] == -999.9) and not rest[:, 1:].any()
            else:
                assert not rest.any()
    # the root's device accumulators hold the sums (its model rows and count stay its own); other ranks are untouched
    for f in INT_FIELDS + ("vp_mean", "vs_mean", "vpvs_mean"):
        assert np.array_equal(after[0][f], m0[f]), f
    assert int(after[0]["nmod"]) == counts[0]
    assert np.array_equal(after[0]["vp_model"], own[0]["vp_model"])
    for r in range(1, world):
        for f in own[r].files:
            assert np.array_equal(after[r][f], own[r][f]), (r, f)
            if f != "nmod_rank":
                assert np.array_equal(merged[r][f], own[r][f]), (r, f)


@pytest.mark.parametrize("world,sequential", [(2, False), (3, False), (3, True)])
def test_posterior_merge_over_the_communicator(tmp_path, world, sequential):
    """rf_comm_post_reduce / rf_comm_post_gather (the mpi_reduce / mpi_gather block of src/mcmc_out.f90:52-93 on the
    device accumulators) with several ranks on the box's one GPU over tests/c/rccl_double.cpp (real RCCL refuses two
    ranks on one device): the merged result on the root is the sum / rank-ordered concatenation of what the ranks held.
    sequential: rf_comm_set_option("sequential_reduce", 1) -- the reduce's calls one by one instead of as one group."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = str(tmp_path / "librccl_double.so")
    subprocess.run([hipcc, "-shared", "-fPIC", "-O2", "-o", lib, os.path.join(ROOT, "tests", "c", "rccl_double.cpp")],
                   check=True, capture_output=True, timeout=300)
    _run_post_merge(tmp_path, world, [0] * world, rccl_library=lib, sequential=sequential)


def test_posterior_merge_between_two_gpus(tmp_path):
    """The same over real RCCL: needs two GPUs."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    _run_post_merge(tmp_path, 2, [0, 1])


@pytest.mark.parametrize("sequential", [False, True])
def test_posterior_merge_entry_points_over_real_rccl_on_one_rank(oracle, golden_dir, sequential):
    """rf_comm_post_gather + rf_comm_post_reduce against the REAL RCCL on a communicator of one rank -- all a one-GPU box
    can form: the twelve in-place ncclReduce calls in one group (and one by one, rf_comm_set_option "sequential_reduce"),
    the all-gather of the model counts, the root's copy-out.  With one rank the merged result is the rank's own; what this
    pins is that RCCL accepts the calls as librfgpu issues them (the multi-rank sums are tested over the test double)."""
    from dataclasses import fields

    from rf_inv_amd import RFEngine
    from rf_inv_amd.posterior import Posterior

    p, ref, mcfg = _setup(golden_dir, 2.0, 0, 10)
    ntrc, n, nm = 2, 40, 60
    p.ntrc, p.nsmp = ntrc, 101
    p.sig_mode, p.sig_min, p.sig_max = [1, 0], [0.005, 0.01], [0.08, 0.01]
    p.amp_min, p.amp_max, p.nbin_amp = -0.05, 0.25, 40
    p.nbin_z, p.nbin_vs, p.nbin_vp, p.nbin_vpvs, p.nbin_sig = 37, 25, 20, 15, 11
    p.nchains, p.niter, p.ncorr = n, 2, 1
    rng = np.random.default_rng(901)
    k, z, dvp, dvs = _valid_states(oracle, rng, p, mcfg, n)
    sig = np.stack([rng.uniform(0.005, 0.0799, n), np.full(n, 0.01)], axis=1).copy()
    pad = p.k_max + 2
    nlay, lay = _layers(oracle, mcfg, k, z, dvp, dvs, pad)
    obs = np.random.default_rng(7).normal(0, 0.05, (ntrc, 101))
    ids = np.arange(n, dtype=np.int32)
    with RFEngine(nfft=256, delta=DELTA, t_start=0.0, deconv_mode=0, sdep=2.0, rayps=[0.06, 0.075], a_gus=[4.0, 3.0],
                  ipha=[1, 1], obs=obs, nsmp=101, max_walkers=n, nlay_max=pad) as eng:
        logl = eng.eval_batch(ids, nlay, lay, sig)
        eng.commit(ids, np.ones(n, dtype=np.int32))
        eng.set_model(p, ref)
        post = Posterior(eng, p, max_models=nm)
        post.record(ids, k, z, dvp, dvs, sig, logl)
        own = post.read()
        eng.comm_init(RFEngine.comm_unique_id(), 0, 1)
        assert eng.comm_info()["nranks"] == 1 and eng.comm_info()["rccl_version"]
        if sequential:
            eng.comm_set_option("sequential_reduce", 1)
        with pytest.raises(Exception, match="unknown option"):
            eng.comm_set_option("no_such_option", 1)
        merged = post.merge_over_comm(root=0)
        after = post.read()
        eng.comm_destroy()
    assert list(merged.nmod_rank) == [own.nmod] and merged.nmod == own.nmod == n
    for f in fields(own):
        if f.name in ("nmod_rank", "vp_model", "vs_model", "all_likelihood"):
            continue
        assert np.array_equal(np.asarray(getattr(merged, f.name)), np.asarray(getattr(own, f.name))), f.name
        assert np.array_equal(np.asarray(getattr(after, f.name)), np.asarray(getattr(own, f.name))), f.name
    rows = min(own.nmod, nm)
    assert np.array_equal(merged.vp_model[:rows], own.vp_model[:rows]) and np.array_equal(merged.vs_model[:rows], own.vs_model[:rows])
    assert np.array_equal(merged.all_likelihood[:rows], own.all_likelihood[:rows])
