"""The bind(C) Fortran shim modules `forward` / `likelihood` (rf_inv_amd/fortran) driven by
the reference's own host modules (params, mt19937, model ... compiled unmodified into
oracle/_ref/ by rf_inv_amd/fortran/Makefile) on the shipped sample_syn params.in."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import logl_tol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "drive_shim")


def test_shim_sources_cite_and_export_the_reference_interface():
    """Static check (no GPU): the shim modules export exactly the reference's public names."""
    fwd = open(os.path.join(ROOT, "rf_inv_amd", "fortran", "forward.f90")).read()
    lik = open(os.path.join(ROOT, "rf_inv_amd", "fortran", "likelihood.F90")).read()
    for name in ("module forward", "flt(:,:)", "is_ray_common", "subroutine init_forward(verb)",
                 "subroutine calc_rf(chain_id, nlay, n, ntrc, rayps, alpha, beta, rho, h, rft)"):
        assert name in fwd, name
    for name in ("module likelihood", "sig(:,:)", "rft(:,:,:)", "log_likelihood(:)",
                 "subroutine init_likelihood(verb)", "subroutine calc_likelihood(chain_id, fwd_flag, prop_k, prop_z"):
        assert name in lik, name


def _parse_dump(path):
    tok = open(path).read().split()
    it = iter(tok)
    nx = lambda: next(it)
    nchains, ntrc, nfft, nsmp, k_max = (int(nx()) for _ in range(5))
    delta = float(nx())
    common = int(nx())
    chains = []
    for _ in range(nchains):
        nlay = int(nx())
        lay = np.array([[float(nx()) for _ in range(4)] for _ in range(nlay)])
        sig = np.array([float(nx()) for _ in range(ntrc)])
        ll = float(nx())
        rft = np.array([[float(nx()) for _ in range(nfft)] for _ in range(ntrc)])
        ll2 = float(nx())
        same_trace = int(nx())
        same_rf = int(nx())
        chains.append(dict(nlay=nlay, lay=lay, sig=sig, ll=ll, rft=rft, ll2=ll2, same_trace=same_trace, same_rf=same_rf))
    flt = np.array([float(nx()) for _ in range(nfft // 2 + 1)])
    return dict(nchains=nchains, ntrc=ntrc, nfft=nfft, nsmp=nsmp, delta=delta, common=common, chains=chains, flt=flt)


@pytest.mark.gpu
def test_fortran_dropin_modules_on_sample_syn(oracle, golden_dir, tmp_path):
    if not os.path.exists(DRIVER):
        pytest.skip("oracle/_ref/drive_shim not built (no Fortran compiler / reference tree at build time)")
    work = tmp_path / "sample_syn"
    shutil.copytree(os.path.join(golden_dir, "sample_syn"), work)
    os.makedirs(work / "rslt")
    env = dict(os.environ)   # the driver carries rpaths for librfgpu and MPICH; keep the environment as is
    r = subprocess.run([DRIVER, "params.in"], cwd=work, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "drive_shim: ok" in r.stdout, r.stdout + r.stderr
    d = _parse_dump(work / "shim_dump.txt")
    assert (d["nchains"], d["ntrc"], d["nfft"], d["nsmp"]) == (5, 2, 256, 101)
    assert d["delta"] == float(np.float32(0.05)) and d["common"] == 0

    from rf_inv_amd.engine import compute_r_inv

    cfg = dict(nfft=256, deconv_mode=0, delta=d["delta"], t_start=0.0, sdep=2.0,
               rayps=np.array([0.06, 0.08]), a_gus=np.array([4.0, 4.0]), ipha=np.array([1, 1], dtype=np.int32))
    obs = np.stack([oracle.read_sac(os.path.join(golden_dir, "sample_syn", "data", f), 0.0, 5.0)[0]
                    for f in ("sample_1.trc", "sample_2.trc")])
    r1, rank = compute_r_inv(101, 4.0, d["delta"])   # the shim's default: librfgpu's own init_r_inv
    r_inv = np.stack([r1, r1])
    assert np.allclose(d["flt"], oracle.init_filter(256, d["delta"], [4.0])[0], rtol=1e-15, atol=0)
    for c in d["chains"]:
        a, b, rho, h = c["lay"].T
        assert b[0] == -999.0 and a[0] == 1.5 and h[0] == 2.0       # ocean layer from the host's format_model
        ref = oracle.calc_rf(cfg, a, b, rho, h)
        assert np.abs(c["rft"] - ref).max() <= 1e-12 * np.abs(ref).max()
        ll = oracle.log_likelihood(ref, obs, r_inv, c["sig"], 101)
        assert abs(c["ll"] - ll) <= logl_tol(ll), (c["ll"], ll)
        ll2 = oracle.log_likelihood(ref, obs, r_inv, 2 * c["sig"], 101)
        assert abs(c["ll2"] - ll2) <= logl_tol(ll2)
        assert c["same_trace"] == 1 and c["same_rf"] == 1


RFINV = os.path.join(ROOT, "oracle", "_ref", "drive_rfinv")
RESULT_FILES = ["all_models", "likelihood", "num_interface.ppd", "syn_trace.ppd", "interface_depth.ppd", "sigma.ppd",
                "vs_z.ppd", "vp_z.ppd", "vpvs_z.ppd", "vs_z.mean", "vp_z.mean", "vpvs_z.mean"]


@pytest.mark.gpu
def test_reference_sampler_on_top_of_the_dropin_modules(oracle, golden_dir, tmp_path):
    """The reference's OWN pt_mcmc.f90 (compiled unmodified) running its sequential PT RJ-MCMC
    loop on top of our forward / likelihood modules on the GPU, shipped sample_syn params.in.
    Checked against (a) the value SURVEY.md section 8c(4) records from a pure-reference run and
    (b) the batched Python driver (rf_inv_amd/mcmc.py) on the same engine: identical
    trajectory (every accept/reject decision) over 300 iterations."""
    if not os.path.exists(RFINV):
        pytest.skip("oracle/_ref/drive_rfinv not built (no Fortran compiler / reference tree at build time)")
    n_it = 300
    work = tmp_path / "sample_syn"
    shutil.copytree(os.path.join(golden_dir, "sample_syn"), work)
    os.makedirs(work / "rslt")
    r = subprocess.run([RFINV, "params.in", "0", str(n_it), "0"], cwd=work, env=dict(os.environ),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "drive_rfinv: ok" in r.stdout, r.stdout + r.stderr
    tok = iter(open(work / "rfinv_dump.txt").read().split())
    n, ntype, ncool = int(next(tok)), int(next(tok)), int(next(tok))
    hist = np.array([float(next(tok)) for _ in range(n)])
    nprop = np.array([int(next(tok)) for _ in range(ntype)])
    nacc = np.array([int(next(tok)) for _ in range(ntype)])
    assert (n, ntype, ncool) == (n_it, 4, 1)
    assert abs(hist[0] - (-1044.33907794324)) < 1e-7     # R^-1 here is librfgpu's own SVD, not LAPACK

    from rf_inv_amd import RFEngine, get_params, read_obs, read_ref_model
    from rf_inv_amd.mcmc import RJMCMC, EngineEvaluator
    from rf_inv_amd.mt19937 import MT19937

    p = get_params(os.path.join(golden_dir, "sample_syn", "params.in"))
    read_obs(p)
    p.nburn, p.niter = 0, n_it
    ref = read_ref_model(os.path.join(p.base_dir, p.vel_file))
    with RFEngine.from_params(p) as eng:               # same default R^-1 as the Fortran shim
        m = RJMCMC(p, ref, EngineEvaluator(eng, p.k_max + 2), MT19937(p.iseed))
        m.init_model(); m.init_likelihood(); m.init_pt_mcmc()
        vals = []
        for it in range(1, n_it + 1):
            m.iterate(it)
            vals.append(m.mean_t1_likelihood(it))
    assert np.array_equal(m.counters.nprop[1:ntype + 1], nprop)
    assert np.array_equal(m.counters.naccept[1:ntype + 1], nacc)
    assert np.allclose(hist, vals, rtol=1e-12, atol=1e-9)


@pytest.mark.gpu
def test_fortran_batched_sampler_equals_reference_sampler(golden_dir, tmp_path):
    """rf_inv_amd/fortran/pt_mcmc_batched.f90 -- mode 1: its default two-segment pipeline (one half of the chains is
    being evaluated through rf_eval_models_begin / rf_eval_wait while the host judges and re-proposes the other, the
    swap's draws made at their place in the stream and its decision once both halves are through); mode 2: propose
    all -> evaluate all -> judge all -- against
    the reference's own sequential pt_control on the same GPU engine: the complete dumps --
    likelihood history, proposal/accept counters, posterior checksums, final temperatures and
    log-likelihoods -- are identical, burn-in and recording phases included; and so is every result
    file the reference's own output_results (src/mcmc_out.f90, compiled unmodified) writes: mode 0
    fills the histograms on the host as the reference does (src/pt_mcmc.f90:204-286), mode 1 on the
    device (rf_post_record), traces never leaving HBM."""
    if not os.path.exists(RFINV):
        pytest.skip("oracle/_ref/drive_rfinv not built (no Fortran compiler / reference tree at build time)")
    dumps = []
    for mode in ("0", "1", "2"):
        work = tmp_path / f"run{mode}"
        shutil.copytree(os.path.join(golden_dir, "sample_syn"), work)
        os.makedirs(work / "rslt")
        r = subprocess.run([RFINV, "params.in", "60", "240", mode, "out"], cwd=work, env=dict(os.environ),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "drive_rfinv: ok" in r.stdout, r.stdout + r.stderr
        dumps.append(open(work / "rfinv_dump.txt").read())
    assert len(dumps[0].split()) > 300
    assert dumps[0] == dumps[1] and dumps[0] == dumps[2]
    for name in RESULT_FILES:
        a, b = open(tmp_path / "run0" / "rslt" / name).read(), open(tmp_path / "run1" / "rslt" / name).read()
        assert a == b, name
        assert a == open(tmp_path / "run2" / "rslt" / name).read(), name
        assert len(a) > 0 or name == "sigma.ppd", name


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 3])
def test_fortran_batched_sampler_several_mpi_ranks(golden_dir, tmp_path, nranks):
    """Several MPI ranks (all on the one GPU of the test box) against the reference's pt_control, rank by rank and
    result file by result file: the cross-rank temperature exchange of pt_control_batched with its two-segment pipeline
    (mode 1) and without it (mode 2); every rank launches for itself.  Same trajectories, same files."""
    mpiexec = "/opt/conda/bin/mpiexec"
    if not os.path.exists(RFINV) or not os.path.exists(mpiexec):
        pytest.skip("drive_rfinv or mpiexec not available")
    dumps = {}
    for mode in ("0", "1", "2"):
        work = tmp_path / f"mpi{mode}"
        shutil.copytree(os.path.join(golden_dir, "sample_syn"), work)
        os.makedirs(work / "rslt")
        r = subprocess.run([mpiexec, "-np", str(nranks), RFINV, "params.in", "40", "160", mode, "out"], cwd=work,
                           env=dict(os.environ), capture_output=True, text=True, timeout=900)
        if r.returncode != 0 and ("hydra" in r.stderr.lower() or "unable" in r.stderr.lower()):
            pytest.skip("mpiexec cannot start processes here: " + r.stderr[-200:])
        assert r.returncode == 0 and r.stdout.count("drive_rfinv: ok") == nranks, r.stdout + r.stderr
        dumps[mode] = [open(work / f"rfinv_dump_{k}.txt").read() for k in range(nranks)]
    for mode in ("1", "2"):
        assert dumps[mode] == dumps["0"], mode
        for name in RESULT_FILES:   # output_results' mpi_reduce / mpi_gather over the ranks' accumulators
            assert open(tmp_path / "mpi0" / "rslt" / name).read() == open(tmp_path / f"mpi{mode}" / "rslt" / name).read(), (mode, name)
    assert dumps["0"][0] != dumps["0"][1]      # the ranks run different chains (seed depends on rank)
    # temperatures moved between ranks at least once: rank 0 started with [1, tempered...]
    t0 = [float(x) for x in dumps["0"][0].split()[-5:]]
    assert len(t0) == 5 and all(x >= 1.0 for x in t0)


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 8])
def test_fortran_batched_sampler_two_mpi_ranks_over_the_rccl_entry_points(golden_dir, tmp_path, nranks):
    """(nranks = 8: the rank count of the driver's 8-GPU node, 40 chains in all.)
    The same ranks with the exchange on librfgpu's communicator -- the branch a one-GPU-per-rank run takes
    (open_temperature_exchange: rf_comm_probe / rf_comm_get_unique_id / mpi_bcast of the id / rf_comm_init;
    propose_temperature_swap: rf_comm_bcast_i32 + rf_pt_swap_exchange) -- over the RCCL test double
    (tests/c/rccl_double.cpp; real RCCL refuses two ranks on the one GPU of the box): rank by rank and file by file
    equal to the reference's pt_control under `mpiexec -np 2`."""
    mpiexec = "/opt/conda/bin/mpiexec"
    if not os.path.exists(RFINV) or not os.path.exists(mpiexec):
        pytest.skip("drive_rfinv or mpiexec not available")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = str(tmp_path / "librccl_double.so")
    subprocess.run([hipcc, "-shared", "-fPIC", "-O2", "-o", lib, os.path.join(ROOT, "tests", "c", "rccl_double.cpp")],
                   check=True, capture_output=True, timeout=300)
    dumps = []
    for mode in ("0", "1"):
        work = tmp_path / f"mpi{mode}"
        shutil.copytree(os.path.join(golden_dir, "sample_syn"), work)
        os.makedirs(work / "rslt")
        r = subprocess.run([mpiexec, "-np", str(nranks), RFINV, "params.in", "40", "160", mode, "out"] + ([lib] if mode == "1" else []),
                           cwd=work, env=dict(os.environ), capture_output=True, text=True, timeout=1500)
        if r.returncode != 0 and ("hydra" in r.stderr.lower() or "unable" in r.stderr.lower()):
            pytest.skip("mpiexec cannot start processes here: " + r.stderr[-200:])
        assert r.returncode == 0 and r.stdout.count("drive_rfinv: ok") == nranks, r.stdout + r.stderr
        if mode == "1":
            assert f"Temperature exchange: RCCL ({nranks} ranks" in r.stderr, r.stderr[-500:]
        dumps.append([open(work / f"rfinv_dump_{k}.txt").read() for k in range(nranks)])
    assert dumps[0] == dumps[1]
    for name in RESULT_FILES:
        assert open(tmp_path / "mpi0" / "rslt" / name).read() == open(tmp_path / "mpi1" / "rslt" / name).read(), name


@pytest.mark.gpu
def test_fortran_batched_sampler_many_chains(golden_dir, tmp_path):
    """600 chains, 100 of them non-tempered: the batched sampler's evaluation goes through the large-batch
    launch plan (misfits to HBM, quadratic forms and logL by follow-up kernels) while the reference's
    pt_control evaluates chain by chain inside the fused kernel.  Both arithmetic paths are the same
    operation for operation, so the dumps and the result files are identical."""
    if not os.path.exists(RFINV):
        pytest.skip("oracle/_ref/drive_rfinv not built (no Fortran compiler / reference tree at build time)")
    outs = []
    for mode in ("0", "1"):
        work = tmp_path / f"big{mode}"
        shutil.copytree(os.path.join(golden_dir, "sample_syn"), work)
        os.makedirs(work / "rslt")
        txt = open(work / "params.in").read().splitlines()
        vals = [i for i, line in enumerate(txt) if line.strip() and not line.lstrip().startswith("#")]
        txt[vals[4]], txt[vals[5]] = "600", "100"          # N_CHAINS, N_COOL
        open(work / "params.in", "w").write("\n".join(txt) + "\n")
        r = subprocess.run([RFINV, "params.in", "10", "30", mode, "out"], cwd=work, env=dict(os.environ),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "drive_rfinv: ok" in r.stdout, r.stdout + r.stderr
        outs.append([open(work / "rfinv_dump.txt").read()] + [open(work / "rslt" / n).read() for n in RESULT_FILES])
    assert outs[0] == outs[1]
    assert int(outs[0][0].split()[2]) == 100               # ncool


# ---------------------------------------------------------------------------------------------------------------
# The reference's OWN main programs, compiled unmodified from /root/reference/src/{rf_inv,make_syn}.f90 and linked
# with the object lists of the reference Makefile (Makefile:28-35) in which fftw, forward and likelihood are the
# drop-in modules of rf_inv_amd/fortran (rf_inv_amd/fortran/Makefile): "no line of the host changes" as a test.
# ---------------------------------------------------------------------------------------------------------------
REF_RFINV = os.path.join(ROOT, "oracle", "_ref", "rf_inv")
REF_MAKESYN = os.path.join(ROOT, "oracle", "_ref", "make_syn")
MPIEXEC = "/opt/conda/bin/mpiexec"


def test_fftw_dropin_exports_the_reference_names():
    """Static check (no GPU): module fftw keeps the reference's public names and kinds (src/fftw.f90:28-48), includes
    no FFTW header, and the legacy entry point make_syn.f90 calls on the two plans exists."""
    src = open(os.path.join(ROOT, "rf_inv_amd", "fortran", "fftw.f90")).read()
    code = "\n".join(l.split("!")[0] for l in src.splitlines())
    for name in ("module fftw", "use params, only: nfft", "complex(kind(0d0)), allocatable :: cx(:)",
                 "real(kind(0d0)), allocatable :: rx(:)", "integer(8) :: ifft, ifft2", "subroutine init_fftw()",
                 "subroutine dfftw_execute(plan)"):
        assert name in code, name
    assert "fftw3.f" not in code and "dfftw_plan" not in code


def _sample_syn_with_iterations(golden_dir, work, nburn, niter):
    """A copy of the shipped sample_syn directory whose params.in asks for nburn + niter iterations."""
    shutil.copytree(os.path.join(golden_dir, "sample_syn"), work)
    os.makedirs(work / "rslt")
    lines = open(work / "params.in").read().split("\n")
    i = next(j for j, l in enumerate(lines) if l.startswith("# N_BURN"))
    assert lines[i + 1].strip() == "3000" and lines[i + 2].startswith("# N_ITER") and lines[i + 3].strip() == "8000"
    lines[i + 1], lines[i + 3] = str(nburn), str(niter)
    open(work / "params.in", "w").write("\n".join(lines))


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [1, 2])
def test_the_reference_main_program_runs_unmodified_on_the_dropin_modules(golden_dir, tmp_path, nranks):
    """bin/rf_inv as the reference builds it -- src/rf_inv.f90 (`use fftw`, `call init_fftw()`, init_forward ...
    pt_control, output_results: src/rf_inv.f90:28-107) with pt_mcmc.f90, mcmc_out.f90 and the host modules, all
    unmodified -- on module fftw / forward / likelihood of rf_inv_amd/fortran: runs under mpiexec with 1 and 2 ranks,
    writes the 12 result files, and they are byte-identical to those of the look-alike driver tests/fortran/drive_rfinv
    (mode 0: the same pt_control) that the other tests of this file use; iteration 1 of rslt/likelihood is the value a
    pure-reference run recorded."""
    if not (os.path.exists(REF_RFINV) and os.path.exists(RFINV) and os.path.exists(MPIEXEC)):
        pytest.skip("oracle/_ref/rf_inv not built (no Fortran compiler / reference tree at build time) or no mpiexec")
    nburn, niter = 60, 240
    runs = {}
    for tag, cmd in (("main", [REF_RFINV, "params.in"]),
                     ("driver", [RFINV, "params.in", str(nburn), str(niter), "0", "out"])):
        work = tmp_path / tag
        _sample_syn_with_iterations(golden_dir, work, nburn, niter)
        r = subprocess.run([MPIEXEC, "-np", str(nranks)] + cmd, cwd=work, env=dict(os.environ), capture_output=True,
                           text=True, timeout=900)
        if r.returncode != 0 and ("hydra" in r.stderr.lower() or "unable" in r.stderr.lower()):
            pytest.skip("mpiexec cannot start processes here: " + r.stderr[-200:])
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        runs[tag] = r
    # the main program's own stdout: the reference's verbose init messages and iteration counter (rank 0)
    assert "--- Reading observed data ---" in runs["main"].stdout and "Iteration" in runs["main"].stdout
    for name in RESULT_FILES:
        a = open(tmp_path / "main" / "rslt" / name, "rb").read()
        assert a == open(tmp_path / "driver" / "rslt" / name, "rb").read(), name
        assert len(a) > 0 or name == "sigma.ppd", name
    lk = np.loadtxt(tmp_path / "main" / "rslt" / "likelihood")
    assert lk.shape == (nburn + niter, 2) and np.array_equal(lk[:, 0], np.arange(1, nburn + niter + 1))
    if nranks == 1:
        assert abs(lk[0, 1] - (-1044.33907794324)) < 1e-7


@pytest.mark.gpu
def test_the_reference_make_syn_runs_unmodified_on_the_dropin_modules(golden_dir, tmp_path):
    """bin/make_syn as the reference builds it (src/make_syn.f90, unmodified; Makefile:32-35): it executes the two FFTW
    plans of module fftw itself -- dfftw_execute(ifft2), a product with flt, dfftw_execute(ifft), src/make_syn.f90:91-95,
    107-111 -- which the drop-in module runs on the GPU (rf_fft_r2c / rf_fft_c2r).  Its files against the Python
    mirror of the program (rf_inv_amd.make_syn.make_syn_program: same stream, same engine, same transforms): the
    noise-free and the noisy SAC files byte for byte, test_vel value for value."""
    if not os.path.exists(REF_MAKESYN):
        pytest.skip("oracle/_ref/make_syn not built (no Fortran compiler / reference tree at build time)")
    from rf_inv_amd import RFEngine, get_params, read_obs, read_ref_model
    from rf_inv_amd.make_syn import make_syn_program

    work = tmp_path / "fortran"
    shutil.copytree(os.path.join(golden_dir, "sample_syn"), work)
    os.makedirs(work / "rslt")          # get_params copies params.in into the output directory (src/params.f90)
    r = subprocess.run([REF_MAKESYN, "params.in"], cwd=work, env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Noise level of trace" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    p = get_params(os.path.join(golden_dir, "sample_syn", "params.in"))
    read_obs(p)
    ref = read_ref_model(os.path.join(p.base_dir, p.vel_file))
    with RFEngine.from_params(p, max_walkers=p.nchains) as eng:
        out = make_syn_program(p, ref, eng, str(tmp_path / "python"))
    said = [float(l.split(":")[-1]) for l in r.stdout.splitlines() if "Noise level of trace" in l]
    assert np.allclose(said, out["noise_sigma"], rtol=1e-12, atol=0)
    for t in range(1, p.ntrc + 1):
        for name in (f"test_trace{t:02d}", f"test_trace{t:02d}wn"):
            a, b = open(work / name, "rb").read(), open(tmp_path / "python" / name, "rb").read()
            assert len(a) == 4 * (158 + p.nsmp)
            assert a == b, name
    assert np.array_equal(np.loadtxt(work / "test_vel"), np.loadtxt(tmp_path / "python" / "test_vel"))


# the WHOLE reference on the CPU (oracle/Makefile.ref: all twelve sources unmodified, its own module fftw on MKL's FFTW3
# interface, dgesvd from MKL; no product code, no GPU) -- round 5 ran the reference's modules on the drop-in module fftw here
REF_OWN = os.path.join(ROOT, "oracle", "_ref", "cpu_o2", "rf_inv")
DROPIN_LAPACK = os.path.join(ROOT, "oracle", "_ref", "rf_inv_lapack")


@pytest.mark.gpu
@pytest.mark.parametrize("nranks,nburn,niter", [(1, 60, 240), (2, 60, 240), (1, 3000, 8000)])
def test_dropin_run_equals_the_run_of_the_reference_on_its_own_modules(golden_dir, tmp_path, nranks, nburn, niter):
    """The whole program, end to end, three ways on the shipped sample_syn directory (iteration counts reduced, and --
    one rank -- the shipped example at its full length: 3000 + 8000 iterations x 5 chains, 55 000 evaluations):
      reference : the reference itself, all twelve sources unmodified, on the host CPU (oracle/_ref/cpu_o2/rf_inv: its own
                  module fftw on MKL's FFTW3 interface, dgesvd from MKL; no product code in it)
      drop-in   : the same main program and host modules on module forward / likelihood of rf_inv_amd/fortran (the HIP
                  kernels; R^-1 by librfgpu's own SVD)
      drop-in + LAPACK : the drop-in built with -DRFGPU_USE_LAPACK (R^-1 by the host's dgesvd, as the reference forms it)
    A chain's random stream depends on every accept / reject decision, i.e. on every log-likelihood: the three runs take
    the SAME trajectory -- the eleven result files that hold models, histograms and means are byte-identical -- and
    rslt/likelihood (the mean log-likelihood of the non-tempered chains per iteration, printed to 17 digits) agrees to
    1e-11 relative (LAPACK variant: the kernels' rounding only; default: plus two SVDs' rounding in R^-1)."""
    if not all(os.path.exists(x) for x in (REF_OWN, REF_RFINV, DROPIN_LAPACK, MPIEXEC)):
        pytest.skip("oracle/_ref mains not built (no Fortran compiler / reference tree / MKL at build time) or no mpiexec")
    for tag, exe in (("reference", REF_OWN), ("dropin", REF_RFINV), ("dropin_lapack", DROPIN_LAPACK)):
        work = tmp_path / tag
        _sample_syn_with_iterations(golden_dir, work, nburn, niter)
        r = subprocess.run([MPIEXEC, "-np", str(nranks), exe, "params.in"], cwd=work, env=dict(os.environ),
                           capture_output=True, text=True, timeout=1800)
        if r.returncode != 0 and ("hydra" in r.stderr.lower() or "unable" in r.stderr.lower()):
            pytest.skip("mpiexec cannot start processes here: " + r.stderr[-200:])
        assert r.returncode == 0, (tag, r.stdout[-1500:] + r.stderr[-1500:])
    lk = {t: np.loadtxt(tmp_path / t / "rslt" / "likelihood") for t in ("reference", "dropin", "dropin_lapack")}
    assert lk["reference"].shape == (nburn + niter, 2)
    if nranks == 1:
        assert abs(lk["reference"][0, 1] - (-1044.33907794324)) < 1e-7       # SURVEY.md 8c(4), a pure-reference run's record
    for t in ("dropin", "dropin_lapack"):
        for name in RESULT_FILES:
            if name == "likelihood":
                continue
            a = open(tmp_path / "reference" / "rslt" / name, "rb").read()
            assert a == open(tmp_path / t / "rslt" / name, "rb").read(), (t, name)
        rel = np.abs(lk[t][:, 1] - lk["reference"][:, 1]) / np.abs(lk["reference"][:, 1])
        assert np.array_equal(lk[t][:, 0], lk["reference"][:, 0]) and rel.max() <= 1e-11, (t, rel.max())
    rel_lap = (np.abs(lk["dropin_lapack"][:, 1] - lk["reference"][:, 1]) / np.abs(lk["reference"][:, 1])).max()
    rel_def = (np.abs(lk["dropin"][:, 1] - lk["reference"][:, 1]) / np.abs(lk["reference"][:, 1])).max()
    print(f"{nranks} rank(s), {nburn + niter} iterations: max relative difference of rslt/likelihood to the reference's own run: "
          f"drop-in + LAPACK {rel_lap:.2e}, drop-in {rel_def:.2e}")


@pytest.mark.gpu
def test_make_syn_on_the_dropin_modules_equals_make_syn_on_the_reference_modules(golden_dir, tmp_path):
    """`program make_syn` (src/make_syn.f90, unmodified) twice: the reference itself on the CPU (oracle/_ref/cpu_o2/make_syn:
    every source the reference's, FFTW3 interface and LAPACK from MKL) and on the drop-in modules (oracle/_ref/make_syn).  Same random stream, same model,
    same noise: the SAC files -- float32 samples of chain 1's synthetic trace, with and without noise -- are byte-identical
    (the CPU and the GPU traces agree to 1e-14 of their scale), test_vel line for line."""
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "cpu_o2", "make_syn")
    if not (os.path.exists(ref_exe) and os.path.exists(REF_MAKESYN)):
        pytest.skip("oracle/_ref/cpu_o2/make_syn not built (no Fortran compiler / reference tree / MKL at build time)")
    outs = {}
    for tag, exe in (("reference", ref_exe), ("dropin", REF_MAKESYN)):
        work = tmp_path / tag
        shutil.copytree(os.path.join(golden_dir, "sample_syn"), work)
        os.makedirs(work / "rslt")
        r = subprocess.run([exe, "params.in"], cwd=work, env=dict(os.environ), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "Noise level of trace" in r.stdout, (tag, r.stdout[-1500:] + r.stderr[-1500:])
        outs[tag] = [l for l in r.stdout.splitlines() if "Noise level" in l]
    assert outs["reference"] == outs["dropin"]
    for name in ("test_trace01", "test_trace01wn", "test_trace02", "test_trace02wn", "test_vel"):
        a, b = open(tmp_path / "reference" / name, "rb").read(), open(tmp_path / "dropin" / name, "rb").read()
        assert len(a) > 0 and a == b, name


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["c4", "c5"])
def test_dropin_runs_equal_the_reference_run_at_the_benchmark_shapes(tmp_path, shape):
    """End to end at the north-star geometry, not only on the shipped nfft-256 example: run directories with the shape of
    BASELINE configs 4 and 5 (tests/tools/shape_run.py: nfft 4096; c4 = traces P .06 / P .08 / S .10, k_max 30; c5 = 2 km
    of water, traces P / P / S / S; 8 chains, 2 of them at T = 1, 100 iterations) through
      (a) the reference itself on the CPU (oracle/_ref/cpu_o2/rf_inv: all of its sources, MKL's FFTW3 interface and LAPACK),
      (b) the same main program on the drop-in modules (oracle/_ref/rf_inv_lapack: per-call evaluation on the GPU),
      (c) the batched sampler pt_control_batched on the drop-in modules (tests/fortran/drive_rfinv mode 1).
    Same trajectory in all three: the eleven model / histogram / mean files byte-identical, rslt/likelihood to 1e-11."""
    import sys

    if not all(os.path.exists(x) for x in (REF_OWN, DROPIN_LAPACK, RFINV)):
        pytest.skip("oracle/_ref mains not built (no Fortran compiler / reference tree / MKL at build time)")
    runs = {"reference": [REF_OWN, "params.in"], "dropin": [DROPIN_LAPACK, "params.in"],
            "batched": [RFINV, "params.in", "0", "100", "1", "out"]}
    for tag, cmd in runs.items():
        work = tmp_path / tag
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "shape_run.py"), shape, "8", str(work), "2"],
                           capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout + r.stderr
        r = subprocess.run(cmd, cwd=work, env=dict(os.environ), capture_output=True, text=True, timeout=1800)
        assert r.returncode == 0, (tag, r.stdout[-1500:] + r.stderr[-1500:])
    lk = {t: np.loadtxt(tmp_path / t / "rslt" / "likelihood") for t in runs}
    assert lk["reference"].shape == (100, 2) and np.isfinite(lk["reference"]).all()
    for t in ("dropin", "batched"):
        for name in RESULT_FILES:
            if name == "likelihood":
                continue
            a = open(tmp_path / "reference" / "rslt" / name, "rb").read()
            assert a == open(tmp_path / t / "rslt" / name, "rb").read(), (shape, t, name)
        rel = np.abs(lk[t][:, 1] - lk["reference"][:, 1]) / np.abs(lk["reference"][:, 1])
        assert rel.max() <= 1e-11, (shape, t, rel.max())
    print(f"{shape} shape, 8 chains x 100 iterations: rslt/likelihood against the reference's own run: per-call drop-in "
          f"{(np.abs(lk['dropin'][:, 1] - lk['reference'][:, 1]) / np.abs(lk['reference'][:, 1])).max():.2e}, batched sampler "
          f"{(np.abs(lk['batched'][:, 1] - lk['reference'][:, 1]) / np.abs(lk['reference'][:, 1])).max():.2e}")
