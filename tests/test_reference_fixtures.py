"""The committed reference fixtures (tests/golden/ref/, generator: oracle/gen_golden.py).

Every output in them was computed by the reference's OWN twelve source files, compiled unmodified and run on the CPU --
FFTW3's Fortran interface and LAPACK from the image's Intel MKL, no product object on any link line, no GPU
(oracle/Makefile.ref; the one stand-in is the two-line include file oracle/fftw3_include/fftw3.f, so the formal pin
of the oracle stays tests/test_oracle_kat.py's reference-held vectors and these are supplementary: DESIGN.md section 5).

  -m "not gpu":  the CPU oracle (oracle/rf_oracle.c) against the fixtures -- the oracle checked against the reference
                 without a GPU and without any product code
  -m gpu:        the HIP path through the C ABI against the same fixtures

A missing or altered fixture FAILS (never skips): the list of expected files is in this module."""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import logl_tol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "tests", "golden", "ref")
FORWARD = ["c1_ocean_2P_nfft256", "land_S_nfft256_decon", "c2_land_P", "c2d_land_P_decon", "c4_land_PPS",
           "c4_land_PPS_decon_tstart", "c5_ocean_PPSS", "c5_ocean_PPSS_decon", "c4common_land_3P_one_ray",
           "common_ocean_3S_one_ray_nfft2048", "common_land_2S_decon_nfft512", "odd_length_nfft1000_S",
           "long_nfft8192_PS", "bluestein_nfft3000_ocean_P_decon", "reference_forward_test", "evanescent_nan"]
PATH = ["c1", "c2", "c2d", "c4", "c4d", "c4common", "c5", "c5d", "c4w20"]
TRACE_TOL = 1e-12            # of max|trace| per trace (tests/helpers.py)


def load(kind, name):
    path = os.path.join(REF, f"{kind}_{name}.npz")
    assert os.path.exists(path), f"{path} is missing: the reference fixtures are part of the repository (oracle/gen_golden.py)"
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def cfg_of(f):
    return dict(nfft=int(f["nfft"]), deconv_mode=int(f["deconv_mode"]), delta=float(f["delta"]), t_start=float(f["t_start"]),
                sdep=float(f["sdep"]), rayps=f["rayps"].astype(float), a_gus=f["a_gus"].astype(float),
                ipha=f["ipha"].astype(np.int32))


def engine_kw(f):
    c = cfg_of(f)
    return dict(nfft=c["nfft"], delta=c["delta"], t_start=c["t_start"], deconv_mode=c["deconv_mode"], sdep=c["sdep"],
                rayps=c["rayps"], a_gus=c["a_gus"], ipha=c["ipha"], nsmp=int(f["nsmp"]))


def trace_close(got, ref, what, tol=TRACE_TOL):
    """|got - ref| <= tol * max|ref| per trace; NaN traces must be NaN on both sides."""
    got, ref = np.asarray(got), np.asarray(ref)
    nan_ref = np.isnan(ref).all(axis=-1)
    assert np.array_equal(np.isnan(got).all(axis=-1), nan_ref), (what, "NaN traces differ")
    ok = ~nan_ref
    scale = np.abs(ref[ok]).max(axis=-1, keepdims=True)
    err = (np.abs(got[ok] - ref[ok]) / scale).max() if ok.any() else 0.0
    assert np.isfinite(got[ok]).all() and err <= tol, (what, float(err))
    return float(err)


def model_setup(f, tmp_path):
    """params + reference model + the oracle's format_model configuration from the fixture's own params.in text."""
    from rf_inv_amd import get_params, read_ref_model

    pin = tmp_path / "params.in"
    pin.write_text(str(f["params_in"]))
    p = get_params(str(pin))
    refm = read_ref_model(os.path.join(ROOT, "tests", "golden", "sample_syn", "model", "sample.velmod"))
    mcfg = dict(k_max=p.k_max, vp_mode=p.vp_mode, sdep=p.sdep, z_max=p.z_max, h_min=p.h_min, z_ref_min=refm.z_ref_min,
                dz_ref=refm.dz_ref, vp_min=p.vp_min, vp_max=p.vp_max, vs_min=p.vs_min, vs_max=p.vs_max,
                vpvs_min=p.vpvs_min, vpvs_max=p.vpvs_max, vp_ref=refm.vp_ref, vs_ref=refm.vs_ref)
    return p, refm, mcfg


# ---------------------------------------------------------------------------------------------------------------
# CPU: integrity of the fixture set, and the oracle against it
# ---------------------------------------------------------------------------------------------------------------
def test_fixture_set_is_complete_and_unaltered():
    man = json.load(open(os.path.join(REF, "MANIFEST.json")))["files"]
    want = {f"forward_{n}.npz" for n in FORWARD} | {f"path_{n}.npz" for n in PATH} | {"run_sample_syn.npz", "format_model.npz"}
    assert set(man) == want
    for name, rec in man.items():
        raw = open(os.path.join(REF, name), "rb").read()
        assert hashlib.sha256(raw).hexdigest() == rec["sha256"], name


def test_the_cpu_reference_recipe_links_no_product_code():
    """oracle/Makefile.ref: every source compiled comes from $(REF) (or is one of the two dumpers), and no compile or
    link line names the product (librfgpu, rf_inv_amd/, the drop-in Fortran modules)."""
    lines = [l for l in open(os.path.join(ROOT, "oracle", "Makefile.ref")).read().splitlines() if not l.lstrip().startswith("#")]
    body = "\n".join(lines)
    assert "rfgpu" not in body and "rf_inv_amd" not in body and "-lrf" not in body
    assert "$(REF)/$$m.f90" in body and "$(REF)/$$p.f90" in body and "-lmkl_gf_lp64" in body
    inc = [l for l in open(os.path.join(ROOT, "oracle", "fftw3_include", "fftw3.f")).read().splitlines() if not l.startswith("!")]
    assert [l.split() for l in inc] == [["INTEGER", "FFTW_ESTIMATE"], ["PARAMETER", "(FFTW_ESTIMATE=64)"]]


@pytest.mark.parametrize("name", FORWARD)
def test_oracle_calc_rf_against_the_reference_fixture(oracle, name):
    f = load("forward", name)
    cfg = cfg_of(f)
    # the filter table expression for expression (src/forward.f90:95-119), is_ray_common (check_ray, :59-91)
    assert np.array_equal(oracle.init_filter(cfg["nfft"], cfg["delta"], cfg["a_gus"]), f["flt"])
    assert bool(f["is_ray_common"]) == (len(set(cfg["rayps"])) == 1 and len(set(cfg["ipha"])) == 1)
    worst = 0.0
    for i in range(len(f["nlay"])):
        n = int(f["nlay"][i])
        st = tuple(f["layers"][i, r, :n] for r in range(4))
        got, npre, _, _ = oracle.calc_rf(cfg, *st, want_stages=True)
        if np.isfinite(f["tp"][i]).all():
            assert np.array_equal(npre, f["npre"][i]), (name, i, npre, f["npre"][i])           # integer: bit-exact
        worst = max(worst, trace_close(got, f["rft"][i], (name, i)))
    print(f"forward_{name}: oracle against the reference's calc_rf, max |d trace| / max|trace| = {worst:.2e}")


@pytest.mark.parametrize("workload", PATH)
def test_oracle_calc_likelihood_against_the_reference_fixture(oracle, tmp_path, workload):
    f = load("path", workload)
    cfg, nsmp, ntrc = cfg_of(f), int(f["nsmp"]), int(f["ntrc"])
    p, refm, mcfg = model_setup(f, tmp_path)
    n = len(f["k"])
    r_inv = f["r_inv"][f["r_index"]]
    # (0) the pseudo-inverse: the reference's init_r_inv with MKL's dgesvd against scipy's OpenBLAS dgesvd in the oracle's
    # builder -- same rank, same matrix to the rounding of two SVDs of an ill-conditioned matrix
    mine = oracle.build_r_inv(nsmp, cfg["a_gus"], cfg["delta"])
    assert np.abs(mine - r_inv).max() <= 1e-8 * np.abs(r_inv).max()
    # (1) format_model: integer bookkeeping and layer values bit for bit (src/model.f90:175-290)
    pad = f["layers"].shape[2]
    nlay = np.zeros(n, dtype=np.int32)
    layers = np.ones((n, 4, pad))
    for i in range(n):
        nl, a, b, r, h, ok = oracle.format_model(mcfg, int(f["k"][i]), f["z"][i], f["dvp"][i], f["dvs"][i])
        assert ok and nl == f["nlay"][i], (workload, i)
        for row, arr in enumerate((a, b, r, h)):
            assert np.array_equal(arr, f["layers"][i, row, :nl]), (workload, i, row)
            layers[i, row, :nl] = arr
        nlay[i] = nl
    # (2) the sigma-only branch (fwd_flag = .false., src/likelihood.f90:81,87-98) on host-stored traces
    for j in range(len(f["probe_logl"])):
        tr = np.zeros((ntrc, cfg["nfft"]))
        tr[:, :nsmp] = f["probe_window"][j]
        want = oracle.log_likelihood(tr, f["obs"], r_inv, f["probe_sig"][j], nsmp)
        assert abs(want - f["probe_logl"][j]) <= logl_tol(f["probe_logl"][j]), (workload, j, want, f["probe_logl"][j])
    # (3) calc_likelihood(fwd_flag = .true.): logL, the traces, the integer shifts
    ll, rft = oracle.eval_batch(cfg, f["obs"], r_inv, nlay, layers, f["sig"], nsmp, want_rft=True, nthreads=oracle.max_threads())
    _, kap = oracle.eval_batch(cfg, f["obs"], r_inv, nlay, layers, f["sig"], nsmp, nthreads=oracle.max_threads(), want_kappa=True)
    check_path(f, workload, "oracle", ll, rft, kap)
    for i in range(0, n, 7):
        st = tuple(layers[i, r, :nlay[i]] for r in range(4))
        assert np.array_equal(oracle.calc_rf(cfg, *st, want_stages=True)[1], f["npre"][i]), (workload, i)


def format_cases(tmp_path):
    """(p, refm, mcfg, arrays) of every case of format_model.npz; the reference velocity table is the fixture's (non-uniform)."""
    from rf_inv_amd import get_params, read_ref_model

    f = load("format", "model")
    vel = tmp_path / "fixture.velmod"
    vel.write_text(str(f["velmod"]))
    refm = read_ref_model(str(vel))
    assert np.array_equal(refm.vp_ref, f["vp_ref"]) and np.array_equal(refm.vs_ref, f["vs_ref"])
    for ci in range(len(f["cases"])):
        pin = tmp_path / f"params_{ci}.in"
        pin.write_text(str(f[f"c{ci}_params_in"]))
        p = get_params(str(pin))
        assert (p.sdep, p.vp_mode, p.k_max) == (f["cases"][ci][0], int(f["cases"][ci][1]), int(f["cases"][ci][2]))
        mcfg = dict(k_max=p.k_max, vp_mode=p.vp_mode, sdep=p.sdep, z_max=p.z_max, h_min=p.h_min, z_ref_min=refm.z_ref_min,
                    dz_ref=refm.dz_ref, vp_min=p.vp_min, vp_max=p.vp_max, vs_min=p.vs_min, vs_max=p.vs_max,
                    vpvs_min=p.vpvs_min, vpvs_max=p.vpvs_max, vp_ref=refm.vp_ref, vs_ref=refm.vs_ref)
        yield ci, p, refm, mcfg, {k: f[f"c{ci}_{k}"] for k in ("k", "z", "dvp", "dvs", "nlay", "valid", "layers")}


def test_oracle_format_model_against_the_reference_fixture(oracle, tmp_path):
    """src/model.f90:175-290 through the reference itself on 3 x 700 proposals (valid and invalid, equal interface depths,
    ocean, vp_mode 0 / 1, a non-uniform reference table): layer count, verdict and every layer value bit for bit."""
    n_valid = 0
    for ci, p, refm, mcfg, a in format_cases(tmp_path):
        for i in range(len(a["k"])):
            nl, al, be, rh, h, ok = oracle.format_model(mcfg, int(a["k"][i]), a["z"][i], a["dvp"][i], a["dvs"][i])
            assert ok == bool(a["valid"][i]), (ci, i)
            if ok:                                                   # (an invalid model's partial stack is never used)
                assert nl == a["nlay"][i], (ci, i)
                for row, arr in enumerate((al, be, rh, h)):
                    assert np.array_equal(arr, a["layers"][i, row, :nl]), (ci, i, row)
                n_valid += 1
    assert n_valid > 900


def check_path(f, workload, who, ll, rft, kap):
    """logL within the north star's tolerance of the reference's calc_likelihood, traces within 1e-12 of their scale;
    an item whose normalising maximum is ill-conditioned (kappa >= 1000, tests/helpers.py) gets kappa / 1000 of slack."""
    nsmp = int(f["nsmp"])
    d, tol = np.abs(ll - f["logl"]), logl_tol(f["logl"])
    slack = np.where(kap >= 1000.0, kap / 1000.0, 1.0)
    bad = np.nonzero(~(d <= tol * slack))[0]
    assert bad.size == 0, (workload, who, [(int(i), float(ll[i]), float(f["logl"][i]), float(kap[i])) for i in bad[:4]])
    worst_plain = float((d / tol)[kap < 1000.0].max()) if (kap < 1000.0).any() else 0.0
    nfull = f["rft_full"].shape[0]
    for i in range(len(ll)):
        trace_close(rft[i][:, :nsmp], f["rft_window"][i], (workload, who, i, "window"), tol=TRACE_TOL * slack[i])
        if i < nfull:
            trace_close(rft[i], f["rft_full"][i], (workload, who, i, "full trace"), tol=TRACE_TOL * slack[i])
    ref_spread = float((np.abs(f["logl_o2"] - f["logl"]) / tol).max())
    print(f"path_{workload}: {who} against the reference's calc_likelihood on {len(ll)} models: worst |dlogL| / tolerance "
          f"{worst_plain:.3f} (items with kappa < 1000), {int((d > tol).sum())} item(s) used the kappa allowance; the reference's own "
          f"-O2 build against its -O0 build: {ref_spread:.3f} of the tolerance")
    return worst_plain


# ---------------------------------------------------------------------------------------------------------------
# GPU: the HIP path through the C ABI against the same fixtures
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", FORWARD)
def test_hip_calc_rf_against_the_reference_fixture(name):
    from rf_inv_amd import RFEngine

    f = load("forward", name)
    n, ntrc, nsmp = len(f["nlay"]), int(f["ntrc"]), int(f["nsmp"])
    with RFEngine(obs=np.zeros((ntrc, nsmp)), max_walkers=n, nlay_max=f["layers"].shape[2], **engine_kw(f)) as eng:
        assert np.array_equal(eng.flt.T, f["flt"]) and eng.is_ray_common == bool(f["is_ray_common"])
        eng.eval_batch(np.arange(n), f["nlay"], f["layers"], np.full((n, ntrc), 0.02))
        got = eng.get_rft_batch(np.arange(n), which=1)
        worst = 0.0
        for i in range(n):
            worst = max(worst, trace_close(got[i], f["rft"][i], (name, "batch", i)))
            nl = int(f["nlay"][i])
            one = eng.calc_rf(nl, *(f["layers"][i, r, :nl] for r in range(4))).T        # the single-call drop-in entry
            trace_close(one, f["rft"][i], (name, "rf_calc_rf", i))
    print(f"forward_{name}: HIP against the reference's calc_rf, max |d trace| / max|trace| = {worst:.2e}")


@pytest.mark.gpu
@pytest.mark.parametrize("workload", PATH)
def test_hip_calc_likelihood_against_the_reference_fixture(oracle, tmp_path, workload):
    from rf_inv_amd import RFEngine

    f = load("path", workload)
    cfg, nsmp, ntrc = cfg_of(f), int(f["nsmp"]), int(f["ntrc"])
    p, refm, _ = model_setup(f, tmp_path)
    n = len(f["k"])
    r_inv = f["r_inv"][f["r_index"]]
    ids = np.arange(n, dtype=np.int32)
    kz = max(p.k_max - 1, 1)
    with RFEngine(obs=f["obs"], r_inv=r_inv, max_walkers=n, nlay_max=p.k_max + 2, **engine_kw(f)) as eng:
        eng.set_model(p, refm)
        ll = eng.eval_models(ids, f["k"], f["z"][:, :kz], f["dvp"], f["dvs"], f["sig"])         # format_model on the device too
        got = eng.get_rft_batch(ids, which=1)
        ll2 = eng.eval_batch(ids, f["nlay"], f["layers"], f["sig"])                              # the reference's own layer stacks
        assert np.array_equal(ll, ll2)
        # the sigma-only branch on traces the host stored (src/likelihood.f90:81,87-98)
        for j in range(len(f["probe_logl"])):
            tr = np.zeros((cfg["nfft"], ntrc))
            tr[:nsmp] = f["probe_window"][j].T
            lp = eng.calc_likelihood_of_trace(tr, f["probe_sig"][j])
            assert abs(lp - f["probe_logl"][j]) <= logl_tol(f["probe_logl"][j]), (workload, j, lp, f["probe_logl"][j])
    _, kap = oracle.eval_batch(cfg, f["obs"], r_inv, f["nlay"], f["layers"], f["sig"], nsmp, nthreads=oracle.max_threads(),
                               want_kappa=True)
    worst = check_path(f, workload, "HIP", ll, got, kap)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"reference_fixture_{workload}.json"), "w") as fh:
            json.dump({"workload": workload, "models": int(n), "worst_fraction_of_tolerance_hip": worst,
                       "max_abs_dlogl_hip": float(np.abs(ll - f["logl"]).max()),
                       "max_rel_dlogl_hip": float((np.abs(ll - f["logl"]) / np.abs(f["logl"])).max())}, fh)


@pytest.mark.gpu
@pytest.mark.parametrize("nranks,nburn,niter", [(1, 60, 240), (2, 60, 240), (1, 3000, 8000)])
def test_dropin_main_program_reproduces_the_cpu_references_result_files(tmp_path, nranks, nburn, niter):
    """The whole program against the whole reference: run_sample_syn.npz holds what the reference's rf_inv -- every source
    its own, on the CPU -- wrote for the shipped sample_syn directory (sha256 of the eleven model / histogram / mean files,
    the rslt/likelihood table).  The same main program linked against the drop-in modules (oracle/_ref/rf_inv: fftw /
    forward / likelihood of rf_inv_amd/fortran, the HIP kernels) must take the same trajectory: the eleven files byte for
    byte, rslt/likelihood (17 digits) to 1e-11 relative; 1 and 2 MPI ranks, and the shipped length (11 000 iterations)."""
    import shutil
    import subprocess

    exe = os.path.join(ROOT, "oracle", "_ref", "rf_inv")
    mpiexec = "/opt/conda/bin/mpiexec"
    assert os.path.exists(exe) and os.path.exists(mpiexec), "oracle/_ref/rf_inv (build()) or mpiexec is missing on a GPU box"
    f = load("run", "sample_syn")
    tag = f"np{nranks}_{nburn}_{niter}"
    work = tmp_path / "run"
    shutil.copytree(os.path.join(ROOT, "tests", "golden", "sample_syn"), work)
    os.makedirs(work / "rslt")
    lines = open(work / "params.in").read().split("\n")
    i = next(j for j, l in enumerate(lines) if l.startswith("# N_BURN"))
    lines[i + 1], lines[i + 3] = str(nburn), str(niter)
    open(work / "params.in", "w").write("\n".join(lines))
    r = subprocess.run([mpiexec, "-np", str(nranks), exe, "params.in"], cwd=work, env=dict(os.environ), capture_output=True,
                       text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    for name, want in zip(f["files"], f[f"{tag}_sha256"]):
        got = hashlib.sha256(open(work / "rslt" / str(name), "rb").read()).hexdigest()
        assert got == str(want), (tag, str(name))
    lk, ref = np.loadtxt(work / "rslt" / "likelihood"), f[f"{tag}_likelihood"]
    assert lk.shape == ref.shape == (nburn + niter, 2) and np.array_equal(lk[:, 0], ref[:, 0])
    rel = np.abs(lk[:, 1] - ref[:, 1]) / np.abs(ref[:, 1])
    assert rel.max() <= 1e-11, (tag, rel.max())
    print(f"{tag}: eleven result files byte-identical to the CPU reference's; rslt/likelihood within {rel.max():.1e}")


@pytest.mark.gpu
def test_hip_format_model_against_the_reference_fixture(tmp_path):
    """format_model_kernel (rf_format_models_device) against the reference's own format_model: bit for bit."""
    import torch

    from rf_inv_amd import RFEngine

    dev = torch.device("cuda", 0)
    n_valid = 0
    for ci, p, refm, mcfg, a in format_cases(tmp_path):
        nb, pad = len(a["k"]), p.k_max + 2
        with RFEngine(nfft=256, delta=float(np.float32(0.05)), t_start=0.0, deconv_mode=0, sdep=p.sdep, rayps=[0.06], a_gus=[4.0],
                      ipha=[1], obs=np.zeros((1, 101)), nsmp=101, max_walkers=nb, nlay_max=pad) as eng:
            eng.set_model(p, refm)
            t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
            nlay = torch.zeros(nb, dtype=torch.int32, device=dev)
            layers = torch.zeros((nb, 4, pad), dtype=torch.float64, device=dev)
            valid = torch.zeros(nb, dtype=torch.int32, device=dev)
            eng.format_models_device(t(a["k"].astype(np.int32)), t(a["z"]), t(a["dvp"]), t(a["dvs"]), nlay, layers, valid)
            torch.cuda.synchronize()
            nlay, layers, valid = nlay.cpu().numpy(), layers.cpu().numpy(), valid.cpu().numpy()
        assert np.array_equal(valid != 0, a["valid"] != 0), ci
        for i in np.nonzero(a["valid"])[0]:
            nl = int(a["nlay"][i])
            assert nlay[i] == nl and np.array_equal(layers[i, :, :nl], a["layers"][i, :, :nl]), (ci, int(i))
            n_valid += 1
    assert n_valid > 900


@pytest.mark.gpu
def test_baseline_configs0_on_the_dropin_modules(tmp_path):
    """BASELINE.json configs[0] -- the shipped params.in with one P trace, 8 MPI ranks x 4 chains (one of them at T = 1) --
    through the reference's main program on the drop-in modules (eight ranks share the one GPU, each with its own context;
    the swap is the reference's own MPI exchange): the eleven result files of the reference itself (CPU build, fixture)
    byte for byte, rslt/likelihood to 1e-11."""
    import shutil
    import subprocess

    exe, mpiexec = os.path.join(ROOT, "oracle", "_ref", "rf_inv"), "/opt/conda/bin/mpiexec"
    assert os.path.exists(exe) and os.path.exists(mpiexec), "oracle/_ref/rf_inv (build()) or mpiexec is missing on a GPU box"
    f = load("run", "sample_syn")
    work = tmp_path / "run"
    shutil.copytree(os.path.join(ROOT, "tests", "golden", "sample_syn"), work)
    os.makedirs(work / "rslt")
    (work / "params.in").write_text(str(f["configs0_np8_params_in"]))
    r = subprocess.run([mpiexec, "-np", "8", exe, "params.in"], cwd=work, env=dict(os.environ), capture_output=True, text=True,
                       timeout=1800)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    for name, want in zip(f["files"], f["configs0_np8_sha256"]):
        assert hashlib.sha256(open(work / "rslt" / str(name), "rb").read()).hexdigest() == str(want), str(name)
    lk, ref = np.loadtxt(work / "rslt" / "likelihood"), f["configs0_np8_likelihood"]
    assert lk.shape == ref.shape == (300, 2)
    assert (np.abs(lk[:, 1] - ref[:, 1]) / np.abs(ref[:, 1])).max() <= 1e-11
