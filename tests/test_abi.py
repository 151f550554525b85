"""The C-ABI library loads and exports every symbol include/rfgpu.h (the section-8b contract) and include/rfgpu_ext.h
(extensions) declare.  CPU only:
no compute entry is called (rf_compute_r_inv is a host-only helper)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(names=("rfgpu.h", "rfgpu_ext.h")):
    out = set()
    for name in names:
        txt = open(os.path.join(ROOT, "include", name)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        out |= set(re.findall(r"\b(rf_[a-z_0-9]+)\s*\(", txt))
    return sorted(out)


# SURVEY.md section 8b, last row: what a C-ABI replacement must export at least
SURVEY_8B_MINIMUM = {"rf_ctx_create", "rf_calc_rf", "rf_calc_likelihood", "rf_eval_batch", "rf_commit", "rf_get_rft",
                     "rf_pt_swap_device", "rf_ctx_destroy"}
# measured slower than the default and removed in round 5 (What's weak #7): must not come back unnoticed
REMOVED = {"rf_host_alloc_shared", "rf_host_unlink_shared", "rf_host_free_shared", "rf_release_gpu", "rf_post_sets",
           "rf_post_select"}


def test_header_symbols_exported_and_bound():
    from rf_inv_amd import _lib

    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in rfgpu.h but not exported by librfgpu.so"
        assert n in _lib.SYMBOLS, f"{n} not bound in rf_inv_amd/_lib.py"
    assert sorted(_lib.SYMBOLS) == names
    assert lib.rf_abi_version() == 6
    contract = set(_header_functions(("rfgpu.h",)))
    assert SURVEY_8B_MINIMUM <= contract and len(contract) <= 32
    # the contract header stands alone: nothing of the extensions leaks into it
    assert not any(n.startswith(("rf_post_", "rf_eval_models", "rf_host_", "rf_set_option", "rf_profile_")) for n in contract)
    assert not (REMOVED & set(names)) and not any(hasattr(lib, n) for n in REMOVED)


def test_config_struct_layout_matches_header():
    from rf_inv_amd import _lib

    # struct rf_config: 4 int32, 3 double, 3 ptr, ptr, int32(+pad), ptr, 3 int32(+pad)
    assert C.sizeof(_lib.RFConfig) == 16 + 24 + 24 + 8 + 8 + 8 + 16
    assert _lib.RFConfig.delta.offset == 16 and _lib.RFConfig.rayps.offset == 40
    assert _lib.RFConfig.ldobs.offset == 72 and _lib.RFConfig.r_inv.offset == 80
    assert _lib.RFConfig.max_walkers.offset == 88


def test_posterior_struct_layouts_match_header():
    from rf_inv_amd import _lib

    # struct rf_post_config: 6 int32, 3 double, 3 ptr, int64
    assert C.sizeof(_lib.RFPostConfig) == 24 + 24 + 24 + 8
    assert _lib.RFPostConfig.amp_min.offset == 24 and _lib.RFPostConfig.sig_min.offset == 48
    assert _lib.RFPostConfig.max_models.offset == 72
    # struct rf_post_result: 15 pointers
    assert C.sizeof(_lib.RFPostResult) == 15 * 8 and _lib.RFPostResult.amp_out_of_range.offset == 14 * 8


def test_no_cpu_fallback_ctx_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the loud-failure path is for GPU-less hosts")
    from rf_inv_amd import RFEngine, RFGPUError

    with pytest.raises(RFGPUError, match="no HIP device|CPU fallback|hip"):
        RFEngine(nfft=256, delta=0.05, t_start=0.0, deconv_mode=0, sdep=0.0, rayps=[0.06], a_gus=[4.0],
                 ipha=[1], obs=np.zeros((1, 101)), nsmp=101)


def test_ctx_create_argument_validation():
    from rf_inv_amd import RFEngine, RFGPUError

    with pytest.raises(RFGPUError, match="nfft must be >= 8"):
        RFEngine(nfft=6, delta=0.05, t_start=0.0, deconv_mode=0, sdep=0.0, rayps=[0.06], a_gus=[4.0],
                 ipha=[1], obs=np.zeros((1, 3)), nsmp=3)
    with pytest.raises(RFGPUError, match="ipha"):
        RFEngine(nfft=256, delta=0.05, t_start=0.0, deconv_mode=0, sdep=0.0, rayps=[0.06], a_gus=[4.0],
                 ipha=[0], obs=np.zeros((1, 101)), nsmp=101)


def test_compute_r_inv_host_helper(oracle):
    from rf_inv_amd.engine import compute_r_inv

    d = float(np.float32(0.05))
    r, rank = compute_r_inv(101, 4.0, d)
    ref, ranks = oracle.build_r_inv(101, [4.0], d, return_rank=True)
    assert rank == ranks[0] == 40
    assert np.abs(r - ref[0]).max() <= 1e-11 * np.abs(ref[0]).max()
    # rank cut sits in a safe gap for the benchmark geometry
    idx = np.arange(101)
    s = np.linalg.svd(np.exp(-16.0 * d * d) ** ((idx[:, None] - idx[None, :]) ** 2.0), compute_uv=False)
    assert s[39] > 1.2e-3 and s[40] < 0.99e-3


def test_product_never_imports_oracle():
    """The product package must not reach into oracle/ (test infrastructure only)."""
    pkg = os.path.join(ROOT, "rf_inv_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".f90", ".F90")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "rf_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_header_is_plain_c_and_a_c_host_links(tmp_path):
    """include/rfgpu.h compiles as C99 and a C program links against librfgpu.so (tests/c/abi_smoke.c)."""
    import subprocess

    from rf_inv_amd import _lib

    _lib.load()
    exe = tmp_path / "abi_smoke"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", str(exe), "-L", libdir, "-lrfgpu",
                           "-lm", f"-Wl,-rpath,{libdir}"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi 6 rank" in r.stdout
    assert "no context: rf_ctx_create: no HIP device" in r.stdout or "calc_rf rc 0" in r.stdout


def test_product_library_reads_no_environment_variables():
    """Every launch-plan knob is an explicit rf_set_option call echoed by rf_get_launch_plan; a stray
    variable in the environment cannot change what the library computes (or skips)."""
    import glob

    for f in glob.glob(os.path.join(ROOT, "rf_inv_amd", "csrc", "*")):
        assert "getenv" not in open(f).read(), f
    src = open(os.path.join(ROOT, "rf_inv_amd", "_lib.py")).read()
    assert "os.environ" not in src and "getenv" not in src
    # the diagnostics exits exist only under the RFGPU_DIAGNOSTICS macro
    k = open(os.path.join(ROOT, "rf_inv_amd", "csrc", "rfgpu_kernels.hip")).read()
    assert not re.search(r"if \(P\.ablate", k.split("#endif", 1)[1])


@pytest.mark.parametrize("a_gus", [2.5, 4.0, 8.0])
@pytest.mark.parametrize("nsmp", [61, 101, 401])
def test_builtin_r_inv_agrees_with_lapack_in_rank_and_values(oracle, a_gus, nsmp):
    """librfgpu's one-sided Jacobi SVD against LAPACK dgesvd (the reference's route, src/likelihood.f90:196-222)
    across filter widths and window lengths: same rank at the hard 1e-3 cut, values to 1e-11, and a reported gap
    at the cut that is wide compared with SVD rounding (so the rank cannot depend on the SVD algorithm)."""
    from rf_inv_amd.engine import compute_r_inv

    delta = float(np.float32(0.05))
    r, rank, gap = compute_r_inv(nsmp, a_gus, delta, with_gap=True)
    ref, ranks = oracle.build_r_inv(nsmp, [a_gus], delta, return_rank=True)
    assert rank == ranks[0]
    assert np.abs(r - ref[0]).max() <= 1e-11 * np.abs(ref[0]).max()
    assert gap > 1e-3


@pytest.mark.parametrize("a_gus,nsmp", [(4.0, 101), (2.5, 161), (8.0, 61)])
def test_r_inv_against_a_symmetric_eigendecomposition(oracle, a_gus, nsmp):
    """An independent route to the reference's truncated pseudo-inverse (src/likelihood.f90:180-222: dgesvd of the
    noise correlation matrix, 1/s for s > 1e-3, V diag U^T): the matrix is symmetric, so its eigen-decomposition
    gives R+ = Q diag(1/lambda | lambda > 1e-3) Q^T and the quadratic form as sum (q_k . m)^2 / lambda_k -- no SVD,
    no matrix product of factors.  Both builders (the oracle's LAPACK route and librfgpu's Jacobi SVD) and the
    quadratic form of the log-likelihood agree with it."""
    from rf_inv_amd.engine import compute_r_inv

    delta = float(np.float32(0.05))
    idx = np.arange(nsmp)
    r = np.exp(-a_gus ** 2 * delta ** 2)
    R = r ** ((idx[:, None] - idx[None, :]) ** 2.0)
    lam, Q = np.linalg.eigh(R)
    keep = lam > 1.0e-3
    want = (Q[:, keep] / lam[keep]) @ Q[:, keep].T
    ref, ranks = oracle.build_r_inv(nsmp, [a_gus], delta, return_rank=True)
    own, rank = compute_r_inv(nsmp, a_gus, delta)
    assert ranks[0] == rank == int(keep.sum())
    scale = np.abs(want).max()
    assert np.abs(ref[0].T - want).max() <= 1e-9 * scale          # conditioning 1e3: ~1e-13 * 1e3 of slack
    assert np.abs(own.T - want).max() <= 1e-9 * scale
    rng = np.random.default_rng(nsmp)
    m = rng.standard_normal(nsmp)
    phi = float(np.sum((Q[:, keep].T @ m) ** 2 / lam[keep]))
    sig = np.array([0.37])
    ll = oracle.log_likelihood(np.concatenate([m, np.zeros(7)])[None, :], np.zeros((1, nsmp)), ref, sig, nsmp)
    assert abs(ll - (-0.5 * phi / sig[0] ** 2 - nsmp * np.log(sig[0]))) <= 1e-9 * abs(ll)


def _a_gus_with_singular_value_on_the_cut(nsmp, delta):
    """Bisect the filter width until a singular value of the noise matrix sits on the 1e-3 cut-off."""
    from rf_inv_amd.engine import compute_r_inv

    lo, hi = 4.0, 4.2
    rank_lo = compute_r_inv(nsmp, lo, delta)[1]
    assert compute_r_inv(nsmp, hi, delta)[1] > rank_lo
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        _, rank, gap = compute_r_inv(nsmp, mid, delta, with_gap=True)
        if gap < 1e-9:
            return mid, gap
        if rank > rank_lo:
            hi = mid
        else:
            lo = mid
    return mid, gap


def test_r_inv_gap_reports_a_singular_value_on_the_cut():
    """The rank of the pseudo-inverse flips where a singular value crosses 1e-3; rf_compute_r_inv reports how
    close the nearest one is, so that a caller (and rf_ctx_create) can refuse an ill-defined R^-1."""
    delta = float(np.float32(0.05))
    a, gap = _a_gus_with_singular_value_on_the_cut(61, delta)
    assert gap < 1e-7


@pytest.mark.gpu
def test_ctx_create_refuses_to_build_an_ill_defined_r_inv():
    from rf_inv_amd import RFEngine, RFGPUError

    delta = float(np.float32(0.05))
    a, gap = _a_gus_with_singular_value_on_the_cut(61, delta)
    kw = dict(nfft=256, delta=delta, t_start=0.0, deconv_mode=0, sdep=0.0, rayps=np.array([0.06]),
              ipha=np.array([1], dtype=np.int32), obs=np.zeros((1, 61)), nsmp=61, max_walkers=1)
    with pytest.raises(RFGPUError, match="rank cut-off"):
        RFEngine(a_gus=np.array([a]), **kw)
    with RFEngine(a_gus=np.array([4.0]), **kw) as eng:          # the ordinary case reports rank and gap
        rank, g = eng.r_inv_info
        assert rank[0] == 25 and g[0] > 0.1
    with RFEngine(a_gus=np.array([a]), r_inv=np.zeros((1, 61, 61)), **kw) as eng:   # the host's own r_inv is taken as is
        rank, g = eng.r_inv_info
        assert rank[0] == -1 and np.isnan(g[0])


def test_bench_refuses_a_rank_count_it_cannot_honour():
    """`bench.py --gpus N` must never report fewer GPUs than it was asked for (the flag was parsed and
    ignored).  Without a launcher it starts the ranks itself -- and refuses, before anything touches a GPU, when the
    node shows fewer than N devices (this container shows none); under a launcher WORLD_SIZE must equal --gpus."""
    import subprocess
    import sys

    import torch

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RFGPU_BENCH_BACKEND")}
    bench = os.path.join(ROOT, "bench.py")
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300,
                           env=env, cwd=ROOT)
        assert r.returncode != 0 and "needs 2 visible GPUs" in r.stderr
        # (round 6: a refused run says so in ONE JSON line -- value null, never a number)
        import json

        d = json.loads(r.stdout.strip())
        assert d["value"] is None and d["n_gpus"] == 2 and "visible GPUs" in d["error"]
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="3", RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "must agree" in r.stderr and "{" not in r.stdout
    r = subprocess.run([sys.executable, bench, "--gpus", "0"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
