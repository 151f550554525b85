"""Run the reference's OWN code on the CPU (test infrastructure; never imported by rf_inv_amd/).

The builds of oracle/Makefile.ref -- all twelve /root/reference/src/*.f90 compiled unmodified, FFTW3's Fortran
interface and LAPACK from the image's Intel MKL, no product object on any link line, no GPU -- driven through the two
dumpers tests/fortran/ref_forward_dump.f90 (calc_rf, src/forward.f90:123-208) and ref_path_dump.f90 (calc_likelihood,
src/likelihood.f90:56-101).  This module writes their run directories (params.in in the reference's positional format,
SAC traces, the velocity model file), starts them and parses what they write.  Callers: oracle/gen_golden.py (the
committed fixtures under tests/golden/ref/), bench.py's cpu_baseline leg, tests/tools/kappa_reference_spread.py.

The binaries are prebuilt where /root/reference exists and travel with the snapshot (oracle/_ref/ is git-ignored);
nothing here reads /root/reference at run time."""
import copy
import os
import shutil
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "sample_syn")
BUILDS = ("cpu_o0", "cpu_o2")
REF_NLAY_MAX = 200          # src/params.f90:44


def exe(build, name):
    """Path of a program of oracle/Makefile.ref's build `build` ("cpu_o0" = the reference Makefile's default -O0
    class, "cpu_o2" = optimised, no value-changing flags)."""
    assert build in BUILDS
    return os.path.join(ROOT, "oracle", "_ref", build, name)


def available(build="cpu_o0"):
    return all(os.path.exists(exe(build, n)) for n in ("ref_forward_dump", "ref_path_dump", "rf_inv"))


def clean_env():
    """Children run without this process's OpenMP binding (inherited OMP_PLACES / OMP_PROC_BIND would pin every child's
    main thread to the first core) and with MKL on one thread (the reference is one thread per MPI rank)."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("OMP_")}
    env.update(MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
    return env


def write_run_dir(work, p, obs=None, header="written by oracle/refrun.py", dvs_prior=0.3, velmod=None):
    """A run directory for the reference's programs: params.in for the geometry of `p` (an rf_inv_amd.params.Params),
    SAC files data/t<i>.trc holding obs[t, :nsmp] (zeros when obs is None), model/sample.velmod (the shipped reference
    model).  dvs_prior: init_model (src/model.f90:66-95, run by the dumper's init sequence) redraws whole models until one is
    valid; the narrow prior keeps that short and touches nothing calc_rf / calc_likelihood compute for GIVEN models.
    velmod: text of another reference velocity table (rows "z vp vs").  Returns the text of params.in."""
    from rf_inv_amd import write_params
    from rf_inv_amd.make_syn import write_sac

    for d in ("data", "rslt", "model"):
        os.makedirs(os.path.join(work, d), exist_ok=True)
    if velmod is None:
        shutil.copy(os.path.join(GOLDEN, "model", "sample.velmod"), os.path.join(work, "model", "sample.velmod"))
    else:
        open(os.path.join(work, "model", "sample.velmod"), "w").write(velmod)
    q = copy.copy(p)
    q.out_dir, q.nchains, q.ncool, q.nburn, q.niter, q.dvs_prior = "./rslt", 1, 1, 0, 10, dvs_prior
    q.vel_file, q.obs_files = "model/sample.velmod", [f"data/t{t + 1}.trc" for t in range(p.ntrc)]
    for t, f in enumerate(q.obs_files):
        tr = np.zeros(p.nsmp) if obs is None else np.asarray(obs)[t, :p.nsmp]
        write_sac(os.path.join(work, f), tr, p.delta, p.t_start, p.t_end)
    write_params(os.path.join(work, "params.in"), q, header=header)
    return open(os.path.join(work, "params.in")).read()


def write_stacks(path, stacks):
    """stacks: list of (alpha, beta, rho, h).  repr() of a float64 round-trips: the reference reads the same doubles."""
    with open(path, "w") as fh:
        fh.write(f"{len(stacks)}\n")
        for st in stacks:
            fh.write(f"{len(st[0])}\n")
            for j in range(len(st[0])):
                fh.write(" ".join(repr(float(st[r][j])) for r in range(4)) + "\n")


def write_models(path, k_max, m_k, m_z, m_dvp, m_dvs, sig):
    n = len(m_k)
    with open(path, "w") as fh:
        fh.write(f"{n}\n")
        for i in range(n):
            fh.write(f"{int(m_k[i])}\n")
            for arr in (m_z[i, :max(k_max - 1, 1)], m_dvp[i, :k_max], m_dvs[i, :k_max], sig[i]):
                fh.write(" ".join(repr(float(x)) for x in arr) + "\n")


def _run(cmd, work, timeout):
    r = subprocess.run(cmd, cwd=work, capture_output=True, text=True, timeout=timeout, env=clean_env())
    if r.returncode != 0 or ": ok" not in r.stdout:
        raise RuntimeError(f"{cmd[0]} failed ({r.returncode}): {r.stdout[-800:]} {r.stderr[-800:]}")
    return r.stdout


def run_forward(build, work, n, nfft, ntrc, timeout=900):
    """ref_forward_dump in `work` (params.in + stacks.txt there).  Returns dict(flt[ntrc, nh], rft[n, ntrc, nfft],
    tp[n, ntrc], npre[n, ntrc], common)."""
    out = _run([exe(build, "ref_forward_dump"), "params.in", "stacks.txt", "ref.bin", "0", "extras.bin"], work, timeout)
    raw = open(os.path.join(work, "ref.bin"), "rb").read()
    nh = nfft // 2 + 1
    assert tuple(np.frombuffer(raw[:16], dtype="<i4")) == (nfft, ntrc, nh, n)
    body = np.frombuffer(raw[16:], dtype="<f8")
    ex = open(os.path.join(work, "extras.bin"), "rb").read()
    rec = 8 * ntrc + 4 * ntrc
    assert len(ex) == n * rec
    tp = np.stack([np.frombuffer(ex[i * rec:i * rec + 8 * ntrc], dtype="<f8") for i in range(n)])
    npre = np.stack([np.frombuffer(ex[i * rec + 8 * ntrc:(i + 1) * rec], dtype="<i4") for i in range(n)])
    return dict(flt=body[:nh * ntrc].reshape(ntrc, nh).copy(), rft=body[nh * ntrc:].reshape(n, ntrc, nfft).copy(),
                tp=tp, npre=npre, common="T" in out.split("ok")[-1])


def run_path(build, work, n, p, nlay_max=REF_NLAY_MAX, reps=0, timeout=1800, out_name="ref.bin"):
    """ref_path_dump in `work` (params.in + models.txt there).  Returns dict(r_inv[ntrc, nsmp, nsmp] (r_inv[t].ravel()
    == Fortran r_inv(:, :, t)), logl[n], rft[n, ntrc, nfft], probe_logl[m], probe_trace[m, ntrc, nfft], probe_sig[m,
    ntrc], nlay[n], valid[n], layers[n, 4, nlay_max], tp[n, ntrc], npre[n, ntrc], seconds, evaluations)."""
    out = _run([exe(build, "ref_path_dump"), "params.in", "models.txt", out_name, str(int(reps)), "extras_" + out_name],
               work, timeout)
    raw = open(os.path.join(work, out_name), "rb").read()
    nfft, ntrc, nsmp, n_out, m = (int(x) for x in np.frombuffer(raw[:20], dtype="<i4"))
    assert (nfft, ntrc, nsmp, n_out) == (p.nfft, p.ntrc, p.nsmp, n), (nfft, ntrc, nsmp, n_out)
    body = np.frombuffer(raw[20:], dtype="<f8")
    o = nsmp * nsmp * ntrc
    rec = body[o:o + n * (1 + nfft * ntrc)].reshape(n, 1 + nfft * ntrc)
    prob = body[o + n * (1 + nfft * ntrc):].reshape(m, 1 + nfft * ntrc + ntrc)
    ex = open(os.path.join(work, "extras_" + out_name), "rb").read()
    sz = 8 + 8 * 4 * nlay_max + 8 * ntrc + 4 * ntrc
    assert len(ex) == n * sz, (len(ex), n, sz)
    nlay = np.zeros(n, dtype=np.int32); valid = np.zeros(n, dtype=np.int32)
    layers = np.zeros((n, 4, nlay_max)); tp = np.zeros((n, ntrc)); npre = np.zeros((n, ntrc), dtype=np.int32)
    for i in range(n):
        b = ex[i * sz:(i + 1) * sz]
        nlay[i], valid[i] = np.frombuffer(b[:8], dtype="<i4")
        layers[i] = np.frombuffer(b[8:8 + 32 * nlay_max], dtype="<f8").reshape(4, nlay_max)
        tp[i] = np.frombuffer(b[8 + 32 * nlay_max:8 + 32 * nlay_max + 8 * ntrc], dtype="<f8")
        npre[i] = np.frombuffer(b[8 + 32 * nlay_max + 8 * ntrc:], dtype="<i4")
    res = dict(r_inv=body[:o].reshape(ntrc, nsmp, nsmp).copy(), logl=rec[:, 0].copy(),
               rft=rec[:, 1:].reshape(n, ntrc, nfft).copy(), probe_logl=prob[:, 0].copy(),
               probe_trace=prob[:, 1:1 + nfft * ntrc].reshape(m, ntrc, nfft).copy(), probe_sig=prob[:, 1 + nfft * ntrc:].copy(),
               nlay=nlay, valid=valid, layers=layers, tp=tp, npre=npre, seconds=None, evaluations=None)
    line = [l for l in out.splitlines() if "ref_path_dump: seconds" in l]
    if line:
        tok = line[0].split()
        res["seconds"], res["evaluations"] = float(tok[2]), int(tok[4])
    return res
