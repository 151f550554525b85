"""Row f-2: format_model on the device (rf_format_models_device / rf_eval_models_device) against
the oracle's restatement of src/model.f90:175-290 -- bit-exact bookkeeping."""
import copy
import os

import numpy as np
import pytest

from helpers import DELTA, logl_tol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(golden_dir, sdep, vp_mode, k_max):
    from rf_inv_amd import get_params, read_obs, read_ref_model

    p = get_params(os.path.join(golden_dir, "sample_syn", "params.in"))
    read_obs(p)
    p.sdep, p.vp_mode, p.k_max = sdep, vp_mode, k_max
    ref = read_ref_model(os.path.join(p.base_dir, p.vel_file))
    ref = copy.copy(ref)
    ref.vp_ref = 5.0 + 0.03 * np.arange(ref.vp_ref.size)     # non-uniform: the iz look-ups matter
    ref.vs_ref = 2.8 + 0.02 * np.arange(ref.vs_ref.size)
    mcfg = dict(k_max=p.k_max, vp_mode=vp_mode, sdep=sdep, z_max=p.z_max, h_min=p.h_min, z_ref_min=ref.z_ref_min,
                dz_ref=ref.dz_ref, vp_min=p.vp_min, vp_max=p.vp_max, vs_min=p.vs_min, vs_max=p.vs_max,
                vpvs_min=p.vpvs_min, vpvs_max=p.vpvs_max, vp_ref=ref.vp_ref, vs_ref=ref.vs_ref)
    return p, ref, mcfg


def _proposals(rng, p, nb, ties=False):
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


@pytest.mark.parametrize("sdep,vp_mode,k_max,ties", [(2.0, 0, 10, False), (0.0, 1, 30, False), (0.0, 0, 12, True)])
def test_format_models_device_bit_exact(oracle, golden_dir, sdep, vp_mode, k_max, ties):
    import torch

    from rf_inv_amd import RFEngine

    p, ref, mcfg = _setup(golden_dir, sdep, vp_mode, k_max)
    rng = np.random.default_rng(k_max)
    nb = 2000
    k, z, dvp, dvs = _proposals(rng, p, nb, ties)
    dev = torch.device("cuda", 0)
    with RFEngine(nfft=256, delta=DELTA, t_start=0.0, deconv_mode=0, sdep=sdep, rayps=[0.06], a_gus=[4.0], ipha=[1],
                  obs=np.zeros((1, 101)), nsmp=101, max_walkers=nb, nlay_max=k_max + 2) as eng:
        eng.set_model(p, ref)
        pad = k_max + 2
        t = lambda a: torch.from_numpy(a).to(dev)
        nlay = torch.zeros(nb, dtype=torch.int32, device=dev)
        layers = torch.zeros((nb, 4, pad), dtype=torch.float64, device=dev)
        valid = torch.zeros(nb, dtype=torch.int32, device=dev)
        eng.format_models_device(t(k), t(z), t(dvp), t(dvs), nlay, layers, valid)
        torch.cuda.synchronize()
        nlay, layers, valid = nlay.cpu().numpy(), layers.cpu().numpy(), valid.cpu().numpy()
    n_valid = 0
    for i in range(nb):
        nl, a, b, r, h, ok = oracle.format_model(mcfg, int(k[i]), z[i], dvp[i], dvs[i])
        assert nlay[i] == nl and bool(valid[i]) == ok, i
        assert np.array_equal(layers[i, 0, :nl], a) and np.array_equal(layers[i, 1, :nl], b), i
        assert np.array_equal(layers[i, 2, :nl], r) and np.array_equal(layers[i, 3, :nl], h), i
        n_valid += ok
    assert 0 < n_valid < nb   # both outcomes exercised


def test_eval_models_device_matches_host_format_plus_eval(oracle, golden_dir):
    """(k, z, dVp, dVs) straight to logL on the device == host format_model + rf_eval_batch;
    invalid models are skipped (valid 0, logL NaN), sigma-only items keep working."""
    import torch

    from rf_inv_amd import RFEngine, format_model
    from rf_inv_amd.likelihood import init_r_inv

    p, ref, mcfg = _setup(golden_dir, 2.0, 0, 10)
    from rf_inv_amd import read_ref_model
    ref = read_ref_model(os.path.join(p.base_dir, p.vel_file))   # the shipped uniform table
    rng = np.random.default_rng(4)
    nb = 256
    k, z, dvp, dvs = _proposals(rng, p, nb)
    sig = np.full((nb, p.ntrc), 0.02)
    dev = torch.device("cuda", 0)
    with RFEngine.from_params(p, r_inv=init_r_inv(p.nsmp, p.a_gus, p.delta), max_walkers=nb) as eng:
        eng.set_model(p, ref)
        t = lambda a: torch.from_numpy(a).to(dev)
        ids = torch.arange(nb, dtype=torch.int32, device=dev)
        logl = torch.zeros(nb, dtype=torch.float64, device=dev)
        valid = torch.zeros(nb, dtype=torch.int32, device=dev)
        eng.eval_models_device(ids, t(k), t(z), t(dvp), t(dvs), t(sig), logl, valid)
        torch.cuda.synchronize()
        ll, ok = logl.cpu().numpy(), valid.cpu().numpy().astype(bool)
        # host path on the valid ones
        stacks, rows = [], []
        for i in range(nb):
            nl, a, b, r, h, v = format_model(p, ref, k[i], z[i], dvp[i], dvs[i])
            assert v == ok[i]
            if v:
                rows.append(i); stacks.append((a, b, r, h))
        assert 10 < len(rows) < nb
        assert np.all(np.isnan(ll[~ok]))
        from helpers import pack_layers
        nlay, layers = pack_layers(stacks, p.k_max + 2)
        eng.commit(np.arange(nb), np.ones(nb, dtype=np.int32))   # accept: proposals become current traces
        ref_ll = eng.eval_batch(np.array(rows), nlay, layers, sig[rows])
        assert np.array_equal(ll[rows], ref_ll)
        # sigma-only items through the model entry point re-use the committed traces
        ff = torch.zeros(nb, dtype=torch.int32, device=dev)
        eng.eval_models_device(ids, t(k), t(z), t(dvp), t(dvs), t(2 * sig), logl, valid, fwd_flag=ff)
        torch.cuda.synchronize()
        ll2 = logl.cpu().numpy()
        phi = -2 * (ll[rows] + 2 * p.nsmp * np.log(0.02)) * 0.02 ** 2     # sum over the 2 traces
        want = -0.5 * phi / 0.04 ** 2 - 2 * p.nsmp * np.log(0.04)
        assert np.allclose(ll2[rows], want, rtol=1e-11, atol=1e-8)


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("ldz_full", [False, True])
def test_eval_models_from_host_arrays(oracle, golden_dir, pinned, ldz_full):
    """rf_eval_models (host arrays in the batched sampler's column-per-chain layout; pageable, or pinned through
    rf_host_alloc; z with leading dimension k_max - 1 or k_max) == host format_model + rf_eval_batch, bit for bit;
    null proposals (fwd_flag < 0) and invalid models come back NaN, sigma-only items re-use the committed trace; and
    rf_commit, which no longer waits for the device, is ordered in front of everything that follows it."""
    from rf_inv_amd import RFEngine, format_model, read_ref_model
    from rf_inv_amd.engine import host_alloc
    from rf_inv_amd.likelihood import init_r_inv
    from helpers import pack_layers

    p, ref, mcfg = _setup(golden_dir, 0.0, 0, 12)
    ref = read_ref_model(os.path.join(p.base_dir, p.vel_file))
    rng = np.random.default_rng(7)
    nb = 700
    k, z, dvp, dvs = _proposals(rng, p, nb)
    if ldz_full:
        z = np.concatenate([z, rng.uniform(0, 20, (nb, 1))], axis=1)      # z(k_max, nb): the last row is never read
    sig = rng.uniform(0.01, 0.03, (nb, p.ntrc))
    ff = np.ones(nb, dtype=np.int32)
    ff[::7] = -1
    if pinned:
        def pin(a):
            b = host_alloc(a.shape, a.dtype)
            b[...] = a
            return b
        k, z, dvp, dvs, sig, ff = (pin(a) for a in (k, z, dvp, dvs, sig, ff))
    with RFEngine.from_params(p, r_inv=init_r_inv(p.nsmp, p.a_gus, p.delta), max_walkers=nb) as eng:
        eng.set_model(p, ref)
        ids_in = np.arange(nb, dtype=np.int32)
        if pinned:
            ids_in = pin(ids_in)
        ll, ok = eng.eval_models(ids_in, k, z, dvp, dvs, sig, fwd_flag=ff, want_valid=True)
        # pinned arrays go down by DMA as they are (also from an interior pointer: the second half of the batch);
        # pageable ones through the context's arena: ids, k, fwd_flag, z, dvs, sig (dvp is not read at vp_mode 0)
        assert eng.launch_plan["staged_host_arrays"] == (0 if pinned else 6)
        if pinned:
            h = nb // 2
            half = eng.eval_models(ids_in[h:], k[h:], z[h:], dvp[h:], dvs[h:], sig[h:], fwd_flag=ff[h:])
            assert eng.launch_plan["staged_host_arrays"] == 0
            assert np.array_equal(half, ll[h:], equal_nan=True)
        stacks, rows = [], []
        for i in range(nb):
            if ff[i] < 0:
                assert np.isnan(ll[i])
                continue
            nl, a, b, r, h, v = format_model(p, ref, k[i], z[i, :p.k_max - 1], dvp[i], dvs[i])
            assert v == bool(ok[i]), i
            if v:
                rows.append(i); stacks.append((a, b, r, h))
            else:
                assert np.isnan(ll[i])
        assert 30 < len(rows) < nb
        rows = np.array(rows)
        nlay, layers = pack_layers(stacks, p.k_max + 2)
        acc = np.zeros(nb, dtype=np.int32)
        acc[rows[::2]] = 1
        eng.commit(np.arange(nb), acc)                   # returns at once; what follows is ordered behind it
        ref_ll = eng.eval_batch(rows, nlay, layers, sig[rows])
        assert np.array_equal(ll[rows], ref_ll)
        # the committed half: a sigma-only proposal re-uses the committed trace
        ff2 = np.zeros(nb, dtype=np.int32)
        ll2 = eng.eval_models(np.arange(nb), k, z, dvp, dvs, 2 * sig, fwd_flag=ff2)
        got = rows[::2]
        tr = eng.get_rft_batch(got, which=0, n=p.nsmp)                 # [n, ntrc, nsmp] current traces
        obs, r_inv = p.obs[:, :p.nsmp], init_r_inv(p.nsmp, p.a_gus, p.delta)
        for j, i in enumerate(got[:40]):
            want = oracle.log_likelihood(np.pad(tr[j], ((0, 0), (0, p.nfft - p.nsmp))), obs, r_inv, 2 * sig[i], p.nsmp)
            assert abs(ll2[i] - want) <= logl_tol(want), (i, ll2[i], want)


@pytest.mark.parametrize("pinned", [False, True])
def test_copy_stream_option_same_results_with_evaluations_in_flight(oracle, golden_dir, pinned):
    """rf_set_option("copy_stream", 1): rf_eval_models_begin transfers a batch's host arrays on a stream of the
    context's own, under the kernels of the evaluation before it.  Three evaluations in flight (different halves of the
    chains, as the sampler's pipeline issues them; pageable arrays go through the slot's arena), commits in between:
    the same bits as in line."""
    import ctypes as C

    from rf_inv_amd import RFEngine, read_ref_model
    from rf_inv_amd.engine import _dptr, _iptr, host_alloc
    from rf_inv_amd.likelihood import init_r_inv

    p, ref, mcfg = _setup(golden_dir, 0.0, 0, 12)
    ref = read_ref_model(os.path.join(p.base_dir, p.vel_file))
    rng = np.random.default_rng(23)
    nb = 600
    k, z, dvp, dvs = _proposals(rng, p, nb)
    sig = rng.uniform(0.01, 0.03, (nb, p.ntrc))
    ids = np.arange(nb, dtype=np.int32)
    arrs = [ids, k, z, dvp, dvs, sig]
    if pinned:
        def pin(a):
            b = host_alloc(a.shape, a.dtype)
            b[...] = a
            return b
        arrs = [pin(a) for a in arrs]
    ids, k, z, dvp, dvs, sig = arrs
    parts = [slice(0, 200), slice(200, 400), slice(400, 600)]
    out = {}
    for on in (0, 1):
        with RFEngine.from_params(p, r_inv=init_r_inv(p.nsmp, p.a_gus, p.delta), max_walkers=nb) as eng:
            eng.set_option("copy_stream", on)
            eng.set_model(p, ref)
            L = eng._lib
            res = np.full((2, nb), np.nan)
            for rep in range(2):                       # the second round re-uses the slots and their device arrays
                tk = []
                for sl in parts:
                    t = C.c_int32(-1)
                    eng._chk(L.rf_eval_models_begin(eng._ctx, 200, _iptr(ids[sl]), None, _iptr(k[sl]), _dptr(z[sl]),
                                                    int(z.shape[1]), _dptr(dvp[sl]), _dptr(dvs[sl]), _dptr(sig[sl]), 0,
                                                    C.byref(t)))
                    tk.append(t)
                for sl, t in zip(parts, tk):
                    buf = np.empty(200)
                    eng._chk(L.rf_eval_wait(eng._ctx, t, _dptr(buf), None))
                    res[rep, sl] = buf
                eng.commit(ids, (np.arange(nb) % 2).astype(np.int32))
            assert np.array_equal(res[0], res[1], equal_nan=True)
            out[on] = res[0]
    assert np.array_equal(out[0], out[1], equal_nan=True) and np.isfinite(out[0]).sum() > 30
