"""Offline sweep (CPU only): the oracle against the reference itself (oracle/_ref/cpu_o0, oracle/Makefile.ref) on many more
random contexts than tests/test_reference_live.py runs -- 40 seeds x 8 contexts x 6 stacks.
    python tests/tools/oracle_vs_live_reference_sweep.py  ->  one summary line (profiles/r06_oracle_vs_live_reference_sweep.txt)"""
import sys, os, tempfile, numpy as np, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_reference_live as L
from oracle import rf_oracle as oracle
oracle.build()
HIP = "--hip" in sys.argv          # also the HIP path (rf_eval_batch through the C ABI) against the same reference outputs
if HIP:
    from helpers import DELTA, pack_layers
    from rf_inv_amd import RFEngine
worst=0; n=0; bad=0; worst_hip=0
for seed in range(100,140):
    rng=np.random.default_rng(seed)
    for case in range(8):
        ctx=L.random_context(rng)
        with tempfile.TemporaryDirectory() as d:
            p,cfg,r=L.reference_traces(ctx, pathlib.Path(d))
        if HIP:
            ns, ntrc = len(ctx['stacks']), p.ntrc
            nlay, layers = pack_layers(ctx['stacks'], 33)
            with RFEngine(nfft=ctx['nfft'], delta=DELTA, t_start=ctx['t_start'], deconv_mode=ctx['deconv'], sdep=ctx['sdep'],
                          rayps=cfg['rayps'], a_gus=cfg['a_gus'], ipha=cfg['ipha'], obs=np.zeros((ntrc, p.nsmp)), nsmp=p.nsmp,
                          max_walkers=ns, nlay_max=33) as eng:
                eng.eval_batch(np.arange(ns), nlay, layers, np.full((ns, ntrc), 0.02))
                hip = eng.get_rft_batch(np.arange(ns), which=1)
        for i,st in enumerate(ctx['stacks']):
            got,npre,_,_=oracle.calc_rf(cfg,*st,want_stages=True)
            assert np.array_equal(npre,r['npre'][i]),(seed,case,i)
            scale=np.abs(r['rft'][i]).max(axis=1,keepdims=True)
            err=(np.abs(got-r['rft'][i])/scale).max()
            tol=L.allowance(oracle,cfg,st)
            n+=1
            if not err<=tol: bad+=1; print('BAD',seed,case,i,err,tol,ctx['nfft'],ctx['ipha'],ctx['deconv'],ctx['sdep'])
            if tol==1e-12: worst=max(worst,err)
            if HIP:
                eh=(np.abs(hip[i]-r['rft'][i])/scale).max()
                if not eh<=tol: bad+=1; print('BAD HIP',seed,case,i,eh,tol,ctx['nfft'],ctx['ipha'],ctx['deconv'],ctx['sdep'])
                if tol==1e-12: worst_hip=max(worst_hip,eh)
print('stacks',n,'bad',bad,'worst (well-conditioned)',worst, ('HIP worst %.3e' % worst_hip) if HIP else '')
