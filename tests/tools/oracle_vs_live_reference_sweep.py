"""Offline sweep (CPU only): the oracle against the reference itself (oracle/_ref/cpu_o0, oracle/Makefile.ref) on many more
random contexts than tests/test_reference_live.py runs -- 40 seeds x 8 contexts x 6 stacks.
    python tests/tools/oracle_vs_live_reference_sweep.py  ->  one summary line (profiles/r06_oracle_vs_live_reference_sweep.txt)"""
import sys, os, tempfile, numpy as np, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_reference_live as L
from oracle import rf_oracle as oracle
oracle.build()
worst=0; n=0; bad=0
for seed in range(100,140):
    rng=np.random.default_rng(seed)
    for case in range(8):
        ctx=L.random_context(rng)
        with tempfile.TemporaryDirectory() as d:
            p,cfg,r=L.reference_traces(ctx, pathlib.Path(d))
        for i,st in enumerate(ctx['stacks']):
            got,npre,_,_=oracle.calc_rf(cfg,*st,want_stages=True)
            assert np.array_equal(npre,r['npre'][i]),(seed,case,i)
            scale=np.abs(r['rft'][i]).max(axis=1,keepdims=True)
            err=(np.abs(got-r['rft'][i])/scale).max()
            tol=L.allowance(oracle,cfg,st)
            n+=1
            if not err<=tol: bad+=1; print('BAD',seed,case,i,err,tol,ctx['nfft'],ctx['ipha'],ctx['deconv'],ctx['sdep'])
            if tol==1e-12: worst=max(worst,err)
print('stacks',n,'bad',bad,'worst (well-conditioned)',worst)
