"""Throughput of the split launch plan on long series: spectra_kernel -> trace_kernel (in-LDS FFT, nfft 8192),
trace_long_kernel (four-step transform: 16384 .. 65536; Bluestein: any other length beyond 2048) and, for comparison,
the direct DFT of trace_anyn_kernel at 2000.  Evaluations / s of a batch of `nb` walkers of 10 layers, one P trace.
usage: python tests/tools/long_series_rate.py [nb]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import DELTA, pack_layers, random_stack  # noqa: E402
from rf_inv_amd import RFEngine  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(7)
stacks = [random_stack(rng, 10) for _ in range(nb)]
nlay, layers = pack_layers(stacks, 12)
dev = torch.device("cuda", 0)
for nfft in (2000, 4096, 8192, 16384, 32768, 65536, 2500, 10007, 20000, 32767):
    with RFEngine(nfft=nfft, delta=DELTA, t_start=0.0, deconv_mode=0, sdep=0.0, rayps=np.array([0.06]), a_gus=np.array([4.0]),
                  ipha=np.array([1], dtype=np.int32), obs=np.zeros((1, 101)), nsmp=101, max_walkers=nb, nlay_max=12,
                  options={"fused": 0}) as eng:
        d_ids = torch.arange(nb, dtype=torch.int32, device=dev)
        d_nlay, d_layers = torch.from_numpy(nlay).to(dev), torch.from_numpy(layers).to(dev)
        d_sig = torch.full((nb, 1), 0.02, dtype=torch.float64, device=dev)
        d_logl = torch.empty(nb, dtype=torch.float64, device=dev)
        for _ in range(3):
            eng.eval_batch_device(d_ids, d_nlay, d_layers, d_sig, d_logl)
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.eval_batch_device(d_ids, d_nlay, d_layers, d_sig, d_logl)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        assert torch.isfinite(d_logl).all()
        print(f"nfft {nfft:6d}: {1e3 * dt:9.3f} ms per batch of {nb} -> {nb / dt:12.0f} evals/s")
