"""Can the reference's own calc_likelihood (oracle/_ref/ref_path_time: its c2r goes through the drop-in module fftw, i.e.
a synchronous round trip to the GPU per transform) be timed on ALL host cores at once?  P concurrent processes, one per
core, with and without HSA settings that reduce the queues a process opens (HSA_ENABLE_SDMA=0, GPU_MAX_HW_QUEUES=1): more
host processes than the GPU keeps queues mapped for are time-sliced, and the round trips then measure the slicing.
usage: python tests/tools/reference_path_scaling.py [workload, default c4] [reps, default 2] [hold: the parent keeps an
       engine context open on the GPU like bench.py does while it times its CPU leg]"""
import copy
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import bench
    from oracle import rf_oracle as orc
    from rf_inv_amd import format_model, read_ref_model, write_params
    from rf_inv_amd.make_syn import write_sac

    wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
    many = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    hold = len(sys.argv) > 3 and sys.argv[3] == "hold"
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_path_time")
    p = bench.make_params(dict(bench.WORKLOADS[wl]))
    golden = os.path.join(ROOT, "tests", "golden", "sample_syn")
    ref = read_ref_model(os.path.join(golden, "model", "sample.velmod"))
    n = 24
    nlay, _, (m_k, m_z, m_dvp, m_dvs) = bench.draw_walkers(p, ref, 0, n, return_models=True)
    orc.build()
    cfg = dict(nfft=p.nfft, deconv_mode=p.deconv_mode, delta=p.delta, t_start=p.t_start, sdep=p.sdep, rayps=p.rayps,
               a_gus=p.a_gus, ipha=p.ipha)
    zt = np.zeros(max(p.k_max - 1, 1)); dvt = np.zeros(p.k_max); dst = np.zeros(p.k_max)
    zt[:3] = [3.1 + p.sdep, 7.7 + p.sdep, 14.2 + p.sdep]; dst[:3] = [-0.6, 0.2, 0.5]; dst[p.k_max - 1] = 0.9
    nl_t, a_t, b_t, r_t, h_t, ok = format_model(p, ref, 3, zt, dvt, dst)
    obs = orc.calc_rf(cfg, a_t, b_t, r_t, h_t)
    with tempfile.TemporaryDirectory() as work:
        for d in ("data", "rslt", "model"):
            os.makedirs(os.path.join(work, d))
        shutil.copy(os.path.join(golden, "model", "sample.velmod"), os.path.join(work, "model", "sample.velmod"))
        q = copy.copy(p)
        q.out_dir, q.nchains, q.ncool, q.nburn, q.niter, q.dvs_prior = "./rslt", 1, 1, 0, 10, 0.3
        q.vel_file, q.obs_files = "model/sample.velmod", [f"data/t{t + 1}.trc" for t in range(p.ntrc)]
        for t, f in enumerate(q.obs_files):
            write_sac(os.path.join(work, f), obs[t, :p.nsmp], p.delta, p.t_start, p.t_end)
        write_params(os.path.join(work, "params.in"), q)
        with open(os.path.join(work, "models.txt"), "w") as fh:
            fh.write(f"{n}\n")
            for i in range(n):
                fh.write(f"{int(m_k[i])}\n")
                for arr in (m_z[i, :max(p.k_max - 1, 1)], m_dvp[i, :p.k_max], m_dvs[i, :p.k_max], np.full(p.ntrc, 0.01)):
                    fh.write(" ".join(repr(float(x)) for x in arr) + "\n")
        cores = bench.physical_cores()
        eng = None
        if hold:
            import torch

            from rf_inv_amd import RFEngine

            eng = RFEngine(nfft=p.nfft, delta=p.delta, t_start=p.t_start, deconv_mode=p.deconv_mode, sdep=p.sdep, rayps=p.rayps,
                           a_gus=p.a_gus, ipha=p.ipha, obs=obs[:, :p.nsmp].copy(), nsmp=p.nsmp, max_walkers=8192, nlay_max=p.k_max + 2)
            x = torch.zeros(1 << 20, device="cuda")
            torch.cuda.synchronize()
            print("parent holds an engine context (8192 walkers) and a torch context", flush=True)
        for label, env in (("default", {}), ("HSA_ENABLE_SDMA=0", {"HSA_ENABLE_SDMA": "0"}),
                           ("HSA_ENABLE_SDMA=0 GPU_MAX_HW_QUEUES=1", {"HSA_ENABLE_SDMA": "0", "GPU_MAX_HW_QUEUES": "1"})):
            for procs in (1, 4, cores):
                reps = 2 * many if procs == 1 else many
                t0 = time.perf_counter()
                runs = [subprocess.Popen([exe, "params.in", "models.txt", f"ref_{i}.bin", str(reps)], cwd=work,
                                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=dict(os.environ, **env))
                        for i in range(procs)]
                outs = [r.communicate(timeout=1200)[0] for r in runs]
                wall = time.perf_counter() - t0
                each = []
                for o in outs:
                    line = [l for l in o.splitlines() if "ref_path_dump: seconds" in l]
                    if line:
                        tok = line[0].split()
                        each.append((float(tok[2]), int(tok[4])))
                if len(each) != procs:
                    print(f"{wl} {label:40s} procs {procs:3d}: {procs - len(each)} process(es) failed", flush=True)
                    continue
                secs, evals = max(e[0] for e in each), sum(e[1] for e in each)
                print(f"{wl} {label:40s} procs {procs:3d}: {evals / secs:8.1f} evals/s in all, {evals / secs / procs:7.2f} per process "
                      f"(slowest {secs:.1f} s, wall {wall:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
