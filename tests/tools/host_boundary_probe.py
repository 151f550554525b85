"""Why did the PCIe-inclusive leg of bench.py (`c4host`) read 3.57 ms per step on the driver's box (20 steps after 5
warm-up steps) and 2.05 here (200 steps)?  Run length / clock state, or the box (NUMA placement of the pinned arrays)?

Prints where the GPU and the pinned memory sit (NUMA nodes), then c4host and c3host at 20 / 200 / 2000 timed steps, each
twice: the round-4 protocol (5 warm-up steps, no pre-warm by time) and the current one (>= 1 s of untimed steps first).
usage: python tests/tools/host_boundary_probe.py [workloads, default c4,c3]"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import bench

    print("host:", os.uname().nodename, "cores", bench.physical_cores(), "affinity", len(os.sched_getaffinity(0)))
    for f in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")):
        print(f, open(f).read().strip())
    for f in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")):
        props = dict(l.split()[:2] for l in open(f) if len(l.split()) >= 2)
        if int(props.get("simd_count", "0")) > 0:
            print(f, "simd_count", props["simd_count"], "location_id", props.get("location_id"), "domain", props.get("domain"))
    try:
        print(subprocess.run(["numactl", "--hardware"], capture_output=True, text=True, timeout=20).stdout.strip())
    except (OSError, subprocess.TimeoutExpired):
        print("numactl: not available;", "nodes:", sorted(os.path.basename(d) for d in glob.glob("/sys/devices/system/node/node*")))
    try:
        st = open("/proc/self/status").read()
        print([l for l in st.splitlines() if l.startswith(("Cpus_allowed_list", "Mems_allowed_list"))])
    except OSError:
        pass
    for wl in (sys.argv[1] if len(sys.argv) > 1 else "c4,c3").split(","):
        for steps in (20, 200, 2000):
            for name, kw in (("r04 protocol (5 warm-up steps)", dict(prewarm_s=0.0, warmup=5)),
                             ("pre-warmed >= 1 s", dict(prewarm_s=1.0, warmup=5))):
                r = bench.run_host_boundary(wl, steps, 0, **kw)
                ph = r["phase_ms"]
                print(f"{wl}host steps {steps:5d} {name:32s}: {r['ms_per_step']:.3f} ms/step {r['value'] / 1e6:.3f} M evals/s | "
                      f"eval_models call {ph['eval_models_call']:.3f} (kernels {ph['kernels']:.3f}) commit call "
                      f"{ph['commit_call']:.3f} | pre-warm steps {r['prewarm']['steps']}", flush=True)


if __name__ == "__main__":
    main()
