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
    try:
        import torch

        pr = torch.cuda.get_device_properties(0)
        bus = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        print("device 0:", pr.name, "pci", bus)
        for f in sorted(glob.glob("/sys/class/drm/card*/device")):
            if os.path.basename(os.path.realpath(f)).startswith(bus):
                print("  ->", f, "numa_node", open(os.path.join(f, "numa_node")).read().strip(),
                      "local_cpulist", open(os.path.join(f, "local_cpulist")).read().strip())
    except Exception as e:      # noqa: BLE001 (a probe: say what could not be read and go on)
        print("device lookup failed:", e)
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
                print(f"{wl}host steps {steps:5d} {name:32s}: {r['ms_per_step']:.3f} ms/step (median {r['ms_per_step_median']:.3f}, "
                      f"max {r['ms_per_step_max']:.3f}) {r['value'] / 1e6:.3f} M evals/s | "
                      f"eval_models call {ph['eval_models_call']:.3f} (kernels {ph['kernels']:.3f}) commit call "
                      f"{ph['commit_call']:.3f} | pre-warm steps {r['prewarm']['steps']}", flush=True)


if __name__ == "__main__":
    main()
