#!/usr/bin/env python3
"""The headline table of profiles/README.md from a committed full bench record (profiles/rNN_bench_n1_detail.json), so
that every number in the table is one the file holds.   usage: tools/bench_table.py profiles/r05_bench_n1_detail.json"""
import json
import sys

NAMES = {"c4": "**C4 (default)**", "c2": "C2", "c2d": "C2, water-level deconvolution", "c3": "C3",
         "c5": "C5 (32768 walkers, ocean, 4 traces)", "c4common": "C4 common ray (single FWD)",
         "c4w20": "C4, 20 s window (nsmp 401)", "c4w60": "C4, 60 s window (nsmp 1201)", "c4win": "C4 with `trace_window`",
         "c5win": "C5 with `trace_window`", "c4stale": "C4, depths changing every step",
         "c4d": "C4, water-level deconvolution", "c5d": "C5, water-level deconvolution",
         "c4full": "**ALL of BASELINE configs[3] on one GPU** (65 536 walkers x 3 traces)",
         "c5full": "**ALL of BASELINE configs[4] on one GPU** (262 144 walkers x 4 traces, ocean; 68.7 GB of traces)"}


def row(name, r):
    roof = r.get("roofline") or {}
    km = r.get("kernel_ms") or {}
    f = lambda x, n=3: "—" if x is None else f"{x:.{n}f}"
    follow = km.get("quadratic_form_logl")
    gemm = r.get("quadratic_form_gemm")
    extra = f"{follow:.3f}" if follow else "—"
    if gemm:
        extra = f"GEMM + logL {gemm['ms']:.3f} ({gemm['frac']:.2f} of the FP64 matrix peak in executed multiply-adds)"
    traffic = "—" if roof.get("traffic") is None else f"{roof['traffic'] / 1e6:.0f}"
    return (f"| {NAMES.get(name, name)} | {r['value'] / 1e6:.2f} M | {r['ms_per_step']:.4g} | `{(roof.get('kernel') or '').replace('rfgpu::', '')}` | "
            f"{f(roof.get('kernel_ms'), 4)} | {f(roof.get('frac'))} ({f(roof.get('frac_step'))}) | {f(roof.get('clock_ghz'))} ({f(roof.get('frac_at_clock'))}) | "
            f"{traffic} | {extra} |")


def main():
    d = json.load(open(sys.argv[1]))
    print("| workload | evals/s/GPU | ms/step | dominant kernel | kernel ms | frac of 78.6 TF (step-level) | clock GHz in the counter pass (frac at it) | HBM MB/launch | follow-up kernel ms |")
    print("|---|---|---|---|---|---|---|---|---|")
    print(row("c4", d))
    for k, v in d.get("also", {}).items():
        if k.endswith("host"):
            ph = v.get("phase_ms", {})
            print(f"| {k}: the walkers from pinned HOST arrays every step (PCIe included, synchronous, no swap) | {v['value'] / 1e6:.2f} M | "
                  f"{v['ms_per_step']:.4g} (median {v.get('ms_per_step_median', 0):.4g}, max {v.get('ms_per_step_max', 0):.3g} of {v['steps']}) | — | "
                  f"kernels {ph.get('kernels', 0):.3f} of the {ph.get('eval_models_call', 0):.3f} ms `rf_eval_models` call; `rf_commit` call "
                  f"{ph.get('commit_call', 0):.3f} | — | — | {v['config']['host_bytes_in_per_step'] / 1e6:.1f} MB in, "
                  f"{v['config']['host_bytes_out_per_step'] / 1e3:.0f} KB out per step | — |")
        else:
            print(row(k, v))
    c = d.get("cpu_baseline") or {}
    print()
    print(f"cpu_baseline: {c.get('value', 0):.0f} evals/s on {c.get('cores')} cores (kind {c.get('kind')}, {c.get('per_core', 0):.1f} per core, "
          f"{c.get('single_core', 0):.1f} on one core alone)", end="")
    q = c.get("port")
    if q:
        r = c.get("reference_over_port") or {}
        print(f"; the C port beside it: {q['value']:.0f} evals/s on {q['cores']} cores ({q['per_core']:.0f} per core; {q.get('single_core_same_walkers', 0):.0f} on "
              f"one core on the reference's sample); reference / port: {r.get('all_cores', 0):.2f} on all cores, {r.get('single_core_same_walkers', 0):.2f} "
              f"on one; max relative |dlogL| port vs reference {c.get('max_rel_dlogl_port_vs_reference', 0):.2e}")
    else:
        print()
    p = d.get("parity_in_bench") or {}
    print(f"parity_in_bench: n {p.get('n')}, max relative |dlogL| {p.get('max_rel_dlogl'):.2e}, within tolerance {p.get('within_tolerance')}, "
          f"kappa allowance used by {p.get('n_used_kappa_allowance')}")


if __name__ == "__main__":
    main()
