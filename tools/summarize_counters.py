#!/usr/bin/env python3
"""Condense a tools/collect_counters.sh output directory: per (kernel, launch shape) the mean of every
counter over that shape's launches -> counters.json (+ a text summary with rocprofv3's kernel stats and
the derived figures bench.py's roofline uses).  Launch shapes separate the workloads of one bench.py run
(and its 1-walker set-up launches) from each other."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

out, root = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
meta = {}
gui = defaultdict(list)      # (kernel, grid, wg) -> [(GRBM_GUI_ACTIVE summed over the 8 XCDs, duration in ns)], every kernel
XCDS = 8
for f in glob.glob(os.path.join(out, "p*/**/*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            k = k.split("(")[0].replace("void ", "").strip()
            key = (k, int(r["Grid_Size"]), int(r["Workgroup_Size"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("End_Timestamp"):
                gui[key].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
            if "rfgpu" not in k:
                continue
            a = agg[key][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            meta[key] = {"lds": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"]), "vgpr": int(r["VGPR_Count"]),
                         "sgpr": int(r["SGPR_Count"])}
lib = os.path.join(root, "rf_inv_amd", "lib", "librfgpu.so")
sys.path.insert(0, root)
from rf_inv_amd._lib import kernels_sha256  # noqa: E402  (ELF parsing only: loads nothing)

doc = {"lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None,
       # the library's .hip_fatbin section alone: counters are a property of the kernels, bench.py matches on this
       "kernels_sha256": kernels_sha256(lib) if os.path.exists(lib) else None,
       "source": "tools/collect_counters.sh: rocprofv3 --pmc <set> (one set per run; FETCH_SIZE and WRITE_SIZE in separate "
                 "passes) over `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --prewarm-seconds 0`; values are "
                 "means per launch over the launches of that (kernel, grid) shape; FETCH_SIZE / WRITE_SIZE in KiB as "
                 "rocprofv3 reports them (HBM bytes = 2048 * FETCH_SIZE + 1024 * WRITE_SIZE: gfx950 read correction)",
       "kernels": []}
# Shader clock a kernel actually ran at = busy cycles per XCD / its duration in the same pass.  Every dispatch of a
# --pmc pass carries a constant bracket of busy cycles (the shortest dispatches are nothing else): subtracted.  Only
# for shapes that run > 100 us, where that bracket is < 10 % of the count.  `clock_reference`: the longest
# non-rfgpu dispatch of the pass (an HBM-bound fill) -- what the clock is when the vector ALUs are not the load.
bracket = min((g / XCDS for v in gui.values() for g, _ in v), default=0.0)


def clock_ghz(v):
    d = sum(t for _, t in v) / len(v)
    return round(sum(g / XCDS - bracket for g, _ in v) / sum(t for _, t in v), 3) if d > 1e5 else None


other = [(sum(t for _, t in v) / len(v), k) for k, v in gui.items() if "rfgpu" not in k[0]]
if other:
    d, k = max(other)
    doc["clock_reference"] = {"kernel": k[0], "grid_threads": k[1], "mean_us": round(d / 1e3, 1), "clock_ghz": clock_ghz(gui[k])}
doc["clock_method"] = ("GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3) / 8, minus the per-dispatch bracket "
                       f"{bracket:.0f} cycles, / (End_Timestamp - Start_Timestamp) of the same dispatch")
for (k, grid, wg), ctrs in sorted(agg.items()):
    e = {"kernel": k, "grid_threads": grid, "workgroup": wg, **meta[(k, grid, wg)],
         "launches": max(n for n, _ in ctrs.values()), "counters": {c: v / n for c, (n, v) in sorted(ctrs.items())}}
    if gui.get((k, grid, wg)):
        e["clock_ghz"] = clock_ghz(gui[(k, grid, wg)])
        e["mean_s"] = sum(t for _, t in gui[(k, grid, wg)]) / len(gui[(k, grid, wg)]) * 1e-9   # dispatch duration in that pass
    doc["kernels"].append(e)
json.dump(doc, open(os.path.join(out, "counters.json"), "w"), indent=1)

print("== rocprofv3 --kernel-trace --stats ==")
for f in glob.glob(os.path.join(out, "trace/**/*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r.get('Name', '?')[:80]:80s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} pct={r.get('Percentage')} "
              f"min={r.get('MinNs')} max={r.get('MaxNs')}")
# per-shape average durations from the trace (the stats table mixes the shapes of one kernel)
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "trace/**/*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "rfgpu" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            dur[(k, int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("\n== per launch shape: mean duration under the kernel trace ==")
for (k, g), v in sorted(dur.items()):
    v = sorted(v)
    print(f"{k[:60]:60s} grid={g:9d} n={len(v):4d} mean={sum(v) / len(v) / 1e3:10.1f} us  median={v[len(v) // 2] / 1e3:10.1f} us")
print("\n== derived per launch shape ==")
for e in doc["kernels"]:
    c = e["counters"]
    line = f"{e['kernel'][:50]:50s} grid={e['grid_threads']:9d} vgpr={e['vgpr']} lds={e['lds']} scratch={e['scratch']}"
    if "SQ_INSTS_VALU_FMA_F64" in c:
        fl = 64 * (c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_TRANS_F64", 0)
                   + 2 * c["SQ_INSTS_VALU_FMA_F64"])
        f64 = sum(c.get(x, 0) for x in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64",
                                        "SQ_INSTS_VALU_FMA_F64"))
        line += f" exec_fp64_gflop={fl / 1e9:.3f} fp64_wave_insts={f64:.3e} valu_wave_insts={c.get('SQ_INSTS_VALU', 0):.3e}"
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        line += (f" wait_any={c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:.2f}"
                 f" active_valu={c.get('SQ_ACTIVE_INST_VALU', 0) / c['SQ_WAVE_CYCLES']:.2f}")
    if c.get("SQ_INSTS_MFMA") or c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # FP64 matrix cores (phi_gemm_kernel): MFMA wave-instructions, the pipe's busy cycles per instruction and its
        # utilisation = busy cycles / (dispatch duration x clock x 1024 SIMDs), when the clock of the shape is known
        line += f" mfma_insts={c.get('SQ_INSTS_MFMA', 0):.3e} mfma_busy_cycles={c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3e}"
        if c.get("SQ_INSTS_VALU_MFMA_MOPS_F64"):
            line += f" mfma_mops_f64={c['SQ_INSTS_VALU_MFMA_MOPS_F64']:.3e}"
        if c.get("SQ_INSTS_MFMA") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            line += f" busy_cycles_per_mfma={c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_INSTS_MFMA']:.1f}"
    if c.get("TCC_REQ_sum"):
        line += (f" l2_req={c['TCC_REQ_sum']:.3e} l2_hit_rate="
                 f"{c.get('TCC_HIT_sum', 0) / max(1.0, c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0)):.3f}")
    if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
        line += f" lds_conflict={c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.2f}"
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        line += f" hbm_MB={(2048 * c['FETCH_SIZE'] + 1024 * c['WRITE_SIZE']) / 1e6:.1f}"
    if e.get("clock_ghz"):
        line += f" clock_ghz={e['clock_ghz']}"
    print(line)
if doc.get("clock_reference"):
    print(f"\nclock reference: {doc['clock_reference']}\n({doc['clock_method']})")
