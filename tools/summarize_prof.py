#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh output directory into a text summary:
per-kernel stats (rocprofv3 --stats) and per-launch HBM bytes from the PMC passes
(FETCH_SIZE doubled per MI355X_MICROARCH.md: on gfx950 it reports half of the bytes of
a wide coalesced read; both counters are in KiB-like units of 1024 B... see below)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== rocprofv3 --kernel-trace --stats ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows:
        print(f"{r.get('Name', '?')[:90]:90s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} "
              f"avg_ns={r.get('AverageNs')} pct={r.get('Percentage')} min={r.get('MinNs')} max={r.get('MaxNs')}")

for f in find("trace/**/*kernel_trace.csv"):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    seen = {}
    for r in rows:
        k = r.get("Kernel_Name", "?")
        if k not in seen:
            seen[k] = r
    print("\n== per-kernel launch shape (first dispatch) ==")
    for k, r in seen.items():
        print(f"{k[:70]:70s} grid={r.get('Grid_Size_X', r.get('Grid_Size'))} wg={r.get('Workgroup_Size_X', r.get('Workgroup_Size'))} "
              f"lds={r.get('LDS_Block_Size')} vgpr={r.get('VGPR_Count')} accum_vgpr={r.get('Accum_VGPR_Count')} "
              f"sgpr={r.get('SGPR_Count')} scratch={r.get('Scratch_Size', r.get('Private_Segment_Size'))}")

for name, pat in (("FETCH_SIZE", "pmc_fetch/**/*counter_collection.csv"), ("WRITE_SIZE", "pmc_write/**/*counter_collection.csv")):
    for f in find(pat):
        agg = defaultdict(lambda: [0, 0.0])
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != name:
                    continue
                k = r.get("Kernel_Name", "?")
                agg[k][0] += 1
                agg[k][1] += float(r.get("Counter_Value", 0))
        print(f"\n== {name} per launch (raw counter value; rocprofv3 reports it in KiB) ==")
        for k, (n, v) in agg.items():
            per = v / max(n, 1)
            note = " (x2 gfx950 correction for reads => %.1f KiB)" % (2 * per) if name == "FETCH_SIZE" else ""
            print(f"{k[:70]:70s} launches={n} avg={per:.1f}{note}")


# machine-readable HBM traffic of the full-size launches (max over launches drops the tiny
# set-up launch): bytes = FETCH_SIZE * 1024 * 2 (gfx950 read correction, MI355X_MICROARCH.md) +
# WRITE_SIZE * 1024
import json
traffic = {}
for name, pat in (("FETCH_SIZE", "pmc_fetch/**/*counter_collection.csv"), ("WRITE_SIZE", "pmc_write/**/*counter_collection.csv")):
    for f in find(pat):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != name or "rfgpu" not in r.get("Kernel_Name", ""):
                    continue
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                d = traffic.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
                d[name] = max(d[name], float(r["Counter_Value"]))
for k, d in traffic.items():
    d["hbm_bytes_per_launch"] = d["FETCH_SIZE"] * 1024 * 2 + d["WRITE_SIZE"] * 1024
with open(os.path.join(out, "hbm_traffic.json"), "w") as fh:
    json.dump(traffic, fh, indent=1)
print("\n== HBM bytes per full-size launch (FETCH x2 + WRITE, KiB -> bytes) ==")
for k, d in traffic.items():
    print(f"{k[:60]:60s} {d['hbm_bytes_per_launch'] / 1e6:10.2f} MB")
