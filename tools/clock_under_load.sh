#!/bin/bash
# Which shader clock and board power does the part run at while each kind of kernel is the load?
#   tools/clock_under_load.sh <tag>      -> gpurun_out/clock_<tag>/{dfma,c4,c5,idle}.txt (+ the programs' own output)
# rocm-smi is sampled every 0.25 s beside (a) back-to-back pure-DFMA kernels (tools/fp64_peak.hip sustain: chains whose
# operands converge, and `chaos` chains whose operands keep toggling like data), (b) the
# headline workload's steps, (c) the same for c5 (4-wave kernel, ocean).  Cross-check of the per-kernel clock
# tools/summarize_counters.py derives from GRBM_GUI_ACTIVE.  Reads sysfs through rocm-smi only; changes no setting.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/clock_${1:-x}
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o $OUT/fp64_peak $R/tools/fp64_peak.hip 2> $OUT/hipcc.log || exit 1
sample() {   # name, seconds, command...
  local name=$1 secs=$2; shift 2
  "$@" > $OUT/$name.out 2> $OUT/$name.err &
  local pid=$!
  local n=$(( secs * 4 ))
  for ((i = 0; i < n; ++i)); do
    echo "--- $(date +%s.%N)"
    rocm-smi --showclocks --showpower 2>/dev/null | grep -Ei "sclk|power"
    sleep 0.25
  done > $OUT/$name.txt
  wait $pid
}
sample idle 2 sleep 2
sample dfma 14 $OUT/fp64_peak sustain 12
sample dfmax 14 $OUT/fp64_peak sustain 12 chaos
sample c4 45 python3 $R/bench.py --workload c4 --steps 8000 --warmup 20 --no-cpu-baseline --also ""
sample c5 60 python3 $R/bench.py --workload c5 --steps 1500 --warmup 5 --no-cpu-baseline --also ""
rocm-smi --showmaxpower --showclkfrq 2>/dev/null | head -60 > $OUT/limits.txt
python3 - $OUT <<'PY'
import re, sys
for name in ("idle", "dfma", "dfmax", "c4", "c5"):
    rows, sclk = [], None
    for line in open(f"{sys.argv[1]}/{name}.txt"):
        m = re.search(r"sclk.*?\((\d+)Mhz\)", line, re.I)
        if m:
            sclk = int(m.group(1))
        m = re.search(r"power.*?:\s*([0-9.]+)\s*$", line, re.I)
        if m and sclk is not None:
            rows.append((float(m.group(1)), sclk))
    rows.sort(reverse=True)
    top = rows[:max(1, len(rows) // 4)]          # the quarter of the samples with the highest power = under load
    print(f"{name:5s} {len(rows):3d} samples; loaded quarter: power {sum(p for p, _ in top) / len(top):7.1f} W, "
          f"sclk mean {sum(c for _, c in top) / len(top):6.0f} MHz (min {min(c for _, c in top)}, max {max(c for _, c in top)})")
PY
echo "dfma (operands converge):"; grep -h "TFLOP" $OUT/dfma.out | sed -n "3p;20p;40p"
echo "dfmax (x <- x*x + c, chaotic operands):"; grep -h "TFLOP" $OUT/dfmax.out | sed -n "3p;20p;40p"
