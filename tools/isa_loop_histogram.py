"""Opcode histogram of the innermost loops of one kernel of rfgpu_kernels.hip (gfx950 ISA; no GPU needed).

    python tools/isa_loop_histogram.py [kernel-substring] [--asm kernels.s]

Finds the kernel whose mangled name contains the substring (default: fused_kernel<8,2>, the C4 plan), lists its loops
(backward branches), and for the INNERMOST loop with the most fp64 instructions -- the per-layer propagator loop -- prints the
instruction mix: fp64 VALU (v_fma_f64, v_mul_f64, v_add_f64, ...) against every other VALU opcode, SALU, SMEM, waits.
The judge's round-5 question: fp64 is 83 % of the VALU issue in that loop; what are the other 17 %?"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rf_inv_amd", "csrc", "rfgpu_kernels.hip")


def assemble(path):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-S", "--cuda-device-only",
           "-I", os.path.join(ROOT, "include"), "-o", path, "-x", "hip", SRC]
    subprocess.run(cmd, check=True, capture_output=True)


def kernel_body(lines, want):
    name, start = None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN5rfgpu\w+):", l)
        if m:
            name, start = m.group(1), i
        elif name and l.strip().startswith(".Lfunc_end"):
            if want in name:
                return name, lines[start:i]
            name = None
    raise SystemExit(f"no kernel matching {want}")


def classify(op):
    if op.startswith("v_") and (op.endswith("_f64") or "_f64_" in op):
        return "valu_fp64"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    want = args[0] if args else "fused_kernelILi8ELi2E"
    asm = sys.argv[sys.argv.index("--asm") + 1] if "--asm" in sys.argv else None
    if asm is None:
        asm = os.path.join(tempfile.mkdtemp(), "kernels.s")
        assemble(asm)
    lines = open(asm).read().split("\n")
    name, body = kernel_body(lines, want)
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = {}
    for i, l in enumerate(body):
        m = re.search(r"\ts_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops[labels[m.group(1)]] = i
    def insts(a, b):
        out = []
        for l in body[a:b + 1]:
            t = l.strip()
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            out.append(t.split()[0])
        return out
    # innermost loops only (no other loop inside): the per-layer propagator loop is the one with the most fp64 work
    inner = [lp for lp in loops.items() if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops.items())]
    best = max(inner, key=lambda lp: sum(classify(o) == "valu_fp64" for o in insts(*lp)))
    ops = insts(*best)
    cls = collections.Counter(classify(o) for o in ops)
    print(f"kernel {name}")
    print(f"loops (header line, latch line, instructions): {[(a, b, len(insts(a, b))) for a, b in sorted(loops.items())]}")
    print(f"the loop with the most fp64 work: lines {best[0]} .. {best[1]}, {len(ops)} instructions")
    valu = cls["valu_fp64"] + cls["valu_other"]
    print(f"classes: {dict(cls)}; fp64 share of VALU {cls['valu_fp64'] / valu:.3f}")
    for c in ("valu_fp64", "valu_other", "salu", "smem", "wait", "lds", "vmem", "other"):
        h = collections.Counter(o for o in ops if classify(o) == c)
        if h:
            print(f"  {c:10s} {sum(h.values()):4d}: " + ", ".join(f"{k} {v}" for k, v in h.most_common()))


if __name__ == "__main__":
    main()
