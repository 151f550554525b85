"""Compile-time checks on the gfx950 ISA of rfgpu_kernels.hip (no GPU needed: hipcc cross-compiles).

The chained-phase loop prefetches the next layer's constants into the scalar cache with inline-asm `s_load_dword`s
whose destination SGPRs the compiler believes to be written when the asm statement ends; the hardware writes them
when the load lands.  The kernel is only correct if nothing else is allocated to those registers while a load can
be in flight, i.e. anywhere in the loop and up to the `s_waitcnt` after it.  The source arranges that by making the
three registers loop-carried in/out operands; this test checks that the compiler actually kept them reserved, in
every kernel that contains the loop, and that the fused kernels a default launch plan selects do not spill."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rf_inv_amd", "csrc", "rfgpu_kernels.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "kernels.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-S", "--cuda-device-only",
           "-I", os.path.join(ROOT, "include"), "-o", str(out), "-x", "hip", SRC]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text().split("\n")


def _functions(lines):
    """name -> (first line, last line) of every kernel body."""
    out, name, start = {}, None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN5rfgpu\w+):", l)
        if m:
            name, start = m.group(1), i
        elif name and (l.strip().startswith(".Lfunc_end") or l.strip().startswith(".end_amdhsa_kernel")):
            out.setdefault(name, (start, i))
            name = None
    return out


def _sgprs(operand):
    """SGPR numbers named by one operand: s7, s[8:9], vcc_lo ... -> set of ints (vcc etc.: empty)."""
    m = re.fullmatch(r"s(\d+)", operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def test_scalar_prefetch_registers_stay_reserved(isa):
    funcs = _functions(isa)
    checked = 0
    for name, (a, b) in funcs.items():
        body = isa[a:b]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        # innermost loops: backward branches
        latch = {}   # loop header line -> last backward branch to it
        for i, l in enumerate(body):
            m = re.search(r"\ts_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                latch[labels[m.group(1)]] = i
        loops = sorted(latch.items())
        i = 0
        while i < len(body):
            if "s_load_dword " in body[i] and body[i - 1].strip().startswith(";;#ASMSTART"):
                dests = set()
                j = i
                while not body[j].strip().startswith(";;#ASMEND"):
                    ops = body[j].split(None, 1)[1].split(",")
                    dests |= _sgprs(ops[0].strip())
                    j += 1
                assert len(dests) == 3, (name, body[i:j])
                enclosing = [lp for lp in loops if lp[0] <= i <= lp[1]]
                assert enclosing, (name, "prefetch outside a loop")
                lo, hi = min(enclosing, key=lambda lp: lp[1] - lp[0])

                def touches(k):
                    t = body[k].strip()
                    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                        return False
                    parts = t.split(None, 1)
                    return len(parts) == 2 and any(_sgprs(op.strip().lstrip("-|").rstrip("|")) & dests
                                                   for op in parts[1].split(","))

                # (a) nothing else names the three registers inside the loop
                for k in list(range(lo, i - 1)) + list(range(j + 1, hi + 1)):
                    assert not touches(k), (name, "prefetch destination touched inside the loop", body[k].strip())
                # (b) every way out of the loop reaches the retiring s_waitcnt asm before anything names them
                exits = []
                for k in range(lo, hi + 1):
                    m = re.search(r"\ts_c?branch\w*\s+(\.LBB\d+_\d+)", body[k])
                    if m and not (lo <= labels[m.group(1)] <= hi):
                        exits.append(labels[m.group(1)])
                if not body[hi].strip().startswith("s_branch"):
                    exits.append(hi + 1)
                assert exits, name
                for e in exits:
                    k, steps, hops = e, 0, 0
                    while not ("s_waitcnt lgkmcnt(0)" in body[k] and body[k - 1].strip().startswith(";;#ASMSTART")):
                        assert not touches(k), (name, "prefetch destination touched before the loads are retired", body[k].strip())
                        jump = re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", body[k])
                        if jump:
                            k = labels[jump.group(1)]      # an unconditional jump is one path: follow it
                            hops += 1
                            continue
                        assert not re.search(r"\ts_c?branch|s_endpgm|s_setpc", body[k]), (name, "no retiring wait on a loop exit", body[k].strip())
                        k += 1
                        steps += 1
                        assert steps < 60 and hops < 4, (name, "retiring wait not found after the loop")
                checked += 1
                i = j
            i += 1
    assert checked >= 10, checked   # every chained-phase kernel variant carries the loop


def test_default_plan_kernels_do_not_spill(isa):
    text = "\n".join(isa)
    for kern in ("_ZN5rfgpu12fused_kernelILi8ELi2EEEvNS_11FusedParamsE", "_ZN5rfgpu12fused_kernelILi4ELi3EEEvNS_11FusedParamsE",
                 "_ZN5rfgpu13fused8_kernelILi2EEEvNS_11FusedParamsE", "_ZN5rfgpu14spectra_kernelILi4ELi2EEEvNS_13SpectraParamsE"):
        # (fused_kernel<8, 3>, the ocean default since round 3, holds 192 state registers: see the next test)
        m = re.search(r"\.name:\s+" + kern + r"\n(?:.*\n){1,20}?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)
        assert m, kern
        assert int(m.group(2)) == 0, (kern, "vgpr spills", m.group(2))
    m = re.search(r"\.name:\s+_ZN5rfgpu13fused8_kernelILi2EEEvNS_11FusedParamsE\n(?:.*\n){1,20}?\s+\.vgpr_count:\s+(\d+)", text)
    assert int(m.group(1)) <= 128, "fused8_kernel must fit four waves per SIMD"
    # the common-ray kernel keeps 32 registers of spectra alive across its trace tails: a handful of spills, 128 VGPRs
    m = re.search(r"\.name:\s+_ZN5rfgpu13fusedc_kernelILi2EEEvNS_11FusedParamsE\n(?:.*\n){1,20}?\s+\.vgpr_count:\s+(\d+)\n"
                  r"\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert m and int(m.group(1)) <= 128 and int(m.group(2)) <= 8, m and m.groups()


def test_ocean_eight_bin_chain_spills_only_outside_the_layer_loop(isa):
    """fused_kernel<8, 3> (three propagated columns x eight bins = 192 state VGPRs) may park a few loop-invariant
    values in scratch around the layer loop, never inside it."""
    text = "\n".join(isa)
    kern = "_ZN5rfgpu12fused_kernelILi8ELi3EEEvNS_11FusedParamsE"
    m = re.search(r"\.name:\s+" + kern + r"\n(?:.*\n){1,20}?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert m, kern
    assert int(m.group(2)) <= 24, ("vgpr spills", m.group(2))
    a, b = _functions(isa)[kern]
    body = isa[a:b]
    labels = {mm.group(1): i for i, l in enumerate(body) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
    loops = []
    for i, l in enumerate(body):
        mm = re.search(r"\ts_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i))
    pref = [i for i, l in enumerate(body) if "s_load_dword " in l and body[i - 1].strip().startswith(";;#ASMSTART")]
    assert pref
    for i in pref:
        lo, hi = min((lp for lp in loops if lp[0] <= i <= lp[1]), key=lambda lp: lp[1] - lp[0])
        assert not any("scratch_" in l for l in body[lo:hi + 1]), "scratch access inside the layer loop"
