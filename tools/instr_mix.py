#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels (no GPU needed): compiles a .hip source of plonkathon_amd/csrc to assembly
with the product's flags and reports, per kernel, instruction count, VGPRs / spills, multiplier instructions and — for
the kernels whose body is one big loop (the MSM kernels) — the mix of the largest loops.  This is where DESIGN.md 3's
"2 420 instructions per mixed addition" comes from.

usage: python tools/instr_mix.py msm.hip [kernel-substring]      (writes nothing; prints a report)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(REPO, "plonkathon_amd", "csrc")


def compile_asm(src):
    out = os.path.join(tempfile.mkdtemp(prefix="instr_mix_"), "k.s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-S", "--cuda-device-only", "-o", out,
           os.path.join(CSRC, src)]
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def instructions(lines):
    for line in lines:
        t = line.strip()
        if line.startswith("\t") and t and not t.startswith((".", ";")):
            yield t.split()[0]


def kernels(asm):
    for m in re.finditer(r"^(_Z\w+):.*?; @", asm, re.M):
        name, i = m.group(1), m.start()
        body = asm[i:asm.find("s_endpgm", i)].split("\n")
        meta = asm[asm.find(".name:           " + name) - 1500:][:3000]
        grab = lambda key: (re.search(key + r":\s+(\d+)", meta) or [None, "?"])[1]
        yield name, body, grab(r"\.vgpr_count"), grab(r"\.vgpr_spill_count"), grab(r"\.sgpr_count")


def loops(body):
    labels = {m.group(1): n for n, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    out = []
    for n, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            out.append((n - labels[m.group(1)], labels[m.group(1)], n))
    return sorted(out, reverse=True)


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "msm.hip"
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    asm = compile_asm(src)
    for name, body, vgpr, spill, sgpr in kernels(asm):
        if want not in name:
            continue
        ins = list(instructions(body))
        c = collections.Counter(ins)
        mads = sum(v for k, v in c.items() if k.startswith(("v_mad_u64", "v_mad_i64")))
        print("%s\n  %d instructions, %d VALU, %d v_mad_[ui]64, %d v_mul_lo_u32, %d v_lshl_add_u64; VGPRs %s (spilled %s), SGPRs %s" % (
            name, len(ins), sum(v for k, v in c.items() if k.startswith("v_")), mads, c["v_mul_lo_u32"], c["v_lshl_add_u64"],
            vgpr, spill, sgpr))
        seen = set()
        for size, a, b in loops(body)[:6]:
            if size < 500 or any(a >= x and b <= y for x, y in seen):  # nested inside a reported loop
                continue
            seen.add((a, b))
            lc = collections.Counter(instructions(body[a:b + 1]))
            top = ", ".join("%s %d" % kv for kv in lc.most_common(8))
            print("  loop of %d instructions (%d VALU): %s" % (sum(lc.values()), sum(v for k, v in lc.items() if k.startswith("v_")), top))


if __name__ == "__main__":
    main()
