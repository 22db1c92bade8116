#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of libplonk_hip.so's sources, from hipcc's own
`-Rpass-analysis=kernel-resource-usage` remarks (static, no GPU needed).

usage: python tools/resource_usage.py [file.hip ...] > profiles/rNN_kernel_resource_usage.txt"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "plonkathon_amd", "csrc")
FILES = sys.argv[1:] or ["ntt.hip", "ntt_bls.hip", "g1_codec.hip", "fr_ops.hip", "msm.hip", "prover.hip", "api.hip", "transcript_api.hip"]
KEYS = ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "SGPRs Spill", "VGPRs Spill")

print("%-78s %6s %6s %6s %8s %5s %8s %7s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS", "sSpill", "vSpill"))
for f in FILES:
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-Rpass-analysis=kernel-resource-usage",
                        "-c", f, "-o", "/dev/null"], cwd=CSRC, capture_output=True, text=True)
    cur = None
    rows = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur).replace("void ", "")
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    for name, d in rows.items():
        print("%-78s %6s %6s %6s %8s %5s %8s %7s %7s" % (((f + ": " + name)[:78],) + tuple(d.get(k, "?") for k in KEYS)))
