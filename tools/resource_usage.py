#!/usr/bin/env python3
"""Per-kernel register / scratch / spill table from `hipcc -Rpass-analysis=kernel-resource-usage` remarks.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2> remarks.txt ; python tools/resource_usage.py remarks.txt"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
try:
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
except FileNotFoundError:
    dem = names
print(f"{'kernel':64s} VGPR AGPR SGPR scratch sspill vspill occ   LDS")
for b, nm in zip(blocks, dem):
    def f(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    nm = re.sub(r"\(.*", "", nm).replace("void ", "")
    print(f"{nm[:64]:64s} {f('VGPRs'):4d} {f('AGPRs'):4d} {f('SGPRs'):4d} {f('ScratchSize .bytes/lane.'):7d} "
          f"{f('SGPRs Spill'):6d} {f('VGPRs Spill'):6d} {f('Occupancy .waves/SIMD.'):3d} {f('LDS Size .bytes/block.'):5d}")
