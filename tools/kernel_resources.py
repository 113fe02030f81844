#!/usr/bin/env python3
"""Per-kernel register / scratch / spill table of a BUILT library (no recompilation): carves the gfx950 code object out of the
.so's clang offload bundle and reads the kernel metadata notes.  usage: python tools/kernel_resources.py [lib.so] [name filter]"""
import re
import struct
import subprocess
import sys
import tempfile

lib = sys.argv[1] if len(sys.argv) > 1 else "mfas_amd/csrc/libmfas_hip.so"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blob = open(lib, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
at = blob.find(magic)
assert at >= 0, "no offload bundle"
n = struct.unpack_from("<Q", blob, at + len(magic))[0]
pos = at + len(magic) + 8
co = None
for _ in range(n):
    off, size, tl = struct.unpack_from("<QQQ", blob, pos)
    triple = blob[pos + 24:pos + 24 + tl].decode()
    pos += 24 + tl
    if "gfx950" in triple:
        co = blob[at + off:at + off + size]
assert co, "no gfx950 code object"
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(co)
    f.flush()
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
names = []
rows = []
for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
    def f(k):
        m = re.search(r"\." + k + r":\s+(\d+)", blk)
        return int(m.group(1)) if m else -1
    nm = re.search(r"\.name:\s+(\S+)", blk).group(1)
    names.append(nm)
    rows.append((f("vgpr_count"), int(re.match(r"\s*(\d+)", blk).group(1)), f("sgpr_count"), f("private_segment_fixed_size"), f("sgpr_spill_count"), f("vgpr_spill_count"),
                 f("group_segment_fixed_size")))
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
print(f"{'kernel':72s} VGPR AGPR SGPR scratch sspill vspill  LDS")
for nm, r in zip(dem, rows):
    nm = re.sub(r"\(.*", "", nm).replace("void ", "")
    if flt in nm:
        print(f"{nm[:72]:72s} {r[0]:4d} {r[1]:4d} {r[2]:4d} {r[3]:7d} {r[4]:6d} {r[5]:6d} {r[6]:5d}")
