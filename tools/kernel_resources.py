#!/usr/bin/env python3
"""Diagnostic: registers / spills / scratch of every kernel instance in the built library (from the code object's notes).
   python tools/kernel_resources.py [substring]      (LIB=path: another build)"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("LIB") or os.path.join(ROOT, "barbay.jl_amd", "lib", "libbarbay_hip.so")
data = open(lib, "rb").read()
notes = ""
pos = 0
while True:                     # one offload bundle per translation unit of the library
    o = data.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
    if o < 0:
        break
    pos = o + 24
    n = struct.unpack_from("<Q", data, o + 24)[0]
    p = o + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, p)
        p += 24
        trip = data[p:p + tl].decode()
        p += tl
        if "gfx950" in trip and size:
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(data[o + off:o + off + size])
                f.flush()
                notes += subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
want = sys.argv[1] if len(sys.argv) > 1 else ""
for e in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", e)
    if not g("name"):
        continue
    nm = subprocess.run(["c++filt", g("name").group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
    if want in nm:
        print(f"{nm:60s} vgpr {int(g('vgpr_count').group(1)):4d}  spilled {int(g('vgpr_spill_count').group(1)):4d}  sgpr {int(g('sgpr_count').group(1)):4d}"
              f"  scratch {int(g('private_segment_fixed_size').group(1)):5d} B")
