#!/usr/bin/env python
"""Registers / scratch of the kernels of a built library whose mangled name contains a substring (from the code object's notes):
    python tools/kernel_regs.py <lib.so> <substring>"""
import re
import struct
import subprocess
import sys
from pathlib import Path

lib, pat = Path(sys.argv[1]), sys.argv[2]
data = lib.read_bytes()
start = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
(count,) = struct.unpack_from("<Q", data, start + 24)
off = start + 32
co = None
for _ in range(count):
    o, size, length = struct.unpack_from("<QQQ", data, off)
    off += 24
    triple = data[off:off + length].decode()
    off += length
    if "gfx950" in triple:
        co = data[start + o:start + o + size]
tmp = Path("/tmp/_regs.co")
tmp.write_bytes(co)
notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", str(tmp)], capture_output=True, text=True, check=True).stdout
for block in notes.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", block)
    if not name or pat not in name.group(1):
        continue
    def f(k):
        m = re.search(k + r":\s+(\d+)", block)
        return m.group(1) if m else "?"
    print(f"{name.group(1)[:100]:100s} vgpr {f(r'.vgpr_count')} sgpr {f(r'.sgpr_count')} spill {f(r'.vgpr_spill_count')} scratch {f(r'.private_segment_fixed_size')} lds {f(r'.group_segment_fixed_size')}")
