#!/usr/bin/env python
"""Disassemble one kernel of a built library:  python tools/disasm_kernel.py <lib.so> <substring of the mangled name> [out.s]"""
import re
import struct
import subprocess
import sys
from pathlib import Path

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
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
tmp = Path("/tmp/_disasm.co")
tmp.write_bytes(co)
dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(tmp)], capture_output=True, text=True, check=True).stdout
kernels = re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", dis)
hits = [k for k in kernels if pat in k.split("\n", 1)[0]]
text = "\n".join(hits)
if len(sys.argv) > 3:
    Path(sys.argv[3]).write_text(text)
print(f"{len(hits)} kernel(s), {text.count(chr(10))} lines")
