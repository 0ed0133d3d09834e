"""The dynamic walk's tickets (feinsum_amd/csrc/fe_common.h) come back from the memory side microseconds after the atomic was
issued, into VGPRs v255 / v254 that only the inline-asm statements name.  That is safe as long as the compiler keeps nothing of
its own there -- which holds because those kernels need fewer registers, not because anything forces it.  This test makes it a
checked property of the BUILT library: it disassembles the gfx950 code object and looks at every kernel that takes tickets."""
import re
import shutil
import struct
import subprocess
from pathlib import Path

import pytest

from feinsum_amd import _hip

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _gfx950_code_object(lib: Path) -> bytes:
    data = lib.read_bytes()
    start = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert start >= 0, "no offload bundle in the library"
    (count,) = struct.unpack_from("<Q", data, start + 24)
    off = start + 32
    for _ in range(count):
        o, size, length = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + length].decode()
        off += length
        if "gfx950" in triple:
            return data[start + o:start + o + size]
    raise AssertionError("no gfx950 code object in the library")


@pytest.mark.skipif(shutil.which(OBJDUMP) is None, reason="llvm-objdump of the ROCm toolchain not found")
def test_ticket_registers_are_not_used_by_the_compiler(tmp_path):
    lib = _hip.library_path()
    if not lib.exists():
        pytest.skip("library not built")
    co = tmp_path / "gfx950.co"
    co.write_bytes(_gfx950_code_object(lib))
    dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True, check=True).stdout
    kernels = re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", dis)
    takers = [k for k in kernels if "global_atomic_add v255" in k]
    assert len(takers) >= 5, "expected the grad / div / face-mass / fused kernels with a dynamic walk"
    for k in takers:
        name = k.split("\n", 1)[0]
        for line in k.splitlines()[1:]:
            text = line.split("//")[0]
            if "global_atomic_add v25" in text or ("v_readfirstlane_b32" in text and re.search(r"\bv25[45]\b", text)):
                continue
            assert not re.search(r"\bv25[45]\b", text), (name, line)
            for m in re.finditer(r"v\[(\d+):(\d+)\]", text):
                assert int(m.group(2)) < 254, (name, line)


@pytest.mark.skipif(shutil.which(OBJDUMP) is None, reason="llvm-objdump of the ROCm toolchain not found")
def test_the_asm_stores_carry_their_wait_states(tmp_path):
    """The write-through stores are inline asm (fe_common.h: FE_STORE16_WRITE_THROUGH), so hipcc's hazard recognizer does not know
    that they are vector-memory stores of 16 bytes: whatever it schedules next may write the store's data registers while the
    hardware still reads them (two wait states on gfx9), and the store's SGPR base comes from a VALU instruction (five wait states
    in front).  Round 5 lost the low dword of four doubles of a tile to the first hazard, in some launches, after an unrelated
    change had moved the code behind a tile's last store.  Checked on the BUILT library: behind every such store comes its
    ``s_nop 1``, and the first store of a batch follows an ``s_nop 4``."""
    lib = _hip.library_path()
    if not lib.exists():
        pytest.skip("library not built")
    co = tmp_path / "gfx950.co"
    co.write_bytes(_gfx950_code_object(lib))
    dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True, check=True).stdout
    lines = [ln.split("//")[0].strip() for ln in dis.splitlines()]
    stores = [k for k, ln in enumerate(lines) if ln.startswith("global_store_dwordx4") and "sc0 sc1" in ln and " nt" not in ln]
    assert len(stores) >= 30, "expected the grad kernels' write-through stores"
    firsts = 0
    for k in stores:
        assert lines[k + 1].startswith("s_nop 1"), (lines[k], lines[k + 1])
        if lines[k - 1].startswith("s_nop 4"):
            firsts += 1
    assert firsts * 5 >= len(stores), (firsts, len(stores))          # batches of at most five stores (one plane of a tile)
