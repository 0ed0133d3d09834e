"""The C-ABI library: it loads on a CPU-only box, exports every symbol that
include/feinsum_hip.h declares, and its argument checks work without a GPU."""

import ctypes
import re
from pathlib import Path

import pytest

from feinsum_amd import _hip
from feinsum_amd.diagnostics import HipLibraryError, InvalidParameterError

ROOT = Path(__file__).resolve().parents[1]


def _declared_symbols():
    text = (ROOT / "include" / "feinsum_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _hip.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 14
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/feinsum_hip.h but not exported"
    assert set(declared) == set(_hip.EXPORTED_SYMBOLS)


def test_version_and_flop_counters():
    lib = _hip.load_library()
    assert lib.fe_version() == 1000
    assert _hip.flops_per_element(1, 35) == 7980            # grad
    assert _hip.flops_per_element(2, 35) == 7980            # div
    assert _hip.flops_per_element(3, 35) == 15960           # fused grad + div
    assert _hip.flops_per_element(4, 35, 4, 15, 4) == 17040  # face-mass x4
    assert _hip.flops_per_element(5, 35) == 7455             # div component (x3 = 22365)
    assert _hip.flops_per_element(6, 35, b=12) == 89460      # cross-product batch: 12 planes
    assert _hip.flops_per_element(7, 35, b=4) == 9940        # e,ij,ej->ei x4
    assert _hip.flops_per_element(99, 35) == -1


def test_argument_validation_without_gpu():
    # FE_EINVAL -> InvalidParameterError, FE_EUNSUPPORTED -> NotImplementedError; none of
    # these reach the HIP runtime
    with pytest.raises(InvalidParameterError, match="E must be"):
        _hip.grad3d(0, 0, 0, 0, -1, 35)
    with pytest.raises(InvalidParameterError, match="Np must be"):
        _hip.div3d(0, 0, 0, 0, 10, 0)
    with pytest.raises(InvalidParameterError, match="null"):
        _hip.grad3d(0, 0, 0, 0, 10, 35)
    with pytest.raises(NotImplementedError, match="unknown variant"):
        _hip.grad3d(8, 8, 8, 8, 10, 35, variant=7)
    with pytest.raises(InvalidParameterError, match="8-byte aligned"):
        _hip.div3d(8, 8, 12, 8, 10, 35)
    with pytest.raises(InvalidParameterError, match="operator flags"):
        _hip.grad3d(8, 8, 8, 8, 10, 35, op_flags=4)
    with pytest.raises(InvalidParameterError, match="operator flags"):
        _hip.divcomp3d(8, 8, 8, 8, 10, 35, op_flags=8)
    with pytest.raises(InvalidParameterError, match="unknown kernel variant"):
        _hip.grad3d(8, 8, 8, 8, 10, 35, variant="fastest")
    with pytest.raises(InvalidParameterError, match="layout"):
        _hip.facemass(8, 8, [8], [8], 10, 35, 4, 15, layout_flags=8)
    with pytest.raises(InvalidParameterError, match="as many outputs"):
        _hip.facemass(8, 8, [8, 8], [8], 10, 35, 4, 15)
    # E == 0 is a valid no-op for every family (no launch, no HIP call)
    _hip.grad3d(0, 0, 0, 0, 0, 35)
    _hip.div3d(0, 0, 0, 0, 0, 35)
    _hip.facemass(0, 0, [0, 0], [0, 0], 0, 35, 4, 15)
    assert _hip.load_library().fe_last_error() is not None


def test_prepared_operator_argument_checks_without_gpu():
    with pytest.raises(InvalidParameterError, match="null"):
        _hip.prepare_operator(1, 0, 35, 0, 0, 0, 0)
    with pytest.raises(InvalidParameterError, match="aligned"):
        _hip.prepare_operator(1, 8, 35, 0, 0, 0, 24)
    with pytest.raises(NotImplementedError, match="p = 1..4"):
        _hip.prepare_operator(1, 8, 56, 0, 0, 0, 64)
    with pytest.raises(NotImplementedError, match="no prepared form"):
        _hip.prepare_operator(5, 8, 35, 0, 0, 0, 64)
    with pytest.raises(InvalidParameterError, match="flags"):
        _hip.prepare_operator(4, 8, 35, 4, 15, 1, 64)          # FE_FM_J_FE is not an operator flag
    # a buffer this process never prepared is refused before anything is launched
    lib = _hip.load_library()
    ptrs = _hip._ptr_array([8])
    rc = lib.fe_grad3d_prepared_f64(8, 8, 4096, ptrs, ptrs, 10, 35, 1, 0, 0, 0)
    assert rc == _hip.FE_EINVAL and b"fe_prepare_operator" in lib.fe_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setenv("FEINSUM_HIP_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_hip, "_lib", None)
    with pytest.raises(HipLibraryError, match="not found"):
        _hip.load_library()
    monkeypatch.undo()
    assert isinstance(_hip.load_library(), ctypes.CDLL)


def test_product_package_does_not_import_the_oracle():
    for py in (ROOT / "feinsum_amd").rglob("*.py"):
        src = py.read_text()
        assert "import oracle" not in src and "from oracle" not in src, py


def test_one_hip_runtime_whatever_the_import_order():
    """torch's wheel brings its own libamdhip64.so and asks for it by file name; this library asks for libamdhip64.so.7.  Loaded in
    the wrong order the process held TWO HIP runtimes and every launch here failed with "no ROCm-capable device is detected"
    (seen with a tool that loaded the library before anything imported torch).  load_library() imports torch first."""
    import subprocess
    import sys

    code = ("from feinsum_amd import _hip\n_hip.load_library()\nimport torch\n"
            "print(len({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(Path(__file__).resolve().parents[1]),
                         timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().splitlines()[-1] == "1", out.stdout
