"""CPU: libclapgpu.so loads and exports exactly what include/clapgpu.h declares.
No compute call is made (there is no GPU here)."""
import os
import re
import subprocess

import pytest

from clap_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "clapgpu.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(clapgpu_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_header_functions_all_bound_and_exported(built_lib):
    declared = declared_functions()
    assert len(declared) >= 15
    assert sorted(_lib.SYMBOLS) == declared, "clap_amd/_lib.py SYMBOLS must mirror include/clapgpu.h"
    for name in declared:
        assert hasattr(built_lib, name), f"libclapgpu.so does not export {name}"


def test_exports_are_c_abi(built_lib):
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True)
    exported = {l.split()[-1] for l in out.stdout.splitlines() if " T " in l}
    for name in declared_functions():
        assert name in exported, f"{name} is not an unmangled exported symbol"


def test_abi_version(built_lib):
    assert built_lib.clapgpu_abi_version() == _lib.ABI_VERSION


def test_code_object_is_gfx950_only():
    """The fat binary's bundle ids name the ISAs it carries: gfx950 and nothing else."""
    blob = open(_lib.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_error_codes_match_reference_cerr_enum():
    # core/error.h:12-49 of the reference
    assert (_lib.ERR_NOMEM, _lib.ERR_INVALID_ARGUMENTS, _lib.ERR_NOT_SUPPORTED) == (-1, -2, -3)
    assert (_lib.ERR_TOO_LARGE, _lib.ERR_INIT_FAILED, _lib.ERR_OUT_OF_BOUNDS, _lib.ERR_UNKNOWN) == (-11, -14, -26, -32)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.ClapGpuError):
        _lib.lib()
