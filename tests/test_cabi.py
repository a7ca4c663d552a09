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


def test_experiment_switches_cannot_reach_a_release_build():
    """The A/B and sensitivity macros of the kernels (some compile arithmetic out: wrong results on purpose) need
    -DCLAPGPU_EXPERIMENT, and an experiment build reports an ABI version with the top bit set, which _lib refuses."""
    import subprocess
    src = os.path.join(ROOT, "clap_amd", "csrc", "runtime.hip")
    base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-E", src,
            "--cuda-host-only", "-o", "-"]
    for macro in ("CLAPGPU_EXP_NO_INVERT", "CLAPGPU_EXP_NO_AABB", "CLAPGPU_PLAIN_STORES", "BP_SEARCH_IN_FLIGHT=2"):
        p = subprocess.run(base + ["-D" + macro], capture_output=True, text=True)
        assert p.returncode != 0 and "experiment switches" in p.stderr, macro
    p = subprocess.run(base + ["-DCLAPGPU_EXPERIMENT", "-DCLAPGPU_EXP_NO_AABB"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-500:]
    line = [l for l in p.stdout.splitlines() if "clapgpu_abi_version(void)" in l and "return" in l][-1]
    assert "0x80000000u" in line
    p = subprocess.run(base, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if "clapgpu_abi_version(void)" in l and "return" in l][-1]
    assert p.returncode == 0 and "0x80000000u" not in line
