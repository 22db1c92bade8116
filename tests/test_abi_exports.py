"""CPU suite: the C-ABI shared library loads and exports every symbol include/plonk_hip.h declares
(no compute calls — there is no GPU in the build container)."""
import ctypes
import os
import re

from conftest import REPO


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "plonk_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(plonk_[a-z0-9_]+)\s*\(", text)))


def test_header_and_ctypes_table_agree():
    from plonkathon_amd import _lib

    assert _declared_symbols() == sorted(_lib.SIGNATURES)


def test_hip_library_exports_the_abi():
    from plonkathon_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):  # a fresh checkout: the library is a build product (hipcc cross-compiles)
        import shutil
        import subprocess

        import pytest

        if not shutil.which("hipcc"):
            pytest.skip("libplonk_hip.so is not built and hipcc is not on PATH")
        subprocess.run(["make", "-s", "-C", os.path.join(REPO, "plonkathon_amd", "csrc")], check=True)
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(cdll, name), name
    cdll.plonk_abi_version.restype = ctypes.c_int
    assert cdll.plonk_abi_version() == 2


def test_product_has_no_cpu_fallback():
    """Nothing under plonkathon_amd/ imports the oracle or the emulation build."""
    pkg = os.path.join(REPO, "plonkathon_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libplonk_emu" not in src, f
