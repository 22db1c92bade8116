import ctypes
import os
import subprocess
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")
EMU_DIR = os.path.join(REPO, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libplonk_emu.so")


# CPU runs go through the fiber emulation, where building even the default 4 GiB lookup table would take
# minutes: without a GPU the automatic policy gets a budget nothing fits in (bucket method); the lookup path is
# covered by the forced-size tests (mode 2).  GPU runs keep the library default.
try:
    import subprocess as _sp

    _HAS_GPU = _sp.run(["rocminfo"], capture_output=True, timeout=30).returncode == 0 and os.path.exists("/dev/kfd")
except Exception:  # pragma: no cover
    _HAS_GPU = False
if not _HAS_GPU:
    os.environ.setdefault("PLONK_MSM_TABLE_GB", "0.0001")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def emu_cdll():
    """Host build of the kernel sources against tests/emu/hip_emu.h (test infrastructure only)."""
    subprocess.run(["make", "-s", "-C", EMU_DIR, "-j8"], check=True)
    return ctypes.CDLL(EMU_LIB)


@pytest.fixture(autouse=True)
def _backend_binding(request):
    """GPU-marked tests talk to the real libplonk_hip.so; everything else that touches
    plonkathon_amd is bound to the CPU emulation build, injected here from the test side (the
    package itself has no switch for it)."""
    from plonkathon_amd import _lib, backend

    want_gpu = request.node.get_closest_marker("gpu") is not None
    uses_emu = "emu" in request.fixturenames or "emu_cdll" in request.fixturenames
    if want_gpu:
        if getattr(_lib, "_bound_kind", None) != "hip":
            backend.set_context(None)
            _lib._lib = None
            _lib.lib()
            _lib._bound_kind = "hip"
    elif uses_emu:
        if getattr(_lib, "_bound_kind", None) != "emu":
            backend.set_context(None)
            _lib.bind(request.getfixturevalue("emu_cdll"))
            _lib._bound_kind = "emu"
    yield


@pytest.fixture
def emu(emu_cdll):
    """Marker fixture: the test runs the product's Python layer over the emulated kernels."""
    return emu_cdll
