"""The C-ABI driven from plain C (tests/c_abi/abi_demo.c): no Python between the caller and the library.

CPU suite: linked against the emulated build of the kernel sources (logic + linkage of every symbol the demo
uses); `-m gpu`: linked against libplonk_hip.so on the MI355X.  Expected output: the reference's K1 commitment
(test.py:23-28) and a fft(ifft(x)) round trip."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, "c_abi", "abi_demo.c")
PTAU = os.path.join(HERE, "golden", "srs_2048.ptau")
K1 = (16120260411117808045030798560855586501988622612038310041007562782458075125622,
      3125847109934958347271782137825877642397632921923926105820408033549219695465)


def _build_and_run(libdir, libname, tmp_path):
    exe = str(tmp_path / "abi_demo")
    subprocess.run(["gcc", "-O1", "-Wall", "-I", os.path.join(REPO, "include"), SRC, "-o", exe, "-L", libdir, "-l" + libname,
                    "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([exe, PTAU], check=True, capture_output=True, text=True, timeout=600).stdout.split("\n")
    tag, x, y = out[0].split()
    assert tag == "K1" and (int(x, 16), int(y, 16)) == K1
    assert out[1] == "roundtrip ok"


def test_c_caller_against_the_emulated_build(tmp_path):
    emu = os.path.join(HERE, "emu")
    subprocess.run(["make", "-s", "-C", emu, "libplonk_emu.so"], check=True)
    _build_and_run(emu, "plonk_emu", tmp_path)


@pytest.mark.gpu
def test_c_caller_against_the_hip_library(tmp_path):
    libdir = os.path.join(REPO, "plonkathon_amd")
    assert os.path.exists(os.path.join(libdir, "libplonk_hip.so")), "build the library first (python __graft_entry__.py)"
    _build_and_run(libdir, "plonk_hip", tmp_path)
