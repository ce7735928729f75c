"""The C++ host-side mirror of the reference's Row/Bitmap/Container interface
(include/fbk_roaring.hpp) restating row_test.go and executor_test.go vectors; it drives the
GPU through the C ABI, so the run is GPU-marked; the compile check runs everywhere."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_row_api.cpp")
BIN = os.path.join(ROOT, "build", "test_row_api")


def compile_it():
    import __graft_entry__ as g

    g.build()
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    lib = os.path.join(ROOT, "featurebase_amd", "csrc")
    subprocess.check_call(
        ["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), SRC, "-L", lib, "-lfbk", f"-Wl,-rpath,{lib}", "-Wl,-rpath-link,/opt/rocm/lib", "-o", BIN]
    )


def test_cpp_host_mirror_compiles():
    compile_it()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_cpp_host_mirror_row_vectors_on_gpu():
    compile_it()
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "row api ok" in out.stdout
