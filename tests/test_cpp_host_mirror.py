"""The C++ host-side mirror of the reference's Row/Bitmap/Container interface
(include/fbk_roaring.hpp) restating row_test.go and executor_test.go vectors; it drives the
GPU through the C ABI, so the run is GPU-marked; the compile check runs everywhere."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_row_api.cpp")
BIN = os.path.join(ROOT, "build", "test_row_api")
SRC_EXEC = os.path.join(ROOT, "tests", "cpp", "test_executor_api.cpp")
BIN_EXEC = os.path.join(ROOT, "build", "test_executor_api")
SRC_THR = os.path.join(ROOT, "tests", "cpp", "test_threads.cpp")
BIN_THR = os.path.join(ROOT, "build", "test_threads")
SRC_FORK = os.path.join(ROOT, "tests", "cpp", "test_forks.cpp")
BIN_FORK = os.path.join(ROOT, "build", "test_forks")


def compile_it(src=SRC, out=BIN):
    import __graft_entry__ as g

    g.build()
    os.makedirs(os.path.dirname(out), exist_ok=True)
    lib = os.path.join(ROOT, "featurebase_amd", "csrc")
    subprocess.check_call(
        ["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), src, "-L", lib, "-lfbk", f"-Wl,-rpath,{lib}", "-Wl,-rpath-link,/opt/rocm/lib", "-o", out]
    )


def test_cpp_host_mirror_compiles():
    compile_it()
    assert os.path.exists(BIN)
    compile_it(SRC_EXEC, BIN_EXEC)
    assert os.path.exists(BIN_EXEC)
    compile_it(SRC_THR, BIN_THR)
    assert os.path.exists(BIN_THR)
    compile_it(SRC_FORK, BIN_FORK)
    assert os.path.exists(BIN_FORK)


@pytest.mark.gpu
def test_one_context_many_threads_on_gpu():
    """8 OS threads x 25 rounds of Row operations on ONE context (the cgo calling pattern)."""
    compile_it(SRC_THR, BIN_THR)
    out = subprocess.run([BIN_THR], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "threads ok" in out.stdout


@pytest.mark.gpu
def test_forked_contexts_overlap_on_gpu():
    """8 threads x 8 forked contexts finish the same queries in < 0.5 x the serial wall time;
    error messages are per context (tests/cpp/test_forks.cpp)."""
    compile_it(SRC_FORK, BIN_FORK)
    out = subprocess.run([BIN_FORK], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "forks ok" in out.stdout


@pytest.mark.gpu
def test_cpp_host_mirror_row_vectors_on_gpu():
    compile_it()
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "row api ok" in out.stdout


@pytest.mark.gpu
def test_cpp_executor_mirror_vectors_on_gpu():
    """executor_test.go vectors (set-ops, TopK, Sum, Min/Max, ranges, GroupBy) through
    include/fbk_executor.hpp, every operator on the GPU."""
    compile_it(SRC_EXEC, BIN_EXEC)
    out = subprocess.run([BIN_EXEC], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "executor api ok" in out.stdout
