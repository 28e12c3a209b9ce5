"""Memory-safety checks under AddressSanitizer + UBSan, CPU build only (the GPU pool has no ASan): the CPU oracle, and
the PRODUCT's host code (swmi_api.cpp, swmi_multi.cpp) on every path of the C ABI that needs no device."""
import os
import shutil
import subprocess

import pytest

from conftest import PKG, ROOT


def test_oracle_under_asan_ubsan(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "oracle_selftest")
    src = [os.path.join(ROOT, "tests", "native", "oracle_selftest.c"), os.path.join(ROOT, "oracle", "sw_oracle.c"),
           os.path.join(ROOT, "oracle", "sg_oracle.c")]
    build = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fopenmp",
                            "-o", exe] + src, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if build.returncode != 0 and "asan" in build.stdout.lower():
        pytest.skip("sanitizer runtime not installed: " + build.stdout[-200:])
    assert build.returncode == 0, build.stdout
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", OMP_NUM_THREADS="2"))
    assert run.returncode == 0, run.stdout
    assert "oracle selftest ok" in run.stdout


def test_product_host_code_under_asan_ubsan(tmp_path):
    """swmi_api.cpp + swmi_multi.cpp compiled by g++ with -fsanitize=address,undefined, linked with the kernels' objects
    as they ship, driven through the C ABI by tests/native/abi_hostpaths.cpp: argument / domain checks, the shard rule,
    init / queue / sharded-batch failure paths, thread-local error text, racing schedule writers."""
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    csrc = os.path.join(PKG, "csrc")
    objs = [os.path.join(PKG, "lib", "sw_kernels.o"), os.path.join(PKG, "lib", "sg_kernels.o")]
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("kernel objects not built (run __graft_entry__.build())")
    san = ["-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
    hip = ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
    lib = str(tmp_path / "libswmi_asan.so")
    build = subprocess.run(["g++"] + san + hip + ["-fPIC", "-shared", "-o", lib, os.path.join(csrc, "swmi_api.cpp"),
                            os.path.join(csrc, "swmi_multi.cpp")] + objs + ["-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-lpthread"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if build.returncode != 0 and "asan" in build.stdout.lower() and "cannot find" in build.stdout.lower():
        pytest.skip("sanitizer runtime not installed: " + build.stdout[-200:])
    assert build.returncode == 0, build.stdout[-3000:]
    exe = str(tmp_path / "abi_hostpaths")
    build = subprocess.run(["g++"] + san + ["-o", exe, os.path.join(ROOT, "tests", "native", "abi_hostpaths.cpp"), lib,
                            "-Wl,-rpath," + str(tmp_path), "-Wl,-rpath,/opt/rocm/lib", "-lpthread"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert build.returncode == 0, build.stdout[-3000:]
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert run.returncode == 0, run.stdout[-3000:]
    assert "abi hostpaths ok" in run.stdout
    # again with a librccl that cannot be loaded: load_rccl()'s failure path (one dlerror() call, a reason string) under ASan
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", SWMI_RCCL_LIB="/nonexistent/librccl-not-here.so"))
    assert run.returncode == 0, run.stdout[-3000:]
    assert "abi hostpaths ok" in run.stdout
