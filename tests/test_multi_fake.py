"""The multi-GPU host logic (swmi_multi.cpp, swmi_api.cpp) on three FAKE GPUs and a fake RCCL, no device needed: the
branches a one-GPU box cannot reach -- peer copies between distinct devices, the three-rank RCCL all-gather, the grouped
broadcasts of ragged shards, empty shards, an ncclCommInitAll failure -- run through the real host code, which is compiled
with g++ against tests/native/fake_hip.cpp instead of the HIP runtime and the kernels.  The fake RCCL checks what a grouped
RCCL sequence must satisfy (the same collectives, roots and counts in the same order on every rank)."""
import os
import shutil
import subprocess

import pytest

from conftest import PKG, ROOT


@pytest.fixture(scope="module")
def fake_build(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    tmp = tmp_path_factory.mktemp("multi_fake")
    csrc = os.path.join(PKG, "csrc")
    native = os.path.join(ROOT, "tests", "native")
    flags = ["-O1", "-g", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-fsanitize=address,undefined",
             "-fno-sanitize-recover=all"]
    rccl = str(tmp / "libfake_rccl.so")
    b = subprocess.run(["g++"] + flags + ["-fPIC", "-shared", "-o", rccl, os.path.join(native, "fake_rccl.cpp")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if b.returncode != 0 and "asan" in b.stdout.lower() and "cannot find" in b.stdout.lower():
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stdout[-3000:]
    exe = str(tmp / "multi_fake")
    b = subprocess.run(["g++"] + flags + ["-o", exe, os.path.join(native, "multi_fake.cpp"), os.path.join(native, "fake_hip.cpp"),
                                          os.path.join(csrc, "swmi_api.cpp"), os.path.join(csrc, "swmi_multi.cpp"), "-ldl", "-lpthread"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert b.returncode == 0, b.stdout[-3000:]
    return exe, rccl


def _run(exe, rccl, **env):
    clean = {k: v for k, v in os.environ.items() if not k.startswith("SWMI_")}
    return subprocess.run([exe, rccl], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          env=dict(clean, ASAN_OPTIONS="detect_leaks=1", FAKE_HIP_DEVICES="3", **env))


def test_three_fake_gpus_peer_copies_and_rccl_gather(fake_build):
    run = _run(*fake_build)
    assert run.returncode == 0, run.stdout[-3000:]
    assert "multi fake ok" in run.stdout


def test_three_fake_gpus_when_the_rccl_clique_cannot_be_created(fake_build):
    run = _run(*fake_build, FAKE_NCCL_FAIL_INIT="1")
    assert run.returncode == 0, run.stdout[-3000:]
    assert "multi fake ok" in run.stdout
