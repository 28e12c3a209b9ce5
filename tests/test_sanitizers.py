"""Memory-safety check of the CPU oracle under AddressSanitizer + UBSan (CPU build only; the GPU pool has no ASan)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


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
