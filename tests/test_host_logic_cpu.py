"""CPU: the device-independent host logic of the runtime (upload layout/packing, kernel-class choice,
arena offsets, LPT work lists) through tests/cpp/host_logic_test.cpp, built with plain g++."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = tmp_path / "host_logic_test"
    subprocess.run([gxx, "-O1", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "ngmlr_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "host_logic_test.cpp"), "-o", str(exe)],
                   check=True, capture_output=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_logic_test: ok" in r.stdout
