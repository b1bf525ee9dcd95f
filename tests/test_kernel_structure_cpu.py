"""Guards on the compiled shape of the gfx950 kernels (no GPU needed: hipcc cross-compiles).

hipcc (ROCm 7.2) turns the persistent tile loop of fill_ring_kernel into a divergent loop that
never terminates as soon as its epilogue grows certain constructs (value-returning atomics, extra
branches, calls).  The good form has exactly two loop levels (tile loop, step-group loop); the
bad one has three.  A hang on the GPU box is expensive, so the shape is pinned here.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ngmlr_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm") / "cvx_kernels.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-fPIC",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only",
           os.path.join(CSRC, "cvx_kernels.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out.read_text()


def _kernels(txt, prefix):
    for m in re.finditer(r"^(%s\w*):" % prefix, txt, re.M):
        body = txt[m.start():]
        yield m.group(1), body[:body.index(".end_amdhsa_kernel")]


def test_fill_kernels_keep_two_loop_levels(device_asm):
    seen = 0
    for name, body in _kernels(device_asm, "_ZN3cvx16fill_ring_kernel"):
        depths = re.findall(r"Loop Header: Depth=(\d+)", body)
        assert depths == ["1", "2"], "%s: loop levels %s (divergent tile loop?)" % (name, depths)
        seen += 1
    assert seen == 22          # 7 single-wave + 4 multi-wave classes, each with and without int16 wrap


def test_register_budgets(device_asm):
    def vgprs(body):
        return int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
    fill = dict(_kernels(device_asm, "_ZN3cvx16fill_ring_kernelILi3ELi1ELb0E"))
    assert len(fill) == 1
    assert vgprs(next(iter(fill.values()))) <= 80       # six waves per SIMD
    walk = dict(_kernels(device_asm, "_ZN3cvx16backtrack_kernel"))
    assert len(walk) == 1
    assert vgprs(next(iter(walk.values()))) <= 32
