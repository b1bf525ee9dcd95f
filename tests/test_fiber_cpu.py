"""CPU: the user-level context runtime (ngmlr_amd/csrc/cvx_fiber.{h,cpp}) on its own -- reads in flight that
are not OS threads: items run on fibers pinned to a few carrier threads, park while "their launch" is with a dispatcher thread
and are resumed by its wake (the shape of SharedAligner::SingleAlign under Convex::AlignPool; inside ngmlr's own long-read stage:
tests/test_pool_cpu.py).  tests/cpp/fiber_test.cpp checks every item's arithmetic chain, fiber-locals across parks, the stack
below a parked frame, park / wake pairing, slot life cycle and the statistics; a lost wake-up is a hang (timeout)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "ngmlr_amd", "fiber_test")


def _build():
    res = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "ngmlr_amd", "csrc"), os.path.join(ROOT, "ngmlr_amd", "fiber_test")],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout[-2000:]


@pytest.mark.parametrize("args", [
    ["4", "512", "20000", "3"],              # the pool's shape: many more fibers than carriers, a few parks per read
    ["8", "4096", "40000", "2"],             # thousands of reads in flight
    ["1", "8", "4000", "3"],                 # one carrier: every switch on one thread
    ["4", "64", "20000", "4", "immediate"],  # the dispatcher wakes at once: Wake before Park on nearly every request
    ["2", "16", "2000", "0"],                # reads that never park (a read without an alignment)
])
def test_fibers_park_and_resume(args):
    _build()
    res = subprocess.run([BINARY] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.startswith("ok:"), res.stdout[-1000:]


def test_window_note_travels_with_the_read():
    """Convex::DeviceWindows (convex_align_hip.h): the (buffer, position, length) note window_decode_binding.inc leaves for
    ConvexAlignHip::Prepare is fiber-local under the pool's user-level contexts -- it survives parks while other reads note
    their own windows on the same carrier -- and thread-local on plain worker threads; a foreign buffer is never recognised
    (tests/cpp/windows_test.cpp, linked against libcvxalign.so; no device call)."""
    binary = os.path.join(ROOT, "ngmlr_amd", "windows_test")
    res = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "ngmlr_amd", "csrc"), binary], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout[-2000:]
    for args in (["4", "256", "20000"], ["1", "8", "2000"], ["8", "2048", "30000"]):
        res = subprocess.run([binary] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert res.returncode == 0 and res.stdout.startswith("ok:"), res.stdout[-1000:]
