"""GPU (-m gpu): configs[0] "plumbing", end to end.  The reference's own ngmlr binary, built
from its sources with the ONE change INTEGRATION.md describes (ConvexAlignHip constructed
instead of ConvexAlignFast; tools/build_ngmlr_hip.sh), maps the reference's own test reads
with every convex alignment running on the MI355X; the SAM records must be identical to
those of the unmodified reference (tests/golden/test_*.sam, written by tools/make_golden.sh)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "ngmlr_hip")
E2E = os.path.join(ROOT, "tests", "golden", "e2e")


def _records(text):
    return [l for l in text.splitlines() if l and not l.startswith("@")]


def _run(args, tmp_path):
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/ngmlr_hip not built (tools/build_ngmlr_hip.sh needs /root/reference)")
    res = subprocess.run([BIN, "--skip-write"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-3000:]
    maps = open("/proc/self/maps").read()  # noqa: F841  (the child loaded libcvxalign.so via its RUNPATH)
    return _records(res.stdout), res.stderr


def test_test_2_sam_identical_to_reference(built, tmp_path):
    got, err = _run(["-t", "1", "-r", os.path.join(E2E, "ref_chr21_20kb.fa"),
                     "-q", os.path.join(E2E, "reads_100_2200bp.fa")], tmp_path)
    want = _records(open(os.path.join(ROOT, "tests", "golden", "test_2.sam")).read())
    assert len(want) == 12
    assert got == want


def test_test_4_sam_identical_to_reference(built, tmp_path):
    got, err = _run(["-x", "pacbio", "-t", "1", "-r", os.path.join(E2E, "test_4_reference.fasta.gz"),
                     "-q", os.path.join(E2E, "test_4_read.fa.gz")], tmp_path)
    want = _records(open(os.path.join(ROOT, "tests", "golden", "test_4.sam")).read())
    assert len(want) == 1
    assert got == want


def test_binary_links_the_device_library(built):
    if not os.path.exists(BIN):
        pytest.skip("ngmlr_hip not built")
    out = subprocess.run(["ldd", BIN], stdout=subprocess.PIPE, text=True).stdout
    assert "libcvxalign.so" in out and "not found" not in out.split("libcvxalign.so")[1].splitlines()[0]
    syms = subprocess.run(["nm", "-C", BIN], stdout=subprocess.PIPE, text=True).stdout
    assert "Convex::ConvexAlignHip::SingleAlign" in syms
