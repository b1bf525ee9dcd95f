"""CPU: Convex::AlignPool (ngmlr_amd/csrc/align_pool.{h,cpp}; SURVEY 8 f1, second half) inside the reference's own
binary with the reference's CPU aligners: processLongReadLIS / processShortRead taken off the CS threads
(reference src/ScoreBuffer.cpp:152-159, src/CS.cpp:293-297) and run on many more alignment contexts than `-t`.
The set of SAM records must not change (oracle/_ref/ngmlr_pool_cpu, built by tools/build_ngmlr_hip.sh when
/root/reference is present; the same patch with the device aligners is tests/test_gpu_e2e.py's ngmlr_hip_pool)."""
import gzip
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
E2E = os.path.join(GOLDEN, "e2e")
BINARY = os.path.join(ROOT, "oracle", "_ref", "ngmlr_pool_cpu")


def _records(text):
    return [l.rstrip("\n") for l in text.splitlines() if l.strip() and not l.startswith("@")]


def _run(tmp_path, args, contexts):
    env = dict(os.environ, CVX_POOL_CONTEXTS=str(contexts))
    res = subprocess.run([BINARY, "--skip-write"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=900, cwd=str(tmp_path), env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    return _records(res.stdout), res.stderr


@pytest.mark.skipif(not os.path.exists(BINARY), reason="oracle/_ref/ngmlr_pool_cpu not built (tools/build_ngmlr_hip.sh needs /root/reference)")
@pytest.mark.parametrize("threads,contexts", [(1, 1), (2, 48), (8, 256)])
def test_pool_keeps_the_reference_sam(tmp_path, threads, contexts):
    # short reads (<= 256 bp: processShortRead) and long ones (processLongReadLIS) of test_2, one CS thread
    got, _ = _run(tmp_path, ["-t", "1", "-r", os.path.join(E2E, "ref_chr21_20kb.fa"), "-q", os.path.join(E2E, "reads_100_2200bp.fa")], contexts)
    assert sorted(got) == sorted(_records(open(os.path.join(GOLDEN, "test_2.sam")).read())) and len(got) == 12
    # test_3: 142 PacBio reads, 985 convex alignments, split reads, both strands, unmapped reads
    fq = os.path.join(str(tmp_path), "test_3.fq")
    with gzip.open(os.path.join(E2E, "test_3_reads.fq.gz"), "rb") as f, open(fq, "wb") as o:
        o.write(f.read())
    got, err = _run(tmp_path, ["-x", "pacbio", "-t", str(threads), "-R", "0.01", "--no-progress",
                               "-r", os.path.join(E2E, "test_3_reference.fasta.gz"), "-q", fq], contexts)
    with gzip.open(os.path.join(GOLDEN, "test_3.sorted.sam.gz"), "rt") as f:
        want = [l.rstrip("\n") for l in f if l.strip() and not l.startswith("@")]
    assert sorted(got) == want and len(want) > 200
    assert "AlignPool: 142 reads on" in err, err[-600:]


@pytest.mark.skipif(not os.path.exists(BINARY) or not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ngmlr_ref")),
                    reason="oracle/_ref/ngmlr_pool_cpu / ngmlr_ref not built (tools/build_ngmlr_hip.sh needs /root/reference)")
def test_pool_keeps_the_sam_of_split_reads(tmp_path):
    """ONT-like reads with inversions / deletions / insertions (tools/e2e_rates.py write_sv_workload, -x ont): ngmlr's split-read
    path -- several intervals and alignments per read, supplementary records -- through the alignment contexts against the
    unmodified reference, both on the CPU aligners."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_rates
    fa, fq = str(tmp_path / "sv_ref.fa"), str(tmp_path / "sv_reads.fq")
    e2e_rates.write_sv_workload(fa, fq, 60, seed=78, L=1_000_000)
    args = ["-x", "ont", "-t", "8", "-R", "0.01", "--no-progress", "-r", fa, "-q", fq]
    res = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ngmlr_ref"), "--skip-write"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=900, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-2000:]
    want = _records(res.stdout)
    got, err = _run(tmp_path, args, 96)
    assert sorted(got) == sorted(want) and len(want) >= 60
    assert any(int(l.split("\t")[1]) & 2048 for l in want)          # split reads among them
