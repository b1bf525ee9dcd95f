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


PARKED = os.path.join(ROOT, "oracle", "_ref", "ngmlr_pool_parked")


def _run(tmp_path, args, contexts, binary=BINARY, fibers=True, carriers=None):
    env = dict(os.environ, CVX_POOL_CONTEXTS=str(contexts), CVX_POOL_FIBERS="1" if fibers else "0")
    if carriers:
        env["CVX_POOL_CARRIERS"] = str(carriers)
    res = subprocess.run([binary, "--skip-write"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=900, cwd=str(tmp_path), env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    return _records(res.stdout), res.stderr


@pytest.mark.skipif(not os.path.exists(BINARY), reason="oracle/_ref/ngmlr_pool_cpu not built (tools/build_ngmlr_hip.sh needs /root/reference)")
@pytest.mark.parametrize("threads,contexts,fibers", [(1, 1, True), (2, 48, False), (8, 256, False), (8, 4096, True)])
def test_pool_keeps_the_reference_sam(tmp_path, threads, contexts, fibers):
    """Contexts as pthreads (CVX_POOL_FIBERS=0, round 4's form) and as user-level contexts on carrier threads (the default)."""
    # short reads (<= 256 bp: processShortRead) and long ones (processLongReadLIS) of test_2, one CS thread
    got, _ = _run(tmp_path, ["-t", "1", "-r", os.path.join(E2E, "ref_chr21_20kb.fa"), "-q", os.path.join(E2E, "reads_100_2200bp.fa")], contexts, fibers=fibers)
    assert sorted(got) == sorted(_records(open(os.path.join(GOLDEN, "test_2.sam")).read())) and len(got) == 12
    # test_3: 142 PacBio reads, 985 convex alignments, split reads, both strands, unmapped reads
    fq = os.path.join(str(tmp_path), "test_3.fq")
    with gzip.open(os.path.join(E2E, "test_3_reads.fq.gz"), "rb") as f, open(fq, "wb") as o:
        o.write(f.read())
    got, err = _run(tmp_path, ["-x", "pacbio", "-t", str(threads), "-R", "0.01", "--no-progress",
                               "-r", os.path.join(E2E, "test_3_reference.fasta.gz"), "-q", fq], contexts, fibers=fibers)
    with gzip.open(os.path.join(GOLDEN, "test_3.sorted.sam.gz"), "rt") as f:
        want = [l.rstrip("\n") for l in f if l.strip() and not l.startswith("@")]
    assert sorted(got) == want and len(want) > 200
    assert "AlignPool: 142 reads on" in err, err[-600:]
    assert ("user-level contexts" in err) == fibers


@pytest.mark.skipif(not os.path.exists(PARKED), reason="oracle/_ref/ngmlr_pool_parked not built (tools/build_ngmlr_hip.sh needs /root/reference)")
@pytest.mark.parametrize("contexts,carriers", [(4096, 8), (3, 2), (64, 1)])
def test_reads_park_and_resume_inside_the_long_read_stage(tmp_path, contexts, carriers):
    """VERDICT r5 item 1 without a GPU: every SingleAlign of ngmlr's own processLongReadLIS gives the read's user-level context back
    to its carrier thread and is resumed later (tests/cpp/parking_cpu_aligner.h: the reference's CPU aligner behind the control
    flow of SharedAligner::SingleAlign) -- 985 park / resume cycles on test_3, in the middle of the interval loop
    (reference src/AlignmentBuffer.cpp:3361-3406) and of the retry loop (:291-425); the SAM must not change."""
    fq = os.path.join(str(tmp_path), "test_3.fq")
    with gzip.open(os.path.join(E2E, "test_3_reads.fq.gz"), "rb") as f, open(fq, "wb") as o:
        o.write(f.read())
    got, err = _run(tmp_path, ["-x", "pacbio", "-t", "4", "-R", "0.01", "--no-progress",
                               "-r", os.path.join(E2E, "test_3_reference.fasta.gz"), "-q", fq], contexts, binary=PARKED, carriers=carriers)
    with gzip.open(os.path.join(GOLDEN, "test_3.sorted.sam.gz"), "rt") as f:
        want = [l.rstrip("\n") for l in f if l.strip() and not l.startswith("@")]
    assert sorted(got) == want
    assert "ParkingCpuAligner: 985 parks, 985 wakes" in err, err[-800:]
    assert "985 parks;" in err and "over %d carrier threads" % carriers in err, err[-800:]


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
    if os.path.exists(PARKED):      # the same reads with a park / resume in front of every alignment of the split-read path
        got, err = _run(tmp_path, args, 2048, binary=PARKED)
        assert sorted(got) == sorted(want)
        assert "ParkingCpuAligner:" in err and " 0 parks" not in err
    assert any(int(l.split("\t")[1]) & 2048 for l in want)          # split reads among them


@pytest.mark.skipif(not os.path.exists(BINARY) or not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ngmlr_ref")),
                    reason="oracle/_ref/ngmlr_pool_cpu / ngmlr_ref not built (tools/build_ngmlr_hip.sh needs /root/reference)")
@pytest.mark.parametrize("split", ["1", "0"])
def test_input_records_read_under_the_lock_objects_built_outside(tmp_path, split):
    """ngmlr_amd/csrc/input_batch_binding.inc: _NGM::GetNextReadBatch (reference src/NGM.cpp:190-244) reads only the raw record
    under NGM's input lock and runs the reference's own NextRead (copyToRead, splitRead) after the unlock.  Same reads, ids and
    SAM as the unmodified reference on FASTQ input with long and short reads -- and on an input whose fourth record is malformed
    (quality string shorter than the sequence): the reference reports the record and terminates the run (Log.Error), and so does the
    binding, from the thread that finishes the record."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_rates
    from ngmlr_amd import synth
    rng = np.random.default_rng(41)
    fa, fq = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq")
    e2e_rates.write_plain_workload(fa, fq, 40, rng, 300_000)
    ref = synth.random_ref(np.random.default_rng(2025), 10)       # (write_plain_workload drew the reference first from its own rng)
    with open(fq, "a") as f:                                       # short reads (processShortRead) behind the long ones
        for i in range(10):
            f.write("@short%d\n%s\n+\n%s\n" % (i, "ACGTTGCATG" * 15, "I" * 150))
    bad = str(tmp_path / "bad.fq")
    recs = open(fq).read().split("\n")
    recs[4 * 3 + 3] = recs[4 * 3 + 3][:-7]                         # fourth record: quality shorter than the sequence
    open(bad, "w").write("\n".join(recs))
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ngmlr_ref")
    for q in (fq, bad):
        args = ["-x", "pacbio", "-t", "4", "-R", "0.01", "--no-progress", "-r", fa, "-q", q]
        res = subprocess.run([ref_bin, "--skip-write"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=str(tmp_path))
        env = dict(os.environ, CVX_POOL_CONTEXTS="64", CVX_INPUT_SPLIT=split)
        got = subprocess.run([BINARY, "--skip-write"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=str(tmp_path), env=env)
        if q == bad:
            msg = "Read r3_%s: Length of read not equal length of quality values." % recs[12].split("_")[1]
            assert res.returncode != 0 and msg in res.stderr and "Terminating" in res.stderr, res.stderr[-1500:]
            assert got.returncode != 0 and msg in got.stderr and "Terminating" in got.stderr, got.stderr[-1500:]
            continue
        assert res.returncode == 0, res.stderr[-2000:]
        assert got.returncode == 0, got.stderr[-2000:]
        want = _records(res.stdout)
        assert sorted(_records(got.stdout)) == sorted(want) and len(want) == 50
        assert "ngmlr's input lock" in got.stderr
