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
BIN_BATCHED = os.path.join(ROOT, "oracle", "_ref", "ngmlr_hip_batched")
BIN_FULL = os.path.join(ROOT, "oracle", "_ref", "ngmlr_hip_full")
BIN_POOL = os.path.join(ROOT, "oracle", "_ref", "ngmlr_hip_pool")
BIN_ALL = os.path.join(ROOT, "oracle", "_ref", "ngmlr_hip_all")
E2E = os.path.join(ROOT, "tests", "golden", "e2e")


def _records(text):
    return [l for l in text.splitlines() if l and not l.startswith("@")]


def _run(args, tmp_path, binary=BIN, env=None):
    if not os.path.exists(binary):
        pytest.skip("%s not built (tools/build_ngmlr_hip.sh needs /root/reference)" % os.path.relpath(binary, ROOT))
    res = subprocess.run([binary, "--skip-write"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600, cwd=str(tmp_path), env=dict(os.environ, **(env or {})))
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


def _test_3_args(tmp_path, threads):
    import gzip
    fq = os.path.join(str(tmp_path), "test_3.fq")          # FASTQ: FASTA + a reverse-strand hit crashes the reference
    with gzip.open(os.path.join(E2E, "test_3_reads.fq.gz"), "rb") as f, open(fq, "wb") as o:
        o.write(f.read())
    return ["-x", "pacbio", "-t", str(threads), "-R", "0.01", "--no-progress",
            "-r", os.path.join(E2E, "test_3_reference.fasta.gz"), "-q", fq]


def _test_3_want():
    import gzip
    with gzip.open(os.path.join(ROOT, "tests", "golden", "test_3.sorted.sam.gz"), "rt") as f:
        return [l.rstrip("\n") for l in f if l.strip()]


def test_test_3_sam_identical_to_reference(built, tmp_path):
    """test/test_3.sh's input (142 PacBio reads, 985 convex alignments), one private aligner per worker."""
    got, err = _run(_test_3_args(tmp_path, 1), tmp_path)
    want = _test_3_want()
    assert len(want) == 202
    assert sorted(got) == want


@pytest.mark.parametrize("threads", [16, 64])
def test_test_3_batched_pipeline_sam_identical(built, tmp_path, threads):
    """SURVEY 8 f1 inside the real pipeline: ngmlr -t N with every worker's AlignmentBuffer sharing ONE
    BatchingAligner (Convex::SharedAligner constructed at src/AlignmentBuffer.h:355): many tiles per
    device launch, SAM records identical to the unmodified reference (sorted: order is thread-dependent)."""
    import re
    got, err = _run(_test_3_args(tmp_path, threads), tmp_path, binary=BIN_BATCHED)
    assert sorted(got) == _test_3_want()
    m = re.search(r"SharedAligner: (\d+) alignments in (\d+) device launches", err)
    assert m, err[-2000:]
    requests, launches = int(m.group(1)), int(m.group(2))
    assert requests == 985
    assert launches < requests          # batching happened
    print("test_3 -t %d: %d alignments in %d launches (%.1f per launch)" % (threads, requests, launches, requests / launches))


def test_test_3_both_plugins_on_the_device(built, tmp_path):
    """Alignment (Convex::SharedAligner at src/AlignmentBuffer.h:355) AND sub-read scoring (StrippedSWHip at
    NGM::CreateAlignment, src/NGM.cpp:355, i.e. ScoreBuffer's BatchScore of 1024-pair batches) on the MI355X: the
    reference's pipeline around them unchanged, SAM records identical to the unmodified reference."""
    import re
    got, err = _run(_test_3_args(tmp_path, 16), tmp_path, binary=BIN_FULL)
    assert sorted(got) == _test_3_want()
    m = re.search(r"SharedAligner: (\d+) alignments in (\d+) device launches", err)
    assert m and int(m.group(1)) == 985, err[-2000:]
    syms = subprocess.run(["nm", "-C", BIN_FULL], stdout=subprocess.PIPE, text=True).stdout
    assert "StrippedSWHip::BatchScore" in syms


@pytest.mark.parametrize("target,fibers", [(0, "1"), (64, "1"), (64, "0")])
def test_test_3_alignment_contexts_off_the_cs_threads(built, tmp_path, target, fibers):
    """SURVEY 8 f1, second half (Convex::AlignPool, ngmlr_amd/csrc/align_pool.h): `-t 16` CS threads, 256 alignment
    contexts -- processLongReadLIS runs on a pool context instead of on the CS thread that scored the read's last
    sub-read (reference src/ScoreBuffer.cpp:152-159), so reads in flight are no longer bounded by -t.  Alignment,
    scoring and SAM records on the drop-ins; with and without a batch target for the dispatcher.  SAM records
    identical to the unmodified reference (sorted)."""
    import re
    # contexts as user-level contexts on carrier threads (round 6, cvx_fiber.h: the read parks inside SharedAligner::SingleAlign by
    # switching back to its carrier) and as pthreads (round 4's form, CVX_POOL_FIBERS=0)
    env = {"CVX_POOL_CONTEXTS": "256", "CVX_BATCH_TARGET": str(target), "CVX_BATCH_HOLD_US": "20000", "CVX_POOL_FIBERS": fibers}      # (target 0 = none)
    got, err = _run(_test_3_args(tmp_path, 16), tmp_path, binary=BIN_POOL, env=env)
    assert sorted(got) == _test_3_want()
    m = re.search(r"SharedAligner: (\d+) alignments in (\d+) device launches", err)
    assert m and int(m.group(1)) == 985, err[-2000:]
    p = re.search(r"AlignPool: (\d+) reads on (\d+) (?:user-level )?contexts \(limit 256\).*at most (\d+) reads in flight", err)
    assert p and int(p.group(1)) == 142, err[-2000:]
    assert int(p.group(3)) > 16          # more reads in flight than CS threads: the point of the pool
    assert ("user-level contexts" in err) == (fibers == "1")
    if fibers == "1":
        assert re.search(r"; 985 parks;", err), err[-2000:]      # every alignment gave its carrier thread back once
    print("test_3 -t 16, pool of 256 (target %d): %s alignments in %s launches, %s reads in flight at most" % (target, m.group(1), m.group(2), p.group(3)))


def test_candidate_search_bound_in_the_pipeline(built, tmp_path):
    """SURVEY 8 f4 (search half) where it belongs: CS::RunBatch hands every CS thread's batch of sub-reads to
    Convex::CandidateSearchHip (cvx_search_batch_ex over ngmlr's own k-mer table resident in HBM) instead of voting read
    by read in CS::RunRead (reference src/CS.cpp:324-398); alignment contexts, scoring and SAM records on the drop-ins as
    well (oracle/_ref/ngmlr_hip_all).  test_2 (short and long reads), test_4 and test_3 (-t 16, 256 contexts): every SAM
    record identical to the unmodified reference's."""
    import re
    got, err = _run(["-t", "1", "-r", os.path.join(E2E, "ref_chr21_20kb.fa"), "-q", os.path.join(E2E, "reads_100_2200bp.fa")], tmp_path, binary=BIN_ALL)
    assert sorted(got) == sorted(_records(open(os.path.join(ROOT, "tests", "golden", "test_2.sam")).read())) and len(got) == 12
    assert re.search(r"CandidateSearchHip: [1-9]\d* search calls", err), err[-1500:]
    got, err = _run(["-x", "pacbio", "-t", "1", "-r", os.path.join(E2E, "test_4_reference.fasta.gz"),
                     "-q", os.path.join(E2E, "test_4_read.fa.gz")], tmp_path, binary=BIN_ALL)
    assert got == _records(open(os.path.join(ROOT, "tests", "golden", "test_4.sam")).read()) and len(got) == 1
    got, err = _run(_test_3_args(tmp_path, 16), tmp_path, binary=BIN_ALL, env={"CVX_POOL_CONTEXTS": "256"})
    assert sorted(got) == _test_3_want()
    m = re.search(r"CandidateSearchHip: (\d+) search calls, (\d+) reads", err)
    assert m and int(m.group(2)) >= 5663, err[-1500:]          # every sub-read of test_3's 142 reads went through the device search (5 663 lists were recorded from the reference)
    assert re.search(r"SharedAligner: 985 alignments", err), err[-1500:]


def test_every_stage_on_two_logical_devices(built, tmp_path):
    """with CVX_ALIAS_DEVICES=2 (two logical devices on the one GPU of a test box -- the N-device code path, not
    a scaling measurement) the whole pipeline binary deals EVERY stage over the devices: alignment contexts (SharedAligner: two
    backends, two dispatchers), and each CS thread's candidate search and sub-read scoring on its own logical device (the k-mer
    table resident on both, service_device.h).  SAM identical on test_3; the statistics lines say what ran where."""
    import re
    got, err = _run(_test_3_args(tmp_path, 8), tmp_path, binary=BIN_ALL, env={"CVX_POOL_CONTEXTS": "256", "CVX_ALIAS_DEVICES": "2"})
    assert sorted(got) == _test_3_want()
    dev = re.findall(r"CandidateSearchHip: device (\d) \(physical 0\): (\d+) search calls, (\d+) reads", err)
    assert sorted(d for d, _, _ in dev) == ["0", "1"], err[-2500:]
    assert all(int(c) > 0 for _, c, _ in dev) and sum(int(r) for _, _, r in dev) >= 5663
    sc = re.findall(r"StrippedSWHip: (\d+) scoring calls on device (\d) \(physical 0\)", err)
    assert sorted(d for _, d in sc) == ["0", "1"] and all(int(c) > 0 for c, _ in sc), err[-2500:]
    assert re.search(r"SharedAligner: 985 alignments", err), err[-1500:]


@pytest.mark.parametrize("extra", [[], ["--subread-corridor", "80"]])
def test_split_reads_with_structural_variants(built, tmp_path, extra):
    """BASELINE.json configs[4]'s shape end to end: ONT-like reads of 8-30 kb (20 % error) on a random reference, a third of them
    with an inverted segment, a deletion or a foreign insertion, `-x ont` -- ngmlr's split-read path (several intervals per read,
    reverse-strand segments, realignment, supplementary records).  The unmodified reference (CPU, in this very test) against
    the binary with alignment, scoring, candidate search and SAM records on the drop-ins and the alignment contexts: every SAM
    record identical.  Second case: configs[4]'s own flag, `--subread-corridor 80` (reference src/ScoreBuffer.h:65-72: the
    scoring windows grow to 256 + 80 + 12 characters, so StrippedSWHip scores 348-character windows)."""
    import sys
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ngmlr_ref")
    if not os.path.exists(ref_bin) or not os.path.exists(BIN_ALL):
        pytest.skip("oracle/_ref/ngmlr_ref / ngmlr_hip_all not built")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_rates
    fa, fq = str(tmp_path / "sv_ref.fa"), str(tmp_path / "sv_reads.fq")
    e2e_rates.write_sv_workload(fa, fq, 160, seed=77)
    args = ["-x", "ont", "-R", "0.01", "--no-progress"] + extra + ["-r", fa, "-q", fq]
    want, _ = _run(["-t", "16"] + args, tmp_path, binary=ref_bin)
    got, err = _run(["-t", "8"] + args, tmp_path, binary=BIN_ALL, env={"CVX_POOL_CONTEXTS": "128"})
    assert sorted(got) == sorted(want)
    assert "StrippedSWHip:" in err, err[-1500:]
    flags = [int(l.split("\t")[1]) for l in want]
    assert sum(1 for f in flags if f & 2048) >= 20 and any(f & 16 for f in flags), "the workload must exercise split and reverse-strand records"
    assert "CandidateSearchHip:" in err and "AlignPool: 160 reads" in err, err[-1500:]


def test_text_stage_on_the_device_in_the_pipeline(built, tmp_path):
    """CVX_DEVICE_TEXT=1: the dispatcher runs cvx_job_text + cvx_job_nm_profile once per launch (CIGAR, MD, the scalar fields and
    nmPerPosition from the ops and sequences still in HBM) and the workers only copy (ConvexAlignHip::FinishText) instead of
    formatting their own tile on a host core (Finish).  test_3 and the split-read workload (whose small-inversion detection reads
    nmPerPosition, reference src/AlignmentBuffer.cpp:1267-1341): every SAM record identical to the unmodified reference's."""
    import re
    import sys
    env = {"CVX_POOL_CONTEXTS": "256", "CVX_DEVICE_TEXT": "1"}
    got, err = _run(_test_3_args(tmp_path, 16), tmp_path, binary=BIN_ALL, env=env)
    assert sorted(got) == _test_3_want()
    m = re.search(r"text stage on the device for (\d+) launches", err)
    k = re.search(r"SharedAligner: 985 alignments in (\d+) device launches", err)
    assert m and k and int(m.group(1)) == int(k.group(1)), err[-1500:]
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ngmlr_ref")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/ngmlr_ref not built")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_rates
    fa, fq = str(tmp_path / "sv_ref.fa"), str(tmp_path / "sv_reads.fq")
    e2e_rates.write_sv_workload(fa, fq, 120, seed=78)
    args = ["-x", "ont", "-R", "0.01", "--no-progress", "-r", fa, "-q", fq]
    want, _ = _run(["-t", "16"] + args, tmp_path, binary=ref_bin)
    got, err = _run(["-t", "8"] + args, tmp_path, binary=BIN_ALL, env=dict(env, CVX_POOL_CONTEXTS="128"))
    assert sorted(got) == sorted(want)
    assert re.search(r"text stage on the device for [1-9]\d* launches", err), err[-1500:]


def test_reference_windows_decoded_on_the_device_in_the_pipeline(built, tmp_path):
    """VERDICT r4 / r5 item 4 (the default of ngmlr_hip_all; CVX_DEVICE_DECODE=0 turns it off): extractReferenceSequenceForAlignment (reference src/AlignmentBuffer.cpp:199-223)
    hands computeAlignment a placeholder of the window's length instead of running DecodeRefSequenceExact on the worker's core
    (window_decode_binding.inc); every launch of the dispatcher travels through cvx_submit_windows (the device decodes the window
    from ngmlr's own 4-bit genome, uploaded once per device); the workers' host text stage reads the decoded characters that came
    back with the launch's results (cvx_job_window_refs) -- or, with CVX_DEVICE_TEXT=1, CIGAR / MD / nmPerPosition come from the
    device text stage, which runs on its own thread under the kernels of the next launch.  test_2 (short reads: windows at chromosome edges), test_4,
    test_3 (-t 16, 256 contexts), the split-read workload (reverse-strand windows, realignment windows) and the repeat-rich
    one: every SAM record identical to the unmodified reference's, and no tile took its reference as characters."""
    import re
    import sys
    env = {"CVX_POOL_CONTEXTS": "256"}

    def windows(err, alignments=None, device_text=False):
        m = re.search(r"SharedAligner: (\d+) tiles in (\d+) launches took their reference as windows of the genome in HBM .*, (\d+) mixed", err)
        k = re.search(r"SharedAligner: (\d+) alignments in (\d+) device launches", err)
        assert m and k, err[-2500:]
        assert int(m.group(1)) == int(k.group(1)) and int(m.group(2)) == int(k.group(2)) and int(m.group(3)) == 0, (m.groups(), k.groups())
        if alignments is not None:
            assert int(k.group(1)) == alignments
        t = re.search(r"text stage on the device for (\d+) launches", err)
        if device_text:
            assert t and int(t.group(1)) == int(k.group(2)), err[-2500:]
        else:
            assert t is None

    got, err = _run(["-t", "1", "-r", os.path.join(E2E, "ref_chr21_20kb.fa"), "-q", os.path.join(E2E, "reads_100_2200bp.fa")], tmp_path, binary=BIN_ALL, env=env)
    assert sorted(got) == sorted(_records(open(os.path.join(ROOT, "tests", "golden", "test_2.sam")).read())) and len(got) == 12
    windows(err)
    got, err = _run(["-x", "pacbio", "-t", "1", "-r", os.path.join(E2E, "test_4_reference.fasta.gz"),
                     "-q", os.path.join(E2E, "test_4_read.fa.gz")], tmp_path, binary=BIN_ALL, env=env)
    assert got == _records(open(os.path.join(ROOT, "tests", "golden", "test_4.sam")).read()) and len(got) == 1
    windows(err)
    got, err = _run(_test_3_args(tmp_path, 16), tmp_path, binary=BIN_ALL, env=env)
    assert sorted(got) == _test_3_want()
    windows(err, 985)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ngmlr_ref")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/ngmlr_ref not built")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_rates
    fa, fq = str(tmp_path / "sv_ref.fa"), str(tmp_path / "sv_reads.fq")
    e2e_rates.write_sv_workload(fa, fq, 120, seed=79)
    args = ["-x", "ont", "-R", "0.01", "--no-progress", "-r", fa, "-q", fq]
    want, _ = _run(["-t", "16"] + args, tmp_path, binary=ref_bin)
    got, err = _run(["-t", "8"] + args, tmp_path, binary=BIN_ALL, env=dict(env, CVX_POOL_CONTEXTS="128"))
    assert sorted(got) == sorted(want)
    windows(err)
    fa, fq = str(tmp_path / "rep_ref.fa"), str(tmp_path / "rep_reads.fq")
    e2e_rates.write_repeat_workload(fa, fq, 160, seed=92)
    args = ["-x", "pacbio", "-R", "0.01", "--no-progress", "-r", fa, "-q", fq]
    want, _ = _run(["-t", "16"] + args, tmp_path, binary=ref_bin)
    got, err = _run(["-t", "8"] + args, tmp_path, binary=BIN_ALL, env=dict(env, CVX_POOL_CONTEXTS="128"))
    assert sorted(got) == sorted(want)
    windows(err)
    # plain worker threads instead of user-level contexts: the note of a window is thread-local there
    got, err = _run(_test_3_args(tmp_path, 16), tmp_path, binary=BIN_ALL, env=dict(env, CVX_POOL_FIBERS="0"))
    assert sorted(got) == _test_3_want()
    windows(err, 985)
    # off: the reference's decode on the workers' cores, no launch carries a window
    got, err = _run(_test_3_args(tmp_path, 16), tmp_path, binary=BIN_ALL, env=dict(env, CVX_DEVICE_DECODE="0"))
    assert sorted(got) == _test_3_want()
    assert "windows of the genome" not in err and re.search(r"SharedAligner: 985 alignments", err)
    # the text stage on the device as well (its own thread, under the next launch's kernels): nothing of the window returns
    got, err = _run(_test_3_args(tmp_path, 16), tmp_path, binary=BIN_ALL, env=dict(env, CVX_DEVICE_TEXT="1"))
    assert sorted(got) == _test_3_want()
    windows(err, 985, device_text=True)
    got, err = _run(["-t", "8"] + args, tmp_path, binary=BIN_ALL, env=dict(env, CVX_POOL_CONTEXTS="128", CVX_DEVICE_TEXT="1"))
    assert sorted(got) == sorted(want)
    windows(err, device_text=True)


def test_reads_at_contig_ends_of_a_multi_contig_reference(built, tmp_path, monkeypatch):
    """A reference of six contigs, a fifth of the reads flush with a contig's first or last base: their alignment windows reach
    into the 1000-N spacers ngmlr puts between sequences and beyond the chromosome the read lies on -- the branches of
    DecodeRefSequenceExact (reference src/SequenceProvider.cpp:493-565: start in a spacer, end past the chromosome, 'x' fill)
    that the windows decoded on the device inside the pipeline must reproduce.  Unmodified reference against ngmlr_hip_all with
    the device decode (default) and without: every SAM record identical."""
    import re
    import sys
    import numpy as np
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ngmlr_ref")
    if not os.path.exists(ref_bin) or not os.path.exists(BIN_ALL):
        pytest.skip("oracle/_ref/ngmlr_ref / ngmlr_hip_all not built")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_rates
    monkeypatch.setenv("E2E_CONTIGS", "6")
    monkeypatch.setenv("E2E_READ_LEN", "3000:6000")
    fa, fq = str(tmp_path / "ctg_ref.fa"), str(tmp_path / "ctg_reads.fq")
    e2e_rates.write_plain_workload(fa, fq, 400, np.random.default_rng(606), 600000)
    args = ["-x", "pacbio", "-R", "0.01", "--no-progress", "-r", fa, "-q", fq]
    want, _ = _run(["-t", "16"] + args, tmp_path, binary=ref_bin)
    assert len({l.split("\t")[2] for l in want}) == 6, "reads must map to every contig"
    got, err = _run(["-t", "8"] + args, tmp_path, binary=BIN_ALL, env={"CVX_POOL_CONTEXTS": "128"})
    assert sorted(got) == sorted(want)
    m = re.search(r"SharedAligner: (\d+) tiles in (\d+) launches took their reference as windows", err)
    k = re.search(r"SharedAligner: (\d+) alignments in", err)
    assert m and k and m.group(1) == k.group(1) and int(k.group(1)) >= 400, err[-2000:]
    got, err = _run(["-t", "8"] + args, tmp_path, binary=BIN_ALL, env={"CVX_POOL_CONTEXTS": "128", "CVX_DEVICE_DECODE": "0"})
    assert sorted(got) == sorted(want) and "windows of the genome" not in err


def test_repeat_rich_reference(built, tmp_path):
    """What a k-mer vote sees on a real genome: repeat families of 8-20 diverged copies and microsatellites, so that sub-reads cast
    10^4..10^5 votes, overflow the wave kernel's LDS map (forced to its smallest size here: the HBM-table form runs) and reads get several close candidates (MAPQ
    spread over 10..60).  The unmodified reference against the binary with everything on the drop-ins: every SAM record identical."""
    import re
    import sys
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ngmlr_ref")
    if not os.path.exists(ref_bin) or not os.path.exists(BIN_ALL):
        pytest.skip("oracle/_ref/ngmlr_ref / ngmlr_hip_all not built")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_rates
    fa, fq = str(tmp_path / "rep_ref.fa"), str(tmp_path / "rep_reads.fq")
    e2e_rates.write_repeat_workload(fa, fq, 240, seed=91)
    args = ["-x", "pacbio", "-R", "0.01", "--no-progress", "-r", fa, "-q", fq]
    want, _ = _run(["-t", "16"] + args, tmp_path, binary=ref_bin)
    got, err = _run(["-t", "8"] + args, tmp_path, binary=BIN_ALL, env={"CVX_POOL_CONTEXTS": "128", "CVX_SEARCH_TRACE": "1", "CVX_TUNE_SEARCH_LOG2": "9"})      # (the smallest LDS map: since round 6 the default one holds these sub-reads)
    assert sorted(got) == sorted(want)
    mapq = [int(l.split("\t")[4]) for l in want]
    assert sum(1 for q in mapq if q < 40) >= 20 and sum(1 for q in mapq if q >= 40) >= 20, "the workload must produce ambiguous and unambiguous reads"
    to_hbm = sum(int(x) for x in re.findall(r"wave \d+ -> (\d+) to hbm", err))
    assert to_hbm >= 1, "no sub-read overflowed the LDS map: the HBM-table form of the vote did not run"
    assert "CandidateSearchHip:" in err and "AlignPool: 240 reads" in err, err[-1500:]


def test_binary_links_the_device_library(built):
    if not os.path.exists(BIN):
        pytest.skip("ngmlr_hip not built")
    out = subprocess.run(["ldd", BIN], stdout=subprocess.PIPE, text=True).stdout
    assert "libcvxalign.so" in out and "not found" not in out.split("libcvxalign.so")[1].splitlines()[0]
    syms = subprocess.run(["nm", "-C", BIN], stdout=subprocess.PIPE, text=True).stdout
    assert "Convex::ConvexAlignHip::SingleAlign" in syms
