"""Dev tool (GPU box): the reference's ngmlr end to end on test/test_3.sh's reads (142 PacBio reads, 985
convex alignments) -- unmodified (CPU ConvexAlignFast), with a private ConvexAlignHip per worker, and with
all workers sharing one BatchingAligner -- wall clock, alignments per device launch, and
whether the sorted SAM records are identical.  The run is dominated by ngmlr's start-up (reference
encoding + index of a 130 kb genome) and its candidate search; it says how the drop-in behaves inside
the real pipeline, not how fast the kernels are."""
import gzip
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E2E = os.path.join(ROOT, "tests", "golden", "e2e")
tmp = tempfile.mkdtemp()
fq = os.path.join(tmp, "test_3.fq")
open(fq, "wb").write(gzip.open(os.path.join(E2E, "test_3_reads.fq.gz"), "rb").read())
want = [l.rstrip("\n") for l in gzip.open(os.path.join(ROOT, "tests", "golden", "test_3.sorted.sam.gz"), "rt") if l.strip()]
threads = [int(x) for x in sys.argv[1:]] or [1, 16, 64]
for name in ("ngmlr_ref", "ngmlr_hip", "ngmlr_hip_batched"):
    binary = os.path.join(ROOT, "oracle", "_ref", name)
    if not os.path.exists(binary):
        print("%-18s not built" % name)
        continue
    for t in threads:
        ref_copy = os.path.join(tmp, "%s_%d.fasta.gz" % (name, t))      # own copy: every run encodes its reference afresh
        open(ref_copy, "wb").write(open(os.path.join(E2E, "test_3_reference.fasta.gz"), "rb").read())
        t0 = time.perf_counter()
        res = subprocess.run([binary, "--skip-write", "-x", "pacbio", "-t", str(t), "-R", "0.01", "--no-progress", "-r", ref_copy, "-q", fq],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp, timeout=900)
        dt = time.perf_counter() - t0
        got = sorted(l for l in res.stdout.splitlines() if l and not l.startswith("@"))
        m = re.search(r"SharedAligner: (\d+) alignments in (\d+) device launches", res.stderr)
        extra = "  %s alignments in %s launches" % (m.group(1), m.group(2)) if m else ""
        print("%-18s -t %-3d wall %6.2f s  rc %d  SAM %s%s" % (name, t, dt, res.returncode, "identical" if got == want else "DIFFERENT (%d records)" % len(got), extra), flush=True)
