#!/bin/bash
# tools/make_golden.sh -- harvest golden tiles from the UNMODIFIED reference pipeline.
#
# Copies /root/reference to a fresh /tmp directory, wraps the one
# `new Convex::ConvexAlignFast(...)` (src/AlignmentBuffer.h:355) in
# tools/ref_recorder/recording_aligner.h (a pure pass-through decorator), builds ngmlr
# with its own CMake, runs it on the reference's own test data and converts the recorded
# SingleAlign calls into tests/golden/*.npz via tools/pack_golden.py.  Nothing is written
# to /root/reference; no reference source enters this repository.  Needs /root/reference,
# cmake, zlib (this container only).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REPO="$(dirname "$HERE")"
WORK="$(mktemp -d /tmp/ngmlr_rec.XXXXXX)"
echo "work dir $WORK"
cp -r /root/reference "$WORK/src_tree"
T="$WORK/src_tree"
cp "$HERE/ref_recorder/recording_aligner.h" "$T/src/"
python3 - "$T/src/AlignmentBuffer.h" <<'PY'
import sys, re
p = sys.argv[1]
s = open(p).read()
s = s.replace('#include "ConvexAlignFast.h"', '#include "ConvexAlignFast.h"\n#include "recording_aligner.h"', 1)
pat = re.compile(r'aligner = new Convex::ConvexAlignFast\((.*?)\);', re.S)
m = pat.search(s)
assert m, "construction site not found"
s = s[:m.start()] + 'aligner = new RecordingAligner(new Convex::ConvexAlignFast(' + m.group(1) + '));' + s[m.end():]
open(p, 'w').write(s)
PY
# the reference's own 4-bit genome encoding and every window it decodes for an alignment
# (SURVEY 8 f4): two more pass-through hooks in the temporary copy, switched on by environment variables
python3 - "$T/src" <<'PY'
import sys
src = sys.argv[1]
p = src + '/SequenceProvider.cpp'
s = open(p).read()
anchor = 'refStartPos[j] = refStartPos[j - 1] + SequenceProvider.GetRefLen(refCount - 1) + 1000;'
assert s.count(anchor) == 1
s = s.replace(anchor, anchor + """
	if (getenv("CVX_RECORD_GENOME")) {   /* recorder hook (tools/make_golden.sh), not part of the reference */
		FILE * gf = fopen(getenv("CVX_RECORD_GENOME"), "wb");
		unsigned long long nib = binRefIndex, ns = (unsigned long long) (j + 1);
		fwrite(&nib, 8, 1, gf); fwrite(&ns, 8, 1, gf);
		for (int q = 0; q <= j; ++q) { unsigned long long v = refStartPos[q]; fwrite(&v, 8, 1, gf); }
		fwrite(binRef, 1, (size_t) (nib / 2), gf);
		fclose(gf);
	}
""")
open(p, 'w').write(s)
p = src + '/AlignmentBuffer.cpp'
s = open(p).read()
anchor = 'if (!SequenceProvider.DecodeRefSequenceExact(refSeq, onRefStart, refSeqLength, 0)) {'
assert s.count(anchor) == 1
s = s.replace(anchor, """{ bool cvx_ok = SequenceProvider.DecodeRefSequenceExact(refSeq, onRefStart, refSeqLength, 0);
			if (cvx_ok && getenv("CVX_RECORD_DECODE")) {   /* recorder hook, not part of the reference */
				FILE * df = fopen(getenv("CVX_RECORD_DECODE"), "ab");
				unsigned long long pos = (unsigned long long) onRefStart; int len = refSeqLength;
				fwrite(&pos, 8, 1, df); fwrite(&len, 4, 1, df); fwrite(refSeq, 1, (size_t) len, df);
				fclose(df);
			}
		if (!cvx_ok) {""")
# close the extra brace after the if-block: the block ends with "refSeq = 0;\n\t\t}"
tail = 'delete[] refSeq;\n\t\t\trefSeq = 0;\n\t\t}'
assert s.count(tail) >= 1
i = s.index(tail, s.index('cvx_ok'))
s = s[:i + len(tail)] + ' }' + s[i + len(tail):]
open(p, 'w').write(s)
PY
mkdir -p "$T/build" && cd "$T/build"
cmake .. -DCMAKE_POLICY_VERSION_MINIMUM=3.5 -DCMAKE_BUILD_TYPE=RELWITHDEBINFO > "$WORK/cmake.log" 2>&1
make -j16 > "$WORK/make.log" 2>&1
BIN=$(ls "$T"/bin/ngmlr-*/ngmlr)
echo "built $BIN"
D="$T/test/data"
run() { # name, args...
  local name=$1; shift
  CVX_RECORD="$WORK/$name.rec" CVX_RECORD_GENOME="$WORK/$name.genome" CVX_RECORD_DECODE="$WORK/$name.decode" \
    "$BIN" --skip-write "$@" > "$WORK/$name.sam" 2> "$WORK/$name.log" || true
  echo "$name: $(grep -vc '^@' "$WORK/$name.sam") SAM records, $(stat -c %s "$WORK/$name.rec" 2>/dev/null || echo 0) bytes recorded"
}
run test_2 -t 1 -r "$D/test_2/ref_chr21_20kb.fa" -q "$D/test_2/reads_100_2200bp.fa"
run test_4 -x pacbio -t 1 -r "$D/test_4/reference.fasta.gz" -q "$D/test_4/read.fa.gz"
# test_3 as FASTQ (FASTA + reverse strand crashes the reference)
python3 - "$D/test_3/read.fa.gz" "$WORK/test_3.fq" <<'PY'
import sys, gzip
name = None; seq = []
out = open(sys.argv[2], 'w')
def flush():
    if name is not None:
        s = ''.join(seq)
        out.write('@%s\n%s\n+\n%s\n' % (name, s, 'I' * len(s)))
for line in gzip.open(sys.argv[1], 'rt'):
    line = line.rstrip()
    if line.startswith('>'):
        flush(); name = line[1:]; seq = []
    else:
        seq.append(line)
flush(); out.close()
PY
run test_3 -x pacbio -t 1 -R 0.01 -r "$D/test_3/reference.fasta.gz" -q "$WORK/test_3.fq"
# (the @PG header line carries the temporary directory: normalise it so that regenerating is a no-op)
for t in test_2 test_4; do sed "s#$WORK#/tmp/ngmlr_rec.HzLyKR#g" "$WORK/$t.sam" > "$REPO/tests/golden/$t.sam"; done
# test_3 end to end (tests/test_gpu_e2e.py): the FASTQ form of the reads, the reference genome and the
# unmodified reference's SAM records (sorted: the output order depends on the thread count)
gzip -9 -n -c "$WORK/test_3.fq" > "$REPO/tests/golden/e2e/test_3_reads.fq.gz"
cp "$D/test_3/reference.fasta.gz" "$REPO/tests/golden/e2e/test_3_reference.fasta.gz"
grep -v '^@' "$WORK/test_3.sam" | LC_ALL=C sort | gzip -9 -n > "$REPO/tests/golden/test_3.sorted.sam.gz"
python3 "$HERE/pack_golden.py" "$WORK" "$REPO/tests/golden"
# every recorded test_3 call (985 tiles): too large for the history, kept beside the other
# reference-derived build artefacts (git-ignored, travels to the GPU box with the snapshot)
mkdir -p "$REPO/oracle/_ref/golden_full"
python3 - "$HERE" "$WORK" "$REPO/oracle/_ref/golden_full" <<'PY'
import os, sys
sys.path.insert(0, sys.argv[1])
import pack_golden
pack_golden.pack(pack_golden.read_records(os.path.join(sys.argv[2], 'test_3.rec')), os.path.join(sys.argv[3], 'ref_test_3_full.npz'))
pack_golden.pack_decode(sys.argv[2], 'test_3', os.path.join(sys.argv[3], 'decode_test_3_full.npz'))
PY
rm -rf "$WORK"
