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
mkdir -p "$T/build" && cd "$T/build"
cmake .. -DCMAKE_POLICY_VERSION_MINIMUM=3.5 -DCMAKE_BUILD_TYPE=RELWITHDEBINFO > "$WORK/cmake.log" 2>&1
make -j8 > "$WORK/make.log" 2>&1
BIN=$(ls "$T"/bin/ngmlr-*/ngmlr)
echo "built $BIN"
D="$T/test/data"
run() { # name, args...
  local name=$1; shift
  CVX_RECORD="$WORK/$name.rec" "$BIN" --skip-write "$@" > "$WORK/$name.sam" 2> "$WORK/$name.log" || true
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
cp "$WORK/test_2.sam" "$WORK/test_4.sam" "$REPO/tests/golden/" 2>/dev/null || true
python3 "$HERE/pack_golden.py" "$WORK" "$REPO/tests/golden"
