#!/bin/bash
# tools/make_golden_cs.sh -- harvest candidate-search fixtures from the UNMODIFIED reference pipeline.
#
# A fresh /tmp copy of /root/reference gets two pass-through recorder hooks, switched on by environment variables:
#   CS::CollectResultsStd (src/CS.cpp:219-268)   every sub-read it was called for and the LocationScore list it produced
#                                                 (in list order), plus maxHitNumber / the threshold it applied, kCount -- the
#                                                 k-mers of the read found in neither orientation, summed over the attempts of
#                                                 the retry ladder (src/CS.cpp:26,67-69,338; a global the CS threads share: the
#                                                 run is -t 1, where it is the read's own count) -- and the table size the
#                                                 read's first attempt ran with (c_SrchTableBitLen at src/CS.cpp:338, which the
#                                                 reference adapts per batch, :482-489)
#   CompactPrefixTable::CompactPrefixTable        the k-mer table the search ran on, in compact form: the used prefixes with
#   (src/PrefixTable.cpp:97-131)                  their slot counts and the RefTable (the 4^13 + 1 entry index is a running
#                                                 sum over those, GetRefEntry reads nothing else: src/PrefixTable.cpp:476-532)
# ngmlr then maps the reference's own test_3 reads; tools/pack_golden_cs.py turns the dumps into tests/golden/cs_test_3.npz
# (a sample) and oracle/_ref/golden_full/cs_test_3_full.npz (every sub-read).  Nothing is written to /root/reference; no
# reference source enters this repository.  Needs /root/reference, cmake, zlib (this container only).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REPO="$(dirname "$HERE")"
WORK="$(mktemp -d /tmp/ngmlr_cs.XXXXXX)"
cp -r /root/reference "$WORK/src_tree"
T="$WORK/src_tree"
python3 - "$T/src" <<'PY'
import sys
src = sys.argv[1]
p = src + '/CS.cpp'
s = open(p).read()
anchor = '\tstatic int const maxScores = Config.getMaxCMRs();'
assert s.count(anchor) == 1
s = s.replace(anchor, """\tif (getenv("CVX_RECORD_CS")) {   /* recorder hook (tools/make_golden_cs.sh), not part of the reference */
		static pthread_mutex_t cvx_m = PTHREAD_MUTEX_INITIALIZER;
		pthread_mutex_lock(&cvx_m);
		FILE * cf = fopen(getenv("CVX_RECORD_CS"), "ab");
		int len = read->length, nn = index, rl = rListLength, kc = kCount, fb = cvx_first_bits;
		float mh = maxHitNumber, th = mi_Threshhold;
		fwrite(&len, 4, 1, cf); fwrite(read->Seq, 1, (size_t) len, cf);
		fwrite(&mh, 4, 1, cf); fwrite(&th, 4, 1, cf); fwrite(&rl, 4, 1, cf); fwrite(&nn, 4, 1, cf); fwrite(&kc, 4, 1, cf); fwrite(&fb, 4, 1, cf);
		for (int q = 0; q < nn; ++q) {
			unsigned long long l = tmp[q].Location.m_Location; float f = tmp[q].Score.f; int r = tmp[q].Location.isReverse() ? 1 : 0;
			fwrite(&l, 8, 1, cf); fwrite(&f, 4, 1, cf); fwrite(&r, 4, 1, cf);
		}
		fclose(cf);
		pthread_mutex_unlock(&cvx_m);
	}
""" + anchor)
s = s.replace('#include <memory.h>', '#include <memory.h>\n#include <pthread.h>\nstatic int cvx_first_bits = 0;   /* recorder: table size of the read\'s first attempt */', 1)
anchor = '\tkCount = 0;'
assert s.count(anchor) == 1
s = s.replace(anchor, anchor + ' cvx_first_bits = c_SrchTableBitLen;   /* recorder hook */')
open(p, 'w').write(s)
p = src + '/PrefixTable.cpp'
s = open(p).read()
anchor = '\tdelete[] cacheFile;\n\tcacheFile = 0;'
assert s.count(anchor) == 1
s = s.replace(anchor, """\tif (getenv("CVX_RECORD_TABLE")) {   /* recorder hook (tools/make_golden_cs.sh), not part of the reference */
		FILE * tf = fopen(getenv("CVX_RECORD_TABLE"), "wb");
		unsigned int k = m_PrefixLength, units = (unsigned int) m_UnitCount, skip = m_RefSkip;
		unsigned int indexLength = (unsigned int) pow(4.0, (double) m_PrefixLength) + 1;
		fwrite(&k, 4, 1, tf); fwrite(&units, 4, 1, tf); fwrite(&skip, 4, 1, tf);
		for (int u = 0; u < m_UnitCount; ++u) {
			TableUnit & cu = m_Units[u];
			unsigned long long off = cu.Offset; unsigned int tl = cu.cRefTableLen, nused = 0;
			for (unsigned int q = 0; q < indexLength - 1; ++q) if (cu.RefTableIndex[q].used()) nused++;
			fwrite(&off, 8, 1, tf); fwrite(&tl, 4, 1, tf); fwrite(&nused, 4, 1, tf);
			for (unsigned int q = 0; q < indexLength - 1; ++q) if (cu.RefTableIndex[q].used()) {
				unsigned int cnt = cu.RefTableIndex[q + 1].m_TabIndex - cu.RefTableIndex[q].m_TabIndex, ti = cu.RefTableIndex[q].m_TabIndex;
				signed char rc = cu.RefTableIndex[q].m_RevCompIndex;
				fwrite(&q, 4, 1, tf); fwrite(&ti, 4, 1, tf); fwrite(&cnt, 4, 1, tf); fwrite(&rc, 1, 1, tf);
			}
			fwrite(cu.RefTable, 4, (size_t) tl, tf);
		}
		fclose(tf);
	}
""" + anchor)
open(p, 'w').write(s)
PY
mkdir -p "$T/build" && cd "$T/build"
cmake .. -DCMAKE_POLICY_VERSION_MINIMUM=3.5 -DCMAKE_BUILD_TYPE=RELWITHDEBINFO > "$WORK/cmake.log" 2>&1
make -j16 > "$WORK/make.log" 2>&1 || { tail -30 "$WORK/make.log"; exit 1; }
BIN=$(ls "$T"/bin/ngmlr-*/ngmlr)
if [ "${1:-}" = "--big" ]; then
# the one regime without a reference-made vector -- a k-mer table that leaves every cache.  The unmodified
# reference maps 20 PacBio-like 10 kb reads (800 sub-reads) on ngmlr_amd.synth.big_reference (512 Mbp in 8 contigs, repeat families,
# microsatellites: the reference of bench.py's candidate_search_big), recorder hooks on.  The 1 GB table is NOT stored: the packer
# checks that cvx_index_build produces the very table the reference built (every used prefix, its slot count and weight byte, every
# location) and keeps its SHA-256; the tests rebuild the table from the same generator, check the hash and search over it.
# -> tests/golden/cs_big.npz (the recorded calls + the hashes; small enough to commit)
python3 - "$REPO" "$WORK/big.fa" "$WORK/big.fq" <<'PY'
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from ngmlr_amd import synth
contigs = synth.big_reference(512 << 20, n_contigs=8)
with open(sys.argv[2], "wb") as f:
    for i, c in enumerate(contigs):
        f.write(b">big%d\n" % i)
        b = c.tobytes()
        for a in range(0, len(b), 1 << 20):
            f.write(b[a:a + (1 << 20)] + b"\n")
rng = np.random.default_rng(77)
with open(sys.argv[3], "w") as f:
    for i in range(20):
        c = contigs[int(rng.integers(0, len(contigs)))]
        a = int(rng.integers(0, len(c) - 12000))
        q = synth.mutate(rng, c[a:a + 10400], 0.15, (6, 3, 1))
        if rng.random() < 0.5:
            q = synth.revcomp(q)
        f.write("@big%d_%d\n%s\n+\n%s\n" % (i, a, q.tobytes().decode(), "I" * len(q)))
PY
CVX_RECORD_CS="$WORK/big.cs" CVX_RECORD_TABLE="$WORK/big.table" "$BIN" --skip-write -x pacbio -t 1 -R 0.01 --no-progress \
	-r "$WORK/big.fa" -q "$WORK/big.fq" > "$WORK/big.sam" 2> "$WORK/big.log" || true
echo "big: $(stat -c %s "$WORK/big.cs") bytes of sub-read records, $(stat -c %s "$WORK/big.table") bytes of table"
python3 "$HERE/pack_golden_cs.py" --big "$WORK/big.cs" "$WORK/big.table" "$REPO/tests/golden/cs_big.npz"
else
D="$T/test/data"
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
CVX_RECORD_CS="$WORK/test_3.cs" CVX_RECORD_TABLE="$WORK/test_3.table" "$BIN" --skip-write -x pacbio -t 1 -R 0.01 --no-progress \
	-r "$D/test_3/reference.fasta.gz" -q "$WORK/test_3.fq" > "$WORK/test_3.sam" 2> "$WORK/test_3.log" || true
echo "test_3: $(stat -c %s "$WORK/test_3.cs") bytes of sub-read records, $(stat -c %s "$WORK/test_3.table") bytes of table"
mkdir -p "$REPO/oracle/_ref/golden_full"
python3 "$HERE/pack_golden_cs.py" "$WORK/test_3.cs" "$WORK/test_3.table" "$REPO/tests/golden/cs_test_3.npz" "$REPO/oracle/_ref/golden_full/cs_test_3_full.npz"
# round 5: the same recording on a repeat-rich reference (repeat families of diverged copies, microsatellites: tools/e2e_rates.py
# write_repeat_workload at 1.2 Mbp, plus one 400-bp unit in 700 copies) -- sub-reads with 10^3..10^5 votes, lists with many close candidates, table sizes adapted UP by the
# reference, reads that overflow the device kernel's LDS vote map and take its HBM-table form.  -> tests/golden/cs_rep.npz (every 2nd
# recorded call; the table in compact form as above) and oracle/_ref/golden_full/cs_rep_full.npz (all of them)
python3 - "$REPO" "$WORK/rep.fa" "$WORK/rep.fq" <<'PY'
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tools")
import e2e_rates
e2e_rates.write_repeat_workload(sys.argv[2], sys.argv[3], 64, seed=4711, L=1_200_000, dense=700)
PY
CVX_RECORD_CS="$WORK/rep.cs" CVX_RECORD_TABLE="$WORK/rep.table" "$BIN" --skip-write -x pacbio -t 1 -R 0.01 --no-progress \
	-r "$WORK/rep.fa" -q "$WORK/rep.fq" > "$WORK/rep.sam" 2> "$WORK/rep.log" || true
echo "repeat-rich: $(stat -c %s "$WORK/rep.cs") bytes of sub-read records, $(stat -c %s "$WORK/rep.table") bytes of table"
python3 "$HERE/pack_golden_cs.py" "$WORK/rep.cs" "$WORK/rep.table" "$REPO/tests/golden/cs_rep.npz" "$REPO/oracle/_ref/golden_full/cs_rep_full.npz" 2
fi
rm -rf "$WORK"
