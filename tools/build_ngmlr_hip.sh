#!/bin/bash
# tools/build_ngmlr_hip.sh -- the drop-in, end to end: builds the reference's own ngmlr
# (its CMake, its sources, in a fresh /tmp copy) with exactly the change INTEGRATION.md
# describes -- Convex::ConvexAlignHip constructed instead of Convex::ConvexAlignFast at
# src/AlignmentBuffer.h:355 -- linked against this repository's libcvxalign.so.  The
# binary lands in oracle/_ref/ngmlr_hip (git-ignored build artefact, travels to the GPU
# box like the .so files); tests/test_gpu_e2e.py runs it on the reference's test data and
# compares the SAM with the unmodified reference's output (tests/golden/test_*.sam).
# Needs /root/reference, cmake, zlib (this container only).  No reference source enters
# the repository.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REPO="$(dirname "$HERE")"
make -s -C "$REPO/ngmlr_amd/csrc"
WORK="$(mktemp -d /tmp/ngmlr_hip.XXXXXX)"
mkdir -p "$REPO/oracle/_ref"
# build_variant OUT_NAME ALIGNER_CLASS: the class constructed at src/AlignmentBuffer.h:355
#   ngmlr_hip          Convex::ConvexAlignHip  one private aligner per worker, one tile per launch
#   ngmlr_hip_batched  Convex::SharedAligner   all -t N workers share one BatchingAligner per device
#   ngmlr_hip_full     Convex::SharedAligner + StrippedSWHip at NGM::CreateAlignment (src/NGM.cpp:355): alignment AND
#                      sub-read scoring on the device
#                      and the SAM records through cvx_sam_record_text (src/SAMWriter.cpp:87)
#   ngmlr_sam          the reference's own CPU aligners; only SAMWriter::DoWriteReadGeneric goes through cvx_sam_record_text
#                      (runs without a GPU: tests/test_sam_cpu.py)
#   ngmlr_hip_pool     ngmlr_hip_full + Convex::AlignPool: processLongReadLIS / processShortRead run on K >> t alignment
#                      contexts instead of on the CS thread (align_pool.h; SURVEY 8 f1's second half), the main loop polls
#                      for the end of the run every 20 ms instead of every 2 s, SAM buffers flush at 256 kB instead of 10 MB (thousands of contexts each own one)
#   ngmlr_hip_all      ngmlr_hip_pool + the candidate search of every CS thread's batch on the device (Convex::CandidateSearchHip,
#                      cs_search_binding.inc at the top of CS::RunBatch, src/CS.cpp:400): alignment, sub-read scoring, k-mer vote
#                      and SAM records on the drop-ins (SURVEY 8 f1 + f2 + f3 + f4's search half)
#                      -- and the k-mer table itself built by cvx_index_build (index_build_binding.inc in CompactPrefixTable::CreateTable)
#   ngmlr_index_cpu    the reference's CPU code with only that table builder bound: the table file it writes against the unmodified
#                      binary's, without a GPU (tests/test_index_cpu.py)
#   ngmlr_pool_cpu     the reference's CPU aligners + the same pool: the pool's own correctness without a GPU (tests/test_pool_cpu.py)
#   ngmlr_pool_parked  the same with a park / wake of the read's user-level context in front of every SingleAlign
#                      (tests/cpp/parking_cpu_aligner.h): the fiber runtime under ngmlr's own long-read stage, no GPU
#   ngmlr_ref          (nothing changed)       the unmodified reference, for wall-clock comparison only
build_variant() {
local OUT_NAME=$1 CLASS=$2 SCORER=${3:-} SAM=${4:-} POOL=${5:-} SEARCH=${6:-} INDEX=${7:-}
local T="$WORK/$OUT_NAME"
cp -r /root/reference "$T"
if [ "$CLASS" != "unmodified" ]; then
python3 - "$T" "$REPO" "$CLASS" "$SCORER" "$SAM" "$POOL" "$SEARCH" "$INDEX" <<'PY'
import re, sys
T, REPO, CLASS, SCORER, SAM, POOL, SEARCH, INDEX = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6], sys.argv[7], sys.argv[8]
def sub1(s, old, new, what):
    assert s.count(old) == 1, (what, s.count(old))
    return s.replace(old, new, 1)
if INDEX:
    # the k-mer table built by cvx_index_build (ngmlr_amd/csrc/index_build_binding.inc at the top of CompactPrefixTable::CreateTable)
    p = T + '/src/SequenceProvider.h'
    s = open(p).read()
    s = sub1(s, '\tChromosome getChrStart(uloc const position);\n', '\tChromosome getChrStart(uloc const position);\n\t/* the 4-bit genome as Init built it (read access for cvx_index_build) */\n'
             '\tchar const * cvxBinRef() const { return binRef; }\n', 'SequenceProvider.h accessor')
    open(p, 'w').write(s)
    p = T + '/src/PrefixTable.cpp'
    s = open(p).read()
    s = sub1(s, '#include "PrefixTable.h"', '#include "PrefixTable.h"\n#include <vector>\n#include <stdlib.h>\n#include <string.h>\n#include <stdint.h>\n#include "cvx_align.h"', 'PrefixTable.cpp includes')
    s = sub1(s, 'void CompactPrefixTable::CreateTable(uint const length) {\n', 'void CompactPrefixTable::CreateTable(uint const length) {\n#include "index_build_binding.inc"\n', 'CompactPrefixTable::CreateTable')
    open(p, 'w').write(s)
if INDEX and POOL:
    # the reference window of an alignment decoded on the device (CVX_DEVICE_DECODE=0 turns it off; ngmlr_amd/csrc/window_decode_binding.inc,
    # Convex::DeviceWindows in convex_align_hip.h): the encoded genome is announced where _SequenceProvider::Init has finished it
    p = T + '/src/SequenceProvider.cpp'
    s = open(p).read()
    s = sub1(s, '#include "SequenceProvider.h"', '#include "SequenceProvider.h"\n#include "convex_align_hip.h"', 'SequenceProvider.cpp include')
    s = sub1(s, '\trefStartPos[j] = refStartPos[j - 1] + SequenceProvider.GetRefLen(refCount - 1) + 1000;\n',
             '\trefStartPos[j] = refStartPos[j - 1] + SequenceProvider.GetRefLen(refCount - 1) + 1000;\n'
             '\tConvex::DeviceWindows::SetGenome(binRef, (unsigned long long) binRefIndex, refStartPos, j + 1);\n', 'SequenceProvider::Init genome')
    open(p, 'w').write(s)
    p = T + '/src/AlignmentBuffer.cpp'
    s = open(p).read()
    s = sub1(s, 'char const * const AlignmentBuffer::extractReferenceSequenceForAlignment(Interval const*& interval, int & refSeqLength) {\n',
             'char const * const AlignmentBuffer::extractReferenceSequenceForAlignment(Interval const*& interval, int & refSeqLength) {\n#include "window_decode_binding.inc"\n', 'extractReferenceSequenceForAlignment')
    open(p, 'w').write(s)
if SEARCH:
    # the k-mer vote of a CS thread's batch on the device (ngmlr_amd/csrc/candidate_search_hip.h, cs_search_binding.inc)
    p = T + '/src/PrefixTable.h'
    s = open(p).read()
    s = sub1(s, '\tstatic int maxPrefixFreq;\n', '\tstatic int maxPrefixFreq;\n\t/* read access to the table units for the device search (what GetRefEntry reads) */\n'
             '\tTableUnit const * cvxUnits(uint & count) const { count = m_UnitCount; return m_Units; }\n', 'PrefixTable.h accessor')
    open(p, 'w').write(s)
    p = T + '/src/CS.cpp'
    s = open(p).read()
    s = sub1(s, '#include "AlignmentBuffer.h"', '#include "AlignmentBuffer.h"\n#include "cs_search_binding.h"', 'CS include (search)')
    s = sub1(s, 'int x_SrchTableLen = (int) pow(2, x_SrchTableBitLen);', 'int x_SrchTableLen = cvxHostVoteTableLen(NGM.GetRefProvider(m_TID), (int) pow(2, x_SrchTableBitLen));', 'CS::DoRun vote table')
    s = sub1(s, '\tint nScoresSum = 0;\n\tfor (size_t i = 0; i < m_CurrentBatch.size(); ++i) {', '#include "cs_search_binding.inc"\n\tint nScoresSum = 0;\n\tfor (size_t i = 0; i < m_CurrentBatch.size(); ++i) {', 'CS::RunBatch')
    s = sub1(s, 'void CS::Cleanup() {', 'void CS::Cleanup() {\n\tConvex::CandidateSearchHip::Shutdown();', 'CS::Cleanup')
    # reads per CS batch = reads per device search / scoring call: the reference's 10 unless CVX_CS_BATCH says otherwise (a measurement knob)
    s = sub1(s, 'int const cBatchSize = 10;', 'int const cBatchSize = cvxCsBatchSize(10);', 'cBatchSize')
    # DoRun's downward adaptation of the vote-table size stops where the device's first attempt does (cs_search_binding.h)
    s = sub1(s, 'if (m_Overflows <= 5 && !up && c_SrchTableBitLen > 8) {', 'if (m_Overflows <= 5 && !up && c_SrchTableBitLen > cvxMinTableBits(m_RefProvider, 8)) {', 'DoRun adaptation floor')
    open(p, 'w').write(s)
if POOL:
    # reads in flight decoupled from the CS threads (ngmlr_amd/csrc/align_pool.h)
    p = T + '/src/CS.cpp'
    s = open(p).read()
    s = sub1(s, '#include "AlignmentBuffer.h"', '#include "AlignmentBuffer.h"\n#include "align_pool.h"', 'CS include')
    s = sub1(s, 'void CS::DoRun() {', 'void CS::DoRun() {\n\tConvex::AlignPool::Attach();', 'CS::DoRun entry')
    s = sub1(s, '\tdelete scoreBuffer;\n\tscoreBuffer = 0;', '\tConvex::AlignPool::Detach();\n\tdelete scoreBuffer;\n\tscoreBuffer = 0;', 'CS::DoRun exit')
    s = sub1(s, 'out->processLongReadLIS(read->group);', 'Convex::AlignPool::Submit(read->group);', 'CS.cpp:296')
    open(p, 'w').write(s)
    p = T + '/src/ScoreBuffer.cpp'
    s = open(p).read()
    s = sub1(s, '#include "AlignmentBuffer.h"', '#include "AlignmentBuffer.h"\n#include "align_pool.h"', 'ScoreBuffer include')
    s = sub1(s, 'out->processLongReadLIS(group);', 'Convex::AlignPool::Submit(group);', 'ScoreBuffer.cpp:155')
    s = sub1(s, 'out->processShortRead(cur_read);', 'Convex::AlignPool::SubmitShort(cur_read);', 'ScoreBuffer.cpp:159')
    s = sub1(s, 'out->processShortRead(read);', 'Convex::AlignPool::SubmitShort(read);', 'ScoreBuffer.cpp:283')
    open(p, 'w').write(s)
    p = T + '/src/NGM.cpp'
    s = open(p).read()
    s = sub1(s, 'Sleep(2000);', 'for (int cvxPoll = 0; cvxPoll < 100 && Running(); ++cvxPoll) Sleep(20);', 'NGM::MainLoop')
    # measurement only: how long the CS threads wait for / hold the lock under which reads are parsed and split (align_pool.h)
    s = sub1(s, 'std::vector<MappedRead*> _NGM::GetNextReadBatch(int desBatchSize) {\n\tNGMLock(&m_Mutex);', 'std::vector<MappedRead*> _NGM::GetNextReadBatch(int desBatchSize) {\n\tlong long const cvxT0 = Convex::AlignPool::ProbeNow();\n\tNGMLock(&m_Mutex);\n\tlong long const cvxT1 = Convex::AlignPool::ProbeNow();', 'GetNextReadBatch lock')
    s = sub1(s, '\tm_CurCount -= desBatchSize;\n\n\tNGMUnlock(&m_Mutex);', '\tm_CurCount -= desBatchSize;\n\n\tConvex::AlignPool::InputLockTimes(cvxT0, cvxT1, Convex::AlignPool::ProbeNow(), count);\n\tNGMUnlock(&m_Mutex);', 'GetNextReadBatch unlock')
    s = sub1(s, '#include "NGM.h"', '#include "NGM.h"\n#include "align_pool.h"\n#include <vector>\n#include <stdlib.h>\n#include <string.h>', 'NGM.cpp include')
    # the read loop in two halves: the record under the input lock, the MappedRead objects outside it (input_batch_binding.inc)
    s = sub1(s, '\tint i = 0;\n\twhile (count < desBatchSize && !eof) {', '\tint i = 0;\n#include "input_batch_binding.inc"\n\twhile (count < desBatchSize && !eof) {', 'GetNextReadBatch loop')
    open(p, 'w').write(s)
    p = T + '/src/IParser.h'
    s = open(p).read()
    s = sub1(s, '\tint parseRead(MappedRead * pRead) {', '\t/* cvx (input_batch_binding.inc): parseRead in two halves -- the raw record (kseq state: under the caller\'s lock), and the MappedRead from it */\n'
             '\tvirtual int cvxReadRecord(kseq_t * rec) { (void) rec; return -3; }\n'
             '\tint cvxFinishRead(MappedRead * pRead, kseq_t * rec, int const l) { return copyToRead(pRead, rec, l); }\n\n\tint parseRead(MappedRead * pRead) {', 'IParser accessors')
    open(p, 'w').write(s)
    p = T + '/src/FastxParser.h'
    s = open(p).read()
    s = sub1(s, '\tvirtual int doParseRead(MappedRead * read) {', '\t/* cvx: the next record, its strings swapped out of the parser (which goes on with the caller\'s previous buffers) */\n'
             '\tvirtual int cvxReadRecord(kseq_t * rec) {\n\t\tint l = kseq_read(tmp);\n\t\t{      /* (also for l < 0: copyToRead reads the name of a record whose quality string has the wrong length) */\n'
             '\t\t\tkstring_t t;\n\t\t\tt = rec->name; rec->name = tmp->name; tmp->name = t;\n\t\t\tt = rec->comment; rec->comment = tmp->comment; tmp->comment = t;\n'
             '\t\t\tt = rec->seq; rec->seq = tmp->seq; tmp->seq = t;\n\t\t\tt = rec->qual; rec->qual = tmp->qual; tmp->qual = t;\n\t\t}\n\t\treturn l;\n\t}\n\n'
             '\tvirtual int doParseRead(MappedRead * read) {', 'FastXParser::cvxReadRecord')
    open(p, 'w').write(s)
    p = T + '/src/ReadProvider.h'
    s = open(p).read()
    s = sub1(s, '\tvirtual MappedRead * NextRead(IParser * parser, int const id);', '\tvirtual MappedRead * NextRead(IParser * parser, int const id, kseq_t * cvxRec = 0, int const cvxL = 0);\npublic:\n'
             '\t/* cvx (input_batch_binding.inc): the raw record under the input lock, NextRead on it outside */\n'
             '\tint cvxReadRecord(kseq_t * rec) { return parser1->cvxReadRecord(rec); }\n'
             '\tMappedRead * cvxGenerateRead(int const readid, kseq_t * rec, int const l) { return NextRead(parser1, readid, rec, l); }\nprivate:', 'ReadProvider accessors')
    open(p, 'w').write(s)
    p = T + '/src/ReadProvider.cpp'
    s = open(p).read()
    s = sub1(s, 'MappedRead * ReadProvider::NextRead(IParser * parser, int const id) {', 'MappedRead * ReadProvider::NextRead(IParser * parser, int const id, kseq_t * cvxRec, int const cvxL) {', 'NextRead signature')
    s = sub1(s, '\t\tl = parser->parseRead(read);', '\t\tl = cvxRec ? parser->cvxFinishRead(read, cvxRec, cvxL) : parser->parseRead(read);', 'NextRead parse')
    open(p, 'w').write(s)
    p = T + '/src/NGM.cpp'
    s = open(p).read()
    open(p, 'w').write(s)
    p = T + '/src/GenericReadWriter.h'
    s = open(p).read()
    s = sub1(s, 'BUFFER_LIMIT =  10000000;', 'BUFFER_LIMIT =  262144;', 'BUFFER_LIMIT')
    open(p, 'w').write(s)
if SAM:
    p = T + '/src/SAMWriter.cpp'
    s = open(p).read()
    m = re.search(r'void SAMWriter::DoWriteReadGeneric\([^)]*\)\s*\{', s)
    assert m and len(re.findall(r'void SAMWriter::DoWriteReadGeneric\(', s)) == 1
    s = s[:m.end()] + '\n#include "sam_writer_binding.inc"\n' + s[m.end():]
    s = s.replace('#include "SAMWriter.h"', '#include "SAMWriter.h"\n#include <vector>\n#include <string.h>\n#include "cvx_align.h"', 1)
    assert 'cvx_align.h' in s
    open(p, 'w').write(s)
if SCORER:
    p = T + '/src/NGM.cpp'
    s = open(p).read()
    assert s.count('instance = new StrippedSW();') == 1
    s = s.replace('instance = new StrippedSW();', 'instance = new %s();' % SCORER, 1)
    s = s.replace('#include "StrippedSW.h"', '#include "StrippedSW.h"\n#include "stripped_sw_hip.h"', 1)
    assert 'stripped_sw_hip.h' in s
    open(p, 'w').write(s)
if CLASS != 'cpu':
    p = T + '/src/AlignmentBuffer.h'
    s = open(p).read()
    s = s.replace('#include "ConvexAlignFast.h"', '#include "ConvexAlignFast.h"\n#include "convex_align_hip.h"\n#include "batching_aligner.h"' + ('\n#include "parking_cpu_aligner.h"' if 'Parking' in CLASS else ''), 1)
    pat = re.compile(r'aligner = new Convex::ConvexAlignFast\(', re.S)
    assert len(pat.findall(s)) == 1
    s = pat.sub('aligner = new %s(' % CLASS, s)
    open(p, 'w').write(s)
p = T + '/src/CMakeLists.txt'
c = open(p).read()
c = c.replace('add_executable(ngmlr', ('add_definitions(-DCVX_IN_NGMLR_TREE)\ninclude_directories(${CMAKE_CURRENT_SOURCE_DIR} %s/include %s/ngmlr_amd/csrc REPO_TESTS_CPP)\nadd_executable(ngmlr\n%s/ngmlr_amd/csrc/convex_align_hip.cpp\n%s/ngmlr_amd/csrc/batching_aligner.cpp\n%s/ngmlr_amd/csrc/stripped_sw_hip.cpp\n%s/ngmlr_amd/csrc/candidate_search_hip.cpp\n%s/ngmlr_amd/csrc/cvx_fiber.cpp%s' % (REPO, REPO, REPO, REPO, REPO, REPO, REPO, ('\n%s/ngmlr_amd/csrc/align_pool.cpp' % REPO) if POOL else '')).replace('REPO_TESTS_CPP', REPO + '/tests/cpp'), 1)
c = c.replace('TARGET_LINK_LIBRARIES(ngmlr ${ZLIB_LIBRARIES})', 'TARGET_LINK_LIBRARIES(ngmlr ${ZLIB_LIBRARIES})\nTARGET_LINK_LIBRARIES(ngmlr %s/ngmlr_amd/libcvxalign.so)\nset_target_properties(ngmlr PROPERTIES BUILD_RPATH "\\$ORIGIN/../../ngmlr_amd;/opt/rocm/lib" SKIP_BUILD_RPATH FALSE)' % REPO, 1)
open(p, 'w').write(c)
PY
fi
mkdir -p "$T/build" && cd "$T/build"
cmake .. -DCMAKE_POLICY_VERSION_MINIMUM=3.5 -DCMAKE_BUILD_TYPE=RELWITHDEBINFO > "$WORK/$OUT_NAME.cmake.log" 2>&1
make -j16 > "$WORK/$OUT_NAME.make.log" 2>&1 || { grep -m5 -B2 -A6 "error" "$WORK/$OUT_NAME.make.log"; exit 1; }
local BIN=$(ls "$T"/bin/ngmlr-*/ngmlr)
cp "$BIN" "$REPO/oracle/_ref/$OUT_NAME"
echo "built $REPO/oracle/_ref/$OUT_NAME"
}
# NGMLR_VARIANTS="ngmlr_pool_cpu ngmlr_hip_all" rebuilds only those (default: all nine)
want() { [ -z "${NGMLR_VARIANTS:-}" ] || [[ " $NGMLR_VARIANTS " == *" $1 "* ]]; }
bv() { if want "$1"; then build_variant "$@" & fi; }
bv ngmlr_hip Convex::ConvexAlignHip
bv ngmlr_hip_batched Convex::SharedAligner
bv ngmlr_hip_full Convex::SharedAligner StrippedSWHip sam
bv ngmlr_sam cpu "" sam
bv ngmlr_hip_pool Convex::SharedAligner StrippedSWHip sam pool
bv ngmlr_pool_cpu cpu "" "" pool
bv ngmlr_pool_parked Convex::ParkingCpuAligner "" "" pool      # CPU aligner behind a park / wake per SingleAlign (tests/cpp/parking_cpu_aligner.h)
bv ngmlr_hip_all Convex::SharedAligner StrippedSWHip sam pool search index
bv ngmlr_index_cpu cpu "" "" "" "" index
bv ngmlr_ref unmodified      # the reference as it is: wall-clock yardstick of tools/e2e_rates.py
wait
for v in ngmlr_hip ngmlr_hip_batched ngmlr_hip_full ngmlr_sam ngmlr_hip_pool ngmlr_pool_cpu ngmlr_pool_parked ngmlr_hip_all ngmlr_index_cpu ngmlr_ref; do
	test -x "$REPO/oracle/_ref/$v" || { echo "missing oracle/_ref/$v"; exit 1; }
done
readelf -d "$REPO/oracle/_ref/ngmlr_hip" | grep -E "RPATH|RUNPATH|NEEDED" | head
rm -rf "$WORK"
