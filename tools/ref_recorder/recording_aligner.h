/*
 * recording_aligner.h -- an IAlignment decorator that forwards every call to the
 * reference's own aligner and appends (inputs, outputs) of each corridor SingleAlign
 * to the file named by $CVX_RECORD.  Used ONLY by tools/make_golden.sh, which
 * compiles it into a /tmp copy of the unmodified reference to harvest golden tiles
 * (tests/golden/).  Never part of the product or of the oracle.
 *
 * Record layout (little endian):
 *   int32 magic 0x43565854, refLen, qryLen, height, extQStart, extQEnd
 *   char ref[refLen], qry[qryLen]; int32 off[height], len[height]
 *   int32 ret; float score; int32 PositionOffset, QStart, QEnd, NM, alignmentLength,
 *   cigarOpCount, svType, first_ref, first_read, last_ref, last_read; float identity
 *   int32 cigarLen, mdLen; char cigar[cigarLen], md[mdLen]
 *   int32 nmCount; int32 nm[nmCount*3]
 */
#ifndef RECORDING_ALIGNER_H
#define RECORDING_ALIGNER_H

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <pthread.h>

#include "IAlignment.h"

class RecordingAligner: public IAlignment {
	IAlignment * inner;
	static pthread_mutex_t & mtx() { static pthread_mutex_t m = PTHREAD_MUTEX_INITIALIZER; return m; }
public:
	RecordingAligner(IAlignment * a) : inner(a) {}
	virtual ~RecordingAligner() { delete inner; }
	virtual int GetScoreBatchSize() const { return inner->GetScoreBatchSize(); }
	virtual int GetAlignBatchSize() const { return inner->GetAlignBatchSize(); }
	virtual int BatchScore(int const mode, int const batchSize, char const * const * const refSeqList,
			char const * const * const qrySeqList, float * const results, void * extData) {
		return inner->BatchScore(mode, batchSize, refSeqList, qrySeqList, results, extData);
	}
	virtual int BatchAlign(int const mode, int const batchSize, char const * const * const refSeqList,
			char const * const * const qrySeqList, Align * const results, void * extData) {
		return inner->BatchAlign(mode, batchSize, refSeqList, qrySeqList, results, extData);
	}
	virtual int SingleAlign(int const mode, int const corridor, char const * const refSeq,
			char const * const qrySeq, Align & result, void * extData) {
		return inner->SingleAlign(mode, corridor, refSeq, qrySeq, result, extData);
	}
	virtual int SingleScore(int const mode, int const corridor, char const * const refSeq,
			char const * const qrySeq, float & result, void * extData) {
		return inner->SingleScore(mode, corridor, refSeq, qrySeq, result, extData);
	}
	virtual int SingleAlign(int const mode, CorridorLine * corridor, int const corridorHeight,
			char const * const refSeq, char const * const qrySeq, Align & result,
			int const externalQStart, int const externalQEnd, void * extData) {
		int const ret = inner->SingleAlign(mode, corridor, corridorHeight, refSeq, qrySeq, result,
				externalQStart, externalQEnd, extData);
		char const * path = getenv("CVX_RECORD");
		if (path != 0) {
			pthread_mutex_lock(&mtx());
			FILE * f = fopen(path, "ab");
			if (f != 0) {
				int32_t hdr[6] = { 0x43565854, (int32_t) strlen(refSeq), (int32_t) strlen(qrySeq),
						corridorHeight, externalQStart, externalQEnd };
				fwrite(hdr, 4, 6, f);
				fwrite(refSeq, 1, hdr[1], f);
				fwrite(qrySeq, 1, hdr[2], f);
				for (int i = 0; i < corridorHeight; ++i) { int32_t v = corridor[i].offset; fwrite(&v, 4, 1, f); }
				for (int i = 0; i < corridorHeight; ++i) { int32_t v = corridor[i].length; fwrite(&v, 4, 1, f); }
				int32_t r = ret;
				fwrite(&r, 4, 1, f);
				fwrite(&result.Score, 4, 1, f);
				int32_t a[11] = { result.PositionOffset, result.QStart, result.QEnd, result.NM,
						result.alignmentLength, result.cigarOpCount, result.svType,
						result.firstPosition.refPosition, result.firstPosition.readPosition,
						result.lastPosition.refPosition, result.lastPosition.readPosition };
				fwrite(a, 4, 11, f);
				fwrite(&result.Identity, 4, 1, f);
				int32_t cl = ret >= 0 ? (int32_t) strlen(result.pBuffer1) : 0;
				int32_t ml = ret >= 0 ? (int32_t) strlen(result.pBuffer2) : 0;
				fwrite(&cl, 4, 1, f);
				fwrite(&ml, 4, 1, f);
				fwrite(result.pBuffer1, 1, cl, f);
				fwrite(result.pBuffer2, 1, ml, f);
				int32_t n = 0;
				if (ret >= 0) {
					n = result.alignmentLength < result.nmPerPostionLength ? result.alignmentLength : result.nmPerPostionLength;
				}
				fwrite(&n, 4, 1, f);
				for (int i = 0; i < n; ++i) {
					int32_t t[3] = { result.nmPerPosition[i].refPosition, result.nmPerPosition[i].readPosition, result.nmPerPosition[i].nm };
					fwrite(t, 4, 3, f);
				}
				fclose(f);
			}
			pthread_mutex_unlock(&mtx());
		}
		return ret;
	}
};

#endif
