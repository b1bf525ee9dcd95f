/*
 * ngmlr_abi.h -- binary-compatible mirror of the types ngmlr's aligner plugins see.
 *
 * When this backend is compiled inside the ngmlr tree, define CVX_IN_NGMLR_TREE and the
 * real header is used instead (src/IAlignment.h).  Stand-alone (this repository's tests)
 * the declarations below reproduce the *layout and vtable order* of the reference's
 * PositionNM / CorridorLine / Align / IAlignment (reference src/IAlignment.h:16-33,
 * :112-191, :211-247) so that an object built here could be handed to ngmlr unchanged.
 * Interval is only ever carried as a pointer by the hot path, so it stays opaque.
 */
#ifndef CVX_NGMLR_ABI_H
#define CVX_NGMLR_ABI_H

#ifdef CVX_IN_NGMLR_TREE
#include "IAlignment.h"
#else

struct Interval;            /* opaque here: Align::mappedInterval is never dereferenced */

struct PositionNM {
	int refPosition;
	int readPosition;
	int nm;
	PositionNM() : refPosition(0), readPosition(0), nm(0) {}
};

struct CorridorLine {       /* 16 bytes: two ints + the row's prefix-sum slot */
	int offset;
	int length;
	unsigned long offsetInMatrix;
};

struct Align {
	Align() : pBuffer1(0), pBuffer2(0), nmPerPosition(0), mappedInterval(0),
			nmPerPostionLength(0), alignmentLength(0), PositionOffset(0), QStart(0), QEnd(0),
			Score(0.0f), Identity(0.0f), NM(0), MQ(0), cigarOpCount(0), maxBufferLength(20000),
			maxMdBufferLength(20000), skip(false), primary(false), svType(0) {}
	virtual ~Align() {}

	char * pBuffer1;            /* CIGAR text */
	char * pBuffer2;            /* MD text */
	PositionNM * nmPerPosition;
	Interval * mappedInterval;
	PositionNM firstPosition;
	PositionNM lastPosition;
	int nmPerPostionLength;
	int alignmentLength;
	int PositionOffset;
	int QStart;
	int QEnd;
	float Score;
	float Identity;
	int NM;
	int MQ;
	int cigarOpCount;
	int maxBufferLength;
	int maxMdBufferLength;
	bool skip;
	bool primary;
	int svType;
};

class IAlignment {
public:
	virtual int GetScoreBatchSize() const = 0;
	virtual int GetAlignBatchSize() const = 0;
	virtual int BatchScore(int const mode, int const batchSize, char const * const * const refSeqList,
			char const * const * const qrySeqList, float * const results, void * extData) = 0;
	virtual int SingleAlign(int const mode, int const corridor, char const * const refSeq,
			char const * const qrySeq, Align & result, void * extData) { return 0; }
	virtual int SingleAlign(int const mode, CorridorLine * corridor, int const corridorHeight,
			char const * const refSeq, char const * const qrySeq, Align & result,
			int const externalQStart, int const externalQEnd, void * extData) { return 0; }
	virtual int SingleScore(int const mode, int const corridor, char const * const refSeq,
			char const * const qrySeq, float & result, void * extData) { return 0; }
	virtual int BatchAlign(int const mode, int const batchSize, char const * const * const refSeqList,
			char const * const * const qrySeqList, Align * const results, void * extData) = 0;
	virtual ~IAlignment() {}
};

#endif  /* CVX_IN_NGMLR_TREE */
#endif
