/*
 * stripped_sw_hip.h -- IAlignment implementation of ngmlr's sub-read SCORING calls on the
 * MI355X.  Takes the place of StrippedSW where ngmlr only scores:
 * ScoreBuffer's aligner (reference src/CS.cpp:416, src/ScoreBuffer.cpp:124: BatchScore of
 * up to 1024 (window, sub-read) pairs) and the SingleScore checks
 * (src/AlignmentBuffer.cpp:1224-1225, :2538, src/ScoreBuffer.cpp:267).  Same return values as
 * StrippedSW::BatchScore / SingleScore (src/StrippedSW.cpp:118-203); the alignment entry
 * points throw, as this backend is score-only.
 *
 * ngmlr creates one scoring aligner per worker thread (NGM::CreateAlignment, src/NGM.cpp:350-361, called
 * from src/CS.cpp:416); the StrippedSWHip instances of a process share a small set of device handles per
 * device (a handle's staging buffers are reused call after call, so calls on ONE handle are serialised by a
 * mutex; workers are dealt round-robin over the handles, calls on different handles overlap).
 */
#ifndef STRIPPED_SW_HIP_H
#define STRIPPED_SW_HIP_H

#include "ngmlr_abi.h"
#include "cvx_align.h"

class StrippedSWHip: public IAlignment {
public:
	/* deviceId < 0 (the default): the logical device of the constructing thread (service_device.h: CS thread k of the process
	 * scores and searches on device k mod n) */
	explicit StrippedSWHip(int const deviceId = -1);
	virtual ~StrippedSWHip();

	virtual int GetScoreBatchSize() const { return 1024; }   /* src/StrippedSW.h:53-55 */
	virtual int GetAlignBatchSize() const { return 1024; }

	virtual int BatchScore(int const mode, int const batchSize, char const * const * const refSeqList,
			char const * const * const qrySeqList, float * const results, void * extData);
	virtual int SingleScore(int const mode, int const corridor, char const * const refSeq,
			char const * const qrySeq, float & result, void * extData);
	virtual int BatchAlign(int const, int const, char const * const * const, char const * const * const,
			Align * const, void *) { throw "StrippedSWHip: score-only backend"; }
	virtual int SingleAlign(int const, int const, char const * const, char const * const, Align &, void *) {
		throw "StrippedSWHip: score-only backend";
	}

private:
	int device;
	int lane;      /* which of the device's scoring handles this worker uses */
};

#endif
