/* stripped_sw_hip.cpp -- see stripped_sw_hip.h */
#include "stripped_sw_hip.h"

#include <cstdio>

StrippedSWHip::StrippedSWHip(int const deviceId) : handle(0) {
	/* the scoring kernel has fixed weights; the handle only needs a valid scoring triple */
	cvx_params p = { 2.0f, -5.0f, -5.0f, -5.0f, -1.0f, 0.15f };
	if (cvx_create(deviceId, &p, 0, &handle) != CVX_OK) {
		fprintf(stderr, "StrippedSWHip: %s\n", cvx_last_error());
		throw "StrippedSWHip: no usable MI355X";
	}
}

StrippedSWHip::~StrippedSWHip() {
	cvx_destroy(handle);
}

int StrippedSWHip::BatchScore(int const mode, int const batchSize, char const * const * const refSeqList,
		char const * const * const qrySeqList, float * const results, void * extData) {
	(void) mode; (void) extData;
	if (cvx_score_batch(handle, batchSize, refSeqList, qrySeqList, results) != CVX_OK) {
		fprintf(stderr, "StrippedSWHip: %s\n", cvx_last_error());
		throw 1;
	}
	return batchSize;
}

int StrippedSWHip::SingleScore(int const mode, int const corridor, char const * const refSeq,
		char const * const qrySeq, float & result, void * extData) {
	(void) corridor;
	char const * r[1] = { refSeq };
	char const * q[1] = { qrySeq };
	float s = -1.0f;
	BatchScore(mode, 1, r, q, &s, extData);
	result = s;
	return s == -1.0f ? 0 : 1;     /* the reference returns 0 when a sequence is too long */
}
