/* stripped_sw_hip.cpp -- see stripped_sw_hip.h */
#include "stripped_sw_hip.h"

#include <cstdio>

#include <mutex>

namespace {
const int kMaxDevices = 64;
std::mutex g_mtx[kMaxDevices];
std::mutex g_tableMtx;
cvx_handle g_handle[kMaxDevices] = {0};
int g_users[kMaxDevices] = {0};
}

StrippedSWHip::StrippedSWHip(int const deviceId) : device(deviceId >= 0 && deviceId < kMaxDevices ? deviceId : 0) {
	std::lock_guard<std::mutex> g(g_tableMtx);
	if (g_handle[device] == 0) {
		/* the scoring kernel has fixed weights; the handle only needs a valid scoring triple */
		cvx_params p = { 2.0f, -5.0f, -5.0f, -5.0f, -1.0f, 0.15f };
		if (cvx_create(device, &p, 0, &g_handle[device]) != CVX_OK) {
			fprintf(stderr, "StrippedSWHip: %s\n", cvx_last_error());
			g_handle[device] = 0;
			throw "StrippedSWHip: no usable MI355X";
		}
	}
	g_users[device] += 1;
}

StrippedSWHip::~StrippedSWHip() {
	std::lock_guard<std::mutex> g(g_tableMtx);
	if (--g_users[device] == 0) {
		std::lock_guard<std::mutex> d(g_mtx[device]);
		cvx_destroy(g_handle[device]);
		g_handle[device] = 0;
	}
}

int StrippedSWHip::BatchScore(int const mode, int const batchSize, char const * const * const refSeqList,
		char const * const * const qrySeqList, float * const results, void * extData) {
	(void) mode; (void) extData;
	std::lock_guard<std::mutex> d(g_mtx[device]);
	if (cvx_score_batch(g_handle[device], batchSize, refSeqList, qrySeqList, results) != CVX_OK) {
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
