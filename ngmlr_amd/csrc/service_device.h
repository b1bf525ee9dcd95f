/*
 * service_device.h -- which device a thread's pipeline stages run on (SURVEY.md 8 e).
 *
 * In the reference every `-t` thread runs the whole pipeline on its own reads (src/CS.cpp:412-503, src/NGM.cpp:340-348).
 * With the stages on the device, a node with several MI355X must not align on eight of them and vote and score on one:
 * SharedAligner deals the alignment contexts over the devices, and the CS threads' search and scoring calls follow the
 * same rule -- CS thread number k of the process uses logical device k mod n for both (its k-mer table resident there,
 * its scoring handles there).  Logical devices are the physical ones unless CVX_ALIAS_DEVICES=k puts k of them on the
 * devices present (the N-device code path on a one-GPU box: tests, not a scaling measurement); CVX_DEVICES=k uses only the
 * first k.  One definition, shared by batching_aligner.cpp, candidate_search_hip.cpp and stripped_sw_hip.cpp.
 */
#ifndef CVX_SERVICE_DEVICE_H
#define CVX_SERVICE_DEVICE_H

#include <atomic>
#include <cstdlib>

#include "cvx_align.h"

namespace Convex {

static const int kMaxLogicalDevices = 64;

/* logical devices of this process and the physical devices they live on (0 / 0 without a device) */
inline void DeviceLayout(int & nLogical, int & nPhysical) {
	int nDev = cvx_device_count();
	if (const char * e = getenv("CVX_DEVICES")) nDev = atoi(e) < nDev ? atoi(e) : nDev;      /* use only the first k devices */
	if (nDev > kMaxLogicalDevices) nDev = kMaxLogicalDevices;
	if (nDev < 0) nDev = 0;
	nPhysical = nDev;
	nLogical = nDev;
	if (const char * e = getenv("CVX_ALIAS_DEVICES")) {
		if (atoi(e) > 0 && nPhysical > 0) nLogical = atoi(e) < kMaxLogicalDevices ? atoi(e) : kMaxLogicalDevices;
	}
}

inline int PhysicalDeviceOf(int logical) {
	int nl = 0, np = 0;
	DeviceLayout(nl, np);
	return np > 0 ? logical % np : logical;
}

/* the logical device of the calling thread's search and scoring calls: threads are numbered in the order they first ask */
inline int ServiceDeviceOfThisThread() {
	static std::atomic<int> next(0);
	thread_local int mine = -1;
	if (mine < 0) {
		int nl = 0, np = 0;
		DeviceLayout(nl, np);
		mine = nl > 0 ? next.fetch_add(1) % nl : 0;
	}
	return mine;
}

}  // namespace Convex

#endif
