/*
 * cs_search_binding.h -- ngmlr-tree side of the candidate-search binding (see cs_search_binding.inc): which table the
 * device search can take.  Only meaningful inside ngmlr's tree (it names the reference's own types); included by the
 * patched src/CS.cpp (tools/build_ngmlr_hip.sh).
 */
#ifndef CS_SEARCH_BINDING_H
#define CS_SEARCH_BINDING_H

#include <stdlib.h>

#include "PrefixTable.h"
#include "candidate_search_hip.h"

/* the one table unit of a genome below 4 Gbp (src/PrefixTable.h:53-75), or 0: several units keep the reference's search */
static inline TableUnit const * cvxSearchableUnit(IRefProvider const * rp) {
	CompactPrefixTable const * table = dynamic_cast<CompactPrefixTable const *>(rp);
	uint unitCount = 0;
	TableUnit const * units = table ? table->cvxUnits(unitCount) : 0;
	return (units != 0 && unitCount == 1) ? units : 0;
}

/* CS::DoRun sizes the thread's host vote table with this (src/CS.cpp:422-432: 2^24 entries + list = 320 MB per CS thread,
 * written once at start-up): a thread whose votes run on the device never touches it */
static inline int cvxHostVoteTableLen(IRefProvider const * rp, int referenceLen) {
	return cvxSearchableUnit(rp) != 0 ? 1 : referenceLen;
}

/* The smallest size DoRun's per-batch adaptation (src/CS.cpp:482-489) may take the thread's vote table down to.  The reference
 * goes down to 2^8 while few first attempts overflow; with the device search bound the first attempt never runs below 2^16
 * (cs_search_binding.inc), so the thread's own value stops there as well: the overflows it counts and the size they were
 * counted at then agree, and the upward adaptation does not have to climb from 8 to 16 on overflows measured at 2^16 before
 * the binding follows it. */
static inline int cvxMinTableBits(IRefProvider const * rp, int referenceValue) {
	return cvxSearchableUnit(rp) != 0 ? 16 : referenceValue;
}

/* reads per CS batch (src/CS.cpp:34: 10, "reduced batch size for low read number PacBio samples") -- with the vote on the device
 * it is also the size of a search call.  The reference's value unless CVX_CS_BATCH is set (measurements: a call of 100 reads
 * amortises the host round trips of the ladder ten times over; DoRun's per-batch adaptation thresholds scale with it) */
static inline int cvxCsBatchSize(int referenceValue) {
	const char * e = getenv("CVX_CS_BATCH");
	return (e && atoi(e) > 0) ? atoi(e) : referenceValue;
}

#endif
