/*
 * candidate_search_hip.h -- ngmlr's candidate search (the k-mer vote of CS::RunRead, reference src/CS.cpp:324-398 over
 * CS::PrefixSearch :57-99, AddLocationStd :101-149, CollectResultsStd :219-268) for a whole batch of sub-reads on the
 * MI355X, over ngmlr's own k-mer table resident in HBM (search half; C ABI: cvx_index_upload /
 * cvx_search_batch_ex, include/cvx_align.h).
 *
 * In the real pipeline the k-mer vote is what the CS threads spend their time on once alignment and scoring run on the
 * device (measured on the bench box, 20 000 reads of 10 kb: 117-200 of 150-230 CPU seconds of the whole run, on a host
 * that may use 16 cores).  CS::RunBatch hands a CS thread's batch -- cBatchSize reads, i.e. a few hundred 256-bp
 * sub-reads -- to Search() in one call and then does for every read what RunRead does after its search: MappedRead::s,
 * the kCount rule for the mapping quality, AllocScores with the LocationScore list in the reference's order, SendToBuffer
 * (the binding is ngmlr_amd/csrc/cs_search_binding.inc, inserted by tools/build_ngmlr_hip.sh and shown in INTEGRATION.md).
 *
 * One instance per process.  Per DEVICE (round 6, service_device.h: CS thread number k searches and scores on logical device
 * k mod n, as SharedAligner deals the alignment contexts): the table unit is uploaded when a thread of that device first
 * searches, and the device's threads are dealt round-robin over its sixteen handles (own persistent staging; the streams are
 * the process's shared service streams), calls on different handles overlap, a call sleeps while the device works.
 * No CPU path: a device error throws (the CS thread ends, as it would on any other hard error).
 */
#ifndef CANDIDATE_SEARCH_HIP_H
#define CANDIDATE_SEARCH_HIP_H

#include <stdint.h>
#include <vector>

#include "cvx_align.h"

namespace Convex {

class CandidateSearchHip {
public:
	/* The searcher over one unit of the CompactPrefixTable (TableUnit::RefTableIndex: 4^k + 2 packed 5-byte Index records,
	 * TableUnit::RefTable, cRefTableLen, Offset: reference src/PrefixTable.h:15-75), created on first use.  Throws when the
	 * device or its memory is not there: no silent host path. */
	static CandidateSearchHip * Get(int kmerLength, void const * refTableIndex, uint32_t const * refTable, uint32_t nLocations,
			uint64_t unitOffset);
	/* frees the handles and the table (end of the run) and prints the statistics line */
	static void Shutdown();

	/* a thread's reusable buffers: in = seqs / lens, out = everything else */
	struct Batch {
		std::vector<char const *> seqs;      /* NUL-terminated reads (MappedRead::Seq) */
		std::vector<int32_t> lens;           /* MappedRead::length */
		std::vector<int32_t> nCand;          /* entries of read i's list, -1: "too many candidates" (the reference gives up) */
		std::vector<uint64_t> begin;         /* its first entry in cands */
		std::vector<cvx_candidate> cands;    /* (location, score, reverse) in the order CollectResultsStd produces */
		std::vector<float> maxHit;           /* maxHitNumber -> MappedRead::s */
		std::vector<int32_t> kmerMisses;     /* kCount: k-mers of the read the table knows in neither orientation */
		std::vector<int32_t> attempts;       /* table sizes tried: > 1 = the first attempt overflowed (CS::m_Overflows counts those) */
	};
	/* throws 1 on a device error */
	void Search(Batch & b, float sensitivity, float minKmerHits, int binShift, int firstTableBits = 16);

private:
	CandidateSearchHip() { }
	/* what Get() was given: a device's copy of the table is uploaded on that device's first search */
	int kmerLength; void const * refTableIndex; uint32_t const * refTable; uint32_t nLocations; uint64_t unitOffset;
};

}  // namespace Convex

#endif
