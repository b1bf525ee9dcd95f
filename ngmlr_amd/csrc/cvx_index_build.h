/* cvx_index_build.h -- the device build of ngmlr's k-mer table (cvx_index.hip), called by cvx_index_build_device (cvx_runtime.cpp) */
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cvx {
/* arguments as cvx_index_build (include/cvx_align.h); host buffers in and out, everything on the device allocated and released
 * inside; err: room for a message when the result is not CVX_OK.  resident_rows / resident_locs (or NULL): the table as the search
 * reads it (uint2 rows[4^k + 1], the locations) left on the device for the caller to own (hipFree) */
int index_build_device(const uint8_t *bin_ref, uint64_t n_nibbles, const uint64_t *start_table, const uint64_t *seq_lengths, int32_t n_seqs,
		int32_t kmer_len, int32_t ref_skip, int32_t bin_shift, void *ref_table_index, uint32_t *ref_table, uint64_t ref_table_capacity,
		uint64_t *n_locations, hipStream_t st, char *err, size_t err_len, void **resident_rows, uint32_t **resident_locs);
}
