/*
 * cvx_index.hip -- one unit of ngmlr's k-mer table built on the device (gfx950; SURVEY.md 8 f4: the data format the
 * candidate search reads).  Same contract as the host builder cvx_index_build (cvx_genome_host.cpp), byte for byte:
 * what CompactPrefixTable::CreateTable leaves in TableUnit::RefTableIndex / RefTable (reference src/PrefixTable.cpp:324-352:
 * createRefTableIndex :265-322 over CountKmerFreq / CountKmer :199-226, :372-393; Generate / BuildPrefixTable /
 * SaveToRefTable :228-263, :405-463; the walk over a sequence is CS::PrefixIteration, src/CSstatic.cpp:23-73).
 *
 * The reference walks a sequence serially; everything it does is local, so it parallelises over positions:
 *   - a sequence of `len` is decoded into len bytes of which the first len - 2 are bases ('N' for a nibble above 3), byte
 *     len - 2 is 'x' when that count is odd and the rest NUL (SequenceProvider::DecodeRefSequence, :569-611); the walk runs
 *     over all len bytes and encodes every byte that is not 'N' as (c >> 1) & 3 -- 'x' and NUL count as A;
 *   - PrefixIteration visits the windows [q, q + k) without an 'N'; with --kmer-skip s it takes every (s + 1)-th window of a
 *     stretch between 'N's, counted from the stretch's first window (the recursion behind an 'N' resets the skip counter);
 *     the sequence's last window is lost when exactly k bytes follow a run of 'N's that is two or more long or starts the
 *     sequence (`n_skip >= length - prefixBasecount`, :33; kmer_windows() in cvx_search.hip is the same closed form);
 *   - of the k-mers so sampled, one is dropped when it equals its two predecessors in the sequence and lies in the bin of
 *     the one before it (CountKmer / BuildPrefixTable's lastPrefix / lastBin, which start a sequence at 111111 / -1);
 *   - a k-mer is indexed while it and its reverse complement occur fewer than 1000 times together; its weight byte is
 *     (char) ((1000 - total) * 100.0f / 1000), 0 = unused from 991 occurrences on -- though a row keeps its (zeroed) slots up to 999;
 *   - a row lists its locations in the order of the walk: ascending.
 * Passes (all HBM-bound integer work): last 'N' per chunk + a max-scan -> sampled windows per chunk, counted, scanned,
 * written out compactly (position, k-mer) -> the drop rule against the two neighbours in that list + a histogram -> totals,
 * weights, row starts (5-byte records) -> a stable radix sort of (k-mer, position) = the rows, ascending.
 * Scan and sort come from rocPRIM.
 */
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#include "cvx_align.h"
#include "cvx_index_build.h"

namespace cvx {
namespace {

const int kChunk = 2048;                 /* window positions per workgroup */
const int kPerThread = 8;                /* ... per thread (256 threads) */
const int kMaxK = 15;

struct SeqDesc { unsigned long long start; unsigned long long len; unsigned long long first_chunk; };

struct IbArgs {
	const uint8_t *bin;
	const SeqDesc *seqs;
	const uint32_t *chunk_seq;             /* chunk -> sequence */
	const uint32_t *chunk_at;              /* chunk -> first window position inside the sequence, / kChunk */
	int n_chunks, n_seqs;
	int k, skip, bin_shift;
	unsigned long long *chunk_scan;        /* per chunk: (sequence << 40) | (last 'N' in the chunk + 1, or 0) -> inclusive max-scan */
	uint32_t *chunk_count;                 /* sampled windows per chunk */
	const unsigned long long *chunk_off;   /* exclusive scan of chunk_count */
	uint32_t *pos, *pre;                   /* the sampled windows, compact, in walk order: absolute position (low 32 bits), k-mer */
};

/* byte i of the decoded sequence: code 0..3 ((c >> 1) & 3 of 'A' 'T' 'G' 'C' = 0 2 3 1), 4 for 'N' */
__device__ __forceinline__ unsigned code_at(const uint8_t *bin, const SeqDesc &s, const unsigned long long i) {
	if (i + 2ull >= s.len) return 0u;      /* 'x' / NUL behind the len - 2 bases */
	const unsigned long long p = s.start + i;
	const unsigned b = bin[p >> 1];
	const unsigned nib = (p & 1ull) ? (b & 15u) : (b >> 4);
	return nib == 0u ? 0u : nib == 1u ? 2u : nib == 2u ? 3u : nib == 3u ? 1u : 4u;
}

/* block-wide inclusive max-scan of one value per thread (256 threads) */
__device__ __forceinline__ long long block_incl_max(long long v, long long *sh) {
	const int t = threadIdx.x;
	sh[t] = v;
	__syncthreads();
	for (int d = 1; d < 256; d <<= 1) {
		const long long u = t >= d ? sh[t - d] : -1ll;
		__syncthreads();
		if (u > sh[t]) sh[t] = u;
		__syncthreads();
	}
	return sh[t];
}
__device__ __forceinline__ uint32_t block_excl_sum(uint32_t v, uint32_t *sh, uint32_t &total) {
	const int t = threadIdx.x;
	sh[t] = v;
	__syncthreads();
	for (int d = 1; d < 256; d <<= 1) {
		const uint32_t u = t >= d ? sh[t - d] : 0u;
		__syncthreads();
		sh[t] += u;
		__syncthreads();
	}
	total = sh[255];
	return sh[t] - v;
}

/* pass 1: the last 'N' of every chunk */
__global__ void __launch_bounds__(256)
ib_last_n_kernel(const IbArgs a) {
	__shared__ long long sh[256];
	const int c = blockIdx.x;
	const uint32_t si = a.chunk_seq[c];
	const SeqDesc s = a.seqs[si];
	const unsigned long long q0 = (unsigned long long) a.chunk_at[c] * kChunk + (unsigned long long) threadIdx.x * kPerThread;
	long long last = -1;
	for (int j = 0; j < kPerThread; ++j) {
		const unsigned long long q = q0 + (unsigned long long) j;
		if (q < s.len && code_at(a.bin, s, q) == 4u) last = (long long) q;
	}
	const long long m = block_incl_max(last, sh);
	if (threadIdx.x == 255) a.chunk_scan[c] = ((unsigned long long) si << 40) | (unsigned long long) (m + 1);
}

/* passes 2 and 3: the sampled windows of a chunk, counted (WRITE = false) or written out */
template <bool WRITE>
__global__ void __launch_bounds__(256)
ib_sample_kernel(const IbArgs a) {
	__shared__ long long sh[256];
	__shared__ uint32_t shc[256];
	const int c = blockIdx.x;
	const uint32_t si = a.chunk_seq[c];
	const SeqDesc s = a.seqs[si];
	const int k = a.k;
	const unsigned long long q0 = (unsigned long long) a.chunk_at[c] * kChunk + (unsigned long long) threadIdx.x * kPerThread;
	/* the last 'N' in front of this thread's first window: in the chunks before (same sequence) ... */
	long long before = -1;
	if ((unsigned long long) c > s.first_chunk) {
		const unsigned long long v = a.chunk_scan[c - 1];
		if ((v >> 40) == (unsigned long long) si) before = (long long) (v & ((1ull << 40) - 1ull)) - 1;
	}
	/* ... and in this chunk: the bytes of this thread's windows, [q0, q0 + kPerThread + k - 1) */
	unsigned codes[kPerThread + kMaxK - 1];
	long long own_last = -1;
	for (int j = 0; j < kPerThread + k - 1; ++j) {
		const unsigned long long q = q0 + (unsigned long long) j;
		codes[j] = q < s.len ? code_at(a.bin, s, q) : 4u;
		if (j < kPerThread && codes[j] == 4u) own_last = (long long) q;
	}
	const long long incl = block_incl_max(own_last, sh);
	__syncthreads();
	long long last_n = threadIdx.x > 0 ? sh[threadIdx.x - 1] : -1ll;       /* in the chunk, in front of this thread */
	if (before > last_n) last_n = before;
	(void) incl;
	/* the sequence's last window falls to the reference's loop-top test when a run of 'N's of two or more, or one that starts the
	 * sequence, ends right in front of it */
	const long long n_win = (long long) s.len - k + 1;
	long long n_eff = n_win;
	if (n_win >= 2) {
		const unsigned long long q = (unsigned long long) (n_win - 1);
		if (code_at(a.bin, s, q - 1) == 4u && (q == 1ull || code_at(a.bin, s, q - 2) == 4u)) n_eff = n_win - 1;
	}
	bool take[kPerThread];
	uint32_t pre[kPerThread];
	uint32_t n_take = 0;
	const uint32_t mask = (uint32_t) ((1ull << (2 * k)) - 1ull);
	for (int j = 0; j < kPerThread; ++j) {
		const long long q = (long long) q0 + j;
		bool ok = q < n_eff;
		uint32_t p = 0;
		for (int b = 0; b < k; ++b) { const unsigned cd = codes[j + b]; ok = ok && cd != 4u; p = (p << 2) | (cd & 3u); }
		const long long stretch = last_n + 1;                       /* first byte behind the last 'N' (or the sequence's first) */
		ok = ok && ((q - stretch) % (long long) (a.skip + 1)) == 0;
		take[j] = ok; pre[j] = p & mask;
		n_take += ok ? 1u : 0u;
		if (codes[j] == 4u) last_n = q;
	}
	uint32_t total = 0;
	const uint32_t rank = block_excl_sum(n_take, shc, total);
	if (!WRITE) {
		if (threadIdx.x == 0) a.chunk_count[c] = total;
		return;
	}
	unsigned long long at = a.chunk_off[c] + rank;
	for (int j = 0; j < kPerThread; ++j) {
		if (!take[j]) continue;
		a.pos[at] = (uint32_t) (s.start + q0 + (unsigned long long) j);
		a.pre[at] = pre[j];
		at += 1;
	}
}

/* pass 4: the drop rule (lastPrefix / lastBin) against the two predecessors in the sequence's list, and the histogram */
__global__ void __launch_bounds__(256)
ib_keep_kernel(const uint32_t *pos, const uint32_t *pre, unsigned long long n, const unsigned long long *seq_first, int n_seqs,
		int bin_shift, uint8_t *kept, uint32_t *freq) {
	const unsigned long long j = (unsigned long long) blockIdx.x * 256ull + threadIdx.x;
	if (j >= n) return;
	int lo = 0, hi = n_seqs;                                  /* the sequence j belongs to: last one with seq_first <= j */
	while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seq_first[mid] <= j) lo = mid; else hi = mid; }
	const unsigned long long first = seq_first[lo];
	const uint32_t p0 = pre[j];
	const uint32_t p1 = j >= first + 1ull ? pre[j - 1] : 111111u;         /* lastPrefix starts a sequence at 111111 */
	bool drop = false;
	if (p0 == p1 && j >= first + 1ull) {
		const bool run = j >= first + 2ull ? pre[j - 2] == p1 : p1 == 111111u;      /* the one before took the "equal" branch: lastBin is its bin */
		drop = run && (pos[j] >> bin_shift) == (pos[j - 1] >> bin_shift);
	}
	kept[j] = drop ? 0 : 1;
	if (!drop) atomicAdd(&freq[p0], 1u);
}

/* pass 5a: total occurrences of a k-mer and its reverse complement -> weight byte, listed count */
__global__ void __launch_bounds__(256)
ib_weight_kernel(const uint32_t *freq, const uint8_t *weight_of_total, int k, unsigned long long n_prefix, uint8_t *weight, uint32_t *listed) {
	const unsigned long long i = (unsigned long long) blockIdx.x * 256ull + threadIdx.x;
	if (i >= n_prefix) return;
	uint64_t c = (i ^ 0xAAAAAAAAAAAAAAAAull) & (n_prefix - 1ull), r = 0;      /* revComp, PrefixTable.cpp:69-89 */
	for (int b = 0; b < k; ++b) { r = (r << 2) | (c & 3ull); c >>= 2; }
	const uint32_t f = freq[i];
	const unsigned long long total = (unsigned long long) f + (unsigned long long) freq[r];
	const uint8_t w = (f > 0u && total < 1000ull) ? weight_of_total[total] : (uint8_t) 0;
	weight[i] = w;
	listed[i] = (f > 0u && total < 1000ull) ? f : 0u;
}

/* pass 5b: the 5-byte records (uint m_TabIndex; char m_RevCompIndex), n_prefix + 2 of them */
__global__ void __launch_bounds__(256)
ib_records_kernel(const unsigned long long *row_start, const uint8_t *weight, unsigned long long n_prefix, unsigned long long next, uint8_t *idx) {
	const unsigned long long i = (unsigned long long) blockIdx.x * 256ull + threadIdx.x;
	if (i > n_prefix + 1ull) return;
	uint32_t tab = 0;
	uint8_t w = 0;
	if (i < n_prefix) { tab = (uint32_t) (row_start[i] + 1ull); w = weight[i]; }
	else if (i == n_prefix) tab = (uint32_t) (next + 1ull);                /* the end marker; the record behind it stays zero (Index()) */
	uint8_t *o = idx + 5ull * i;
	o[0] = (uint8_t) tab; o[1] = (uint8_t) (tab >> 8); o[2] = (uint8_t) (tab >> 16); o[3] = (uint8_t) (tab >> 24); o[4] = w;
}

/* pass 6: sort keys -- the k-mer of a sampled window that is kept and whose row has slots, no_kmer = 4^k (behind every real
 * k-mer) otherwise.  A k-mer with 991..999 occurrences has slots and weight 0: BuildPrefixTable skips it (!used()), its slots
 * keep the zeros the table was allocated with -- it takes part in the sort with location 0. */
__global__ void __launch_bounds__(256)
ib_keys_kernel(uint32_t *pre, uint32_t *pos, const uint8_t *kept, const uint8_t *weight, const uint32_t *listed, unsigned long long n, const uint32_t no_kmer) {
	const unsigned long long j = (unsigned long long) blockIdx.x * 256ull + threadIdx.x;
	if (j >= n) return;
	const uint32_t p = pre[j];
	const bool row = kept[j] && listed[p] != 0u;
	pre[j] = row ? p : no_kmer;
	if (row && weight[p] == 0) pos[j] = 0u;
}

/* for a caller that goes on to search this table on the same device: the rows as the search reads them (cvx_index_upload's
 * records: row start, row length | used << 31; 4^k + 1 of them) */
__global__ void __launch_bounds__(256)
ib_rows_kernel(const unsigned long long *row_start, const uint32_t *listed, const uint8_t *weight, unsigned long long n_prefix, uint2 *rows) {
	const unsigned long long i = (unsigned long long) blockIdx.x * 256ull + threadIdx.x;
	if (i > n_prefix) return;
	const bool used = i < n_prefix && weight[i] != 0;
	rows[i] = make_uint2(used ? (uint32_t) row_start[i] : 0u, used ? (listed[i] | 0x80000000u) : 0u);
}

struct DevMem {      /* everything this build allocates on the device, released together */
	std::vector<void *> blocks;
	template <class T> hipError_t get(T **p, size_t n) {
		void *v = nullptr;
		const hipError_t e = hipMalloc(&v, (n ? n : 1) * sizeof(T));
		if (e == hipSuccess) blocks.push_back(v);
		*p = static_cast<T *>(v);
		return e;
	}
	void keep(void *p) { for (void *&b : blocks) if (b == p) b = nullptr; }      /* (leaves with the caller) */
	~DevMem() { for (void *b : blocks) if (b) (void) hipFree(b); }
};

}  // namespace

#define IB_HIP(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(err, err_len, "%s: %s", #expr, hipGetErrorString(e_)); return e_ == hipErrorOutOfMemory ? CVX_ERR_OOM : CVX_ERR_HIP; } } while (0)

int index_build_device(const uint8_t *bin_ref, uint64_t n_nibbles, const uint64_t *start_table, const uint64_t *seq_lengths, int32_t n_seqs,
		int32_t kmer_len, int32_t ref_skip, int32_t bin_shift, void *ref_table_index, uint32_t *ref_table, uint64_t ref_table_capacity,
		uint64_t *n_locations, hipStream_t st, char *err, size_t err_len, void **resident_rows, uint32_t **resident_locs) {
	const int k = kmer_len;
	if (resident_rows) *resident_rows = nullptr;
	if (resident_locs) *resident_locs = nullptr;
	const unsigned long long n_prefix = 1ull << (2 * k);
	/* chunks never straddle sequences */
	std::vector<SeqDesc> seqs((size_t) n_seqs);
	std::vector<uint32_t> chunk_seq, chunk_at;
	for (int32_t s = 0; s < n_seqs; ++s) {
		seqs[(size_t) s].start = start_table[s];
		seqs[(size_t) s].len = seq_lengths[s];
		seqs[(size_t) s].first_chunk = chunk_seq.size();
		const unsigned long long nch = (seq_lengths[s] + kChunk - 1) / kChunk;
		if (seq_lengths[s] >= (1ull << 40) || nch + chunk_seq.size() >= 0x7FFFFFFFull) { snprintf(err, err_len, "sequence %d is too long for one table unit", s); return CVX_ERR_ARG; }
		for (unsigned long long c = 0; c < nch; ++c) { chunk_seq.push_back((uint32_t) s); chunk_at.push_back((uint32_t) c); }
	}
	const int n_chunks = (int) chunk_seq.size();
	uint8_t weight_of_total[1000];
	for (int t = 0; t < 1000; ++t) weight_of_total[t] = (uint8_t) (char) ((float) (1000 - t) * 100.0f / (float) 1000);      /* PrefixTable.cpp:299, in the host's own float arithmetic */
	*n_locations = 0;
	if (n_chunks == 0) {
		memset(ref_table_index, 0, (size_t) (n_prefix + 2) * 5);
		uint8_t *idx = static_cast<uint8_t *>(ref_table_index);
		for (unsigned long long i = 0; i <= n_prefix; ++i) idx[5 * i] = 1;       /* every row starts at 1 */
		return CVX_OK;
	}
	DevMem mem;
	IbArgs a;
	memset(&a, 0, sizeof(a));
	uint8_t *d_bin; SeqDesc *d_seqs; uint32_t *d_cseq, *d_cat, *d_ccount; unsigned long long *d_cscan, *d_coff;
	const size_t bin_bytes = (size_t) ((n_nibbles + 1) / 2);
	IB_HIP(mem.get(&d_bin, bin_bytes + 16));
	IB_HIP(mem.get(&d_seqs, (size_t) n_seqs));
	IB_HIP(mem.get(&d_cseq, (size_t) n_chunks)); IB_HIP(mem.get(&d_cat, (size_t) n_chunks)); IB_HIP(mem.get(&d_ccount, (size_t) n_chunks + 1));
	unsigned long long *d_cscan_in;
	IB_HIP(mem.get(&d_cscan_in, (size_t) n_chunks)); IB_HIP(mem.get(&d_cscan, (size_t) n_chunks)); IB_HIP(mem.get(&d_coff, (size_t) n_chunks + 1));
	IB_HIP(hipMemsetAsync(d_bin + bin_bytes, 0, 16, st));
	IB_HIP(hipMemcpyAsync(d_bin, bin_ref, bin_bytes, hipMemcpyHostToDevice, st));
	IB_HIP(hipMemcpyAsync(d_seqs, seqs.data(), sizeof(SeqDesc) * (size_t) n_seqs, hipMemcpyHostToDevice, st));
	IB_HIP(hipMemcpyAsync(d_cseq, chunk_seq.data(), 4 * (size_t) n_chunks, hipMemcpyHostToDevice, st));
	IB_HIP(hipMemcpyAsync(d_cat, chunk_at.data(), 4 * (size_t) n_chunks, hipMemcpyHostToDevice, st));
	IB_HIP(hipMemsetAsync(d_ccount, 0, 4 * ((size_t) n_chunks + 1), st));
	a.bin = d_bin; a.seqs = d_seqs; a.chunk_seq = d_cseq; a.chunk_at = d_cat; a.n_chunks = n_chunks; a.n_seqs = n_seqs;
	a.k = k; a.skip = ref_skip; a.bin_shift = bin_shift; a.chunk_scan = d_cscan_in; a.chunk_count = d_ccount; a.chunk_off = d_coff;
	/* 1: last 'N' per chunk, max-scanned (the sequence number in the high bits keeps sequences apart) */
	hipLaunchKernelGGL(ib_last_n_kernel, dim3(n_chunks), dim3(256), 0, st, a);
	IB_HIP(hipGetLastError());
	{
		size_t tmp_bytes = 0;
		IB_HIP(rocprim::inclusive_scan(nullptr, tmp_bytes, d_cscan_in, d_cscan, (size_t) n_chunks, rocprim::maximum<unsigned long long>(), st));
		void *tmp; IB_HIP(mem.get(reinterpret_cast<uint8_t **>(&tmp), tmp_bytes));
		IB_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, d_cscan_in, d_cscan, (size_t) n_chunks, rocprim::maximum<unsigned long long>(), st));
		a.chunk_scan = d_cscan;
	}
	/* 2: sampled windows per chunk, scanned */
	hipLaunchKernelGGL(ib_sample_kernel<false>, dim3(n_chunks), dim3(256), 0, st, a);
	IB_HIP(hipGetLastError());
	{
		size_t tmp_bytes = 0;
		IB_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, d_ccount, d_coff, 0ull, (size_t) n_chunks + 1, rocprim::plus<unsigned long long>(), st));
		void *tmp; IB_HIP(mem.get(reinterpret_cast<uint8_t **>(&tmp), tmp_bytes));
		IB_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, d_ccount, d_coff, 0ull, (size_t) n_chunks + 1, rocprim::plus<unsigned long long>(), st));
	}
	std::vector<unsigned long long> h_coff((size_t) n_chunks + 1);
	IB_HIP(hipMemcpyAsync(h_coff.data(), d_coff, 8 * ((size_t) n_chunks + 1), hipMemcpyDeviceToHost, st));
	IB_HIP(hipStreamSynchronize(st));
	const unsigned long long n_sampled = h_coff[(size_t) n_chunks];
	/* 3: the sampled windows, compact */
	uint32_t *d_pos, *d_pre, *d_pos2 = nullptr, *d_pre2, *d_freq, *d_listed;
	uint8_t *d_kept, *d_weight, *d_wot, *d_idx;
	unsigned long long *d_seq_first, *d_row;
	IB_HIP(mem.get(&d_pos, (size_t) n_sampled)); IB_HIP(mem.get(&d_pre, (size_t) n_sampled)); IB_HIP(mem.get(&d_kept, (size_t) n_sampled));
	a.pos = d_pos; a.pre = d_pre;
	hipLaunchKernelGGL(ib_sample_kernel<true>, dim3(n_chunks), dim3(256), 0, st, a);
	IB_HIP(hipGetLastError());
	/* 4: drop rule + histogram */
	std::vector<unsigned long long> seq_first((size_t) n_seqs);
	for (int32_t s = 0; s < n_seqs; ++s) seq_first[(size_t) s] = seqs[(size_t) s].first_chunk < (unsigned long long) n_chunks ? h_coff[(size_t) seqs[(size_t) s].first_chunk] : n_sampled;
	IB_HIP(mem.get(&d_seq_first, (size_t) n_seqs));
	IB_HIP(hipMemcpyAsync(d_seq_first, seq_first.data(), 8 * (size_t) n_seqs, hipMemcpyHostToDevice, st));
	IB_HIP(mem.get(&d_freq, (size_t) n_prefix)); IB_HIP(mem.get(&d_listed, (size_t) n_prefix + 1)); IB_HIP(mem.get(&d_weight, (size_t) n_prefix));
	IB_HIP(mem.get(&d_wot, 1000)); IB_HIP(mem.get(&d_row, (size_t) n_prefix + 1)); IB_HIP(mem.get(&d_idx, (size_t) (n_prefix + 2) * 5));
	IB_HIP(hipMemsetAsync(d_freq, 0, 4 * (size_t) n_prefix, st));
	IB_HIP(hipMemsetAsync(d_listed, 0, 4 * ((size_t) n_prefix + 1), st));
	IB_HIP(hipMemcpyAsync(d_wot, weight_of_total, 1000, hipMemcpyHostToDevice, st));
	if (n_sampled) {
		hipLaunchKernelGGL(ib_keep_kernel, dim3((unsigned) ((n_sampled + 255) / 256)), dim3(256), 0, st, d_pos, d_pre, n_sampled, d_seq_first, n_seqs, bin_shift, d_kept, d_freq);
		IB_HIP(hipGetLastError());
	}
	/* 5: weights, listed counts, row starts, records */
	hipLaunchKernelGGL(ib_weight_kernel, dim3((unsigned) ((n_prefix + 255) / 256)), dim3(256), 0, st, d_freq, d_wot, k, n_prefix, d_weight, d_listed);
	IB_HIP(hipGetLastError());
	{
		size_t tmp_bytes = 0;
		IB_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, d_listed, d_row, 0ull, (size_t) n_prefix + 1, rocprim::plus<unsigned long long>(), st));
		void *tmp; IB_HIP(mem.get(reinterpret_cast<uint8_t **>(&tmp), tmp_bytes));
		IB_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, d_listed, d_row, 0ull, (size_t) n_prefix + 1, rocprim::plus<unsigned long long>(), st));
	}
	unsigned long long next = 0;
	IB_HIP(hipMemcpyAsync(&next, d_row + n_prefix, 8, hipMemcpyDeviceToHost, st));
	IB_HIP(hipStreamSynchronize(st));
	hipLaunchKernelGGL(ib_records_kernel, dim3((unsigned) ((n_prefix + 2 + 255) / 256)), dim3(256), 0, st, d_row, d_weight, n_prefix, next, d_idx);
	IB_HIP(hipGetLastError());
	IB_HIP(hipMemcpyAsync(ref_table_index, d_idx, (size_t) (n_prefix + 2) * 5, hipMemcpyDeviceToHost, st));
	*n_locations = next;
	if (next > 0xFFFFFFFFull) { IB_HIP(hipStreamSynchronize(st)); snprintf(err, err_len, "%llu locations: more than one table unit holds", next); return CVX_ERR_ARG; }
	if (next > ref_table_capacity || (next > 0 && !ref_table)) { IB_HIP(hipStreamSynchronize(st)); snprintf(err, err_len, "table of %llu locations, room for %llu", next, (unsigned long long) ref_table_capacity); return CVX_ERR_CAPACITY; }
	/* 6: the rows = the kept, listed windows sorted by k-mer; the sort is stable, so every row stays in walk order */
	if (n_sampled && next) {
		hipLaunchKernelGGL(ib_keys_kernel, dim3((unsigned) ((n_sampled + 255) / 256)), dim3(256), 0, st, d_pre, d_pos, d_kept, d_weight, d_listed, n_sampled, (uint32_t) n_prefix);
		IB_HIP(hipGetLastError());
		IB_HIP(mem.get(&d_pos2, (size_t) n_sampled + 1)); IB_HIP(mem.get(&d_pre2, (size_t) n_sampled));
		size_t tmp_bytes = 0;
		IB_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_pre, d_pre2, d_pos, d_pos2, (size_t) n_sampled, 0, 2 * k + 1, st));
		void *tmp; IB_HIP(mem.get(reinterpret_cast<uint8_t **>(&tmp), tmp_bytes));
		IB_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, d_pre, d_pre2, d_pos, d_pos2, (size_t) n_sampled, 0, 2 * k + 1, st));
		IB_HIP(hipMemcpyAsync(ref_table, d_pos2, 4 * (size_t) next, hipMemcpyDeviceToHost, st));
	}
	uint2 *d_rows = nullptr;
	if (resident_rows && resident_locs) {
		IB_HIP(mem.get(&d_rows, (size_t) n_prefix + 1));
		hipLaunchKernelGGL(ib_rows_kernel, dim3((unsigned) ((n_prefix + 1 + 255) / 256)), dim3(256), 0, st, d_row, d_listed, d_weight, n_prefix, d_rows);
		IB_HIP(hipGetLastError());
		if (!(n_sampled && next)) IB_HIP(mem.get(&d_pos2, 1));
	}
	IB_HIP(hipStreamSynchronize(st));
	if (d_rows) {
		mem.keep(d_rows); mem.keep(d_pos2);
		*resident_rows = d_rows; *resident_locs = d_pos2;
	}
	return CVX_OK;
}

}  // namespace cvx
