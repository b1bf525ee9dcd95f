/*
 * cvx_genome_host.cpp -- host half of the resident-genome entry points of include/cvx_align.h:
 * the 4-bit encoding ngmlr keeps its reference in.  No device code, no HIP.
 *
 * Follows _SequenceProvider::Init (reference src/SequenceProvider.cpp:333-386; enc4 :76-89) and the
 * refStartPos table of :415-424; checked byte for byte against the binRef the unmodified reference
 * builds from its own test genomes (tests/test_decode_cpu.py, fixtures from tools/make_golden.sh).
 */
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "cvx_align.h"

namespace {

const int kMinRefSeqLen = 10;          /* SequenceProvider.h:79: shorter sequences are skipped */
const int kSpacerBytes = 500;          /* 1000 N in front of the first sequence and after every one */

inline unsigned code_of(char c) {      /* enc4: case-insensitive, everything but ACGT is N */
	switch (c) {
	case 'A': case 'a': return 0u;
	case 'T': case 't': return 1u;
	case 'G': case 'g': return 2u;
	case 'C': case 'c': return 3u;
	default: return 4u;
	}
}

}  // namespace

extern "C" {

uint64_t cvx_genome_encoded_bytes(int32_t n, const uint64_t *lengths) {
	uint64_t b = kSpacerBytes;
	for (int32_t i = 0; i < n; ++i)
		if (lengths[i] > (uint64_t) kMinRefSeqLen) b += (lengths[i] + 1) / 2 + kSpacerBytes;
	return b;
}

int cvx_genome_encode(int32_t n, const char *const *seqs, const uint64_t *lengths, uint8_t *bin_ref,
		uint64_t *n_nibbles, uint64_t *start_table, int32_t *n_starts) {
	if (n < 0 || (n > 0 && (!seqs || !lengths)) || !bin_ref || !n_nibbles || !start_table || !n_starts) return CVX_ERR_ARG;
	const uint8_t spacer = (uint8_t) ((4u << 4) | 4u);
	uint64_t at = 0;
	for (int q = 0; q < kSpacerBytes; ++q) bin_ref[at++] = spacer;
	int32_t kept = 0;
	uint64_t last_len = 0;
	for (int32_t i = 0; i < n; ++i) {
		const uint64_t len = lengths[i];
		if (len <= (uint64_t) kMinRefSeqLen) continue;
		if (!seqs[i]) return CVX_ERR_ARG;
		start_table[kept++] = at * 2;                    /* first base of the sequence, in nibbles */
		last_len = len;
		const char *s = seqs[i];
		const uint64_t pairs = len / 2;
		for (uint64_t k = 0; k < pairs; ++k) bin_ref[at++] = (uint8_t) ((code_of(s[2 * k]) << 4) | code_of(s[2 * k + 1]));
		if (len & 1) bin_ref[at++] = (uint8_t) ((code_of(s[len - 1]) << 4) | 4u);
		for (int q = 0; q < kSpacerBytes; ++q) bin_ref[at++] = spacer;
	}
	*n_nibbles = at * 2;
	if (kept == 0) { *n_starts = 0; return CVX_ERR_ARG; }
	start_table[kept] = start_table[kept - 1] + last_len + 1000;   /* upper bound for positions on the last sequence */
	*n_starts = kept + 1;
	return CVX_OK;
}

/*
 * cvx_index_build -- one unit of ngmlr's k-mer table from the encoded genome, on the host: what
 * CompactPrefixTable::CreateTable leaves in TableUnit::RefTableIndex / RefTable (reference src/PrefixTable.cpp:324-352:
 * createRefTableIndex :265-322 over CountKmerFreq / CountKmer :199-226, 372-393; Generate / BuildPrefixTable /
 * SaveToRefTable :228-263, 405-463; the walk over a sequence is CS::PrefixIteration, src/CSstatic.cpp:23-73).  The
 * library does not need it in production -- ngmlr builds (and caches) its own table and cvx_index_upload takes that --
 * but a table the size of a real genome's is needed to measure the device search where its accesses miss every cache,
 * and ngmlr takes minutes to build one.  Restated with the reference's quirks, all of which shape the table:
 *   - a sequence is decoded with DecodeRefSequence(seq, id, start, len) into a buffer of len bytes, which decodes len - 2
 *     characters (:569), turns the last of them into 'x' when that count is odd (:609-611) and leaves the rest NUL; the walk
 *     then runs over all len bytes and encodes every byte that is not 'N' as (c >> 1) & 3 -- 'x' and NUL count as A;
 *   - of a run of equal k-mers at consecutive sampled positions only the first one per bin (position >> bin_shift) is
 *     kept (lastPrefix / lastBin, reset per sequence and by every other k-mer);
 *   - a k-mer is indexed while it and its reverse complement occur fewer than 1000 times together, its weight byte is
 *     (char) ((1000 - total) * 100.0f / 1000) -- which is 0, i.e. "unused", from 991 occurrences on although the slots
 *     stay reserved.
 * Pinned byte for byte against the table the unmodified reference writes to <ref>-ht-13-2.2.ngm (tests/test_index_cpu.py).
 */
int cvx_index_build(const uint8_t *bin_ref, uint64_t n_nibbles, const uint64_t *start_table, const uint64_t *seq_lengths, int32_t n_seqs,
		int32_t kmer_len, int32_t ref_skip, int32_t bin_shift, void *ref_table_index, uint32_t *ref_table, uint64_t ref_table_capacity,
		uint64_t *n_locations) try {
	if (!bin_ref || !start_table || !seq_lengths || n_seqs <= 0 || kmer_len < 4 || kmer_len > 15 || ref_skip < 0 || bin_shift < 0 || bin_shift > 30 ||
			!ref_table_index || !n_locations) return CVX_ERR_ARG;
	const uint64_t n_prefix = 1ull << (2 * kmer_len);
	const uint64_t length = n_prefix + 1;                 /* indexLength (:102) */
	const uint64_t mask = n_prefix - 1;
	const int kMaxFreq = 1000;                            /* CompactPrefixTable::maxPrefixFreq (:28) */
	static const char dec4[] = { 'A', 'T', 'G', 'C', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N' };
	uint64_t longest = 0;
	for (int32_t s = 0; s < n_seqs; ++s) {
		if (start_table[s] & 1ull) return CVX_ERR_ARG;     /* sequences start on a byte (SeqStart = binRefIndex * 2, :350) */
		if (start_table[s] + seq_lengths[s] > n_nibbles) return CVX_ERR_ARG;
		if (seq_lengths[s] > longest) longest = seq_lengths[s];
	}
	std::vector<int32_t> freq;
	try { freq.assign((size_t) length, 0); } catch (const std::bad_alloc &) { return CVX_ERR_OOM; }
	/* sequences are independent (the walk's lastPrefix / lastBin state is reset per sequence): one host thread per sequence
	 * at a time, counters shared through relaxed atomic adds.  CVX_INDEX_THREADS overrides min(sequences, hardware threads, 16). */
	int n_threads = (int) std::thread::hardware_concurrency();
	if (n_threads > 16) n_threads = 16;
	if (const char *e = getenv("CVX_INDEX_THREADS")) n_threads = atoi(e);
	if (n_threads < 1) n_threads = 1;
	if (n_threads > 64) n_threads = 64;                   /* (an unbounded value spawned that many threads per pass: ADVICE r5) */
	const int n_range_threads = n_threads;                /* the passes over the 4^k records do not care how many sequences there are */
	if (n_threads > n_seqs) n_threads = n_seqs;
	std::atomic<bool> oom(false);                         /* written by the workers */

	/* DecodeRefSequence(buf, id, start, len): len - 2 characters, the rest NUL */
	auto decode = [&](int32_t s, std::vector<char> &buf) {
		const uint64_t len = seq_lengths[s], pos = start_table[s];
		buf.assign((size_t) len + 2, 0);
		if (len < 2) return;
		const uint64_t n = len - 2, first = pos / 2;
		uint64_t at = 0;
		for (uint64_t i = 0; i < (n + 1) / 2; ++i) {
			const uint8_t b = bin_ref[first + i];
			buf[(size_t) at++] = dec4[b >> 4];
			buf[(size_t) at++] = dec4[b & 15];
		}
		if (n & 1) buf[(size_t) at - 1] = 'x';
	};
	/* CS::PrefixIteration(seq, len, fn, 0, 0, data, ref_skip, offset), its recursion on N written as a loop */
	auto walk = [&](const char *seq, uint64_t len, uint64_t offset, auto &&fn) {
		const uint64_t k = (uint64_t) kmer_len;
		for (;;) {
			if (len < k) return;
			if (*seq == 'N') {
				uint64_t n_skip = 1;
				while (seq[n_skip] == 'N') ++n_skip;
				seq += n_skip;
				if (n_skip >= len - k) return;
				len -= n_skip; offset += n_skip;
			}
			uint64_t prefix = 0, i = 0;
			bool restart = false;
			for (; i < k - 1; ++i) {
				const char c = seq[i];
				if (c == 'N') { restart = true; break; }
				prefix = (prefix << 2) | (uint64_t) ((c >> 1) & 3);
			}
			if (!restart) {
				uint32_t skipcount = (uint32_t) ref_skip;
				for (i = k - 1; i < len; ++i) {
					const char c = seq[i];
					if (c == 'N') { restart = true; break; }
					prefix = ((prefix << 2) | (uint64_t) ((c >> 1) & 3)) & mask;
					if (skipcount == (uint32_t) ref_skip) { fn(prefix, offset + i + 1 - k); skipcount = 0; }
					else ++skipcount;
				}
				if (!restart) return;
			}
			seq += i + 1; len -= i + 1; offset += i + 1;      /* PrefixIteration(sequence + i + 1, length - i - 1, ..., offset + i + 1) */
		}
	};
	/* every sequence once, on n_threads threads; fn(prefix, pos) sees only the occurrences the reference keeps */
	auto for_kept_kmers = [&](auto &&fn) {
		std::atomic<int32_t> next_seq(0);
		auto work = [&]() {
			std::vector<char> buf;
			for (;;) {
				const int32_t s = next_seq.fetch_add(1);
				if (s >= n_seqs) return;
				try { decode(s, buf); } catch (const std::bad_alloc &) { oom = true; return; }
				uint64_t last_prefix = 111111;
				int64_t last_bin = -1;
				walk(buf.data(), seq_lengths[s], start_table[s], [&](uint64_t prefix, uint64_t pos) {
					if (prefix == last_prefix) {
						const int64_t bin = (int64_t) (pos >> bin_shift);
						if (bin != last_bin || last_bin == -1) fn(prefix, pos);
						last_bin = bin;
					} else {
						last_bin = -1;
						fn(prefix, pos);
					}
					last_prefix = prefix;
				});
			}
		};
		std::vector<std::thread> ths;
		/* (a thread that cannot be created is work the others pick up: the sequences come off a shared counter) */
		for (int t = 1; t < n_threads; ++t) { try { ths.emplace_back(work); } catch (const std::system_error &) { break; } }
		work();
		for (std::thread &t : ths) t.join();
	};
	auto parallel_ranges = [&](uint64_t n, auto &&fn) {      /* fn(begin, end, piece) over [0, n) in n_range_threads pieces */
		std::vector<std::thread> ths;
		const uint64_t per = (n + (uint64_t) n_range_threads - 1) / (uint64_t) n_range_threads;
		int started = 1;
		for (int t = 1; t < n_range_threads; ++t) {
			try { ths.emplace_back([&, t] { fn(std::min(n, per * (uint64_t) t), std::min(n, per * (uint64_t) (t + 1)), t); }); started = t + 1; }
			catch (const std::system_error &) { break; }
		}
		fn(0, std::min(n, per), 0);
		for (int t = started; t < n_range_threads; ++t) fn(std::min(n, per * (uint64_t) t), std::min(n, per * (uint64_t) (t + 1)), t);      /* pieces whose thread could not be created */
		for (std::thread &t : ths) t.join();
	};

	const bool trace = getenv("CVX_INDEX_TRACE") != nullptr;
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (!trace) return;
		const std::chrono::steady_clock::time_point t1 = std::chrono::steady_clock::now();
		fprintf(stderr, "cvx_index_build: %s %.2f s (%d threads)\n", what, std::chrono::duration<double>(t1 - t0).count(), n_threads);
		t0 = t1;
	};
	/* pass 1: CountKmer */
	if (n_threads == 1) for_kept_kmers([&](uint64_t prefix, uint64_t) { freq[(size_t) prefix] += 1; });
	else for_kept_kmers([&](uint64_t prefix, uint64_t) { __atomic_fetch_add(&freq[(size_t) prefix], 1, __ATOMIC_RELAXED); });
	if (oom) return CVX_ERR_OOM;
	lap("count");
	/* createRefTableIndex: 5-byte records (uint m_TabIndex; char m_RevCompIndex), length + 1 of them, zero-initialised (Index()).
	 * revComp (:69-89): complement is xor 2 per base (A 0, C 1, T 2, G 3), then the bases in reverse order.  freq[revComp(i)]
	 * for i in order is a cache miss per entry (the last base of i is the first of its reverse complement); the sums are
	 * therefore formed middle bases outermost: with the k-mer split into head | middle | tail, i = head middle tail and its
	 * reverse complement rc(tail) rc(middle) rc(head) both stay inside two small windows while head and tail vary. */
	std::vector<int32_t> total;
	try { total.assign((size_t) n_prefix, 0); } catch (const std::bad_alloc &) { return CVX_ERR_OOM; }
	{
		const int nt = kmer_len >= 8 ? 4 : kmer_len / 2;          /* bases in the tail and in the head */
		const int nm = kmer_len - 2 * nt;                          /* bases in the middle */
		auto rc_bits = [](uint64_t v, int bases) { uint64_t c = v ^ 0xAAAAAAAAAAAAAAAAull, r = 0; for (int b = 0; b < bases; ++b) { r = (r << 2) | (c & 3); c >>= 2; } return r; };
		const uint64_t n_t = 1ull << (2 * nt), n_m = 1ull << (2 * nm);
		std::vector<uint32_t> rc_t((size_t) n_t);
		for (uint64_t v = 0; v < n_t; ++v) rc_t[(size_t) v] = (uint32_t) rc_bits(v, nt);
		parallel_ranges(n_m, [&](uint64_t m0, uint64_t m1, int) {
			for (uint64_t m = m0; m < m1; ++m) {
				const uint64_t rm = rc_bits(m, nm);
				for (uint64_t h = 0; h < n_t; ++h)
					for (uint64_t t = 0; t < n_t; ++t) {
						const uint64_t i = (h << (2 * (nm + nt))) | (m << (2 * nt)) | t;
						const uint64_t r = ((uint64_t) rc_t[(size_t) t] << (2 * (nm + nt))) | (rm << (2 * nt)) | (uint64_t) rc_t[(size_t) h];
						total[(size_t) i] = freq[(size_t) i] + freq[(size_t) r];
					}
			}
		});
	}
	uint8_t *idx = static_cast<uint8_t *>(ref_table_index);
	auto put_tab = [&](uint64_t i, uint64_t v) { const uint32_t t = (uint32_t) v; memcpy(idx + 5 * i, &t, 4); };
	/* the running sum `next` of the reference's loop (:283-309), as a prefix sum in pieces: every piece sums the frequencies of
	 * the k-mers it keeps, then writes its records from the sum of the pieces before it */
	std::vector<uint64_t> piece_sum((size_t) n_range_threads + 1, 0);
	parallel_ranges(length - 1, [&](uint64_t b, uint64_t e, int piece) {
		uint64_t sum = 0;
		for (uint64_t i = b; i < e; ++i) {
			const int f = freq[(size_t) i];
			if (f > 0 && total[(size_t) i] < kMaxFreq) sum += (uint64_t) f;
		}
		piece_sum[(size_t) piece + 1] = sum;
	});
	for (int t = 0; t < n_range_threads; ++t) piece_sum[(size_t) t + 1] += piece_sum[(size_t) t];
	parallel_ranges(length - 1, [&](uint64_t b, uint64_t e, int piece) {
		uint64_t next = piece_sum[(size_t) piece];
		for (uint64_t i = b; i < e; ++i) {
			const int f = freq[(size_t) i];
			put_tab(i, next + 1);
			idx[5 * i + 4] = 0;
			if (f > 0 && total[(size_t) i] < kMaxFreq) {
				idx[5 * i + 4] = (uint8_t) (char) ((float) (kMaxFreq - total[(size_t) i]) * 100.0f / (float) kMaxFreq);
				next += (uint64_t) f;
			}
		}
	});
	const uint64_t next = piece_sum[(size_t) n_range_threads];
	memset(idx + 5 * (length - 1), 0, 10);                 /* records length - 1 (the end marker) and length (never written by the reference: Index()) */
	put_tab(length - 1, next + 1);
	total = std::vector<int32_t>();
	lap("index");
	*n_locations = next;
	if (next > 0xFFFFFFFFull) return CVX_ERR_ARG;          /* one table unit */
	if (next > ref_table_capacity || (next > 0 && !ref_table)) return CVX_ERR_CAPACITY;
	if (next) memset(ref_table, 0, (size_t) next * 4);
	/* pass 2: BuildPrefixTable (positions relative to the unit's offset 0).  SaveToRefTable takes the first unused slot of the
	 * k-mer's run, i.e. the run fills in the order of the walk: ascending positions.  With several threads the slots are
	 * handed out by an atomic cursor per k-mer (the counter array of pass 1, reused) and every run is sorted afterwards. */
	parallel_ranges(length, [&](uint64_t b, uint64_t e, int) { memset(freq.data() + b, 0, (size_t) (e - b) * sizeof(int32_t)); });
	for_kept_kmers([&](uint64_t prefix, uint64_t pos) {
		if (idx[5 * prefix + 4] == 0) return;              /* RefTableIndex[prefix].used() */
		uint32_t tab;
		memcpy(&tab, idx + 5 * prefix, 4);
		const int32_t slot = n_threads == 1 ? freq[(size_t) prefix]++ : __atomic_fetch_add(&freq[(size_t) prefix], 1, __ATOMIC_RELAXED);
		ref_table[(size_t) (tab - 1) + (size_t) slot] = (uint32_t) pos;
	});
	if (oom) return CVX_ERR_OOM;
	if (n_threads > 1) {
		parallel_ranges(n_prefix, [&](uint64_t p0, uint64_t p1, int) {
			for (uint64_t p = p0; p < p1; ++p) {
				const int32_t n = freq[(size_t) p];
				if (n < 2) continue;
				uint32_t tab;
				memcpy(&tab, idx + 5 * p, 4);
				std::sort(ref_table + (tab - 1), ref_table + (tab - 1) + n);
			}
		});
	}
	lap("fill");
	return CVX_OK;
} catch (const std::bad_alloc &) {      /* nothing leaves an extern "C" function as an exception (a worker's own failures are flagged, see oom) */
	return CVX_ERR_OOM;
} catch (...) {
	return CVX_ERR_OOM;
}

}  /* extern "C" */
