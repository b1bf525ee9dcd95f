/*
 * cvx_text.hip -- device-side text stage (gfx950): run-length ops -> SAM CIGAR and
 * MD strings plus the scalar Align fields, for every tile of a finished batch, from the ops and the
 * sequences that are still resident in HBM.
 *
 * Behavioural contract = convertCigar, reference src/ConvexAlignFast.cpp:112-333 (EQ and X merged
 * into M, flushed before every I / D and at the end; leading / trailing S with the external
 * clips; MD "%d%c" per mismatched base, "%d^BASES" per deletion, trailing "%d"; NM, Identity,
 * alignmentLength, cigarOpCount, first/last positions) and the N-clip flags of :493-528 -- the same
 * contract as the host form cvx_format_alignment (cvx_format.cpp), against which it is tested field
 * by field and byte by byte.  The per-position mismatch profile (nmPerPosition, 12 bytes per
 * alignment column, several GB per batch; only detectMisalignment reads it) has its own kernel
 * (nm_profile_kernel below) and its own entry point: a caller that wants it on the host pays 12 B per
 * column of PCIe, a consumer on the device does not.
 *
 * One wave per tile, 64 ops per step.  A CIGAR piece is the sum of a maximal EQ/X run -- a
 * segmented sum: inclusive wave scan minus the scan value in front of the run's first lane, found
 * with a ballot -- and the MD match counter in front of an X or D op is the same construction over
 * the EQ lengths since the previous X / D.  Text offsets are a third scan over the piece lengths;
 * every lane then writes its own piece.  text_size_kernel only measures (lengths + fields), a
 * one-workgroup scan turns the lengths into offsets, text_write_kernel writes.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cvx_types.h"
#include "cvx_launch.h"

namespace cvx {

#define TXT_DEV __device__ __forceinline__
typedef unsigned long long u64;

TXT_DEV int wave_scan(int v, const int lane) {          /* inclusive prefix sum over the wave */
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const int t = __shfl_up(v, d, 64);
		if (lane >= d) v += t;
	}
	return v;
}
TXT_DEV int wave_sum(int v) {
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
	return v;
}
TXT_DEV int ndigits(unsigned v) {
	return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6 :
			v < 10000000u ? 7 : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}
TXT_DEV void put_num(uint8_t *p, unsigned v, const int nd) {
	for (int k = nd - 1; k >= 0; --k) { p[k] = (uint8_t) ('0' + v % 10u); v /= 10u; }
}

/* One tile.  WRITE = false: lengths and fields only.  cig / md: where the strings go (WRITE). */
template <bool WRITE>
TXT_DEV void text_tile(const int lane, const TileOut o, const int32_t *ops, const uint8_t *ref, const int ref_len,
		const int ext_qstart, const int ext_qend, uint8_t *cig, uint8_t *md, TextRec &out) {
	const int n = o.n_ops;
	const uint8_t *rs = ref + o.ref_position;            /* convertCigar sees refSeq + ref_position */
	const int qstart = o.qstart + ext_qstart, qend = o.qend + ext_qend;
	int cg_at = 0, md_at = 0, n_pieces = 0;
	if (qstart > 0) {
		const int nd = ndigits((unsigned) qstart);
		if (WRITE && lane == 0) { put_num(cig, (unsigned) qstart, nd); cig[nd] = 'S'; }
		cg_at = nd + 1;
		n_pieces = 1;
	}
	int m_carry = 0;          /* EQ/X run still open at the end of the previous step */
	bool m_open = false;
	int mc_carry = 0;         /* EQ bases since the last X / D op */
	int ref_base = 0, read_base = 0;      /* reference / read bases consumed by earlier steps */
	int matches = 0, columns = 0, m_and_i = 0, nm_count = 0;
	for (int c0 = 0; c0 < n; c0 += 64) {
		const int k = c0 + lane;
		const bool valid = k < n;
		const unsigned w = valid ? (unsigned) ops[k] : 0u;
		const unsigned wn = (k + 1 < n) ? (unsigned) ops[k + 1] : 0u;
		const int len = (int) (w >> 4), type = (int) (w & 15u);
		const int tn = (int) (wn & 15u);
		const bool isEQ = valid && type == 7, isX = valid && type == 8, isI = valid && type == 1, isD = valid && type == 2;
		const bool isM = isEQ || isX;
		const bool nextM = (tn == 7 || tn == 8);
		/* ---- CIGAR: one piece per maximal EQ/X run, one per I / D */
		const bool endsM = isM && !nextM;
		const int S = wave_scan(isM ? len : 0, lane);
		const int prevM_i = __shfl_up((int) isM, 1, 64);
		const bool prevM = lane > 0 ? prevM_i != 0 : m_open;
		const u64 startmask = __builtin_amdgcn_ballot_w64(isM && !prevM);
		const u64 sm = startmask & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
		const int s_lane = sm ? 63 - __builtin_clzll(sm) : 0;
		const int S_excl = S - (isM ? len : 0);
		const int S_at_start = __shfl(S_excl, s_lane, 64);
		const int runsum = S - (sm ? S_at_start : -m_carry);
		const bool piece = endsM || isI || isD;
		const unsigned pval = (unsigned) (endsM ? runsum : len);
		const int pch = piece ? ndigits(pval) + 1 : 0;
		const int P = wave_scan(pch, lane);
		if (WRITE && piece) {
			uint8_t *p = cig + cg_at + P - pch;
			put_num(p, pval, pch - 1);
			p[pch - 1] = endsM ? 'M' : (isI ? 'I' : 'D');
		}
		cg_at += __builtin_amdgcn_readlane(P, 63);
		n_pieces += __builtin_popcountll(__builtin_amdgcn_ballot_w64(piece));
		{
			/* only the step's last valid op can leave a run open (any other EQ/X op that does not end a
			 * run is followed by one in the next lane) */
			const int lastl = (n - c0 > 64) ? 63 : (n - c0 - 1);
			const int open_i = (isM && !endsM) ? 1 : 0;
			m_open = __shfl(open_i, lastl, 64) != 0;
			const int rs_last = __shfl(runsum, lastl, 64);
			m_carry = m_open ? rs_last : 0;
		}
		/* ---- positions in front of every op */
		const bool refc = isEQ || isX || isD, readc = isEQ || isX || isI;
		const int RC = wave_scan(refc ? len : 0, lane), RD = wave_scan(readc ? len : 0, lane);
		const int pr0 = ref_base + RC - (refc ? len : 0);            /* reference bases consumed before this op */
		const int pq0 = o.qstart + read_base + RD - (readc ? len : 0);
		/* ---- MD: X and D ops are events; the counter in front of one = EQ bases since the previous event */
		const bool evt = isX || isD;
		const int E = wave_scan(isEQ ? len : 0, lane);
		const u64 evmask = __builtin_amdgcn_ballot_w64(evt);
		const u64 em = evmask & ((1ull << lane) - 1ull);
		const int p_lane = em ? 63 - __builtin_clzll(em) : 0;
		const int E_at_p = __shfl(E, p_lane, 64);
		const int mc = em ? E - E_at_p : E + mc_carry;
		const int ech = evt ? ndigits((unsigned) mc) + 1 + (isX ? 2 * (len - 1) : len) : 0;
		const int Q = wave_scan(ech, lane);
		if (WRITE && evt) {
			uint8_t *p = md + md_at + Q - ech;
			const int nd = ndigits((unsigned) mc);
			put_num(p, (unsigned) mc, nd);
			p += nd;
			if (isX) {
				*p++ = rs[pr0];
				for (int j = 1; j < len; ++j) { *p++ = '0'; *p++ = rs[pr0 + j]; }
			} else {
				*p++ = '^';
				for (int j = 0; j < len; ++j) *p++ = rs[pr0 + j];
			}
		}
		md_at += __builtin_amdgcn_readlane(Q, 63);
		{
			const int E63 = __builtin_amdgcn_readlane(E, 63);
			if (evmask) {
				const int pl = 63 - __builtin_clzll(evmask);
				mc_carry = E63 - __shfl(E, pl, 64);
			} else {
				mc_carry += E63;
			}
		}
		/* ---- NM-profile entries the host form would write: columns with both positions past 16 (:76-98) */
		int cnt = 0;
		if (isEQ || isX) {
			const int mn = pr0 < pq0 ? pr0 : pq0;
			const int skip = 17 - mn > 0 ? 17 - mn : 0;
			cnt = len - skip > 0 ? len - skip : 0;
		} else if (isD && pq0 > 16) {
			const int skip = 17 - pr0 > 0 ? 17 - pr0 : 0;
			cnt = len - skip > 0 ? len - skip : 0;
		}
		nm_count += wave_sum(cnt);
		matches += wave_sum(isEQ ? len : 0);
		columns += wave_sum(valid ? len : 0);
		m_and_i += wave_sum((isM || isI) ? len : 0);
		ref_base += __builtin_amdgcn_readlane(RC, 63);
		read_base += __builtin_amdgcn_readlane(RD, 63);
	}
	/* trailing MD counter, trailing clip */
	{
		const int nd = ndigits((unsigned) mc_carry);
		if (WRITE && lane == 0) put_num(md + md_at, (unsigned) mc_carry, nd);
		md_at += nd;
	}
	if (qend > 0) {
		const int nd = ndigits((unsigned) qend);
		if (WRITE && lane == 0) { put_num(cig + cg_at, (unsigned) qend, nd); cig[cg_at + nd] = 'S'; }
		cg_at += nd + 1;
		n_pieces += 1;
	}
	if (WRITE && lane == 0) { cig[cg_at] = 0; md[md_at] = 0; }

	/* N-clip flags, src/ConvexAlignFast.cpp:493-528 as the host form reproduces them (both set bit 0x1,
	 * both look for 'X') */
	int sv = 0;
	{
		int lo = o.ref_position - 100;
		if (lo < 0) lo = 0;
		const int probes = o.ref_position - lo;              /* k = ref_position, ..., lo + 1 */
		int hit = 0;
		for (int q = lane; q < probes; q += 64) { const int kk = o.ref_position - q; if (kk < ref_len && ref[kk] == 'X') hit++; }
		hit = wave_sum(hit);
		if ((float) hit > (float) probes * 0.8f) sv |= 1;
		int hi = ref_base + 100;
		if (hi > ref_len - o.ref_position) hi = ref_len - o.ref_position;
		const int probes2 = hi - ref_base > 0 ? hi - ref_base : 0;
		int hit2 = 0;
		for (int q = lane; q < probes2; q += 64) if (rs[ref_base + q] == 'X') hit2++;
		hit2 = wave_sum(hit2);
		if ((float) hit2 > (float) probes2 * 0.8f) sv |= 1;
	}
	out.ret = qstart + m_and_i + qend;
	out.score = o.score;
	out.position_offset = o.ref_position;
	out.qstart = qstart;
	out.qend = qend;
	out.nm = columns - matches;
	out.identity = __fdiv_rn((float) matches * 1.0f, (float) columns);
	out.alignment_length = columns;
	out.cigar_op_count = n_pieces;
	out.sv_type = sv;
	out.first_ref = 0;
	out.first_read = o.qstart;
	out.last_ref = ref_base;
	out.last_read = o.qstart + read_base;
	out.nm_count = nm_count;
	out.cigar_len = cg_at;
	out.md_len = md_at;
}

template <bool WRITE>
__global__ void __launch_bounds__(64)
text_kernel(const TextArgs a) {
	const int t = blockIdx.x;
	if (t >= a.n_tiles) return;
	const int lane = threadIdx.x;
	const TileOut o = a.tout[t];
	TextRec rec;
	if (o.status != 0) {
		/* no valid alignment: SingleAlign returns -1 with Score -1.0f, both strings empty */
		rec.ret = -1; rec.score = -1.0f; rec.position_offset = 0; rec.qstart = 0; rec.qend = 0; rec.nm = 0; rec.identity = 0.0f;
		rec.alignment_length = 0; rec.cigar_op_count = 0; rec.sv_type = 0; rec.first_ref = 0; rec.first_read = 0;
		rec.last_ref = 0; rec.last_read = 0; rec.nm_count = 0; rec.cigar_len = 0; rec.md_len = 0;
		if (WRITE && lane == 0) { a.text[a.text_off[t]] = 0; a.text[a.text_off[t] + 1] = 0; }
	} else {
		const TileIn ti = a.tin[t];
		const TileRun tr = a.trun[t];
		uint8_t *cig = nullptr, *md = nullptr;
		if (WRITE) {
			cig = a.text + a.text_off[t];
			md = cig + a.recs[t].cigar_len + 1;
		}
		text_tile<WRITE>(lane, o, a.ops + tr.ops_off + o.ops_first, a.seq + ti.ref_off, ti.W,
				a.ext_qstart ? a.ext_qstart[t] : 0, a.ext_qend ? a.ext_qend[t] : 0, cig, md, rec);
	}
	if (lane == 0) {
		if (!WRITE) { a.recs[t] = rec; a.text_len[t] = (unsigned long long) (rec.cigar_len + 1 + rec.md_len + 1); }
	}
}

/* ------------------------------------------------------------------ nmPerPosition
 *
 * src/ConvexAlignFast.cpp:76-98,186-269 as the host form restates it (cvx_format.cpp): a 32-bit shift register over
 * the alignment columns (every op base is a column) with a bit for every mismatched base and for the first base of
 * every I / D op; an entry (refPosition - 16, readPosition - 16, Yi) per EQ / X / D column once both positions passed
 * 16.  Yi of an EQ / X column is the register's population count; the first base of a gap op makes it Yi + 1 (not a
 * recount), its other bases leave it alone.
 *
 * One wave per tile, one op per lane.  No register travels along the alignment: the number of set bits among the 32
 * columns in front of an op comes from walking back over at most 32 ops, and while a lane emits its op's columns a
 * second cursor follows 32 columns behind and takes the bits out again.  Entry offsets are a wave scan over the
 * per-op entry counts (the same counts text_kernel adds up into nm_count).
 */
struct OpCursor {          /* the op that contains a given column */
	int k, start, len, type;
};

/* set bits among the columns [end - 32, end) where op j ends at column `end`; cur = the op holding column end - 32
 * (op 0 when that column does not exist) */
TXT_DEV int window_bits(const int32_t *ops, int j, const int end, OpCursor &cur) {
	const int lo = end - 32;
	int bits = 0, e = end;
	cur.k = 0; cur.start = 0; cur.len = 0; cur.type = 0;
	bool have = false;
	while (j >= 0 && e > lo) {
		const unsigned w = (unsigned) ops[j];
		const int l = (int) (w >> 4), t = (int) (w & 15u), s = e - l;
		if (t == 8) bits += e - (s > lo ? s : lo);
		else if (t == 1 || t == 2) bits += (s >= lo) ? 1 : 0;
		cur.k = j; cur.start = s; cur.len = l; cur.type = t;
		have = true;
		e = s;
		--j;
	}
	if (!have) {       /* nothing in front: the cursor starts at op 0 */
		const unsigned w = (unsigned) ops[0];
		cur.len = (int) (w >> 4); cur.type = (int) (w & 15u);
	}
	return bits;
}

TXT_DEV void nm_tile(const int lane, const TileOut o, const int32_t *ops, int32_t *out) {
	const int n = o.n_ops;
	int ref_base = 0, read_base = 0, col_base = 0, ent_base = 0;
	for (int c0 = 0; c0 < n; c0 += 64) {
		const int k = c0 + lane;
		const bool valid = k < n;
		const unsigned w = valid ? (unsigned) ops[k] : 0u;
		const int len = (int) (w >> 4), type = (int) (w & 15u);
		const bool isEQ = valid && type == 7, isX = valid && type == 8, isI = valid && type == 1, isD = valid && type == 2;
		const bool refc = isEQ || isX || isD, readc = isEQ || isX || isI;
		const int RC = wave_scan(refc ? len : 0, lane), RD = wave_scan(readc ? len : 0, lane), CS = wave_scan(valid ? len : 0, lane);
		const int pr0 = ref_base + RC - (refc ? len : 0);
		const int pq0 = o.qstart + read_base + RD - (readc ? len : 0);
		const int cs = col_base + CS - (valid ? len : 0);            /* alignment columns in front of this op */
		int cnt = 0, skip = 0;
		if (isEQ || isX) {
			const int mn = pr0 < pq0 ? pr0 : pq0;
			skip = 17 - mn > 0 ? 17 - mn : 0;
			cnt = len - skip > 0 ? len - skip : 0;
		} else if (isD && pq0 > 16) {
			skip = 17 - pr0 > 0 ? 17 - pr0 : 0;
			cnt = len - skip > 0 ? len - skip : 0;
		}
		const int EN = wave_scan(cnt, lane);
		int32_t *dst = out + 3 * (size_t) (ent_base + EN - cnt);
		if (cnt > 0) {
			OpCursor cur;
			if (isD) {
				/* Yi in front of the gap ops that end with this one, plus one per gap op */
				int j = k - 1, g = 1, end = cs;
				while (j >= 0) {
					const unsigned wj = (unsigned) ops[j];
					const int tj = (int) (wj & 15u);
					if (tj != 1 && tj != 2) break;
					end -= (int) (wj >> 4);
					++g;
					--j;
				}
				const int yi = (j >= 0 ? window_bits(ops, j, end, cur) : 0) + g;
				for (int i = skip; i < len; ++i) {
					dst[0] = pr0 + i - 16; dst[1] = pq0 - 16; dst[2] = yi;
					dst += 3;
				}
			} else {
				int P = window_bits(ops, k - 1, cs, cur);
				const int add = isX ? 1 : 0;
				for (int i = 0; i < len; ++i) {
					const int lc = cs + i - 32;          /* the column that leaves the register */
					P += add;
					if (lc >= 0) {
						while (lc >= cur.start + cur.len) {
							cur.start += cur.len;
							cur.k += 1;
							const unsigned wc = (unsigned) ops[cur.k];
							cur.len = (int) (wc >> 4); cur.type = (int) (wc & 15u);
						}
						P -= (cur.type == 8 || ((cur.type == 1 || cur.type == 2) && lc == cur.start)) ? 1 : 0;
					}
					if (i >= skip) {
						dst[0] = pr0 + i - 16; dst[1] = pq0 + i - 16; dst[2] = P;
						dst += 3;
					}
				}
			}
		}
		ent_base += __builtin_amdgcn_readlane(EN, 63);
		ref_base += __builtin_amdgcn_readlane(RC, 63);
		read_base += __builtin_amdgcn_readlane(RD, 63);
		col_base += __builtin_amdgcn_readlane(CS, 63);
	}
}

/* entry counts of the tiles [first, first + count) as 64-bit lengths for text_scan_kernel */
__global__ void __launch_bounds__(256)
nm_count_kernel(const TextRec *recs, unsigned long long *len, int first, int count) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i < count) len[i] = (unsigned long long) (recs[first + i].nm_count > 0 ? recs[first + i].nm_count : 0);
}

__global__ void __launch_bounds__(64)
nm_profile_kernel(const TextArgs a, const unsigned long long *ent_off, int32_t *triples, int first) {
	const int t = first + (int) blockIdx.x;
	const TileOut o = a.tout[t];
	if (o.status != 0) return;
	nm_tile((int) threadIdx.x, o, a.ops + a.trun[t].ops_off + o.ops_first, triples + 3 * (size_t) ent_off[blockIdx.x]);
}

/* exclusive prefix sum of the per-tile text lengths (one workgroup, a chunk of tiles per thread) */
__global__ void __launch_bounds__(256)
text_scan_kernel(const unsigned long long *len, unsigned long long *off, unsigned long long *total, int n) {
	constexpr int T = 256;
	__shared__ unsigned long long s_part[T];
	const int tid = threadIdx.x;
	const int per = (n + T - 1) / T;
	const int t0 = min(n, tid * per), t1 = min(n, t0 + per);
	unsigned long long mine = 0;
	for (int t = t0; t < t1; ++t) mine += len[t];
	s_part[tid] = mine;
	__syncthreads();
	for (int d = 1; d < T; d <<= 1) {
		const unsigned long long v = (tid >= d) ? s_part[tid - d] : 0ull;
		__syncthreads();
		s_part[tid] += v;
		__syncthreads();
	}
	unsigned long long at = s_part[tid] - mine;
	for (int t = t0; t < t1; ++t) { off[t] = at; at += len[t]; }
	if (tid == T - 1) *total = s_part[T - 1];
}

hipError_t launch_text_size(const TextArgs &a, hipStream_t st) {
	if (a.n_tiles <= 0) return hipSuccess;
	hipLaunchKernelGGL(text_kernel<false>, dim3(a.n_tiles), dim3(64), 0, st, a);
	hipLaunchKernelGGL(text_scan_kernel, dim3(1), dim3(256), 0, st, a.text_len, a.text_off, a.text_total, a.n_tiles);
	return hipGetLastError();
}

hipError_t launch_nm_offsets(const TextArgs &a, int first, int count, unsigned long long *len, unsigned long long *off, unsigned long long *total, hipStream_t st) {
	if (count <= 0) return hipSuccess;
	hipLaunchKernelGGL(nm_count_kernel, dim3((count + 255) / 256), dim3(256), 0, st, a.recs, len, first, count);
	hipLaunchKernelGGL(text_scan_kernel, dim3(1), dim3(256), 0, st, len, off, total, count);
	return hipGetLastError();
}

hipError_t launch_nm_profile(const TextArgs &a, int first, int count, const unsigned long long *off, int32_t *triples, hipStream_t st) {
	if (count <= 0) return hipSuccess;
	hipLaunchKernelGGL(nm_profile_kernel, dim3(count), dim3(64), 0, st, a, off, triples, first);
	return hipGetLastError();
}

hipError_t launch_text_write(const TextArgs &a, hipStream_t st) {
	if (a.n_tiles <= 0) return hipSuccess;
	hipLaunchKernelGGL(text_kernel<true>, dim3(a.n_tiles), dim3(64), 0, st, a);
	return hipGetLastError();
}

}  // namespace cvx
