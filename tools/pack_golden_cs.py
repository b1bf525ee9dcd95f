"""tools/pack_golden_cs.py CS_DUMP TABLE_DUMP SAMPLE.npz FULL.npz [STRIDE] -- the recorder dumps of tools/make_golden_cs.sh as fixtures:
the k-mer table in compact form (used prefixes, slot counts, RefTable, unit offset) and the recorded candidate-search calls
(sub-read -> LocationScore list in the reference's order, maxHitNumber, threshold, kCount, table size of the first attempt).  SAMPLE keeps every STRIDE-th sub-read (default 6)."""
import struct
import sys

import numpy as np


def read_table(path):
    d = open(path, 'rb').read()
    k, units, skip = struct.unpack_from('<3I', d, 0)
    pos = 12
    assert units == 1, "fixtures cover one table unit (genomes below 4 Gbp)"
    off, tl, nused = struct.unpack_from('<QII', d, pos); pos += 16
    rec = np.frombuffer(d, dtype=np.dtype([('prefix', '<u4'), ('tab', '<u4'), ('cnt', '<u4'), ('rc', 'i1')]), count=nused, offset=pos)
    pos += nused * 13
    locs = np.frombuffer(d, dtype='<u4', count=tl, offset=pos).copy()
    return dict(k=k, skip=skip, offset=off, prefix=rec['prefix'].copy(), tab=rec['tab'].copy(), cnt=rec['cnt'].copy(), rc=rec['rc'].copy(), locs=locs)


def read_cs(path):
    d = open(path, 'rb').read()
    pos = 0
    out = []
    while pos < len(d):
        n, = struct.unpack_from('<i', d, pos); pos += 4
        seq = d[pos:pos + n]; pos += n
        mh, th, rl, nn, kc, fb = struct.unpack_from('<ffiiii', d, pos); pos += 24
        rec = np.frombuffer(d, dtype=np.dtype([('loc', '<u8'), ('score', '<f4'), ('rev', '<i4')]), count=nn, offset=pos).copy()
        pos += 16 * nn
        out.append((seq, mh, th, rl, rec, kc, fb))
    return out


def pack(table, calls, out):
    seqs = b''.join(c[0] for c in calls)
    lens = np.array([len(c[0]) for c in calls], dtype=np.int32)
    cnt = np.array([len(c[4]) for c in calls], dtype=np.int32)
    recs = np.concatenate([c[4] for c in calls]) if calls else np.zeros(0, dtype=[('loc', '<u8'), ('score', '<f4'), ('rev', '<i4')])
    np.savez_compressed(out, k=np.int32(table['k']), ref_skip=np.int32(table['skip']), unit_offset=np.uint64(table['offset']),
                        prefix=table['prefix'], tab=table['tab'], cnt=table['cnt'], rc=table['rc'], locs=table['locs'],
                        seqs=np.frombuffer(seqs, dtype=np.uint8), seq_len=lens, max_hit=np.array([c[1] for c in calls], dtype=np.float32),
                        thresh=np.array([c[2] for c in calls], dtype=np.float32), rlist_len=np.array([c[3] for c in calls], dtype=np.int32),
                        n_scores=cnt, loc=recs['loc'], score=recs['score'], rev=recs['rev'],
                        kmer_misses=np.array([c[5] for c in calls], dtype=np.int32), first_bits=np.array([c[6] for c in calls], dtype=np.int32))


def pack_big(cs_path, table_path, out):
    """--big: the recorded calls on ngmlr_amd.synth.big_reference + the SHA-256 of the table, after checking that cvx_index_build
    (what the tests will rebuild it with) produces exactly the table the reference dumped"""
    import hashlib
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ngmlr_amd import capi, synth
    table = read_table(table_path)
    calls = read_cs(cs_path)
    contigs = synth.big_reference(512 << 20, n_contigs=8)
    idx5, locs, starts = synth.kmer_table(capi.load(), contigs, k=int(table['k']), skip=int(table['skip']))
    rec = idx5.reshape(-1, 5)
    tab = rec[:, :4].copy().view('<u4').ravel()
    used = rec[:, 4] != 0
    pre = np.flatnonzero(used[:-2]).astype(np.uint32)
    assert np.array_equal(pre, table['prefix']), "used prefixes differ from the reference's table"
    assert np.array_equal(tab[pre], table['tab']) and np.array_equal(tab[pre + 1] - tab[pre], table['cnt']), "table rows differ"
    assert np.array_equal(rec[pre, 4].view('i1'), table['rc']), "weight bytes differ"
    assert np.array_equal(locs, table['locs']), "locations differ"
    seqs = b''.join(c[0] for c in calls)
    recs = np.concatenate([c[4] for c in calls])
    np.savez_compressed(out, k=np.int32(table['k']), ref_skip=np.int32(table['skip']), unit_offset=np.uint64(table['offset']),
                        reference=np.array("ngmlr_amd.synth.big_reference(512 << 20, n_contigs=8)"),
                        index_sha256=np.array(hashlib.sha256(idx5.tobytes()).hexdigest()), locs_sha256=np.array(hashlib.sha256(locs.tobytes()).hexdigest()),
                        n_locations=np.int64(len(locs)), n_used_prefixes=np.int64(len(pre)),
                        seqs=np.frombuffer(seqs, dtype=np.uint8), seq_len=np.array([len(c[0]) for c in calls], dtype=np.int32),
                        max_hit=np.array([c[1] for c in calls], dtype=np.float32), thresh=np.array([c[2] for c in calls], dtype=np.float32),
                        rlist_len=np.array([c[3] for c in calls], dtype=np.int32), n_scores=np.array([len(c[4]) for c in calls], dtype=np.int32),
                        loc=recs['loc'], score=recs['score'], rev=recs['rev'],
                        kmer_misses=np.array([c[5] for c in calls], dtype=np.int32), first_bits=np.array([c[6] for c in calls], dtype=np.int32))
    print("big: k=%d, %d used prefixes, %d locations = the table cvx_index_build makes; %d recorded sub-reads, %d candidates" % (
        table['k'], len(pre), len(locs), len(calls), len(recs)))


if __name__ == '__main__':
    if sys.argv[1] == '--big':
        pack_big(sys.argv[2], sys.argv[3], sys.argv[4])
        sys.exit(0)
    table = read_table(sys.argv[2])
    calls = read_cs(sys.argv[1])
    print("table: k=%d, %d used prefixes, %d locations; %d recorded sub-reads, %d candidates" % (
        table['k'], len(table['prefix']), len(table['locs']), len(calls), sum(len(c[4]) for c in calls)))
    pack(table, calls, sys.argv[4])
    pack(table, calls[::int(sys.argv[5]) if len(sys.argv) > 5 else 6], sys.argv[3])
