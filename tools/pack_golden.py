"""tools/pack_golden.py -- turn the recorder dumps of tools/make_golden.sh into the small
fixtures under tests/golden/ (one .npz per reference test, tiles kept whole)."""
import os
import struct
import sys

import numpy as np


def read_records(path):
    data = open(path, 'rb').read()
    pos = 0
    recs = []
    while pos < len(data):
        magic, W, H, height, eqs, eqe = struct.unpack_from('<6i', data, pos); pos += 24
        assert magic == 0x43565854
        ref = data[pos:pos + W]; pos += W
        qry = data[pos:pos + H]; pos += H
        off = np.frombuffer(data, dtype='<i4', count=height, offset=pos).copy(); pos += 4 * height
        ln = np.frombuffer(data, dtype='<i4', count=height, offset=pos).copy(); pos += 4 * height
        ret, = struct.unpack_from('<i', data, pos); pos += 4
        score_bits, = struct.unpack_from('<I', data, pos); pos += 4
        a = struct.unpack_from('<11i', data, pos); pos += 44
        ident_bits, = struct.unpack_from('<I', data, pos); pos += 4
        cl, ml = struct.unpack_from('<2i', data, pos); pos += 8
        cigar = data[pos:pos + cl]; pos += cl
        md = data[pos:pos + ml]; pos += ml
        n, = struct.unpack_from('<i', data, pos); pos += 4
        nm = np.frombuffer(data, dtype='<i4', count=3 * n, offset=pos).copy().reshape(n, 3); pos += 12 * n
        recs.append(dict(ref=ref, qry=qry, off=off, len=ln, height=height, eqs=eqs, eqe=eqe, ret=ret,
                         score_bits=score_bits, fields=np.array(a, dtype=np.int32), ident_bits=ident_bits,
                         cigar=cigar, md=md, nm=nm))
    return recs


def pack(recs, out, max_tiles=None, max_cells=None):
    keep = []
    cells = 0
    for r in recs:
        c = int(r['len'].astype(np.int64).sum())
        if max_cells is not None and c > max_cells:
            continue
        keep.append(r)
        cells += c
        if max_tiles is not None and len(keep) >= max_tiles:
            break
    d = {'n': np.int32(len(keep))}
    for i, r in enumerate(keep):
        p = 't%d_' % i
        d[p + 'ref'] = np.frombuffer(r['ref'], dtype=np.uint8)
        d[p + 'qry'] = np.frombuffer(r['qry'], dtype=np.uint8)
        d[p + 'off'] = r['off']
        d[p + 'len'] = r['len']
        d[p + 'meta'] = np.array([r['eqs'], r['eqe'], r['ret']], dtype=np.int32)
        d[p + 'bits'] = np.array([r['score_bits'], r['ident_bits']], dtype=np.uint32)
        d[p + 'fields'] = r['fields']
        d[p + 'cigar'] = np.frombuffer(r['cigar'], dtype=np.uint8)
        d[p + 'md'] = np.frombuffer(r['md'], dtype=np.uint8)
        d[p + 'nm'] = r['nm']
    np.savez_compressed(out, **d)
    print('%s: %d tiles (%d recorded), %.1f Mcells, %d bytes' % (out, len(keep), len(recs), cells / 1e6, os.path.getsize(out)))


def read_genome(path):
    data = open(path, 'rb').read()
    nib, ns = struct.unpack_from('<2Q', data, 0)
    starts = np.frombuffer(data, dtype='<u8', count=ns, offset=16).copy()
    binref = np.frombuffer(data, dtype=np.uint8, count=nib // 2, offset=16 + 8 * ns).copy()
    return nib, starts, binref


def read_decodes(path):
    data = open(path, 'rb').read()
    pos = 0
    out = []
    while pos < len(data):
        p, ln = struct.unpack_from('<Qi', data, pos); pos += 12
        out.append((p, ln, data[pos:pos + ln])); pos += ln
    return out


def pack_decode(work, name, out, max_windows=None):
    """The reference's encoded genome (binRef nibbles, chromosome start table) and the windows
    DecodeRefSequenceExact produced for its alignments: position, length, the decoded bytes."""
    nib, starts, binref = read_genome(os.path.join(work, name + '.genome'))
    dec = read_decodes(os.path.join(work, name + '.decode'))
    if max_windows is not None and len(dec) > max_windows:
        # keep the interesting ones first: windows holding 'x' (outside a chromosome) or 'N', odd start positions
        key = lambda d: (-(b'x' in d[2][:-1]), -(b'N' in d[2][:-1]), -(d[0] & 1))  # noqa: E731
        dec = sorted(dec, key=key)[:max_windows]
    d = {'nibbles': np.uint64(nib), 'starts': starts, 'binref': binref, 'n': np.int32(len(dec)),
         'pos': np.array([x[0] for x in dec], dtype=np.uint64), 'len': np.array([x[1] for x in dec], dtype=np.int32),
         'off': np.concatenate([[0], np.cumsum([x[1] for x in dec])]).astype(np.int64),
         'bytes': np.frombuffer(b''.join(x[2] for x in dec), dtype=np.uint8)}
    np.savez_compressed(out, **d)
    print('%s: genome %d nibbles, %d chromosome starts, %d decoded windows, %d bytes' % (out, nib, len(starts), len(dec), os.path.getsize(out)))


if __name__ == '__main__':
    work, outdir = sys.argv[1], sys.argv[2]
    os.makedirs(outdir, exist_ok=True)
    pack(read_records(os.path.join(work, 'test_2.rec')), os.path.join(outdir, 'ref_test_2.npz'))
    pack(read_records(os.path.join(work, 'test_4.rec')), os.path.join(outdir, 'ref_test_4.npz'))
    # test_3: ~985 calls; keep the first 60 below 2.5 Mcells (fixture size)
    pack(read_records(os.path.join(work, 'test_3.rec')), os.path.join(outdir, 'ref_test_3.npz'), max_tiles=60, max_cells=2500000)
    # genome encoding + decoded windows: all of test_2 / test_4, a sample of test_3
    pack_decode(work, 'test_2', os.path.join(outdir, 'decode_test_2.npz'))
    pack_decode(work, 'test_4', os.path.join(outdir, 'decode_test_4.npz'))
    pack_decode(work, 'test_3', os.path.join(outdir, 'decode_test_3.npz'), max_windows=40)
