"""CPU: the byte-table form of decode_windows_kernel (ngmlr_amd/csrc/cvx_genome.hip, late round 6) checked without a device.  The
kernel expands sixteen 4-bit codes to characters with v_perm_b32 used as an eight-entry byte table (`lut4`), a second permute
whose selectors 12 / 13 read 0x00 / 0xFF to mask the codes the table does not hold, and two permutes that interleave the
characters of the even and odd nibbles (`expand8`).  This test takes the selector and table constants OUT OF THE SOURCE, runs
them through a bit-level model of v_perm_b32 (D.byte[i] = {S0,S1}.byte[sel.byte[i]] for selectors 0-7, 0x00 for 12, 0xFF for
13-15) and compares with dec4 of the reference (src/SequenceProvider.cpp:90-104: A T G C N; anything above 4 is not a code the
encoder produces, the kernel writes '?') -- for every byte value in every position, both nibble parities of a piece's start."""
import os
import re
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "ngmlr_amd", "csrc", "cvx_genome.hip")).read()


def perm(s0, s1, sel):
    pool = [(s1 >> (8 * i)) & 0xFF for i in range(4)] + [(s0 >> (8 * i)) & 0xFF for i in range(4)]
    out = 0
    for b in range(4):
        k = (sel >> (8 * b)) & 0xFF
        if k <= 7:
            v = pool[k]
        elif k == 12:
            v = 0x00
        elif k >= 13:
            v = 0xFF
        else:
            raise AssertionError("selector %d (sign replication) is not one the kernel may use" % k)
        out |= v << (8 * b)
    return out


def constants():
    m = re.search(r"__builtin_amdgcn_perm\((0x[0-9A-Fa-f]+)u, (0x[0-9A-Fa-f]+)u, x & (0x[0-9A-Fa-f]+)u\)", SRC)
    k = re.search(r"__builtin_amdgcn_perm\(0u, 0u, \(\(x >> 3\) & (0x[0-9A-Fa-f]+)u\) \| (0x[0-9A-Fa-f]+)u\)", SRC)
    q = re.search(r"\(c & ~m\) \| \((0x[0-9A-Fa-f]+)u & m\)", SRC)
    e = re.findall(r"__builtin_amdgcn_perm\(ch, cl, (0x[0-9A-Fa-f]+)u\)", SRC)
    assert m and k and q and len(e) == 2, "lut4 / expand8 not found in cvx_genome.hip as this test knows them"
    return [int(x, 16) for x in m.groups()], [int(x, 16) for x in k.groups()], int(q.group(1), 16), [int(x, 16) for x in e]


def lut4(x, c):
    (t_hi, t_lo, keep), (bit, base), fill, _ = c
    ch = perm(t_hi, t_lo, x & keep)
    m = perm(0, 0, ((x >> 3) & bit) | base)
    return (ch & ~m & 0xFFFFFFFF) | (fill & m)


def expand8(d, c):
    ch, cl = lut4((d >> 4) & 0x0F0F0F0F, c), lut4(d & 0x0F0F0F0F, c)
    return perm(ch, cl, c[3][0]), perm(ch, cl, c[3][1])


def piece(src9, odd, c):
    """the kernel's fast path for one piece: eight genome bytes (+ the ninth when the piece starts on a low nibble) -> 16 characters"""
    nb = int.from_bytes(src9[:8], "big")                      # __builtin_bswap64 of the little-endian 8-byte load
    if odd:
        nb = ((nb << 4) & 0xFFFFFFFFFFFFFFFF) | (src9[8] >> 4)
    w0, w1 = expand8(nb >> 32, c)
    w2, w3 = expand8(nb & 0xFFFFFFFF, c)
    return struct.pack("<4I", w0, w1, w2, w3)


def dec4(n):
    return b"ATGCN"[n] if n < 5 else ord("?")


def want(src9, odd):
    nibs = []
    for b in src9:
        nibs += [b >> 4, b & 15]
    return bytes(dec4(n) for n in nibs[odd:odd + 16])


def test_every_byte_value_in_every_position():
    c = constants()
    for pos in range(9):
        for v in range(256):
            src = bytearray(b"\x01\x23\x40\x12\x34\x02\x31\x44\x20")
            src[pos] = v
            for odd in (0, 1):
                assert piece(bytes(src), odd, c) == want(bytes(src), odd), (pos, v, odd)


def test_random_genome_bytes_valid_and_not():
    c = constants()
    rng = np.random.default_rng(5)
    valid = np.array([a << 4 | b for a in range(5) for b in range(5)], dtype=np.uint8)
    for it in range(4000):
        src = bytes(rng.choice(valid, 9)) if it % 2 else bytes(rng.integers(0, 256, 9, dtype=np.uint8))
        for odd in (0, 1):
            assert piece(src, odd, c) == want(src, odd), (src.hex(), odd)


def test_the_table_is_dec4():
    (t_hi, t_lo, keep), (bit, base), fill, _ = constants()
    table = struct.pack("<II", t_lo, t_hi)
    assert table == b"ATGCN???" and keep == 0x07070707 and bit == 0x01010101 and base == 0x0C0C0C0C and fill == 0x3F3F3F3F
