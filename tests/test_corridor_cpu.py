"""CPU: the closed forms of the reference's corridor builders (cvx_tile.corridor_kind, restated in
ngmlr_amd/synth.py for the generators and in cvx_types.h for the device / host planning) reproduce every corridor
the unmodified reference was recorded building -- the committed golden tiles of test_2 / test_3 / test_4 and, when
generated, all 985 SingleAlign calls of test_3 -- bit for bit; tests/util.fit_corridor recovers the builder and its
parameters from the recorded rows.  (The device's own evaluation is checked in tests/test_gpu_corridor.py.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from ngmlr_amd import capi, synth
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows_of(desc, H):
    if desc[0] == capi.CORRIDOR_AFFINE:
        return synth.affine_rows(H, desc[1], desc[2], desc[3], desc[5])
    return np.full(H, desc[4], np.int32), np.full(H, desc[5], np.int32)


@pytest.mark.parametrize("name", ["ref_test_2.npz", "ref_test_4.npz", "ref_test_3.npz", "full"])
def test_recorded_corridors_have_a_closed_form(name):
    if name == "full":
        name = util.full_golden_path()
        if name is None:
            pytest.skip("oracle/_ref/golden_full not generated")
    kinds = {}
    for t, _ in util.load_golden(name):
        d = util.fit_corridor(t.row_offset, t.row_length, t.H, t.W)
        assert d is not None, t.tag
        off, ln = _rows_of(d, t.H)
        assert np.array_equal(off, t.row_offset) and np.array_equal(ln, t.row_length), t.tag
        kinds[d[0]] = kinds.get(d[0], 0) + 1
    assert kinds.get(capi.CORRIDOR_AFFINE, 0) > 0


@pytest.mark.parametrize("name", ["ref_test_2.npz", "ref_test_4.npz", "ref_test_3.npz", "full"])
def test_library_recognises_every_recorded_corridor(name):
    """cvx_corridor_fit -- what Convex::ConvexAlignHip::Prepare runs on the CorridorLine[] ngmlr hands it --
    finds a closed form for every corridor the unmodified reference was recorded building (985 / 985 on test_3), and the form
    regenerates the recorded rows bit for bit (numpy float32 here; the device's evaluation: tests/test_gpu_corridor.py)."""
    if name == "full":
        name = util.full_golden_path()
        if name is None:
            pytest.skip("oracle/_ref/golden_full not generated")
    tiles = util.load_golden(name)
    hits = 0
    for t, _ in tiles:
        d = capi.corridor_fit(t.row_offset, t.row_length, t.W, t.H)
        assert d[0] != capi.CORRIDOR_ROWS, t.tag
        off, ln = _rows_of(d, t.H)
        assert np.array_equal(off, t.row_offset) and np.array_equal(ln, t.row_length), (t.tag, d)
        hits += 1
    assert hits == len(tiles) and hits > 0


def test_library_fit_on_generated_and_irregular_corridors():
    rng = np.random.default_rng(11)
    # every builder of the generators (anchors with scattered `right`, endpoints, linear, full), CorridorLine stride included
    for t in util.tile_zoo(n=64) + synth.workload_short(20) + synth.workload_ultralong_sv(4, read_len=3000):
        d = capi.corridor_fit(t.row_offset, t.row_length, t.W, t.H)
        assert d[0] == t.desc[0], (t.tag, d, t.desc)
        off, ln = _rows_of(d, t.H)
        assert np.array_equal(off, t.row_offset) and np.array_equal(ln, t.row_length), (t.tag, d)
        lines = np.zeros((t.H, 4), dtype=np.int32)      # CorridorLine { int offset; int length; unsigned long offsetInMatrix; }
        lines[:, 0], lines[:, 1] = t.row_offset, t.row_length
        d16 = capi.corridor_fit(lines[:, 0], lines[:, 1], t.W, t.H, stride_bytes=16)
        assert d16 == d
    # anchors corridors at read lengths and shifts of the real workloads: any float inside the feasible interval will do
    for _ in range(300):
        H = int(rng.integers(2, 40000))
        W = max(1, int(H * rng.uniform(0.8, 1.3)))
        k = np.float32(H) * np.float32(1.0) / np.float32(W)
        right = np.float32(rng.uniform(128.0, 4000.0))
        w = int(rng.integers(200, 9000))
        off, ln = synth.affine_rows(H, k, np.float32(0.0), right, w)
        d = capi.corridor_fit(off, ln, W, H)
        assert d[0] == capi.CORRIDOR_AFFINE and d[5] == w, (H, W, right, d)
        o2, _ = _rows_of(d, H)
        assert np.array_equal(o2, off), (H, W, right, d)
    # rows no builder makes: one offset moved, one length changed, a k that is not qry / ref, offsets that jump back
    t = util.tile_zoo(n=4)[0]
    for mutate in ("offset", "length", "k", "reverse"):
        off, ln = t.row_offset.copy(), t.row_length.copy()
        W = t.W
        if mutate == "offset":
            off[t.H // 2] += 1
        elif mutate == "length":
            ln[t.H // 3] += 1
        elif mutate == "k":
            W = t.W + 97
        else:
            off = off[::-1].copy()
        d = capi.corridor_fit(off, ln, W, t.H)
        if mutate == "k" and d[0] != capi.CORRIDOR_ROWS:      # (a different k may still reproduce a short tile's rows exactly: then it is a form)
            o2, l2 = _rows_of(d, t.H)
            assert np.array_equal(o2, off) and np.array_equal(l2, ln)
        else:
            assert d[0] == capi.CORRIDOR_ROWS, (mutate, d)
    assert capi.corridor_fit(np.zeros(0, np.int32), np.zeros(0, np.int32), 10, 0)[0] == capi.CORRIDOR_ROWS


def test_library_fit_scalar_and_avx2_forms_agree(tmp_path):
    """cvx_corridor_fit picks an AVX2 form of its two passes where the host has it; the scalar form (CVX_CORRIDOR_NO_AVX2=1, read when
    the library is loaded) must return the same closed forms."""
    code = (
        "import numpy as np, json, sys\n"
        "sys.path.insert(0, %r)\n"
        "from ngmlr_amd import capi, synth\n"
        "rng = np.random.default_rng(3)\n"
        "out = []\n"
        "for _ in range(120):\n"
        "    H = int(rng.integers(2, 30000)); W = max(1, int(H * rng.uniform(0.8, 1.3)))\n"
        "    k = np.float32(H) * np.float32(1.0) / np.float32(W)\n"
        "    off, ln = synth.affine_rows(H, k, np.float32(0.0), np.float32(rng.uniform(128.0, 3000.0)), int(rng.integers(200, 5000)))\n"
        "    if rng.random() < 0.15: off[int(rng.integers(0, H))] += 1\n"
        "    d = capi.corridor_fit(off, ln, W, H)\n"
        "    o2 = synth.affine_rows(H, np.float32(d[1]), np.float32(d[2]), np.float32(d[3]), d[5])[0] if d[0] == 1 else None\n"
        "    out.append([d[0], bool(o2 is None or np.array_equal(o2, off))])\n"
        "print(json.dumps(out))\n") % ROOT
    import json
    res = {}
    for name, env in (("avx2", {}), ("scalar", {"CVX_CORRIDOR_NO_AVX2": "1"})):
        r = subprocess.run(["python", "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env), timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        res[name] = json.loads(r.stdout)
    assert res["avx2"] == res["scalar"]
    assert all(ok for _, ok in res["scalar"]) and sum(1 for k_, _ in res["scalar"] if k_ == 1) >= 90 and any(k_ == 0 for k_, _ in res["scalar"])


def test_generators_carry_their_closed_form():
    for t in util.tile_zoo(n=48) + synth.workload_short(20) + synth.workload_ultralong_sv(4, read_len=3000):
        assert t.desc is not None
        off, ln = _rows_of(t.desc, t.H)
        assert np.array_equal(off, t.row_offset) and np.array_equal(ln, t.row_length), (t.tag, t.desc)
        d = util.fit_corridor(t.row_offset, t.row_length, t.H, t.W)
        assert d is not None
        off, ln = _rows_of(d, t.H)
        assert np.array_equal(off, t.row_offset) and np.array_equal(ln, t.row_length), (t.tag, d)


def test_host_side_closed_form_matches_numpy(tmp_path):
    """affine_row_offset of cvx_types.h compiled for the host (what the chain planning uses) against numpy float32."""
    src = tmp_path / "aff.cpp"
    src.write_text('#include "cvx_types.h"\nextern "C" void rows(int H, float d, float k, float r, int *out) '
                   '{ for (int y = 0; y < H; ++y) out[y] = cvx::affine_row_offset(y, d, k, r); }\n')
    so = tmp_path / "aff.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "ngmlr_amd", "csrc"),
                    str(src), "-o", str(so)], check=True)
    lib = C.CDLL(str(so))
    lib.rows.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
    rng = np.random.default_rng(5)
    for _ in range(200):
        H = int(rng.integers(1, 30000))
        W = int(rng.integers(1, 30000))
        k = np.float32(H) * np.float32(1.0) / np.float32(W)
        d = np.float32(rng.choice([0.0, 154.5, 1024.0]))
        r = np.float32(rng.choice([0.0, 156.16, 312.32, 901.7]))
        out = np.zeros(H, dtype=np.int32)
        lib.rows(H, float(d), float(k), float(r), out.ctypes.data)
        want, _ = synth.affine_rows(H, k, d, r, 1)
        assert np.array_equal(out, want)


def test_tileset_table_layout_and_subset():
    tiles = util.tile_zoo(n=12)
    ts = synth.tileset_from_tiles(tiles)
    assert ts.desc is not None and len(ts) == 12
    tab = ts.table()
    assert tab.dtype.itemsize == C.sizeof(capi.CvxTile) and np.all(tab["corridor_kind"] == 0) and np.all(tab["row_stride_bytes"] == 4)
    ts.use_closed_form()
    tab = ts.table()
    assert np.all(tab["corridor_kind"] > 0) and np.all(tab["row_offset"] == 0)
    sub = ts.subset([5, 2, 9])
    for k, i in enumerate([5, 2, 9]):
        a, b = sub.tile(k), tiles[i]
        assert a.ref == b.ref and a.qry == b.qry and np.array_equal(a.row_offset, b.row_offset) and a.desc[0] == b.desc[0]
        assert abs(a.desc[1] - np.float32(b.desc[1])) == 0 and a.desc[5] == b.desc[5]
