"""GPU (-m gpu): behaviour of the C ABI itself -- statuses, capacities, staged reuse."""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def test_statuses(hip_aligner, port_oracle):
    from ngmlr_amd import synth
    rng = np.random.default_rng(8)
    ok = synth.make_tile(rng, 500, corridor="anchors")
    # best cell in row 0 -> invalid (reference src/ConvexAlignFast.cpp:338)
    row0 = synth.Tile(b"ACGTACGTAC", b"A" + b"T" * 0, *synth.corridor_linear(1, 30), tag="row0")
    # decreasing offsets: not a shape any reference caller builds -> catch-all kernel, still exact
    H = 120
    weird = synth.Tile(synth.random_ref(rng, 400).tobytes(), synth.random_ref(rng, H).tobytes(),
                       (300 - 2 * np.arange(H)).astype(np.int32), np.full(H, 60, np.int32), tag="decreasing")
    empty = synth.Tile(b"ACGT" * 10, b"ACGT" * 5, np.full(20, 100, np.int32), np.full(20, 30, np.int32), tag="outside")
    got = hip_aligner.batch_align([ok, row0, weird, empty])
    assert got[0]["status"] == 0 and got[0]["ret"] == ok.H
    assert got[1]["status"] == 1 and got[1]["ret"] == -1 and got[1]["score"] == -1.0
    from oracle.pyoracle import same_alignment
    assert got[2]["status"] != -1 and same_alignment(port_oracle.align(weird), got[2]) is None
    assert got[3]["status"] == 5 and got[3]["ret"] == -1
    assert port_oracle.align(row0)["ret"] == -1 and port_oracle.align(empty)["ret"] == -1


def test_too_large_matrix_is_rejected_like_the_reference(built):
    """maxMatrixSizeMB (src/AlignmentMatrixFast.cpp:45,55-57), here lowered to 1 MB."""
    from ngmlr_amd import synth
    from ngmlr_amd.aligner import ConvexAlignHip
    al = ConvexAlignHip(device=0, max_matrix_mb=1)
    rng = np.random.default_rng(1)
    small = synth.make_tile(rng, 1000, corridor="anchors")     # 0.3 MB
    big = synth.make_tile(rng, 8000, corridor="anchors")       # 2.5 MB
    got = al.batch_align([small, big])
    assert got[0]["status"] == 0 and got[1]["status"] == 4 and got[1]["ret"] == -1
    al.close()


def test_capacity_error_and_retry(hip_aligner):
    from ngmlr_amd import capi, synth
    lib = capi.load()
    tiles = synth.workload_ont(8, seed=3, max_len=1200)
    batch = hip_aligner.upload(tiles)
    batch.run()
    total = C.c_uint64()
    capi.check(lib.cvx_batch_ops_total(batch.b, C.byref(total)))
    assert total.value > 8
    res = (capi.CvxResult * 8)()
    small = np.zeros(4, dtype=np.uint32)
    used = C.c_uint64()
    rc = lib.cvx_batch_download(hip_aligner.h, batch.b, res, small.ctypes.data, 4, C.byref(used))
    assert rc == -6 and used.value == total.value
    ops = np.zeros(int(used.value), dtype=np.uint32)
    assert lib.cvx_batch_download(hip_aligner.h, batch.b, res, ops.ctypes.data, len(ops), C.byref(used)) == 0
    assert sum(r.n_ops for r in res) == total.value
    batch.free()


def test_staged_batch_can_be_rerun(hip_aligner):
    from ngmlr_amd import synth
    from oracle.pyoracle import same_alignment
    tiles = synth.workload_ont(16, seed=6, max_len=2500)
    batch = hip_aligner.upload(tiles)
    batch.run()
    a = batch.alignments(want_nm=False)
    batch.results = None
    t = batch.run()
    b = batch.alignments(want_nm=False)
    assert t.n_fill_launches >= 1 and t.fill_ms > 0 and t.cells == sum(x.cells for x in tiles)
    for x, y in zip(a, b):
        assert same_alignment(x, y, keys=("ret", "score_bits", "cigar", "md")) is None
    batch.free()


def test_corridorline_stride_is_accepted(hip_aligner, port_oracle):
    """The C++ shim passes &CorridorLine[0].offset with stride 16 (src/IAlignment.h:29-33)."""
    from ngmlr_amd import capi, synth
    from ngmlr_amd.aligner import format_alignment
    from oracle.pyoracle import same_alignment
    lib = capi.load()
    rng = np.random.default_rng(12)
    t = synth.make_tile(rng, 700, corridor="anchors", scatter=40)
    lines = np.zeros((t.H, 4), dtype=np.int32)       # {int offset; int length; unsigned long offsetInMatrix}
    lines[:, 0] = t.row_offset
    lines[:, 1] = t.row_length
    tile = capi.CvxTile()
    tile.ref, tile.qry = t.ref, t.qry
    tile.row_offset = lines.ctypes.data
    tile.row_length = lines.ctypes.data + 4
    tile.ref_len, tile.qry_len, tile.row_stride_bytes = t.W, t.H, 16
    res = (capi.CvxResult * 1)()
    ops = np.zeros(t.H + t.W + 8, dtype=np.uint32)
    used = C.c_uint64()
    capi.check(lib.cvx_align_batch(hip_aligner.h, 1, C.byref(tile), res, ops.ctypes.data, len(ops), C.byref(used)))
    got = format_alignment(lib, res[0], ops, t)
    assert same_alignment(port_oracle.align(t), got) is None


def test_more_ops_than_the_dense_arena_was_sized_for(built):
    """The dense ops arena is sized for a third of H + W per tile; with a mild mismatch penalty an alignment
    can alternate match / mismatch base by base (one op per base).  The batch summary then reports the
    overflow and the ops are compacted again into a larger arena (cvx_runtime.cpp stage_ops) -- same results."""
    from ngmlr_amd import synth
    from ngmlr_amd.aligner import ConvexAlignHip
    from oracle.pyoracle import Oracle, same_alignment
    sc = dict(match=2.0, mismatch=-1.0, gap_open=-5.0, gap_extend=-5.0, gap_extend_min=-1.0, gap_decay=0.15)
    rng = np.random.default_rng(3)
    comp = np.frombuffer(bytes.maketrans(b"ACGT", b"CATG"), dtype=np.uint8)
    tiles = []
    for k in range(6):
        ref = synth.random_ref(rng, 2500)
        qry = ref.copy()
        qry[1::2] = comp[qry[1::2]]                      # every other base substituted
        off, ln = synth.corridor_anchors(len(qry), len(ref))
        tiles.append(synth.Tile(ref.tobytes(), qry.tobytes(), off, ln, tag="alternating%d" % k))
    al = ConvexAlignHip(device=0, **sc)
    got = al.batch_align(tiles)
    al.close()
    orc = Oracle("port", (sc["match"], sc["mismatch"], sc["gap_open"], sc["gap_extend"], sc["gap_extend_min"], sc["gap_decay"]))
    for t, g in zip(tiles, got):
        w = orc.align(t)
        assert same_alignment(w, g) is None, (t.tag, same_alignment(w, g))
        assert g["ret"] == t.H and g["nm"] > 1000      # ~2500 one-base ops each: 15 000 > the 10 400 the arena was sized for


def test_streaming_jobs_waited_out_of_order_and_released_unwaited(hip_aligner, port_oracle):
    """cvx_submit / cvx_wait / cvx_job_release: several jobs in flight on one handle; waiting for a later job
    first, and releasing one that was never waited for, must neither hang nor disturb the others."""
    from ngmlr_amd import capi
    from ngmlr_amd.aligner import format_alignment
    from oracle.pyoracle import same_alignment
    sets = [util.tile_zoo(seed=200 + k, n=20, max_w=1500) for k in range(4)]
    jobs = [hip_aligner.submit(ts) for ts in sets]

    def check(job, tiles):
        res, ops = job.wait()
        for i, t in enumerate(tiles):
            r = capi.CvxResult.from_buffer_copy(res[i].tobytes())
            assert same_alignment(port_oracle.align(t), format_alignment(hip_aligner.lib, r, ops, t)) is None, t.tag
    check(jobs[2], sets[2])
    check(jobs[0], sets[0])
    jobs[1].release()                      # never waited for
    check(jobs[3], sets[3])
    jobs[2].wait()                         # waiting twice is harmless
    for j in (jobs[0], jobs[2], jobs[3]):
        j.release()
    # the handle is still good
    got = hip_aligner.batch_align(sets[1][:5])
    for t, g in zip(sets[1][:5], got):
        assert same_alignment(port_oracle.align(t), g) is None


def test_a_failed_job_keeps_its_own_error(built):
    """a failure while a job's kernels are being queued belongs to that job -- its cvx_wait returns the
    error (every time), its handle stays valid until cvx_job_release, and the jobs around it deliver their own,
    correct results.  CVX_TUNE_FAIL_COMPUTE=2 makes the second compute stage of a handle fail before anything is queued."""
    import ctypes as C
    import os
    import numpy as np
    from ngmlr_amd import capi, synth
    from ngmlr_amd.aligner import ConvexAlignHip
    rng = np.random.default_rng(11)
    tiles = [synth.make_tile(rng, 900, corridor="anchors", scatter=10.0) for _ in range(6)]
    ref = ConvexAlignHip(device=0)
    want = ref.batch_align(tiles, want_nm=False)
    ref.close()
    os.environ["CVX_TUNE_FAIL_COMPUTE"] = "2"
    try:
        al = ConvexAlignHip(device=0)
    finally:
        del os.environ["CVX_TUNE_FAIL_COMPUTE"]
    jobs = [al.submit(tiles) for _ in range(3)]
    res0, _ = jobs[0].wait()
    for _ in range(2):                                   # the failed job answers with its own error, twice, and stays valid
        with pytest.raises(capi.CvxError) as e:
            jobs[1].wait()
        assert e.value.code == -4 and "CVX_TUNE_FAIL_COMPUTE" in str(e.value)
    res2, _ = jobs[2].wait()
    for res in (res0, res2):
        for i, w in enumerate(want):
            assert int(res[i]["status"]) == w["status"]
            assert int(np.float32(res[i]["score"]).view(np.uint32)) == w["fwd_score_bits"]
    # a polled job: done flips to 1 without blocking in cvx_wait
    j3 = al.submit(tiles)
    done = C.c_int32(0)
    for _ in range(20000):
        capi.check(al.lib.cvx_job_poll(al.h, j3.j, C.byref(done)))
        if done.value:
            break
    assert done.value == 1
    r3, _ = j3.wait()
    assert int(r3[0]["status"]) == want[0]["status"]
    for j in jobs + [j3]:
        j.release()
    # destroy with a job the caller never released: nothing leaks, nothing crashes
    j4 = al.submit(tiles)
    al.close()
    j4.j = None
