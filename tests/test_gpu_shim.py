"""GPU (-m gpu): the C++ IAlignment drop-in (Convex::ConvexAlignHip) driven like
AlignmentBuffer::computeAlignment drives the reference aligner, on tiles whose expected
Align contents come from the CPU oracle and from the recorded reference pipeline."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_records(path, pairs):
    """Same record layout as tools/ref_recorder/recording_aligner.h."""
    with open(path, "wb") as f:
        for t, e in pairs:
            f.write(struct.pack("<6i", 0x43565854, t.W, t.H, t.H, t.ext_qstart, t.ext_qend))
            f.write(t.ref); f.write(t.qry)
            f.write(np.ascontiguousarray(t.row_offset, dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(t.row_length, dtype="<i4").tobytes())
            f.write(struct.pack("<iI", e["ret"], e["score_bits"]))
            if e["ret"] >= 0:
                f.write(struct.pack("<11i", *[e[k] for k in util.FIELD_NAMES]))
                f.write(struct.pack("<I", int(np.float32(e["identity"]).view(np.uint32))))
                c, m = e["cigar"].encode(), e["md"].encode()
                nm = np.ascontiguousarray(e["nm_per_position"], dtype="<i4")
            else:
                f.write(struct.pack("<11i", *([0] * 11)))
                f.write(struct.pack("<I", 0))
                c, m, nm = b"", b"", np.zeros((0, 3), dtype="<i4")
            f.write(struct.pack("<2i", len(c), len(m)))
            f.write(c); f.write(m)
            f.write(struct.pack("<i", len(nm)))
            f.write(nm.tobytes())


@pytest.mark.parametrize("mode", ["single", "batch"])
def test_cpp_shim_matches_oracle(built, port_oracle, tmp_path, mode):
    exe = os.path.join(ROOT, "ngmlr_amd", "shim_test")
    assert os.path.exists(exe), "shim_test not built"
    tiles = util.tile_zoo(seed=77, n=80, max_w=2500) + util.edge_tiles()
    pairs = [(t, port_oracle.align(t)) for t in tiles]
    pairs += util.load_golden("ref_test_2.npz") + util.load_golden("ref_test_4.npz")
    rec = str(tmp_path / "tiles.bin")
    write_records(rec, pairs)
    cmd = [exe, rec] + (["batch"] if mode == "batch" else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr[-2000:]
    assert "0 mismatches" in res.stdout


@pytest.mark.parametrize("fit", ["1", "0"])
def test_shim_recognises_the_recorded_corridors(built, port_oracle, tmp_path, fit):
    """ConvexAlignHip::Prepare recovers the closed form behind every CorridorLine[] the unmodified reference
    was recorded passing to SingleAlign (all 985 calls of test_3 when oracle/_ref/golden_full travelled, else the committed
    sample) and sends 32 bytes instead of the rows; a corridor no builder makes (one row moved, one of another width) still
    travels as rows.  Results identical to the recorded reference output either way, and with the recogniser off."""
    import re
    exe = os.path.join(ROOT, "ngmlr_amd", "shim_test")
    full = util.full_golden_path()
    pairs = util.load_golden(full if full else "ref_test_3.npz")
    n_golden = len(pairs)
    assert n_golden >= 60
    irregular = []
    for t in util.tile_zoo(seed=5, n=6, max_w=1500):
        off, ln = t.row_offset.copy(), t.row_length.copy()
        if len(irregular) % 2 == 0:
            off[t.H // 2] += 1
        else:
            ln[t.H // 3] += 2
        irregular.append(util.synth.Tile(t.ref, t.qry, off, ln, t.ext_qstart, t.ext_qend, tag=t.tag + "+irregular"))
    pairs = pairs + [(t, port_oracle.align(t)) for t in irregular]
    rec = str(tmp_path / "tiles.bin")
    write_records(rec, pairs)
    res = subprocess.run([exe, rec, "batch"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900,
                         env=dict(os.environ, CVX_CORRIDOR_FIT=fit))
    assert res.returncode == 0, res.stdout + res.stderr[-2000:]
    assert "0 mismatches" in res.stdout
    m = re.search(r"(\d+) of (\d+) corridors travelled as closed forms", res.stdout)
    assert m, res.stdout
    assert (int(m.group(1)), int(m.group(2))) == ((n_golden, n_golden + len(irregular)) if fit == "1" else (0, n_golden + len(irregular))), res.stdout


def test_batching_aligner_many_threads(built, port_oracle, tmp_path):
    """24 worker threads call the blocking SingleAlign of one shared
    BatchingAligner; requests coalesce into a few device launches, results stay exact."""
    exe = os.path.join(ROOT, "ngmlr_amd", "batching_test")
    assert os.path.exists(exe), "batching_test not built"
    tiles = util.tile_zoo(seed=91, n=160, max_w=1800)
    pairs = [(t, port_oracle.align(t)) for t in tiles]
    rec = str(tmp_path / "tiles.bin")
    write_records(rec, pairs)
    res = subprocess.run([exe, rec, "24"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr[-2000:]
    assert "0 mismatches" in res.stdout


@pytest.mark.parametrize("mode,threads,env", [("shared", 24, {"CVX_ALIAS_DEVICES": "2"}), ("handles", 2, {}), ("handles", 4, {})])
def test_two_devices_worth_of_handles_on_one_gpu(built, port_oracle, tmp_path, mode, threads, env):
    """The N-device code paths on the one device a test box has: `shared` = ngmlr's form, every
    worker thread constructs its own Convex::SharedAligner and CVX_ALIAS_DEVICES=2 deals them over two logical devices
    (two backends + two dispatcher threads, both on device 0); `handles` = bench.py --gpus N's form, one
    ConvexAlignHip handle per host thread, all launching concurrently.  Results identical to the serial oracle."""
    import re
    exe = os.path.join(ROOT, "ngmlr_amd", "batching_test")
    assert os.path.exists(exe), "batching_test not built"
    tiles = util.tile_zoo(seed=123, n=180, max_w=1800)
    pairs = [(t, port_oracle.align(t)) for t in tiles]
    rec = str(tmp_path / "tiles.bin")
    write_records(rec, pairs)
    res = subprocess.run([exe, rec, str(threads), mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                         env=dict(os.environ, **env))
    assert res.returncode == 0, res.stdout + res.stderr[-2000:]
    assert "0 mismatches" in res.stdout
    m = re.search(r"(\d+) logical devices in use", res.stdout)
    assert m and int(m.group(1)) == (2 if mode == "shared" else threads), res.stdout
