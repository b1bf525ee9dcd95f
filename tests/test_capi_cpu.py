"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/*.h
declares, and refuses to compute without a device (no silent fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "cvx_align.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cvx_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(built):
    from ngmlr_amd import capi
    lib = capi.load()
    names = declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(capi.EXPORTS)
    assert lib.cvx_abi_version() == 9


def test_library_carries_gfx950_code_object(built):
    from ngmlr_amd import capi
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", capi.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert ".hip_fatbin" in out
    strs = subprocess.run(["strings", "-n", "6", capi.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "gfx950" in strs
    assert "fill_ring_kernel" in strs


def test_product_does_not_link_the_oracle(built):
    from ngmlr_amd import capi
    out = subprocess.run(["ldd", capi.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out
    nm = subprocess.run(["nm", "-D", capi.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle_align" not in nm
    for root, _, files in os.walk(os.path.join(ROOT, "ngmlr_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert not re.search(r"#include\s*[<\"][^>\"]*oracle", txt), f
                assert "dlopen" not in txt and "libcvx_oracle" not in txt, f


def test_no_device_is_a_loud_error(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ngmlr_amd import capi
    from ngmlr_amd.aligner import ConvexAlignHip
    with pytest.raises(capi.CvxError) as e:
        ConvexAlignHip()
    assert e.value.code == -1 and "no CPU fallback" in str(e.value)


def test_any_finite_scoring_is_accepted(built):
    """The reference runs whatever --match/--mismatch/--gap-* it is given.  Scoring outside the regime
    in which its SSE path equals the scalar recurrence (gap_open + gap_ext_min >= mismatch, SURVEY
    Appendix A) is no longer refused: those handles route every tile to the catch-all kernel's
    SSE-variant instantiation (GPU parity: tests/test_gpu_parity.py::test_sse_variant_scoring).  Without
    a device the only possible answers are "no device" and, for non-finite values, "bad parameters"."""
    import torch
    from ngmlr_amd import capi
    lib = capi.load()
    h = C.c_void_p()
    for odd in [(2, -10, -5, -5, -1, 0.15), (2, -5, -5, -5, 1, 0.15), (-2, -5, -5, -5, -1, 0.15), (2, -5, -5, -1, -5, 0.15)]:
        p = capi.CvxParams(*odd)
        rc = lib.cvx_create(0, C.byref(p), 0, C.byref(h))
        if torch.cuda.is_available():
            assert rc == 0, odd
            lib.cvx_destroy(h)
        else:
            assert rc == -1, odd           # CVX_ERR_NO_DEVICE, not CVX_ERR_PARAMS
    p = capi.CvxParams(2, float("nan"), -5, -5, -1, 0.15)
    assert lib.cvx_create(0, C.byref(p), 0, C.byref(h)) == -2
    assert b"not finite" in lib.cvx_last_error()


def test_bench_fails_loudly_when_devices_are_missing(built):
    """`python bench.py --gpus N` launched directly must drive N devices or refuse before doing any work
    (it used to report n_gpus 1 whatever N was)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert res.returncode == 2
    assert "needs 2 visible MI355X device(s)" in res.stderr and res.stdout.strip() == ""


def test_runtime_regime_without_a_device(built):
    """cvx_runtime_regime answers without a device: what the library did to the process's environment when it was loaded."""
    from ngmlr_amd import capi
    lib = capi.load()
    r = capi.CvxRegime()
    assert lib.cvx_runtime_regime(0, C.byref(r)) == 0
    assert r.hw_queues_env >= 1 and r.blocking_sync == -1 and r.service_streams == 4 and r.runtime_up_at_load == 0
    assert r.hw_queues_set_by_library == (0 if os.environ.get("CVX_TEST_HWQ_PRESET") else 1) or r.hw_queues_env != 16
    assert lib.cvx_runtime_regime(0, None) != 0


def test_source_ids_and_the_committed_counters(built):
    """cvx_source_id: twelve hex digits per kernel family, the whole-library id for anything else.  bench.py takes HBM traffic and
    bytes per vote from a profiles/r*_pmc.json only while the id of that kernel family is the one the counters were collected on
    (null otherwise, never a stale number); when the newest committed file belongs to the kernels of this tree, say so -- when it
    does not, that is a fact about the profiles, not a failure of the library."""
    import glob
    import json
    from ngmlr_amd import capi
    lib = capi.load()
    ids = {f: lib.cvx_source_id(f.encode()).decode() for f in ("fill", "search", "anything else")}
    assert all(re.fullmatch(r"[0-9a-f]{12}", v) for v in ids.values()), ids
    assert ids["anything else"] == lib.cvx_build_id().decode() and len(set(ids.values())) == 3
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc.json")), reverse=True)
    assert files, "no counters committed"
    pm = json.load(open(files[0]))
    src = pm.get("source_ids", {})
    if pm.get("build_id") != ids["anything else"] and (src.get("fill") != ids["fill"] or src.get("search") != ids["search"]):
        pytest.skip("%s was collected on other fill / search kernels (%s) than this tree's (%s): bench.py reports null traffic until the passes are repeated"
                    % (os.path.basename(files[0]), src or pm.get("build_id"), ids))
