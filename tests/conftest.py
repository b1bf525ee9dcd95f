import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """In-tree build of libcvxalign.so + the oracle's C restatement (+ oracle/_ref when the
    reference tree is present).  hipcc cross-compiles gfx950 without a GPU."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def port_oracle(built):
    from oracle.pyoracle import Oracle
    return Oracle("port")


@pytest.fixture(scope="session")
def ref_oracle(built):
    from oracle.pyoracle import Oracle, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this box and no prebuilt .so)")
    return Oracle("reference")


@pytest.fixture(scope="session")
def hip_aligner(built):
    from ngmlr_amd.aligner import ConvexAlignHip
    al = ConvexAlignHip(device=0)   # no fallback: raises without the library or the GPU
    yield al
    al.close()
