"""CPU: ngmlr_amd/csrc/ngmlr_abi.h must be layout- and vtable-compatible with the
reference's src/IAlignment.h (the drop-in is compiled against the real header inside the
ngmlr tree; stand-alone it uses the mirror).  Needs /root/reference (this container)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"

PROBE = r'''
#include <cstdio>
#include <cstddef>
#include HEADER
struct Impl : IAlignment {
	int GetScoreBatchSize() const { return 11; }
	int GetAlignBatchSize() const { return 12; }
	int BatchScore(int const, int const, char const * const * const, char const * const * const, float * const, void *) { return 13; }
	int SingleAlign(int const, int const, char const * const, char const * const, Align &, void *) { return 14; }
	int SingleAlign(int const, CorridorLine *, int const, char const * const, char const * const, Align &, int const, int const, void *) { return 15; }
	int SingleScore(int const, int const, char const * const, char const * const, float &, void *) { return 16; }
	int BatchAlign(int const, int const, char const * const * const, char const * const * const, Align * const, void *) { return 17; }
};
#define O(T, f) printf(#T "." #f " %zu\n", offsetof(T, f))
int main() {
	printf("sizeof Align %zu PositionNM %zu CorridorLine %zu IAlignment %zu\n", sizeof(Align), sizeof(PositionNM), sizeof(CorridorLine), sizeof(IAlignment));
	O(CorridorLine, offset); O(CorridorLine, length); O(CorridorLine, offsetInMatrix);
	O(PositionNM, refPosition); O(PositionNM, readPosition); O(PositionNM, nm);
	O(Align, pBuffer1); O(Align, pBuffer2); O(Align, nmPerPosition); O(Align, mappedInterval);
	O(Align, firstPosition); O(Align, lastPosition); O(Align, nmPerPostionLength); O(Align, alignmentLength);
	O(Align, PositionOffset); O(Align, QStart); O(Align, QEnd); O(Align, Score); O(Align, Identity);
	O(Align, NM); O(Align, MQ); O(Align, cigarOpCount); O(Align, maxBufferLength); O(Align, maxMdBufferLength);
	O(Align, skip); O(Align, primary); O(Align, svType);
	Align a; printf("defaults %d %d %d\n", a.maxBufferLength, a.maxMdBufferLength, a.svType);
	/* vtable slot order: call through raw slots */
	Impl impl; IAlignment * p = &impl;
	typedef int (*fn0)(const IAlignment *);
	void ** vt = *(void ***) p;
	printf("slot0 %d slot1 %d\n", ((fn0) vt[0])(p), ((fn0) vt[1])(p));
	Align al; float fl = 0;
	printf("calls %d %d %d %d %d\n", p->BatchScore(0, 0, 0, 0, 0, 0), p->SingleAlign(0, 0, "", "", al, 0),
			p->SingleAlign(0, (CorridorLine *) 0, 0, "", "", al, 0, 0, 0), p->SingleScore(0, 0, "", "", fl, 0), p->BatchAlign(0, 0, 0, 0, 0, 0));
	return 0;
}
'''


def _run(header, incdir, tmp):
    src = os.path.join(tmp, "probe.cpp")
    open(src, "w").write(PROBE.replace("HEADER", '"%s"' % header))
    exe = os.path.join(tmp, "probe")
    subprocess.run(["g++", "-std=c++11", "-w", "-fno-strict-aliasing", "-I", incdir, "-o", exe, src], check=True)
    return subprocess.run([exe], stdout=subprocess.PIPE, text=True, check=True).stdout


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "IAlignment.h")), reason="reference tree not present")
def test_mirror_header_matches_reference_layout(tmp_path):
    a = tmp_path / "a"; b = tmp_path / "b"
    a.mkdir(); b.mkdir()
    mine = _run("ngmlr_abi.h", os.path.join(ROOT, "ngmlr_amd", "csrc"), str(a))
    theirs = _run("IAlignment.h", REF, str(b))
    assert mine == theirs
    assert "sizeof Align" in mine and "slot0 11 slot1 12" in mine
