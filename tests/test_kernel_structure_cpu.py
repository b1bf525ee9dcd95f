"""Guards on the compiled shape of the gfx950 kernels (no GPU needed: hipcc cross-compiles).

Round 1's fill kernel was persistent (an atomic tile cursor around the step loop) and hipcc
(ROCm 7.2) could turn that tile loop into a divergent loop that never terminated.  The kernel now
takes ONE tile per workgroup, so the only loops left are the two step-group loops (untracked and
exactly tracked phase); they must stay flat (depth 1) and wave-uniform (closed by a scalar
branch), otherwise a step would run under a partial EXEC mask.  A hang on the GPU box is
expensive, so the shape is pinned here.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ngmlr_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm") / "cvx_kernels.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-fPIC",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only",
           os.path.join(CSRC, "cvx_kernels.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out.read_text()


def _kernels(txt, prefix):
    for m in re.finditer(r"^(%s\w*):" % prefix, txt, re.M):
        body = txt[m.start():]
        yield m.group(1), body[:body.index(".end_amdhsa_kernel")]


def test_fill_kernels_have_flat_uniform_step_loops(device_asm):
    seen = 0
    for name, body in _kernels(device_asm, "_ZN3cvx16fill_ring_kernel"):
        depths = re.findall(r"Loop Header: Depth=(\d+)", body)
        # (a gang's wave spins -- bounded -- on its neighbour's record in front of the step's last slot: one wave-uniform loop inside each step)
        gang = re.search(r"ELi[23]EEEvNS_8FillArgsE$", name) is not None
        assert depths and set(depths) <= ({"1", "2"} if gang else {"1"}), "%s: loop levels %s" % (name, depths)
        lines = body.split("\n")
        labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
        step_loops = 0
        for i, l in enumerate(lines):
            m = re.search(r"(s_c?branch\w*) (\.LBB\d+_\d+)", l)
            if not m or m.group(2) not in labels or labels[m.group(2)] >= i:
                continue
            seg = lines[labels[m.group(2)]:i]
            # (a backward jump to a block that only ends the program -- the shared exit of an early return -- is not a loop)
            tgt = [q.strip() for q in lines[labels[m.group(2)] + 1:labels[m.group(2)] + 4] if q.strip() and not q.strip().startswith(";")]
            if tgt and tgt[0] == "s_endpgm":
                continue
            if sum(1 for q in seg if "v_addc_co_u32_e64" in q) >= 8:      # holds the plane updates of a group
                step_loops += 1
                # (vccz / vccnz test the whole VCC for zero -- a ballot -- and are as wave-uniform as scc; EXEC-based
                # branches are what a divergent loop would be closed by)
                assert m.group(1) in ("s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz"), \
                    "%s: step loop closed by %s (divergent?)" % (name, m.group(1))
        assert step_loops >= 1, name
        seen += 1
    # 4 ring classes x {float, int16 runs} x {two-phase, exact} + 4 x the two-phase float form with the penalty table (TAB) + 3 chain classes x {float, int16}
    # + gangs of 2 and 3 waves x {two-phase, two-phase with the table, exact}
    assert seen == 32


def test_register_budgets(device_asm):
    def vgprs(body):
        return int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
    fill = dict(_kernels(device_asm, "_ZN3cvx16fill_ring_kernelILi3ELb0E"))
    assert len(fill) == 9                                # {one wave, gangs of 2 and 3} x {two-phase arithmetic, two-phase with the LDS penalty table, exact}
    for name, body in fill.items():
        assert vgprs(body) <= 80, name                   # six waves per SIMD
    tab = dict(_kernels(device_asm, "_ZN3cvx16fill_ring_kernelILi3ELb0ELi0ELb1ELi1E"))
    assert len(tab) == 1 and vgprs(next(iter(tab.values()))) <= 72      # the PacBio launch: seven waves per SIMD
    # round 6: no fill kernel touches scratch.  Round 5's builds spilled 6-26 dwords per lane and reloaded them in the step loop's rare
    # paths (row staging, the direction flush): the tile's wave-uniform constants came out of vector loads and what was derived from
    # them -- 64-bit addresses -- lived in vector registers; they go through readfirstlane once now (in_sgpr, cvx_kernels.hip)
    meta = re.findall(r"\.name:\s+(_ZN3cvx16fill_ring_kernel\S+)\n\s+\.private_segment_fixed_size: (\d+)\n(?:.*\n){1,6}?\s+\.vgpr_spill_count: (\d+)", device_asm)
    assert len(meta) == 32
    for name, scratch, spilled in meta:
        gang_or_wide = re.search(r"ELi[23]EEEvNS_8FillArgsE$", name) or "ILi4E" in name
        assert (int(scratch), int(spilled)) == (0, 0) or gang_or_wide, (name, scratch, spilled)
    one_wave_m3 = [m_ for m_ in meta if "ILi3E" in m_[0] and m_[0].endswith("ELi1EEEvNS_8FillArgsE")]
    assert len(one_wave_m3) == 5 and all((int(sc_), int(sp_)) == (0, 0) for _, sc_, sp_ in one_wave_m3)
    walk = dict(_kernels(device_asm, "_ZN3cvx16backtrack_kernel"))
    assert len(walk) == 1
    assert vgprs(next(iter(walk.values()))) <= 32
