#!/usr/bin/env python3
"""Generates tools/ubench_pipes_bodies.inc for tools/ubench_pipes.hip: instruction streams with the
fill kernel's real operand forms (SGPR-mask v_cndmask, v_cmp -> SGPR pair, v_addc with an SGPR carry,
SDWA compare, SALU mask logic between them) in several orders, to find out what keeps a full-rate
VALU op from hiding under a half-rate one inside the kernel.

One "cell" below is the cell update of one ring slot as hipcc emits it for fill_ring_kernel<3,false,0>
(27 VALU + 16 SALU); a step is three cells on disjoint registers.  Orders:

  seq     slot after slot (what hipcc emits)
  rr      the three slots interleaved instruction by instruction
  alt     greedy list schedule that alternates half-rate and full-rate VALU ops whenever the
          dependences allow, SALU ops as early as possible
  noF     seq without the full-rate ops      noS   seq without the half-rate ops
  noSALU  seq without the scalar ops
"""
import sys

FULL = ("v_add_f32", "v_mul_f32", "v_add_u32", "v_mov_b32", "v_fma_f32", "v_and_b32")      # (classification used for the S= / F= counts printed beside each body; the new probes print as S)


def cell(k):
    """(opcode text, writes, reads) of one slot's cell update on its own registers"""
    vb = 16 + 20 * k
    names = "CW Q T DC DG HC UC MX SC R UI DRUN P E Z O ACCA ACCB CNT LEN".split()
    v = {n: "v%d" % (vb + i) for i, n in enumerate(names)}
    sb = 34 + 22 * k          # s32 / s33 are reserved; GAP reuses EU2's pair (dead by then)
    snames = "EU EG A B EU2 ND NIX NI CR ISDL ISIU".split()
    s = {n: "s[%d:%d]" % (sb + 2 * i, sb + 2 * i + 1) for i, n in enumerate(snames)}
    s["GAP"] = s["EU2"]
    MIS, MAT, DECAY, GEXT, GEM, GO = "v8", "v9", "v10", "v11", "v12", "v13"
    I = []

    def op(text, w, r):
        I.append((text, set(w), set(r)))

    op("v_cmp_eq_u32_sdwa vcc, %s, %s src0_sel:BYTE_0 src1_sel:DWORD" % (v["CW"], v["Q"]), ["vcc"], [v["CW"], v["Q"]])
    op("v_cndmask_b32_e32 %s, %s, %s, vcc" % (v["T"], MIS, MAT), [v["T"]], ["vcc"])
    op("v_add_f32_e32 %s, %s, %s" % (v["DC"], v["DG"], v["T"]), [v["DC"]], [v["DG"], v["T"]])
    op("v_max_f32_e32 %s, %s, %s" % (v["T"], v["HC"], v["DC"]), [v["T"]], [v["HC"], v["DC"]])
    op("v_max3_f32 %s, %s, %s, 0" % (v["MX"], v["T"], v["UC"]), [v["MX"]], [v["T"], v["UC"]])
    op("v_cmp_eq_f32_e64 %s, %s, %s" % (s["EU"], v["MX"], v["UC"]), [s["EU"]], [v["MX"], v["UC"]])
    op("v_cmp_eq_f32_e64 %s, %s, %s" % (s["EG"], v["MX"], v["DC"]), [s["EG"]], [v["MX"], v["DC"]])
    op("s_and_b64 %s, %s, %s" % (s["A"], s["EU"], s["ISIU"]), [s["A"], "scc"], [s["EU"], s["ISIU"]])
    op("s_or_b64 %s, %s, %s" % (s["A"], s["A"], s["EG"]), [s["A"], "scc"], [s["A"], s["EG"]])
    op("v_cmp_eq_f32_e32 vcc, %s, %s" % (v["MX"], v["HC"]), ["vcc"], [v["MX"], v["HC"]])
    op("s_orn2_b64 %s, %s, %s" % (s["A"], s["ISDL"], s["A"]), [s["A"], "scc"], [s["ISDL"], s["A"]])
    op("s_orn2_b64 %s, %s, %s" % (s["B"], s["ISIU"], s["EG"]), [s["B"], "scc"], [s["ISIU"], s["EG"]])
    op("s_and_b64 %s, vcc, %s" % (s["A"], s["A"]), [s["A"], "scc"], ["vcc", s["A"]])
    op("v_cmp_lt_u32_e32 vcc, %s, %s" % (v["CNT"], v["LEN"]), ["vcc"], [v["CNT"], v["LEN"]])
    op("s_and_b64 %s, %s, %s" % (s["EU2"], s["B"], s["EU"]), [s["EU2"], "scc"], [s["B"], s["EU"]])
    op("s_and_b64 %s, %s, vcc" % (s["ND"], s["A"]), [s["ND"], "scc"], [s["A"], "vcc"])
    op("s_and_b64 %s, %s, vcc" % (s["NIX"], s["EU2"]), [s["NIX"], "scc"], [s["EU2"], "vcc"])
    op("s_andn2_b64 %s, %s, %s" % (s["NI"], s["NIX"], s["ND"]), [s["NI"], "scc"], [s["NIX"], s["ND"]])
    op("v_cndmask_b32_e32 %s, 0, %s, vcc" % (v["SC"], v["MX"]), [v["SC"]], [v["MX"], "vcc"])
    op("s_and_b64 vcc, %s, %s" % (s["NI"], s["ISIU"]), ["vcc", "scc"], [s["NI"], s["ISIU"]])
    op("v_cndmask_b32_e32 %s, 1.0, %s, vcc" % (v["R"], v["UI"]), [v["R"]], [v["UI"], "vcc"])
    op("s_and_b64 vcc, %s, %s" % (s["ND"], s["ISDL"]), ["vcc", "scc"], [s["ND"], s["ISDL"]])
    op("v_cndmask_b32_e32 %s, %s, %s, vcc" % (v["R"], v["R"], v["DRUN"]), [v["R"]], [v["R"], v["DRUN"], "vcc"])
    op("v_mul_f32_e32 %s, %s, %s" % (v["P"], DECAY, v["R"]), [v["P"]], [v["R"]])
    op("v_add_f32_e32 %s, %s, %s" % (v["P"], GEXT, v["P"]), [v["P"]], [v["P"]])
    op("v_min_f32_e32 %s, %s, %s" % (v["P"], GEM, v["P"]), [v["P"]], [v["P"]])
    op("s_or_b64 %s, %s, %s" % (s["GAP"], s["ND"], s["NIX"]), [s["GAP"], "scc"], [s["ND"], s["NIX"]])
    op("s_or_b64 %s, %s, %s" % (s["CR"], s["NIX"], s["EG"]), [s["CR"], "scc"], [s["NIX"], s["EG"]])
    op("v_add_f32_e32 %s, %s, %s" % (v["E"], v["SC"], v["P"]), [v["E"]], [v["SC"], v["P"]])
    op("v_mul_f32_e32 %s, 0xf1800000, %s" % (v["Z"], v["SC"]), [v["Z"]], [v["SC"]])
    op("s_andn2_b64 %s, %s, %s" % (s["CR"], s["CR"], s["ND"]), [s["CR"], "scc"], [s["CR"], s["ND"]])
    op("v_addc_co_u32_e64 %s, vcc, %s, %s, %s" % (v["ACCA"], v["ACCA"], v["ACCA"], s["GAP"]), [v["ACCA"], "vcc"], [v["ACCA"], s["GAP"]])
    op("v_add_f32_e32 %s, %s, %s" % (v["O"], GO, v["SC"]), [v["O"]], [v["SC"]])
    op("v_addc_co_u32_e64 %s, vcc, %s, %s, %s" % (v["ACCB"], v["ACCB"], v["ACCB"], s["CR"]), [v["ACCB"], "vcc"], [v["ACCB"], s["CR"]])
    op("v_max_f32_e32 %s, %s, %s" % (v["E"], v["E"], v["Z"]), [v["E"]], [v["E"], v["Z"]])
    # new slot state: the next iteration's inputs
    op("v_cndmask_b32_e64 %s, %s, %s, %s" % (v["UC"], v["O"], v["E"], s["NI"]), [v["UC"]], [v["O"], v["E"], s["NI"]])
    op("v_cndmask_b32_e64 %s, %s, %s, %s" % (v["HC"], v["O"], v["E"], s["ND"]), [v["HC"]], [v["O"], v["E"], s["ND"]])
    op("v_add_f32_e32 %s, 1.0, %s" % (v["DRUN"], v["R"]), [v["DRUN"]], [v["R"]])
    op("v_add_u32_e32 %s, 1, %s" % (v["CNT"], v["CNT"]), [v["CNT"]], [v["CNT"]])
    op("v_mov_b32_dpp %s, %s wave_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (v["DG"], v["SC"]), [v["DG"]], [v["SC"]])
    op("s_mov_b64 %s, %s" % (s["ISDL"], s["ND"]), [s["ISDL"]], [s["ND"]])
    op("s_mov_b64 %s, %s" % (s["ISIU"], s["NI"]), [s["ISIU"]], [s["NI"]])
    return I


def klass(text):
    o = text.split()[0]
    if o.startswith("s_"):
        return "SALU"
    if o.startswith(FULL) and "dpp" not in o and "sdwa" not in o:
        return "F"
    return "S"


def deps(instrs):
    """dependence lists (RAW, WAR, WAW) in program order"""
    n = len(instrs)
    pred = [set() for _ in range(n)]
    for i in range(n):
        _, wi, ri = instrs[i]
        for j in range(i):
            _, wj, rj = instrs[j]
            if (wj & ri) or (wi & rj) or (wi & wj):
                pred[i].add(j)
    return pred


def schedule_alt(instrs):
    pred = deps(instrs)
    n = len(instrs)
    done, order = set(), []
    last = "F"
    while len(order) < n:
        ready = [i for i in range(n) if i not in done and pred[i] <= done]
        salu = [i for i in ready if klass(instrs[i][0]) == "SALU"]
        if salu:
            pick = salu[0]
        else:
            want = "S" if last == "F" else "F"
            c = [i for i in ready if klass(instrs[i][0]) == want]
            pick = c[0] if c else ready[0]
            last = klass(instrs[pick][0])
        done.add(pick)
        order.append(pick)
    return [instrs[i] for i in order]


def emit(name, instrs, out):
    nS = sum(1 for t, _, _ in instrs if klass(t) == "S")
    nF = sum(1 for t, _, _ in instrs if klass(t) == "F")
    nX = sum(1 for t, _, _ in instrs if klass(t) == "SALU")
    out.write("/* %s: %d half-rate, %d full-rate VALU, %d SALU */\n" % (name, nS, nF, nX))
    out.write("#define BODY_%s \\\n" % name)
    for t, _, _ in instrs:
        out.write('\t"%s\\n" \\\n' % t)
    out.write("\n#define COUNT_%s %d, %d, %d\n\n" % (name, nS, nF, nX))


def simple(text_fn, n):
    """n independent instances of an op on rotating registers"""
    return [(text_fn(i), set(), set()) for i in range(n)]


def main():
    cells = [cell(k) for k in range(3)]
    seq = cells[2] + cells[1] + cells[0]
    rr = []
    for i in range(len(cells[0])):
        for k in (2, 1, 0):
            rr.append(cells[k][i])
    kinds = []      # (name, instrs)
    F = lambda i: "v_add_f32 v%d, v%d, v8" % (16 + i % 24, 16 + i % 24)
    sg = lambda i: "s[%d:%d]" % (34 + 2 * (i % 8), 35 + 2 * (i % 8))
    skinds = {
        "v_max_f32": lambda i: "v_max_f32 v%d, v%d, v9" % (16 + i % 24, 16 + i % 24),
        "v_max3_f32": lambda i: "v_max3_f32 v%d, v%d, v%d, 0" % (16 + i % 24, 16 + i % 24, 40 + i % 8),
        "v_cmp_eq_f32 -> sgpr": lambda i: "v_cmp_eq_f32_e64 %s, v%d, v%d" % (sg(i), 16 + i % 24, 40 + i % 8),
        "v_cmp_eq_f32 -> vcc": lambda i: "v_cmp_eq_f32_e32 vcc, v%d, v%d" % (16 + i % 24, 40 + i % 8),
        "v_cmp_eq_u32_sdwa -> vcc": lambda i: "v_cmp_eq_u32_sdwa vcc, v%d, v%d src0_sel:BYTE_%d src1_sel:DWORD" % (16 + i % 24, 40 + i % 8, i % 4),
        "v_cndmask <- sgpr": lambda i: "v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (16 + i % 24, 16 + i % 24, 40 + i % 8, sg(i)),
        "v_cndmask <- vcc": lambda i: "v_cndmask_b32_e32 v%d, v%d, v%d, vcc" % (16 + i % 24, 16 + i % 24, 40 + i % 8),
        "v_addc sgpr carry -> vcc": lambda i: "v_addc_co_u32_e64 v%d, vcc, v%d, v%d, %s" % (16 + i % 24, 16 + i % 24, 16 + i % 24, sg(i)),
        "v_addc sgpr carry -> sgpr": lambda i: "v_addc_co_u32_e64 v%d, %s, v%d, v%d, %s" % (16 + i % 24, sg(i + 4), 16 + i % 24, 16 + i % 24, sg(i)),
        "v_addc vcc -> vcc (e32)": lambda i: "v_addc_co_u32_e32 v%d, vcc, v%d, v%d, vcc" % (16 + i % 24, 16 + i % 24, 16 + i % 24),
        "v_lshl_or_b32": lambda i: "v_lshl_or_b32 v%d, v%d, 1, v%d" % (16 + i % 24, 16 + i % 24, 40 + i % 8),
        "v_mov_dpp wave_ror:1": lambda i: "v_mov_b32_dpp v%d, v%d wave_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (16 + i % 24, 40 + i % 8),
        "v_mov_dpp row_shr:1": lambda i: "v_mov_b32_dpp v%d, v%d row_shr:1 row_mask:0xf bank_mask:0xf" % (16 + i % 24, 40 + i % 8),
        "v_add_f32_dpp wave_ror:1": lambda i: "v_add_f32_dpp v%d, v%d, v%d wave_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (16 + i % 24, 40 + i % 8, 16 + i % 24),
        "v_mul_f32": lambda i: "v_mul_f32 v%d, v%d, v9" % (16 + i % 24, 16 + i % 24),
        "v_add_u32": lambda i: "v_add_u32 v%d, 1, v%d" % (16 + i % 24, 16 + i % 24),
        "v_mov_b32": lambda i: "v_mov_b32 v%d, v%d" % (16 + i % 24, 40 + i % 8),
        "v_fma_f32": lambda i: "v_fma_f32 v%d, v%d, v8, v9" % (16 + i % 24, 16 + i % 24),
        "v_cvt_f32_ubyte0": lambda i: "v_cvt_f32_ubyte%d v%d, v%d" % (i % 4, 16 + i % 24, 40 + i % 8),
        "v_and_b32": lambda i: "v_and_b32 v%d, v%d, v%d" % (16 + i % 24, 16 + i % 24, 40 + i % 8),
        "v_min_f32": lambda i: "v_min_f32 v%d, v%d, v9" % (16 + i % 24, 16 + i % 24),
        "v_med3_f32": lambda i: "v_med3_f32 v%d, v%d, v8, v9" % (16 + i % 24, 16 + i % 24),
        "v_readlane": lambda i: "v_readlane_b32 s%d, v%d, 5" % (34 + i % 16, 16 + i % 24),
        # round 3, second batch: which integer forms are full-rate (candidates for replacing float min / max / compares)
        "v_max_i32": lambda i: "v_max_i32 v%d, v%d, v9" % (16 + i % 24, 16 + i % 24),
        "v_max_u32": lambda i: "v_max_u32 v%d, v%d, v9" % (16 + i % 24, 16 + i % 24),
        "v_min_u32": lambda i: "v_min_u32 v%d, v%d, v9" % (16 + i % 24, 16 + i % 24),
        "v_max3_i32": lambda i: "v_max3_i32 v%d, v%d, v%d, 0" % (16 + i % 24, 16 + i % 24, 40 + i % 8),
        "v_sub_co_u32 -> sgpr": lambda i: "v_sub_co_u32_e64 v%d, %s, v%d, v%d" % (16 + i % 24, sg(i), 16 + i % 24, 40 + i % 8),
        "v_add_co_u32 -> vcc (e32)": lambda i: "v_add_co_u32_e32 v%d, vcc, v%d, v%d" % (16 + i % 24, 16 + i % 24, 40 + i % 8),
        "v_cmp_lt_u32 -> sgpr": lambda i: "v_cmp_lt_u32_e64 %s, v%d, v%d" % (sg(i), 16 + i % 24, 40 + i % 8),
        "v_cmp_eq_u32 -> sgpr": lambda i: "v_cmp_eq_u32_e64 %s, v%d, v%d" % (sg(i), 16 + i % 24, 40 + i % 8),
        "v_pk_add_f32": lambda i: "v_pk_add_f32 v[%d:%d], v[%d:%d], v[8:9]" % (16 + 2 * (i % 12), 17 + 2 * (i % 12), 16 + 2 * (i % 12), 17 + 2 * (i % 12)),
        "v_pk_mul_f32": lambda i: "v_pk_mul_f32 v[%d:%d], v[%d:%d], v[8:9]" % (16 + 2 * (i % 12), 17 + 2 * (i % 12), 16 + 2 * (i % 12), 17 + 2 * (i % 12)),
        "v_pk_fma_f32": lambda i: "v_pk_fma_f32 v[%d:%d], v[%d:%d], v[8:9], v[10:11]" % (16 + 2 * (i % 12), 17 + 2 * (i % 12), 16 + 2 * (i % 12), 17 + 2 * (i % 12)),
        "v_bfe_u32": lambda i: "v_bfe_u32 v%d, v%d, 8, 8" % (16 + i % 24, 40 + i % 8),
        "v_xor_b32": lambda i: "v_xor_b32 v%d, v%d, v%d" % (16 + i % 24, 16 + i % 24, 40 + i % 8),
        "v_lshlrev_b32": lambda i: "v_lshlrev_b32 v%d, 1, v%d" % (16 + i % 24, 16 + i % 24),
        "v_lshl_add_u32": lambda i: "v_lshl_add_u32 v%d, v%d, 1, v%d" % (16 + i % 24, 16 + i % 24, 40 + i % 8),
        "v_add3_u32": lambda i: "v_add3_u32 v%d, v%d, v%d, 1" % (16 + i % 24, 16 + i % 24, 40 + i % 8),
        "v_and_or_b32": lambda i: "v_and_or_b32 v%d, v%d, v%d, v9" % (16 + i % 24, 16 + i % 24, 40 + i % 8),
        "v_bfi_b32": lambda i: "v_bfi_b32 v%d, v8, v%d, v%d" % (16 + i % 24, 16 + i % 24, 40 + i % 8),
        "v_perm_b32": lambda i: "v_perm_b32 v%d, v%d, v%d, v9" % (16 + i % 24, 16 + i % 24, 40 + i % 8),
        "v_sub_f32": lambda i: "v_sub_f32 v%d, v%d, v9" % (16 + i % 24, 16 + i % 24),
        "v_cvt_f32_u32": lambda i: "v_cvt_f32_u32 v%d, v%d" % (16 + i % 24, 40 + i % 8),
        "v_cvt_f32_i32": lambda i: "v_cvt_f32_i32 v%d, v%d" % (16 + i % 24, 40 + i % 8),
        "v_max_i16": lambda i: "v_max_i16 v%d, v%d, v9" % (16 + i % 24, 16 + i % 24),
        "v_pk_max_i16": lambda i: "v_pk_max_i16 v%d, v%d, v9" % (16 + i % 24, 16 + i % 24),
        "v_pk_add_u16": lambda i: "v_pk_add_u16 v%d, v%d, v9" % (16 + i % 24, 16 + i % 24),
        "v_cndmask sdwa?skip": None,
        "s_and_b64": lambda i: "s_and_b64 %s, %s, %s" % (sg(i), sg(i), sg(i + 3)),
        "s_lshl_b64": lambda i: "s_lshl_b64 %s, %s, 1" % (sg(i), sg(i)),
        "s_mov_b64": lambda i: "s_mov_b64 %s, %s" % (sg(i), sg(i + 3)),
        "s_and_b32": lambda i: "s_and_b32 s%d, s%d, s%d" % (34 + i % 16, 34 + i % 16, 52 + i % 8),
        "s_nop 0": lambda i: "s_nop 0",
    }
    kinds.append(("F x24 (v_add_f32)", [(F(i), set(), set()) for i in range(24)]))
    for nm, fn in skinds.items():
        if fn is None:
            continue
        kinds.append(("%s x24" % nm, simple(fn, 24)))
        alt = []
        for i in range(12):
            alt.append((fn(i), set(), set()))
            alt.append((F(i + 12), set(), set()))
        kinds.append(("%s | v_add_f32 alternating x12" % nm, alt))
    for nm in ("v_max_f32", "v_cndmask <- sgpr", "v_cmp_eq_f32 -> sgpr"):
        fn = skinds[nm]
        kinds.append(("%s | s_and_b64 alternating x12" % nm, sum(([(fn(i), set(), set()), (skinds["s_and_b64"](i), set(), set())] for i in range(12)), [])))
        kinds.append(("%s, v_add_f32, s_and_b64 x8" % nm, sum(([(fn(i), set(), set()), (F(i + 12), set(), set()), (skinds["s_and_b64"](i), set(), set())] for i in range(8)), [])))
    kinds.append(("cell x3: compiler order", seq))
    kinds.append(("cell x3: slots interleaved", rr))
    kinds.append(("cell x3: S/F alternated (list schedule)", schedule_alt(seq)))
    kinds.append(("cell x3: without full-rate ops", [x for x in seq if klass(x[0]) != "F"]))
    kinds.append(("cell x3: without half-rate ops", [x for x in seq if klass(x[0]) != "S"]))
    kinds.append(("cell x3: without SALU", [x for x in seq if klass(x[0]) != "SALU"]))
    kinds.append(("cell x3: without addc and dpp", [x for x in seq if "addc" not in x[0] and "dpp" not in x[0]]))
    kinds.append(("cell x3: VALU only, without addc and dpp", [x for x in seq if "addc" not in x[0] and "dpp" not in x[0] and klass(x[0]) != "SALU"]))
    # marginal cost of every VALU instruction of the cell in its context: the first k VALU instructions of each slot
    valu = [x for x in cells[0] if klass(x[0]) != "SALU"]
    for k in range(1, len(valu) + 1):
        pre = []
        for c in (2, 1, 0):
            pre += [x for x in cells[c] if klass(x[0]) != "SALU"][:k]
        kinds.append(("prefix %2d: + %s" % (k, valu[k - 1][0].split()[0]), pre))
    with open(sys.argv[1] if len(sys.argv) > 1 else "tools/ubench_pipes_bodies.inc", "w") as out:
        out.write("/* generated by tools/gen_ubench_pipes.py -- do not edit */\n\n")
        for i, (nm, ins) in enumerate(kinds):
            nS = sum(1 for t, _, _ in ins if klass(t) == "S")
            nF = sum(1 for t, _, _ in ins if klass(t) == "F")
            nX = sum(1 for t, _, _ in ins if klass(t) == "SALU")
            out.write("KIND(%d, \"%s\",\n" % (i, nm))
            for t, _, _ in ins:
                out.write('\t"%s\\n"\n' % t)
            out.write("\t, %d, %d, %d)\n" % (nS, nF, nX))
        out.write("static const int kKinds = %d;\n" % len(kinds))


if __name__ == "__main__":
    main()
