"""CPU: SAM record assembly (cvx_sam_record_text / cvx_sam_batch / cvx_sam_unmapped_text) against the
records the unmodified reference wrote (tests/golden/test_{2,3,4}.sam: 215 records with SA:Z lists, both strands, unmapped reads,
real PacBio quality strings), against hand-computed records for the branches its test data never takes (hard clipping,
read group, the 65 536-operation CG:B:I form, unmapped reads, a read written twice on the reverse strand), and -- when the
reference is present -- against the reference's own binary with only SAMWriter::DoWriteReadGeneric rebound
(oracle/_ref/ngmlr_sam, tools/build_ngmlr_hip.sh)."""
import ctypes as C
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

from ngmlr_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
E2E = os.path.join(GOLDEN, "e2e")


def _unmapped_text(lib, line):
    f = line.split("\t")
    assert f[2] == "*" and len(f) == 11
    u = capi.CvxSamUnmapped(read_name=f[0].encode(), seq=f[9].encode(), qual=None if f[10] == "*" else f[10].encode(), read_length=len(f[9]),
                            flags=int(f[1]) & ~4, ref_name=None, ref_name_len=0, location=int(f[3]) - 1, mate_ref=f[6].encode(),
                            mate_location=int(f[7]) - 1, template_length=int(f[8]), rg_id=None)
    out = C.create_string_buffer(len(line) + 64)
    n = C.c_uint64()
    assert lib.cvx_sam_unmapped_text(C.byref(u), out, len(line) + 64, C.byref(n)) == 0
    return out.raw[:n.value].decode()


def _golden_lines(mapped_only=True):
    lines = _all_golden_lines()
    return [l for l in lines if not int(l.split("\t")[1]) & 4] if mapped_only else lines


def _all_golden_lines():
    lines = []
    for name in ("test_2.sam", "test_4.sam"):
        lines += [l.rstrip("\n") for l in open(os.path.join(GOLDEN, name)) if l.strip() and not l.startswith("@")]
    with gzip.open(os.path.join(GOLDEN, "test_3.sorted.sam.gz"), "rt") as f:
        lines += [l.rstrip("\n") for l in f if l.strip() and not l.startswith("@")]
    return lines


class Rec:
    """cvx_sam_record + the Python objects that keep its pointers alive"""

    def __init__(self, line=None, **kw):
        self.keep = []
        self.r = capi.CvxSamRecord()
        if line is not None:
            self._from_sam(line)
        for k, v in kw.items():
            self.set(k, v)

    def set(self, k, v):
        if isinstance(v, str):
            v = v.encode()
        if k == "qual":
            if v is None:
                self.r.qual = None
                return
            buf = C.create_string_buffer(v)
            self.keep.append(buf)
            self.qual_buf = buf
            self.r.qual = C.addressof(buf)
            return
        if k == "others":
            arr = (capi.CvxSamOther * max(len(v), 1))()
            for i, o in enumerate(v):
                name, loc, rev, cig, mq, nm = o
                arr[i].ref_name = name.encode(); arr[i].ref_name_len = len(name); arr[i].location = loc
                arr[i].reverse = rev; arr[i].cigar = cig.encode(); arr[i].mq = mq; arr[i].nm = nm
            self.keep.append(arr)
            self.r.others = arr
            self.r.n_others = len(v)
            return
        if isinstance(v, bytes):
            self.keep.append(v)
        setattr(self.r, k, v)

    def _from_sam(self, line):
        """the writer's inputs, recovered from a record the reference wrote"""
        f = line.split("\t")
        tags = {t[:2]: t[5:] for t in f[11:]}
        flag = int(f[1])
        cigar = f[5]
        ops = [(int(n), c) for n, c in re.findall(r"(\d+)([MIDS])", cigar)]
        aln = sum(n for n, c in ops if c in "MID")
        nm = int(tags["NM"])
        L = len(f[9])
        self.set("read_name", f[0]); self.set("seq", f[9])
        rev = 1 if flag & 0x10 else 0
        # the function reverses the caller's quality string in place before printing a reverse-strand record
        self.set("qual", None if f[10] == "*" else (f[10][::-1] if rev else f[10]))
        self.set("read_length", L); self.set("flags", flag & ~(0x800 | 0x10)); self.set("primary", 0 if flag & 0x800 else 1)
        self.set("reverse", rev); self.set("ref_name", f[2]); self.set("ref_name_len", len(f[2]))
        self.set("location", int(f[3]) - 1); self.set("mq", int(f[4])); self.set("cigar", cigar); self.set("md", tags["MD"])
        self.set("cigar_op_count", len(ops)); self.set("mate_ref_name", f[6]); self.set("mate_location", int(f[7]) - 1)
        self.set("template_length", int(f[8])); self.set("score", float(int(tags["AS"]))); self.set("nm", nm)
        # Align::Identity = matches * 1.0f / alignmentLength (src/ConvexAlignFast.cpp:323), binary32
        self.set("identity", float(np.float32(aln - nm) / np.float32(aln)))
        self.set("qstart", int(tags["QS"])); self.set("qend", L - int(tags["QE"]))
        self.set("sv_type", int(tags["SV"]) if "SV" in tags else -1)
        others = []
        for e in tags.get("SA", "").split(";"):
            if e:
                name, pos, strand, cig, mq, onm = e.split(",")
                others.append((name, int(pos) - 1, 1 if strand == "-" else 0, cig, int(mq), int(onm)))
        self.set("others", others)
        self.set("rg_id", None)

    def text(self, lib, cap=1 << 22):
        out = C.create_string_buffer(cap)
        n = C.c_uint64()
        rc = lib.cvx_sam_record_text(C.byref(self.r), out, cap, C.byref(n))
        return rc, out.raw[:min(n.value, cap)].decode(), n.value


@pytest.fixture(scope="module")
def lib(built):
    return capi.load()


def test_records_equal_the_reference_writers_on_its_own_output(lib):
    lines = _golden_lines()
    assert len(lines) > 150
    n_sa = n_rev = 0
    for line in lines:
        rc, got, n = Rec(line).text(lib)
        assert rc == 0 and got == line + "\n", (line[:80], got[:80])
        n_sa += "\tSA:Z:" in line
        n_rev += bool(int(line.split("\t")[1]) & 0x10)
    assert n_sa > 20 and n_rev > 50
    unmapped = [l for l in _golden_lines(mapped_only=False) if int(l.split("\t")[1]) & 4]
    for line in unmapped:
        assert _unmapped_text(lib, line) == line + "\n"
    assert len(unmapped) >= 1


def test_batch_equals_one_by_one_including_shared_quality_strings(lib):
    lines = _golden_lines()
    recs = [Rec(l) for l in lines]
    # two reverse-strand records of one read share a quality buffer: the second is printed with the qualities turned back
    shared = [i for i, l in enumerate(lines) if int(l.split("\t")[1]) & 0x10][:2]
    a, b = recs[shared[0]], recs[shared[1]]
    L = min(a.r.read_length, b.r.read_length)
    b.r.qual = a.r.qual
    b.r.read_length = a.r.read_length = L
    one_by_one = []
    for r in [Rec(l) for l in lines]:
        one_by_one.append(r)
    one_by_one[shared[1]].r.qual = one_by_one[shared[0]].r.qual
    one_by_one[shared[1]].r.read_length = one_by_one[shared[0]].r.read_length = L
    want = [r.text(lib)[1] for r in one_by_one]
    arr = (capi.CvxSamRecord * len(recs))(*[r.r for r in recs])
    off = np.zeros(len(recs) + 1, dtype=np.uint64)
    assert lib.cvx_sam_batch(len(recs), arr, None, 0, off.ctypes.data) == -6          # CVX_ERR_CAPACITY: offsets[n] = the need
    need = int(off[len(recs)])
    assert need == sum(len(w) for w in want)
    before = C.string_at(a.r.qual, L)
    out = C.create_string_buffer(need)
    assert lib.cvx_sam_batch(len(recs), arr, out, need, off.ctypes.data) == 0
    got = [out.raw[int(off[i]):int(off[i + 1])].decode() for i in range(len(recs))]
    assert got == want
    assert got[shared[0]].split("\t")[10] == got[shared[1]].split("\t")[10][::-1]
    # reversed twice: the shared buffer is back where it started, exactly as after the reference's two in-place reversals
    assert C.string_at(a.r.qual, L) == before == C.string_at(one_by_one[shared[0]].r.qual, L)


def _base(**kw):
    d = dict(read_name="r1", seq="ACGTACGTAC", qual="0123456789", read_length=10, flags=0, primary=1, reverse=0,
             ref_name="chr1 extra", ref_name_len=4, location=99, mq=60, cigar="2S6M2S", md="6", cigar_op_count=3,
             mate_ref_name="*", mate_location=-1, template_length=0, score=12.9, nm=0, identity=1.0, qstart=2, qend=2,
             sv_type=-1, others=[], rg_id=None, hard_clip=0, bam_cigar_fix=0, skip=0)
    d.update(kw)
    return Rec(**d)


def test_branches_the_reference_test_data_never_takes(lib):
    # plain record: tags in the reference's order, (int) score, %g identity, %f coverage
    rc, got, _ = _base().text(lib)
    assert got == "r1\t0\tchr1\t100\t60\t2S6M2S\t*\t0\t0\tACGTACGTAC\t0123456789\tAS:i:12\tNM:i:0\tXI:f:1\tXS:i:0\tXE:i:12\tXR:i:6\tMD:Z:6\tQS:i:2\tQE:i:8\tCV:f:60.000000\n"
    # reverse strand, supplementary, read group, SV tag, hard clip (sequence and the REVERSED qualities clipped by QStart / QEnd)
    rc, got, _ = _base(reverse=1, primary=0, rg_id="grp", sv_type=1, hard_clip=1, identity=0.87654321, nm=3).text(lib)
    assert got == ("r1\t2064\tchr1\t100\t60\t2S6M2S\t*\t0\t0\tGTACGT\t765432\tRG:Z:grp\tAS:i:12\tNM:i:3\tXI:f:0.8765\tXS:i:0\tXE:i:12\tXR:i:6\t"
                   "MD:Z:6\tSV:i:1\tQS:i:2\tQE:i:8\tCV:f:60.000000\n")
    # no qualities; mate fields as DoWritePair passes them; SA list
    rc, got, _ = _base(qual=None, mate_ref_name="=", mate_location=499, template_length=-250, flags=0x1 | 0x40,
                       others=[("chr2", 9, 1, "10M", 3, 1), ("chrUn_x", 0, 0, "5M5S", 0, 0)]).text(lib)
    assert got == ("r1\t65\tchr1\t100\t60\t2S6M2S\t=\t500\t-250\tACGTACGTAC\t*\tAS:i:12\tNM:i:0\tXI:f:1\tXS:i:0\tXE:i:12\tXR:i:6\tMD:Z:6\t"
                   "SA:Z:chr2,10,-,10M,3,1;chrUn_x,1,+,5M5S,0,0;\tQS:i:2\tQE:i:8\tCV:f:60.000000\n")
    # 65 536 operations and more with the BAM fix: "<length>S" in the CIGAR column, the real one as CG:B:I words
    n_ops = 0x10000
    cigar = "1M1I" * (n_ops // 2)
    seq = "A" * n_ops
    r = _base(seq=seq, qual="I" * n_ops, read_length=n_ops, cigar=cigar, cigar_op_count=n_ops, bam_cigar_fix=1, qstart=0, qend=0, md="0")
    rc, got, n = r.text(lib)
    assert rc == 0 and n == len(got)
    f = got.rstrip("\n").split("\t")
    assert f[5] == "%dS" % n_ops and f[-1] == "CG:B:I" + ",16,17" * (n_ops // 2)
    # ... one operation fewer: the CIGAR stays where it is
    r = _base(seq=seq, qual="I" * n_ops, read_length=n_ops, cigar=cigar[:-2], cigar_op_count=n_ops - 1, bam_cigar_fix=1, qstart=0, qend=0)
    assert r.text(lib)[1].split("\t")[5] == cigar[:-2]
    # too small a buffer: the need comes back, the quality string is not reversed by the failed call
    r = _base(reverse=1)
    rc, _, need = r.text(lib, cap=16)
    assert rc == -6 and need > 100 and r.qual_buf.value == b"0123456789"
    rc, got, n = r.text(lib, cap=int(need))
    assert rc == 0 and n == need and r.qual_buf.value == b"9876543210" and got.split("\t")[10] == "9876543210"
    # the same read written again on the reverse strand: the reference reverses its buffer once more
    assert r.text(lib)[1].split("\t")[10] == "0123456789"
    # bad arguments
    bad = _base()
    bad.r.cigar = None
    assert bad.text(lib)[0] == -3


def test_unmapped_record(lib):
    u = capi.CvxSamUnmapped(read_name=b"r9", seq=b"ACGTN", qual=b"IIII#", read_length=5, flags=0, ref_name=None, ref_name_len=0,
                            location=-1, mate_ref=b"*", mate_location=-1, template_length=0, rg_id=None)
    out = C.create_string_buffer(256)
    n = C.c_uint64()
    assert lib.cvx_sam_unmapped_text(C.byref(u), out, 256, C.byref(n)) == 0
    assert out.raw[:n.value] == b"r9\t4\t*\t0\t0\t*\t*\t0\t0\tACGTN\tIIII#\n"
    u.qual = None
    u.rg_id = b"g"
    u.ref_name = b"chr3"
    u.ref_name_len = 4
    u.location = 41
    assert lib.cvx_sam_unmapped_text(C.byref(u), out, 256, C.byref(n)) == 0
    assert out.raw[:n.value] == b"r9\t4\tchr3\t42\t0\t*\t*\t0\t0\tACGTN\t*\tRG:Z:g\n"


def _records(text):
    return [l for l in text.splitlines() if l and not l.startswith("@")]


def test_reference_binary_with_the_writer_rebound(tmp_path):
    """The reference's own ngmlr, CPU aligners and all, with DoWriteReadGeneric going through cvx_sam_record_text
    (ngmlr_amd/csrc/sam_writer_binding.inc): the SAM must not change."""
    binary = os.path.join(ROOT, "oracle", "_ref", "ngmlr_sam")
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref/ngmlr_sam not built (tools/build_ngmlr_hip.sh needs /root/reference)")

    def run(args):
        res = subprocess.run([binary, "--skip-write"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                             timeout=900, cwd=str(tmp_path))
        assert res.returncode == 0, res.stderr[-2000:]
        return _records(res.stdout)

    got = run(["-t", "1", "-r", os.path.join(E2E, "ref_chr21_20kb.fa"), "-q", os.path.join(E2E, "reads_100_2200bp.fa")])
    assert got == _records(open(os.path.join(GOLDEN, "test_2.sam")).read()) and len(got) == 12
    fq = os.path.join(str(tmp_path), "test_3.fq")
    with gzip.open(os.path.join(E2E, "test_3_reads.fq.gz"), "rb") as f, open(fq, "wb") as o:
        o.write(f.read())
    got = run(["-x", "pacbio", "-t", "8", "-R", "0.01", "--no-progress", "-r", os.path.join(E2E, "test_3_reference.fasta.gz"), "-q", fq])
    with gzip.open(os.path.join(GOLDEN, "test_3.sorted.sam.gz"), "rt") as f:
        want = [l.rstrip("\n") for l in f if l.strip() and not l.startswith("@")]
    assert sorted(got) == want and len(want) > 200


def _restated_record(d, qual_reversed):
    """src/SAMWriter.cpp:87-224 restated with Python's printf (%d, %u, %g, %f are libc's): the fields of a record -> its text.
    Test infrastructure: the checker of the fuzz test below."""
    flags = d["flags"] | (0 if d["primary"] else 0x800) | (0x10 if d["reverse"] else 0)
    L = d["read_length"]
    clipped = L - d["qstart"] - d["qend"]
    out = ["%s\t%d\t%s\t%u\t%d\t" % (d["read_name"], flags, d["ref_name"][:d["ref_name_len"]], (d["location"] + 1) & 0xffffffff, d["mq"])]
    long_cigar = bool(d["bam_cigar_fix"]) and not d["skip"] and d["cigar_op_count"] >= 0x10000
    out.append(("%dS\t" % (clipped if d["hard_clip"] else L)) if long_cigar else d["cigar"] + "\t")
    out.append("%s\t%u\t%d\t" % (d["mate_ref_name"], (d["mate_location"] + 1) & 0xffffffff, d["template_length"]))
    seq = d["seq"]
    out.append((seq[d["qstart"]:d["qstart"] + max(clipped, 0)] if d["hard_clip"] else seq[:L]) + "\t")
    if d["qual"] is None:
        out.append("*\t")
    else:
        q = d["qual"][:L][::-1] + d["qual"][L:] if qual_reversed else d["qual"]
        out.append((q[d["qstart"]:d["qstart"] + max(clipped, 0)] if d["hard_clip"] else q[:L]) + "\t")
    if d["rg_id"] is not None:
        out.append("RG:Z:%s\t" % d["rg_id"])
    score = int(np.float32(d["score"]))                                   # (int) float: truncation
    ident = float(np.float32(np.round(np.float32(d["identity"]) * np.float32(10000.0)) / np.float32(10000.0)))
    out.append("AS:i:%d\tNM:i:%d\tXI:f:%g\tXS:i:0\tXE:i:%d\tXR:i:%d\tMD:Z:%s\t" % (score, d["nm"], ident, score, clipped, d["md"]))
    if d["sv_type"] > -1:
        out.append("SV:i:%d\t" % d["sv_type"])
    if d["others"]:
        out.append("SA:Z:" + "".join("%s,%d,%s,%s,%d,%d;" % (n, loc + 1, "-" if rev else "+", cig, mq, nm) for n, loc, rev, cig, mq, nm in d["others"]) + "\t")
    covered = float(np.float32(np.float32(L - d["qstart"] - d["qend"]) * np.float32(100.0)) / np.float32(L))
    out.append("QS:i:%d\tQE:i:%d\tCV:f:%f" % (d["qstart"], L - d["qend"], float(np.float32(covered))))
    if long_cigar:
        words = [(int(n) << 4) | "MIDNSH=X".index(c) if c in "MIDNSH" else (int(n) << 4) | {"=": 7, "X": 8}[c]
                 for n, c in re.findall(r"(\d+)([MIDNSH=X])", d["cigar"])][:d["cigar_op_count"]]
        out.append("\tCG:B:I" + "".join(",%d" % w for w in words))
    return "".join(out) + "\n"


def test_fuzzed_records_against_the_restated_writer(lib):
    rng = np.random.default_rng(2718)
    names = ["chr1", "2/83370031_83380361", "x", "scaffold_12 extra words"]
    n_checked = 0
    for it in range(600):
        L = int(rng.integers(1, 400))
        seq = "".join(rng.choice(list("ACGTN"), size=L))
        qs = int(rng.integers(0, L))
        qe = int(rng.integers(0, L - qs))
        has_q = rng.random() < 0.8
        qual = "".join(chr(int(c)) for c in rng.integers(33, 74, size=L)) if has_q else None
        rname = names[int(rng.integers(0, len(names)))]
        others = [(names[int(rng.integers(0, len(names)))].split(" ")[0], int(rng.integers(0, 2**31 - 2)), int(rng.integers(0, 2)),
                   "%dM%dS" % (int(rng.integers(1, 300)), int(rng.integers(1, 300))), int(rng.integers(0, 61)), int(rng.integers(0, 5000)))
                  for _ in range(int(rng.integers(0, 4)))]
        d = dict(read_name="read/%d" % it, seq=seq, qual=qual, read_length=L, flags=int(rng.choice([0, 1, 0x41, 0x81, 0x100])),
                 primary=int(rng.integers(0, 2)), reverse=int(rng.integers(0, 2)), ref_name=rname, ref_name_len=int(rng.integers(1, len(rname) + 1)),
                 location=int(rng.integers(0, 2**32 - 1)), mq=int(rng.integers(0, 61)), cigar="%dS%dM%dS" % (qs, L - qs - qe, qe) if qs and qe else "%dM" % L,
                 md=str(int(rng.integers(0, 500))), cigar_op_count=int(rng.integers(1, 4)), mate_ref_name=str(rng.choice(["*", "=", "chr9"])),
                 mate_location=int(rng.integers(-1, 2**31 - 2)), template_length=int(rng.integers(-5000, 5000)),
                 score=float(np.float32(rng.uniform(-10, 30000))), nm=int(rng.integers(0, 5000)), identity=float(np.float32(rng.uniform(0, 1))),
                 qstart=qs, qend=qe, sv_type=int(rng.integers(-1, 4)), others=others, rg_id=(None if rng.random() < 0.7 else "rg%d" % it),
                 hard_clip=int(rng.random() < 0.3), bam_cigar_fix=int(rng.random() < 0.5), skip=0)
        r = Rec(**d)
        rc, got, n = r.text(lib)
        want = _restated_record(d, qual_reversed=bool(d["reverse"]) and has_q)
        assert rc == 0 and got == want, (it, got[:200], want[:200])
        n_checked += 1
    assert n_checked == 600
