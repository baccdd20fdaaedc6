"""VCF / BED / FASTA readers of include/vcfdist_io.h against the Python restatement of the reference's record logic
(tests/demo_pipeline.py), on the demo files and on a hand-made VCF with the edge cases of variant.cpp:556-875."""
import gzip
import os

import numpy as np
import pytest

import demo_pipeline as D
from vcfdist_amd import api, io as IO


def columns(h):
    pool = h["pool"]
    refs = [bytes(pool[o:o + n]).decode() for o, n in zip(h["ref_off"], h["ref_len"])]
    alts = [bytes(pool[o:o + n]).decode() for o, n in zip(h["alt_off"], h["alt_len"])]
    return h["pos"].tolist(), h["rlen"].tolist(), h["type"].tolist(), refs, alts, h["var_qual"].tolist()


def same_as_python(path, bed_path):
    bed_py = D.Bed(bed_path) if bed_path else None
    bed_c = IO.Bed(bed_path) if bed_path else None

    class NoBed:
        def contains(self, *a): return D.BED_INSIDE
    py, st = D.parse_vcf(path, bed_py or NoBed())
    c = IO.read_vcf(path, bed_c)
    assert c["contigs"] == ["chr1"]
    for hp in range(2):
        pos, rlen, typ, refs, alts, q = columns(c["vars"][0][hp])
        P = py[hp]
        assert pos == P["pos"] and rlen == P["rlen"] and typ == P["type"] and refs == P["ref"] and alts == P["alt"]
        assert np.array_equal(np.asarray(q, np.float32), np.asarray(P["qual"], np.float32))
    return c, st


def test_exports():
    import ctypes as C
    L = C.CDLL(api.build())
    for name in IO.EXPORTED:
        assert hasattr(L, name), name


def test_demo_vcfs_match_python_restatement():
    bed = os.path.join(D.DEMO, "nist-v4.2.1_chr1_5Mb.bed")
    c, st = same_as_python(os.path.join(D.DEMO, "query.vcf"), bed)
    assert c["stats"]["n_records"] == 10430 and c["stats"]["n_unphased"] == st["unphased"] == 451
    assert c["stats"]["n_bed_outside"] == st["outside"] and c["stats"]["n_bed_border"] == st["border"]
    assert c["lengths"] == [248956422] and c["sample"] == "QUERY" and c["ploidy"] == [2]
    c, st = same_as_python(os.path.join(D.DEMO, "nist-v4.2.1_chr1_5Mb.vcf.gz"), bed)      # gzip input
    assert c["stats"]["n_records"] == 6676 and c["sample"] == "TRUTH"


def test_bed_contains_matches_python():
    bed_path = os.path.join(D.DEMO, "nist-v4.2.1_chr1_5Mb.bed")
    py, c = D.Bed(bed_path), IO.Bed(bed_path)
    rng = np.random.RandomState(0)
    m = {D.BED_INSIDE: 1, D.BED_BORDER: 2, D.BED_OUTSIDE: 0, D.BED_OFFCTG: 3}
    for _ in range(3000):
        s = int(rng.randint(590000, 5000000)); e = s + int(rng.choice([0, 1, 2, 50, 5000])); t = int(rng.choice([1, 2, 3]))
        assert c.contains("chr1", s, e, t) == m[py.contains("chr1", s, e, t)]
    assert c.contains("chrZ", 5, 6, 1) == 3


EDGE_VCF = """##fileformat=VCFv4.2
##contig=<ID=chr1,length=1000>
##FILTER=<ID=LowQual,Description="x">
##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">
##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="q">
##FORMAT=<ID=PS,Number=1,Type=Integer,Description="ps">
#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1
chr1\t10\t.\tA\tG\t30\tPASS\t.\tGT:GQ:PS\t0|1:20:7
chr1\t20\t.\tAC\tA\t.\tPASS\t.\tGT:GQ\t1|1:5
chr1\t30\t.\tA\tATT,AG\t99\tPASS\t.\tGT\t1|2
chr1\t40\t.\tACG\tTTT\t50\tPASS\t.\tGT\t1|0
chr1\t41\t.\tC\tT\t50\tPASS\t.\tGT\t1|0
chr1\t50\t.\tA\tG\t50\tPASS\t.\tGT\t0/1
chr1\t60\t.\tA\tG\t50\tPASS\t.\tGT\t1/1
chr1\t70\t.\tA\t*\t50\tPASS\t.\tGT\t1|1
chr1\t80\t.\tAGG\tAG\t50\tLowQual\t.\tGT\t1|1
chr1\t90\t.\tACGT\tATGT\t70.5\tPASS\t.\tGT\t.|1
chr1\t100\t.\tA\tA\t50\tPASS\t.\tGT\t1|1
"""


def test_edge_cases(tmp_path):
    p = tmp_path / "edge.vcf.gz"
    with gzip.open(p, "wt") as fh:
        fh.write(EDGE_VCF)
    c, st = same_as_python(str(p), None)
    h0, h1 = (columns(c["vars"][0][k]) for k in range(2))
    # hap 0: 1|1 DEL (trimmed), 1|2 -> INS "TT", CPX ACG>TTT -> INS + DEL (the SNP at 41 then overlaps), 1/1 SNP, LowQual kept
    # (no filter selected), multi-base SUB
    assert h0[0][:2] == [20, 30] and h0[2][:2] == [3, 2] and h0[4][1] == "TT"
    assert (39, 0, 2, "", "TTT") in list(zip(*h0[:5])) and (39, 3, 3, "ACG", "") in list(zip(*h0[:5]))
    assert c["stats"]["n_overlap"] == 1 and c["stats"]["n_unphased"] == 1 and c["stats"]["n_spanning_del"] == 2
    assert c["stats"]["n_ref_call"] == 2 and c["stats"]["n_unknown_allele"] == 1 and c["stats"]["n_complex"] == 2
    assert h1[0][0] == 9 and c["vars"][0][1]["phase_set"][0] == 7 and c["vars"][0][1]["gt_qual"][0] == 20
    # ACGT>ATGT: equal lengths, tails differ -> CPX (variant.cpp:783-790), i.e. INS of the whole ALT + DEL of the whole REF
    assert (89, 0, 2, "", "ATGT") in list(zip(*h1[:5])) and (89, 4, 3, "ACGT", "") in list(zip(*h1[:5]))
    assert max(h0[5]) == 60.0                                              # QUAL clamped to max_qual
    # with a FILTER selection the LowQual record is dropped
    c2 = IO.read_vcf(str(p), None, filters=("PASS",))
    assert c2["stats"]["n_failed_filter"] == 1


def test_fasta_reader(tmp_path):
    p = tmp_path / "x.fa"
    p.write_text(">chr1 description here\nacgtNN\nACGT\n>chr2\nTTTT\n")
    f = IO.read_fasta(str(p))
    assert bytes(f["chr1"]) == b"ACGTNNACGT" and bytes(f["chr2"]) == b"TTTT"


HDR = "##fileformat=VCFv4.2\n##contig=<ID=chr1,length=1000>\n##contig=<ID=chr2,length=1000>\n"
COLS = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1"


@pytest.mark.parametrize("name,text,needle", [
    # the reference's ERROR() cases (variant.cpp:397-1004): each ends the run there, an IOError with the message here
    ("two_samples", HDR + COLS + "\tS2\nchr1\t10\t.\tA\tG\t30\tPASS\t.\tGT\t1|1\t0|1\n", "1 sample"),
    ("unsorted_contigs", HDR + COLS + "\nchr1\t10\t.\tA\tG\t30\tPASS\t.\tGT\t1|1\nchr2\t10\t.\tA\tG\t30\tPASS\t.\tGT\t1|1\n"
                                      "chr1\t50\t.\tA\tG\t30\tPASS\t.\tGT\t1|1\n", "already parsed"),
    ("polyploid", HDR + COLS + "\nchr1\t10\t.\tA\tG\t30\tPASS\t.\tGT\t0|1|1\n", "ploidy 3"),
    ("record_before_header", HDR + "chr1\t10\t.\tA\tG\t30\tPASS\t.\tGT\t1|1\n", "before the #CHROM"),
    ("short_record", HDR + COLS + "\nchr1\t10\t.\tA\tG\t30\n", "fields"),
    ("allele_index", HDR + COLS + "\nchr1\t10\t.\tA\tG\t30\tPASS\t.\tGT\t1|2\n", "out of range"),
    # GT declared in the header but absent from a record (variant.cpp:637-640)
    ("missing_gt", HDR + '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n' + COLS +
                   "\nchr1\t10\t.\tA\tG\t30\tPASS\t.\tGQ\t20\n", "Failed to read GT"),
])
def test_vcf_format_errors_are_reported(tmp_path, name, text, needle):
    p = tmp_path / (name + ".vcf")
    p.write_text(text)
    with pytest.raises(IOError, match=needle):
        IO.read_vcf(str(p), None)


def test_record_without_gt_is_monoploid_when_the_header_declares_no_gt(tmp_path):
    p = tmp_path / "nogt.vcf"
    p.write_text(HDR + COLS + "\nchr1\t10\t.\tA\tG\t30\tPASS\t.\tGQ\t20\n")
    c = IO.read_vcf(str(p), None)                      # variant.cpp:631-636: a warning, ploidy 1
    assert sum(len(columns(c["vars"][0][k])[0]) for k in range(2)) >= 1


def test_missing_files_and_short_bed_lines(tmp_path):
    with pytest.raises(IOError):
        IO.read_vcf(str(tmp_path / "nope.vcf"), None)
    with pytest.raises(IOError):
        IO.read_fasta(str(tmp_path / "nope.fa"))
    with pytest.raises(IOError):
        IO.Bed(str(tmp_path / "nope.bed"))
    bad = tmp_path / "bad.bed"
    bad.write_text("chr1\t10\n")
    with pytest.raises(IOError, match="fewer than 3"):
        IO.Bed(str(bad))


@pytest.mark.parametrize("text,needle", [
    ("chr1\t0\t100\nchr1\t50\t200\n", "BED overlap detected: regions chr1:0-100 and chr1:50-200"),
    ("chr1\t300\t400\nchr1\t0\t100\n", "BED is unsorted; region chr1:300-400 precedes chr1:0-100"),
    ("chr1\t10\t10\n", "BED region chr1:10-10 length zero"),
    ("chr1\t20\t10\n", "BED region chr1:20-10 stop precedes start"),
])
def test_bed_check_refuses_what_the_reference_refuses(tmp_path, text, needle):
    """bedData::check (bed.cpp:36-71): the reference ERRORs on empty, unsorted and overlapping regions -- the containment
    test bisects the starts and stops, so silently accepting them would misclassify variants"""
    p = tmp_path / "regions.bed"
    p.write_text(text)
    with pytest.raises(IOError, match=needle):
        IO.Bed(str(p))


def test_bed_abutting_regions_only_warn(tmp_path, capfd):
    p = tmp_path / "abut.bed"
    p.write_text("chr1\t0\t100\nchr1\t100\t200\nchr2\t5\t9\n")
    b = IO.Bed(str(p))
    assert "BED regions chr1:0-100 and chr1:100-200 should be merged." in capfd.readouterr().err
    assert b.contains("chr1", 10, 20, 1) == 1      # VIO_BED_INSIDE
